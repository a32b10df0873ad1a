"""Worker of tests/test_gpu_dist2.py: one of several ranks that SHARE ONE GPU (torch.distributed over gloo --
RCCL refuses two ranks on a device), launched through ``python -m torch.distributed.run``.  Exercises the
multi-rank code paths of the product with the real HIP engine: DistributedCholesky / exact_gp_posterior at P = 2,
reconstruct_slices, gather_to_root with device tensors, the sharded acquisition ranking.
Every rank checks its results; rank 0 prints "DIST2 OK" when all checks of all ranks passed."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gpim_amd                                                           # noqa: E402
from gpim_amd import _lib, dist as gdist                                   # noqa: E402
from gpim_amd.dist_chol import DistributedCholesky, exact_gp_posterior     # noqa: E402
from oracle import gpim_oracle as O                                        # noqa: E402
from problems import hyperspectral_cube                                    # noqa: E402


def main():
    rank, world, local_rank = gdist.init_from_env()
    assert world >= 2 and dist.get_backend() == "gloo", (world, dist.get_backend())
    dev = torch.device("cuda", local_rank)
    ok = True
    msgs = []

    def check(name, cond, detail=""):
        nonlocal ok
        if not cond:
            ok = False
            msgs.append("rank %d: %s FAILED %s" % (rank, name, detail))

    # 1. factorisation at P = world against gpimhip_potrf and torch (three sizes incl. a ragged last panel)
    for n in (700, 1500, 2600):
        rng = np.random.default_rng(n)
        Bm = rng.standard_normal((n, n // 3))
        Ah = torch.from_numpy(Bm @ Bm.T + n * np.eye(n))
        A = Ah.to(dev)
        ch = DistributedCholesky(n)
        ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
        Ld = ch.gather_lower()
        # reference on the host (torch.linalg.cholesky on the device goes through the vendor solver library, which
        # returned a wrong factor once while the other rank was busy on the same GPU)
        ref = torch.linalg.cholesky(Ah).to(dev)
        scale = ref.abs().max().item()
        check("factor n=%d vs torch" % n, (Ld - ref).abs().max().item() <= 1e-11 * scale,
              "%.2e" % ((Ld - ref).abs().max().item() / scale))
        H = _lib.Handle()
        Lp = A.clone()
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Lp), n, n, _lib.ptr(info)))
        check("factor n=%d vs gpimhip_potrf" % n, (Ld - torch.tril(Lp)).abs().max().item() <= 1e-11 * scale)
        ld_ref = 2 * torch.log(torch.diagonal(ref)).sum().item()
        check("logdet n=%d" % n, abs(ch.logdet() - ld_ref) <= 1e-11 * abs(ld_ref))
        y = torch.from_numpy(rng.standard_normal(n)).to(dev)
        alpha = ch.solve(y)
        aref = torch.cholesky_solve(y[:, None], ref)[:, 0]
        check("solve n=%d" % n, (alpha - aref).abs().max().item() <= 1e-8 * aref.abs().max().item())
        H.close()
    # 2. not positive definite: every rank raises
    Abad = torch.eye(900, dtype=torch.float64, device=dev)
    Abad[650, 650] = -2.0
    try:
        DistributedCholesky(900).set_from_function(lambda c0, c1: Abad[:, c0:c1]).factor()
        check("non-PD raises", False)
    except torch.linalg.LinAlgError:
        pass
    # 3. posterior mean / sd / NLL of one exact GP across the ranks against the dense oracle
    rng = np.random.default_rng(2)
    pts = np.unique(rng.integers(0, 60, size=(4000, 2)), axis=0)
    pts = pts[rng.permutation(len(pts))[:1300]].astype(np.float64)
    y = np.sin(pts[:, 0] / 7.0) * np.cos(pts[:, 1] / 9.0) + 0.05 * rng.standard_normal(len(pts))
    Xs = rng.uniform(0, 59, size=(777, 2))
    ls, var, noise = [6.0, 8.0], 1.3, 0.02
    mean, sd, nll = exact_gp_posterior(pts, y, Xs, kernel="Matern52", lengthscale=ls, variance=var, noise=noise,
                                       chunk_bytes=150 * 8 * 1408)          # several chunks of test columns per rank
    kp = O.KernelParams("Matern52", 2, [[0., 0.], [12., 16.]])
    with torch.no_grad():
        kp.u_var.copy_(torch.logit(torch.tensor((var - 1e-4) / (10 - 1e-4), dtype=torch.float64)))
        kp.u_ls.copy_(torch.zeros(2, dtype=torch.float64))          # sigmoid(0) = 1/2 -> ls = hi / 2
        kp.u_noise.copy_(torch.log(torch.tensor(noise, dtype=torch.float64)))
    gp = O.ExactGP(torch.from_numpy(pts), torch.from_numpy(y), kp, 1e-5)
    mref, vref = gp.predict(torch.from_numpy(Xs))
    check("posterior mean vs oracle", np.abs(mean - mref.numpy()).max() < 1e-9, "%.2e" % np.abs(mean - mref.numpy()).max())
    check("posterior sd vs oracle", np.abs(sd - vref.sqrt().numpy()).max() < 1e-9)
    nref = (gp.loss() - kp.neg_log_prior()).item()
    check("NLL vs oracle", abs(nll - nref) <= 1e-11 * abs(nref), "%.3e" % (abs(nll - nref) / abs(nref)))
    # 3b. distributed training: loss / gradient at P = world against the single-GPU entry point, and a short Adam run
    from gpim_amd.dist_chol import exact_gp_fit, exact_gp_nll_grad
    from gpim_amd.kernels import KernelSpec
    import ctypes
    lsb = [[1., 1.], [20., 20.]]
    spec = KernelSpec("Matern52", 2, lsb, jitter=1e-5)
    u0 = spec.draw_initial_u(torch.Generator().manual_seed(3))
    loss_d, grad_d = exact_gp_nll_grad(pts, y, u0, kernel="Matern52", lengthscale=lsb)
    H = _lib.Handle()
    mm = spec.struct()
    Xd, yd, ud = (torch.from_numpy(a).to(dev) for a in (pts, y, u0.numpy()))
    out = torch.zeros(1 + spec.n_params, dtype=torch.float64, device=dev)
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(mm), _lib.ptr(Xd), _lib.ptr(yd), len(y), _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu().numpy()
    check("distributed loss", abs(loss_d - o[0]) <= 1e-11 * abs(o[0]), "%.3e" % (abs(loss_d - o[0]) / abs(o[0])))
    check("distributed gradient", np.abs(grad_d - o[1:]).max() <= 1e-8 * np.abs(o[1:]).max(),
          "%.3e" % (np.abs(grad_d - o[1:]).max() / np.abs(o[1:]).max()))
    hist = torch.zeros((6, spec.n_params), dtype=torch.float64, device=dev)
    us = ud.clone()
    _lib.check(H.lib.gpimhip_fit_exact(H.h, ctypes.byref(mm), _lib.ptr(Xd), _lib.ptr(yd), len(y), _lib.ptr(us), 0.1, 6,
                                       _lib.ptr(hist), None))
    hyper, uf = exact_gp_fit(pts, y, kernel="Matern52", lengthscale=lsb, learning_rate=0.1, iterations=6, u0=u0)
    hs = hist.cpu().numpy()
    got = np.concatenate([hyper["variance"][:, None], hyper["lengthscale"], hyper["noise"][:, None]], axis=1)
    check("distributed Adam trajectory", np.abs(got / hs - 1).max() <= 1e-8, "%.3e" % np.abs(got / hs - 1).max())
    ulist = [torch.empty_like(uf) for _ in range(world)]
    gdist.all_gather(ulist, uf)
    check("every rank holds the same parameters", all(torch.equal(ulist[0], t) for t in ulist))
    H.close()
    # 4. independent slices dealt to the ranks, gathered to rank 0 (device tensors through gather_to_root)
    cube, _ = hyperspectral_cube(size=24, nspec=6)
    kw = dict(kernel="RBF", lengthscale=[[1., 1.], [10., 10.]], learning_rate=0.1, iterations=5)
    res = gdist.reconstruct_slices(cube, axis=-1, batch=4, **kw)
    if rank == 0:
        m0, s0 = res
        check("reconstruct_slices shape", m0.shape == cube.shape and np.isfinite(m0).all() and np.isfinite(s0).all())
        from gpim_amd import gprutils
        k = 3
        Rk = cube[..., k]
        mk, sk, _ = gpim_amd.reconstructor(gprutils.get_sparse_grid(Rk), Rk, gprutils.get_full_grid(Rk), verbose=0, **kw).run()
        check("slice 3 equals its stand-alone reconstruction", np.array_equal(m0[..., k], mk) and np.array_equal(s0[..., k], sk))
    else:
        check("reconstruct_slices returns None off the root", res is None)
    # 5. sharded acquisition sweep: every rank ends up with the same ranking as an unsharded optimizer
    from problems import bo_test_problem
    bo_kw = dict(acquisition_function="ei", exploration_steps=2, batch_size=20, gp_iterations=30, verbose=0,
                 filename=os.path.join("/tmp", "dist2_bo_%d" % rank))
    f, Z = bo_test_problem()
    X0, Xf = gpim_amd.utils.get_sparse_grid(Z), gpim_amd.utils.get_full_grid(Z)
    bo_s = gpim_amd.boptimizer(X0, Z.copy(), Xf, f, shard_candidates=True, **bo_kw)
    bo_s.run()
    bo_1 = gpim_amd.boptimizer(X0, Z.copy(), Xf, f, **bo_kw)
    bo_1.run()
    check("sharded BO queries the same points", bo_s.indices_all == bo_1.indices_all,
          "%s vs %s" % (bo_s.indices_all, bo_1.indices_all))
    # 6. the reflection blocks of a complete grid dealt to the ranks (gpim_amd.dist_symm): no data-path collective, one
    #    all-reduce of eleven doubles per Adam iteration; against the single-process structured reconstructor
    from gpim_amd.dist_symm import symm_gp_fit, symm_gp_posterior
    for shape in ((12, 10), (9, 6, 1)):
        rs = np.random.default_rng(sum(shape))
        g = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        R = np.cos(g[0] / 3.0) * np.sin(g[1] / 2.0 + 0.2) + 0.05 * rs.standard_normal(shape)
        Xg = gpim_amd.utils.get_full_grid(R)
        kws = dict(kernel="Matern52", lengthscale=[[1.] * len(shape), [6.] * len(shape)])
        hyp, us = symm_gp_fit(Xg, R, learning_rate=0.1, iterations=6, **kws)
        rec = gpim_amd.reconstructor(Xg, R, Xg, structured=True, learning_rate=0.1, iterations=6, verbose=0, **kws)
        m1, s1, h1 = rec.run()
        check("sharded blocks %s: loss history" % (shape,), np.allclose(hyp["loss"], rec.loss_all, rtol=1e-11, atol=0),
              "%s vs %s" % (hyp["loss"][-1], rec.loss_all[-1]))
        check("sharded blocks %s: lengthscales" % (shape,),
              np.allclose(hyp["lengthscale"], np.asarray(h1["lengthscale"]), rtol=1e-10, atol=0))
        mp, sp = symm_gp_posterior(Xg, R, Xg.reshape(len(shape), -1).T, us, **kws)
        check("sharded blocks %s: posterior" % (shape,), np.abs(mp - m1.ravel()).max() < 1e-9 and np.abs(sp - s1.ravel()).max() < 1e-9,
              "%.2e %.2e" % (np.abs(mp - m1.ravel()).max(), np.abs(sp - s1.ravel()).max()))
        mg, sg = symm_gp_posterior(Xg, R, None, us, **kws)
        check("sharded blocks %s: posterior on the grid (mirrored variance)" % (shape,),
              np.abs(mg - m1.ravel()).max() < 1e-9 and np.abs(sg - s1.ravel()).max() < 1e-9,
              "%.2e %.2e" % (np.abs(mg - m1.ravel()).max(), np.abs(sg - s1.ravel()).max()))
        ul = [torch.empty_like(us.cpu()) for _ in range(world)]
        dist.all_gather(ul, us.cpu())
        check("sharded blocks %s: every rank holds the same parameters" % (shape,), all(torch.equal(ul[0], t) for t in ul))
    # 7. fault injection: a 300 us spin kernel in front of EVERY engine launch -- the panel chain, pack and vector solves on
    #    the side stream, every consumer on the main stream (GPIM_DIST_FAULT_DELAY_US, gpim_amd/dist_chol.py).  The two
    #    streams, the two broadcast buffers and the pair buffers are ordered by explicit event / Work edges only; with the
    #    launches pushed apart a missing edge reads or overwrites a buffer at the wrong time.  Every product of the schedule
    #    must come out BITWISE as in the undelayed run.
    def schedule_products(n, seed):
        rng7 = np.random.default_rng(seed)
        Bm = rng7.standard_normal((n, n // 3))
        A7 = torch.from_numpy(Bm @ Bm.T + n * np.eye(n)).to(dev)
        y7 = torch.from_numpy(rng7.standard_normal(n)).to(dev)
        Bq = torch.from_numpy(rng7.standard_normal((n, 200)))
        ch = DistributedCholesky(n)
        ch.set_from_function(lambda c0, c1: A7[:, c0:c1]).factor()
        Lg = ch.gather_lower().clone()
        al = ch.solve(y7).clone()
        Bp = torch.zeros((ch.layout.np, 200), dtype=torch.float64, device=dev)
        Bp[:n] = Bq.to(dev)
        q = ch.solve_colsumsq(Bp).clone()
        Xl = ch.inverse()
        Xc = Xl.clone()
        Kl = ch.kinv(Xl).clone()
        torch.cuda.synchronize()
        return Lg, al, q, Xc, Kl
    for n in (2600, 3300):
        os.environ["GPIM_DIST_FAULT_DELAY_US"] = "0"
        base = schedule_products(n, n)
        os.environ["GPIM_DIST_FAULT_DELAY_US"] = "300"
        slow = schedule_products(n, n)
        os.environ["GPIM_DIST_FAULT_DELAY_US"] = "0"
        for name, a, b in zip(("factor", "solve", "colsumsq", "inverse", "kinv"), base, slow):
            check("fault injection n=%d: %s unchanged under 300 us delays" % (n, name), torch.equal(a, b),
                  "%.3e" % (a - b).abs().max().item())
    flag = torch.tensor([0 if ok else 1], dtype=torch.int64)
    dist.all_reduce(flag)
    for m in msgs:
        print(m, flush=True)
    gdist.barrier()
    if rank == 0:
        print("DIST2 OK" if flag.item() == 0 else "DIST2 FAILED (%d ranks)" % flag.item(), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 0 else 1)


if __name__ == "__main__":
    main()
