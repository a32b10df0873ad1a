"""The fused launch (gradient contraction + finalize step, next theta carried) against the two launches of rounds 1-5
(GPIMHIP_NO_FUSED_FINALIZE=1): hyper-parameter and loss histories of T iterations must be BITWISE equal (same reductions in
the same order), at sizes on both sides of the graph-replay / eager switch and of the alpha-partials switch (np <= 8192), for
a lock-step batch, and repeated (a stale read of another workgroup's partial sums would show as a difference)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))


def run(N, T, kind, B):
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    dev = torch.device("cuda:0")
    H = _lib.Handle(); lib = H.lib
    side = int(np.ceil(np.sqrt(N * 4)))
    rng = np.random.default_rng(N)
    out = []
    Xs, ys = [], []
    for b in range(B):
        flat = rng.choice(side * side, size=N, replace=False); flat.sort()
        X = np.stack([flat // side, flat % side], 1).astype(np.float64)
        Xs.append(X); ys.append(np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N))
    Xd, yd = torch.from_numpy(np.stack(Xs)).to(dev), torch.from_numpy(np.stack(ys)).to(dev)
    spec = KernelSpec(kind, 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    m = spec.struct()
    for rep in range(3):
        u = spec.draw_initial_u(torch.Generator().manual_seed(1)).repeat(B, 1).to(dev).contiguous()
        hist = torch.zeros(B, T, spec.n_params, dtype=torch.float64, device=dev)
        loss = torch.zeros(B, T, dtype=torch.float64, device=dev)
        _lib.check(lib.gpimhip_fit_exact_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), N * 2, _lib.ptr(yd), N, B, _lib.ptr(u), 0.1, T,
                                                 _lib.ptr(hist), _lib.ptr(loss)))
        out.append(np.concatenate([hist.cpu().numpy().ravel(), loss.cpu().numpy().ravel(), u.cpu().numpy().ravel()]))
    assert all(np.array_equal(out[0], o) for o in out[1:]), "run-to-run difference"
    return out[0]


if __name__ == "__main__":
    if len(sys.argv) > 1:
        N, T, kind, B = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
        np.save(sys.argv[5], run(N, T, kind, B))
        sys.exit(0)
    ok = True
    for N, T, kind, B in ((300, 200, "RBF", 1), (1207, 100, "RBF", 16), (2500, 60, "Matern52", 1), (4212, 40, "RBF", 1),
                          (6000, 10, "RationalQuadratic", 1), (8300, 6, "Matern52", 1), (1207, 100, "Matern52", 5)):
        res = []
        for env in ({}, {"GPIMHIP_NO_FUSED_FINALIZE": "1"}):
            f = "/tmp/r6_bits_%d_%s.npy" % (N, "two" if env else "fused")
            subprocess.run([sys.executable, os.path.abspath(__file__), str(N), str(T), kind, str(B), f], check=True,
                           env=dict(os.environ, **env))
            res.append(np.load(f))
        same = np.array_equal(res[0], res[1])
        ok &= same
        print("N=%d T=%d %s B=%d: fused == two launches bitwise: %s (max rel %.2e)" % (N, T, kind, B, same, np.max(np.abs(res[0] - res[1]) / (np.abs(res[1]) + 1e-300))), flush=True)
    print("FUSED BITS OK" if ok else "FUSED BITS DIFFER")
