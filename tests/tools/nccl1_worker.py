"""Worker of tests/test_gpu_dist2.py::test_rccl_world1: ONE rank under the ``nccl`` backend (= RCCL; a one-rank communicator
is legal) on the one GPU of the test box.  What gloo cannot show -- it stages device tensors through the host and blocks
the host -- runs here with RCCL's real stream semantics:

  * gpim_amd.dist.all_gather / gather / barrier / gather_to_root's collective on DEVICE tensors (dist.py: the branches
    north_star's "RCCL only for the argmax/gather" names);
  * the distributed Cholesky with every broadcast / all-reduce of its schedule ISSUED (GPIM_DIST_FORCE_COLLECTIVES=1):
    ``async_op=True`` broadcasts started on the side stream and consumed on the main stream through their Work handles,
    with a 300 us spin kernel in front of every engine launch (GPIM_DIST_FAULT_DELAY_US) so that a missing
    ``work.wait()`` / event edge changes the result.  Compared bitwise with the plain one-process run.
Prints "NCCL1 OK"."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpim_amd import dist as gdist                                          # noqa: E402
from gpim_amd.dist_chol import DistributedCholesky                          # noqa: E402


def products(n, seed, dev):
    rng = np.random.default_rng(seed)
    Bm = rng.standard_normal((n, n // 3))
    A = torch.from_numpy(Bm @ Bm.T + n * np.eye(n)).to(dev)
    y = torch.from_numpy(rng.standard_normal(n)).to(dev)
    Bq = torch.from_numpy(rng.standard_normal((n, 200)))
    ch = DistributedCholesky(n)
    ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
    Lg = ch.gather_lower().clone()
    al = ch.solve(y).clone()
    Bp = torch.zeros((ch.layout.np, 200), dtype=torch.float64, device=dev)
    Bp[:n] = Bq.to(dev)
    q = ch.solve_colsumsq(Bp).clone()
    Xl = ch.inverse()
    Xc = Xl.clone()
    Kl = ch.kinv(Xl).clone()
    torch.cuda.synchronize()
    return Lg, al, q, Xc, Kl


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    # 1. the schedule's products in a plain process (no process group: no collective is issued)
    base = {n: products(n, n, dev) for n in (2600, 3300)}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    ok = True

    def check(name, cond, detail=""):
        nonlocal ok
        if not cond:
            ok = False
            print("FAILED", name, detail, flush=True)

    # 2. device-tensor collectives of gpim_amd.dist under RCCL
    gdist.barrier()
    t = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4)
    parts = [torch.empty_like(t)]
    gdist.all_gather(parts, t)
    check("all_gather (device tensors, RCCL)", torch.equal(parts[0], t))
    pairs = torch.stack([torch.tensor([3.5, 1.25, -7.0], dtype=torch.float64, device=dev),
                         torch.tensor([11.0, 5.0, 2.0], dtype=torch.float64, device=dev)])
    got = [torch.empty_like(pairs)]
    gdist.all_gather(got, pairs)                                            # (value, index) pairs of the sharded ranking
    check("all_gather of (value, index) pairs", torch.equal(got[0], pairs))
    parts = [torch.zeros_like(t)]
    gdist.gather(t, parts, dst=0)
    check("gather (device tensors, RCCL)", torch.equal(parts[0], t))
    red = torch.arange(11, dtype=torch.float64, device=dev)
    dist.all_reduce(red)
    check("all_reduce", torch.equal(red, torch.arange(11, dtype=torch.float64, device=dev)))
    # 3. the distributed Cholesky with its collectives issued on RCCL's stream and the launches pushed apart
    os.environ["GPIM_DIST_FORCE_COLLECTIVES"] = "1"
    for delay in ("0", "300"):
        os.environ["GPIM_DIST_FAULT_DELAY_US"] = delay
        for n in (2600, 3300):
            got = products(n, n, dev)
            for name, a, b in zip(("factor", "solve", "colsumsq", "inverse", "kinv"), base[n], got):
                check("RCCL world 1, delay %s us, n=%d: %s equals the plain run bitwise" % (delay, n, name), torch.equal(a, b),
                      "%.3e" % (a - b).abs().max().item())
    os.environ["GPIM_DIST_FAULT_DELAY_US"] = "0"
    os.environ["GPIM_DIST_FORCE_COLLECTIVES"] = "0"
    # 4. the sharded training loop's all-reduce on device tensors (exact_gp_fit at world 1 skips it by design; the reflection
    #    blocks' driver likewise): covered by 2.
    gdist.barrier()
    dist.destroy_process_group()
    print("NCCL1 OK" if ok else "NCCL1 FAILED", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
