"""Root-causing the 'wrong factor once' event of round 3 (two ranks sharing one GPU, tests/tools/dist2_worker.py): REPS
repetitions of the P = 2 factorisation with BOTH references -- torch.linalg.cholesky on the device (vendor solver) and on
the host -- and the residual |L L^T - A| / |A| of every candidate, so that a mismatch names the side that is wrong.
    GPIM_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/r4_dist2_stress.py REPS
Prints one line per mismatch and a summary per rank."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from gpim_amd import dist as gdist
from gpim_amd.dist_chol import DistributedCholesky

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rank, world, local_rank = gdist.init_from_env()
    dev = torch.device("cuda", local_rank)
    sizes = (700, 1500, 2600)
    bad = {"product": 0, "device_ref": 0, "host_ref": 0}
    worst = {"product": 0.0, "device_ref": 0.0, "host_ref": 0.0}
    mats = {}
    for n in sizes:
        rng = np.random.default_rng(n)
        Bm = rng.standard_normal((n, n // 3))
        Ah = torch.from_numpy(Bm @ Bm.T + n * np.eye(n))
        mats[n] = (Ah, Ah.to(dev), torch.linalg.cholesky(Ah))
    for rep in range(reps):
        n = sizes[rep % 3]
        Ah, A, Lh = mats[n]
        ch = DistributedCholesky(n)
        ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
        Lp = ch.gather_lower()
        Ld = torch.linalg.cholesky(A)                      # device reference (vendor solver), while the other rank is busy
        torch.cuda.synchronize()
        na = Ah.abs().max().item()
        for name, L in (("product", Lp.cpu()), ("device_ref", Ld.cpu()), ("host_ref", Lh)):
            res = (L @ L.T - Ah).abs().max().item() / na   # residual on the HOST
            worst[name] = max(worst[name], res)
            if not res < 1e-12:
                bad[name] += 1
                print("rank %d rep %d n=%d: %s residual %.3e  (max|L - host| %.3e)" % (rank, rep, n, name, res,
                      (L - Lh).abs().max().item()), flush=True)
        del ch, Lp, Ld
    print("rank %d: %d repetitions; wrong factors: %s; worst residuals: %s; NO_CACHING=%s" % (
        rank, reps, bad, {k: "%.2e" % v for k, v in worst.items()}, os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING", "0")), flush=True)
    gdist.barrier()
    import torch.distributed as dist
    dist.destroy_process_group()

main()
