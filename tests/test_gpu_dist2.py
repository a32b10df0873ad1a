"""
The multi-rank code paths with the real HIP engine on a ONE-GPU box: two ranks share the device over gloo
(GPIM_DIST_BACKEND=gloo; RCCL refuses two ranks on one GPU).  What runs: the block-column-cyclic Cholesky and the
distributed posterior at P = 2 (tests/tools/dist2_worker.py), slices dealt to ranks + gather_to_root with device
tensors, the sharded acquisition ranking, and bench.py's --gpus 2 branches (c2 / c3 / c2full at reduced iteration
counts: the point is that every line of the world > 1 paths executes once before the driver's 8-GPU run).
No scaling number comes out of this: both ranks compete for the same GPU.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, timeout):
    env = dict(os.environ, GPIM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          timeout=timeout)


def test_two_ranks_one_gpu_product_paths(ensure_built):
    r = _torchrun([os.path.join(ROOT, "tests", "tools", "dist2_worker.py")], 900)
    assert r.returncode == 0 and "DIST2 OK" in r.stdout, r.stdout[-4000:]


def test_rccl_world1(ensure_built):
    """One rank under the nccl backend (RCCL) on the one GPU: the device-tensor collectives of gpim_amd.dist and the
    distributed Cholesky with every broadcast / all-reduce issued on RCCL's own stream, launches pushed apart by spin
    kernels -- bitwise equal to the plain run (tests/tools/nccl1_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("GPIM_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "nccl1_worker.py")], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "NCCL1 OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.parametrize("workload,extra", [("c3", ["--iterations", "5"]), ("c2", ["--iterations", "2"]),
                                            ("c1", ["--iterations", "5"]), ("c2full", [])])
def test_bench_two_ranks(ensure_built, workload, extra):
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "1", "--warmup", "0"]
                  + extra, 1500)
    assert r.returncode == 0, r.stdout[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["unit"] == "grid-points/s"
    assert out["scaling"] == ("weak" if workload in ("c1", "c2") else "strong")
