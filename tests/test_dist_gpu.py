"""The RCCL branch of bench.py on real hardware.  A one-GPU box cannot host two ranks (RCCL refuses two ranks on one device), but it can
run the N > 1 CODE PATH with a one-rank communicator: process-group initialisation over RCCL, the all-gather of the final latents, the
barrier-bracketed timed loop and the max-over-ranks reduction -- under the driver's launcher form (torch.distributed.run) and plain."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--gpus", "1", "--rccl-selftest", "--steps", "1", "--warmup", "1", "--batch", "1", "--size", "512", "--no-cpu-baseline", "--no-fp8-lines",
         "--no-roofline"]


def _line(r):
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[0])


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_bench_rccl_path_with_one_rank_under_torch_distributed_run():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py")] + FLAGS
    ln = _line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT))
    assert ln["n_gpus"] == 1 and ln["rccl_ranks"] == 1 and len(ln["rank_ms_per_step"]) == 1 and ln["value"] > 0
    assert abs(ln["ms_per_step"] - ln["rank_ms_per_step"][0]) < 1e-2
