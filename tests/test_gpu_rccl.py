"""The exchange code over RCCL on the one GPU of the test box: a process group of one rank (backend "nccl") with the
exchanges forced, and bench.py's two exchange-step workloads started with two ranks on that GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_exchanges_over_rccl_one_rank():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-u", os.path.join(HERE, "rccl_child.py")], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl crt ok" in r.stdout and "rccl fourstep ok" in r.stdout


@pytest.mark.parametrize("workload", ["crt", "fourstep"])
def test_bench_two_ranks_on_one_device(workload):
    """bench.py --gpus 2 with both ranks on GPU 0 (collectives staged through host memory over gloo: RCCL needs one GPU
    per rank): rank spawning, the sharded product of the workload, its oracle / evaluation check and the line."""
    env = dict(os.environ, NTTX_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "2",
                        "--warmup", "1", "--no-cpu"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and "error" not in out
    assert "rehearsal" in out
