"""bench.py's N > 1 path end to end, rehearsed with two ranks on the one GPU of the test box.  `python bench.py --gpus 2`
with no RANK / WORLD_SIZE set starts its two ranks itself (the command the driver runs); one case goes through
torch.distributed.run as well.  The RCCL group cannot come up with two ranks on one device (or is not asked for),
so the exchange runs over gloo with host-staged records -- the same per-level protocol, the same engine calls
(ocrgpu_step with engine-written headers, ocrgpu_exchange_apply_strided), and bench.py's own checks (every EDT fired at
the level of its layer on every rank, counts add up).  Collected late (file name) so that trouble here cannot hide the
parity tests from a run with -x."""
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


# "nccl": the self-test must notice it cannot run and fall back, on every rank
@pytest.mark.parametrize("backend,launcher", [("gloo", "self"), ("nccl", "self"), ("gloo", "torchrun")])
def test_bench_two_ranks_on_one_gpu(backend, launcher):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OCRGPU_BENCH_BACKEND=backend, MASTER_ADDR="127.0.0.1")
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n-edt", "400000", "--width", "4000"]
    if launcher == "self":       # bench.py spawns its own ranks before it touches the GPU
        cmd = [sys.executable] + bench
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + bench
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["n_edt"] == 2 * 400000 + 1 and 100 <= out["config"]["levels"] <= 103   # 100 layers + latch/sink
    wl = out["config"]["workload"]
    assert "all-to-all (gloo" in wl or "all-to-all (RCCL" in wl   # one GPU: gloo (RCCL refuses two ranks on one device); more: RCCL may come up
    if backend == "gloo":
        assert "all-to-all (gloo" in wl
    assert out["roofline"]["frac"] > 0


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
