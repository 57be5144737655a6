"""bench.py's N > 1 path end to end, rehearsed with two ranks on the one GPU of the test box.  `python bench.py --gpus 2`
with no RANK / WORLD_SIZE set starts its two ranks itself (the command the driver runs); one case goes through
torch.distributed.run as well.  Peer windows (hipIpcOpenMemHandle) work between two processes on one device, so
the default transport is the one N ranks on N GPUs get; the RCCL group cannot come up with two ranks on one device, so that case
checks the fall-back to gloo with host-staged records -- the same per-level protocol, the same engine calls
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


# transport "auto": peer windows come up (two processes may map each other's windows on one GPU too) and carry the exchange;
# "rccl" with the nccl backend: the self-test must notice it cannot run with two ranks on one device and fall back to gloo, on
# every rank; "gloo": records staged through the host
@pytest.mark.parametrize("backend,launcher,transport", [("gloo", "self", "auto"), ("nccl", "self", "auto"), ("nccl", "self", "rccl"),
                                                        ("gloo", "self", "gloo"), ("gloo", "torchrun", "auto")])
def test_bench_two_ranks_on_one_gpu(backend, launcher, transport):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OCRGPU_BENCH_BACKEND=backend, MASTER_ADDR="127.0.0.1", OCRGPU_BENCH_TRANSPORT=transport)
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
    assert "digest" in out["config"]["parity"]
    if transport == "auto":
        assert "peer windows" in wl and out["config"]["transport"] == "windows"
    else:
        assert "all-to-all (gloo" in wl                          # one GPU: RCCL refuses two ranks on one device
    assert out["roofline"]["frac"] > 0


def test_bench_falls_back_when_the_windows_fail_mid_run():
    """rank 1 drops out of the first resolve over the windows: rank 0's wait gives up (OCRGPU_PEER_TIMEOUT_MS), both ranks drop
    the windows together and the run completes over the next transport"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OCRGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OCRGPU_BENCH_INJECT_FAILURE="windows", OCRGPU_PEER_TIMEOUT_MS="300")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n-edt", "200000", "--width", "4000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    # the next transport of the chain: the same windows with every wave releasing at system scope (OCRGPU_PEER_FENCE=wave)
    assert out["config"]["transport"] == "windows-wave-fence" and "failed its first resolve" in r.stderr
    assert "digest" in out["config"]["parity"]


@pytest.mark.parametrize("bad,then", [("windows", "windows-wave-fence"), ("windows,windows-wave-fence", "gloo")])
def test_bench_drops_a_transport_whose_digest_is_wrong(bad, then):
    """one rank's digest over (EDT id, level, HASH value) is off by one bit after the proof resolve over `bad`: every rank drops that
    transport (counts and levels were right -- only the record differs) and the run completes over the next one whose digest matches
    the single-partition resolve of the same DAG"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OCRGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OCRGPU_BENCH_INJECT_DIGEST=bad)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n-edt", "200000", "--width", "4000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"]["transport"] == then and "the single-partition resolve leaves" in r.stderr
    assert "digest" in out["config"]["parity"]


def test_bench_fails_when_no_transport_leaves_the_right_record():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(OCRGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", OCRGPU_BENCH_INJECT_DIGEST="gloo", OCRGPU_BENCH_TRANSPORT="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n-edt", "200000", "--width", "4000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode != 0 and "PARITY FAILURE" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
