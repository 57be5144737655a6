#!/usr/bin/env python3
"""bench.py -- dependence satisfactions/sec of the MI355X engine on BASELINE config 2
(layered random DAG, 10M EDTs, mean fan-in 4, ONCE output events).

One "step" = one full resolve of the built DAG: restore the dependence state (re-arm), satisfy the
start event through the C-ABI, run the worker loop until the sink EDT has fired.  The DAG is built
once, before the timed region, and is resident in HBM.  Prints ONE JSON line (see DESIGN.md
"Measurement" for what each field means and how roofline.achieved is computed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n-edt 10000000] [--width 10000]
                    [--no-cpu-baseline] [--no-parity]

`--gpus N` with N > 1 and no RANK / WORLD_SIZE in the environment: this process starts N fresh rank processes
itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE it imports torch or
touches HIP, relays rank 0's JSON line and exits non-zero if any rank does -- the way the reference's
ocrrun_x86-mpi launches its N policy domains on one node (ocr/build/x86-mpi/ocrrun_x86-mpi:77-108,147).
Launched under torch.distributed.run (RANK / WORLD_SIZE set), it is one of the ranks and WORLD_SIZE must equal N.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-edt", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=10_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-sample-edts", type=int, default=0, help="EDTs of the CPU baseline's DAG (0 = the full workload)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))          # the parent never initialises the GPU

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: n_gpus in the JSON line would not be what was asked for")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libocrgpu has no CPU path")
    local = local % torch.cuda.device_count()    # rehearsals put several ranks on one GPU (gloo backend)
    torch.cuda.set_device(local)
    backend = os.environ.get("OCRGPU_BENCH_BACKEND", "nccl")
    if world > 1:
        # The default group is a gloo one (it always comes up); the exchange runs on an RCCL group created next to it,
        # and only after that group has passed a self-test on every rank -- otherwise the job stays on gloo with
        # host-staged records and says so in config.workload.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))

    from ocr_amd import engine, opstream
    if world > 1:
        return main_partitioned(args, torch, dist, rank, world, local, backend)

    # ---- build: every rank owns one partition = one independent layered DAG (seeded per rank) ----
    seed = opstream.SEED_BASE + 2 + 1000 * rank
    s = opstream.layered_dag(args.n_edt, args.width, seed=seed)
    start_op = s.main[-1:]                      # the final ocrEventSatisfy(start)
    built = s.with_main(s.main[:-1])
    t0 = time.time()
    prog = engine.StreamProgram(built, device=local, flags=engine.F_TIMING).build()
    eng = prog.eng
    eng.snapshot()
    eng.sync()
    build_s = time.time() - t0
    start_guid = prog._guids(start_op["a"])

    def step():
        eng.restore()
        eng.event_satisfy_slot(start_guid)
        return eng.resolve()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        st = step()
    k0 = eng.stats().step_kernel_ns
    l0 = eng.stats().launches
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step_kernel_s = (eng.stats().step_kernel_ns - k0) * 1e-9 / args.steps
    launches = (eng.stats().launches - l0) / args.steps
    # which worker loop ran: the barrier-free kernel (k_async: ONE launch fires the whole DAG; the launches before it only apply the
    # main program's satisfy and are not in the kernel time) or one k_step launch per level
    barrier_free = eng.stats().async_runs >= args.steps
    kernel_launches = 1.0 if barrier_free else launches
    kernel_name = ("k_async (the one launch of a resolve, HIP events on the engine stream)" if barrier_free
                   else "k_step (all launches of one resolve, HIP events on the engine stream)")

    n_sat = s.meta["n_satisfactions"]          # EDT pre-slot + latch pre-slot satisfies per resolve
    assert st.fired == s.n_edt, (st.fired, s.n_edt)
    assert sum(st.err) == 0, list(st.err)
    assert st.edt_sat + s.meta["n_leaves"] == n_sat

    parity = "skipped"
    if not args.no_parity and rank == 0:
        from oracle import ref as O            # checker only: the CPU restatement, full size
        ref = O.run_restatement(s)
        diff = prog.result().diff(ref)
        if diff:
            raise SystemExit(f"PARITY FAILURE vs restatement: {diff}")
        parity = f"bit-exact vs CPU restatement at full size (digest {ref.digest():016x})"

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = n_sat * world / (dt / args.steps)
        # algorithmic bytes: 16 B per edge satisfied + 20 B per EDT fired
        alg_bytes = 16 * n_sat + 20 * s.n_edt
        out = {
            "metric": "dependence satisfactions/sec (EDTs fired/sec) at 1/2/4/8 MI355X",
            "value": value, "unit": "satisfactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"layered random DAG, {args.n_edt} EDTs in {args.n_edt // args.width} layers of {args.width}, "
                                   f"depc=1+popcount(6 bits) (mean 4), ONCE output events, HASH bodies, latch+sink; "
                                   f"one independent partition per GPU",
                       "n_edt": s.n_edt, "n_satisfactions": n_sat, "levels": int(st.levels),
                       "edts_fired_per_s": s.n_edt * world / (dt / args.steps), "build_s": round(build_s, 2),
                       "worker_loop": "barrier-free (k_async)" if barrier_free else "level-synchronous (k_step)",
                       "launches_per_resolve": launches, "parity": parity},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / step_kernel_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_bytes / step_kernel_s / 1e9 / 8000.0,
                         "traffic": profiled_traffic(args, kernel="k_async" if barrier_free else "k_step")[0],
                         "traffic_source": profiled_traffic(args, kernel="k_async" if barrier_free else "k_step")[1],
                         "kernel": kernel_name,
                         "launches_per_resolve": kernel_launches, "alg_bytes_per_launch": alg_bytes / kernel_launches,
                         "us_per_launch": step_kernel_s / kernel_launches * 1e6,
                         "alg_bytes_per_resolve": alg_bytes, "kernel_s_per_resolve": step_kernel_s},
        }
        # the limit that actually binds scattered slot counting (SURVEY.md s8d "second roofline"): returning device-scope
        # atomics on random 64-byte lines, measured on an MI355X with tools/latbench.hip: 27e9 per second
        out["roofline_atomics"] = {"bound": "scattered device-scope atomics (measured, tools/latbench.hip)", "achieved": n_sat / step_kernel_s / 1e9,
                                   "peak": 27.0, "unit": "G atomics/s", "frac": n_sat / step_kernel_s / 27e9}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def launch_ranks(n):
    """Start n rank processes of this same command line (one per GPU) and relay rank 0's JSON line.  Nothing here
    imports torch or calls HIP: the children are started with subprocess (fresh interpreters), never exec'd over a
    process that has touched the GPU.  Returns the exit code for the parent (0 only if every rank exited 0 and rank 0
    printed exactly one JSON line with n_gpus == n)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        codes = [None] * n
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes):     # one rank failed: the others would wait for it in a collective
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.terminate()                       # the exact processes started above, nothing else
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        try:
                            codes[i] = p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[i] = p.wait()
                break
            time.sleep(0.1)
        out0.seek(0)
        text = out0.read()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    for ln in text.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != n:
        print(f"bench.py: expected one JSON line with n_gpus={n} from rank 0, got {lines!r}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def c_transport_up(part, torch, dist, rank, world):
    """Bring up libocrgpu's own RCCL communicator (ocrgpu_comm_init: one rank per engine) and run one content-checked
    all-to-all and one empty exchange round on it.  The verdict is agreed over the gloo default group: the C transport is
    used only if EVERY rank passed; otherwise the exchange stays on gloo with host-staged records."""
    from ocr_amd import engine
    ok, why = 1, ""
    devs = [None] * world
    dist.all_gather_object(devs, torch.cuda.current_device())
    if len(set(devs)) < world:                               # RCCL wants one device per rank: do not even try (it may not return)
        if rank == 0:
            print(f"[bench] {world} ranks on devices {devs}: ranks share a GPU, the exchange stays on gloo", file=sys.stderr)
        return False
    uid = [None]
    if rank == 0:
        try:
            uid = [engine.comm_get_unique_id()]
        except Exception as ex:  # noqa: BLE001 -- rank 0 still broadcasts (None), or the others would wait for it
            print(f"[bench] no RCCL unique id ({type(ex).__name__}: {str(ex)[:200]})", file=sys.stderr)
    dist.broadcast_object_list(uid, src=0)                   # 128 bytes over gloo: the out-of-band channel
    if uid[0] is None:
        return False
    try:
        part.eng.comm_init(uid[0], rank, world)
        src = (torch.arange(world * 1024, dtype=torch.int64, device="cuda") + 1_000_000 * rank)
        dst = torch.zeros_like(src)
        part.eng.comm_alltoall(src.data_ptr(), dst.data_ptr(), 1024 * 8)
        part.eng.sync()
        want = torch.cat([torch.arange(rank * 1024, (rank + 1) * 1024, dtype=torch.int64) + 1_000_000 * p for p in range(world)])
        assert torch.equal(dst.cpu(), want), "all-to-all content"
        rin, act, sent, queued = part.eng.exchange()
        assert (rin, sent) == (0, 0), (rin, sent)
    except Exception as ex:  # noqa: BLE001 -- whatever RCCL refuses (e.g. two ranks on one device in a rehearsal)
        ok, why = 0, f"{type(ex).__name__}: {str(ex)[:300]}"
    verdict = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
    if int(verdict.item()) == 1:
        return True
    print(f"[bench] rank {rank}: RCCL transport unusable ({why or 'another rank failed'}); the exchange stays on gloo", file=sys.stderr)
    return False


def all_ok(torch, dist, ok):
    v = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.all_reduce(v, op=dist.ReduceOp.MIN)
    return int(v.item()) == 1


def windows_up(part, torch, dist, rank, world):
    """Join the ranks by peer windows (ocrgpu_peer_export / ocrgpu_peer_connect: every rank maps every other rank's receive
    window with hipIpcOpenMemHandle; one node, which is what this bench runs on) and run one empty exchange round over them.
    Used only if EVERY rank got through both stages; otherwise the windows are dropped again and the next transport is tried."""
    from ocr_amd.distributed import connect_peer_windows_dist
    os.environ.setdefault("OCRGPU_PEER_TIMEOUT_MS", "5000")   # ranks enter a resolve within milliseconds of each other; a wait this long is a failure
    why = ""
    try:
        connect_peer_windows_dist(part, dist)
    except Exception as ex:  # noqa: BLE001 -- whatever HIP refuses on this machine
        why = f"{type(ex).__name__}: {str(ex)[:300]}"
    if all_ok(torch, dist, not why):
        try:
            rin, act, sent, queued = part.eng.exchange()
            assert (rin, sent) == (0, 0), (rin, sent)
        except Exception as ex:  # noqa: BLE001
            why = f"{type(ex).__name__}: {str(ex)[:300]}"
        if all_ok(torch, dist, not why):
            return True
    part.eng.peer_fini()
    print(f"[bench] rank {rank}: peer windows unusable ({why or 'another rank failed'})", file=sys.stderr)
    return False


def main_partitioned(args, torch, dist, rank, world, local, backend):
    """N > 1: ONE global DAG of world x n_edt EDTs (same layer count, layers world x wider), GUID space partitioned
    round-robin over the ranks; satisfies whose destination lives on another GPU are exchanged per level with an
    all-to-all of 32-byte records (RCCL over xGMI).  Weak scaling: every rank owns n_edt EDTs."""
    from ocr_amd import engine
    from ocr_amd.distributed import resolve_distributed, distributed_exchange
    from ocr_amd.workloads import LayeredPartition
    grp = None                                   # control collectives below: tiny tensors on the default (gloo) group

    def allreduce_sum(x):
        t = torch.tensor([int(x)], dtype=torch.int64)
        dist.all_reduce(t, group=grp)
        return int(t.item())

    N, width = args.n_edt * world, args.width * world
    t0 = time.time()
    part = LayeredPartition(N, width, rank, world, device=local, flags=engine.F_TIMING, allreduce_sum=allreduce_sum)
    part.build()
    distributed_exchange(part, dist)             # deliver what construction sent across (nothing for this DAG)
    # the per-level loop and the exchange run in C (ocrgpu_resolve_exchange) over peer windows or libocrgpu's own RCCL communicator;
    # when neither comes up (a rehearsal with several ranks on one GPU has no RCCL) the same protocol runs from Python over gloo
    part.eng.snapshot()
    part.eng.sync()
    expect_levels = N // width + 1

    # ---- what a resolve must leave behind: the order-independent digest over (EDT id, level, HASH value) of the WHOLE DAG, from
    # a single-partition resolve of the same global DAG on rank 0's GPU (the engine whose N = 1 record bench.py holds to the CPU
    # restatement bit by bit).  Every rank folds its own EDTs on the device (ocrgpu_digest_edts), the sums are added up over
    # gloo, and the total must be this number: records that crossed partitions arrived with their payloads, not just in the right
    # count (hc-dist-policy.c:2208).  Checked in the untimed proof resolve and again after the timed loop.
    want_digest = [None]
    if rank == 0:
        whole = LayeredPartition(N, width, 0, 1, device=local)
        whole.build()
        whole.satisfy_start()
        wst = whole.eng.resolve()
        assert wst.fired == N + 1 and sum(wst.err) == 0, (wst.fired, list(wst.err))
        whole.check()
        want_digest = [whole.eng.digest_edts(0, whole.n_local_edts)]
        whole.eng.close()
        del whole
    dist.broadcast_object_list(want_digest, src=0)
    want_digest = int(want_digest[0])

    def global_digest():
        mine = part.eng.digest_edts(0, part.n_local_edts)
        if transport in os.environ.get("OCRGPU_BENCH_INJECT_DIGEST", "").split(",") and rank == world - 1:
            mine ^= 1 << 17                      # (tests: one rank's record is off by one bit)
        halves = torch.tensor([mine & 0xFFFFFFFF, mine >> 32], dtype=torch.int64)   # (two halves: no signed overflow in the reduction)
        dist.all_reduce(halves, group=grp)
        return (int(halves[0].item()) + (int(halves[1].item()) << 32)) & 0xFFFFFFFFFFFFFFFF

    def step(time_steps=False):
        part.eng.restore()
        part.satisfy_start()
        if transport != "gloo":
            return part.eng.resolve_exchange(no_cascade=True, time_steps=time_steps)
        return resolve_distributed(part, dist)

    # Transports, best first (OCRGPU_BENCH_TRANSPORT = windows | rccl | gloo picks where the chain starts): peer windows (stores
    # into the receiver's memory over xGMI, no collective), the same windows with every wave releasing at system scope itself
    # (OCRGPU_PEER_FENCE=wave), the RCCL all-to-all, records staged through the host over gloo.  One untimed resolve proves each
    # candidate on the real DAG before anything is timed: a transport that fails it on any rank -- an exception, a wrong level
    # count, a digest that is not the single-partition one -- is dropped by all ranks together and the next one takes over.
    want = os.environ.get("OCRGPU_BENCH_TRANSPORT", "auto")
    chain = {"auto": ["windows", "windows-wave-fence", "rccl", "gloo"], "windows": ["windows", "windows-wave-fence", "gloo"],
             "rccl": ["rccl", "gloo"], "gloo": ["gloo"]}[want]

    def bring_up(name):
        if name.startswith("windows"):
            if name == "windows-wave-fence":
                os.environ["OCRGPU_PEER_FENCE"] = "wave"
            else:
                os.environ.pop("OCRGPU_PEER_FENCE", None)
            return windows_up(part, torch, dist, rank, world)
        if name == "rccl":
            return backend == "nccl" and c_transport_up(part, torch, dist, rank, world)
        return True

    def take_down(name):
        if name.startswith("windows"):
            part.eng.peer_fini()
        elif name == "rccl":
            part.eng.comm_fini()
        part.eng.restore()                        # the failed resolve's leftovers (records waiting in the send regions) do not
        part.eng.exchange_reset()                 # reach the next candidate's self-test round
        part.eng.sync()

    transport, proof = None, ""
    for name in chain:
        if not bring_up(name):
            continue
        transport = name
        why = ""
        try:
            if os.environ.get("OCRGPU_BENCH_INJECT_FAILURE") == name and rank == world - 1:
                raise RuntimeError("injected (tests: the other ranks must time out, and everybody falls back together)")
            got = step()
            assert got == expect_levels, f"{got} levels, expected {expect_levels}"
        except Exception as ex:  # noqa: BLE001
            why = f"{type(ex).__name__}: {str(ex)[:300]}"
        if all_ok(torch, dist, not why):         # (everybody finished the resolve: the collective below cannot hang on a rank that threw)
            got_digest = global_digest()
            if got_digest == want_digest:
                proof = f"digest {got_digest:016x} over (EDT id, level, HASH value) of all {N + 1} EDTs, summed over the ranks"
                break
            why = f"digest {got_digest:016x}, the single-partition resolve leaves {want_digest:016x}"
        print(f"[bench] rank {rank}: transport {name} failed its first resolve ({why or 'on another rank'})", file=sys.stderr)
        take_down(name)
        transport = None
    if transport is None:
        raise SystemExit("bench.py: no transport passed its first resolve (the last one tried was gloo with host-staged records): PARITY FAILURE")
    backend = {"windows": "peer windows: records stored into the receiver's memory over xGMI, no collective; C loop ocrgpu_resolve_exchange",
               "windows-wave-fence": "peer windows (OCRGPU_PEER_FENCE=wave: every wave releases at system scope): records stored into the "
                                     "receiver's memory over xGMI, no collective; C loop ocrgpu_resolve_exchange",
               "rccl": "all-to-all (RCCL, C transport: ocrgpu_resolve_exchange)",
               "gloo": "all-to-all (gloo, host-staged records)"}[transport]
    build_s = time.time() - t0

    def fence():
        torch.cuda.synchronize()
        dist.barrier(group=grp)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        levels = step()
    k0 = part.eng.stats().step_kernel_ns
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        levels = step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=grp)
    dt = float(t.item())
    st = part.eng.stats()
    assert sum(st.err) == 0, list(st.err)
    part.check()                                 # every owned EDT fired at the level of its layer
    got_digest = global_digest()                 # ... and the last timed resolve left the record the single-partition resolve leaves
    if got_digest != want_digest:
        raise SystemExit(f"PARITY FAILURE after the timed loop over {transport}: digest {got_digest:016x}, the single-partition resolve leaves {want_digest:016x}")
    fired = allreduce_sum(st.fired)
    edt_sat = allreduce_sum(st.edt_sat)
    assert fired == N + 1, (fired, N + 1)
    n_sat = part.sum_depc_global + 1 + part.n_leaves
    assert edt_sat == part.sum_depc_global + 1, (edt_sat, part.sum_depc_global + 1)
    step_kernel_s = (st.step_kernel_ns - k0) * 1e-9 / args.steps
    kernel_note = "k_step on rank 0 (HIP events on the engine stream, all launches of one resolve)"
    if transport != "gloo":
        # the C loop queues a level's launches ahead without events; one more resolve, outside the timed region, with HIP events
        # around every level's k_step launches (OCRGPU_RX_TIME_STEPS) gives the kernel time the roofline wants
        k1 = part.eng.stats().step_kernel_ns
        step(time_steps=True)
        step_kernel_s = (part.eng.stats().step_kernel_ns - k1) * 1e-9
        kernel_note = "k_step on rank 0 (HIP events around every level's launches, one extra resolve after the timed ones)"
    if rank == 0:
        ms = dt / args.steps * 1e3
        alg_bytes = (16 * n_sat + 20 * (N + 1)) / world
        out = {
            "metric": "dependence satisfactions/sec (EDTs fired/sec) at 1/2/4/8 MI355X",
            "value": n_sat / (dt / args.steps), "unit": "satisfactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"layered random DAG, {N} EDTs in {N // width} layers of {width} ({args.n_edt} EDTs per GPU), "
                                   f"depc=1+popcount(6 bits) (mean 4), ONCE output events, HASH bodies, latch+sink; GUID space "
                                   f"partitioned round-robin over {world} GPUs, cross-partition satisfies exchanged per level by {backend}",
                       "n_edt": N + 1, "n_satisfactions": n_sat, "levels": int(levels), "transport": transport,
                       "us_per_level": round(dt / args.steps / max(int(levels), 1) * 1e6, 1),
                       "edts_fired_per_s": (N + 1) / (dt / args.steps), "build_s": round(build_s, 2),
                       "parity": f"{proof} = the digest a single-partition resolve of the same DAG leaves on rank 0's GPU (compared in the "
                                 f"untimed first resolve and again after the timed loop); every EDT fired at the level of its layer on every rank"},
            "roofline": {"bound": "hbm", "achieved": alg_bytes / step_kernel_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": alg_bytes / step_kernel_s / 1e9 / 8000.0, "traffic": None,
                         "kernel": kernel_note,
                         "alg_bytes_per_resolve": alg_bytes, "kernel_s_per_resolve": step_kernel_s},
        }
        print(json.dumps(out))
    dist.barrier(group=grp)
    dist.destroy_process_group()


PROFILE_TAG = "r03b_config2"


def profiled_traffic(args, tag=PROFILE_TAG, kernel="k_step"):
    """HBM bytes per launch of the dominant kernel (k_async, or k_step) from the committed PMC passes (profiles/<tag>_pmc_summary.txt: FETCH_SIZE and
    WRITE_SIZE in KB, separate rocprofv3 --pmc runs of this same workload).  Counters cannot be read inside this
    process, so the figure is the profiled one, and it is only quoted for the workload and the engine source it was
    taken on: the summary records a hash of ocr_amd/csrc/ocrgpu.hip, and any other source (or workload shape) gives
    None.  Returns (bytes per launch or None, where the figure comes from)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "profiles", tag + "_pmc_summary.txt")
    if args.n_edt != 10_000_000 or args.width != 10_000:
        return None, "not profiled for this workload shape"
    kb, src = {}, None
    try:
        for line in open(path):
            if line.startswith("source:"):
                src = line.strip().split("=")[-1]
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                if (kernel + "(" in line or kernel + "<" in line) and ctr + " sum=" in line:   # ("k_async(", not "k_async_end(")
                    kb[ctr] = float(line.split("per dispatch ")[1].rstrip(")\n"))
        now = hashlib.sha256(open(os.path.join(here, "ocr_amd", "csrc", "ocrgpu.hip"), "rb").read()).hexdigest()[:16]
    except OSError:
        return None, "no committed profile"
    if len(kb) != 2:
        return None, "no committed profile"
    if src != now:
        return None, f"profiles/{tag}_pmc_summary.txt was taken on another engine source ({src}, now {now})"
    # scattered <=64-B requests: raw values (the guide's x2 correction is for 16-B-per-lane coalesced streams only)
    return (kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0, f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload on this engine source, not this run)"


def cpu_baseline(args):
    """The REFERENCE runtime (stock x86 pthread/HC config, PTR guids; oracle/_ref/ocr_replay_perf built from the
    reference sources) resolving a smaller twin of the same DAG on this box's host cores."""
    from oracle import ref as O
    from ocr_amd import opstream
    n = min(args.cpu_sample_edts or args.n_edt, args.n_edt)
    width = min(args.width, n)
    n -= n % width
    twin = opstream.layered_dag(n, width)
    if O.reference_available("perf"):
        # 1 worker and min(8, host share) workers: the reference's HC scheduler does not scale on this workload
        # (one finish latch, work-stealing deques), so the better of the two is reported with its worker count
        tried = {}
        for w in (1, min(8, max(1, (os.cpu_count() or 1)))):
            if w in tried:
                continue
            _, info = O.run_reference(twin, workers=w, guid="PTR", flavour="perf", want_result=False, timeout=900)
            tried[w] = info
        best = max(tried, key=lambda w: tried[w]["sat_per_s"])
        info = tried[best]
        what = "the full workload" if n == args.n_edt else "a smaller twin of the workload"
        other = "; ".join(f"{w} workers: {i['sat_per_s']:.3g}/s" for w, i in tried.items() if w != best)
        return {"value": info["sat_per_s"], "unit": "satisfactions/s", "cores": best, "kind": "reference",
                "sample": f"{what}: layered DAG, {n} EDTs in layers of {width} ({info['edges']} satisfactions in "
                          f"{info['resolve_s']:.2f} s), reference runtime (stock x86 pthread/HC, PTR guids) with {best} HC worker(s), "
                          f"timed first-satisfy..sink; the better of the worker counts tried is reported (also tried: {other})",
                "by_workers": {str(w): round(i["sat_per_s"], 1) for w, i in tried.items()}}
    t0 = time.perf_counter()
    _, st = O.run_restatement(twin, return_stats=True)
    dt = time.perf_counter() - t0
    return {"value": twin.meta["n_satisfactions"] / dt, "unit": "satisfactions/s", "cores": 1, "kind": "port",
            "sample": f"CPU restatement on a {n}-EDT twin, whole run {dt:.2f} s (includes construction)"}


if __name__ == "__main__":
    main()
