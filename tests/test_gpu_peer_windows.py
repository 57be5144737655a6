"""Peer windows (ocrgpu_peer_export / ocrgpu_peer_connect): the exchange of one node without a collective.  Senders' kernels
store their records into the receiver's window, the receiver's kernel waits for one arrival word per source; the loop above it
(ocrgpu_resolve_exchange) is the one the RCCL transport runs.  Exercised (a) with 2-4 partitions in one process (plain pointers
into each other's windows) against the single-partition oracle on the streams the RCCL-protocol tests use, and (b) with two and
three PROCESSES sharing this GPU, windows mapped with hipIpcOpenMemHandle -- the path N ranks on N GPUs take."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import streams
from ocr_amd import distributed as D
from ocr_amd import engine as E
from ocr_amd.opstream import layered_dag
from oracle import ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name", ["layered_1k", "event_chain", "latch_incr2_decr2", "counted_events", "channel_events", "cholesky_T8",
                                  "stencil_16x16x10", "counted_after_null_preslot", "sticky_early_and_late", "wide_fanin"])
def test_windows_in_one_process_are_partition_invariant(name, world):
    s = ALL[name]
    parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
    D.connect_peer_windows(parts)
    D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", ["layered_100k", "powerlaw_100k"])
def test_large_streams_over_windows(name, world):
    """100 000 EDTs over 2 / 4 / 8 partitions, rounds queued ahead: the power-law DAG has latches chained across partitions
    (arrivals fire events that send on: settle rounds, decided on the device), the layered one fills the regions"""
    s = ALL[name]
    parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
    D.connect_peer_windows(parts)
    levels = D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    want = O.run_restatement(s)
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(want) == [] and levels == int(want.level.max()) + 1


@pytest.mark.parametrize("G,steps,world", [(16, 10, 8), (64, 6, 4)])
def test_stencil_blocks_over_windows(G, steps, world):
    """config 4's placement (blocks of tiles per partition, only halo edges cross) with the C loop over the windows"""
    from ocr_amd.opstream import stencil_dag
    s = ALL["stencil_16x16x10"] if (G, steps) == (16, 10) else stencil_dag(G, steps)
    own = D.stencil_block_owner(G, steps, world)
    parts = D.build_lockstep([D.Partition(s, r, world, owner_edt=own) for r in range(world)])
    D.connect_peer_windows(parts)
    D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("mode", ["budget1", "budget2_ahead1", "ahead3", "host_driven", "host_driven_budget1"])
@pytest.mark.parametrize("name", ["layered_1k", "event_chain", "counted_events", "channel_events", "cholesky_T8", "wide_fanin",
                                  "sticky_early_and_late"])
def test_rounds_queued_ahead_decide_on_the_device(name, mode, monkeypatch):
    """Over the windows the level loop queues rounds ahead of the host; whether a round fires the next level, settles, finishes a
    step that ran out of launches, or idles because the run is over is decided on the device from the exchanged sums.  A budget of
    one launch per round makes every step run out; batches of 1 and 3 rounds move the points where the host looks; "host_driven"
    is the loop with one synchronisation per level (OCRGPU_RX_NO_AHEAD), which must agree."""
    if "budget1" in mode:
        monkeypatch.setenv("OCRGPU_RX_LAUNCHES", "1")
    if "budget2" in mode:
        monkeypatch.setenv("OCRGPU_RX_LAUNCHES", "2")
    if "ahead1" in mode:
        monkeypatch.setenv("OCRGPU_RX_AHEAD", "1")
    if "ahead3" in mode:
        monkeypatch.setenv("OCRGPU_RX_AHEAD", "3")
    if "host_driven" in mode:
        monkeypatch.setenv("OCRGPU_RX_NO_AHEAD", "1")
    s = ALL[name]
    parts = D.build_lockstep([D.Partition(s, r, 3) for r in range(3)])
    D.connect_peer_windows(parts)
    levels = D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    want = O.run_restatement(s)
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(want) == []
    assert levels == int(want.level.max()) + 1


@pytest.mark.parametrize("world", [2, 4])
def test_windows_one_exchange_per_level_rearmed(world):
    from ocr_amd.workloads import LayeredPartition
    n, width = 60_000, 1500
    ref = O.run_restatement(layered_dag(n, width))
    parts = [LayeredPartition(n, width, r, world, allreduce_sum=None) for r in range(world)]
    total_leaves = sum(len(p.leaves) for p in parts)
    for p in parts:
        p.n_leaves = total_leaves
        p.build()
    D.connect_peer_windows(parts)
    D._lockstep_exchange(parts)
    for p in parts:
        p.eng.snapshot()
    for rep in range(2):
        for p in parts:
            p.eng.restore()
        parts[0].satisfy_start()
        levels = D.resolve_lockstep_c(parts, no_cascade=True)
        assert levels == int(ref.level.max()) + 1
        for p in parts:
            assert sum(p.eng.stats().err) == 0, list(p.eng.stats().err)
            lv, val = p.check()
            ids = np.r_[p.mine, n] if p.rank == 0 else p.mine
            assert np.array_equal(lv, ref.level[ids]) and np.array_equal(val, ref.val[ids])
    for p in parts:
        p.eng.close()


@pytest.mark.parametrize("host_driven", [False, True])
def test_no_cascade_with_too_few_launches(host_driven, monkeypatch):
    """OCRGPU_RX_NO_CASCADE fires the next level without a settle round.  With ONE launch per level a partition whose launch went
    into the last level's arrivals (here: the leaves' DECRs of the latch on partition 0) has not fired when the others' records of
    the level arrive.  The host-driven loop finishes its step with polls first and stays exact; rounds queued ahead go on when ALL
    partitions are behind alike (nobody has sent the level yet) and otherwise notice (the headers say who is behind) and stop
    with an error instead of firing EDTs a level early: exact or refused, never silently wrong."""
    from ocr_amd.workloads import LayeredPartition
    monkeypatch.setenv("OCRGPU_RX_LAUNCHES", "1")
    if host_driven:
        monkeypatch.setenv("OCRGPU_RX_NO_AHEAD", "1")
    n, width, world = 60_000, 1500, 2
    ref = O.run_restatement(layered_dag(n, width))
    parts = [LayeredPartition(n, width, r, world, allreduce_sum=None) for r in range(world)]
    total_leaves = sum(len(p.leaves) for p in parts)
    for p in parts:
        p.n_leaves = total_leaves
        p.build()
    D.connect_peer_windows(parts)
    D._lockstep_exchange(parts)
    parts[0].satisfy_start()
    try:
        levels = D.resolve_lockstep_c(parts, no_cascade=True)
    except E.OcrGpuError as ex:          # queued ahead only, and only with that reason: never a silently wrong record
        assert not host_driven and "had not fired its level" in str(ex)
    else:
        assert levels == int(ref.level.max()) + 1
        for p in parts:
            lv, val = p.check()
            ids = np.r_[p.mine, n] if p.rank == 0 else p.mine
            assert np.array_equal(lv, ref.level[ids]) and np.array_equal(val, ref.val[ids])
    for p in parts:
        p.eng.close()


def test_a_partition_that_does_not_arrive_fails_the_round(monkeypatch):
    """two partitions connected, only one of them exchanges: its wait gives up after OCRGPU_PEER_TIMEOUT_MS and the call fails"""
    monkeypatch.setenv("OCRGPU_PEER_TIMEOUT_MS", "50")
    s = ALL["layered_1k"]
    parts = [D.Partition(s, r, 2) for r in range(2)]
    D.connect_peer_windows(parts)
    with pytest.raises(E.OcrGpuError, match="did not arrive"):
        parts[0].eng.exchange()
    # the level may be partly applied: the worker-loop calls refuse until a snapshot is restored or fresh windows are joined
    for call in (lambda: parts[0].eng.step(1), lambda: parts[0].eng.exchange(), lambda: parts[0].eng.resolve()):
        with pytest.raises(E.OcrGpuError, match="gave up earlier"):
            call()
    parts[1].eng.step(1)                   # the partition that did not wait is not affected
    for p in parts:
        p.eng.close()


def test_windows_can_be_dropped_and_made_again(monkeypatch):
    """peer_fini frees the window; a new export / connect starts from round 0 with clean arrival words -- also after a wait
    that gave up (the give-up is sticky only for the window it happened on)"""
    monkeypatch.setenv("OCRGPU_PEER_TIMEOUT_MS", "50")
    s = ALL["layered_1k"]
    parts = D.build_lockstep([D.Partition(s, r, 2) for r in range(2)])
    D.connect_peer_windows(parts)
    with pytest.raises(E.OcrGpuError, match="did not arrive"):
        parts[0].eng.exchange()                    # partition 1 never comes: partition 0's window is dead now
    for p in parts:
        p.eng.peer_fini()
    D.connect_peer_windows(parts)
    D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    for p in parts:
        assert sum(p.eng.stats().err) == 0
        p.eng.close()
    assert res.diff(O.run_restatement(s)) == []


def test_handles_of_another_shape_are_refused():
    s = ALL["layered_1k"]
    a, b = D.Partition(s, 0, 2), D.Partition(s, 1, 2)
    ha, hb = a.eng.peer_export(), b.eng.peer_export()
    with pytest.raises(E.OcrGpuError):
        a.eng.peer_connect([hb, ha])              # location order
    with pytest.raises(E.OcrGpuError):
        a.eng.exchange()                          # no transport yet
    a.eng.peer_connect([ha, hb])
    with pytest.raises(E.OcrGpuError):
        a.eng.peer_connect([ha, hb])              # once
    a.eng.close()
    b.eng.close()


def _worker(rank, world, port, n, width, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ocr_amd import distributed as DD
    from ocr_amd.workloads import LayeredPartition

    def allreduce_sum(x):
        t = torch.tensor([int(x)], dtype=torch.int64)
        dist.all_reduce(t)
        return int(t.item())

    out = None
    try:
        part = LayeredPartition(n, width, rank, world, allreduce_sum=allreduce_sum).build()
        DD.connect_peer_windows_dist(part, dist)
        rin, act, sent, queued = part.eng.exchange()          # what construction sent across (nothing for this DAG)
        assert (rin, sent) == (0, 0)
        part.eng.snapshot()
        for _ in range(2):
            part.eng.restore()
            part.satisfy_start()
            dist.barrier()
            levels = part.eng.resolve_exchange(no_cascade=True)
            lv, val = part.check()
            ids = np.r_[part.mine, n] if rank == 0 else part.mine
            out = (rank, ids, lv, val, levels, list(part.eng.stats().err))
        dist.barrier()
        part.eng.close()
    except Exception as ex:  # noqa: BLE001 -- the parent reports it
        out = (rank, repr(ex))
    dist.destroy_process_group()
    q.put(out)


@pytest.mark.parametrize("world,ahead", [(2, None), (3, None), (2, "1"), (3, "2")])
def test_processes_sharing_one_gpu_windows_over_ipc(world, ahead, monkeypatch):
    """(ahead: rounds per batch -- small batches make the processes look at the stop word at many different moments; each
    must still queue exactly as many rounds as the others, or rounds would pair up wrongly across processes)"""
    if ahead:
        monkeypatch.setenv("OCRGPU_RX_AHEAD", ahead)
    width = 1200 if world == 3 else 1000
    n = 36_000 if world == 3 else 40_000
    ref = O.run_restatement(layered_dag(n, width))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, width, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for res in results:
        assert len(res) == 6, res
        rank, ids, lv, val, levels, err = res
        assert sum(err) == 0
        assert levels == n // width + 1
        assert np.array_equal(lv, ref.level[ids]) and np.array_equal(val, ref.val[ids])


def test_rounds_queued_ahead_size_their_launch_budget(monkeypatch):
    """an EDT -> EDT DAG needs ONE step launch per round (its arrivals set nothing off); the loop starts every round with three and
    follows what the device reports -- launches with nothing to do, rounds whose step ran out -- down to one.  Same record either way."""
    s = layered_dag(400_000, 2_000)          # 200 levels: enough batches of 8 rounds for the budget to settle
    want = O.run_restatement(s)
    launches = {}
    for fixed in ("3", None):
        if fixed:
            monkeypatch.setenv("OCRGPU_RX_LAUNCHES", fixed)
        else:
            monkeypatch.delenv("OCRGPU_RX_LAUNCHES", raising=False)
        parts = D.build_lockstep([D.Partition(s, r, 2) for r in range(2)])
        D.connect_peer_windows(parts)
        before = [int(p.eng.stats().launches) for p in parts]
        D.resolve_lockstep_c(parts)
        launches[fixed] = sum(int(p.eng.stats().launches) - b for p, b in zip(parts, before))
        res = D.gather_result_lockstep(s, parts)
        errs = [list(p.eng.stats().err) for p in parts]
        for p in parts:
            p.eng.close()
        assert all(sum(e) == 0 for e in errs), errs
        assert res.diff(want) == []
    assert launches[None] < 0.6 * launches["3"], launches


@pytest.mark.parametrize("host_driven", [False, True])
def test_no_cascade_outside_its_class_is_refused_not_mislabelled(host_driven, monkeypatch):
    """OCRGPU_RX_NO_CASCADE is the caller's promise that nothing that ARRIVES sets off records for other partitions.  The tiled
    Cholesky breaks it at its tail when the sink latch and the sink EDT live on different partitions (the last task's output event
    -> latch -> sink EDT: an event whose waiter lives elsewhere): the sink would fire a level late.  The engine notices records sent
    while a round that is still to fire its level applies what arrived, and stops: exact or refused, never silently wrong."""
    from ocr_amd.opstream import cholesky_dag
    if host_driven:
        monkeypatch.setenv("OCRGPU_RX_NO_AHEAD", "1")
    refused = []
    for T, world in ((8, 2), (16, 2), (16, 3), (24, 2), (32, 3), (32, 4), (48, 2)):
        s = cholesky_dag(T)
        parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
        D.connect_peer_windows(parts)
        try:
            D.resolve_lockstep_c(parts, no_cascade=True)
        except E.OcrGpuError as ex:
            assert "set off records for other partitions" in str(ex)
            refused.append((T, world))
        else:
            res = D.gather_result_lockstep(s, parts)
            assert res.diff(O.run_restatement(s)) == [], (T, world)
        for p in parts:
            p.eng.close()
    assert (16, 2) in refused and (8, 2) not in refused, refused
