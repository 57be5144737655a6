"""The C transport (ocrgpu_comm_*, ocrgpu_exchange, ocrgpu_resolve_exchange) through the C-ABI.

One GPU is what the test box has, and RCCL refuses two ranks on one device, so RCCL itself is exercised with a ONE-rank
communicator (unique id, ncclCommInitRank, grouped ncclSend / ncclRecv to self on the engine's stream, content-checked), and the
protocol that runs on top of it -- headers, equal-split regions, device-side apply, the per-level loop -- is exercised
end to end with 2, 3 and 4 partitions in one process, where device-to-device copies stand in for the all-to-all
(ocrgpu_resolve_exchange_local): bit-exact against the single-partition oracle."""
import numpy as np
import pytest
import torch

import streams
from ocr_amd import distributed as D
from ocr_amd import engine as E
from ocr_amd.opstream import layered_dag
from oracle import ref as O

pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


def test_rccl_one_rank_alltoall_and_resolve():
    s = layered_dag(20_000, 500)
    part = D.Partition(s, 0, 1)
    part.build()
    uid = E.comm_get_unique_id()
    assert len(uid) == 128 and any(uid)
    part.eng.comm_init(uid, 0, 1)
    with pytest.raises(E.OcrGpuError):
        part.eng.comm_init(uid, 0, 1)                       # one communicator per engine
    # the transport alone: bytes go through ncclSend / ncclRecv (to self) on the engine's stream
    src = torch.arange(1 << 20, dtype=torch.int64, device="cuda") * 7 + 3
    dst = torch.zeros_like(src)
    part.eng.comm_alltoall(src.data_ptr(), dst.data_ptr(), src.numel() * 8)
    part.eng.sync()
    assert torch.equal(src, dst)
    # one exchange round with nothing to send, then the whole worker loop over the communicator
    rin, act, sent, queued = part.eng.exchange()
    assert (rin, sent) == (0, 0)
    levels = part.eng.resolve_exchange(no_cascade=True)
    res = D.gather_result_lockstep(s, [part])
    want = O.run_restatement(s)
    assert sum(part.eng.stats().err) == 0 and res.diff(want) == [] and levels == int(want.level.max()) + 1
    part.eng.comm_fini()
    part.eng.close()


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name", ["layered_1k", "event_chain", "latch_incr2_decr2", "counted_events", "channel_events", "cholesky_T8",
                                  "stencil_16x16x10", "counted_after_null_preslot", "sticky_early_and_late", "wide_fanin"])
def test_c_loop_is_partition_invariant(name, world):
    s = ALL[name]
    parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
    D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("mode", ["budget1", "polled"])
@pytest.mark.parametrize("name", ["layered_1k", "event_chain", "counted_events", "channel_events", "cholesky_T8", "wide_fanin"])
def test_c_loop_step_that_runs_out_of_launches(name, mode, monkeypatch):
    """A level's launches are queued ahead without a poll (one synchronisation per level); a step that needs more launches than
    were queued says so in its headers, every partition comes back for a settle round, and the step is finished with polls.
    With a budget of ONE launch every step takes that path.  "polled": the two-synchronisation round, for comparison."""
    if mode == "budget1":
        monkeypatch.setenv("OCRGPU_RX_LAUNCHES", "1")
    else:
        monkeypatch.setenv("OCRGPU_RX_POLLED", "1")
    s = ALL[name]
    parts = D.build_lockstep([D.Partition(s, r, 3) for r in range(3)])
    D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("world", [2, 4])
def test_c_loop_one_exchange_per_level(world):
    """the bench's partition-local layered DAG keeps its latch and sink on one rank and has no event -> remote waiter chains
    (may_cascade = False): OCRGPU_RX_NO_CASCADE, one exchange per level; re-armed and resolved twice"""
    from ocr_amd.workloads import LayeredPartition
    n, width = 60_000, 1500
    ref = O.run_restatement(layered_dag(n, width))
    parts = [LayeredPartition(n, width, r, world, allreduce_sum=None) for r in range(world)]
    total_leaves = sum(len(p.leaves) for p in parts)
    for p in parts:
        p.n_leaves = total_leaves
        p.build()
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


def test_workpile_push_pop_steal():
    """hcWorkpilePush / hcWorkpilePop (hc-workpile.c:103-150): POP = tail, newest first; STEAL = head, oldest first"""
    eng = E.Engine(max_edt=16, max_evt=4, max_db=4)            # capacity max_edt + 1024
    assert len(eng.workpile_pop(4)) == 0                        # empty: nothing (the reference returns NULL_GUID)
    g = np.arange(100, 130, dtype=np.uint64)
    eng.workpile_push(g[:10])
    eng.workpile_push(g[10:20])
    assert list(eng.workpile_pop(3)) == [119, 118, 117]
    assert list(eng.workpile_pop(2, steal=True)) == [100, 101]
    eng.workpile_push(g[20:22], back=True)                      # PUSH_WORKPUSHBACKTYPE: to the head
    assert list(eng.workpile_pop(3, steal=True)) == [120, 121, 102]
    rest = eng.workpile_pop(100)
    assert list(rest) == list(range(116, 102, -1))
    assert len(eng.workpile_pop(1, steal=True)) == 0
    big = np.arange(1, 1041, dtype=np.uint64)                   # wraps around the ring, fills it exactly
    eng.workpile_push(big[:600]); assert list(eng.workpile_pop(500, steal=True)) == list(range(1, 501))
    eng.workpile_push(big[600:1040]); eng.workpile_push(np.arange(5000, 5500, dtype=np.uint64))
    with pytest.raises(E.OcrGpuError) as ei:
        eng.workpile_push(np.array([1], np.uint64))
    assert ei.value.code == -12
    out = eng.workpile_pop(2000, steal=True)
    assert list(out) == list(range(501, 1041)) + list(range(5000, 5500))
    # PUSHBACK moves the head down and POP the tail: with the pile never empty both indices walk down one step per round, past
    # zero after capacity rounds -- the pile must keep working as a deque through that (the indices are positions modulo capacity)
    eng.workpile_push(np.array([7, 8], np.uint64))
    expect = [7, 8]
    for k in range(2 * (16 + 1024) + 50):
        eng.workpile_push(np.array([10_000 + k], np.uint64), back=True)
        expect.insert(0, 10_000 + k)
        assert list(eng.workpile_pop(1)) == [expect.pop()], k
    assert list(eng.workpile_pop(10, steal=True)) == expect
    eng.close()
