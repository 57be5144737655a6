"""The barrier-free resolve (k_async: one resident kernel, the level carried as data -- DESIGN.md s4) must leave the very record the
level-synchronous path leaves: levels, HASH values, event states, datablocks, the fired log grouped by level and the statistics.
Every stream is run three ways -- default (async when the DAG allows), OCRGPU_F_NO_ASYNC, and the CPU restatement -- and the golden
digests recorded from the reference runtime are the fourth witness."""
import json
import os

import numpy as np
import pytest

import streams
from ocr_amd import engine
from ocr_amd.opstream import layered_dag, stencil_dag, powerlaw_dag
from oracle import ref as O

pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


def takes_barrier_free_path(s):
    """The class of programs k_async takes, restated from the op stream alone: ONCE / LATCH events, HASH / NOP bodies without scripts,
    every registered dependence either an EDT pre-slot whose payload is known at registration (DB_MODE_NULL, or the output event of a
    HASH / NOP EDT) or a LATCH DECR slot, no datablock of an EDT that has not run yet handed around by the main program."""
    from ocr_amd.opstream import (BODY_HASH, BODY_NOP, EVT_LATCH, EVT_ONCE, LATCH_DECR, MODE_NULL, OP_ADD_DEP, OP_SATISFY, REF_EDT, REF_EVT,
                                  REF_OUTDB, REF_OUTEVT, ref_cls, ref_idx)
    if len(s.script) or s.db_words:
        return False
    if len(s.evt) and not np.isin(s.evt["kind"], [EVT_ONCE, EVT_LATCH]).all():
        return False
    if not np.isin(s.edt["body"] & 0xFF, [BODY_NOP, BODY_HASH]).all():
        return False
    for op in s.main:
        a, b, mode, slot = int(op["a"]), int(op["b"]), int(op["mode"]), int(op["slot"])
        if op["op"] == OP_SATISFY and ref_cls(b) == REF_OUTDB:
            return False
        if op["op"] != OP_ADD_DEP:
            continue
        if ref_cls(a) == REF_OUTDB:
            return False
        if ref_cls(a) not in (REF_EVT, REF_OUTEVT):
            continue                                     # NULL / datablock sources are immediate satisfies
        if ref_cls(b) == REF_EDT:
            if mode != MODE_NULL and ref_cls(a) != REF_OUTEVT:
                return False
        elif ref_cls(b) == REF_EVT:
            if s.evt["kind"][ref_idx(b)] != EVT_LATCH or slot != LATCH_DECR:
                return False
        else:
            return False
    return True



def _run(s, flags):
    p = engine.StreamProgram(s, flags=flags).build()
    st = p.resolve()
    r = p.result()
    ids, ls = p.eng.read_fired()
    p.eng.close()
    return r, st, ids, ls


@pytest.mark.parametrize("name", sorted(ALL))
def test_both_paths_leave_the_same_record(name, golden_dir):
    s = ALL[name]
    ra, sa, ida, lsa = _run(s, 0)
    rs, ss, ids, lss = _run(s, engine.F_NO_ASYNC)
    assert ss.async_runs == 0
    assert sum(sa.err) == 0 and sum(ss.err) == 0
    assert ra.diff(rs) == [] and ra.diff(O.run_restatement(s)) == []
    with open(os.path.join(golden_dir, "index.json")) as f:
        assert f"{ra.digest():016x}" == json.load(f)[name]["digest"]
    assert (sa.levels, sa.fired, sa.edt_sat, sa.evt_sat) == (ss.levels, ss.fired, ss.edt_sat, ss.evt_sat)
    assert list(lsa) == list(lss)                      # same level boundaries ...
    for L in range(len(lsa) - 1):                      # ... and the same SET in every level (order within a level is nobody's contract)
        assert np.array_equal(np.sort(ida[int(lsa[L]):int(lsa[L + 1])]), np.sort(ids[int(lss[L]):int(lss[L + 1])]))
    max_row = 0
    from ocr_amd.opstream import OP_ADD_DEP, REF_OUTEVT, ref_cls, ref_idx
    m = (s.main["op"] == OP_ADD_DEP) & (ref_cls(s.main["a"]) == REF_OUTEVT)
    if m.any():
        max_row = int(np.bincount(ref_idx(s.main["a"][m]).astype(np.int64)).max())
    # (a DAG whose EDTs have output events with more than 64 waiters is left to the level-synchronous path, which spreads such rows
    # over the grid: a block of long rows on one wave is a chain of dependent passes -- ocrgpu.hip AMAX_ROW)
    assert sa.async_runs == (1 if takes_barrier_free_path(s) and max_row <= 64 and sa.fired else 0)


def test_streams_outside_the_class_stay_level_synchronous():
    for name in ("sticky_early_and_late", "counted_events", "channel_events", "forward_and_nop", "idem_double_satisfy"):
        _, st, _, _ = _run(ALL[name], 0)
        assert st.async_runs == 0, name


@pytest.mark.parametrize("n,width", [(200_000, 100), (1_000_000, 10_000), (1_000_000, 50_000)])
def test_layered_medium(n, width):
    s = layered_dag(n, width)
    r, st, ids, ls = _run(s, 0)
    assert st.async_runs == 1 and sum(st.err) == 0
    assert st.fired == n + 1 and st.levels == n // width + 1
    assert st.edt_sat == s.meta["n_satisfactions"] - s.meta["n_leaves"]
    ref = O.run_restatement(s)
    assert r.diff(ref) == []
    for L in range(int(st.levels)):
        assert np.array_equal(np.sort(ids[int(ls[L]):int(ls[L + 1])]), np.nonzero(ref.level == L)[0])


def test_stencil_and_powerlaw_medium(monkeypatch):
    monkeypatch.setenv("OCRGPU_ASYNC", "2")   # whenever the DAG allows, however wide it starts
    for s in (stencil_dag(128, 12), powerlaw_dag(400_000)):
        r, st, _, _ = _run(s, 0)
        assert sum(st.err) == 0 and r.diff(O.run_restatement(s)) == []
        # (the power-law DAG has output events with thousands of waiters: those keep it on the level-synchronous path)
        assert st.async_runs == (1 if s.meta["kind"] == "stencil" else 0)


def test_rearm_and_resolve_again():
    s = ALL["layered_100k"]
    p = engine.StreamProgram(s.with_main(s.main[:-1])).build()
    p.eng.snapshot()
    ref = O.run_restatement(s)
    for k in range(4):
        p.eng.restore()
        p.eng.event_satisfy_slot(p._guids(s.main[-1:]["a"]))
        st = p.resolve()
        assert st.async_runs == k + 1 and st.fired == s.n_edt and sum(st.err) == 0
        assert p.result().diff(ref) == []
    p.eng.close()


def test_values_travel_between_waves():
    """two layers whose second reads two values of the first each: the consumers' bodies meet granules published by other waves"""
    from ocr_amd.opstream import EVT_ONCE, MODE_NULL, Builder
    b = Builder("two_rounds")
    start = b.evt(EVT_ONCE)
    first = [b.edt(1) for _ in range(40)]
    for e in first:
        b.add_dep(start, e, 0, MODE_NULL)
    second = [b.edt(2) for _ in range(40)]
    for i, e in enumerate(second):
        b.add_dep(b.out(first[i]), e, 0)
        b.add_dep(b.out(first[(i + 7) % 40]), e, 1)
    s = b.finish(second, start=start)
    r, st, _, _ = _run(s, 0)
    assert st.async_runs == 1 and r.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("fan,limit,runs", [(64, None, 1), (65, None, 0), (300, "1024", 1), (1024, "4096", 1), (300, "16", 0)])
def test_long_rows_choose_the_path(monkeypatch, fan, limit, runs):
    """one EDT whose output event has `fan` waiters, and a second round behind them: barrier-free up to OCRGPU_ASYNC_MAX_ROW (default 64;
    values below the 32 from which row lengths are tracked count as 32), level by level beyond -- the same record either way"""
    from ocr_amd.opstream import EVT_ONCE, MODE_NULL, Builder
    if limit is not None:
        monkeypatch.setenv("OCRGPU_ASYNC_MAX_ROW", limit)
    b = Builder(f"hub_{fan}")
    start = b.evt(EVT_ONCE)
    hub = b.edt(1)
    b.add_dep(start, hub, 0, MODE_NULL)
    mid = [b.edt(1) for _ in range(fan)]
    for e in mid:
        b.add_dep(b.out(hub), e, 0)
    last = [b.edt(2) for _ in range(fan)]
    for i, e in enumerate(last):
        b.add_dep(b.out(mid[i]), e, 0)
        b.add_dep(b.out(mid[(i * 7 + 3) % fan]), e, 1)
    s = b.finish(last, start=start)
    r, st, _, _ = _run(s, 0)
    assert sum(st.err) == 0 and st.async_runs == runs
    assert r.diff(O.run_restatement(s)) == []


def test_device_digest_is_the_fold_of_the_record():
    """ocrgpu_digest_edts = sum modulo 2^64 of ocrgpu_digest_term(paramv[0], level, value) over the EDTs, whichever worker loop ran"""
    s = ALL["layered_100k"]
    want = None
    for flags in (0, engine.F_NO_ASYNC):
        p = engine.StreamProgram(s, flags=flags).build()
        p.resolve()
        lv, val = p.eng.read_edts(0, s.n_edt)
        ids = np.arange(s.n_edt, dtype=np.uint64)       # StreamProgram gives every EDT its stream index as paramv[0]
        with np.errstate(over="ignore"):
            host = int(engine.digest_term(ids, lv, val).sum(dtype=np.uint64))
        assert p.eng.digest_edts(0, s.n_edt) == host
        assert p.eng.digest_edts(5, 1000) == int(engine.digest_term(ids[5:1005], lv[5:1005], val[5:1005]).sum(dtype=np.uint64))
        assert p.eng.digest_edts(0, 0) == 0
        want = want or host
        assert host == want
        p.eng.close()


def test_sixty_resolves_leave_one_digest():
    """the order in which the resident kernel fires EDTs differs from run to run; the record may not (tools/soak_async.py at test size)"""
    s = ALL["layered_100k"]
    p = engine.StreamProgram(s.with_main(s.main[:-1])).build()
    p.eng.snapshot()
    want = O.run_restatement(s)
    ids = np.arange(s.n_edt, dtype=np.uint64)
    with np.errstate(over="ignore"):
        digest = int(engine.digest_term(ids, want.level[:s.n_edt], want.val[:s.n_edt]).sum(dtype=np.uint64))
    for k in range(60):
        p.eng.restore()
        p.eng.event_satisfy_slot(p._guids(s.main[-1:]["a"]))
        st = p.resolve()
        assert st.async_runs == k + 1 and sum(st.err) == 0
        assert p.eng.digest_edts(0, s.n_edt) == digest, k
    p.eng.close()
