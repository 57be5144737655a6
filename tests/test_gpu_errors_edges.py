"""Error behaviour and edge cases of the hot path, through the C-ABI.

The user errors the reference runtime ASSERTs on (it aborts: hc-task.c:1243 double satisfy of a pre-slot,
hc-event.c:1078-1083 waiter on a spent ONCE event, :578-586 STICKY satisfied twice, satisfy of a destroyed event)
cannot be golden vectors -- the reference dies.  The engine and the CPU restatement COUNT them instead
(ocrgpu_stats_t.err[]) and go on; here the two must count the same errors on the same ill-formed programs.  Then
the sizes at the edges of the data layout: empty batches, depc = 0, the inline/overflow pre-slot boundary, wide
EDTs, rows around the BIG_ROW split, capacity errors."""
import numpy as np
import pytest

from ocr_amd import engine
from ocr_amd.opstream import (BODY_HASH, BODY_NOP, EVT_IDEM, EVT_LATCH, EVT_ONCE, EVT_STICKY, LATCH_DECR, MODE_NULL, MODE_RO,
                              Builder)
from oracle import ref as O

pytestmark = pytest.mark.gpu


def _ill(name):
    b = Builder("ill_" + name)
    start = b.evt(EVT_ONCE)
    d = b.db(77)
    ok = b.edt(1, BODY_NOP)
    b.add_dep(start, ok, 0, MODE_RO)
    if name == "double_slot":
        e = b.edt(3, BODY_NOP)
        b.add_dep(d, e, 0)
        b.add_dep(start, e, 2)
        b.satisfy(b.evt(EVT_STICKY), 0)          # a different op class in between: the two add-deps are two batches
        b.add_dep(d, e, 0)
    elif name == "once_late_waiter":
        once = b.evt(EVT_ONCE)
        b.satisfy(once, d)                        # nobody waits: the event fires and is gone
        e = b.edt(1, BODY_NOP)
        b.add_dep(once, e, 0)
    elif name == "sticky_twice":
        st = b.evt(EVT_STICKY)
        b.satisfy(st, d)
        b.satisfy(st, 0)
    elif name == "satisfy_destroyed":
        st = b.evt(EVT_STICKY)
        b.destroy(st)
        b.satisfy(st, d)
    elif name == "destroy_twice":
        st = b.evt(EVT_IDEM)
        b.destroy(st)
        b.satisfy(b.evt(EVT_STICKY), 0)
        b.destroy(st)
    elif name == "slot_out_of_range":
        e = b.edt(2, BODY_NOP)
        b.add_dep(d, e, 2)
        b.add_dep(d, e, 7)
    return b.finish([ok], start)


@pytest.mark.parametrize("name", ["double_slot", "once_late_waiter", "sticky_twice", "satisfy_destroyed", "destroy_twice",
                                  "slot_out_of_range"])
def test_ill_formed_programs_count_the_same_errors(name):
    s = _ill(name)
    ref, stats = O.run_restatement(s, return_stats=True)
    want = [int(x) for x in stats[8:]]
    assert sum(want) > 0, "the program is meant to be ill-formed"
    r, st = engine.run_stream(s, flags=engine.F_NANNY, tolerate_faults=True)
    assert list(st.err) == want
    assert r.level[0] == ref.level[0] == 0          # the well-formed part still runs
    # ... and the call during which the device counted the error says so: OCRGPU_EFAULT, after doing its work
    with pytest.raises(engine.OcrGpuError) as ei:
        engine.run_stream(s, flags=engine.F_NANNY)
    assert ei.value.code == engine.EFAULT and "flagged" in str(ei.value)


def test_empty_batches_and_empty_program():
    eng = engine.Engine(max_edt=4, max_evt=4, max_db=4)
    z64, z32 = np.zeros(0, np.uint64), np.zeros(0, np.uint32)
    assert len(eng.event_create(z32)) == 0 and len(eng.db_create(z64)) == 0
    eng.add_dependence(z64, z64, z32, z32)
    eng.event_satisfy_slot(z64)
    st = eng.resolve()
    assert (st.fired, st.levels, sum(st.err)) == (0, 0, 0)
    eng.close()


def test_depc_zero_fires_at_creation():
    """hc-task.c:1129-1134: an EDT without pre-slots is ready when it is created"""
    eng = engine.Engine(max_edt=8, max_evt=4, max_db=4)
    eng.edt_create([0, 0, 1], [BODY_NOP] * 3, [1, 2, 3])
    st = eng.resolve()
    lv, _ = eng.read_edts(0, 3)
    assert st.fired == 2 and list(lv) == [0, 0, -1]
    eng.close()


@pytest.mark.parametrize("depc", [1, 5, 6, 7, 8, 9, 63, 64, 65, 300, 5000])
def test_pre_slot_layout_boundaries(depc):
    """6 pre-slot words live in the EDT's line, more in the overflow array; bodies gather 8 at a time"""
    b = Builder(f"wide_{depc}")
    start = b.evt(EVT_ONCE)
    dbs = [b.db(1000 + i) for i in range(4)]
    e = b.edt(depc, BODY_HASH)
    for k in range(depc):
        src = [start, dbs[k % 4], 0, dbs[(k * 7) % 4]][k % 4] if k else start
        b.add_dep(src, e, k, MODE_NULL if k % 5 == 4 else MODE_RO)
    tail = b.edt(1, BODY_HASH)
    b.add_dep(b.out(e), tail, 0)
    s = b.finish([tail], start)
    ref = O.run_restatement(s)
    for flags in (engine.F_NANNY, 0):
        r, st = engine.run_stream(s, flags=flags)
        assert sum(st.err) == 0 and r.diff(ref) == []


@pytest.mark.parametrize("fan", [1, 2047, 2048, 2049, 4096, 100_000])
def test_waiter_rows_around_the_big_row_split(fan):
    """one event with `fan` waiters (EDT pre-slots and a latch): rows longer than 2048 are fanned out in pieces"""
    b = Builder(f"fan_{fan}")
    start = b.evt(EVT_STICKY)
    lt = b.evt(EVT_LATCH, fan)
    es = [b.edt(1, BODY_NOP) for _ in range(fan)]
    for e in es:
        b.add_dep(start, e, 0)
        b.add_dep(b.out(e), lt, LATCH_DECR, MODE_NULL)
    last = b.edt(1, BODY_NOP)
    b.add_dep(lt, last, 0)
    s = b.finish([last], None)
    s2 = type(s)(s.edt, s.evt, s.db_init, np.concatenate([s.main, Builder_satisfy(start)]), s.script, s.sink_edt, name=s.name)
    ref = O.run_restatement(s2)
    assert ref.n_fired == fan + 2
    r, st = engine.run_stream(s2, flags=0)
    assert sum(st.err) == 0 and r.diff(ref) == []


def Builder_satisfy(evt):
    from ocr_amd.opstream import OP_DT, OP_SATISFY
    out = np.zeros(1, dtype=OP_DT)
    out[0] = (OP_SATISFY, 0, 0, 0, evt, 0)
    return out


def test_capacity_errors_are_reported_not_faulted():
    eng = engine.Engine(max_edt=4, max_evt=2, max_db=2)
    with pytest.raises(engine.OcrGpuError) as ei:
        eng.edt_create([1] * 5, [BODY_NOP] * 5, list(range(5)))
    assert ei.value.code == -12                     # OCRGPU_ENOMEM
    with pytest.raises(engine.OcrGpuError):
        eng.event_create([EVT_ONCE] * 3)
    with pytest.raises(engine.OcrGpuError):
        eng.db_create([1, 2, 3])
    g, _, _ = eng.edt_create([1] * 4, [BODY_NOP] * 4, list(range(4)))     # still usable after the refusals
    eng.add_dependence([0] * 4, g, [0] * 4, [MODE_RO] * 4)
    assert eng.resolve().fired == 4
    eng.close()


def test_interrupted_resolve_leaves_no_body_owed():
    """ocrgpu_resolve with a level budget stops between launches; bodies are run one launch behind their level, so a
    stopped run must pay what it owes before it returns: every EDT that has a level has its value."""
    import numpy as np
    from ocr_amd import engine, opstream
    from oracle import ref as O
    s = opstream.layered_dag(3000, 10)
    want = O.run_restatement(s)
    p = engine.StreamProgram(s, flags=0).build()
    calls = 0
    while True:
        st = p.eng.resolve(max_levels=1)
        calls += 1
        lv, val = p.eng.read_edts(0, s.n_edt)
        got_lv, got_val = np.full(s.n_edt, -1, np.int32), np.zeros(s.n_edt, np.uint64)
        eo = np.asarray(p.edt_order, dtype=np.int64)
        got_lv[eo], got_val[eo] = lv, val
        fired = got_lv >= 0
        assert (got_lv[fired] == want.level[fired]).all()
        assert (got_val[fired] == want.val[fired]).all(), f"call {calls}: values missing for fired EDTs"
        if int(st.fired) == s.n_edt or calls > 100:
            break
    assert calls > 2 and int(st.fired) == s.n_edt
    assert want.diff(p.result()) == []
    p.eng.close()


def test_api_batch_larger_than_the_record_queue():
    """Every element of one ocrgpu_event_satisfy_slot / ocrgpu_add_dependence call can queue a satisfy record; a batch far
    larger than the queue (sized from the engine's capacities) is applied in chunks with a drain between them: nothing
    is dropped, nothing is flagged (round-1 advice: the drop was silent and the call returned 0)."""
    eng = engine.Engine(max_edt=8, max_evt=8, max_db=8, max_edges=200_000)   # record queue: a few thousand entries
    n = 150_000
    lt = eng.event_create([EVT_LATCH], [0])
    g, _, _ = eng.edt_create([1], [BODY_NOP], [5])
    eng.add_dependence(lt, g, [0], [MODE_NULL])
    eng.event_satisfy_slot(np.repeat(lt, n), None, np.ones(n, np.uint32))             # n INCRs in one call
    assert eng.resolve().fired == 0
    eng.event_satisfy_slot(np.repeat(lt, n), None, np.zeros(n, np.uint32))            # n DECRs: the last one fires it
    st = eng.resolve()
    assert sum(st.err) == 0 and st.fired == 1 and st.evt_sat == 2 * n
    # one add-dependence batch of DB-source edges onto IDEM events, each an immediate satisfy record
    ev = eng.event_create([EVT_IDEM] * 4)
    d = eng.db_create([9])
    eng.add_dependence(np.repeat(d, n), np.tile(ev, n // 4 + 1)[:n], np.zeros(n, np.uint32), np.zeros(n, np.uint32))
    assert sum(eng.stats().err) == 0
    eng.close()


def test_restore_after_the_rows_were_rebuilt():
    """snapshot -> resolve -> add_dependence -> resolve (the CSR is sealed again: the rows of the events still alive move)
    -> restore: the restored rows must not point into the rebuilt layout, and the edge budget goes back too
    (round-1 advice on ocrgpu_restore)."""
    b = Builder("restore_reseal")
    start = b.evt(EVT_ONCE)
    keep = b.evt(EVT_STICKY)                      # stays alive across the first resolve
    es = [b.edt(1, BODY_HASH) for _ in range(40)]
    for e in es:
        b.add_dep(start, e, 0, MODE_RO)
    late = [b.edt(1, BODY_HASH) for _ in range(8)]
    for e in late[:4]:
        b.add_dep(keep, e, 0, MODE_RO)
    s = b.finish(es, None)
    from ocr_amd.opstream import OP_ADD_DEP
    n_adddep = int((s.main["op"] == OP_ADD_DEP).sum())
    eng = engine.Engine(max_edt=s.n_edt, max_evt=s.n_evt, max_db=1, max_slots=int(s.edt["depc"].sum()) + 1,
                        max_edges=n_adddep + 1 + 4)          # room for ONE round of the four late edges, not three
    p = engine.StreamProgram(s, engine=eng)
    p.build()
    g_start, g_keep = p._guids(np.array([start, keep], np.uint64))
    eng.snapshot()
    def run():
        eng.event_satisfy_slot([g_start])
        st = eng.resolve()
        assert sum(st.err) == 0
        return p.result()
    first = run()
    for rounds in range(3):                        # more rounds than the edge capacity would allow if it leaked
        eng.add_dependence(np.repeat(g_keep, 4), p._guids(np.array(late[4:], np.uint64)), [0] * 4, [MODE_RO] * 4)
        eng.event_satisfy_slot([g_keep])           # seals the new rows and fans them out
        eng.resolve()
        eng.restore()
        again = run()
        assert first.diff(again) == [], rounds
    eng.close()
