"""Capacity is what is alive, not what was ever created (ocrgpu_reclaim, ocrgpu_db_destroy): rows of finished EDTs, destroyed
events and destroyed datablocks go on free stacks the create calls serve first, their guids leave the table, registrations of
fired events leave the log (destructTaskHc hc-task.c:721, destructEventHc hc-event.c:185-242, releaseGuid
counted-map-guid.c:540-572, hashtable remove hashtable.c:556-575).  Small cases pin down WHEN a row may go; the churn case
runs ten rounds of a million objects through an engine sized for 1.2 million, every round bit-exact against the restatement."""
import numpy as np
import pytest

from ocr_amd import engine, opstream as ops
from ocr_amd.engine import Engine, OcrGpuError
from oracle import ref as O

pytestmark = pytest.mark.gpu

NOP, HASH, HOST, OUTEVT = ops.BODY_NOP, ops.BODY_HASH, 3, ops.BODY_F_OUTEVT
ONCE, STICKY, LATCH = ops.EVT_ONCE, ops.EVT_STICKY, ops.EVT_LATCH
RO, NULL = ops.MODE_RO, ops.MODE_NULL


def _idx(eng, guids):
    return eng.guid_get(np.atleast_1d(np.asarray(guids, np.uint64))).astype(np.int64) - 1


def test_rows_come_back_and_are_served_first():
    eng = Engine(max_edt=8, max_evt=4, max_db=4, max_edges=32)
    ev = eng.event_create([ONCE])
    g, _, _ = eng.edt_create([1] * 8, [NOP] * 8)
    first = _idx(eng, g)
    assert sorted(first) == list(range(8))
    with pytest.raises(OcrGpuError) as ei:                       # the table is full
        eng.edt_create([0], [NOP])
    assert ei.value.code == -12
    eng.add_dependence(np.repeat(ev, 8), g, [0] * 8, [NULL] * 8)
    eng.event_satisfy_slot(ev)
    st = eng.resolve()
    assert st.fired == 8 and sum(st.err) == 0
    r = eng.reclaim()
    assert r["edts_freed"] == 8 and r["events_freed"] == 1 and r["edts_held"] == 0 and r["edges_kept"] == 0
    assert r["free_edt"] == 8 and r["fresh_edt"] == 0 and r["free_evt"] == 1 and r["fresh_evt"] == 3
    assert not eng.guid_get(g).any() and not eng.guid_get(ev).any()   # released guids do not resolve
    # second generation: same rows, new guids, level numbering starts over
    ev2 = eng.event_create([ONCE])
    assert _idx(eng, ev2)[0] == 0
    g2, _, _ = eng.edt_create([1] * 5, [NOP] * 5)
    assert sorted(_idx(eng, g2)) == sorted(first[-5:]) or set(_idx(eng, g2)) <= set(first)
    assert not set(g2.tolist()) & set(g.tolist())
    eng.add_dependence(np.repeat(ev2, 5), g2, [0] * 5, [NULL] * 5)
    eng.event_satisfy_slot(ev2)
    st = eng.resolve()
    assert sum(st.err) == 0
    lv, _ = eng.read_edts(0, 8)
    assert sorted(lv[_idx(eng, g2)]) == [0] * 5 and (lv >= 0).sum() == 5
    r = eng.reclaim()
    assert r["edts_freed"] == 5 and r["free_edt"] == 8
    eng.close()


def test_what_holds_a_row_back():
    eng = Engine(max_edt=8, max_evt=8, max_db=8, max_edges=32)
    start = eng.event_create([ONCE])
    db = eng.db_create([41, 42])
    # a HASH EDT: its output datablock lives in its line -> the line stays until that datablock is destroyed
    gh, oe_h, od_h = eng.edt_create([1], [HASH | OUTEVT])
    # a host-body EDT with an output event: stays until the host has satisfied that event
    gs, oe_s, _ = eng.edt_create([1], [HOST | OUTEVT])
    # a consumer of both, registered on the output events
    gc, _, _ = eng.edt_create([2], [NOP])
    eng.add_dependence([start[0], start[0], oe_h[0], oe_s[0]], [gh[0], gs[0], gc[0], gc[0]], [0, 0, 0, 1], [NULL, NULL, RO, RO])
    eng.event_satisfy_slot(start)
    eng.step(fire_levels=1)
    r = eng.reclaim()                                   # the host EDT fired but was not handed out: it stays in the ready log
    assert r["edts_freed"] == 0 and r["edts_held"] == 1 and r["ready_kept"] == 1 and r["events_freed"] == 1
    assert r["edges_kept"] == 1                         # the host EDT's output event -> consumer slot 1 is still to come
    p0, dc, df, dg, dv = eng.pop_ready()
    assert len(p0) == 1
    r = eng.reclaim()                                   # handed out and (by contract) returned, but its output event is LIVE
    assert r["edts_freed"] == 0 and r["edts_held"] == 2
    eng.event_satisfy_slot(oe_s, db[:1])                # the body's return value reaches the output event
    st = eng.resolve()
    assert sum(st.err) == 0
    lv, _ = eng.read_edts(0, 3)
    assert lv[_idx(eng, gc)[0]] >= 0                    # the consumer fired (levels restarted at the reclaim)
    r = eng.reclaim()
    assert r["edts_freed"] == 2 and r["edts_held"] == 1 # host EDT + consumer gone; the HASH EDT waits for its datablock
    assert eng.guid_get(od_h)[0] != 0
    eng.db_destroy(od_h)
    assert eng.guid_get(od_h)[0] == 0
    r = eng.reclaim()
    assert r["edts_freed"] == 1 and r["edts_held"] == 0 and r["free_edt"] == 3
    eng.db_destroy(db)
    r = eng.reclaim()
    assert r["free_db"] == 2
    with pytest.raises(OcrGpuError) as ei:              # destroyed twice: unknown guid
        eng.db_destroy(db[:1])
    assert ei.value.code == engine.EFAULT
    eng.close()


def test_live_registrations_and_ready_edts_survive():
    eng = Engine(max_edt=16, max_evt=8, max_db=4, max_edges=64)
    a, b = eng.event_create([ONCE, STICKY])
    g1, _, _ = eng.edt_create([1] * 4, [NOP] * 4)       # wait for a
    g2, oe2, _ = eng.edt_create([1] * 4, [NOP | OUTEVT] * 4)   # wait for b; their output events feed g3
    g3, _, _ = eng.edt_create([4], [NOP])
    eng.add_dependence([a] * 4 + [b] * 4, list(g1) + list(g2), [0] * 8, [NULL] * 8)
    eng.add_dependence(oe2, np.repeat(g3, 4), [0, 1, 2, 3], [NULL] * 4)
    eng.event_satisfy_slot([a])
    eng.step(fire_levels=0)                             # events closed, g1 ready but not fired
    r = eng.reclaim()
    assert r["edts_freed"] == 0 and r["events_freed"] == 1 and r["ready_kept"] == 4 and r["edges_kept"] == 8
    g0, _, _ = eng.edt_create([0] * 2, [NOP] * 2)       # fire at creation: appended behind the kept entries
    st = eng.resolve()
    assert sum(st.err) == 0
    lv, _ = eng.read_edts(0, 11)
    assert list(lv[_idx(eng, g1)]) == [0] * 4 and list(lv[_idx(eng, g0)]) == [0] * 2 and list(lv[_idx(eng, g2)]) == [-1] * 4
    r = eng.reclaim()
    assert r["edts_freed"] == 6 and r["edges_kept"] == 8
    eng.event_satisfy_slot([b])
    st = eng.resolve()
    assert sum(st.err) == 0
    lv, _ = eng.read_edts(0, 11)
    assert list(lv[_idx(eng, g2)]) == [0] * 4 and lv[_idx(eng, g3)[0]] == 1
    eng.event_destroy([b])                              # a STICKY event goes when the program destroys it
    r = eng.reclaim()
    assert r["edts_freed"] == 5 and r["events_freed"] == 1 and r["edges_kept"] == 0 and r["free_edt"] == 11 and r["free_evt"] == 2
    eng.close()


def test_overflow_slots_scripts_and_snapshot():
    eng = Engine(max_edt=5, max_evt=1, max_db=9, max_slots=40, max_edges=64)
    for rnd in range(6):                                # 6 x 4 EDTs of 9 pre-slots through 40 overflow words
        b = ops.Builder()
        dbs = [b.db(100 * rnd + k) for k in range(9)]
        es = [b.edt(9, HASH, outevt=False) for _ in range(4)]   # no output event: a script DECRs the sink latch
        for e in es:
            for k, d in enumerate(dbs):
                b.add_dep(d, e, k, RO)
        s = b.finish(es)
        want = O.run_restatement(s)
        p = engine.StreamProgram(s, engine=eng, reuse=True).build()
        if rnd == 3:
            eng.snapshot()
        st = p.resolve()
        assert sum(st.err) == 0
        assert p.result().diff(want) == []
        eng.db_destroy(p.db_guid)
        r = eng.reclaim()
        assert r["edts_freed"] == 5 and r["events_freed"] == 1 and r["slots_used"] == 0 and r["free_db"] == 9, r
        if rnd == 3:
            with pytest.raises(OcrGpuError):            # rows changed hands since the snapshot
                eng.restore()
    eng.close()


def _churn_stream(n_edt, width, n_db, n_evt, seed):
    """config-2-shaped layered DAG + one more pre-slot per EDT fed by an explicit datablock + n_evt of the edges routed through
    an explicit event of their own (ONCE and STICKY alternating): EDTs, events and datablocks all churn."""
    s = ops.layered_dag(n_edt, width, seed)
    N = n_edt
    edt = s.edt.copy()
    extra_slot = edt["depc"][:N].copy()
    edt["depc"][:N] += 1
    r = ops.splitmix64(seed ^ 0xC0FFEE, N)
    which = (r % np.uint64(n_db)).astype(np.int64)
    main = s.main
    dep = np.nonzero((main["op"] == ops.OP_ADD_DEP) & (ops.ref_cls(main["a"]) == ops.REF_OUTEVT) & (ops.ref_cls(main["b"]) == ops.REF_EDT))[0]
    pick = dep[:: max(1, len(dep) // n_evt)][:n_evt]
    x = 2 + np.arange(len(pick))                          # events 0, 1 are start and the sink latch
    to_evt = ops.make_ops(ops.OP_ADD_DEP, main["a"][pick], b=ops.ref(ops.REF_EVT, x), slot=0, mode=ops.MODE_RO)
    rerouted = main.copy()
    rerouted["a"][pick] = ops.ref(ops.REF_EVT, x)
    evt = np.zeros(2 + len(pick), dtype=ops.EVT_DT)
    evt[:2] = s.evt
    evt["kind"][2:] = np.where(np.arange(len(pick)) % 2 == 0, ONCE, STICKY)
    assert int(main["op"][0]) == ops.OP_EVT_CREATE and int(main["op"][-1]) == ops.OP_SATISFY
    n_head = int((main["op"] == ops.OP_EVT_CREATE).sum())
    new_main = np.concatenate([
        ops.make_ops(ops.OP_EVT_CREATE, np.arange(len(evt))),
        ops.make_ops(ops.OP_DB_CREATE, np.arange(n_db)),
        rerouted[n_head:-1],
        to_evt,
        ops.make_ops(ops.OP_ADD_DEP, ops.ref(ops.REF_DB, which), b=ops.ref(ops.REF_EDT, np.arange(N)), slot=extra_slot, mode=ops.MODE_RO),
        main[-1:],
    ])
    return ops.OpStream(edt, evt, 7000 + np.arange(n_db, dtype=np.uint64), new_main, s.script, s.sink_edt, name="churn")


def test_churn_ten_rounds_of_a_million_objects():
    n_edt, n_db, n_evt = 500_000, 250_000, 250_000
    eng = Engine(max_edt=600_000, max_evt=300_000, max_db=300_000, max_slots=2_000_000, max_edges=4_000_000, guid_slots=4_000_000)
    want_by_seed = {}
    for rnd in range(10):
        seed = ops.SEED_BASE + 0x777 + rnd % 2                          # two programs taking turns: rows change tenants
        if seed not in want_by_seed:
            s = _churn_stream(n_edt, 5000, n_db, n_evt, seed)
            assert s.n_edt + s.n_evt + s.n_db > 1_000_000
            want_by_seed[seed] = (s, O.run_restatement(s))
        s, want = want_by_seed[seed]
        p = engine.StreamProgram(s, engine=eng, reuse=True).build()
        st = p.resolve()
        assert sum(st.err) == 0, (rnd, list(st.err))
        got = p.result()
        assert got.diff(want) == [], rnd
        if rnd:
            assert int(p.edt_index.max()) < 600_000 and int(p.edt_index.min()) >= 0
        # the program's clean-up: datablocks (explicit ones and the EDTs' output datablocks) and the STICKY events
        eng.db_destroy(p.db_guid)
        sticky = p.evt_guid[:s.n_evt][s.evt["kind"] == STICKY]
        eng.event_destroy(sticky)
        r = eng.reclaim()
        assert r["edts_freed"] == s.n_edt and r["events_freed"] == s.n_evt and r["edts_held"] == 0 and r["edges_kept"] == 0, (rnd, r)
        assert r["free_edt"] + r["fresh_edt"] == 600_000 and r["free_evt"] + r["fresh_evt"] == 300_000 and r["free_db"] + r["fresh_db"] == 300_000
        assert r["fresh_edt"] == 600_000 - s.n_edt                     # no row beyond the first round's was ever touched
        assert not eng.guid_get(p.edt_guid[::997]).any()
        print(f"round {rnd}: fired {int(st.fired)} levels {int(st.levels)} reclaimed {r['edts_freed']} EDTs {r['events_freed']} events", flush=True)
    st = eng.stats()
    assert sum(st.err) == 0
    eng.close()
