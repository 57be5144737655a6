"""Datablock acquire modes on the ready path on the MI355X engine: OCRGPU_F_LOCKABLE_DB against the
CPU restatement's Lockable mode, which tests/test_oracle_lockable.py pins on the reference runtime itself
(datablockfactory=Lockable, 1 and 4 workers).  Who wins a contended datablock is a race in the reference and here, so
`level` is held to the invariants of tests/lockable.py; everything else in the record is bit-exact."""
import numpy as np
import pytest

import lockable as L

pytestmark = pytest.mark.gpu


def _restatement(s, lockable=False):
    from oracle.ref import run_restatement
    return run_restatement(s, lockable=lockable)


@pytest.mark.parametrize("nanny", [True, False], ids=["nanny", "fast"])
def test_lockable_streams_match_restatement(nanny):
    from ocr_amd.engine import F_LOCKABLE_DB, F_NANNY, run_stream
    for s in L.all_lockable():
        base = _restatement(s)
        r, st = run_stream(s, flags=F_LOCKABLE_DB | (F_NANNY if nanny else 0))
        assert sum(st.err) == 0, (s.name, list(st.err))
        assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all() and (r.db_val == base.db_val).all(), s.name
        L.check_levels(s, r.level, base.level)


def test_regular_datablocks_do_not_arbitrate():
    """Without the flag the same streams run as under the Regular factory: levels from the dependences alone, bit-exact."""
    from ocr_amd.engine import F_NANNY, run_stream
    for s in L.all_lockable()[:8]:
        base = _restatement(s)
        r, st = run_stream(s, flags=F_NANNY)
        assert sum(st.err) == 0 and not base.diff(r), s.name


def test_symmetric_contenders_take_one_level_each():
    from ocr_amd.engine import F_LOCKABLE_DB, run_stream
    for n in (2, 7, 65, 300):   # 65 / 300: contenders spread over several waves and workgroups
        s = L.ew_contenders(n)
        r, st = run_stream(s, flags=F_LOCKABLE_DB)
        assert sum(st.err) == 0
        lv = np.sort(r.level[:n])
        assert (lv == np.arange(n)).all(), (n, lv[:10])
    s = L.ordered_two_dbs()
    r, _ = run_stream(s, flags=F_LOCKABLE_DB)
    assert (np.sort(r.level[:6]) == np.arange(6)).all()
    # the restatement agrees on the multiset
    ref = _restatement(s, lockable=True)
    assert (np.sort(ref.level) == np.sort(r.level)).all()


def test_rearm_restores_lock_state():
    """snapshot / restore (the bench's re-arm) bring the lock words and waiter lists back."""
    from ocr_amd.engine import F_LOCKABLE_DB, StreamProgram
    s = L.random_lockable(3, n=120)
    base = _restatement(s)
    p = StreamProgram(s, flags=F_LOCKABLE_DB).build()
    p.eng.snapshot()          # the start event is satisfied: the first EDTs already hold their datablocks
    for _ in range(3):
        p.eng.restore()
        st = p.eng.resolve()
        assert sum(st.err) == 0
        r = p.result()
        assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all()
        L.check_levels(s, r.level, base.level)
    p.eng.close()


def test_host_bodies_release_through_the_api():
    """OCRBODY_HOST EDTs contending for one datablock in EW mode: each is handed out alone, and only after the previous
    holder was released with ocrgpu_edt_release_dbs."""
    from ocr_amd.engine import F_LOCKABLE_DB, Engine
    from ocr_amd.opstream import EVT_ONCE, MODE_EW, MODE_NULL
    BODY_HOST = 3   # OCRBODY_HOST (include/ocr_opstream.h)
    n = 9
    eng = Engine(max_edt=n, max_evt=4, max_db=2, max_slots=4, max_edges=64, flags=F_LOCKABLE_DB)
    start = eng.event_create([EVT_ONCE], [0])
    db = eng.db_create([5])
    g, _, _ = eng.edt_create([2] * n, [BODY_HOST] * n, param0=np.arange(n))
    eng.add_dependence(np.repeat(start, n), g, [0] * n, [MODE_NULL] * n)
    eng.add_dependence(np.repeat(db, n), g, [1] * n, [MODE_EW] * n)
    eng.event_satisfy_slot(start, [0], [0])
    seen = []
    for _ in range(4 * n):
        eng.step(fire_levels=1)
        p0, dc, df, dg, dv = eng.pop_ready()
        if len(p0) == 0:
            if len(seen) == n:
                break
            continue
        assert len(p0) == 1, "two exclusive holders were handed out together"
        assert dv[df[0] + 1] == 5   # depv[1] is the datablock (its guid column is the tag the creator chose: none here)
        seen.append(int(p0[0]))
        eng.edt_release_dbs(g[[int(p0[0])]])
    assert sorted(seen) == list(range(n))
    st = eng.stats()
    assert sum(st.err) == 0
    eng.close()


def test_contended_layered_dag_200k():
    """config-2-shaped DAG, 200 000 EDTs in 100 layers, every EDT also wants one of 512 shared datablocks in a random mode:
    the dependences alone give 101 levels, the arbitration several hundred."""
    from ocr_amd.engine import F_LOCKABLE_DB, run_stream
    from ocr_amd.opstream import lockable_layered
    s = lockable_layered(200_000, 2000, 512)
    base = _restatement(s)
    r, st = run_stream(s, flags=F_LOCKABLE_DB)
    assert sum(st.err) == 0 and int(st.fired) == s.n_edt
    assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all() and (r.db_val == base.db_val).all()
    L.check_levels(s, r.level, base.level)
    assert int(st.levels) > int(base.level.max()) + 1


def test_too_many_preslots_is_refused():
    """The datablock walk is quadratic in depc: under OCRGPU_F_LOCKABLE_DB an EDT with more than 4096 pre-slots is a
    malformed op (counted, no guid), not a kernel that never ends."""
    from ocr_amd.engine import F_LOCKABLE_DB, Engine
    eng = Engine(max_edt=4, max_evt=4, max_db=2, max_slots=20000, max_edges=64, flags=F_LOCKABLE_DB)
    g, _, _ = eng.edt_create([4096, 4097], [0, 0])
    assert g[0] != 0 and g[1] == 0
    assert eng.stats().err[4] == 1      # ERR_BAD_OP
    eng.close()


def test_config_twins_bit_exact_under_lockable():
    """Cholesky / stencil / layered twins: every hand-over is ordered by dependences, so the Lockable record is fixed
    (the reference agrees at 1 and 4 workers, tests/test_oracle_lockable.py) and the engine must match it bit for bit,
    levels included -- with datablocks arriving through events and FWD0 bodies rather than direct dependences."""
    from ocr_amd import opstream as op
    from ocr_amd.engine import F_LOCKABLE_DB, F_NANNY, run_stream
    for s in (op.cholesky_dag(8), op.cholesky_dag(32), op.stencil_dag(16, 10), op.layered_dag(1000, 100)):
        want = _restatement(s, lockable=True)
        for fl in (F_LOCKABLE_DB, F_LOCKABLE_DB | F_NANNY):
            r, st = run_stream(s, flags=fl)
            assert sum(st.err) == 0 and want.diff(r) == [], (s.name, fl)


@pytest.mark.parametrize("nanny", [True, False], ids=["nanny", "fast"])
def test_fuzz_corpus_bit_exact_under_lockable(nanny):
    """Every event kind, lazy frontier registration, channels, scripts, unfired EDTs -- with each EDT going through the
    acquisition walk before it is made ready.  These programs never contend, so the record is the Regular one."""
    import streams
    from ocr_amd.engine import F_LOCKABLE_DB, F_NANNY, run_stream
    for seed in range(streams.FUZZ_SEEDS):
        for es in (False, True):
            s = streams.random_stream(seed, 36, eager_safe=es)
            want = _restatement(s, lockable=True)
            r, st = run_stream(s, flags=F_LOCKABLE_DB | (F_NANNY if nanny else 0))
            assert sum(st.err) == 0 and want.diff(r) == [], (seed, es, list(st.err), want.diff(r)[:3])
