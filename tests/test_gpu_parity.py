"""Parity tests proper: the HIP engine, driven through the C-ABI, must reproduce bit for bit the
result records of the reference (golden fixtures) and of the CPU restatement on the same streams."""
import json
import os

import numpy as np
import pytest

import streams
from ocr_amd import engine
from ocr_amd.opstream import Result, layered_dag
from oracle import ref as O

pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


@pytest.mark.parametrize("name", sorted(ALL))
@pytest.mark.parametrize("nanny", [True, False])
def test_engine_matches_golden_and_restatement(name, nanny, golden_dir):
    s = ALL[name]
    r, st = engine.run_stream(s, flags=engine.F_NANNY if nanny else 0)
    assert sum(st.err) == 0, f"device flagged errors {list(st.err)}"
    ref = O.run_restatement(s)
    assert r.diff(ref) == []
    with open(os.path.join(golden_dir, "index.json")) as f:
        ent = json.load(f)[name]
    assert f"{r.digest():016x}" == ent["digest"]
    assert r.n_fired == ent["n_fired"] == st.fired
    if ent.get("files"):
        assert r.diff(Result.from_file(os.path.join(golden_dir, name + ".res"))) == []


def test_fired_log_is_the_per_level_set_record():
    s = ALL["layered_1k"]
    p = engine.StreamProgram(s).build()
    st = p.resolve()
    ids, ls = p.eng.read_fired()
    r = p.result()
    assert len(ids) == st.fired == 1001 and len(ls) == st.levels + 1 == 12
    for L in range(int(st.levels)):
        got = np.sort(ids[int(ls[L]):int(ls[L + 1])])
        want = np.nonzero(r.level == L)[0]
        assert np.array_equal(got, want)
    p.eng.close()


def test_restore_rearms_a_built_dag():
    s = ALL["layered_1k"]
    p = engine.StreamProgram(s)
    # build everything except the final start satisfy, snapshot, then run twice
    main = s.main
    s2 = type(s)(s.edt, s.evt, s.db_init, main[:-1], s.script, s.sink_edt)
    p = engine.StreamProgram(s2).build()
    p.eng.snapshot()
    ref = O.run_restatement(s)
    for _ in range(3):
        p.eng.restore()
        p.eng.event_satisfy_slot(p._guids(main[-1:]["a"]))
        st = p.resolve()
        assert st.fired == 1001
        assert p.result().diff(ref) == []
    p.eng.close()


@pytest.mark.parametrize("n,width", [(200_000, 100), (1_000_000, 10_000), (1_000_000, 250_000)])
def test_layered_medium_vs_restatement(n, width):
    s = layered_dag(n, width)
    r, st = engine.run_stream(s, flags=0)
    assert sum(st.err) == 0
    assert st.fired == n + 1 and st.levels == n // width + 1
    assert st.edt_sat == s.meta["n_satisfactions"] - s.meta["n_leaves"]
    assert r.diff(O.run_restatement(s)) == []


def _narrow_wide_stream(wide=300_000):
    """start -> A0 (1 EDT) -> B (wide EDTs, each reads A0's datablock) -> latch -> A1 (reads B[0]'s datablock) -> B' (wide, each
    reads A1's) -> latch' -> sink: narrow and wide levels in turn, with a datablock VALUE crossing every transition."""
    import numpy as np
    from ocr_amd.opstream import (BODY_F_OUTEVT, BODY_F_SINK, BODY_HASH, BODY_NOP, EDT_DT, EVT_DT, EVT_LATCH, EVT_ONCE, LATCH_DECR,
                                  MODE_NULL, MODE_RO, OP_ADD_DEP, OP_EDT_CREATE, OP_EVT_CREATE, OP_SATISFY, OP_DT, REF_EDT, REF_EVT,
                                  REF_OUTEVT, OpStream, make_ops, ref)
    W = wide
    A0, B0, A1, B1, SINK = 0, 1, 1 + W, 2 + W, 2 + 2 * W
    n = SINK + 1
    edt = np.zeros(n, dtype=EDT_DT)
    edt["depc"][:] = 1
    edt["body"][:] = BODY_HASH | BODY_F_OUTEVT
    edt["depc"][A1] = 2
    edt["body"][SINK] = BODY_NOP | BODY_F_SINK
    evt = np.zeros(3, dtype=EVT_DT)
    evt["kind"] = [EVT_ONCE, EVT_LATCH, EVT_LATCH]
    evt["arg"] = [0, W, W]
    b0, b1 = np.arange(B0, B0 + W), np.arange(B1, B1 + W)
    main = np.concatenate([
        make_ops(OP_EVT_CREATE, [0, 1, 2]),
        make_ops(OP_EDT_CREATE, np.arange(n)),
        make_ops(OP_ADD_DEP, [ref(REF_EVT, 0)], b=ref(REF_EDT, A0), slot=0, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, np.full(W, ref(REF_OUTEVT, A0), dtype=np.uint64), b=ref(REF_EDT, b0), slot=0, mode=MODE_RO),
        make_ops(OP_ADD_DEP, ref(REF_OUTEVT, b0[1:]), b=ref(REF_EVT, 1), slot=LATCH_DECR, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, [ref(REF_OUTEVT, b0[0])], b=ref(REF_EDT, A1), slot=1, mode=MODE_RO),
        make_ops(OP_ADD_DEP, [ref(REF_OUTEVT, b0[0])], b=ref(REF_EVT, 1), slot=LATCH_DECR, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, [ref(REF_EVT, 1)], b=ref(REF_EDT, A1), slot=0, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, np.full(W, ref(REF_OUTEVT, A1), dtype=np.uint64), b=ref(REF_EDT, b1), slot=0, mode=MODE_RO),
        make_ops(OP_ADD_DEP, ref(REF_OUTEVT, b1), b=ref(REF_EVT, 2), slot=LATCH_DECR, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, [ref(REF_EVT, 2)], b=ref(REF_EDT, SINK), slot=0, mode=MODE_NULL),
        make_ops(OP_SATISFY, [ref(REF_EVT, 0)], b=0, slot=0),
    ])
    return OpStream(edt, evt, np.zeros(0, dtype=np.uint64), main, np.zeros(0, dtype=OP_DT), SINK, name=f"narrow_wide_{W}")


@pytest.mark.parametrize("nanny", [False, True], ids=["fast", "nanny"])
def test_values_cross_narrow_and_wide_levels(nanny):
    """Bodies of a narrow level are run one launch later by idle workgroups, a wide level runs its own, and a launch that only
    pays what is owed sits between a narrow level and a wide one (DESIGN.md s4 "Body"): the datablock checksums must not care."""
    s = _narrow_wide_stream()
    r, st = engine.run_stream(s, flags=engine.F_NANNY if nanny else 0)
    assert sum(st.err) == 0 and st.fired == s.n_edt
    assert r.diff(O.run_restatement(s)) == []
