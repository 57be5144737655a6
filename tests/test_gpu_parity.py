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
