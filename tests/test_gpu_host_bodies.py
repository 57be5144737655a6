"""Host-run EDT bodies through the C-ABI: OCRBODY_HOST EDTs are parked when they fire, ocrgpu_pop_ready hands them out
with paramv[0] and depv (guid, value), the caller's epilogue satisfies the output event; creation with caller-chosen
guids (ocrgpu_guid_reserve + OCRGPU_GUIDS_GIVEN).  Expected behaviour is the reference's: an EDT runs once all its
pre-slots are satisfied (hc-task.c:1260), depv[i].guid is the guid that travelled down the dependence chain
(hc-task.c:680-690), the output event fires its waiters with the guid the body returned (hc-task.c:1788-1831)."""
import numpy as np
import pytest

from ocr_amd import engine as E
from ocr_amd import opstream as ops

pytestmark = pytest.mark.gpu

BODY_HOST = 3


def test_pop_ready_and_epilogue():
    eng = E.Engine(max_edt=64, max_evt=16, max_db=16)
    dbs = eng.guid_reserve(E.KIND_DB, 3)
    eng.db_create([1000, 2000, 3000], tag=dbs, guids=dbs)
    edt_g = eng.guid_reserve(E.KIND_EDT, 3)
    oev_g = eng.guid_reserve(E.KIND_EVT_ONCE, 3)
    # A(depc 2: db0, db1) ; B(depc 1: A's output) ; C(depc 8: overflow slots, all from db2 except slot 3 = NULL)
    eng.edt_create([2, 1, 8], [BODY_HOST | ops.BODY_F_OUTEVT] * 3, [111, 222, 333], guids=edt_g, outevt_guids=oev_g)
    null = np.uint64(0)
    eng.add_dependence([dbs[0], dbs[1], oev_g[0]], [edt_g[0], edt_g[0], edt_g[1]], [0, 1, 0], [ops.MODE_RO] * 3)
    src = [dbs[2]] * 8
    src[3] = null
    eng.add_dependence(src, [edt_g[2]] * 8, list(range(8)), [ops.MODE_RO] * 8)
    assert len(eng.pop_ready()[0]) == 0                      # nothing fired yet
    eng.step(1)
    p0, dc, df, dg, dv = eng.pop_ready()
    got = {int(p): (list(map(int, dg[f:f + c])), list(map(int, dv[f:f + c]))) for p, c, f in zip(p0, dc, df)}
    assert set(got) == {111, 333}
    assert got[111] == ([int(dbs[0]), int(dbs[1])], [1000, 2000])
    g333, v333 = got[333]
    assert g333 == [int(dbs[2])] * 3 + [0] + [int(dbs[2])] * 4 and v333 == [3000] * 3 + [0] + [3000] * 4
    assert len(eng.pop_ready()[0]) == 0                      # handed out once
    eng.step(1)
    assert len(eng.pop_ready()[0]) == 0                      # B waits for A's epilogue: the engine ran none
    eng.event_satisfy_slot(oev_g[:1], dbs[1:2])              # A returned db1
    eng.step(1)
    p0, dc, df, dg, dv = eng.pop_ready()
    assert list(p0) == [222] and list(dc) == [1] and int(dg[df[0]]) == int(dbs[1]) and int(dv[df[0]]) == 2000
    st = eng.stats()
    assert sum(st.err) == 0 and st.fired == 3


def test_pop_ready_grows_and_bad_given_guid():
    eng = E.Engine(max_edt=5000, max_evt=4, max_db=4)
    n = 3000
    g = eng.guid_reserve(E.KIND_EDT, n)
    eng.edt_create([0] * n, [BODY_HOST] * n, np.arange(n), guids=g)
    eng.step(1)
    p0, dc, df, dg, dv = eng.pop_ready(max_edts=16, max_deps=4)   # too small: the wrapper retries with the reported sizes
    assert sorted(map(int, p0)) == list(range(n)) and len(dg) == 0
    # an event guid where an EDT guid is expected is a malformed op (err[4]), not a silent registration
    bad = eng.guid_reserve(E.KIND_EVT_ONCE, 1)
    eng.edt_create([0], [BODY_HOST], [7], guids=bad)
    assert eng.stats().err[4] == 1
    assert eng.guid_get(bad)[0] == 0
