"""ocrgpu_pop_ready_noalloc (the entry point that allocates nothing: its counters and staging live in a static device
array of the library) against ocrgpu_pop_ready, through the C-ABI: two engines are driven identically, one popped with
each entry point; two calls in a row, the second with a smaller input.  The hand-over order is unspecified (both entry
points say "any order"), so records are compared per task handle.  Collected last (file name)."""
import numpy as np
import pytest

from ocr_amd import engine as E
from ocr_amd.opstream import BODY_F_OUTEVT, MODE_NULL, MODE_RO

BODY_HOST = 3   # OCRBODY_HOST (include/ocr_opstream.h)

pytestmark = pytest.mark.gpu


def _records(p0, dc, df, dg, dv):
    out = {}
    for i in range(len(p0)):
        a, n = int(df[i]), int(dc[i])
        out[int(p0[i])] = (n, tuple(int(x) for x in dg[a:a + n]), tuple(int(x) for x in dv[a:a + n]))
    assert len(out) == len(p0), "a task handle was handed over twice"
    return out


def _drive(eng, n, first_handle, rng_seed):
    """n HOST-body EDTs with 1..5 pre-slots, every pre-slot fed by a datablock (or NULL_GUID): ready after one step"""
    rng = np.random.default_rng(rng_seed)
    depc = rng.integers(1, 6, size=n).astype(np.uint32)
    dbs = eng.db_create(rng.integers(1, 1 << 60, size=64, dtype=np.uint64), np.arange(64, dtype=np.uint64) + 1000 + first_handle)
    g, _, _ = eng.edt_create(depc, np.full(n, BODY_HOST | BODY_F_OUTEVT, np.uint32), np.arange(n, dtype=np.uint64) + first_handle)
    dst = np.repeat(g, depc)
    slot = np.concatenate([np.arange(k, dtype=np.uint32) for k in depc])
    pick = rng.integers(0, 80, size=len(dst))
    src = np.where(pick < 64, dbs[np.minimum(pick, 63)], np.uint64(0))
    mode = np.where(rng.integers(0, 5, size=len(dst)) == 0, MODE_NULL, MODE_RO).astype(np.uint32)
    eng.add_dependence(src, dst, slot, mode)
    eng.step(1)


def test_noalloc_matches_pop_ready_on_two_calls_in_a_row():
    a = E.Engine(max_edt=8192, max_evt=4, max_db=256, max_slots=40000, max_edges=40000)
    b = E.Engine(max_edt=8192, max_evt=4, max_db=256, max_slots=40000, max_edges=40000)
    import torch
    for n, first, seed in ((5000, 0, 11), (700, 100000, 12)):       # the second call is the smaller one
        _drive(a, n, first, seed)
        _drive(b, n, first, seed)
        want = _records(*a.pop_ready())
        b.sync()
        free_before = torch.cuda.mem_get_info(0)[0]                 # hipMemGetInfo: the contract in the name
        got = _records(*b.pop_ready_noalloc())
        assert len(want) == n and got == want
        assert len(b.pop_ready_noalloc()[0]) == 0                   # nothing is handed over twice
        assert torch.cuda.mem_get_info(0)[0] == free_before, "ocrgpu_pop_ready_noalloc changed the device's free memory"
    assert sum(a.stats().err) == 0 and sum(b.stats().err) == 0
    a.close()
    b.close()


def test_noalloc_refuses_more_than_its_scratch_holds():
    eng = E.Engine(max_edt=16, max_evt=4, max_db=4)
    with pytest.raises(E.OcrGpuError) as ei:
        eng.pop_ready_noalloc(max_edts=32769, max_deps=16)
    assert ei.value.code == -22 and "scratch" in str(ei.value)
    with pytest.raises(E.OcrGpuError):
        eng.pop_ready_noalloc(max_edts=16, max_deps=131073)
    assert len(eng.pop_ready_noalloc(max_edts=32768, max_deps=131072)[0]) == 0   # the maximum itself is served
    eng.close()
