"""GUID provider rows a12/a13 through the C-ABI: reserve (generateNextGuid, guid-bitmap-based.c:70-98), put / get /
remove (hashtableConcBucketLocked{Put,Get,Remove}, hashtable.c:461-554 behind counted-map-guid.c:302,394,540) against
a Python dict as the model.  Bit layout is the reference's LOCHOME(10) | KIND(5) | COUNTER(49), guid-bitmap.h:22-71."""
import numpy as np
import pytest

from ocr_amd import engine as E

pytestmark = pytest.mark.gpu


def test_reserve_bit_layout():
    eng = E.Engine(max_edt=16, location=5)
    a = eng.guid_reserve(E.KIND_EDT, 1000)
    b = eng.guid_reserve(E.KIND_EVT_LATCH, 10)
    assert np.all(a >> np.uint64(54) == 5) and np.all((a >> np.uint64(49)) & np.uint64(31) == E.KIND_EDT)
    assert np.all((b >> np.uint64(49)) & np.uint64(31) == E.KIND_EVT_LATCH)
    ca, cb = a & np.uint64((1 << 49) - 1), b & np.uint64((1 << 49) - 1)
    assert np.all(np.diff(ca.astype(np.int64)) == 1) and int(cb[0]) == int(ca[-1]) + 1   # one counter, consecutive
    assert int(ca[0]) >= 1                                                                 # 0 is NULL_GUID's counter


def test_put_get_remove_against_dict_1m():
    n = 1 << 20
    eng = E.Engine(max_edt=16, guid_slots=4 * n)
    rng = np.random.default_rng(7)
    g = eng.guid_reserve(E.KIND_DB, n)
    vals = rng.integers(1, 1 << 62, size=n, dtype=np.uint64)
    eng.guid_put(g, vals)
    model = dict(zip(g.tolist(), vals.tolist()))
    perm = rng.permutation(n)
    assert np.array_equal(eng.guid_get(g[perm]), vals[perm])
    # guids that were never registered (other counters, other location), and the three special values
    foreign = np.concatenate([g[:1000] + np.uint64(n), g[:1000] | (np.uint64(9) << np.uint64(54)),
                              np.array([0, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFE], np.uint64)])
    assert not eng.guid_get(foreign).any()
    # remove a random half; the rest must still resolve through the tombstones
    gone = perm[: n // 2]
    eng.guid_remove(g[gone])
    for k in g[gone].tolist():
        del model[k]
    got = eng.guid_get(g)
    want = np.array([model.get(k, 0) for k in g.tolist()], np.uint64)
    assert np.array_equal(got, want)
    # re-register a quarter with new values, overwrite some live ones
    back = gone[: n // 4]
    nv = rng.integers(1, 1 << 62, size=len(back), dtype=np.uint64)
    eng.guid_put(g[back], nv)
    model.update(zip(g[back].tolist(), nv.tolist()))
    live = perm[n // 2: n // 2 + 5000]
    ov = rng.integers(1, 1 << 62, size=len(live), dtype=np.uint64)
    eng.guid_put(g[live], ov)
    model.update(zip(g[live].tolist(), ov.tolist()))
    want = np.array([model.get(k, 0) for k in g.tolist()], np.uint64)
    assert np.array_equal(eng.guid_get(g), want)
    assert sum(eng.stats().err) == 0


def test_table_full_is_reported():
    eng = E.Engine(max_edt=16, max_evt=16, max_db=16, guid_slots=1024)
    g = eng.guid_reserve(E.KIND_DB, 1100)
    v = np.arange(1, 1101, dtype=np.uint64)
    eng.guid_put(g[:1024], v[:1024])                       # load factor 1.0 still resolves (linear probing)
    assert np.array_equal(eng.guid_get(g[:1024]), v[:1024])
    assert eng.stats().err[5] == 0
    eng.guid_put(g[1024:], v[1024:])                       # no room: counted, never silently dropped
    assert eng.stats().err[5] == 76
    assert not eng.guid_get(g[1024:]).any()


def test_engine_objects_resolve_and_dead_events_do_not():
    from ocr_amd import opstream as ops
    eng = E.Engine(max_edt=16, max_evt=16, max_db=16)
    ev = eng.event_create([ops.EVT_ONCE, ops.EVT_STICKY, ops.EVT_LATCH])
    db = eng.db_create([11, 22])
    edt, oe, od = eng.edt_create([1], [ops.BODY_HASH | ops.BODY_F_OUTEVT])
    assert list(eng.guid_get(ev)) == [1, 2, 3] and list(eng.guid_get(db)) == [1, 2] and list(eng.guid_get(edt)) == [1]
    eng.event_satisfy_slot(ev[:2])                          # ONCE self-destructs (hc-event.c:450-464), STICKY persists
    assert list(eng.guid_get(ev)) == [0, 2, 3]
    eng.event_destroy(ev[1:2])
    assert list(eng.guid_get(ev)) == [0, 0, 3]


def test_removed_slots_are_reused_and_tryput():
    """hashtable.c:498-554: remove gives capacity back (a put takes over the first tombstone on its way once it knows the
    key is not further down), and TryPut inserts only when the key is absent and reports the value the table holds."""
    eng = E.Engine(max_edt=16, max_evt=16, max_db=16, guid_slots=1024)
    g = eng.guid_reserve(E.KIND_DB, 6000)
    live = {}
    rng = np.random.default_rng(7)
    for rnd in range(6):                                     # 6 x 900 puts through a 1024-slot table: only works if tombstones are reused
        blk = g[rnd * 1000: rnd * 1000 + 900]
        v = rng.integers(1, 1 << 62, size=len(blk), dtype=np.uint64)
        eng.guid_put(blk, v)
        live.update(zip(blk.tolist(), v.tolist()))
        assert np.array_equal(eng.guid_get(blk), v)
        gone = blk[:880]
        eng.guid_remove(gone)
        for k in gone.tolist():
            del live[k]
    keys = np.array(sorted(live), np.uint64)
    assert np.array_equal(eng.guid_get(keys), np.array([live[k] for k in keys.tolist()], np.uint64))
    assert not eng.guid_get(g[:880]).any()
    assert sum(eng.stats().err) == 0
    # TryPut: present -> the old value is reported and kept; absent -> inserted; duplicates inside one batch -> one winner
    some = keys[:10]
    held = eng.guid_tryput(some, np.full(10, 5, np.uint64))
    assert np.array_equal(held, np.array([live[k] for k in some.tolist()], np.uint64))
    fresh = g[5990:5995]
    batch = np.repeat(fresh, 64)
    vals = np.arange(1, len(batch) + 1, dtype=np.uint64)
    held = eng.guid_tryput(batch, vals)
    for k in range(5):
        h = held[k * 64:(k + 1) * 64]
        assert len(set(h.tolist())) == 1 and h[0] in vals[k * 64:(k + 1) * 64]      # everybody saw the one value that won
    assert np.array_equal(eng.guid_get(fresh), held[::64])
    assert sum(eng.stats().err) == 0
    eng.close()
