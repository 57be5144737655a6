"""Op streams and the checker for datablock acquire modes (iterateDbFrontier hc-task.c:484-544 +
lockable-datablock.c:139-507, the cfg's datablockfactory=Lockable).

WHO gets a contended datablock first is a scheduling accident in the reference (its waiter lists are served in the
order the workers happened to arrive), so `level` is not one fixed vector.  What IS fixed, and what every executor
(reference with 1 and 4 workers, CPU restatement, MI355X engine) is held to:
  * everything but `level` in the result record is bit-identical to the run without arbitration (bodies only read);
  * every EDT fires, no earlier than its dependences alone allow (level >= the Regular-datablock level);
  * per datablock and level, the holders that fire there are compatible: one EW, or only RW, or only CONST (RO never
    arbitrates: lockableAcquireInternal lets it fall through, :171);
  * streams with symmetric contenders have a fixed level multiset (n EW contenders ready together -> n different levels).
"""
import numpy as np

from ocr_amd.opstream import (BODY_HASH, BODY_NOP, EVT_LATCH, EVT_ONCE, EVT_STICKY, LATCH_DECR, MODE_CONST, MODE_EW, MODE_NULL,
                              MODE_RO, MODE_RW, OP_ADD_DEP, REF_DB, REF_EDT, Builder, ref_cls, ref_idx)


def ew_contenders(n=6):
    """n EDTs become ready together and all want datablock d exclusively: one per level."""
    b = Builder("lock_ew_contenders")
    start = b.evt(EVT_ONCE)
    d = b.db(7)
    es = []
    for _ in range(n):
        e = b.edt(2)
        b.add_dep(start, e, 0, MODE_NULL)
        b.add_dep(d, e, 1, MODE_EW)
        es.append(e)
    return b.finish(es, start)


def readers_then_writer():
    """CONST readers share; an EW waits for every reader (numUsers != 0, :193) and readers wait for the writer (:174)."""
    b = Builder("lock_readers_then_writer")
    start = b.evt(EVT_ONCE)
    d = b.db(11)
    rs = []
    for _ in range(4):
        e = b.edt(2)
        b.add_dep(start, e, 0, MODE_NULL)
        b.add_dep(d, e, 1, MODE_CONST)
        rs.append(e)
    w = b.edt(2)
    b.add_dep(start, w, 0, MODE_NULL)
    b.add_dep(d, w, 1, MODE_EW)
    late = []
    for _ in range(3):            # readers that become ready one level later: the writer may be ahead of them
        e = b.edt(2)
        b.add_dep(b.out(rs[0]), e, 0, MODE_NULL)
        b.add_dep(d, e, 1, MODE_CONST)
        late.append(e)
    return b.finish(rs + [w] + late, start)


def rw_share_ew_excludes():
    """RW holders of one location share the ITW lock (:216-220); EW excludes them; RO walks through everything."""
    b = Builder("lock_rw_share")
    start = b.evt(EVT_ONCE)
    d = b.db(13)
    es = []
    for m in (MODE_RW, MODE_RW, MODE_RW, MODE_EW, MODE_RO, MODE_RO, MODE_EW, MODE_CONST):
        e = b.edt(2)
        b.add_dep(start, e, 0, MODE_NULL)
        b.add_dep(d, e, 1, m)
        es.append(e)
    return b.finish(es, start)


def ordered_two_dbs():
    """Every EDT wants the same two datablocks exclusively, named in opposite slot orders: acquisition in ascending
    guid order (sortRegNode :459-477) cannot deadlock; a datablock named twice is acquired once (:495-499)."""
    b = Builder("lock_two_dbs")
    start = b.evt(EVT_ONCE)
    d0, d1 = b.db(1), b.db(2)
    es = []
    for k in range(6):
        e = b.edt(4)
        b.add_dep(start, e, 0, MODE_NULL)
        b.add_dep(d1 if k % 2 else d0, e, 1, MODE_EW)
        b.add_dep(d0 if k % 2 else d1, e, 2, MODE_EW)
        b.add_dep(d0, e, 3, MODE_EW)       # duplicate
        es.append(e)
    return b.finish(es, start)


def pipeline_with_locks(n=5):
    """A chain hands a datablock down in EW mode while side readers hold it in CONST: dependences and locks together."""
    b = Builder("lock_pipeline")
    start = b.evt(EVT_ONCE)
    d = b.db(21)
    prev, es = start, []
    for _ in range(n):
        e = b.edt(2)
        b.add_dep(prev, e, 0, MODE_NULL)
        b.add_dep(d, e, 1, MODE_EW)
        es.append(e)
        r = b.edt(2)
        b.add_dep(prev, r, 0, MODE_NULL)
        b.add_dep(d, r, 1, MODE_CONST)
        es.append(r)
        prev = b.out(e)
    return b.finish(es, start)


def random_lockable(seed, n=40):
    """Random layered program: control dependences through output events (MODE_NULL), datablocks named directly in random modes."""
    rng = np.random.default_rng(0x10C4AB1E + seed)
    b = Builder(f"lock_random_{seed}")
    start = b.evt(EVT_ONCE)
    dbs = [b.db(100 + i) for i in range(int(rng.integers(1, 5)))]
    es = []
    for _ in range(n):
        n_ctl = int(rng.integers(1, 3)) if es else 1
        n_db = int(rng.integers(0, 4))
        e = b.edt(n_ctl + n_db, BODY_HASH if rng.random() < 0.7 else BODY_NOP)
        for k in range(n_ctl):
            src = start if (not es or rng.random() < 0.25) else b.out(es[int(rng.integers(0, len(es)))])
            b.add_dep(src, e, k, MODE_NULL)
        for k in range(n_db):
            m = (MODE_EW, MODE_RW, MODE_CONST, MODE_RO, MODE_EW)[int(rng.integers(0, 5))]
            b.add_dep(dbs[int(rng.integers(0, len(dbs)))], e, n_ctl + k, m)
        es.append(e)
    return b.finish(es, start)


HAND = [ew_contenders, readers_then_writer, rw_share_ew_excludes, ordered_two_dbs, pipeline_with_locks]
N_RANDOM = 24


def all_lockable():
    return [f() for f in HAND] + [random_lockable(i) for i in range(N_RANDOM)]


def holders(s):
    """(db[], edt[], mode[]): what each EDT acquires, from the stream's direct (db -> EDT pre-slot) dependences; the mode of
    a datablock named on several pre-slots of one EDT is that of its lowest pre-slot (stable sort)."""
    m = s.main
    sel = (m["op"] == OP_ADD_DEP) & (ref_cls(m["a"]) == REF_DB) & (ref_cls(m["b"]) == REF_EDT) & (m["mode"] != MODE_NULL)
    db, edt = ref_idx(m["a"][sel]).astype(np.int64), ref_idx(m["b"][sel]).astype(np.int64)
    slot, mode = m["slot"][sel].astype(np.int64), m["mode"][sel].astype(np.int64)
    order = np.lexsort((slot, edt, db))
    db, edt, mode = db[order], edt[order], mode[order]
    first = np.ones(len(db), bool)
    first[1:] = (db[1:] != db[:-1]) | (edt[1:] != edt[:-1])
    return db[first], edt[first], mode[first]


def check_levels(s, level, base_level):
    """Raise AssertionError unless `level` is a valid outcome of `s` under Lockable datablocks (see module docstring)."""
    level = np.asarray(level).astype(np.int64)
    base_level = np.asarray(base_level).astype(np.int64)
    assert ((level >= 0) == (base_level >= 0)).all(), "the same EDTs must fire"
    assert (level >= base_level).all(), "an EDT fired before its dependences allow"
    db, edt, mode = holders(s)
    keep = (mode != MODE_RO) & (level[edt] >= 0)
    db, lv, mode = db[keep], level[edt][keep], mode[keep]
    if len(db) == 0:
        return
    order = np.lexsort((lv, db))
    db, lv, mode = db[order], lv[order], mode[order]
    new = np.ones(len(db), bool)
    new[1:] = (db[1:] != db[:-1]) | (lv[1:] != lv[:-1])
    gid = np.cumsum(new) - 1                               # one group per (datablock, level)
    size = np.bincount(gid)
    n_ew = np.bincount(gid, weights=(mode == MODE_EW))
    n_rw = np.bincount(gid, weights=(mode == MODE_RW))
    n_const = np.bincount(gid, weights=(mode == MODE_CONST))
    ok = ((n_ew == 1) & (size == 1)) | (n_rw == size) | (n_const == size)
    if not ok.all():
        g = int(np.nonzero(~ok)[0][0])
        i = int(np.nonzero(gid == g)[0][0])
        raise AssertionError(f"{s.name}: datablock {int(db[i])} level {int(lv[i])}: incompatible holders {mode[gid == g].tolist()}")
