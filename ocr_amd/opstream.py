"""Op streams: OCR programs as data (format: include/ocr_opstream.h), and the synthetic
DAG generators of SURVEY.md s8(d).

An :class:`OpStream` is the list of public OCR API calls a ``mainEdt`` would issue
(``ocrEventCreate``/``ocrDbCreate``/``ocrEdtCreate``/``ocrAddDependence``/
``ocrEventSatisfySlot``, ocr/inc/ocr-edt.h:63-347, ocr/inc/ocr-db.h:78) plus a per-EDT
script of the satisfies its body performs.  It is the one input every executor sees:
the reference runtime (oracle/ocr_replay.c), the CPU restatement (oracle/engine_ref.c)
and the MI355X engine (ocr_amd.engine through include/ocrgpu.h).
"""
from __future__ import annotations

import numpy as np

MAGIC = 0x003153504F52434F
RES_MAGIC = 0x0031534552524F43

OP_EVT_CREATE, OP_DB_CREATE, OP_EDT_CREATE, OP_ADD_DEP, OP_SATISFY, OP_EVT_DESTROY = 1, 2, 3, 4, 5, 6
REF_NULL, REF_EDT, REF_EVT, REF_DB, REF_OUTEVT, REF_OUTDB, REF_DEPV, REF_UNINIT = range(8)
BODY_NOP, BODY_HASH, BODY_FWD0, BODY_HOST, BODY_TILE = 0, 1, 2, 3, 4
BODY_F_OUTEVT, BODY_F_SINK = 0x100, 0x200
EVT_ONCE, EVT_IDEM, EVT_STICKY, EVT_LATCH, EVT_COUNTED, EVT_CHANNEL = 1, 2, 3, 4, 5, 6
LATCH_DECR, LATCH_INCR = 0, 1
MODE_NULL, MODE_CONST, MODE_RW, MODE_EW, MODE_RO = 0, 1, 2, 4, 8
EVS_DESTROYED, EVS_LIVE_UNSAT, EVS_LIVE_SAT, EVS_ABSENT = 0, 1, 2, 3

HDR_DT = np.dtype([("magic", "<u8"), ("n_edt", "<u4"), ("n_evt", "<u4"), ("n_db", "<u4"),
                   ("n_main", "<u4"), ("n_script", "<u4"), ("sink_edt", "<u4"), ("db_words", "<u4"), ("pad", "<u4", 7)])
EDT_DT = np.dtype([("depc", "<u4"), ("body", "<u4"), ("script_begin", "<u4"), ("script_len", "<u4")])
EVT_DT = np.dtype([("kind", "<u4"), ("arg", "<i4")])
OP_DT = np.dtype([("op", "u1"), ("mode", "u1"), ("pad", "<u2"), ("slot", "<u4"), ("a", "<u8"), ("b", "<u8")])
RES_HDR_DT = np.dtype([("magic", "<u8"), ("n_edt", "<u4"), ("n_evt_total", "<u4"), ("n_db_total", "<u4"),
                       ("n_fired", "<u4"), ("pad", "<u4", 10)])
assert HDR_DT.itemsize == 64 and EDT_DT.itemsize == 16 and OP_DT.itemsize == 24 and RES_HDR_DT.itemsize == 64


def ref(cls, idx=0):
    """Object reference (class in the top byte). Works on ints and numpy arrays."""
    if isinstance(idx, np.ndarray):
        return (np.uint64(cls) << np.uint64(56)) | idx.astype(np.uint64)
    return (int(cls) << 56) | int(idx)


def ref_cls(r):
    return r >> 56 if not isinstance(r, np.ndarray) else (r >> np.uint64(56)).astype(np.uint32)


def ref_idx(r):
    m = 0x00FFFFFFFFFFFFFF
    return r & m if not isinstance(r, np.ndarray) else (r & np.uint64(m))


def make_ops(op, a, b=0, slot=0, mode=0):
    """Build an OP_DT array from scalars/arrays (broadcast against ``a``)."""
    a = np.atleast_1d(np.asarray(a, dtype=np.uint64))
    out = np.zeros(a.shape[0], dtype=OP_DT)
    out["op"] = op
    out["a"] = a
    out["b"] = b
    out["slot"] = slot
    out["mode"] = mode
    return out


class OpStream:
    """SoA container for one program. ``main`` and ``script`` are OP_DT arrays."""

    def __init__(self, edt, evt, db_init, main, script, sink_edt, name="", meta=None, db_words=0):
        self.db_words = int(db_words)   # 0: datablocks are {value, id}; 64: each also carries an 8x8 int64 tile (OCRBODY_TILE)
        self.edt = np.ascontiguousarray(edt, dtype=EDT_DT)
        self.evt = np.ascontiguousarray(evt, dtype=EVT_DT)
        self.db_init = np.ascontiguousarray(db_init, dtype=np.uint64)
        self.main = np.ascontiguousarray(main, dtype=OP_DT)
        self.script = np.ascontiguousarray(script, dtype=OP_DT)
        self.sink_edt = int(sink_edt)
        self.name = name
        self.meta = dict(meta or {})

    n_edt = property(lambda s: len(s.edt))
    n_evt = property(lambda s: len(s.evt))
    n_db = property(lambda s: len(s.db_init))
    n_evt_total = property(lambda s: len(s.evt) + len(s.edt))
    n_db_total = property(lambda s: len(s.db_init) + len(s.edt))

    def with_main(self, main):
        """The same program with another main-op list (e.g. without its last op, the start satisfy)."""
        return OpStream(self.edt, self.evt, self.db_init, main, self.script, self.sink_edt, name=self.name, meta=self.meta,
                        db_words=self.db_words)

    def header(self):
        hdr = np.zeros(1, dtype=HDR_DT)
        hdr["magic"] = MAGIC
        hdr["n_edt"], hdr["n_evt"], hdr["n_db"] = self.n_edt, self.n_evt, self.n_db
        hdr["n_main"], hdr["n_script"], hdr["sink_edt"], hdr["db_words"] = len(self.main), len(self.script), self.sink_edt, self.db_words
        return hdr

    def n_satisfy_ops(self):
        """Dependence satisfactions a full resolve applies (SURVEY.md s8d metric numerator):
        every (dst,slot) satisfy applied to an EDT pre-slot or an event pre-slot."""
        return self.meta.get("n_satisfactions")

    def to_file(self, path):
        hdr = self.header()
        with open(path, "wb") as f:
            for arr in (hdr, self.edt, self.evt, self.db_init, self.main, self.script):
                f.write(arr.tobytes())

    @classmethod
    def from_file(cls, path):
        buf = np.fromfile(path, dtype=np.uint8)
        hdr = buf[:64].view(HDR_DT)[0]
        assert int(hdr["magic"]) == MAGIC, "not an op-stream file"
        off = 64
        out = []
        for dt, n in ((EDT_DT, hdr["n_edt"]), (EVT_DT, hdr["n_evt"]), (np.dtype("<u8"), hdr["n_db"]),
                      (OP_DT, hdr["n_main"]), (OP_DT, hdr["n_script"])):
            nb = int(n) * dt.itemsize
            out.append(buf[off:off + nb].view(dt).copy())
            off += nb
        return cls(out[0], out[1], out[2], out[3], out[4], int(hdr["sink_edt"]), db_words=int(hdr["db_words"]))


class Result:
    """The parity record (include/ocr_opstream.h): levels, HASH values, event states, DB values."""

    def __init__(self, level, val, evt_state, db_val):
        self.level = np.asarray(level, dtype=np.int32)
        self.val = np.asarray(val, dtype=np.uint64)
        self.evt_state = np.asarray(evt_state, dtype=np.uint64)
        self.db_val = np.asarray(db_val, dtype=np.uint64)

    @property
    def n_fired(self):
        return int((self.level >= 0).sum())

    def to_file(self, path):
        hdr = np.zeros(1, dtype=RES_HDR_DT)
        hdr["magic"] = RES_MAGIC
        hdr["n_edt"], hdr["n_evt_total"], hdr["n_db_total"] = len(self.level), len(self.evt_state), len(self.db_val)
        hdr["n_fired"] = self.n_fired
        with open(path, "wb") as f:
            for arr in (hdr, self.level, self.val, self.evt_state, self.db_val):
                f.write(np.ascontiguousarray(arr).tobytes())

    @classmethod
    def from_file(cls, path):
        buf = np.fromfile(path, dtype=np.uint8)
        hdr = buf[:64].view(RES_HDR_DT)[0]
        assert int(hdr["magic"]) == RES_MAGIC, "not a result file"
        ne, nv, nd = int(hdr["n_edt"]), int(hdr["n_evt_total"]), int(hdr["n_db_total"])
        off = 64
        level = buf[off:off + 4 * ne].view("<i4").copy(); off += 4 * ne
        val = buf[off:off + 8 * ne].view("<u8").copy(); off += 8 * ne
        evs = buf[off:off + 8 * nv].view("<u8").copy(); off += 8 * nv
        dbv = buf[off:off + 8 * nd].view("<u8").copy()
        return cls(level, val, evs, dbv)

    def level_hashes(self):
        """64-bit FNV-1a over the sorted fired-id set of each level (the per-level parity key)."""
        fired = np.nonzero(self.level >= 0)[0]
        out = {}
        if len(fired) == 0:
            return out
        lv = self.level[fired]
        order = np.lexsort((fired, lv))
        fired, lv = fired[order], lv[order]
        bounds = np.nonzero(np.diff(lv))[0] + 1
        for ids, l in zip(np.split(fired, bounds), lv[np.r_[0, bounds]]):
            out[int(l)] = (len(ids), fnv1a64(ids.astype("<u4").tobytes()))
        return out

    def digest(self):
        """One 64-bit digest of the whole record (checksum of checksums)."""
        h = 0xCBF29CE484222325
        for arr in (self.level, self.val, self.evt_state, self.db_val):
            h ^= fnv1a64(np.ascontiguousarray(arr).tobytes())
            h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h

    def diff(self, other):
        """Human-readable list of mismatches against another Result ([] == bit-exact)."""
        msgs = []
        for nm in ("level", "val", "evt_state", "db_val"):
            a, b = getattr(self, nm), getattr(other, nm)
            if a.shape != b.shape:
                msgs.append(f"{nm}: shape {a.shape} vs {b.shape}")
                continue
            bad = np.nonzero(a != b)[0]
            if len(bad):
                i = int(bad[0])
                msgs.append(f"{nm}: {len(bad)} mismatches, first at [{i}]: {int(a[i])} vs {int(b[i])}")
        return msgs


def fnv1a64(data: bytes) -> int:
    """FNV-1a, vectorised by 8-byte words for speed: digest of a byte string.

    (Not the byte-serial FNV: the word-wise fold below is what every checker here uses.)"""
    pad = (-len(data)) % 8
    w = np.frombuffer(data + b"\0" * pad, dtype="<u8")
    if len(w) == 0:
        return 0xCBF29CE484222325
    # position-dependent multiply then xor-fold: order sensitive, O(n) in numpy
    idx = np.arange(1, len(w) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (w + idx * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return int(np.bitwise_xor.reduce(z)) ^ (len(data) * 0x100000001B3 & 0xFFFFFFFFFFFFFFFF)


# ----------------------------------------------------------------------------------------------
# SplitMix64, counter-based (vectorised): element i of stream `seed` is mix(seed + (i+1)*golden)
# ----------------------------------------------------------------------------------------------
def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


SEED_BASE = 0x0C5200000000  # seed = 0x0C5200000000 + config#


def layered_dag(n_edt: int, width: int, seed: int = SEED_BASE + 2, name=None) -> OpStream:
    """BASELINE config 2: layered random DAG with ONCE output events.

    ``n_edt`` EDTs in ``n_edt/width`` layers of ``width``.  An EDT in layer l>0 has
    depc = 1 + popcount(6 random bits) (1..7, mean 4) sources drawn uniformly from layer
    l-1, each dependence an ``ocrAddDependence(outputEvent(src), edt, slot, RO)``.  Every
    EDT has a HASH body and one ONCE output event carrying its output datablock.  Layer 0
    hangs off one ONCE ``start`` event (satisfied by the last main op), because a depc==0
    EDT fires inside ocrEdtCreate (hc-task.c:1129-1134) and its output event would be
    satisfied-and-destroyed before its consumers register (hc-event.c:450-464,1078-1083).
    The output events nobody consumes feed the DECR slot of one LATCH (event->event
    dependences; initial count = their number) whose post-slot fires the sink EDT #n_edt,
    which calls ocrShutdown (SURVEY.md s8a rule 10: the sink must depend on all work).
    """
    assert n_edt % width == 0 and n_edt >= width
    n_layers = n_edt // width
    N = n_edt
    depc = np.ones(N, dtype=np.uint32)
    if n_layers > 1:
        bits = splitmix64(seed, N - width) & np.uint64(0x3F)
        pop = np.zeros(N - width, dtype=np.uint32)
        for k in range(6):
            pop += ((bits >> np.uint64(k)) & np.uint64(1)).astype(np.uint32)
        depc[width:] = 1 + pop
    slot_base = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(depc, out=slot_base[1:])
    n_edges = int(slot_base[N]) - width  # layer-0 slots hang off `start`
    # edge list for layers >= 1
    dst = np.repeat(np.arange(width, N, dtype=np.int64), depc[width:])
    slot = (np.arange(n_edges, dtype=np.int64) - (slot_base[dst] - width)).astype(np.uint32)
    layer_of_dst = dst // width
    r = splitmix64(seed ^ 0xA5A5A5A5, n_edges)
    src = (layer_of_dst - 1) * width + (r % np.uint64(width)).astype(np.int64)
    consumed = np.zeros(N, dtype=bool)
    consumed[src] = True
    leaves = np.nonzero(~consumed)[0]
    n_leaves = len(leaves)

    EV_START, EV_LATCH = 0, 1
    sink = N
    edt = np.zeros(N + 1, dtype=EDT_DT)
    edt["depc"][:N] = depc
    edt["body"][:N] = BODY_HASH | BODY_F_OUTEVT
    edt["depc"][N] = 1
    edt["body"][N] = BODY_NOP | BODY_F_SINK
    script = np.zeros(0, dtype=OP_DT)
    evt = np.zeros(2, dtype=EVT_DT)
    evt["kind"] = [EVT_ONCE, EVT_LATCH]
    evt["arg"] = [0, n_leaves]

    main = np.concatenate([
        make_ops(OP_EVT_CREATE, [EV_START, EV_LATCH]),
        make_ops(OP_EDT_CREATE, np.arange(N + 1)),
        make_ops(OP_ADD_DEP, ref(REF_OUTEVT, src), b=ref(REF_EDT, dst), slot=slot, mode=MODE_RO),
        make_ops(OP_ADD_DEP, np.full(width, ref(REF_EVT, EV_START), dtype=np.uint64),
                 b=ref(REF_EDT, np.arange(width)), slot=0, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, ref(REF_OUTEVT, leaves), b=ref(REF_EVT, EV_LATCH), slot=LATCH_DECR, mode=MODE_NULL),
        make_ops(OP_ADD_DEP, [ref(REF_EVT, EV_LATCH)], b=ref(REF_EDT, sink), slot=0, mode=MODE_NULL),
        make_ops(OP_SATISFY, [ref(REF_EVT, EV_START)], b=0, slot=0),
    ])
    # Metric numerator ("E"): satisfies applied to EDT pre-slots (sum of depc
    # over fired EDTs, sink included) + satisfies applied to LATCH pre-slots.  The 1:1
    # epilogue satisfy of each EDT's own output event is NOT counted (conservative).
    n_sat = int(slot_base[N]) + 1 + n_leaves
    meta = dict(kind="layered", n_layers=n_layers, width=width, n_edges=n_edges + width + 1,
                n_leaves=n_leaves, n_satisfactions=n_sat, seed=seed)
    return OpStream(edt, evt, np.zeros(0, dtype=np.uint64), main, script, sink,
                    name=name or f"layered_N{N}_W{width}", meta=meta)


def lockable_layered(n_edt: int, width: int, n_db: int, seed: int = SEED_BASE + 0x10C) -> OpStream:
    """config-2-shaped layered DAG whose EDTs also contend for ``n_db`` shared datablocks: every
    EDT gets one more pre-slot, fed directly by datablock (r mod n_db) in mode EW / RW / CONST / RO (3 : 2 : 2 : 1).
    Under the Regular datablock factory it resolves like layered_dag; under Lockable (OCRGPU_F_LOCKABLE_DB) the EDTs of a
    layer that share a datablock are serialised by their acquire modes."""
    s = layered_dag(n_edt, width, seed)
    N = n_edt
    edt = s.edt.copy()
    extra_slot = edt["depc"][:N].copy()
    edt["depc"][:N] += 1
    r = splitmix64(seed ^ 0x10CAB1E, N)
    which = (r % np.uint64(n_db)).astype(np.int64)
    m = ((r >> np.uint64(40)) % np.uint64(8)).astype(np.int64)
    mode = np.array([MODE_EW, MODE_EW, MODE_EW, MODE_RW, MODE_RW, MODE_CONST, MODE_CONST, MODE_RO], dtype=np.uint8)[m]
    assert int(s.main["op"][-1]) == OP_SATISFY
    main = np.concatenate([
        make_ops(OP_DB_CREATE, np.arange(n_db)),
        s.main[:-1],
        make_ops(OP_ADD_DEP, ref(REF_DB, which), b=ref(REF_EDT, np.arange(N)), slot=extra_slot, mode=mode),
        s.main[-1:],
    ])
    meta = dict(s.meta)
    meta.update(kind="lockable_layered", n_db=n_db, n_satisfactions=s.meta["n_satisfactions"] + N)
    return OpStream(edt, s.evt, 1000 + np.arange(n_db, dtype=np.uint64), main, s.script, s.sink_edt,
                    name=f"lockable_N{N}_W{width}_D{n_db}", meta=meta)


class Builder:
    """Write small op streams by hand (tests, examples): objects are plain ints, refs are made for you."""

    def __init__(self, name=""):
        self.name = name
        self.edts, self.evts, self.dbs, self.main, self.scripts = [], [], [], [], {}
        self._sunk = False

    # -- creation (each also emits the main op) --
    def evt(self, kind, arg=0):
        self.evts.append((kind, arg))
        x = len(self.evts) - 1
        self.main.append((OP_EVT_CREATE, 0, 0, x, 0))
        return ref(REF_EVT, x)

    def db(self, value):
        self.dbs.append(value)
        d = len(self.dbs) - 1
        self.main.append((OP_DB_CREATE, 0, 0, d, 0))
        return ref(REF_DB, d)

    def edt(self, depc, body=BODY_HASH, outevt=True, sink=False):
        b = body | (BODY_F_OUTEVT if outevt else 0) | (BODY_F_SINK if sink else 0)
        self.edts.append([depc, b])
        e = len(self.edts) - 1
        self.main.append((OP_EDT_CREATE, 0, 0, e, 0))
        return ref(REF_EDT, e)

    @staticmethod
    def out(edt_ref):
        return ref(REF_OUTEVT, ref_idx(edt_ref))

    @staticmethod
    def outdb(edt_ref):
        return ref(REF_OUTDB, ref_idx(edt_ref))

    @staticmethod
    def depv(k):
        return ref(REF_DEPV, k)

    # -- main-program calls --
    def add_dep(self, src, dst, slot=0, mode=MODE_RO):
        self.main.append((OP_ADD_DEP, mode, slot, src, dst))

    def satisfy(self, evt, payload=0, slot=0):
        self.main.append((OP_SATISFY, 0, slot, evt, payload))

    def destroy(self, evt):
        self.main.append((OP_EVT_DESTROY, 0, 0, evt, 0))

    # -- calls made by an EDT body before it returns --
    def script_satisfy(self, edt_ref, evt, payload=0, slot=0):
        self.scripts.setdefault(ref_idx(edt_ref), []).append((OP_SATISFY, 0, slot, evt, payload))

    def script_destroy(self, edt_ref, evt):
        self.scripts.setdefault(ref_idx(edt_ref), []).append((OP_EVT_DESTROY, 0, 0, evt, 0))

    def finish(self, expected, start=None):
        """Add the sink: a LATCH counting the EDTs in ``expected`` (through their output event
        when they have one, else through a script DECR) fires the sink EDT.  ``start`` (an
        event ref) is satisfied as the very last main op."""
        latch = self.evt(EVT_LATCH, len(expected))
        sink = self.edt(1, BODY_NOP, outevt=False, sink=True)
        for e in expected:
            if self.edts[ref_idx(e)][1] & BODY_F_OUTEVT:
                self.add_dep(self.out(e), latch, LATCH_DECR, MODE_NULL)
            else:
                self.script_satisfy(e, latch, 0, LATCH_DECR)
        self.add_dep(latch, sink, 0, MODE_NULL)
        if start is not None:
            self.satisfy(start)
        self._sunk = True
        return self.build(ref_idx(sink))

    def build(self, sink_edt=0):
        edt = np.zeros(len(self.edts), dtype=EDT_DT)
        script = []
        for i, (depc, body) in enumerate(self.edts):
            ops = self.scripts.get(i, [])
            edt[i] = (depc, body, len(script), len(ops))
            script += ops
        evt = np.array(self.evts, dtype=EVT_DT) if self.evts else np.zeros(0, dtype=EVT_DT)

        def pack(ops):
            out = np.zeros(len(ops), dtype=OP_DT)
            for i, (op, mode, slot, a, b) in enumerate(ops):
                out[i] = (op, mode, 0, slot, a, b)
            return out
        return OpStream(edt, evt, np.array(self.dbs, dtype=np.uint64), pack(self.main), pack(script), sink_edt,
                        name=self.name)


def channel_ring_slots(s):
    """waiters[] entries the engine sets aside for the rings of the stream's CHANNEL events (maxGen each; unbounded = 1024)"""
    if not len(s.evt):
        return 0
    a = s.evt["arg"][s.evt["kind"] == EVT_CHANNEL].astype(np.int64)
    return int(np.where(a > 0, a, 1024).sum())


def channel_keys(s, src_refs):
    """key per add-dependence op: its source ref when that is a CHANNEL event, else 0 -- for split_conflicts: two
    registrations on one channel are two generations and must stay in program order"""
    src_refs = np.asarray(src_refs, dtype=np.uint64)
    keys = np.zeros(len(src_refs), np.uint64)
    if len(s.evt):
        m = ref_cls(src_refs) == REF_EVT
        idx = ref_idx(src_refs[m]).astype(np.int64)
        keys[m] = np.where(s.evt["kind"][idx] == EVT_CHANNEL, src_refs[m], np.uint64(0))
    return keys


def split_conflicts(keys):
    """Cut points for a run of ops so that no non-zero key repeats inside one sub-batch (ops of one batch are
    unordered on the GPU; two satisfies of the SAME event must stay in program order)."""
    keys = np.asarray(keys)
    if len(keys) < 2 or len(np.unique(keys[keys != 0])) == int((keys != 0).sum()):
        return [(0, len(keys))]
    out, seen, start = [], set(), 0
    for i, k in enumerate(keys.tolist()):
        if k and k in seen:
            out.append((start, i)); start, seen = i, set()
        if k:
            seen.add(k)
    out.append((start, len(keys)))
    return out


def _finish_stream(N, depc, src_ref, dst_idx, slot, mode, extra_main=(), evt_extra=(), name="", meta=None,
                   bodies=None, db_init=None, start_targets=None, db_words=0):
    """Common tail of the bulk generators: EDTs 0..N-1 (HASH bodies, ONCE output events) with dependences
    (src_ref[i] -> EDT dst_idx[i], slot[i], mode[i]); EDTs in ``start_targets`` take the start event on their LAST
    pre-slot; output events nobody consumes DECR the sink latch; sink EDT = N.  Events: 0 = start (ONCE), 1 = sink
    latch, then ``evt_extra``."""
    consumed = np.zeros(N, dtype=bool)
    cls = ref_cls(np.asarray(src_ref, dtype=np.uint64))
    oe = cls == REF_OUTEVT
    consumed[ref_idx(np.asarray(src_ref, dtype=np.uint64))[oe].astype(np.int64)] = True
    for ops_arr in extra_main:
        m = (ops_arr["op"] == OP_ADD_DEP) & (ref_cls(ops_arr["a"]) == REF_OUTEVT)
        consumed[ref_idx(ops_arr["a"][m]).astype(np.int64)] = True
    leaves = np.nonzero(~consumed)[0]
    edt = np.zeros(N + 1, dtype=EDT_DT)
    edt["depc"][:N] = depc
    edt["body"][:N] = (BODY_HASH | BODY_F_OUTEVT) if bodies is None else bodies
    edt["depc"][N] = 1
    edt["body"][N] = BODY_NOP | BODY_F_SINK
    evt = np.zeros(2 + len(evt_extra), dtype=EVT_DT)
    evt["kind"][:2] = [EVT_ONCE, EVT_LATCH]
    evt["arg"][1] = len(leaves)
    for i, (k, a) in enumerate(evt_extra):
        evt[2 + i] = (k, a)
    db_init = np.zeros(0, np.uint64) if db_init is None else np.asarray(db_init, np.uint64)
    st = np.asarray(start_targets if start_targets is not None else [], dtype=np.int64)
    parts = [make_ops(OP_EVT_CREATE, np.arange(len(evt))), make_ops(OP_DB_CREATE, np.arange(len(db_init))),
             make_ops(OP_EDT_CREATE, np.arange(N + 1)),
             make_ops(OP_ADD_DEP, src_ref, b=ref(REF_EDT, np.asarray(dst_idx, dtype=np.int64)), slot=slot, mode=mode)]
    parts += list(extra_main)
    parts += [make_ops(OP_ADD_DEP, np.full(len(st), ref(REF_EVT, 0), dtype=np.uint64), b=ref(REF_EDT, st),
                       slot=(np.asarray(depc)[st] - 1).astype(np.uint32), mode=MODE_NULL),
              make_ops(OP_ADD_DEP, ref(REF_OUTEVT, leaves), b=ref(REF_EVT, 1), slot=LATCH_DECR, mode=MODE_NULL),
              make_ops(OP_ADD_DEP, [ref(REF_EVT, 1)], b=ref(REF_EDT, N), slot=0, mode=MODE_NULL),
              make_ops(OP_SATISFY, [ref(REF_EVT, 0)], b=0, slot=0)]
    main = np.concatenate([p for p in parts if len(p)])
    m = dict(meta or {})
    m.update(n_leaves=len(leaves), n_satisfactions=int(np.asarray(depc, dtype=np.int64).sum()) + 1 + len(leaves))
    return OpStream(edt, evt, db_init, main, np.zeros(0, dtype=OP_DT), N, name=name, meta=m, db_words=db_words)


def cholesky_dag(T: int, name=None, tiles=False) -> OpStream:
    """BASELINE config 3: right-looking tiled Cholesky, T tiles per side, datablock-carried
    dependences.  potrf(k) <- syrk(k-1,k); trsm(k,i) <- potrf(k), gemm(k-1,i,k); syrk(k,i) <- trsm(k,i), syrk(k-1,i);
    gemm(k,i,j) <- trsm(k,i), trsm(k,j), gemm(k-1,i,j)   (i > j > k).  At k = 0 the "previous update" is the initial
    tile, an explicit DB added with ocrAddDependence(db, edt, slot).  T + T(T-1) + T(T-1)(T-2)/6 EDTs, fan-in <= 3.

    ``tiles=False``: HASH bodies (every task folds its inputs into a new 16-byte datablock).  ``tiles=True``: the payload
    s8d.3 names -- one datablock per tile holding an 8x8 tile of 64-bit integers, updated IN PLACE by TILE bodies (the task's
    last tile -= first * second^T over wrapping integers; the task returns the tile it updated, so the same T*T datablocks
    travel down the task chains) -- checksums are exact."""
    k_, i_ = np.triu_indices(T, 1)                       # pairs k < i
    nP, nTr = T, len(k_)
    gi, gj, gk = [], [], []
    for k in range(T):
        jj, ii = np.triu_indices(T - k - 1, 1)            # j < i within (k, T)
        gi.append(ii + k + 1); gj.append(jj + k + 1); gk.append(np.full(len(ii), k))
    gi, gj, gk = (np.concatenate(x).astype(np.int64) if len(x) else np.zeros(0, np.int64) for x in (gi, gj, gk))
    nG = len(gi)
    P0, TR0, SY0, G0 = 0, nP, nP + nTr, nP + 2 * nTr
    N = nP + 2 * nTr + nG
    pair_id = -np.ones((T, T), dtype=np.int64)
    pair_id[k_, i_] = np.arange(nTr)

    def potrf(k): return P0 + k
    def trsm(k, i): return TR0 + pair_id[k, i]
    def syrk(k, i): return SY0 + pair_id[k, i]
    gid = {}
    # gemm id lookup through a dense 3-index table is too big for T=256 (16.7M) -> still fine as int32
    gtab = -np.ones((T, T, T), dtype=np.int32) if T <= 256 else None
    gtab[gk, gi, gj] = np.arange(nG, dtype=np.int32)
    def gemm(k, i, j): return G0 + gtab[k, i, j].astype(np.int64)
    tile_db = lambda i, j: i * T + j                      # explicit DB index of the initial tile (i, j)
    depc = np.zeros(N, dtype=np.uint32)
    depc[P0:P0 + nP] = 1; depc[TR0:TR0 + nTr] = 2; depc[SY0:SY0 + nTr] = 2; depc[G0:] = 3
    S, D, SL = [], [], []
    def add(src, dst, slot): S.append(np.asarray(src, np.uint64)); D.append(np.asarray(dst, np.int64)); SL.append(np.full(len(np.atleast_1d(dst)), slot, np.uint32))
    ks = np.arange(T)
    # potrf(k): slot 0 <- syrk(k-1,k) | A(k,k)
    add(ref(REF_DB, tile_db(ks[:1], ks[:1])), potrf(ks[:1]), 0)
    if T > 1:
        add(ref(REF_OUTEVT, syrk(ks[1:] - 1, ks[1:])), potrf(ks[1:]), 0)
    # trsm(k,i): slot 0 <- potrf(k); slot 1 <- gemm(k-1,i,k) | A(i,k)
    add(ref(REF_OUTEVT, potrf(k_)), trsm(k_, i_), 0)
    m0 = k_ == 0
    add(ref(REF_DB, tile_db(i_[m0], k_[m0])), trsm(k_[m0], i_[m0]), 1)
    add(ref(REF_OUTEVT, gemm(k_[~m0] - 1, i_[~m0], k_[~m0])), trsm(k_[~m0], i_[~m0]), 1)
    # syrk(k,i): slot 0 <- trsm(k,i); slot 1 <- syrk(k-1,i) | A(i,i)
    add(ref(REF_OUTEVT, trsm(k_, i_)), syrk(k_, i_), 0)
    add(ref(REF_DB, tile_db(i_[m0], i_[m0])), syrk(k_[m0], i_[m0]), 1)
    add(ref(REF_OUTEVT, syrk(k_[~m0] - 1, i_[~m0])), syrk(k_[~m0], i_[~m0]), 1)
    # gemm(k,i,j): slot 0 <- trsm(k,i); slot 1 <- trsm(k,j); slot 2 <- gemm(k-1,i,j) | A(i,j)
    if nG:
        g_all = G0 + np.arange(nG)
        add(ref(REF_OUTEVT, trsm(gk, gi)), g_all, 0)
        add(ref(REF_OUTEVT, trsm(gk, gj)), g_all, 1)
        g0 = gk == 0
        add(ref(REF_DB, tile_db(gi[g0], gj[g0])), g_all[g0], 2)
        add(ref(REF_OUTEVT, gemm(gk[~g0] - 1, gi[~g0], gj[~g0])), g_all[~g0], 2)
    src, dst, slot = np.concatenate(S), np.concatenate(D), np.concatenate(SL)
    # DB sources are immediate satisfies: the start event gates every EDT that has ONLY DB inputs? none here need it:
    # potrf(0) has only a DB input and would fire during construction, so give the k=0 layer an extra start slot.
    first = np.r_[potrf(np.array([0])), trsm(k_[m0], i_[m0]), syrk(k_[m0], i_[m0]), (G0 + np.nonzero(gk == 0)[0]) if nG else np.zeros(0, np.int64)]
    depc[first] += 1
    db_init = splitmix64(SEED_BASE + 3, T * T)
    mode = np.where(ref_cls(src) == REF_DB, MODE_RO, MODE_RW).astype(np.uint8)
    if tiles:
        return _finish_stream(N, depc, src, dst, slot, mode, name=name or f"cholesky_tiles_T{T}", db_init=db_init, start_targets=first,
                              meta=dict(kind="cholesky_tiles", T=T), bodies=BODY_TILE | BODY_F_OUTEVT, db_words=64)
    return _finish_stream(N, depc, src, dst, slot, mode, name=name or f"cholesky_T{T}", db_init=db_init,
                          start_targets=first, meta=dict(kind="cholesky", T=T))


def stencil_dag(G: int, steps: int, name=None) -> OpStream:
    """BASELINE config 4: 2D 5-point stencil on G x G tiles for `steps` steps.  EDT(t,i,j) depends
    on (t-1,i,j) [slot 0] and its N,S,W,E neighbours at t-1 [slots 1..]; every dependence is the ONCE output event of
    the producer carrying its datablock (the "halo satisfy")."""
    n_tile = G * G
    N = n_tile * steps
    ii, jj = np.divmod(np.arange(n_tile, dtype=np.int64), G)
    nb = [(ii, jj)]
    for di, dj in ((-1, 0), (1, 0), (0, -1), (0, 1)):
        ni, nj = ii + di, jj + dj
        ok = (ni >= 0) & (ni < G) & (nj >= 0) & (nj < G)
        nb.append((np.where(ok, ni, -1), nj))
    valid = np.stack([n[0] >= 0 for n in nb], axis=1)             # [tile, 5]
    nbr = np.stack([np.where(n[0] >= 0, n[0] * G + n[1], 0) for n in nb], axis=1)
    slot_of = np.cumsum(valid, axis=1) - 1                         # compact slots per tile
    fanin = valid.sum(axis=1).astype(np.uint32)
    depc = np.empty(N, dtype=np.uint32)
    depc[:n_tile] = 1                                              # step 0: start event only
    depc[n_tile:] = np.tile(fanin, steps - 1)
    t_idx, d_idx = np.nonzero(valid)
    S, D, SL = [], [], []
    for t in range(1, steps):
        S.append((t - 1) * n_tile + nbr[t_idx, d_idx]); D.append(t * n_tile + t_idx); SL.append(slot_of[t_idx, d_idx])
    src = ref(REF_OUTEVT, np.concatenate(S)) if S else np.zeros(0, np.uint64)
    dst = np.concatenate(D) if D else np.zeros(0, np.int64)
    slot = (np.concatenate(SL) if SL else np.zeros(0)).astype(np.uint32)
    return _finish_stream(N, depc, src, dst, slot, MODE_RO, name=name or f"stencil_{G}x{G}x{steps}",
                          start_targets=np.arange(n_tile), meta=dict(kind="stencil", G=G, steps=steps))


def powerlaw_dag(n_edt: int, seed: int = SEED_BASE + 5, alpha: float = 2.1, cap: int = 1_000_000, group: int = 1024,
                 name=None, progress=None, stable_argsort=None) -> OpStream:
    """BASELINE config 5: power-law fan-out DAG with latch events.  EDT ids are a topological
    order; EDT e has out-degree ~ Zipf(alpha) (capped) to targets drawn uniformly from (e, n); every `group`
    consecutive EDTs check out of one LATCH (initial count = group size: the INCR-on-create half of a finish scope,
    hc-task.c:1071-1078, done at construction), and the group latches check out of one root latch that fires the
    sink.  EDTs without a data predecessor hang off the start event.

    ``progress(msg)`` is called after every stage (a 125M-EDT stream takes minutes: a caller under a silence limit prints them);
    ``stable_argsort(keys)`` may replace numpy's stable sort of the 1e9 edge keys (tools/bench_configs.py sorts on the GPU)."""
    say = progress or (lambda msg: None)
    N = n_edt
    u = (splitmix64(seed, N).astype(np.float64) + 0.5) / 2.0 ** 64
    deg = np.minimum(np.floor(u ** (-1.0 / (alpha - 1.0))).astype(np.int64), cap)   # P(D >= d) = d^-(alpha-1)
    deg = np.minimum(deg, N - 1 - np.arange(N))
    deg[deg < 0] = 0
    E = int(deg.sum())
    say(f"powerlaw: {N} out-degrees drawn, {E} edges")
    src = np.repeat(np.arange(N, dtype=np.int64), deg)
    r = splitmix64(seed ^ 0x5EED, E)
    span = (N - 1 - src).astype(np.uint64)
    dst = src + 1 + (r % np.maximum(span, np.uint64(1))).astype(np.int64)
    del r, span
    say("powerlaw: targets drawn")
    # edges in (dst, src) order: src is non-decreasing as generated, so a STABLE sort by dst alone is np.lexsort((src, dst))
    order = stable_argsort(dst) if stable_argsort is not None else np.argsort(dst, kind="stable")
    src, dst = src[order], dst[order]
    del order
    say("powerlaw: edges sorted by (target, source)")
    indeg = np.bincount(dst, minlength=N).astype(np.int64)
    first_of = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(indeg, out=first_of[1:])
    slot = (np.arange(E, dtype=np.int64) - first_of[dst]).astype(np.uint32)
    roots = np.nonzero(indeg == 0)[0]
    depc = indeg.astype(np.uint32)
    depc[roots] = 1
    n_groups = (N + group - 1) // group
    gsize = np.bincount(np.arange(N) // group, minlength=n_groups)
    evt_extra = [(EVT_LATCH, n_groups)] + [(EVT_LATCH, int(g)) for g in gsize]   # event 2 = root latch, 3.. = groups
    # every EDT's output event -> its group latch (DECR); group latch -> root latch (DECR); root -> sink latch handled
    # by making the root latch's post-slot the only "leaf": we route root latch -> event 1 (the generic sink latch)
    extra = [make_ops(OP_ADD_DEP, ref(REF_OUTEVT, np.arange(N)), b=ref(REF_EVT, 3 + np.arange(N) // group), slot=LATCH_DECR, mode=MODE_NULL),
             make_ops(OP_ADD_DEP, ref(REF_EVT, 3 + np.arange(n_groups)), b=np.full(n_groups, ref(REF_EVT, 2), np.uint64), slot=LATCH_DECR, mode=MODE_NULL),
             make_ops(OP_ADD_DEP, [ref(REF_EVT, 2)], b=ref(REF_EVT, 1), slot=LATCH_DECR, mode=MODE_NULL)]
    say("powerlaw: pre-slots assigned")
    s = _finish_stream(N, depc, ref(REF_OUTEVT, src), dst, slot, MODE_RO, extra_main=extra, evt_extra=evt_extra,
                       name=name or f"powerlaw_N{N}", start_targets=roots,
                       meta=dict(kind="powerlaw", n_edges=E, n_groups=n_groups, max_outdeg=int(deg.max())))
    # all output events are consumed by group latches -> no generic leaves; the sink latch counts the root latch only
    s.evt["arg"][1] = 1
    s.meta["n_satisfactions"] = int(depc.astype(np.int64).sum()) + 1 + N + n_groups + 1
    return s
