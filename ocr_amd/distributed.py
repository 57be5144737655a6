"""GUID-space partitioning across GPUs: one engine per rank, cross-partition satisfies batched per level.

This is the host-side plumbing that takes the place of the reference's distributed policy domain + MPI
comm-platform (ocr/src/policy-domain/hc-dist/hc-dist-policy.c:1027-1031,2208; ocr/src/comm-platform/mpi/
mpi-comm-platform.c:1055) for PD_MSG_DEP_SATISFY: instead of one marshalled ocrPolicyMsg_t per remote satisfy,
every engine fills per-destination buffers of 32-byte records during a level, and the ranks exchange them with
ONE all-to-all per round (RCCL over xGMI through torch.distributed; device-to-device copies when several
partitions share one process, which is how the single-GPU tests exercise the same code).

Ownership: an object lives on the rank named by its GUID's LOCHOME bits (ocr/src/guid/guid-bitmap-based.c:25-58).
``owner_of`` decides it per logical index; an EDT's output event and output DB live with the EDT.
"""
from __future__ import annotations

import os

# torch ships its own copy of the HIP runtime: load it before libocrgpu.so pulls in /opt/rocm's, so that one runtime
# serves both in this process (the other order leaves torch with "No HIP GPUs are available")
import torch  # noqa: F401

import numpy as np

from . import opstream as ops
from .engine import F_LOCKABLE_DB, F_WIDE_DB, SCRIPT_OP_DT, Engine
from .opstream import OpStream, Result

REC_BYTES = 32
TILE_REC_SLOTS = 23   # slots of a record that carries a tile datablock (ocrgpu.hip: head + 22 slots of 3 words = 64 tile words + home guid)
KIND_DB, KIND_EDT = 2, 3
EVT_GUID_KIND = {ops.EVT_ONCE: 17, ops.EVT_COUNTED: 18, ops.EVT_IDEM: 19, ops.EVT_STICKY: 21, ops.EVT_LATCH: 22,
                 ops.EVT_CHANNEL: 23}


def mkguid(loc, kind, counter):
    return (np.uint64(loc) << np.uint64(54)) | (np.asarray(kind, np.uint64) << np.uint64(49)) | np.asarray(counter, np.uint64)


def default_owner(idx, world):
    """Hash partition of a logical index (Fibonacci hash of the index, top bits)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return ((idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(40)) % np.uint64(world)


def stencil_block_owner(G, steps, world, grid=None):
    """Block partition of BASELINE config 4 (the reference places an object by the LOCHOME bits its creator
    chose, ocr/src/guid/guid-bitmap-based.c:25-58; a stencil program chooses them by tile position): tile (i, j) of every
    time step lives on block (i // bh, j // bw) of a py x px grid of ranks (2 x 4 for 8 GPUs), so only the halo edges along block
    borders cross partitions.  Returns owner_edt for :class:`Partition` (the sink EDT goes to rank 0)."""
    if grid is None:
        py = 1
        while (2 * py) ** 2 <= world and world % (2 * py) == 0:   # the squarest py x px with py <= px: 2 x 4 for 8, 2 x 2 for 4
            py *= 2
        grid = (py, world // py)
    py, px = grid
    assert py * px == world and G % py == 0 and G % px == 0, (grid, G)
    ii, jj = np.divmod(np.arange(G * G, dtype=np.int64), G)
    tile_owner = (ii // (G // py)) * px + jj // (G // px)
    return np.r_[np.tile(tile_owner, steps), 0].astype(np.int64)


class Partition:
    """One rank's share of an op stream.  Every rank walks the same main-op list and executes the ops whose
    executor it is: creations by the owner, ocrAddDependence by the owner of the source (the waiter list lives
    with the source event), NULL-source add-deps by the owner of the destination, satisfies by the owner of the
    payload DB (else of the event).  GUIDs of remote objects are predicted: every engine hands out counters in
    creation order, so all ranks can compute all GUIDs from the stream alone (checked against the real ones)."""

    def __init__(self, s: OpStream, rank: int, world: int, device=0, flags=0, send_cap=None, torch=None, owner_edt=None, proxy_cap=None):
        """owner_edt: rank of every EDT (array of s.n_edt) in place of the hash partition -- the placement hook: an EDT's
        output event and output datablock live with it, so this decides which dependences cross partitions."""
        import torch as _torch
        self.torch = torch or _torch
        self.s, self.rank, self.world = s, rank, world
        self.own_edt = (default_owner(np.arange(s.n_edt), world) if owner_edt is None else np.asarray(owner_edt)).astype(np.int64)
        assert len(self.own_edt) == s.n_edt and self.own_edt.min() >= 0 and self.own_edt.max() < world
        self.own_evt = default_owner(np.arange(s.n_evt) + (1 << 40), world).astype(np.int64)
        self.own_db = default_owner(np.arange(s.n_db) + (2 << 40), world).astype(np.int64)
        self._predict_guids()
        mine_e = self.own_edt == rank
        n_my_edt = int(mine_e.sum())
        n_adddep = int((s.main["op"] == ops.OP_ADD_DEP).sum())
        # streams with tile datablocks (OCRGPU_F_WIDE_DB): a record that carries a datablock takes 23 slots of its region (head + the
        # tile), and a TILE body that updated a copy sends one more home
        tiles = s.db_words == 64
        per_rec = TILE_REC_SLOTS if tiles else 2 if (flags & F_LOCKABLE_DB) else 1   # (Lockable: head + the datablock's home guid)
        if tiles:
            flags |= F_WIDE_DB
        cap = send_cap or max(4096, per_rec * (n_adddep + len(s.script) + (s.n_edt if tiles else 0)) + 1024)
        self.send_cap = cap
        self.send = self.torch.zeros((world, cap, REC_BYTES), dtype=self.torch.uint8, device=f"cuda:{device}")
        self.eng = Engine(max_edt=max(n_my_edt, 1), max_evt=max(int((self.own_evt == rank).sum()), 1),
                          max_db=s.n_db + (proxy_cap if proxy_cap is not None else world * (cap // per_rec + 1)) + 16,
                          max_slots=int(s.edt["depc"][mine_e].sum()) + 1, max_edges=n_adddep + 1 + ops.channel_ring_slots(s), device=device,
                          location=rank, flags=flags)
        self.eng.exchange_config(world, self.send.data_ptr(), cap, n_adddep + len(s.script) + 16)
        self.eng.set_hash_tag_base(s.n_db)
        self.my_edts = np.nonzero(mine_e)[0]          # in creation order == engine index order (checked in build)
        self._created = np.zeros(s.n_edt, dtype=bool)
        self.my_evts, self.my_dbs = [], []

    # -- GUID prediction: replay every rank's creation counters --
    def _predict_guids(self):
        s, W = self.s, self.world
        self.edt_guid = np.zeros(s.n_edt, np.uint64)
        self.evt_guid = np.zeros(s.n_evt_total, np.uint64)
        self.db_guid = np.zeros(s.n_db_total, np.uint64)
        counter = np.ones(W, np.int64)
        main = s.main
        if not len(main):
            return
        cuts = np.nonzero(np.diff(main["op"].astype(np.int16)))[0] + 1
        for b0, b1 in zip(np.r_[0, cuts], np.r_[cuts, len(main)]):
            op = int(main["op"][b0])
            if op not in (ops.OP_EVT_CREATE, ops.OP_DB_CREATE, ops.OP_EDT_CREATE):
                continue
            idx = main["a"][b0:b1].astype(np.int64)
            own = {ops.OP_EVT_CREATE: self.own_evt, ops.OP_DB_CREATE: self.own_db, ops.OP_EDT_CREATE: self.own_edt}[op][idx]
            step = 3 if op == ops.OP_EDT_CREATE else 1
            for r in range(W):
                m = own == r
                k = int(m.sum())
                if not k:
                    continue
                c = counter[r] + step * np.arange(k, dtype=np.int64)
                ii = idx[m]
                if op == ops.OP_EVT_CREATE:
                    kinds = np.array([EVT_GUID_KIND[int(x)] for x in s.evt["kind"][ii]], np.uint64)
                    self.evt_guid[ii] = mkguid(r, kinds, c)
                elif op == ops.OP_DB_CREATE:
                    self.db_guid[ii] = mkguid(r, KIND_DB, c)
                else:
                    self.edt_guid[ii] = mkguid(r, KIND_EDT, c)
                    has_oe = (s.edt["body"][ii] & ops.BODY_F_OUTEVT) != 0
                    self.evt_guid[s.n_evt + ii] = np.where(has_oe, mkguid(r, 17, c + 1), np.uint64(0))
                    self.db_guid[s.n_db + ii] = mkguid(r, KIND_DB, c + 2)
                counter[r] += step * k

    def _guids(self, refs):
        refs = np.asarray(refs, dtype=np.uint64)
        cls, idx = ops.ref_cls(refs), ops.ref_idx(refs).astype(np.int64)
        out = np.zeros(len(refs), np.uint64)
        s = self.s
        for c, table, base in ((ops.REF_EDT, self.edt_guid, 0), (ops.REF_EVT, self.evt_guid, 0),
                               (ops.REF_OUTEVT, self.evt_guid, s.n_evt), (ops.REF_DB, self.db_guid, 0),
                               (ops.REF_OUTDB, self.db_guid, s.n_db)):
            m = cls == c
            if m.any():
                out[m] = table[idx[m] + base]
        return out

    def _owner_of_refs(self, refs):
        """Owner rank of each ref (-1 for NULL)."""
        refs = np.asarray(refs, dtype=np.uint64)
        cls, idx = ops.ref_cls(refs), ops.ref_idx(refs).astype(np.int64)
        out = np.full(len(refs), -1, np.int64)
        for c, own in ((ops.REF_EDT, self.own_edt), (ops.REF_OUTEVT, self.own_edt), (ops.REF_OUTDB, self.own_edt),
                       (ops.REF_EVT, self.own_evt), (ops.REF_DB, self.own_db)):
            m = cls == c
            if m.any():
                out[m] = own[idx[m]]
        return out

    def build(self, sync=None):
        """Replay the main ops.  ``sync()`` is called after every batch that may have produced cross-partition
        records (all ranks reach the same points: the batch boundaries depend on the stream only), so the main
        program's satisfies are delivered in program order."""
        for _ in self.build_steps():
            if sync is not None:
                sync()
        return self

    def build_steps(self):
        s, eng, me = self.s, self.eng, self.rank
        main = s.main
        scripts_done = len(s.script) == 0
        if len(main):
            cuts = np.nonzero(np.diff(main["op"].astype(np.int16)))[0] + 1
            bounds = list(zip(np.r_[0, cuts], np.r_[cuts, len(main)]))
        else:
            bounds = []
        for b0, b1 in bounds:
            blk = main[b0:b1]
            op = int(blk["op"][0])
            if op == ops.OP_EVT_CREATE:
                idx = blk["a"].astype(np.int64)
                idx = idx[self.own_evt[idx] == me]
                if len(idx):
                    g = eng.event_create(s.evt["kind"][idx], s.evt["arg"][idx])
                    assert np.array_equal(g, self.evt_guid[idx]), "GUID prediction drifted (events)"
                    self.my_evts += idx.tolist()
            elif op == ops.OP_DB_CREATE:
                idx = blk["a"].astype(np.int64)
                idx = idx[self.own_db[idx] == me]
                if len(idx):
                    g = eng.db_create(s.db_init[idx], idx.astype(np.uint64))
                    assert np.array_equal(g, self.db_guid[idx]), "GUID prediction drifted (DBs)"
                    self.my_dbs += idx.tolist()
            elif op == ops.OP_EDT_CREATE:
                idx = blk["a"].astype(np.int64)
                idx = idx[self.own_edt[idx] == me]
                if len(idx):
                    g, oe, od = eng.edt_create(s.edt["depc"][idx], s.edt["body"][idx], idx.astype(np.uint64))
                    assert np.array_equal(g, self.edt_guid[idx]) and np.array_equal(oe, self.evt_guid[s.n_evt + idx]) \
                        and np.array_equal(od, self.db_guid[s.n_db + idx]), "GUID prediction drifted (EDTs)"
                    self._created[idx] = True
            elif op == ops.OP_ADD_DEP:
                for c0, c1 in (ops.split_conflicts(ops.channel_keys(s, blk["a"])) if ops.channel_ring_slots(s) else [(0, len(blk))]):   # registrations on one CHANNEL keep program order
                    sub = blk[c0:c1]
                    so, do = self._owner_of_refs(sub["a"]), self._owner_of_refs(sub["b"])
                    m = np.where(so >= 0, so, do) == me
                    if m.any():
                        b = sub[m]
                        eng.add_dependence(self._guids(b["a"]), self._guids(b["b"]), b["slot"], b["mode"].astype(np.uint32))
                    yield op
            elif op == ops.OP_SATISFY:
                for c0, c1 in ops.split_conflicts(blk["a"]):   # two satisfies of one event keep program order
                    sub = blk[c0:c1]
                    po, eo = self._owner_of_refs(sub["b"]), self._owner_of_refs(sub["a"])
                    m = np.where(po >= 0, po, eo) == me
                    if m.any():
                        b = sub[m]
                        eng.event_satisfy_slot(self._guids(b["a"]), self._guids(b["b"]), b["slot"])
                    yield op
            elif op == ops.OP_EVT_DESTROY:
                m = self._owner_of_refs(blk["a"]) == me
                if m.any():
                    eng.event_destroy(self._guids(blk["a"][m]))
                yield op
        if not scripts_done:
            self._set_scripts()

    def _scripted_edts_exist(self):
        s = self.s
        has = np.nonzero((s.edt["script_len"] > 0) & (self.own_edt == self.rank))[0]
        return bool(np.all(self._created[has])) if len(has) else True

    def _set_scripts(self):
        s = self.s
        has = np.nonzero((s.edt["script_len"] > 0) & (self.own_edt == self.rank))[0]
        if not len(has):
            return
        # re-pack the scripts of the EDTs this rank owns
        begins, lens, rows = [], [], []
        for e in has:
            b, n = int(s.edt["script_begin"][e]), int(s.edt["script_len"][e])
            begins.append(len(rows)); lens.append(n); rows += list(range(b, b + n))
        sub = s.script[np.asarray(rows, dtype=np.int64)]
        sc = np.zeros(len(sub), dtype=SCRIPT_OP_DT)
        sc["target"], sc["slot"], sc["op"] = self._guids(sub["a"]), sub["slot"], sub["op"]
        cls = ops.ref_cls(sub["b"])
        pk, pay = np.zeros(len(sc), np.uint8), np.zeros(len(sc), np.uint64)
        m = cls == ops.REF_DB
        if m.any():
            # a body may satisfy with a DB that lives on another partition: satisfies carry DBs by value, so this
            # rank keeps a copy under the DB's own guid (main's DBs never change after creation)
            away = np.unique(ops.ref_idx(sub["b"][m]).astype(np.int64))
            away = away[self.own_db[away] != self.rank]
            if len(away):
                self.eng.db_create(s.db_init[away], away.astype(np.uint64), guids=self.db_guid[away])
            pk[m], pay[m] = 1, self._guids(sub["b"][m])
        pk[cls == ops.REF_OUTDB] = 2
        m = cls == ops.REF_DEPV
        pk[m], pay[m] = 3, ops.ref_idx(sub["b"][m])
        sc["payload_kind"], sc["payload"] = pk, pay
        self.eng.edt_set_scripts(self.edt_guid[has], np.asarray(begins, np.uint32), np.asarray(lens, np.uint32), sc)

    def fill_result(self, level, val, evs, dbv):
        """Write this rank's objects into the global result arrays."""
        s, eng = self.s, self.eng
        n = len(self.my_edts)
        if n:
            lv, v = eng.read_edts(0, n)
            level[self.my_edts], val[self.my_edts] = lv, v
            evs[s.n_evt + self.my_edts] = eng.read_events(1, 0, n)
            dbv[s.n_db + self.my_edts] = eng.read_dbs(1, 0, n)
        if self.my_evts:
            evs[np.asarray(self.my_evts)] = eng.read_events(0, 0, len(self.my_evts))
        if self.my_dbs:
            mine = np.asarray(self.my_dbs)
            idx = eng.guid_get(self.db_guid[mine]).astype(np.int64) - 1     # engine index of each explicit DB this rank made
            assert (idx >= 0).all()
            dbv[mine] = eng.read_dbs(0, 0, int(idx.max()) + 1)[idx]


# ---- the per-level protocol ------------------------------------------------------------------------------------
# One engine call per round (`Engine.step`: close pending satisfies, optionally fire ONE level, close again; its
# single synchronisation also brings back the per-destination send counters and the "EDTs are ready" flag), then
#   1. one small all-gather of [counts to every rank..., ready flag]  -> every rank knows the whole W x W matrix,
#   2. if any record is in flight: one all-to-all-v of the 32-byte records, apply, and another round without firing;
#   3. else if anybody is ready: next level; else the resolve is over.
# Levels stay exact: nobody fires level L+1 before every satisfy of level L (event chains across partitions
# included) has been delivered and closed.

def _lockstep_round(parts, fire):
    W = len(parts)
    M = np.zeros((W, W + 1), dtype=np.int64)
    for r, p in enumerate(parts):
        ready, _, counts = p.eng.step(1 if fire else 0, force_level=bool(fire), want_counts=True)
        M[r, :W], M[r, W] = counts, int(ready)
    return M


def _lockstep_settle(parts, M=None):
    """Exchange until no record is in flight (device-to-device copies stand in for the all-to-all)."""
    torch = parts[0].torch
    W = len(parts)
    if M is None:
        M = _lockstep_round(parts, False)
    while M[:, :W].sum() > 0:
        inbox = []
        for dst in range(W):
            chunks = [parts[src].send[dst, :int(M[src, dst])] for src in range(W) if M[src, dst]]
            inbox.append(torch.cat(chunks).contiguous() if chunks else None)
        for p in parts:
            p.eng.exchange_reset()
        for dst, p in enumerate(parts):
            if inbox[dst] is not None:
                p.eng.exchange_apply(inbox[dst].data_ptr(), inbox[dst].shape[0])
        M = _lockstep_round(parts, False)
    return M


def _lockstep_exchange(parts):
    _lockstep_settle(parts)


def build_lockstep(parts):
    """Build every partition in lockstep, exchanging after each dependence / satisfy batch."""
    gens = [p.build_steps() for p in parts]
    while True:
        alive = [next(g, None) is not None for g in gens]
        assert all(alive) or not any(alive), "partitions walked different batch sequences"
        if not any(alive):
            return parts
        _lockstep_settle(parts)


def resolve_lockstep(parts, count_records=None):
    """Drive several partitions that live in ONE process (tests: all on one GPU).  Same protocol as
    :func:`resolve_distributed`.  ``count_records``: a list that receives the number of records each firing round put on the
    wire (the level's own fan-out; records that cascade in the settle rounds behind it are not counted)."""
    W = len(parts)
    levels = 0
    M = _lockstep_settle(parts)
    while M[:, W].sum() > 0:
        M = _lockstep_round(parts, True)
        if count_records is not None:
            count_records.append(int(M[:, :W].sum()))
        M = _lockstep_settle(parts, M)
        levels += 1
    return levels


def resolve_lockstep_c(parts, no_cascade=False):
    """The same, with the per-level loop and the exchange in C (ocrgpu_resolve_exchange_local: the protocol RCCL ranks run,
    device-to-device copies standing in for the all-to-all)."""
    from .engine import resolve_exchange_local
    return resolve_exchange_local([p.eng for p in parts], no_cascade=no_cascade)


def connect_peer_windows(parts):
    """The partitions of ONE process joined by peer windows (ocrgpu_peer_export / ocrgpu_peer_connect): from here on
    resolve_lockstep_c and the engines' exchange run the protocol that one process per GPU runs across processes."""
    handles = [p.eng.peer_export() for p in parts]
    for p in parts:
        p.eng.peer_connect(handles)
    # From here on the partitions only meet through their windows (arrival words, as between devices), so each gets a stream of its
    # own: on the NULL stream they all share, every kernel of every partition stands in one line (2 partitions on one GPU: 103 us
    # per level that way, 58 with a stream each; 8 partitions: 342 -> 161).  OCRGPU_RIG_SHARED_STREAM=1 keeps the old arrangement.
    if len(parts) > 1 and not os.environ.get("OCRGPU_RIG_SHARED_STREAM"):
        for p in parts:
            p.eng.sync()
        for p in parts:
            p.eng.stream_create()


def connect_peer_windows_dist(part, dist, group=None):
    """One partition per process: the window handles travel over the process group (plain bytes, any backend will do)."""
    try:
        mine, why = part.eng.peer_export(), None
    except Exception as ex:  # noqa: BLE001 -- every rank still takes part in the collective below, or the others would wait for it
        mine, why = None, ex
    handles = [None] * dist.get_world_size(group)
    dist.all_gather_object(handles, mine, group=group)
    if any(h is None for h in handles):
        raise RuntimeError(f"peer windows: rank(s) {[r for r, h in enumerate(handles) if h is None]} could not export a window"
                           + (f" (here: {why})" if why else ""))
    part.eng.peer_connect(handles)


HOST_GROUP = None   # set to a gloo group to route every collective of the exchange through host memory
DEV_GROUP = None    # set to an RCCL ("nccl") group to keep the exchange on the GPUs when the default group is a gloo one


def _grp():
    return DEV_GROUP if DEV_GROUP is not None else HOST_GROUP


def _host(dist):
    if DEV_GROUP is not None:
        return False
    return HOST_GROUP is not None or dist.get_backend() == "gloo"


def exchange_records(dist, send, counts, device):
    """all-to-all-v of 32-byte records when the receive counts are NOT known yet: ``send[dst, :counts[dst]]`` goes
    to rank dst.  Returns (recv tensor, n)."""
    import torch
    W = dist.get_world_size()
    cdev = "cpu" if _host(dist) else device
    cnt_out = torch.as_tensor(np.asarray(counts, dtype=np.int64)).to(cdev)
    cnt_in = torch.empty_like(cnt_out)
    dist.all_to_all_single(cnt_in, cnt_out, group=_grp())
    cin = cnt_in.cpu().numpy()
    return exchange_payload(dist, send, counts, cin, device), int(cin.sum())


def exchange_payload(dist, send, counts_out, counts_in, device):
    """The record all-to-all-v with known splits.  With the "nccl" (= RCCL) backend everything stays on the GPU; with
    "gloo" (CPU tests, or several ranks sharing one GPU) the records are staged through host memory."""
    import torch
    W = dist.get_world_size()
    cdev = "cpu" if _host(dist) else device
    pieces = [send[d, :int(counts_out[d])] for d in range(W) if int(counts_out[d])]
    sendbuf = (torch.cat(pieces) if pieces else send[0, :0]).contiguous().to(cdev)
    recv = torch.empty((int(np.sum(counts_in)), REC_BYTES), dtype=torch.uint8, device=cdev)
    dist.all_to_all_single(recv, sendbuf, output_split_sizes=[int(x) for x in counts_in],
                           input_split_sizes=[int(x) for x in counts_out], group=_grp())
    return recv.to(device)


def _dist_round(part, dist, fire):
    import torch
    W, rank = dist.get_world_size(), dist.get_rank()
    cdev = "cpu" if _host(dist) else part.send.device
    ready, _, counts = part.eng.step(1 if fire else 0, force_level=bool(fire), want_counts=True)
    vec = torch.tensor([int(c) for c in counts] + [int(ready)], dtype=torch.int64, device=cdev)
    rows = [torch.empty_like(vec) for _ in range(W)]
    dist.all_gather(rows, vec, group=_grp())
    return torch.stack(rows).cpu().numpy()


def distributed_exchange(part, dist, M=None):
    """Rounds of [status all-gather, record all-to-all-v, apply, close] until no record is in flight on any rank."""
    W, rank = dist.get_world_size(), dist.get_rank()
    if M is None:
        M = _dist_round(part, dist, False)
    while M[:, :W].sum() > 0:
        recv = exchange_payload(dist, part.send, M[rank, :W], M[:, rank], part.send.device)
        part.eng.exchange_reset()
        if recv.shape[0]:
            part.eng.exchange_apply(recv.data_ptr(), recv.shape[0])
        M = _dist_round(part, dist, False)
    return M


def resolve_distributed(part, dist):
    """One process per GPU (torch.distributed, backend "nccl" == RCCL).  Returns the number of levels fired."""
    if getattr(part, "may_cascade", True) is False:
        return _resolve_one_collective_per_level(part, dist)
    W = dist.get_world_size()
    levels = 0
    M = distributed_exchange(part, dist)
    while M[:, W].sum() > 0:
        M = distributed_exchange(part, dist, _dist_round(part, dist, True))
        levels += 1
    return levels


def _resolve_one_collective_per_level(part, dist):
    """For DAGs that declare ``may_cascade = False``: no satisfy that arrives from another partition can itself send
    a satisfy to a third one (no event -> remote waiter chains), so every record in flight was produced by the level
    just fired and ONE exchange per level settles it.  Per level: one engine step (closes the records applied last
    round, fires the level, one synchronisation) and one EQUAL-SPLIT all-to-all whose regions carry a 32-byte header
    {records for you, my activity} in their last record slot.  Every rank opens every level (forced when it has
    nothing ready); the loop ends after the first level in which no rank fired or sent anything, and that probe level
    is dropped again."""
    import torch
    W, rank = dist.get_world_size(), dist.get_rank()
    host = _host(dist)
    dev = part.send.device
    C = part.send.shape[1] - 1                      # record slots per region; slot C holds the header
    distributed_exchange(part, dist)                # main-time satisfies (the start event's fan-out) settle the general way
    recv = torch.empty(part.send.shape, dtype=torch.uint8, device="cpu" if host else dev)
    levels = 0
    import os
    import time
    timing = bool(os.environ.get("OCRGPU_DIST_TIMING"))      # where a level's wall time goes (rank 0 prints it)
    tm = [0.0] * 5
    while True:
        t0 = time.perf_counter()
        _, _, counts = part.eng.step(1, force_level=True, want_counts=True, write_headers=True)  # the engine fills slot C
        t1 = time.perf_counter()
        if int(counts.max()) > C:
            raise RuntimeError(f"send region overflow: {int(counts.max())} records for one partition, room for {C}")
        t2 = t1
        dist.all_to_all_single(recv.view(-1), (part.send.cpu() if host else part.send).view(-1), group=_grp())
        t3 = time.perf_counter()
        h = recv[:, C, :].cpu().contiguous().view(torch.int64).numpy()
        t4 = time.perf_counter()
        part.eng.exchange_reset()
        cin = h[:, 0].astype(np.uint32)
        if cin.any():
            rdev = recv.to(dev) if host else recv
            part.eng.exchange_apply_strided(rdev.data_ptr(), W, C + 1, cin)
        if timing:
            t5 = time.perf_counter()
            for k, dt in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                tm[k] += dt
        if h[:, 1].sum() == 0:
            break
        levels += 1
    part.eng.step(0, drop_empty_level=True)
    if timing and rank == 0 and levels:
        print("[dist timing] us per level: step %.1f | header %.1f | all-to-all %.1f | header read %.1f | reset+apply %.1f" %
              tuple(1e6 * x / (levels + 1) for x in tm), flush=True)
    return levels


def gather_result_lockstep(s: OpStream, parts) -> Result:
    level = np.full(s.n_edt, -1, np.int32)
    val = np.zeros(s.n_edt, np.uint64)
    evs = np.full(s.n_evt_total, ops.EVS_ABSENT, np.uint64)
    dbv = np.zeros(s.n_db_total, np.uint64)
    for p in parts:
        p.fill_result(level, val, evs, dbv)
    return Result(level, val, evs, dbv)
