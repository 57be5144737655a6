"""Partition-local construction of the BASELINE config-2 DAG (same definition as opstream.layered_dag) for the
multi-GPU bench: every rank derives, from the seed alone, the EDTs it owns and the dependences whose SOURCE it
owns -- no rank ever materialises the global op stream.  Ownership is round-robin (owner(e) = e % world; the start
event, the sink latch and the sink EDT live on rank 0), so remote GUIDs follow in closed form from the engines'
creation-order counters (LOCHOME | KIND | COUNTER, ocr/src/guid/guid-bitmap-based.c:70-98)."""
from __future__ import annotations

import numpy as np

from . import opstream as ops
from .distributed import REC_BYTES, mkguid
from .engine import Engine

KIND_EDT, KIND_ONCE, KIND_LATCH = 3, 17, 22


class LayeredPartition:
    def __init__(self, n_edt, width, rank, world, seed=ops.SEED_BASE + 2, device=0, flags=0, allreduce_sum=None,
                 chunk=1 << 22):
        import torch
        assert n_edt % width == 0
        self.N, self.width, self.rank, self.world, self.seed = n_edt, width, rank, world, seed
        N, W = n_edt, world
        # ---- global depc / slot_base (cheap: one RNG draw per EDT) ----
        depc = np.ones(N, dtype=np.uint32)
        for c0 in range(width, N, chunk):
            c1 = min(N, c0 + chunk)
            bits = ops.splitmix64(seed, c1 - c0, offset=c0 - width) & np.uint64(0x3F)
            pop = np.zeros(c1 - c0, dtype=np.uint32)
            for k in range(6):
                pop += ((bits >> np.uint64(k)) & np.uint64(1)).astype(np.uint32)
            depc[c0:c1] = 1 + pop
        slot_base = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(depc, out=slot_base[1:])
        self.n_edges_global = int(slot_base[N]) - width
        # ---- the dependences whose source this rank owns ----
        srcs, dsts, slots = [], [], []
        for d0 in range(width, N, chunk // 4):
            d1 = min(N, d0 + chunk // 4)
            e0, e1 = int(slot_base[d0]) - width, int(slot_base[d1]) - width
            dst = np.repeat(np.arange(d0, d1, dtype=np.int64), depc[d0:d1])
            r = ops.splitmix64(seed ^ 0xA5A5A5A5, e1 - e0, offset=e0)
            src = (dst // width - 1) * width + (r % np.uint64(width)).astype(np.int64)
            m = (src % W) == rank
            srcs.append(src[m]); dsts.append(dst[m])
            slots.append((np.arange(e0, e1, dtype=np.int64)[m] - (slot_base[dst[m]] - width)).astype(np.uint32))
        self.src = np.concatenate(srcs) if srcs else np.zeros(0, np.int64)
        self.dst = np.concatenate(dsts) if dsts else np.zeros(0, np.int64)
        self.slot = np.concatenate(slots) if slots else np.zeros(0, np.uint32)
        self.mine = np.arange(rank, N, W, dtype=np.int64)
        consumed = np.zeros(len(self.mine), dtype=bool)
        consumed[(self.src - rank) // W] = True
        self.leaves = self.mine[~consumed]
        n_leaves_local = len(self.leaves)
        self.n_leaves = int(allreduce_sum(n_leaves_local)) if allreduce_sum else n_leaves_local
        self.depc_mine = depc[self.mine]
        self.sum_depc_global = int(slot_base[N])
        del depc, slot_base
        # ---- engine ----
        n_my = len(self.mine) + (1 if rank == 0 else 0)
        n_reg = len(self.src) + n_leaves_local + (width + 1 if rank == 0 else 0)
        # records per (level, destination): the remote edges of one level spread evenly over W-1 partitions; 2x the
        # mean + slack (they are Poisson-like: sd ~ sqrt(mean)); the start event's fan-out is width/W per destination.
        # One more slot per region for the header of the one-collective-per-level exchange.
        per_pair = self.n_edges_global // max(N // width - 1, 1) // (W * W)    # from global figures: every rank must get the same cap
        # records one rank sends another per level: binomial around per_pair (10 sigma + 1024 of room); the last layer's
        # EDTs all DECR the sink latch, width/W records from every rank to the latch's owner
        cap = (max(per_pair + 10 * int(per_pair ** 0.5) + 1024, width // W + 1024) + 1) if W > 1 else 16
        self.send_cap = cap
        # no event of this DAG that a remote satisfy can reach has a remote waiter (the latch and the sink EDT both
        # live on rank 0; the start event is satisfied by rank 0 itself before the first level)
        self.may_cascade = False
        self.send = torch.zeros((W, cap, REC_BYTES), dtype=torch.uint8, device=f"cuda:{device}")
        n_in = int(self.depc_mine.sum()) + 16        # upper bound on records this rank can receive per resolve
        self.eng = Engine(max_edt=n_my, max_evt=2, max_db=(n_in if W > 1 else 0) + 16,
                          max_slots=int(self.depc_mine.sum()) + 8, max_edges=n_reg + 8, device=device, location=rank,
                          flags=flags)
        if W > 1:
            self.eng.exchange_config(W, self.send.data_ptr(), cap, n_reg + 8)
        self.eng.set_hash_tag_base(0)

    # closed-form GUIDs (creation order: rank 0 creates start, latch first; then every rank its EDTs in id order)
    def edt_guid(self, e, which=0):
        e = np.asarray(e, dtype=np.int64)
        r = np.where(e == self.N, 0, e % self.world)
        j = np.where(e == self.N, len(range(0, self.N, self.world)), e // self.world)
        base = np.where(r == 0, 3, 1)
        kind = (KIND_EDT, KIND_ONCE, 2)[which]
        return mkguid(r.astype(np.uint64), kind, (base + 3 * j + which).astype(np.uint64))

    @property
    def start_guid(self):
        return mkguid(0, KIND_ONCE, 1)

    @property
    def latch_guid(self):
        return mkguid(0, KIND_LATCH, 2)

    def build(self):
        eng, rank, N, width = self.eng, self.rank, self.N, self.width
        if rank == 0:
            g = eng.event_create([ops.EVT_ONCE, ops.EVT_LATCH], [0, self.n_leaves])
            assert int(g[0]) == int(self.start_guid) and int(g[1]) == int(self.latch_guid)
        ids = self.mine
        depc = self.depc_mine
        body = np.full(len(ids), ops.BODY_HASH | ops.BODY_F_OUTEVT, np.uint32)
        if rank == 0:
            ids = np.r_[ids, N]; depc = np.r_[depc, 1].astype(np.uint32); body = np.r_[body, ops.BODY_NOP | ops.BODY_F_SINK].astype(np.uint32)
        g, oe, od = eng.edt_create(depc, body, ids.astype(np.uint64))
        assert np.array_equal(g, self.edt_guid(ids)), "closed-form GUIDs drifted"
        src = [self.edt_guid(self.src, 1)]; dst = [self.edt_guid(self.dst)]; slot = [self.slot]
        mode = [np.full(len(self.src), ops.MODE_RO, np.uint32)]
        src.append(self.edt_guid(self.leaves, 1)); dst.append(np.full(len(self.leaves), self.latch_guid, np.uint64))
        slot.append(np.full(len(self.leaves), ops.LATCH_DECR, np.uint32)); mode.append(np.zeros(len(self.leaves), np.uint32))
        if rank == 0:
            l0 = np.arange(width, dtype=np.int64)
            src.append(np.full(width, self.start_guid, np.uint64)); dst.append(self.edt_guid(l0))
            slot.append(np.zeros(width, np.uint32)); mode.append(np.zeros(width, np.uint32))
            src.append(np.array([self.latch_guid], np.uint64)); dst.append(self.edt_guid(np.array([N])))
            slot.append(np.zeros(1, np.uint32)); mode.append(np.zeros(1, np.uint32))
        eng.add_dependence(np.concatenate(src), np.concatenate(dst), np.concatenate(slot), np.concatenate(mode))
        self.n_local_edts = len(ids)
        return self

    def satisfy_start(self):
        if self.rank == 0:
            self.eng.event_satisfy_slot(np.array([self.start_guid], np.uint64))

    @property
    def torch(self):
        import torch
        return torch

    def check(self):
        """Size-independent invariants of a finished resolve: every owned EDT fired, at the level of its layer."""
        n = self.n_local_edts
        lv, val = self.eng.read_edts(0, n)
        want = (self.mine // self.width).astype(np.int32)
        if self.rank == 0:
            want = np.r_[want, self.N // self.width].astype(np.int32)   # sink: the level after the last layer (whose outputs DECR the latch)
        bad = np.nonzero(lv != want)[0]
        assert len(bad) == 0, f"rank {self.rank}: {len(bad)} EDTs fired at the wrong level, first idx {bad[:5]} got {lv[bad[:5]]} want {want[bad[:5]]}"
        return lv, val
