"""BASELINE config 4 at its stated LENGTH (5-point stencil, 1000 time steps) on one GPU's share of the
tile grid, as a sliding window over time: the program builds a few steps ahead of the one that fires, and what has
fired is given back (``ocrgpu_db_destroy`` + ``ocrgpu_reclaim``) -- the reference frees an EDT when its body has returned
(destructTaskHc, hc-task.c:721-790) and an event when it is satisfied (destructEventHc, hc-event.c:185-242), so a run of
1000 steps there never holds more than what is in flight either.  The engine is sized for W + 2 steps, not for 1000.

The OCR program is the one ``opstream.stencil_dag(G, steps)`` writes down (same EDT ids in paramv[0], same dependences,
same start event / sink latch / sink EDT), only issued window by window; ``tests/test_gpu_stencil_window.py`` holds the
record it leaves to the CPU restatement of that stream on a twin that fits the restatement.

All batch arrays are made ON the GPU (torch tensors as plain device memory, ``OCRGPU_DEVICE`` pointers): no op stream
of 2e9 EDTs ever exists on the host.

A window:  build steps (fired, fired + W]  ->  ocrgpu_step(fire W levels)  ->  check / fold the W steps that fired  ->
ocrgpu_db_destroy(output datablocks whose readers have all run)  ->  ocrgpu_reclaim.  The newest built step stays READY
but unfired across the reclaim: its ONCE output events must not fire before the next window's EDTs are registered on them
(a ONCE event satisfied without waiters drops the satisfy, hc-event.c:450-464).  Levels start over at every reclaim; the
level of tile-step (t, i, j) in its window is t - (first step the window fired)."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import opstream as ops
from .engine import DEVICE, Engine, digest_term

_M64 = (1 << 64) - 1


def _s64(x):
    """the two's-complement int64 that has the bits of the unsigned 64-bit x (torch has no uint64 arithmetic)"""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def _lsr(z, k):
    """logical shift right of int64 tensors"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def t_mix64(z):
    """ocrops_mix64 (include/ocr_opstream.h) on int64 torch tensors, wrapping like the unsigned original"""
    z = z + _s64(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


def t_digest_term(param0, level, value):
    """ocrgpu_digest_term (include/ocrgpu.h) on int64 torch tensors"""
    z = param0 * _s64(0x9E3779B97F4A7C15) + level
    z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = z ^ value
    z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
    return z ^ _lsr(z, 31)


class StencilWindow:
    """5-point stencil on GX x GY tiles for `steps` steps through an engine that holds `window` + 2 steps."""

    def __init__(self, gx, gy, steps, window, device=0, flags=0, keep_record=False, progress=None):
        import torch
        self.torch = torch
        self.gx, self.gy, self.steps, self.W = gx, gy, steps, max(1, min(window, steps))
        self.dev = torch.device(f"cuda:{device}")
        self.keep_record, self.progress = keep_record, progress
        n = self.n_tile = gx * gy
        self.N = n * steps
        # the dependences of one time step, exactly as opstream.stencil_dag lays them out: direction order self, N, S, W, E;
        # the pre-slots of a tile are its valid directions in that order
        t = torch.arange(n, device=self.dev, dtype=torch.int64)
        ii, jj = t // gy, t % gy
        nbr, valid = [], []
        for di, dj in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)):
            ni, nj = ii + di, jj + dj
            ok = (ni >= 0) & (ni < gx) & (nj >= 0) & (nj < gy)
            nbr.append(torch.where(ok, ni * gy + nj, torch.zeros_like(ni))); valid.append(ok)
        self.nbr, self.valid = torch.stack(nbr, 1), torch.stack(valid, 1)            # [tile, 5]
        self.slot_of = torch.cumsum(self.valid.to(torch.int64), 1) - 1
        self.fanin = self.valid.sum(1).to(torch.int32)
        ti, di = torch.nonzero(self.valid, as_tuple=True)
        self.e_src_tile, self.e_dst_tile = self.nbr[ti, di].contiguous(), ti.contiguous()
        self.e_slot = self.slot_of[ti, di].to(torch.int32).contiguous()
        self.e_mode = torch.full_like(self.e_slot, ops.MODE_RO)
        self.E_step = int(self.e_slot.numel())
        self.body = torch.full((n,), ops.BODY_HASH | ops.BODY_F_OUTEVT, dtype=torch.int32, device=self.dev)
        self.ones = torch.ones(n, dtype=torch.int32, device=self.dev)
        self.n_satisfactions = n + (steps - 1) * self.E_step + n + 1 + 1   # start fan-out, dependences, latch DECRs, latch -> sink, start
        alive = self.W + 2
        self.eng = Engine(max_edt=alive * n + 8, max_evt=4, max_db=4, max_slots=8, max_edges=alive * self.E_step + 2 * n + 64,
                          device=device, flags=flags)
        self.eng.set_hash_tag_base(0)        # no explicit datablocks: the output datablock of EDT e is datablock e, as in the stream
        self.lib, self.h = self.eng.lib, self.eng.h
        # step t -> (EDT guids, output-event guids, output-datablock guids), device tensors, dropped once reclaimed
        self.g = {}
        self.built = self.fired = self.db_gone = 0
        self.rows = self.free_rows = 0        # EDT rows ever used / on the free stack (what ocrgpu_read_edts may be asked for)
        self.digest = 0                       # sum mod 2^64 of ocrgpu_digest_term(id, GLOBAL level = t, value) over what has fired
        self.tm = dict(build=0.0, fire=0.0, check=0.0, reclaim=0.0)
        self.windows = 0
        self.prev_val = None                  # values of the newest fired step (the recurrence's input), [tile]
        self.record_level = np.full(self.N + 1, -1, np.int32) if keep_record else None
        self.record_val = np.zeros(self.N + 1, np.uint64) if keep_record else None
        self.level_errors = self.value_errors = 0

    # ---- C-ABI calls with device pointers ----
    def _chk(self, rc):
        self.eng._chk(rc)

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def _edt_create(self, depc, body, param0):
        torch, n = self.torch, depc.numel()
        g, oe, od = (torch.empty(n, dtype=torch.int64, device=self.dev) for _ in range(3))
        self._chk(self.lib.ocrgpu_edt_create(self.h, C.c_uint64(n), self._p(depc), self._p(body), self._p(param0), self._p(g), self._p(oe),
                                             self._p(od), DEVICE))
        take = min(n, self.free_rows)         # reclaimed rows are served first (ocrgpu.h)
        self.free_rows -= take; self.rows += n - take
        return g, oe, od

    def _add_dep(self, src, dst, slot, mode):
        self._chk(self.lib.ocrgpu_add_dependence(self.h, C.c_uint64(src.numel()), self._p(src), self._p(dst), self._p(slot), self._p(mode), DEVICE))

    # ---- the program ----
    def _prologue(self):
        torch, eng = self.torch, self.eng
        ev = eng.event_create([ops.EVT_ONCE, ops.EVT_LATCH], [0, self.n_tile])
        self.start_guid, self.latch_guid = int(ev[0]), int(ev[1])
        g, _, _ = eng.edt_create([1], [ops.BODY_NOP | ops.BODY_F_SINK], [self.N])
        self.sink_guid = int(g[0])
        self.rows += 1
        eng.add_dependence([self.latch_guid], [self.sink_guid], [0], [ops.MODE_NULL])

    def _build_step(self, t):
        torch, n = self.torch, self.n_tile
        ids = torch.arange(t * n, (t + 1) * n, device=self.dev, dtype=torch.int64)
        g, oe, od = self._edt_create(self.ones if t == 0 else self.fanin, self.body, ids)
        self.g[t] = (g, oe, od)
        if t == 0:
            src = torch.full((n,), _s64(self.start_guid), dtype=torch.int64, device=self.dev)
            zero = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self._add_dep(src, g, zero, zero)                                    # start event -> slot 0, DB_MODE_NULL
        else:
            poe = self.g[t - 1][1]
            self._add_dep(poe[self.e_src_tile], g[self.e_dst_tile], self.e_slot, self.e_mode)
        if t == self.steps - 1:                                                  # the last step's output events check out of the sink latch
            dst = torch.full((n,), _s64(self.latch_guid), dtype=torch.int64, device=self.dev)
            zero = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self._add_dep(oe, dst, zero, zero)                                   # OCR_EVENT_LATCH_DECR_SLOT, DB_MODE_NULL

    def expected_values(self, t, prev_val):
        """the HASH fold of step t from step t-1's values: an independent, vectorised restatement of OCRBODY_HASH
        (ocrops_hash_init / ocrops_hash_step) for this DAG"""
        torch, n = self.torch, self.n_tile
        ids = torch.arange(t * n, (t + 1) * n, device=self.dev, dtype=torch.int64)
        v = t_mix64(ids ^ 0x0C5200000000)
        if t == 0:
            return t_mix64(v)                                                    # one NULL_GUID pre-slot: v ^ (0 + 0 * K + 0)
        K = _s64(ops_HASH_K)
        for d in range(5):
            p = self.nbr[:, d]
            pid = (t - 1) * n + p
            cand = t_mix64(v ^ (prev_val[p] + (pid + 1) * K + self.slot_of[:, d]))
            v = torch.where(self.valid[:, d], cand, v)
        return v

    def _check_window(self, first, k):
        """steps [first, first + k) fired in this window, at levels 0 .. k-1: check every level and every value on the device"""
        torch, n = self.torch, self.n_tile
        ne = self.rows
        lv = torch.empty(ne, dtype=torch.int32, device=self.dev)
        val = torch.empty(ne, dtype=torch.int64, device=self.dev)
        self._chk(self.lib.ocrgpu_read_edts(self.h, C.c_uint64(0), C.c_uint64(ne), self._p(lv), self._p(val), DEVICE))
        for t in range(first, first + k):
            g = self.g[t][0]
            idx = torch.empty(n, dtype=torch.int64, device=self.dev)
            self._chk(self.lib.ocrgpu_guid_get(self.h, C.c_uint64(n), self._p(g), self._p(idx), DEVICE))
            idx = idx - 1
            got_lv, got_val = lv[idx], val[idx]
            self.level_errors += int((got_lv != (t - first)).sum())
            want = self.expected_values(t, self.prev_val)
            self.value_errors += int((got_val != want).sum())
            self.prev_val = got_val
            ids = torch.arange(t * n, (t + 1) * n, device=self.dev, dtype=torch.int64)
            self.digest = (self.digest + (int(t_digest_term(ids, torch.full_like(ids, t), got_val).sum().item()) & _M64)) & _M64
            if self.keep_record:
                self.record_level[t * n:(t + 1) * n] = t
                self.record_val[t * n:(t + 1) * n] = got_val.cpu().numpy().view(np.uint64)
                if self.level_errors:
                    self.record_level[t * n:(t + 1) * n] = (got_lv.cpu().numpy() + first)

    def run(self):
        torch, eng, W, T = self.torch, self.eng, self.W, self.steps
        sync = torch.cuda.synchronize
        t_all = time.perf_counter()
        self._prologue()
        started = False
        while self.fired < T:
            t0 = time.perf_counter()
            hi = min(T, self.fired + W + 1)                   # build up to and including step fired + W
            while self.built < hi:
                self._build_step(self.built); self.built += 1
            if not started:
                eng.event_satisfy_slot([self.start_guid]); started = True
            sync(); t1 = time.perf_counter()
            last = self.built == T
            k = T - self.fired if last else self.built - 1 - self.fired
            if last:
                st = eng.resolve()                            # to the end: the last steps, the sink latch, the sink EDT
                levels = int(st.levels)
            else:
                eng.step(fire_levels=k)
                levels = k
            sync(); t2 = time.perf_counter()
            self._check_window(self.fired, k)
            if last:
                lv, _ = self._sink_level()
                self.sink_level_in_window = lv
                if self.keep_record:
                    self.record_level[self.N] = self.fired + lv
            sync(); t3 = time.perf_counter()
            first = self.fired
            self.fired += k
            # output datablocks whose readers (the next step) have all run: everything below the newest fired step
            gone_to = self.fired if last else self.fired - 1
            if gone_to > self.db_gone:
                od = torch.cat([self.g[t][2] for t in range(self.db_gone, gone_to)])
                self._chk(self.lib.ocrgpu_db_destroy(self.h, C.c_uint64(od.numel()), self._p(od), DEVICE))
                for t in range(self.db_gone, gone_to):
                    del self.g[t]
                self.db_gone = gone_to
            r = eng.reclaim()
            self.free_rows, self.rows = r["free_edt"], int(eng.cfg.max_edt) - r["fresh_edt"]
            sync(); t4 = time.perf_counter()
            self.tm["build"] += t1 - t0; self.tm["fire"] += t2 - t1; self.tm["check"] += t3 - t2; self.tm["reclaim"] += t4 - t3
            self.windows += 1
            self.last_reclaim = r
            if self.progress:
                self.progress(f"window {self.windows}: steps {first}..{self.fired - 1} fired ({levels} levels, {(t2 - t1) * 1e3:.1f} ms), "
                              f"reclaimed {r['edts_freed']} EDTs, held {r['edts_held']}, ready kept {r['ready_kept']}, edges kept {r['edges_kept']}; "
                              f"level errors {self.level_errors}, value errors {self.value_errors}")
        self.wall_s = time.perf_counter() - t_all
        return self

    def _sink_level(self):
        idx = int(self.eng.guid_get(np.array([self.sink_guid], np.uint64))[0]) - 1
        lv, val = self.eng.read_edts(idx, 1)
        return int(lv[0]), int(val[0])

    def summary(self):
        st = self.eng.stats()
        return dict(workload=f"stencil {self.gx}x{self.gy} x {self.steps} steps, window {self.W}", n_edt=self.N + 1,
                    n_satisfactions=self.n_satisfactions, windows=self.windows, wall_s=round(self.wall_s, 3),
                    s_build=round(self.tm["build"], 3), s_fire=round(self.tm["fire"], 3), s_check=round(self.tm["check"], 3),
                    s_reclaim=round(self.tm["reclaim"], 3),
                    sat_per_s_fire_phase=self.n_satisfactions / max(self.tm["fire"], 1e-9),
                    sat_per_s_whole_program=self.n_satisfactions / max(self.tm["build"] + self.tm["fire"] + self.tm["reclaim"], 1e-9),
                    level_errors=self.level_errors, value_errors=self.value_errors, digest=f"{self.digest:016x}",
                    engine_max_edt=int(self.eng.cfg.max_edt), err=[int(x) for x in st.err], fired_total=int(st.fired))


ops_HASH_K = 0xD6E8FEB86659FD93   # OCROPS_HASH_K (include/ocr_opstream.h)


def record_digest(level, val, n):
    """the same digest from a full record (numpy; the restatement's result on the twin): EDTs 0..n-1"""
    return int(digest_term(np.arange(n, dtype=np.uint64), level[:n], val[:n]).sum(dtype=np.uint64))
