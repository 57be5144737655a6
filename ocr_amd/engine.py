"""Host-side mirror of the OCR dependence API over libocrgpu's C-ABI (include/ocrgpu.h).

Names and argument meaning follow the reference's public API (ocr/inc/ocr-edt.h, ocr/inc/ocr-db.h):
``event_create`` = ocrEventCreateParams, ``edt_create`` = ocrEdtCreate, ``add_dependence`` =
ocrAddDependence, ``event_satisfy_slot`` = ocrEventSatisfySlot, ``db_create`` = ocrDbCreate -- but every
call takes arrays (one element per API call the reference would have made).  Errors follow the
reference's convention (0 = ok, OCR_E* codes): a non-zero C return raises :class:`OcrGpuError`.

There is NO CPU path: if libocrgpu.so is missing or no MI355X is visible this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import opstream as ops
from .opstream import OpStream, Result

HOST, DEVICE = 0, 1
GUIDS_GIVEN = 4   # OR into `where` of the create calls: guid arrays are inputs (ocrgpu.h)
F_NANNY, F_TIMING = 1, 2
F_WIDE_DB = 16     # datablocks carry an 8x8 int64 tile that OCRBODY_TILE bodies update in place
F_NO_ASYNC = 32    # ocrgpu_resolve always level by level (by default an eligible, narrow DAG resolves as one barrier-free kernel)
F_LOCKABLE_DB = 8  # datablocks arbitrate their acquire modes (the cfg's datablockfactory=Lockable)
EFAULT = -14       # OCRGPU_EFAULT: the device flagged a user-level error during the call (which completed)


def LATCH_DELTA(k):
    """slot value of one latch op that stands for the net of k INCRs (k < 0: DECRs) -- OCRGPU_LATCH_DELTA (ocrgpu.h)"""
    return 0x40000000 + int(k)


# ocrGuidKind values in the guid KIND bits (ocrgpu.h)
KIND_DB, KIND_EDT, KIND_EVT_ONCE, KIND_EVT_COUNTED, KIND_EVT_IDEM, KIND_EVT_STICKY, KIND_EVT_LATCH = 2, 3, 17, 18, 19, 21, 22
_LIB_PATH = os.environ.get("OCRGPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libocrgpu.so")


class OcrGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libocrgpu error {code}: {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [("max_edt", C.c_uint32), ("max_evt", C.c_uint32), ("max_db", C.c_uint32), ("location", C.c_uint32),
                ("max_slots", C.c_uint64), ("max_edges", C.c_uint64), ("guid_slots", C.c_uint64),
                ("max_levels", C.c_uint32), ("flags", C.c_uint32)]


class Reclaim(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("edts_freed", "events_freed", "edts_held", "edges_kept", "free_edt", "free_evt", "free_db",
                                          "fresh_edt", "fresh_evt", "fresh_db", "slots_used", "ready_kept")]


class Stats(C.Structure):
    _fields_ = [("levels", C.c_uint64), ("fired", C.c_uint64), ("edt_sat", C.c_uint64), ("evt_sat", C.c_uint64),
                ("launches", C.c_uint64), ("step_kernel_ns", C.c_uint64), ("err", C.c_uint64 * 8), ("async_runs", C.c_uint64)]

    def as_dict(self):
        return dict(levels=self.levels, fired=self.fired, edt_sat=self.edt_sat, evt_sat=self.evt_sat,
                    launches=self.launches, step_kernel_ns=self.step_kernel_ns, err=list(self.err), async_runs=self.async_runs)


SCRIPT_OP_DT = np.dtype([("target", "<u8"), ("payload", "<u8"), ("slot", "<u4"), ("op", "u1"), ("payload_kind", "u1"),
                         ("pad", "<u2")])
assert SCRIPT_OP_DT.itemsize == 24

EXPORTS = ["ocrgpu_init", "ocrgpu_fini", "ocrgpu_set_stream", "ocrgpu_stream_create", "ocrgpu_last_error", "ocrgpu_abi_version",
           "ocrgpu_guid_reserve", "ocrgpu_guid_put", "ocrgpu_guid_tryput", "ocrgpu_guid_get", "ocrgpu_guid_remove",
           "ocrgpu_event_create", "ocrgpu_db_create", "ocrgpu_edt_create", "ocrgpu_edt_set_scripts",
           "ocrgpu_set_hash_tag_base", "ocrgpu_add_dependence", "ocrgpu_event_satisfy_slot", "ocrgpu_event_destroy", "ocrgpu_edt_release_dbs",
           "ocrgpu_resolve", "ocrgpu_step", "ocrgpu_step_info", "ocrgpu_pop_ready", "ocrgpu_pop_ready_noalloc", "ocrgpu_exchange_apply_strided", "ocrgpu_exchange_config", "ocrgpu_exchange_counts",
           "ocrgpu_exchange_reset", "ocrgpu_exchange_apply", "ocrgpu_snapshot", "ocrgpu_restore", "ocrgpu_sync", "ocrgpu_read_edts",
           "ocrgpu_read_events", "ocrgpu_read_dbs", "ocrgpu_read_fired", "ocrgpu_get_stats",
           "ocrgpu_comm_get_unique_id", "ocrgpu_comm_init", "ocrgpu_comm_fini", "ocrgpu_comm_alltoall", "ocrgpu_exchange",
           "ocrgpu_exchange_local", "ocrgpu_resolve_exchange", "ocrgpu_resolve_exchange_local", "ocrgpu_workpile_push",
           "ocrgpu_workpile_pop", "ocrgpu_db_destroy", "ocrgpu_reclaim", "ocrgpu_peer_export", "ocrgpu_peer_connect",
           "ocrgpu_peer_fini", "ocrgpu_digest_edts"]
PEER_HANDLE_BYTES = 128
RX_NO_CASCADE = 1
RX_TIME_STEPS = 2
WP_PUSH, WP_PUSHBACK, WP_POP, WP_STEAL = 0, 1, 0, 1

_lib = None


def digest_term(param0, level, value):
    """include/ocrgpu.h:ocrgpu_digest_term on numpy arrays (the host-side twin the tests hold the device fold against)"""
    with np.errstate(over="ignore"):
        z = np.asarray(param0, np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.asarray(level, np.int32).astype(np.uint32).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = z ^ np.asarray(value, np.uint64)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def load_library(path=_LIB_PATH):
    """dlopen libocrgpu.so (no GPU needed for this) and check every symbol include/ocrgpu.h declares."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise OcrGpuError(-2, f"{path} not built (run `python -m ocr_amd.build`); there is no CPU fallback")
    lib = C.CDLL(path)
    for name in EXPORTS:
        try:
            fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        except AttributeError:
            if os.environ.get("OCRGPU_LIB_LAX"):   # A/B runs against an older build of the library (tools/, never the tests)
                continue
            raise
        fn.restype = C.c_int
    lib.ocrgpu_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):  # raw device pointer
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class Engine:
    """One GPU partition of the dependence engine (one policy-domain location)."""

    def __init__(self, max_edt, max_evt=16, max_db=16, max_slots=None, max_edges=None, device=0, location=0,
                 flags=0, max_levels=0, guid_slots=0, tolerate_faults=False):
        """tolerate_faults: a call that returns OCRGPU_EFAULT (the device counted a user-level error, did the rest of its
        work and says so) is recorded in ``self.faults`` instead of raising -- for tests of ill-formed programs."""
        self.lib = load_library()
        self.tolerate_faults, self.faults = tolerate_faults, 0
        cfg = Config(max_edt=max_edt, max_evt=max_evt, max_db=max_db, location=location,
                     max_slots=max_slots if max_slots is not None else 8 * max_edt,
                     max_edges=max_edges if max_edges is not None else 8 * max_edt, guid_slots=guid_slots,
                     max_levels=max_levels, flags=flags)
        self.cfg = cfg
        self.h = C.c_void_p()
        self._chk(self.lib.ocrgpu_init(C.byref(self.h), device, C.byref(cfg)))

    def _chk(self, rc):
        if rc == EFAULT and self.tolerate_faults:
            self.faults += 1
            return
        if rc != 0:
            raise OcrGpuError(rc, (self.lib.ocrgpu_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.ocrgpu_fini(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- GUID provider (ocrGuidProviderFcts_t) ----
    def guid_reserve(self, kind, n):
        out = np.empty(n, dtype=np.uint64)
        self._chk(self.lib.ocrgpu_guid_reserve(self.h, C.c_uint32(kind), C.c_uint64(n), _ptr(out), HOST))
        return out

    def guid_put(self, guids, vals):
        guids, vals = np.ascontiguousarray(guids, np.uint64), np.ascontiguousarray(vals, np.uint64)
        self._chk(self.lib.ocrgpu_guid_put(self.h, C.c_uint64(len(guids)), _ptr(guids), _ptr(vals), HOST))

    def guid_tryput(self, guids, vals):
        """insert unless present; returns the value the table holds for each guid afterwards (hashtable TryPut)"""
        guids, vals = np.ascontiguousarray(guids, np.uint64), np.ascontiguousarray(vals, np.uint64)
        out = np.empty(len(guids), dtype=np.uint64)
        self._chk(self.lib.ocrgpu_guid_tryput(self.h, C.c_uint64(len(guids)), _ptr(guids), _ptr(vals), _ptr(out), HOST))
        return out

    def guid_get(self, guids):
        guids = np.ascontiguousarray(guids, np.uint64)
        out = np.empty(len(guids), dtype=np.uint64)
        self._chk(self.lib.ocrgpu_guid_get(self.h, C.c_uint64(len(guids)), _ptr(guids), _ptr(out), HOST))
        return out

    def guid_remove(self, guids):
        guids = np.ascontiguousarray(guids, np.uint64)
        self._chk(self.lib.ocrgpu_guid_remove(self.h, C.c_uint64(len(guids)), _ptr(guids), HOST))

    # ---- objects ----
    def event_create(self, kind, arg=None, guids=None):
        """guids: caller-chosen guids from guid_reserve (OCRGPU_GUIDS_GIVEN) instead of engine-made ones."""
        kind = np.ascontiguousarray(kind, np.uint32)
        arg = np.zeros(len(kind), np.int32) if arg is None else np.ascontiguousarray(arg, np.int32)
        given = guids is not None
        out = np.ascontiguousarray(guids, np.uint64) if given else np.empty(len(kind), dtype=np.uint64)
        self._chk(self.lib.ocrgpu_event_create(self.h, C.c_uint64(len(kind)), _ptr(kind), _ptr(arg), _ptr(out),
                                               HOST | (GUIDS_GIVEN if given else 0)))
        return out

    def db_create(self, value, tag=None, guids=None):
        value = np.ascontiguousarray(value, np.uint64)
        tag = None if tag is None else np.ascontiguousarray(tag, np.uint64)
        given = guids is not None
        out = np.ascontiguousarray(guids, np.uint64) if given else np.empty(len(value), dtype=np.uint64)
        self._chk(self.lib.ocrgpu_db_create(self.h, C.c_uint64(len(value)), _ptr(value), _ptr(tag), _ptr(out),
                                            HOST | (GUIDS_GIVEN if given else 0)))
        return out

    def edt_create(self, depc, body, param0=None, guids=None, outevt_guids=None):
        depc, body = np.ascontiguousarray(depc, np.uint32), np.ascontiguousarray(body, np.uint32)
        param0 = None if param0 is None else np.ascontiguousarray(param0, np.uint64)
        n = len(depc)
        if guids is not None:
            g = np.ascontiguousarray(guids, np.uint64)
            oe = None if outevt_guids is None else np.ascontiguousarray(outevt_guids, np.uint64)
            self._chk(self.lib.ocrgpu_edt_create(self.h, C.c_uint64(n), _ptr(depc), _ptr(body), _ptr(param0), _ptr(g),
                                                 _ptr(oe), None, HOST | GUIDS_GIVEN))
            return g, oe, None
        g, oe, od = (np.empty(n, dtype=np.uint64) for _ in range(3))
        self._chk(self.lib.ocrgpu_edt_create(self.h, C.c_uint64(n), _ptr(depc), _ptr(body), _ptr(param0), _ptr(g),
                                             _ptr(oe), _ptr(od), HOST))
        return g, oe, od

    def pop_ready_noalloc(self, max_edts=32768, max_deps=131072):
        """ocrgpu_pop_ready_noalloc: the same hand-over on the library's static device scratch (no allocation; the
        capacities are bounded by it and a larger request is refused)."""
        p0, dc, df = np.empty(max_edts, np.uint64), np.empty(max_edts, np.uint32), np.empty(max_edts, np.uint32)
        dg, dv = np.empty(max_deps, np.uint64), np.empty(max_deps, np.uint64)
        ne, nd = C.c_uint64(0), C.c_uint64(0)
        self._chk(self.lib.ocrgpu_pop_ready_noalloc(self.h, C.c_uint64(max_edts), C.c_uint64(max_deps), C.byref(ne), C.byref(nd),
                                                    _ptr(p0), _ptr(dc), _ptr(df), _ptr(dg), _ptr(dv)))
        return p0[:ne.value], dc[:ne.value], df[:ne.value], dg[:nd.value], dv[:nd.value]

    def pop_ready(self, max_edts=1 << 16, max_deps=1 << 18):
        """EDTs with OCRBODY_HOST bodies that fired since the last call: (param0[n], depc[n], dep_first[n],
        dep_guid[m], dep_value[m]); grows its buffers and retries when more is ready than fits."""
        while True:
            p0, dc, df = np.empty(max_edts, np.uint64), np.empty(max_edts, np.uint32), np.empty(max_edts, np.uint32)
            dg, dv = np.empty(max_deps, np.uint64), np.empty(max_deps, np.uint64)
            ne, nd = C.c_uint64(0), C.c_uint64(0)
            rc = self.lib.ocrgpu_pop_ready(self.h, C.c_uint64(max_edts), C.c_uint64(max_deps), C.byref(ne), C.byref(nd),
                                           _ptr(p0), _ptr(dc), _ptr(df), _ptr(dg), _ptr(dv))
            if rc == -12 and (ne.value > max_edts or nd.value > max_deps):
                max_edts, max_deps = max(max_edts, ne.value), max(max_deps, nd.value)
                continue
            self._chk(rc)
            return p0[:ne.value], dc[:ne.value], df[:ne.value], dg[:nd.value], dv[:nd.value]

    def edt_set_scripts(self, edt, begin, length, script_ops):
        edt = np.ascontiguousarray(edt, np.uint64)
        begin, length = np.ascontiguousarray(begin, np.uint32), np.ascontiguousarray(length, np.uint32)
        script_ops = np.ascontiguousarray(script_ops, SCRIPT_OP_DT)
        self._chk(self.lib.ocrgpu_edt_set_scripts(self.h, C.c_uint64(len(edt)), _ptr(edt), _ptr(begin), _ptr(length),
                                                  C.c_uint64(len(script_ops)), _ptr(script_ops), HOST))

    def set_hash_tag_base(self, base):
        self._chk(self.lib.ocrgpu_set_hash_tag_base(self.h, C.c_uint64(base)))

    # ---- dependences ----
    def add_dependence(self, src, dst, slot, mode, where=HOST):
        if where == HOST:
            src, dst = np.ascontiguousarray(src, np.uint64), np.ascontiguousarray(dst, np.uint64)
            slot, mode = np.ascontiguousarray(slot, np.uint32), np.ascontiguousarray(mode, np.uint32)
            n = len(src)
        else:
            src, dst, slot, mode, n = src
        self._chk(self.lib.ocrgpu_add_dependence(self.h, C.c_uint64(n), _ptr(src), _ptr(dst), _ptr(slot), _ptr(mode), where))

    def event_satisfy_slot(self, evt, db=None, slot=None):
        evt = np.ascontiguousarray(evt, np.uint64)
        db = np.zeros(len(evt), np.uint64) if db is None else np.ascontiguousarray(db, np.uint64)
        slot = np.zeros(len(evt), np.uint32) if slot is None else np.ascontiguousarray(slot, np.uint32)
        self._chk(self.lib.ocrgpu_event_satisfy_slot(self.h, C.c_uint64(len(evt)), _ptr(evt), _ptr(db), _ptr(slot), HOST))

    def event_destroy(self, evt):
        evt = np.ascontiguousarray(evt, np.uint64)
        self._chk(self.lib.ocrgpu_event_destroy(self.h, C.c_uint64(len(evt)), _ptr(evt), HOST))

    def db_destroy(self, guids):
        """ocrDbDestroy: guids stop resolving, indices go back (output DBs of HASH EDTs: their EDT's line may go)."""
        g = np.ascontiguousarray(guids, dtype=np.uint64)
        self._chk(self.lib.ocrgpu_db_destroy(self.h, C.c_uint64(len(g)), _ptr(g), HOST))

    def reclaim(self):
        """Take back the rows of finished EDTs and destroyed events; returns ocrgpu_reclaim_t as a dict."""
        out = Reclaim()
        self._chk(self.lib.ocrgpu_reclaim(self.h, C.byref(out)))
        return {n: int(getattr(out, n)) for n, _ in Reclaim._fields_}

    def edt_release_dbs(self, edt):
        """F_LOCKABLE_DB, host-run bodies: the EDTs named have run; release the datablocks they hold."""
        edt = np.ascontiguousarray(edt, np.uint64)
        self._chk(self.lib.ocrgpu_edt_release_dbs(self.h, C.c_uint64(len(edt)), _ptr(edt), HOST))

    # ---- worker loop ----
    def resolve(self, max_levels=0):
        st = Stats()
        self._chk(self.lib.ocrgpu_resolve(self.h, C.c_uint64(max_levels), C.byref(st)))
        return st

    def step(self, fire_levels=1, force_level=False, want_counts=False, drop_empty_level=False, write_headers=False):
        """Close pending satisfies, fire at most ``fire_levels`` levels, close again.  Returns
        (has_ready, pending_records[, send_counts]) -- all from the one synchronisation the run ends with."""
        hr, pr = C.c_uint64(), C.c_uint64()
        counts = np.zeros(self.n_locations, np.uint32) if want_counts else None
        self._chk(self.lib.ocrgpu_step(self.h, C.c_uint32(fire_levels),
                                       C.c_uint32((1 if force_level else 0) | (2 if drop_empty_level else 0) | (4 if write_headers else 0)),
                                       C.byref(hr), C.byref(pr), _ptr(counts)))
        if want_counts:
            return bool(hr.value), int(pr.value), counts
        return bool(hr.value), int(pr.value)

    def step_info(self):
        """(ready-log entries of the level the last step fired, level, levels recorded, queued satisfies)."""
        out = (C.c_uint64 * 4)()
        self._chk(self.lib.ocrgpu_step_info(self.h, out))
        return tuple(int(x) for x in out)

    # ---- cross-partition exchange ----
    def exchange_config(self, n_locations, send_dev_ptr, records_per_dest, max_remote_dsts):
        self.n_locations = n_locations
        self._chk(self.lib.ocrgpu_exchange_config(self.h, C.c_uint32(n_locations), C.c_void_p(send_dev_ptr),
                                                  C.c_uint64(records_per_dest), C.c_uint64(max_remote_dsts)))

    def exchange_counts(self):
        out = np.zeros(self.n_locations, np.uint32)
        self._chk(self.lib.ocrgpu_exchange_counts(self.h, _ptr(out)))
        return out

    def exchange_reset(self):
        self._chk(self.lib.ocrgpu_exchange_reset(self.h))

    def exchange_apply(self, dev_ptr, n):
        self._chk(self.lib.ocrgpu_exchange_apply(self.h, C.c_void_p(dev_ptr), C.c_uint64(n), DEVICE))

    def exchange_apply_strided(self, dev_ptr, n_src, stride_records, counts):
        counts = np.ascontiguousarray(counts, np.uint32)
        self._chk(self.lib.ocrgpu_exchange_apply_strided(self.h, C.c_void_p(dev_ptr), C.c_uint32(n_src),
                                                         C.c_uint64(stride_records), _ptr(counts)))

    # ---- the comm-platform over RCCL (C transport) ----
    def comm_init(self, unique_id: bytes, rank, nranks):
        """ncclCommInitRank for this engine (collective; exchange_config comes first)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.lib.ocrgpu_comm_init(self.h, buf, C.c_int(rank), C.c_int(nranks)))

    def comm_fini(self):
        self._chk(self.lib.ocrgpu_comm_fini(self.h))

    # ---- peer windows: the exchange of one node without a collective (ocrgpu_peer_*) ----
    def peer_export(self) -> bytes:
        """Make this partition's window (exchange_config comes first); the 128-byte handle goes to every partition."""
        buf = C.create_string_buffer(PEER_HANDLE_BYTES)
        self._chk(self.lib.ocrgpu_peer_export(self.h, buf))
        return buf.raw

    def peer_connect(self, handles):
        """handles: the handles of ALL partitions in location order (a list of bytes or one bytes object)."""
        raw = b"".join(handles) if isinstance(handles, (list, tuple)) else bytes(handles)
        buf = C.create_string_buffer(raw, len(raw))
        self._chk(self.lib.ocrgpu_peer_connect(self.h, buf))

    def peer_fini(self):
        self._chk(self.lib.ocrgpu_peer_fini(self.h))

    def comm_alltoall(self, send_dev_ptr, recv_dev_ptr, bytes_per_peer):
        self._chk(self.lib.ocrgpu_comm_alltoall(self.h, C.c_void_p(send_dev_ptr), C.c_void_p(recv_dev_ptr), C.c_uint64(bytes_per_peer)))

    def exchange(self):
        """One exchange round over RCCL: (records received here, activity of all, records sent by all, satisfies queued)."""
        out = [C.c_uint64() for _ in range(4)]
        self._chk(self.lib.ocrgpu_exchange(self.h, *[C.byref(x) for x in out]))
        return tuple(int(x.value) for x in out)

    def resolve_exchange(self, no_cascade=False, time_steps=False):
        """The worker loop across partitions, in C over peer windows or RCCL (collective).  Returns the number of levels fired."""
        lv = C.c_uint64()
        flags = (RX_NO_CASCADE if no_cascade else 0) | (RX_TIME_STEPS if time_steps else 0)
        self._chk(self.lib.ocrgpu_resolve_exchange(self.h, C.c_uint32(flags), C.byref(lv)))
        return int(lv.value)

    # ---- workpile (ocrWorkpileFcts_t push / pop) ----
    def workpile_push(self, guids, back=False):
        guids = np.ascontiguousarray(guids, np.uint64)
        self._chk(self.lib.ocrgpu_workpile_push(self.h, C.c_uint32(WP_PUSHBACK if back else WP_PUSH), C.c_uint64(len(guids)), _ptr(guids), HOST))

    def workpile_pop(self, max_n, steal=False):
        out = np.empty(max_n, np.uint64)
        n = C.c_uint64()
        self._chk(self.lib.ocrgpu_workpile_pop(self.h, C.c_uint32(WP_STEAL if steal else WP_POP), C.c_uint64(max_n), _ptr(out), C.byref(n), HOST))
        return out[:n.value]

    def snapshot(self):
        self._chk(self.lib.ocrgpu_snapshot(self.h))

    def restore(self):
        self._chk(self.lib.ocrgpu_restore(self.h))

    def sync(self):
        self._chk(self.lib.ocrgpu_sync(self.h))

    def stream_create(self):
        """a stream of the engine's own in place of the NULL stream (several engines in one process: ocrgpu.h)"""
        self._chk(self.lib.ocrgpu_stream_create(self.h))

    def stats(self):
        st = Stats()
        self._chk(self.lib.ocrgpu_get_stats(self.h, C.byref(st)))
        return st

    # ---- results ----
    def read_edts(self, first, n):
        lv, val = np.empty(n, np.int32), np.empty(n, np.uint64)
        self._chk(self.lib.ocrgpu_read_edts(self.h, C.c_uint64(first), C.c_uint64(n), _ptr(lv), _ptr(val), HOST))
        return lv, val

    def digest_edts(self, first, n):
        """sum modulo 2^64 of ocrgpu_digest_term(paramv[0], level, value) over EDTs [first, first+n), folded on the device"""
        out = C.c_uint64()
        self._chk(self.lib.ocrgpu_digest_edts(self.h, C.c_uint64(first), C.c_uint64(n), C.byref(out)))
        return out.value

    def read_events(self, which, first, n):
        out = np.empty(n, np.uint64)
        self._chk(self.lib.ocrgpu_read_events(self.h, which, C.c_uint64(first), C.c_uint64(n), _ptr(out), HOST))
        return out

    def read_dbs(self, which, first, n):
        out = np.empty(n, np.uint64)
        self._chk(self.lib.ocrgpu_read_dbs(self.h, which, C.c_uint64(first), C.c_uint64(n), _ptr(out), HOST))
        return out

    def read_fired(self):
        nf, nl = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.ocrgpu_read_fired(self.h, C.byref(nf), C.byref(nl), None, None, HOST))
        ids = np.empty(nf.value, np.uint32)
        ls = np.zeros(nl.value + 1, np.uint64)
        self._chk(self.lib.ocrgpu_read_fired(self.h, C.byref(nf), C.byref(nl), _ptr(ids), _ptr(ls), HOST))
        return ids, ls


def comm_get_unique_id() -> bytes:
    """ncclGetUniqueId through libocrgpu (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.ocrgpu_comm_get_unique_id(buf)
    if rc != 0:
        raise OcrGpuError(rc, (lib.ocrgpu_last_error() or b"").decode())
    return buf.raw


def _engine_array(engines):
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    return arr


def exchange_local(engines):
    """One exchange round among the partitions of this process (locations 0..n-1 on one GPU): (activity, sent, queued)."""
    lib = engines[0].lib
    out = [C.c_uint64() for _ in range(3)]
    engines[0]._chk(lib.ocrgpu_exchange_local(_engine_array(engines), C.c_uint32(len(engines)), *[C.byref(x) for x in out]))
    return tuple(int(x.value) for x in out)


def resolve_exchange_local(engines, no_cascade=False):
    lib = engines[0].lib
    lv = C.c_uint64()
    engines[0]._chk(lib.ocrgpu_resolve_exchange_local(_engine_array(engines), C.c_uint32(len(engines)),
                                                       C.c_uint32(RX_NO_CASCADE if no_cascade else 0), C.byref(lv)))
    return int(lv.value)


class StreamProgram:
    """An op stream loaded into an :class:`Engine`: the same role oracle/ocr_replay.c plays for the
    reference runtime, but every op becomes (part of) one batched C-ABI call."""

    def __init__(self, s: OpStream, device=0, flags=0, engine=None, slack=0, tolerate_faults=False, reuse=False, progress=None):
        self.s = s
        self.progress = progress   # called with a line of text after every chunk of a long build (None: silent)
        # reuse: the engine may hand out reclaimed rows (ocrgpu_reclaim), so creation order is not index order: ask the
        # GUID table for every object's index right after its create call
        self.reuse = reuse
        self.edt_index = np.full(s.n_edt, -1, np.int64)
        self.evt_index = np.full(s.n_evt, -1, np.int64)
        self.db_index = np.full(s.n_db, -1, np.int64)
        n_slots = int(s.edt["depc"].sum())
        n_adddep = int((s.main["op"] == ops.OP_ADD_DEP).sum())
        if s.db_words == 64:
            flags |= F_WIDE_DB
        self.eng = engine or Engine(max_edt=max(s.n_edt, 1) + slack, max_evt=max(s.n_evt, 1), max_db=max(s.n_db, 1),
                                    max_slots=n_slots + 1, max_edges=n_adddep + 1 + ops.channel_ring_slots(s), device=device, flags=flags,
                                    tolerate_faults=tolerate_faults)
        self.edt_guid = np.zeros(s.n_edt, np.uint64)
        self.evt_guid = np.zeros(s.n_evt_total, np.uint64)
        self.db_guid = np.zeros(s.n_db_total, np.uint64)
        self.edt_order = []  # stream EDT idx in engine creation order
        self.evt_order = []
        self.db_order = []
        self.eng.set_hash_tag_base(s.n_db)
        self._built = False

    ADD_DEP_CHUNK = 1 << 26

    def _guids(self, refs, depv_ok=False):
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
        out[cls == ops.REF_UNINIT] = np.uint64(0xFFFFFFFFFFFFFFFE)
        return out

    def build(self):
        """Replay the main ops as batched API calls (consecutive ops of one kind = one call)."""
        s, eng = self.s, self.eng
        main = s.main
        if len(main):
            cuts = np.nonzero(np.diff(main["op"].astype(np.int16)))[0] + 1
            bounds = np.r_[0, cuts, len(main)]
        else:
            bounds = np.array([0])
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            blk = main[b0:b1]
            op = int(blk["op"][0])
            if op == ops.OP_EVT_CREATE:
                idx = blk["a"].astype(np.int64)
                self.evt_guid[idx] = eng.event_create(s.evt["kind"][idx], s.evt["arg"][idx])
                self.evt_order += idx.tolist()
                if self.reuse:
                    self.evt_index[idx] = eng.guid_get(self.evt_guid[idx]).astype(np.int64) - 1
            elif op == ops.OP_DB_CREATE:
                idx = blk["a"].astype(np.int64)
                self.db_guid[idx] = eng.db_create(s.db_init[idx], idx.astype(np.uint64))
                self.db_order += idx.tolist()
                if self.reuse:
                    self.db_index[idx] = eng.guid_get(self.db_guid[idx]).astype(np.int64) - 1
            elif op == ops.OP_EDT_CREATE:
                idx = blk["a"].astype(np.int64)
                g, oe, od = eng.edt_create(s.edt["depc"][idx], s.edt["body"][idx], idx.astype(np.uint64))
                self.edt_guid[idx], self.evt_guid[s.n_evt + idx], self.db_guid[s.n_db + idx] = g, oe, od
                self.edt_order += idx.tolist()
                if self.reuse:
                    self.edt_index[idx] = eng.guid_get(g).astype(np.int64) - 1
            elif op == ops.OP_ADD_DEP:
                for c0, c1 in (ops.split_conflicts(ops.channel_keys(s, blk["a"])) if ops.channel_ring_slots(s) else [(0, len(blk))]):   # registrations on one CHANNEL keep program order
                    for k0 in range(c0, c1, self.ADD_DEP_CHUNK):   # bounded host arrays; a line of progress per chunk
                        b = blk[k0:min(c1, k0 + self.ADD_DEP_CHUNK)]
                        eng.add_dependence(self._guids(b["a"]), self._guids(b["b"]), b["slot"], b["mode"].astype(np.uint32))
                        if self.progress and c1 - c0 > self.ADD_DEP_CHUNK:
                            self.progress(f"add_dependence: {min(c1, k0 + self.ADD_DEP_CHUNK) - c0} of {c1 - c0} registered")
            elif op == ops.OP_SATISFY:
                for c0, c1 in ops.split_conflicts(blk["a"]):   # two satisfies of one event keep program order
                    b = blk[c0:c1]
                    eng.event_satisfy_slot(self._guids(b["a"]), self._guids(b["b"]), b["slot"])
            elif op == ops.OP_EVT_DESTROY:
                eng.event_destroy(self._guids(blk["a"]))
            else:
                raise ValueError(f"bad op {op}")
        self._set_scripts()   # bodies only run in resolve(): by now every event a script names exists
        self._built = True
        return self

    def _set_scripts(self):
        s = self.s
        if getattr(self, "_scripts_done", False) or len(s.script) == 0:
            return
        has = np.nonzero(s.edt["script_len"] > 0)[0]
        if np.any(self.edt_guid[has] == 0):
            return  # some scripted EDT not created yet
        sc = np.zeros(len(s.script), dtype=SCRIPT_OP_DT)
        sc["target"] = self._guids(s.script["a"])
        sc["slot"] = s.script["slot"]
        sc["op"] = s.script["op"]
        cls = ops.ref_cls(s.script["b"])
        pk = np.zeros(len(sc), np.uint8)
        pay = np.zeros(len(sc), np.uint64)
        m = (cls == ops.REF_DB)
        pk[m], pay[m] = 1, self._guids(s.script["b"][m])
        m = (cls == ops.REF_OUTDB)
        pk[m] = 2  # by construction scripts only name their own output DB
        m = (cls == ops.REF_DEPV)
        pk[m], pay[m] = 3, ops.ref_idx(s.script["b"][m])
        sc["payload_kind"], sc["payload"] = pk, pay
        self.eng.edt_set_scripts(self.edt_guid[has], s.edt["script_begin"][has], s.edt["script_len"][has], sc)
        self._scripts_done = True

    def resolve(self):
        return self.eng.resolve()

    def _result_by_index(self) -> Result:
        s, eng = self.s, self.eng
        assert (self.edt_index >= 0).all() and (self.evt_index >= 0).all() and (self.db_index >= 0).all()
        evs = np.full(s.n_evt_total, ops.EVS_ABSENT, np.uint64)
        dbv = np.zeros(s.n_db_total, np.uint64)
        level, v = np.zeros(0, np.int32), np.zeros(0, np.uint64)
        if s.n_edt:
            ne = int(self.edt_index.max()) + 1
            lv, val = eng.read_edts(0, ne)
            level, v = lv[self.edt_index], val[self.edt_index]
            evs[s.n_evt:] = eng.read_events(1, 0, ne)[self.edt_index]
            dbv[s.n_db:] = eng.read_dbs(1, 0, ne)[self.edt_index]
        if s.n_evt:
            evs[:s.n_evt] = eng.read_events(0, 0, int(self.evt_index.max()) + 1)[self.evt_index]
        if s.n_db:
            dbv[:s.n_db] = eng.read_dbs(0, 0, int(self.db_index.max()) + 1)[self.db_index]
        return Result(level, v, evs, dbv)

    def result(self) -> Result:
        if self.reuse:
            return self._result_by_index()
        s, eng = self.s, self.eng
        lv, val = eng.read_edts(0, s.n_edt) if s.n_edt else (np.zeros(0, np.int32), np.zeros(0, np.uint64))
        eo = np.asarray(self.edt_order, dtype=np.int64)
        level = np.full(s.n_edt, -1, np.int32)
        v = np.zeros(s.n_edt, np.uint64)
        level[eo], v[eo] = lv[:len(eo)], val[:len(eo)]
        evs = np.full(s.n_evt_total, ops.EVS_ABSENT, np.uint64)
        if s.n_evt:
            xo = np.asarray(self.evt_order, dtype=np.int64)
            evs[xo] = eng.read_events(0, 0, len(xo))
        if s.n_edt:
            evs[s.n_evt + eo] = eng.read_events(1, 0, len(eo))
        dbv = np.zeros(s.n_db_total, np.uint64)
        if s.n_db:
            do = np.asarray(self.db_order, dtype=np.int64)
            dbv[do] = eng.read_dbs(0, 0, len(do))
        if s.n_edt:
            dbv[s.n_db + eo] = eng.read_dbs(1, 0, len(eo))
        return Result(level, v, evs, dbv)


def run_stream(s: OpStream, device=0, flags=F_NANNY, tolerate_faults=False):
    """Build + resolve + read back. Returns (Result, Stats)."""
    p = StreamProgram(s, device=device, flags=flags, tolerate_faults=tolerate_faults).build()
    st = p.resolve()
    r = p.result()
    p.eng.close()
    return r, st
