"""TEST INFRASTRUCTURE: Python handles on the two checkers.

* :func:`run_restatement` -- oracle/engine_ref.c (the CPU restatement) through ctypes.
* :func:`run_reference`   -- the real reference runtime replaying the stream
  (oracle/_ref/ocr_replay_*, built by oracle/Makefile.ref from /root/reference where present;
  on the GPU box only the prebuilt binaries exist).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import re
import subprocess
import tempfile

import numpy as np

from ocr_amd.opstream import HDR_DT, MAGIC, OpStream, Result

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libengine_ref.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(force=False):
    """Compile the restatement (always) and, when /root/reference is present, the reference runtime."""
    src = os.path.join(HERE, "engine_ref.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), src, "-o", LIB])
    if os.path.isdir("/root/reference/ocr/src"):
        subprocess.check_call(["make", "-s", "-f", "Makefile.ref", "-j8", "all"], cwd=HERE,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call(["make", "-s", "-f", "Makefile.ref", "progs"], cwd=HERE,
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.ocrref_run.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_restatement(s: OpStream, return_stats=False, lockable=False):
    """``lockable``: datablocks arbitrate their acquire modes like the cfg's datablockfactory=Lockable (else Regular)."""
    lib = _load()
    hdr = s.header()
    level = np.empty(s.n_edt, dtype=np.int32)
    val = np.empty(s.n_edt, dtype=np.uint64)
    evs = np.empty(s.n_evt_total, dtype=np.uint64)
    dbv = np.empty(s.n_db_total, dtype=np.uint64)
    stats = np.zeros(16, dtype=np.uint64)
    rc = lib.ocrref_run2(_p(hdr), _p(s.edt), _p(s.evt), _p(s.db_init), _p(s.main), _p(s.script),
                         _p(level), _p(val), _p(evs), _p(dbv), _p(stats), ctypes.c_uint32(1 if lockable else 0))
    assert rc == 0
    r = Result(level, val, evs, dbv)
    return (r, stats) if return_stats else r


def reference_available(flavour="dbg"):
    return os.path.exists(os.path.join(REF_DIR, f"ocr_replay_{flavour}"))


def run_reference(s: OpStream, workers=1, guid="COUNTED_MAP", flavour="dbg", timeout=600, want_result=True, db="Regular",
                  lock_harness=None):
    """Replay ``s`` on the reference runtime. Returns (Result|None, info dict with the timing line).
    ``db``: the cfg's datablock factory, "Regular" or "Lockable" (the latter also switches the harness's clocks to
    count datablock hand-overs and makes it watch for overlapping holders; see oracle/ocr_replay.c)."""
    from oracle.mkcfg import cfg
    exe = os.path.join(REF_DIR, f"ocr_replay_{flavour}")
    with tempfile.TemporaryDirectory(prefix="ocrref_") as td:
        ops, res, cf = (os.path.join(td, n) for n in ("s.ops", "s.res", "m.cfg"))
        s.to_file(ops)
        with open(cf, "w") as f:
            f.write(cfg(workers, guid, db))
        env = dict(os.environ)
        if (db == "Lockable") if lock_harness is None else lock_harness:
            env["OCR_REPLAY_LOCKABLE"] = "1"
        out = subprocess.run([exe, "-ocr:cfg", cf, ops, res if want_result else "-"], capture_output=True,
                             text=True, timeout=timeout, env=env)
        if out.returncode != 0:
            raise RuntimeError(f"reference replay failed rc={out.returncode}: {out.stdout[-500:]} {out.stderr[-500:]}")
        m = re.search(r"REPLAY fired=(\d+) edges=(\d+) resolve_s=([\d.]+) sat_per_s=([\d.]+)", out.stdout)
        info = dict(fired=int(m.group(1)), edges=int(m.group(2)), resolve_s=float(m.group(3)),
                    sat_per_s=float(m.group(4))) if m else {}
        m = re.search(r"REPLAY_LOCKABLE overlaps=(\d+)", out.stdout)
        if m:
            info["overlaps"] = int(m.group(1))
        r = Result.from_file(res) if want_result else None
    return r, info
