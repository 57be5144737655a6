#!/usr/bin/env python3
"""Where does the tail of a k_step launch come from?  Runs one resolve of config 2 on the -DOCRGPU_STAMPS build
(ocr_amd/libocrgpu_stamps.so: hipcc ... -DOCRGPU_STAMPS) with per-workgroup time stamps of 8
consecutive launches, and prints, per launch, the phase times by XCD and by workgroup index.

    OCRGPU_LIB=ocr_amd/libocrgpu_stamps.so OCRGPU_STAMP_K=600 OCRGPU_STAMP_DUMP=gpurun_out/stamps.bin python tools/stamp_tail.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MAX_GRID = 2048
NAMES = ["entry", "exit", "decided", "lines", "scan", "fanout", "loopend", "stats"]


def analyse(path):
    a = np.fromfile(path, dtype=np.uint64)
    tabs = a[16:].reshape(8, MAX_GRID, 8)
    for j in range(8):
        t = tabs[j]
        used = (t[:, 1] != 0) & (t[:, 0] != 0)
        if not used.any():
            continue
        xcc = (t[:, 0] >> np.uint64(60)).astype(int)
        tt = t.copy()
        tt[:, 0] &= np.uint64((1 << 60) - 1)
        t0 = tt[used, 0].min()
        rel = (tt[used].astype(np.int64) - int(t0)) / 100.0       # 100 MHz counter -> us
        idx = np.nonzero(used)[0]
        work = rel[:, 5] > 0                                      # reached the fan-out stamp = had work
        print(f"launch +{j}: {used.sum()} workgroups stamped, {work.sum()} with work; exit us min/med/p90/max = "
              f"{rel[work,1].min():.2f}/{np.median(rel[work,1]):.2f}/{np.percentile(rel[work,1],90):.2f}/{rel[work,1].max():.2f}")
        for ph in range(2, 8):
            v = rel[work, ph]
            v = v[v > 0]
            if len(v):
                print(f"   {NAMES[ph]:8s} min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v,90):6.2f} max {v.max():6.2f}")
        # by XCD
        line = "   exit by XCD (med/max):"
        for x in range(8):
            m = work & (xcc[idx] == x)
            if m.any():
                line += f"  x{x}:{np.median(rel[m,1]):.1f}/{rel[m,1].max():.1f}({m.sum()})"
        print(line)
        # by workgroup index octile
        wi = idx[work]
        order = np.argsort(wi)
        ex = rel[work, 1][order]
        parts = np.array_split(ex, 8)
        print("   exit med by workgroup-index octile: " + " ".join(f"{np.median(p):.1f}" for p in parts))
        # the slowest 5 workgroups, all phases
        slow = np.argsort(-rel[:, 1])[:5]
        for sidx in slow:
            print(f"   slow wg {idx[sidx]:4d} xcd {xcc[idx[sidx]]}: " + " ".join(f"{NAMES[p]}={rel[sidx,p]:.2f}" for p in range(8)))


if __name__ == "__main__":
    dump = os.environ.get("OCRGPU_STAMP_DUMP", "gpurun_out/stamps.bin")
    if len(sys.argv) > 1 and sys.argv[1] == "--analyse":
        analyse(dump)
        raise SystemExit
    import torch  # noqa: F401  (before libocrgpu: one HIP runtime per process)
    from ocr_amd import engine, opstream
    n, w = int(os.environ.get("STAMP_N", 10_000_000)), int(os.environ.get("STAMP_W", 10_000))   # a wide level: STAMP_W=100000
    s = opstream.layered_dag(n, w)
    built = s.with_main(s.main[:-1])
    prog = engine.StreamProgram(built, device=0, flags=0).build()
    eng = prog.eng
    eng.snapshot()
    eng.event_satisfy_slot(prog._guids(s.main[-1:]["a"]))
    st = eng.resolve()
    print("fired", st.fired, "levels", st.levels, "launches", st.launches, flush=True)
    eng.close()                                                   # fini dumps the tables
    analyse(dump)
