#!/usr/bin/env python3
"""BASELINE config 3 "1 and 8 GPUs" on the one-GPU rig: the tiled Cholesky (8x8 int64 tiles updated in place) at T = 256 on `world`
partitions that share this GPU, joined by peer windows, a stream per partition; wall time of the resolve and parity against the CPU
restatement (levels, checksums, final tiles).

    python tools/tiles_partitioned_timing.py [T=256] [world=8]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from ocr_amd import distributed as D
    from ocr_amd import opstream as ops
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    s = ops.cholesky_dag(T, tiles=True)
    want = O.run_restatement(s)
    kw = dict(send_cap=1 << 21, proxy_cap=1_500_000) if T >= 128 else {}
    parts = D.build_lockstep([D.Partition(s, r, world, **kw) for r in range(world)])
    D.connect_peer_windows(parts)
    for p in parts:
        p.eng.sync()
    t0 = time.perf_counter()
    levels = D.resolve_lockstep_c(parts)
    for p in parts:
        p.eng.sync()
    dt = time.perf_counter() - t0
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    ok = not any(sum(e) for e in errs) and res.diff(want) == [] and res.n_fired == s.n_edt
    print(json.dumps({"workload": f"cholesky tiles T={T} on {world} partitions of one GPU (peer windows)", "n_edt": int(s.n_edt), "levels": int(levels),
                      "ms_per_resolve": round(1e3 * dt, 2), "us_per_level": round(1e6 * dt / max(1, int(levels)), 1),
                      "parity": "bit-exact vs CPU restatement (levels, checksums, all tiles)" if ok else "MISMATCH"}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
