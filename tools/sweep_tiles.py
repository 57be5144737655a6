#!/usr/bin/env python3
"""Tile datablocks across partitions beyond the goldens: tiled Cholesky (8x8 int64 tiles updated in place, OCRBODY_TILE) for every
T in [2, 24] on 2..8 partitions, loops in turn, against the CPU restatement (levels, checksums, final contents of every tile).

    python tools/sweep_tiles.py
"""
import json
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))


def main():
    from ocr_amd import distributed as D
    from ocr_amd import opstream as ops
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    bad, runs, t0 = [], 0, time.time()
    for T in range(2, 25):
        s = ops.cholesky_dag(T, tiles=True)
        want = O.run_restatement(s)
        for world in range(2, 9):
            loop = ("lockstep", "c", "windows")[(T + world) % 3]
            parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
            if loop == "lockstep":
                D.resolve_lockstep(parts)
            else:
                if loop == "windows":
                    D.connect_peer_windows(parts)
                D.resolve_lockstep_c(parts)
            res = D.gather_result_lockstep(s, parts)
            errs = [list(p.eng.stats().err) for p in parts]
            for p in parts:
                p.eng.close()
            runs += 1
            if any(sum(e) for e in errs) or res.diff(want) or res.n_fired != s.n_edt:
                bad.append((T, world, loop, errs, res.diff(want)[:3]))
        print(f"[sweep_tiles] T = {T}: {runs} runs, {len(bad)} mismatches, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    for b in bad[:20]:
        print("MISMATCH", b, file=sys.stderr, flush=True)
    print(json.dumps({"what": "cholesky tiles T=2..24 on 2..8 partitions", "runs": runs, "mismatches": len(bad), "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
