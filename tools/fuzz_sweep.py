#!/usr/bin/env python3
"""Fuzz sweep beyond the committed seeds: random well-formed OCR programs (tests/streams.py:random_stream, both flavours, sizes
24..192) on the engine against the CPU restatement -- one partition (plain and NANNY) and, every third seed, 2..4 partitions
over one of the three loops (host-driven lockstep, the C loop, the C loop over peer windows).  One line per 100 seeds, one JSON
line at the end; exit status 1 on any mismatch.

    python tools/fuzz_sweep.py [first_seed=1000] [count=2000]
"""
import json
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
sys.path.insert(0, os.path.join(os.path.dirname(here), "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    import streams
    from ocr_amd import distributed as D
    from ocr_amd import engine
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    bad, runs, t0 = [], 0, time.time()
    for seed in range(first, first + count):
        n = 24 + (seed * 37) % 169
        s = streams.random_stream(seed, n=n, eager_safe=bool(seed & 1))
        want = O.run_restatement(s)
        for flags in ((0, engine.F_NANNY) if seed % 4 == 0 else (0,)):
            r, st = engine.run_stream(s, flags=flags)
            runs += 1
            if sum(st.err) or r.diff(want):
                bad.append((seed, n, f"flags {flags}", list(st.err), r.diff(want)[:3]))
        if seed % 3 == 0:
            world, loop = 2 + (seed // 3) % 3, ("lockstep", "c", "windows")[(seed // 9) % 3]
            parts = D.build_lockstep([D.Partition(s, k, world) for k in range(world)])
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
            if any(sum(e) for e in errs) or res.diff(want):
                bad.append((seed, n, f"{world} partitions, {loop}", errs, res.diff(want)[:3]))
        if (seed - first + 1) % 100 == 0:
            print(f"[fuzz_sweep] {seed - first + 1} / {count} seeds, {runs} runs, {len(bad)} mismatches, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    for b in bad[:20]:
        print("MISMATCH", b, file=sys.stderr, flush=True)
    print(json.dumps({"first_seed": first, "seeds": count, "runs": runs, "mismatches": len(bad), "bad_seeds": [b[0] for b in bad][:50],
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
