#!/usr/bin/env python3
"""Sweep of random programs with contended Lockable datablocks (tests/lockable.py:random_lockable, sizes 20..220) beyond the
committed seeds: one partition and, every second seed, 2..4 partitions over one of the three loops, each held to the bar of
tests/test_gpu_lockable*.py (everything but `level` bit-identical to the restatement; every EDT fires, no earlier than its
dependences allow; holders of a datablock compatible level by level).

    python tools/fuzz_sweep_lockable.py [first_seed=1000] [count=600]
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
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    import lockable as L
    from ocr_amd import distributed as D
    from ocr_amd import engine
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    bad, runs, t0 = [], 0, time.time()

    def check(s, r, base, what):
        try:
            assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all() and (r.db_val == base.db_val).all(), "record"
            L.check_levels(s, r.level, base.level)
        except AssertionError as ex:
            bad.append((s.name, what, str(ex)[:200]))

    for seed in range(first, first + count):
        s = L.random_lockable(seed, n=20 + (seed * 29) % 201)
        base = O.run_restatement(s)
        r, st = engine.run_stream(s, flags=engine.F_LOCKABLE_DB)
        runs += 1
        if sum(st.err):
            bad.append((s.name, "one partition", list(st.err)))
        check(s, r, base, "one partition")
        if seed % 2 == 0:
            world, loop = 2 + (seed // 2) % 3, ("lockstep", "c", "windows")[(seed // 6) % 3]
            parts = D.build_lockstep([D.Partition(s, k, world, flags=engine.F_LOCKABLE_DB) for k in range(world)])
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
            if any(sum(e) for e in errs):
                bad.append((s.name, f"{world} partitions, {loop}", errs))
            check(s, res, base, f"{world} partitions, {loop}")
        if (seed - first + 1) % 50 == 0:
            print(f"[fuzz_sweep_lockable] {seed - first + 1} / {count} seeds, {runs} runs, {len(bad)} failures, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    for b in bad[:20]:
        print("FAIL", b, file=sys.stderr, flush=True)
    print(json.dumps({"first_seed": first, "seeds": count, "runs": runs, "failures": len(bad), "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
