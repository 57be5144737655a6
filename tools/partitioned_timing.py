#!/usr/bin/env python3
"""A workload of tools/bench_configs.py on `world` partitions that share this GPU, joined by peer windows, a stream per partition
(the one-GPU rig of the multi-GPU loop): wall time of the resolve and parity against the CPU restatement.  BASELINE config 3
"1 and 8 GPUs": choleskytiles:256 (8x8 int64 tiles updated in place; levels, checksums and the final tiles are compared).

    python tools/partitioned_timing.py [spec=choleskytiles:256] [world=8]
    python tools/partitioned_timing.py lockable:200000:10000:4096 4        # Lockable datablocks arbitrated by their owners
    python tools/partitioned_timing.py stencil:1024:10 8                   # config 4: block-partitioned 2 x 4 (OCRGPU_STENCIL_BLOCKS=0: by GUID hash)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "choleskytiles:256"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import bench_configs
    from ocr_amd import distributed as D
    from ocr_amd.engine import F_LOCKABLE_DB
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    s = bench_configs.make(spec)
    lockable = spec.startswith("lockable:")
    want = O.run_restatement(s)
    kw = dict(send_cap=1 << 21, proxy_cap=1_500_000) if spec.startswith("choleskytiles:") and int(spec.split(":")[1]) >= 128 else {}
    if lockable:
        kw["flags"] = F_LOCKABLE_DB
    if spec.startswith("stencil:") and os.environ.get("OCRGPU_STENCIL_BLOCKS", "1") != "0":   # config 4's placement: blocks of tiles, only halo edges cross
        G, steps = (int(x) for x in spec.split(":")[1:3])
        kw["owner_edt"] = D.stencil_block_owner(G, steps, world)
        kw["send_cap"] = max(1 << 16, 1 << int(G * G // world).bit_length())   # the start event's fan-out reaches every tile of step 0; after it a round carries one level's halo
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
    if lockable:   # the bar of tests/test_gpu_lockable*.py: everything but `level` bit-identical, levels valid under the checker
        import lockable as L
        ok = not any(sum(e) for e in errs) and bool((res.val == want.val).all() and (res.evt_state == want.evt_state).all() and (res.db_val == want.db_val).all())
        try:
            L.check_levels(s, res.level, want.level)
        except AssertionError:
            ok = False
    else:
        ok = not any(sum(e) for e in errs) and res.diff(want) == [] and res.n_fired == s.n_edt
    print(json.dumps({"workload": f"{spec} on {world} partitions of one GPU (peer windows)", "n_edt": int(s.n_edt), "levels": int(levels),
                      "ms_per_resolve": round(1e3 * dt, 2), "us_per_level": round(1e6 * dt / max(1, int(levels)), 1),
                      "parity": ("values / events / datablocks bit-exact, levels valid under the Lockable checker" if lockable else "bit-exact vs CPU restatement") if ok else "MISMATCH"}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
