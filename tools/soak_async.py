#!/usr/bin/env python3
"""Soak of the barrier-free resolve: config 2 (10M-EDT layered DAG) re-armed and resolved N times in one process; after every
resolve the device digest over (paramv[0], level, value) of all EDTs must equal the first one, and the first is held to the
CPU restatement.  k_async's order of work differs from run to run (ticket queues); what it leaves may not.

    python tools/soak_async.py [N=200] [n_edt=10000000] [width=10000]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n_edt = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
    from ocr_amd import engine
    from ocr_amd.opstream import layered_dag
    from oracle import ref as O                      # the checker (tools/ is not the product path)
    s = layered_dag(n_edt, width)
    p = engine.StreamProgram(s.with_main(s.main[:-1])).build()
    p.eng.snapshot()
    start = p._guids(s.main[-1:]["a"])
    want = O.run_restatement(s)
    ids = np.arange(s.n_edt, dtype=np.uint64)
    with np.errstate(over="ignore"):
        want_digest = int(engine.digest_term(ids, want.level[:s.n_edt], want.val[:s.n_edt]).sum(dtype=np.uint64))
    bad, ms, runs = 0, [], 0
    t_last = time.time()
    for k in range(n_rep):
        p.eng.restore()
        p.eng.event_satisfy_slot(start)
        t0 = time.perf_counter()
        st = p.resolve()
        ms.append(1e3 * (time.perf_counter() - t0))
        runs = int(st.async_runs)
        got = p.eng.digest_edts(0, s.n_edt)
        if got != want_digest or sum(st.err) or st.fired != s.n_edt:
            bad += 1
            print(f"resolve {k}: digest {got:016x} want {want_digest:016x} err {list(st.err)} fired {st.fired}", file=sys.stderr, flush=True)
        if time.time() - t_last > 20:
            print(f"[soak] {k + 1} / {n_rep} resolves, {bad} bad", file=sys.stderr, flush=True)
            t_last = time.time()
    if bad == 0:   # and the full record once more at the end
        assert p.result().diff(want) == []
    p.eng.close()
    ms = np.array(ms)
    print(json.dumps({"workload": f"layered {n_edt} x width {width}", "resolves": n_rep, "barrier_free_runs": runs, "mismatches": bad,
                      "digest": f"{want_digest:016x}", "ms_min": float(ms.min()), "ms_median": float(np.median(ms)), "ms_max": float(ms.max()),
                      "parity": "every resolve's device digest equals the CPU restatement's" if not bad else "MISMATCH"}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
