#!/usr/bin/env python3
"""Single-GPU measurements of the BASELINE configs other than bench.py's default (config 2): same step as bench.py
(restore + satisfy(start) + resolve on a DAG resident in HBM), full-size parity against the CPU restatement, one JSON
line per workload.  Not the contract bench: a measurement tool whose output is quoted in DESIGN.md s5.

    python tools/bench_configs.py cholesky:256 stencil:2048:10 powerlaw:20000000 layered:10000000:100000
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make(spec):
    from ocr_amd import opstream as op
    k, *a = spec.split(":")
    a = [int(x) for x in a]
    if k == "cholesky":
        return op.cholesky_dag(a[0])
    if k == "choleskytiles":   # config 3 with 8x8 int64 tile datablocks updated in place (OCRBODY_TILE)
        return op.cholesky_dag(a[0], tiles=True)
    if k == "stencil":
        return op.stencil_dag(a[0], a[1])
    if k == "powerlaw":
        return op.powerlaw_dag(a[0], progress=stage, stable_argsort=gpu_stable_argsort if a[0] >= 20_000_000 else None)
    if k == "layered":
        return op.layered_dag(a[0], a[1])
    if k == "layerednop":   # config 2's DAG with NOP bodies: dependence resolution alone, no datablock checksums
        s = op.layered_dag(a[0], a[1])
        n = a[0]
        s.edt["body"][:n] = op.BODY_NOP | op.BODY_F_OUTEVT
        return s
    if k == "layeredperm":   # config 2's DAG with the EDT indices shuffled: a layer's lines lie all over the table, not in one 640-KB run
        s = op.layered_dag(a[0], a[1])
        n = a[0]
        perm = np.random.default_rng(7).permutation(n).astype(np.uint64)
        edt = s.edt.copy()
        edt[perm.astype(np.int64)] = s.edt[:n]
        s.edt[:n] = edt[:n]
        for f in ("a", "b"):
            r = s.main[f]
            cls, idx = op.ref_cls(r), op.ref_idx(r)
            m = ((cls == op.REF_EDT) | (cls == op.REF_OUTEVT) | (cls == op.REF_OUTDB)) & (idx < np.uint64(n)) & (s.main["op"] != op.OP_EDT_CREATE)
            s.main[f][m] = (r[m] & np.uint64(0xFF00000000000000)) | perm[idx[m].astype(np.int64)]
        return s
    if k == "lockable":   # lockable:N:width:n_db -- layered DAG + contended datablocks, run with OCRGPU_F_LOCKABLE_DB
        return op.lockable_layered(a[0], a[1], a[2])
    raise SystemExit(f"unknown workload {spec}")


def gpu_stable_argsort(keys):
    """stable argsort of a large int64 key array on the GPU (torch as a sort routine for the TEST-DATA generator: numpy's stable sort
    of config 5's 1.15e9 edge keys takes minutes on the host)"""
    import torch
    k = torch.from_numpy(keys).cuda()
    order = torch.sort(k, stable=True).indices
    del k
    out = order.cpu().numpy()
    del order
    torch.cuda.empty_cache()
    return out


def stage(msg):
    """one line per finished stage, so a long run shows where it is (and a hung stage shows as silence)"""
    print(f"[bench_configs {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def run(spec, steps=3, warmup=1, parity=True):
    from ocr_amd import engine, opstream
    from ocr_amd import opstream as ops
    t0 = time.time()
    s = make(spec)
    gen_s = time.time() - t0
    stage(f"{spec}: generated {s.n_edt} EDTs, {len(s.main)} main ops in {gen_s:.1f} s")
    assert int(s.main["op"][-1]) == ops.OP_SATISFY, "generators end with the start satisfy"
    start_op = s.main[-1:]
    built = s.with_main(s.main[:-1])
    t0 = time.time()
    lockable = spec.startswith("lockable")
    prog = engine.StreamProgram(built, device=0, flags=engine.F_TIMING | (engine.F_LOCKABLE_DB if lockable else 0), progress=stage).build()
    eng = prog.eng
    eng.snapshot()
    eng.sync()
    build_s = time.time() - t0
    stage(f"{spec}: built on the engine in {build_s:.1f} s")
    start_guid = prog._guids(start_op["a"])
    start_db = prog._guids(start_op["b"])

    tm = {"restore": 0.0, "satisfy": 0.0, "resolve": 0.0}

    def step(timed=False):
        t0 = time.perf_counter()
        eng.restore()
        if timed:
            eng.sync()
        t1 = time.perf_counter()
        eng.event_satisfy_slot(start_guid, start_db, start_op["slot"])
        if timed:
            eng.sync()
        t2 = time.perf_counter()
        st = eng.resolve()
        t3 = time.perf_counter()
        if timed:
            tm["restore"] += t1 - t0
            tm["satisfy"] += t2 - t1
            tm["resolve"] += t3 - t2
        return st

    for _ in range(warmup):
        st = step()
    stage(f"{spec}: warm-up done ({int(st.fired)} EDTs fired, {int(st.levels)} levels)")
    k0, l0 = eng.stats().step_kernel_ns, eng.stats().launches
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        st = step()
    eng.sync()
    dt = (time.perf_counter() - t0) / steps
    kern = (eng.stats().step_kernel_ns - k0) * 1e-9 / steps
    launches = (eng.stats().launches - l0) / steps
    assert sum(st.err) == 0, list(st.err)
    for _ in range(steps):
        st = step(timed=True)
    n_sat = int(st.edt_sat + st.evt_sat)
    out = {"workload": spec, "n_edt": int(s.n_edt), "n_evt": int(s.n_evt), "n_db": int(s.n_db), "fired": int(st.fired),
           "levels": int(st.levels), "edt_sat": int(st.edt_sat), "evt_sat": int(st.evt_sat),
           "meta_n_satisfactions": int(s.meta.get("n_satisfactions", 0)),
           "ms_per_resolve": dt * 1e3, "ms_breakdown_synced": {k: round(v / steps * 1e3, 3) for k, v in tm.items()}, "ms_k_step": kern * 1e3, "launches": launches,
           "sat_per_s": int(s.meta.get("n_satisfactions", n_sat)) / dt, "edts_per_s": int(st.fired) / dt,
           "gen_s": round(gen_s, 2), "build_s": round(build_s, 2)}
    stage(f"{spec}: timed {steps} steps, {dt * 1e3:.2f} ms per resolve")
    if parity:
        from oracle import ref as O            # checker only
        t0 = time.time()
        os.environ.setdefault("OCRREF_PROGRESS", "30")   # the restatement of a 125M-EDT stream runs for minutes: one line per 30 s
        ref = O.run_restatement(s)
        got = prog.result()
        if lockable:   # who wins a datablock is a race (here and in the reference): levels through the checker, the rest exactly
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            import lockable as L
            same = (got.val == ref.val).all() and (got.evt_state == ref.evt_state).all() and (got.db_val == ref.db_val).all()
            try:
                L.check_levels(s, got.level, ref.level)
                lv_ok = True
            except AssertionError as ex:
                lv_ok = str(ex)
            r2, st2 = O.run_restatement(s, return_stats=True, lockable=True)
            out["levels_regular"], out["levels_restatement_lockable"] = int(ref.level.max()) + 1, int(st2[0])
            out["parity"] = ("values/events/DBs bit-exact vs CPU restatement, levels valid under the Lockable checker"
                             if same and lv_ok is True else f"MISMATCH same={same} levels={lv_ok}")
        else:
            diff = got.diff(ref)
            out["parity"] = "bit-exact vs CPU restatement" if not diff else f"MISMATCH {diff[:3]}"
        out["restatement_s"] = round(time.time() - t0, 2)
    print(json.dumps(out), flush=True)
    eng.close()          # free the gigabytes NOW: left to the garbage collector, the hipFree lands in the next workload's timed loop
    del prog, eng
    return out


if __name__ == "__main__":
    specs = [a for a in sys.argv[1:] if not a.startswith("--")]
    if any(sp.startswith("powerlaw:") for sp in specs):   # the generator sorts on the GPU through torch, which ships its own HIP runtime:
        import torch  # noqa: F401                          # it has to be loaded before libocrgpu.so pulls in /opt/rocm's (ocr_amd/distributed.py)
    for sp in specs:
        run(sp, parity="--no-parity" not in sys.argv)
