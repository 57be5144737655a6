#!/usr/bin/env python3
"""GUID-provider throughput on device-resident batches (rows a12/a13): put / get / remove of N guids in a table at
load factor N/slots.  Algorithmic bytes per lookup: one 16-B slot probe x expected probes.
    python tools/bench_guid.py [log2_n=26] [log2_slots=28]
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from ocr_amd import engine as E
    lg_n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    lg_s = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    n = 1 << lg_n
    eng = E.Engine(max_edt=16, guid_slots=1 << lg_s)
    lib, h = eng.lib, eng.h
    g = torch.empty(n, dtype=torch.int64, device="cuda")
    v = torch.randint(1, 1 << 60, (n,), dtype=torch.int64, device="cuda")
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    eng._chk(lib.ocrgpu_guid_reserve(h, C.c_uint32(E.KIND_DB), C.c_uint64(n), P(g), E.DEVICE))
    perm = torch.randperm(n, device="cuda")
    gq = g[perm].contiguous()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); eng.sync()
            t0 = time.perf_counter(); fn(); eng.sync()
            best = min(best, time.perf_counter() - t0)
        return best
    t_put = timed(lambda: eng._chk(lib.ocrgpu_guid_put(h, C.c_uint64(n), P(g), P(v), E.DEVICE)), reps=1)
    t_get = timed(lambda: eng._chk(lib.ocrgpu_guid_get(h, C.c_uint64(n), P(gq), P(out), E.DEVICE)))
    assert torch.equal(out, v[perm])
    t_rm = timed(lambda: eng._chk(lib.ocrgpu_guid_remove(h, C.c_uint64(n), P(gq), E.DEVICE)), reps=1)
    eng._chk(lib.ocrgpu_guid_get(h, C.c_uint64(n), P(gq), P(out), E.DEVICE)); eng.sync()
    assert not out.any()
    print(json.dumps({"n": n, "slots": 1 << lg_s, "load": n / (1 << lg_s), "put_per_s": n / t_put, "get_per_s": n / t_get,
                      "remove_per_s": n / t_rm, "get_GBps_at_32B_per_lookup": 32 * n / t_get / 1e9}))


if __name__ == "__main__":
    main()
