#!/usr/bin/env python3
"""Host cost of the per-level loop of ocrgpu_resolve_exchange on ONE GPU: (a) a one-rank RCCL communicator (the grouped
ncclSend / ncclRecv to self is the real transport call path), (b) two partitions in one process with device-to-device copies.
Each with the level's launches queued ahead (one synchronisation per level) and with OCRGPU_RX_POLLED=1 (two).

    OCRGPU_DIST_TIMING=1 python tools/dist_timing.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if not (sys.argv[1:] and sys.argv[1].startswith("windows")):   # (the per-stage timing keeps the loop host-driven: not for the windows runs)
    os.environ.setdefault("OCRGPU_DIST_TIMING", "1")

from ocr_amd import distributed as D          # noqa: E402
from ocr_amd import engine as E               # noqa: E402
from ocr_amd.workloads import LayeredPartition  # noqa: E402


def run(world, rccl, polled, n=1_000_000, width=10_000, windows=False):
    if polled:
        os.environ["OCRGPU_RX_POLLED"] = "1"
    else:
        os.environ.pop("OCRGPU_RX_POLLED", None)
    if rccl:   # one rank: the generic partition of the whole DAG (it configures the exchange), resolved once
        from ocr_amd.opstream import layered_dag
        part = D.Partition(layered_dag(n, width), 0, 1)
        part.build()
        if windows:
            D.connect_peer_windows([part])
        else:
            part.eng.comm_init(E.comm_get_unique_id(), 0, 1)
        part.eng.sync()
        t0 = time.perf_counter()
        levels = part.eng.resolve_exchange(no_cascade=True)
        dt = time.perf_counter() - t0
        print(f"world=1 transport={'peer windows' if windows else 'rccl(1 rank)'} {'polled (2 syncs/level)' if polled else 'queued ahead (1 sync/level)'}: "
              f"{levels} levels, {1e6 * dt / levels:.1f} us per level wall, {n} EDTs", flush=True)
        if not windows:
            part.eng.comm_fini()
        part.eng.close()
        return
    parts = [LayeredPartition(n, width, r, world, allreduce_sum=None) for r in range(world)]
    total_leaves = sum(len(p.leaves) for p in parts)
    for p in parts:
        p.n_leaves = total_leaves
        p.build()
    if world > 1:
        D._lockstep_exchange(parts)
    if windows:
        D.connect_peer_windows(parts)
    for p in parts:
        p.eng.snapshot()
    best = None
    for rep in range(3):
        for p in parts:
            p.eng.restore()
        parts[0].satisfy_start()
        for p in parts:
            p.eng.sync()
        t0 = time.perf_counter()
        levels = D.resolve_lockstep_c(parts, no_cascade=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"world={world} transport={'peer windows' if windows else 'rccl(1 rank)' if rccl else 'local copies'} {'polled (2 syncs/level)' if polled else 'queued ahead (1 sync/level)'}: "
          f"{levels} levels, {1e6 * best / levels:.1f} us per level wall, {n * world} EDTs", flush=True)
    for p in parts:
        p.eng.close()


if __name__ == "__main__":
    if sys.argv[1:] == ["rccl"]:       # the one case a kernel trace wants: python tools/dist_timing.py rccl
        run(1, True, False)
        sys.exit(0)
    if sys.argv[1:] == ["windows2"]:
        run(2, False, False, windows=True)
        sys.exit(0)
    if sys.argv[1:] == ["windows8"]:
        run(8, False, False, windows=True)
        sys.exit(0)
    if sys.argv[1:] == ["windows"]:
        run(1, True, False, windows=True)
        run(2, False, False, windows=True)
        run(4, False, False, windows=True)
        run(8, False, False, windows=True)
        sys.exit(0)
    for world, rccl in ((1, True), (2, False)):
        for polled in (True, False):
            run(world, rccl, polled)
    run(1, True, False, windows=True)
    run(2, False, False, windows=True)
    run(4, False, False, windows=True)
