"""The distributed driver end to end with TWO processes (one partition each) sharing this GPU: torch.distributed over
gloo carries the per-level record exchange (on an 8-GPU node the same code runs over RCCL).  Checks every EDT's level
and HASH value against the CPU restatement of the global stream."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n, width, general, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ocr_amd.distributed import distributed_exchange, resolve_distributed
    from ocr_amd.workloads import LayeredPartition

    def allreduce_sum(x):
        t = torch.tensor([int(x)], dtype=torch.int64)
        dist.all_reduce(t)
        return int(t.item())

    part = LayeredPartition(n, width, rank, world, allreduce_sum=allreduce_sum).build()
    if general:
        part.may_cascade = True                      # the settle-until-quiet protocol instead of one collective per level
    distributed_exchange(part, dist)
    part.eng.snapshot()
    out = None
    for _ in range(2):
        part.eng.restore()
        part.satisfy_start()
        levels = resolve_distributed(part, dist)
        lv, val = part.check()
        ids = np.r_[part.mine, n] if rank == 0 else part.mine
        out = (rank, ids, lv, val, levels, list(part.eng.stats().err))
    dist.barrier()
    dist.destroy_process_group()
    part.eng.close()
    q.put(out)


@pytest.mark.parametrize("world,general", [(2, False), (2, True), (3, False)])
def test_processes_sharing_one_gpu_gloo(world, general):
    from ocr_amd.opstream import layered_dag
    from oracle import ref as O
    width = 1200 if world == 3 else 1000
    n = 36_000 if world == 3 else 40_000
    ref = O.run_restatement(layered_dag(n, width))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, width, general, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ids, lv, val, levels, err in results:
        assert sum(err) == 0
        assert levels == n // width + 1
        assert np.array_equal(lv, ref.level[ids]) and np.array_equal(val, ref.val[ids])
