"""N>1 plumbing on CPU: two gloo ranks exchange 32-byte satisfy records with exchange_records (the all-to-all-v that
stands where the reference's MPI comm-platform sends one ocrPolicyMsg_t per satisfy).  No GPU, no engine."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ocr_amd.distributed import REC_BYTES, exchange_records
    cap = 64
    send = torch.zeros((world, cap, REC_BYTES), dtype=torch.uint8)
    rng = np.random.default_rng(100 + rank)
    for rounds in range(3):
        counts = rng.integers(0, cap, size=world)
        if rounds == 2:
            counts[:] = 0                          # an empty round must not hang
        for d in range(world):                     # record k from rank r to d: bytes = (r, d, k, round, ...)
            for k in range(int(counts[d])):
                send[d, k, :4] = torch.tensor([rank, d, k, rounds], dtype=torch.uint8)
        recv, n = exchange_records(dist, send, counts, "cpu")
        all_counts = [None] * world
        dist.all_gather_object(all_counts, counts.tolist())
        want = sum(all_counts[src][rank] for src in range(world))
        assert n == want == recv.shape[0]
        off = 0
        for src in range(world):                   # arrives grouped by source rank, in order
            for k in range(all_counts[src][rank]):
                assert recv[off, :4].tolist() == [src, rank, k, rounds]
                off += 1
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, "ok"))


def test_exchange_records_gloo_world2():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == [0, 1]
