"""The partition-local generator used by the multi-GPU bench builds the SAME DAG as opstream.layered_dag: resolve it
with 1, 2 and 4 engines on this GPU and compare every EDT's level and HASH value with the CPU restatement of the
global stream."""
import numpy as np
import pytest

from ocr_amd import distributed as D
from ocr_amd.opstream import layered_dag
from ocr_amd.workloads import LayeredPartition
from oracle import ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("n,width", [(2000, 100), (60_000, 1500)])
def test_partition_local_layered_matches_global_restatement(n, width, world):
    ref = O.run_restatement(layered_dag(n, width))
    counts = {}
    parts = [LayeredPartition(n, width, r, world, allreduce_sum=None) for r in range(world)]
    total_leaves = sum(len(p.leaves) for p in parts)
    for p in parts:
        p.n_leaves = total_leaves
        p.build()
    if world > 1:
        D._lockstep_exchange(parts)
    for p in parts:
        p.eng.snapshot()
    for rep in range(2):                           # second pass exercises restore in partitioned mode
        for p in parts:
            p.eng.restore()
        parts[0].satisfy_start()
        if world > 1:
            D.resolve_lockstep(parts)
        else:
            parts[0].eng.resolve()
        for p in parts:
            assert sum(p.eng.stats().err) == 0, list(p.eng.stats().err)
            lv, val = p.check()
            ids = np.r_[p.mine, n] if p.rank == 0 else p.mine
            assert np.array_equal(lv, ref.level[ids])
            assert np.array_equal(val, ref.val[ids])
    for p in parts:
        p.eng.close()
