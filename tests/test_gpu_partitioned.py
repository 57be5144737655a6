"""Partition invariance: the same stream resolved by 1, 2, 3 and 4 engines (one per GUID
partition, all on this GPU, records exchanged per level) must give the bit-identical result record --
multi-GPU parity is checked against the single-PD oracle."""
import numpy as np
import pytest

import streams
from ocr_amd import distributed as D
from ocr_amd.opstream import layered_dag
from oracle import ref as O

pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


def run_partitioned(s, world):
    parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
    D.resolve_lockstep(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    return res, errs


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("name", sorted(n for n in ALL if n not in ("layered_100k", "powerlaw_100k")))
def test_semantic_streams_are_partition_invariant(name, world):
    s = ALL[name]
    res, errs = run_partitioned(s, world)
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name", ["layered_100k", "powerlaw_100k"])
def test_large_streams_partitioned(name, world):
    s = ALL[name]
    res, errs = run_partitioned(s, world)
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []
