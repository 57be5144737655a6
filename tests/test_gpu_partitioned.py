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
@pytest.mark.parametrize("name", sorted(n for n in ALL if n not in ("layered_100k", "powerlaw_100k") and not ALL[n].db_words))   # (tile datablocks: tests/test_gpu_tiles_partitioned.py)
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


@pytest.mark.parametrize("G,steps,world", [(16, 10, 8), (16, 10, 4), (256, 4, 8)])
def test_stencil_block_partition_only_halo_edges_cross(G, steps, world):
    """BASELINE config 4: the stencil is block-partitioned (2 x 4 blocks of tiles over 8 partitions), so that
    only the dependences along block borders cross; bit-exact against the single-partition record, and the records on the
    wire are counted: per time step exactly the halo edges, nothing else."""
    from ocr_amd.opstream import stencil_dag
    s = ALL["stencil_16x16x10"] if (G, steps) == (16, 10) else stencil_dag(G, steps)
    own = D.stencil_block_owner(G, steps, world)
    py = {8: 2, 4: 2}[world]; px = world // py
    assert sorted(np.bincount(own[:-1], minlength=world).tolist()) == [G * G * steps // world] * world   # equal blocks
    parts = D.build_lockstep([D.Partition(s, r, world, owner_edt=own) for r in range(world)])
    sent = []
    D.resolve_lockstep(parts, count_records=sent)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    assert res.diff(O.run_restatement(s)) == []
    # halo edges of one time step: every tile on a block border sends its datablock across, in both directions
    halo = 2 * ((py - 1) * G + (px - 1) * G)
    body = [n for n in sent if n]                      # the level that fires the sink sends nothing
    # levels 0 .. steps-2 feed the next time step; the last step's output events feed the sink latch on rank 0
    assert body[:steps - 1] == [halo] * (steps - 1), (body, halo)
    latch_rank = int(D.default_owner(np.arange(s.n_evt) + (1 << 40), world)[1])      # explicit events are hash-placed; event 1 = the sink latch
    leaves_elsewhere = int((own[(steps - 1) * G * G:-1] != latch_rank).sum())
    assert body[steps - 1] == leaves_elsewhere
