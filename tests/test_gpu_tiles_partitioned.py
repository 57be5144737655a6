"""Tile datablocks (OCRGPU_F_WIDE_DB: the 8x8 int64 tiles SURVEY.md s8d.3 names for the Cholesky) across partitions.  The reference
clones the WHOLE datablock on a remote acquire and writes a released RW / EW copy back (ProxyDb_t, hc-dist-policy.c:382-700); here a
record that carries a tile is a head slot + 22 continuation slots in the same send region, the receiver makes a copy under the home
guid (or, when the datablock comes home, takes it in place), and a TILE body that updated a copy sends the contents home.

Bar: the goldens `cholesky_tiles_T8` / `cholesky_tiles_T32` (recorded from the reference runtime) bit-exact on 2 / 4 / 8 partitions
over every loop (host-driven lockstep, the C loop, the C loop over peer windows); T = 256 -- BASELINE config 3 at full size, 2.83M
tile updates -- on 8 partitions sharing this GPU against the CPU restatement, the checksums of all 65 536 tiles included."""
import json
import os

import numpy as np
import pytest

import streams
from ocr_amd import distributed as D
from ocr_amd import opstream as ops
from ocr_amd.opstream import Result
from oracle import ref as O

pytestmark = pytest.mark.gpu
ALL = {s.name: s for s in streams.all_streams()}


def run(s, world, loop, **kw):
    parts = D.build_lockstep([D.Partition(s, r, world, **kw) for r in range(world)])
    if loop == "lockstep":
        D.resolve_lockstep(parts)
    elif loop == "c":
        D.resolve_lockstep_c(parts)
    else:
        D.connect_peer_windows(parts)
        D.resolve_lockstep_c(parts)
    res = D.gather_result_lockstep(s, parts)
    errs = [list(p.eng.stats().err) for p in parts]
    for p in parts:
        p.eng.close()
    assert all(sum(e) == 0 for e in errs), errs
    return res


@pytest.mark.parametrize("loop", ["lockstep", "c", "windows"])
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", ["cholesky_tiles_T8", "cholesky_tiles_T32"])
def test_tile_goldens_are_partition_invariant(name, world, loop, golden_dir):
    s = ALL[name]
    res = run(s, world, loop)
    assert res.diff(O.run_restatement(s)) == []
    with open(os.path.join(golden_dir, "index.json")) as f:          # what the reference runtime left on this stream
        ent = json.load(f)[name]
    assert f"{res.digest():016x}" == ent["digest"] and res.n_fired == ent["n_fired"]
    if ent.get("files"):
        assert res.diff(Result.from_file(os.path.join(golden_dir, name + ".res"))) == []


def test_a_tile_record_that_does_not_fit_is_an_overflow_not_a_fault():
    s = ALL["cholesky_tiles_T8"]
    parts = [D.Partition(s, r, 2, send_cap=40) for r in range(2)]   # room for one tile record per region
    for p in parts:
        p.eng.tolerate_faults = True
    D.build_lockstep(parts)
    D.resolve_lockstep(parts)
    errs = np.array([list(p.eng.stats().err) for p in parts])
    for p in parts:
        p.eng.close()
    assert errs[:, 5].sum() > 0 and errs[:, [0, 1, 2, 3, 6]].sum() == 0       # counted as queue overflow; nothing else went wrong


def test_cholesky_T256_tiles_on_8_partitions():
    s = ops.cholesky_dag(256, tiles=True)
    want = O.run_restatement(s)
    res = run(s, 8, "windows", send_cap=1 << 21, proxy_cap=1_500_000)
    assert res.n_fired == s.n_edt
    assert res.diff(want) == []
