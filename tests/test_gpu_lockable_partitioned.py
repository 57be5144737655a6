"""Lockable datablocks (OCRGPU_F_LOCKABLE_DB = the cfg's datablockfactory=Lockable) ACROSS partitions: a datablock is arbitrated
where it lives, as under hc-dist (lockableAcquireInternal / lockableRelease, lockable-datablock.c:139-263, 330-507, run on the
owner's policy domain; PD_MSG_DB_ACQUIRE / DB_RELEASE are forwarded there, hc-dist-policy.c:1027-1031).  An EDT whose next
datablock (ascending guid: one order on every partition) is a copy of another partition's asks its owner with a record and goes
on when the grant arrives; the level that fires it sends the releases; the owner hands the datablock on.

Bar (tests/lockable.py, the one the single-partition engine and the reference itself are held to): everything but `level` is
bit-identical to the run without arbitration; every EDT fires, no earlier than its dependences allow; per datablock and level the
holders are compatible -- whichever partitions they run on."""
import numpy as np
import pytest

import lockable as L
from ocr_amd import distributed as D
from ocr_amd.engine import F_LOCKABLE_DB
from oracle import ref as O

pytestmark = pytest.mark.gpu
STREAMS = {s.name: s for s in L.all_lockable()}


def run(s, world, loop):
    parts = D.build_lockstep([D.Partition(s, r, world, flags=F_LOCKABLE_DB) for r in range(world)])
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
    assert all(sum(e) == 0 for e in errs), (s.name, errs)
    return res


@pytest.mark.parametrize("loop", ["lockstep", "c", "windows"])
@pytest.mark.parametrize("world", [2, 3, 4])
def test_lockable_streams_across_partitions(world, loop):
    for s in STREAMS.values():
        base = O.run_restatement(s)
        r = run(s, world, loop)
        assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all() and (r.db_val == base.db_val).all(), s.name
        L.check_levels(s, r.level, base.level)


@pytest.mark.parametrize("world", [2, 4])
def test_symmetric_contenders_take_one_level_each_across_partitions(world):
    for n in (2, 7, 65, 300):           # n EDTs, spread over the partitions, all want one datablock exclusively
        s = L.ew_contenders(n)
        r = run(s, world, "c")
        assert (np.sort(r.level[:n]) == np.arange(n)).all(), (n, np.sort(r.level[:n])[:10])
    s = L.ordered_two_dbs()             # two datablocks named in opposite orders: ascending guid order cannot deadlock
    r = run(s, world, "c")
    assert (np.sort(r.level[:6]) == np.arange(6)).all()


def test_larger_random_programs_across_partitions():
    for seed in (101, 102):
        s = L.random_lockable(seed, n=400)
        base = O.run_restatement(s)
        r = run(s, 4, "windows")
        assert (r.val == base.val).all() and (r.evt_state == base.evt_state).all() and (r.db_val == base.db_val).all()
        L.check_levels(s, r.level, base.level)
