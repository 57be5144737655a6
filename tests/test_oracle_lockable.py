"""Datablock acquire modes: the CPU restatement's Lockable mode is pinned on what the REFERENCE
runtime did with datablockfactory=Lockable (tests/golden/lockable.json, made by tests/golden/make_lockable_golden.py;
1 and 4 workers) and, where the reference build is present, on a live re-run.  The arbitration order is a scheduling
accident in the reference, so levels are compared through the invariants of tests/lockable.py (which the reference's
own levels must satisfy too) and everything else in the record bit for bit."""
import hashlib
import json
import os

import numpy as np
import pytest

import lockable as L
from oracle import ref as O

ALL = {s.name: s for s in L.all_lockable()}


def rest_digest(r):
    h = hashlib.sha256()
    for a in (r.val, r.evt_state, r.db_val):
        h.update(a.tobytes())
    return h.hexdigest()[:16]


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "lockable.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(ALL))
def test_restatement_against_reference_golden(name, golden):
    s, ent = ALL[name], golden[name]
    base = O.run_restatement(s)
    r, stats = O.run_restatement(s, return_stats=True, lockable=True)
    assert int(stats[8:].sum()) == 0
    L.check_levels(s, r.level, base.level)
    for w in ("workers1", "workers4"):
        g = ent[w]
        assert g["overlaps"] == 0, "the reference's own bodies saw holders the Lockable datablock should have kept apart"
        assert g["n_fired"] == r.n_fired and g["rest"] == rest_digest(r) == rest_digest(base)
        L.check_levels(s, np.asarray(g["level"], np.int32), base.level)   # the reference's outcome passes the same checker
    # the watcher has teeth: the same programs on Regular datablocks (no arbitration) do overlap with 4 workers
    assert ent["regular_workers4_overlaps"] > 0


def test_fixed_outcomes():
    """Where the outcome does not depend on who wins: n exclusive contenders ready together take n different levels,
    in the reference (1 and 4 workers) and in the restatement alike."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lockable.json")) as f:
        golden = json.load(f)
    for name, n in (("lock_ew_contenders", 6), ("lock_two_dbs", 6)):
        s = ALL[name]
        r = O.run_restatement(s, lockable=True)
        assert sorted(r.level[:n].tolist()) == list(range(n))
        for w in ("workers1", "workers4"):
            assert sorted(golden[name][w]["level"][:n]) == list(range(n))
    # readers share a level, the writer is alone on its own
    s = ALL["lock_readers_then_writer"]
    r = O.run_restatement(s, lockable=True)
    assert len(set(r.level[:4].tolist())) == 1 and r.level[4] not in set(r.level[:4].tolist()) | set(r.level[5:8].tolist())


def test_checker_rejects_bad_levels():
    s = ALL["lock_ew_contenders"]
    base = O.run_restatement(s)
    with pytest.raises(AssertionError):
        L.check_levels(s, base.level, base.level)       # all six exclusive holders on one level
    r = O.run_restatement(s, lockable=True)
    bad = r.level.copy(); bad[0] = -1
    with pytest.raises(AssertionError):
        L.check_levels(s, bad, base.level)


@pytest.mark.skipif(not O.reference_available(), reason="reference runtime build (oracle/_ref) not present")
@pytest.mark.parametrize("workers", [1, 4])
def test_live_reference_lockable(workers):
    for name in ("lock_ew_contenders", "lock_rw_share", "lock_pipeline", "lock_random_5", "lock_random_17"):
        s = ALL[name]
        base = O.run_restatement(s)
        ref, info = O.run_reference(s, workers=workers, db="Lockable")
        if workers > 1 and rest_digest(ref) != rest_digest(base):   # dump taken while another worker was still inside a releasing
            ref, info = O.run_reference(s, workers=workers, db="Lockable")   # call (oracle/ocr_replay.c settle_before_dump): one more run decides
        assert info["overlaps"] == 0 and ref.n_fired == base.n_fired
        assert rest_digest(ref) == rest_digest(base)
        L.check_levels(s, ref.level, base.level)


def _twins():
    from ocr_amd import opstream as op
    return [op.cholesky_dag(8), op.stencil_dag(16, 10), op.layered_dag(1000, 100)]


def test_config_twins_do_not_move_under_lockable():
    """The BASELINE config twins order every datablock hand-over through dependences already (tile datablocks travel in
    RW / RO modes along FWD0 chains): arbitration finds nothing to do and the record, levels included, is the Regular one."""
    for s in _twins():
        base = O.run_restatement(s)
        r, stats = O.run_restatement(s, return_stats=True, lockable=True)
        assert int(stats[8:].sum()) == 0 and base.diff(r) == [], s.name


@pytest.mark.skipif(not O.reference_available(), reason="reference runtime build (oracle/_ref) not present")
@pytest.mark.parametrize("workers", [1, 4])
def test_config_twins_live_reference_lockable(workers):
    for s in _twins()[:2]:
        ref, info = O.run_reference(s, workers=workers, db="Lockable")
        want = O.run_restatement(s, lockable=True)
        if workers > 1 and want.diff(ref) != []:   # see test_live_reference_lockable
            ref, info = O.run_reference(s, workers=workers, db="Lockable")
        assert info["overlaps"] == 0
        assert want.diff(ref) == [], s.name


def test_fuzz_corpus_does_not_move_under_lockable():
    """The random OCR programs of tests/streams.py name their datablocks in RO / RW / NULL modes and never make two
    incompatible holders ready together: Lockable leaves their records alone (checked live on the reference at 1 and 4
    workers for 40 of them while this was written: 0 overlaps, 0 levels moved).  A regression net for the acquisition
    walk next to every event kind, the lazy frontier and channels."""
    import streams
    for seed in range(streams.FUZZ_SEEDS):
        for es in (False, True):
            s = streams.random_stream(seed, 36, eager_safe=es)
            base = O.run_restatement(s)
            r, stats = O.run_restatement(s, return_stats=True, lockable=True)
            assert int(stats[8:].sum()) == 0 and base.diff(r) == [], (seed, es)
