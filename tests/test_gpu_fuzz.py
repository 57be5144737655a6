"""Fuzz parity on the GPU: random well-formed OCR programs (streams.random_stream, both flavours) resolved by the
HIP engine must give the result record of the CPU restatement, and the digest the reference runtime produced for
the committed seeds (tests/golden/fuzz.json); also split over 2 and 3 GUID partitions."""
import json
import os

import pytest

import streams
from ocr_amd import distributed as D
from ocr_amd import engine
from oracle import ref as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("safe", [True, False])
@pytest.mark.parametrize("nanny", [True, False])
def test_engine_matches_reference_digests_and_restatement(nanny, safe, golden_dir):
    with open(os.path.join(golden_dir, "fuzz.json")) as f:
        fuzz = json.load(f)
    for seed in range(200):
        s = streams.random_stream(seed, n=36 if seed < streams.FUZZ_SEEDS else 64, eager_safe=safe)
        r, st = engine.run_stream(s, flags=engine.F_NANNY if nanny else 0)
        assert sum(st.err) == 0, f"{s.name}: device flagged errors {list(st.err)}"
        assert r.diff(O.run_restatement(s)) == [], s.name
        if seed < streams.FUZZ_SEEDS:
            assert f"{r.digest():016x}" == fuzz[s.name]["digest"], s.name


@pytest.mark.parametrize("world", [2, 3])
def test_fuzz_programs_are_partition_invariant(world):
    for seed in range(60):
        s = streams.random_stream(seed, n=48, eager_safe=bool(seed & 1))
        parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
        D.resolve_lockstep(parts)
        res = D.gather_result_lockstep(s, parts)
        errs = [list(p.eng.stats().err) for p in parts]
        for p in parts:
            p.eng.close()
        assert all(sum(e) == 0 for e in errs), (s.name, errs)
        assert res.diff(O.run_restatement(s)) == [], s.name


def test_latch_ops_of_one_body_keep_their_order():
    """INCR, DECR, DECR by one EDT on a latch holding 1 ends at 0 and fires once (satisfyEventHcLatch,
    hc-event.c:812-868, applied call by call in the reference); the engine folds the run into one net op.  Many such
    EDTs in one level, so the ops would otherwise land in different waves."""
    from ocr_amd.opstream import BODY_NOP, EVT_LATCH, EVT_ONCE, LATCH_DECR, LATCH_INCR, MODE_RO, Builder
    b = Builder("latch_order")
    start = b.evt(EVT_ONCE)
    expect = []
    for i in range(300):
        lt = b.evt(EVT_LATCH, 1 + (i % 3 == 2))
        drv = b.edt(1, BODY_NOP)
        b.add_dep(start, drv, 0, MODE_RO)
        for _ in range(1 + i % 4):
            b.script_satisfy(drv, lt, 0, LATCH_INCR)
        for _ in range(2 + i % 4):
            b.script_satisfy(drv, lt, 0, LATCH_DECR)
        w = b.edt(1, BODY_NOP)
        b.add_dep(lt, w, 0, MODE_RO)
        expect.append(drv)
        if i % 3 != 2:
            expect.append(w)         # the latches that started at 2 are left holding 1
    s = b.finish(expect, start)
    ref = O.run_restatement(s)
    for flags in (engine.F_NANNY, 0):
        r, st = engine.run_stream(s, flags=flags)
        assert sum(st.err) == 0, list(st.err)
        assert r.diff(ref) == []


def test_latch_delta_through_the_api():
    """OCRGPU_LATCH_DELTA(k) on ocrgpu_event_satisfy_slot: one op for the net of several."""
    import numpy as np
    from ocr_amd.opstream import BODY_NOP, MODE_RO
    eng = engine.Engine(max_edt=16, max_evt=16, max_db=4)
    lt = eng.event_create([4, 4, 4], [5, 0, 3])          # OCR_EVENT_LATCH_T, initial counters 5, 0, 3
    edts, _, _ = eng.edt_create([1, 1, 1], [BODY_NOP] * 3, [1, 2, 3])
    eng.add_dependence(lt, edts, [0, 0, 0], [MODE_RO] * 3)
    eng.event_satisfy_slot(lt, None, [engine.LATCH_DELTA(-5), engine.LATCH_DELTA(0), engine.LATCH_DELTA(-2)])
    st = eng.resolve()
    assert sum(st.err) == 0 and st.fired == 2
    lv, _ = eng.read_edts(0, 3)
    assert list(lv >= 0) == [True, True, False]
    eng.close()


def test_lazy_frontier_counted_event():
    """COUNTED event on pre-slot 1 of an EDT whose pre-slot 0 is never satisfied: the registration is never made, the
    event keeps its count (hc-task.c:1286-1338); a second EDT whose frontier does get there uses it up."""
    from ocr_amd.opstream import BODY_NOP, EVT_COUNTED, EVT_ONCE, EVS_DESTROYED, EVS_LIVE_SAT, MODE_RO, Builder
    b = Builder("lazy_frontier_gpu")
    start = b.evt(EVT_ONCE)
    never = b.evt(EVT_ONCE)
    kept, used = b.evt(EVT_COUNTED, 1), b.evt(EVT_COUNTED, 1)
    stuck = b.edt(2, BODY_NOP)
    b.add_dep(never, stuck, 0, MODE_RO)
    b.add_dep(kept, stuck, 1, MODE_RO)
    runs = b.edt(2, BODY_NOP)
    b.add_dep(used, runs, 1, MODE_RO)          # registered before pre-slot 0 has its dependence: held back
    b.add_dep(start, runs, 0, MODE_RO)
    b.satisfy(kept, 0)
    b.satisfy(used, 0)
    s = b.finish([runs], start)
    ref = O.run_restatement(s)
    assert ref.evt_state[2] & 0xFF == EVS_LIVE_SAT and ref.evt_state[3] & 0xFF == EVS_DESTROYED
    for flags in (engine.F_NANNY, 0):
        r, st = engine.run_stream(s, flags=flags)
        assert sum(st.err) == 0 and r.diff(ref) == []


def test_channel_to_event_and_event_to_channel():
    """CHANNEL events as source AND destination of event dependences (a STICKY feeding a channel; a channel feeding a
    ONCE event, a latch, an EDT pre-slot and another channel).  Pinned since round 2: the replay harness clocks a channel's
    generations whatever satisfies them and whoever waits for them (oracle/ocr_replay.c: chan_gen_clk), and
    tests/golden/channel_event_deps.res is what the reference runtime did (1 and 4 workers)."""
    import os
    import streams
    from ocr_amd.opstream import Result
    s = streams.channel_event_deps()
    gold = Result.from_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "channel_event_deps.res"))
    ref = O.run_restatement(s)
    assert list(ref.level[:5]) == [1, 2, 3, 4, 0] and ref.diff(gold) == []
    for flags in (engine.F_NANNY, 0):
        r, stt = engine.run_stream(s, flags=flags)
        assert sum(stt.err) == 0 and r.diff(ref) == [] and r.diff(gold) == []
    for world in (2, 3):
        parts = D.build_lockstep([D.Partition(s, r, world) for r in range(world)])
        D.resolve_lockstep(parts)
        res = D.gather_result_lockstep(s, parts)
        errs = [sum(p.eng.stats().err) for p in parts]
        for p in parts:
            p.eng.close()
        assert sum(errs) == 0 and res.diff(ref) == []
