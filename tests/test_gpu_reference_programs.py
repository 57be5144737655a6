"""The reference's own non-regression programs (events, finish scopes, EDT depv handling, datablocks), compiled
UNMODIFIED against include/ocr_shim/ocr.h + libocrshim + libocrgpu by oracle/build_shimtests.py, run on the MI355X.
Expected exit code and console output come from running the same sources on the reference runtime
(tests/golden/reference_programs.json).  Console lines are compared as a multiset: the order in which independent
EDTs print is not defined by OCR."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_programs.json")))["programs"]
BIN = os.path.join(ROOT, "oracle", "_ref", "shimtests")
UNSUPPORTED = set()
RUNNABLE = sorted(n for n, e in FIX.items() if e.get("shim_builds") and n not in UNSUPPORTED)


def _check(name, env=None):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shimtests not built (python oracle/build_shimtests.py)")
    want = FIX[name]
    r = subprocess.run([exe] + list(want.get("args", [])), capture_output=True, text=True, timeout=60, env=env)
    if want["rc"] == 0:
        assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
        norm = lambda lines: sorted(re.sub(r"0x[0-9a-fA-F]+", "0xGUID", ln) for ln in lines)  # guid VALUES differ between runtimes
        got, exp = r.stdout.splitlines(), want["stdout"]
        if want.get("compare") == "verdict-lines":   # which EDT wins a labeled create is a race in the program itself
            keep = lambda lines: [ln for ln in lines if ln.startswith(("PASSED", "FAILURE"))]
            got, exp = keep(got), keep(exp)
            assert not any(ln.startswith("FAILURE") for ln in got), got
        assert norm(got) == norm(exp), (r.stdout[-800:], want["stdout"][-10:])
    else:  # programs the reference rejects (asserts): the shim must not report success either
        assert r.returncode != 0, (r.stdout[-500:], r.stderr[-500:])


@pytest.mark.parametrize("name", RUNNABLE)
def test_reference_program(name):
    _check(name)


def test_reference_programs_with_lockable_datablocks():
    """The same programs with OCRSHIM_LOCKABLE=1: the engine arbitrates the acquire modes the programs name
    (OCRGPU_F_LOCKABLE_DB) and libocrshim reports each body's end with ocrgpu_edt_release_dbs -- the reference's own
    exclusive-write programs (dist/testDistDbEw*.c, testDistDbEwRo*.c: EW writers, RO / CONST readers that check a
    datablock is never seen half written) among them.  Same exit codes and console lines as under Regular datablocks."""
    env = dict(os.environ, OCRSHIM_LOCKABLE="1")
    for name in RUNNABLE:
        _check(name, env)
    # the arbitration is really on: testDistDbEw2's 40 exclusive writers of one datablock take a level each (2 levels without)
    exe = os.path.join(BIN, "testDistDbEw2")
    levels = {}
    for lock in ("0", "1"):
        r = subprocess.run([exe], capture_output=True, text=True, timeout=60, env=dict(os.environ, OCRSHIM_LOCKABLE=lock, OCRSHIM_VERBOSE="1"))
        levels[lock] = int(re.search(r"levels=(\d+)", r.stderr).group(1))
    assert levels["0"] < 10 and levels["1"] >= 41, levels


def test_fib20():
    """BASELINE config 1 on the engine: same program text as oracle/_ref/ocr_fib_dbg (6765, 21891 + 10945 EDTs)."""
    import sys
    sys.path.insert(0, ROOT)
    from ocr_amd import build as B
    exe = B.build_program(os.path.join(ROOT, "tests", "programs", "fib.c"))   # no-op when ocr_amd/bin/fib is up to date
    r = subprocess.run([exe, "20"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.strip() == "FIB n=20 result=6765 fib_edts=21891 sum_edts=10945"


def test_fib27_in_an_engine_sized_for_the_live_set():
    """fib(27) creates 953 433 EDTs (635 621 fib + 317 810 sum + root + done), but at most ~half of them are alive at any time
    (every `sum` continuation waits for its subtree; a fib EDT is done the moment its body returns).  With OCRSHIM_MAX_EDT at
    that live maximum -- not at the total -- libocrshim takes finished EDTs' rows back between levels (ocrgpu_reclaim) the way
    the reference frees a task when taskExecute returns (hc-task.c:721); the datablocks the program destroys give their rows
    back too (ocrgpu_db_destroy), so the 635 621 of them fit 420 000 rows."""
    import sys
    sys.path.insert(0, ROOT)
    from ocr_amd import build as B
    exe = B.build_program(os.path.join(ROOT, "tests", "programs", "fib.c"))
    env = dict(os.environ, OCRSHIM_MAX_EDT="480000", OCRSHIM_MAX_EVT="700000", OCRSHIM_MAX_DB="420000", OCRSHIM_MAX_EDGES="2000000",
               OCRSHIM_VERBOSE="1")
    r = subprocess.run([exe, "27"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    assert r.stdout.strip() == "FIB n=27 result=196418 fib_edts=635621 sum_edts=317810"
    m = re.search(r"reclaims=(\d+) peak_live_edt<=(\d+) of max_edt=480000", r.stderr)
    assert m and int(m.group(1)) >= 1 and int(m.group(2)) <= 480000, r.stderr[-800:]
    assert "err=[0 0 0 0 0 0 0]" in r.stderr
    print(r.stderr[-600:])


def test_reference_programs_with_worker_threads():
    """OCRSHIM_WORKERS=4: the bodies of a level run on four threads (hc-worker.c:52-265), their deferred calls are merged when
    the level is done.  OCRSHIM_PAR_MIN=2 wakes the workers for every level of two or more bodies, so that these small programs
    really take the threaded path.  Same exit codes and console lines; the programs that wait inside an EDT
    (ocrLegacyBlockProgress under a running level) need one worker and say so."""
    # (with bodies really running side by side, exclusive access is the engine's job: the Lockable arbitration is on, as it
    #  is in the reference's stock configuration -- dist/testDistDbEw*.c check that no datablock is seen half written)
    env = dict(os.environ, OCRSHIM_WORKERS="4", OCRSHIM_PAR_MIN="2", OCRSHIM_LOCKABLE="1")
    needs_one = []
    for name in RUNNABLE:
        exe = os.path.join(BIN, name)
        if not os.path.exists(exe):
            pytest.skip("oracle/_ref/shimtests not built (python oracle/build_shimtests.py)")
        r = subprocess.run([exe] + list(FIX[name].get("args", [])), capture_output=True, text=True, timeout=60, env=env)
        if "needs OCRSHIM_WORKERS=1" in r.stderr:
            needs_one.append(name)
            continue
        _check(name, env)
    assert len(needs_one) <= 8 and all(n.startswith(("testLegacy", "rtapi", "testRt")) or "Legacy" in n for n in needs_one), needs_one


def test_fib27_with_worker_threads():
    import sys
    sys.path.insert(0, ROOT)
    from ocr_amd import build as B
    exe = B.build_program(os.path.join(ROOT, "tests", "programs", "fib.c"))
    out = {}
    for w in ("1", "4"):
        r = subprocess.run([exe, "27"], capture_output=True, text=True, timeout=600, env=dict(os.environ, OCRSHIM_WORKERS=w, OCRSHIM_VERBOSE="1"))
        assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
        assert r.stdout.strip() == "FIB n=27 result=196418 fib_edts=635621 sum_edts=317810"
        assert "err=[0 0 0 0 0 0 0]" in r.stderr and f"workers={w}" in r.stderr
        out[w] = float(re.search(r"seconds: bodies ([0-9.]+)", r.stderr).group(1))
    print("fib 27 host-body seconds by workers:", out)


def test_worker_threads_scale_on_bodies_that_do_work():
    """tests/programs/heavy.c: 8 levels of 4096 independent EDTs whose bodies run a 20 000-step integer recurrence (~16 us each).
    Same result whatever the worker count; four workers take well under the time of one (on the box: 0.53 s -> 0.13 s -> 0.04 s
    of body time with 1 / 4 / 16 workers; fib's 0.2-us bodies, by contrast, do not pay for a thread hand-over)."""
    import sys
    sys.path.insert(0, ROOT)
    from ocr_amd import build as B
    exe = B.build_program(os.path.join(ROOT, "tests", "programs", "heavy.c"))
    secs, lines = {}, set()
    for w in ("1", "4"):
        r = subprocess.run([exe, "4096", "8", "20000"], capture_output=True, text=True, timeout=300, env=dict(os.environ, OCRSHIM_WORKERS=w, OCRSHIM_VERBOSE="1"))
        assert r.returncode == 0, (r.stdout[-300:], r.stderr[-800:])
        lines.add(r.stdout.strip())
        assert "err=[0 0 0 0 0 0 0]" in r.stderr and "bodies=32777" in r.stderr
        secs[w] = float(re.search(r"seconds: bodies ([0-9.]+)", r.stderr).group(1))
    assert len(lines) == 1 and next(iter(lines)).startswith("HEAVY W=4096 L=8 iters=20000 sum="), lines
    assert secs["4"] < 0.7 * secs["1"], secs
