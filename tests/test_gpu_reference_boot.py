"""Row b' of the scope table: UNMODIFIED OCR programs on the REFERENCE runtime booted on this repo's GPU factories, selected by
name from the machine config the way the reference selects any component -- [GuidType0] name=GPUHASH, scheduler heuristic GPU,
workpile GPU (ocr/src/driver/ocr-driver.c:220-287, ocr-machine-description.c:518-530).  oracle/Makefile.mi355x compiles the
reference's own sources where they lie (plain gcc, nothing copied, nothing modified) together with integration/*.c into
oracle/_ref/mi355x/libocr_x86-mi355x.so, which calls libocrgpu.so through include/ocrgpu.h; oracle/build_shimtests.py --mi355x
links the programs.  Test infrastructure only: nothing under ocr_amd/ links or imports any of it.

What is checked: fib(20) = 6765 with the reference's EDT counts (BASELINE config 1), and every program of the reference's
non-regression directories leaves the exit code and the console lines the STOCK reference runtime left
(tests/golden/reference_programs.json) -- with one HC worker and with four.  OCRGPU_BINDING_STATS makes the bindings report at
tear-down how many calls reached the engine, so a run that silently took another provider cannot pass."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_programs.json")))["programs"]
BIN = os.path.join(ROOT, "oracle", "_ref", "mi355xtests")
FIB = os.path.join(ROOT, "oracle", "_ref", "mi355x", "fib")
PROGRAMS = sorted(n for n, e in FIX.items() if e["rc"] not in ("timeout", "no-ref-build"))
STATS = re.compile(r"GPUHASH binding: (\d+) guids made, (\d+) lookups, (\d+) released; GPU work-list: (\d+) pushed, (\d+) popped")


def _need(path):
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/mi355x* not built (python oracle/build_shimtests.py --mi355x, where /root/reference is present)")


def _run(exe, cfg, args, timeout=120):
    env = dict(os.environ, OCRGPU_BINDING_STATS="1")
    return subprocess.run([exe, "-ocr:cfg", os.path.join(BIN, cfg)] + list(args), capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("workers", [1, 4])
def test_fib20_on_the_reference_runtime_with_gpu_factories(workers):
    _need(FIB)
    r = _run(FIB, f"gpu-{workers}w.cfg", ["20"])
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-500:])
    assert "FIB n=20 result=6765 fib_edts=21891 sum_edts=10945" in r.stdout
    m = STATS.search(r.stderr)
    assert m, r.stderr[-500:]
    guids, lookups, released, pushed, popped = map(int, m.groups())
    # 32 836 fib + sum EDTs (+ root, done, main) went through the GPU work-list; every EDT, event, datablock and template has a guid
    assert pushed == popped and pushed >= 21891 + 10945
    assert guids >= 3 * (21891 + 10945) // 2 and lookups > guids and released > 0


def _check(name, cfg):
    exe = os.path.join(BIN, name)
    _need(exe)
    want = FIX[name]
    r = _run(exe, cfg, want.get("args", []), timeout=60)
    if want["rc"] == 0:
        assert r.returncode == 0, (name, r.returncode, r.stdout[-500:], r.stderr[-500:])
        norm = lambda lines: sorted(re.sub(r"0x[0-9a-fA-F]+", "0xGUID", ln) for ln in lines)  # guid VALUES differ between providers
        got, exp = r.stdout.splitlines(), want["stdout"]
        if want.get("compare") == "verdict-lines":   # which EDT wins a labeled create is a race in the program itself
            keep = lambda lines: [ln for ln in lines if ln.startswith(("PASSED", "FAILURE"))]
            got, exp = keep(got), keep(exp)
            assert not any(ln.startswith("FAILURE") for ln in got), (name, got)
        assert norm(got) == norm(exp), (name, r.stdout[-800:], want["stdout"][-10:])
        m = STATS.search(r.stderr)
        assert m and int(m.group(1)) > 0 and int(m.group(4)) == int(m.group(5)), (name, r.stderr[-300:])   # guids came from the engine; what its
        #                                                            work-list took it handed out again (legacy-mode programs schedule nothing)
    else:  # programs the reference rejects (asserts, ocrAbort): the same exit code
        assert r.returncode == want["rc"], (name, r.returncode, want["rc"], r.stderr[-300:])


@pytest.mark.parametrize("name", PROGRAMS)
def test_reference_program_on_gpu_factories(name):
    _check(name, "gpu-1w.cfg")


def test_reference_programs_on_gpu_factories_with_four_workers():
    """the bindings serialise the workers on one lock (integration/ocrgpu-binding.h): same results with four HC workers"""
    for name in PROGRAMS:
        _check(name, "gpu-4w.cfg")
