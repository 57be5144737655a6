"""The checker is checked: oracle/engine_ref.c under AddressSanitizer + UndefinedBehaviorSanitizer (host gcc), over the
hand-written semantic streams, fuzz programs, the Lockable-datablock programs and a contended layered DAG, in both
datablock modes.  Any report fails the test."""
import os
import shutil
import subprocess
import tempfile

import pytest

import lockable as L
import streams
from ocr_amd import opstream as op

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_restatement_is_sanitizer_clean():
    with tempfile.TemporaryDirectory(prefix="ocrsan_") as td:
        exe = os.path.join(td, "san_driver")
        cc = subprocess.run(["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"),
                             os.path.join(ROOT, "oracle", "san_driver.c"), os.path.join(ROOT, "oracle", "engine_ref.c"), "-o", exe],
                            capture_output=True, text=True)
        if cc.returncode != 0 and "asan" in cc.stderr.lower():
            pytest.skip("gcc has no sanitizer runtime here")
        assert cc.returncode == 0, cc.stderr[-2000:]
        ss = L.all_lockable() + [streams.random_stream(i, 48, eager_safe=bool(i & 1)) for i in range(32)] + streams.all_streams()[:16] \
            + [op.lockable_layered(20000, 500, 64)]
        files = []
        for i, s in enumerate(ss):
            files.append(os.path.join(td, f"s{i}.ops"))
            s.to_file(files[-1])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        run = subprocess.run([exe] + files, capture_output=True, text=True, env=env, timeout=600)
        assert run.returncode == 0 and run.stderr.strip() == "", run.stderr[-3000:]
        assert len(run.stdout.strip().splitlines()) == 2 * len(files)
