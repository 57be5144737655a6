"""What the compiler makes of k_step, checked at build level (no GPU): registers, scratch and occupancy of the instantiations
the measurements in DESIGN.md were taken on.  The hot kernel is latency-bound and lives on its 4 waves per SIMD; code inlined
into it from a rare path (it happened: a wavefront-aggregated reservation for cross-partition records cost the tile-body
instantiation 70 VGPRs, 312 bytes of scratch per lane and half its speed) shows up here before it shows up in a benchmark."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_k_step_instantiations_keep_their_registers(tmp_path):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "ocr_amd", "csrc", "ocrgpu.hip"), "-o", str(tmp_path / "lib.so"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    usage = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    steps = {k: v for k, v in usage.items() if k.startswith("_Z6k_stepILb")}
    assert len(steps) == 10, sorted(usage)[:10]
    for k, v in steps.items():
        tiles = "ELi1EE" in k                 # OCRGPU_F_WIDE_DB on one partition: the 8x8 tile bodies are compiled in
        xlock = "ELi2ELi0EE" in k             # OCRGPU_F_LOCKABLE_DB with other partitions: datablocks arbitrated by their owners
        if xlock:                             # a path of its own: no scratch (a kernel with scratch paid 22 us per launch for nothing), 3 waves per SIMD
            assert v["ScratchSize"] == 0 and v["VGPRs"] <= 168 and v["Occupancy"] >= 3, (k, v)
            continue
        # ("ELi2EE": tile datablocks with other partitions -- its bodies run in k_bodies_pre, so it is held to the default kernel's
        # numbers; with the tile code inline in front of the fan-out it had 249 VGPRs and 312 bytes of scratch, 61 us per launch)
        assert v["ScratchSize"] == 0 and v["VGPRs Spill"] == 0, (k, v)
        if tiles:
            assert v["VGPRs"] <= 168 and v["Occupancy"] >= 3, (k, v)
        else:
            assert v["VGPRs"] <= 128 and v["Occupancy"] >= 4, (k, v)
    pre = usage["_Z12k_bodies_pre3Devj"]
    assert pre["ScratchSize"] == 0 and pre["VGPRs"] <= 128, pre
    print({k: (v["VGPRs"], v["ScratchSize"], v["Occupancy"]) for k, v in steps.items()})
