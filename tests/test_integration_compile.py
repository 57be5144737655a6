"""The reference-side bindings (integration/*.c: GUID provider, scheduler heuristic, workpile, comm-platform factories that fill the
reference's own function tables by forwarding to include/ocrgpu.h) are compiled here with gcc against the reference's headers, so
every signature is type-checked against the interface it claims to replace (ocr-guid.h:86-314, ocr-scheduler-heuristic.h:126-200,
ocr-workpile.h:39-84, ocr-comm-platform.h:55-199).  Object files only: nothing is linked, nothing of the reference is copied, and
the test is skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/ocr"
FILES = {"gpuhash-guid.c": ("ENABLE_GUID_GPUHASH", ["newGuidProviderFactoryGpuHash", "ocrgpuBinding"]),
         "gpu-scheduler-heuristic.c": ("ENABLE_SCHEDULER_HEURISTIC_GPU", ["newOcrSchedulerHeuristicFactoryGpu"]),
         "gpu-workpile.c": ("ENABLE_WORKPILE_GPU", ["newOcrWorkpileFactoryGpu"]),
         "rccl-comm-platform.c": ("ENABLE_COMM_PLATFORM_RCCL", ["newCommPlatformFactoryRccl"])}
# the flags the oracle's reference build uses (oracle/Makefile.ref), warnings on, the interesting ones as errors
FLAGS = ["-std=gnu99", "-O2", "-Wall", "-Werror=incompatible-pointer-types", "-Werror=implicit-function-declaration", "-Werror=int-conversion",
         "-Wno-unused-variable", "-Wno-unused-function", "-DCACHE_LINE_SZB=64", "-DGUID_PROVIDER_LOCID_SIZE=10",
         "-DOCR_DEBUG_LVL=DEBUG_LVL_WARN", "-DOCR_ASSERT", "-DOCR_DEBUG"]
INCS = [f"{REF}/build/x86", f"{REF}/inc", f"{REF}/src", f"{REF}/src/inc", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration")]


@pytest.mark.skipif(not os.path.isdir(REF + "/src/inc"), reason="reference tree not present")
@pytest.mark.parametrize("name", sorted(FILES))
def test_binding_compiles_against_the_reference_headers(name):
    define, symbols = FILES[name]
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, name + ".o")
        cmd = ["gcc", "-c"] + FLAGS + [f"-D{define}"] + [x for i in INCS for x in ("-I", i)] + [os.path.join(ROOT, "integration", name), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "warning" not in r.stderr, r.stderr[-3000:]
        nm = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True).stdout
        for sym in symbols:
            assert f" {sym}" in nm, f"{name} does not define {sym}"
        und = subprocess.run(["nm", "--undefined-only", obj], capture_output=True, text=True).stdout
        assert "ocrgpu_" in und, "the binding does not call the C-ABI"


@pytest.mark.skipif(not os.path.isdir(REF + "/src/inc"), reason="reference tree not present")
def test_every_c_abi_symbol_the_bindings_use_is_exported():
    """what the four bindings call is what libocrgpu.so exports (no GPU needed to look)"""
    lib = os.path.join(ROOT, "ocr_amd", "libocrgpu.so")
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    used = set()
    with tempfile.TemporaryDirectory() as tmp:
        for name, (define, _) in FILES.items():
            obj = os.path.join(tmp, name + ".o")
            subprocess.check_call(["gcc", "-c", "-w"] + FLAGS + [f"-D{define}"] + [x for i in INCS for x in ("-I", i)] +
                                  [os.path.join(ROOT, "integration", name), "-o", obj])
            for ln in subprocess.run(["nm", "--undefined-only", obj], capture_output=True, text=True).stdout.splitlines():
                if "ocrgpu_" in ln and "ocrgpuBinding" not in ln:
                    used.add(ln.split()[-1])
    assert len(used) >= 14, sorted(used)
    for sym in sorted(used):
        assert f" {sym}\n" in exported, f"{sym} is used by integration/ but not exported by libocrgpu.so"
