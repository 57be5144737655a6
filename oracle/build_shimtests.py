#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference's own non-regression programs as known-answer tests for libocrshim.

Each program under /root/reference/ocr/tests/non-regression-tests/{events,finish,edt,datablock} is compiled twice from
the source where it lies (nothing is copied): against the reference runtime (oracle/_ref/libocr_x86_dbg.a) into
oracle/_ref/reftests/, and against include/ocr_shim/ocr.h + libocrshim.a + libocrgpu.so into oracle/_ref/shimtests/.
The reference build is run HERE (1 HC worker) and its exit code and console output are written to
tests/golden/reference_programs.json; tests/test_gpu_reference_programs.py runs the shim builds on the MI355X and
expects the same exit code and the same console lines.

    python oracle/build_shimtests.py            # build both, regenerate the fixture
    python oracle/build_shimtests.py --shim     # rebuild the shim side only (after libocrshim / libocrgpu changed)
"""
import glob
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/ocr"
TESTS = os.path.join(REF, "tests", "non-regression-tests")
DIRS = ["events", "finish", "edt", "datablock", "guidlabel", "dist", "lifecycle"]
ARGS = {"testGuidLabelFib": ["12"]}          # command-line arguments a program needs
# programs whose console lines depend on which of two concurrent EDTs wins a labeled create: only exit code and
# PASSED/FAILURE lines are compared
LOOSE_DIRS = {"guidlabel"}
OUT_REF = os.path.join(HERE, "_ref", "reftests")
OUT_SHIM = os.path.join(HERE, "_ref", "shimtests")
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_programs.json")
APP_DEFS = ["-DENABLE_EXTENSION_PARAMS_EVT", "-DENABLE_EXTENSION_COUNTED_EVT", "-DENABLE_EXTENSION_CHANNEL_EVT",
            "-DENABLE_EXTENSION_RTITF", "-DENABLE_EXTENSION_LABELING", "-DENABLE_EXTENSION_AFFINITY", "-DOCR_ASSERT"]


def args_of(src, name):
    """command-line arguments: ARGS, or the <name>.args file the reference keeps next to the program"""
    if name in ARGS:
        return ARGS[name]
    f = os.path.splitext(src)[0] + ".args"
    return open(f).read().split() if os.path.exists(f) else []


def sources():
    out = []
    for d in DIRS:
        out += sorted(glob.glob(os.path.join(TESTS, d, "*.c")))
    return out


def build_ref(src, out):
    cmd = ["gcc", "-O1", "-w", "-std=gnu11"] + APP_DEFS + ["-I", os.path.join(REF, "inc"), "-I", os.path.join(REF, "build", "x86"),
           src, os.path.join(HERE, "_ref", "libocr_x86_dbg.a"), "-lpthread", "-lm", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True).returncode == 0


def build_shim(src, out):
    cmd = ["gcc", "-O1", "-w", "-std=gnu11", "-DENABLE_EXTENSION_PARAMS_EVT", "-DENABLE_EXTENSION_COUNTED_EVT",
           "-DENABLE_EXTENSION_LABELING", "-DENABLE_EXTENSION_AFFINITY", "-DENABLE_EXTENSION_CHANNEL_EVT", "-I", os.path.join(ROOT, "include", "ocr_shim"), "-I", os.path.join(ROOT, "include"),
           src, os.path.join(ROOT, "ocr_amd", "libocrshim.a"), "-L", os.path.join(ROOT, "ocr_amd"), "-locrgpu",
           "-Wl,-rpath,$ORIGIN/../../../ocr_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True).returncode == 0


def main():
    sys.path.insert(0, ROOT)
    from ocr_amd import build as B
    B.build_lib()
    B.build_shim()
    os.makedirs(OUT_REF, exist_ok=True)
    os.makedirs(OUT_SHIM, exist_ok=True)
    shim_only = "--shim" in sys.argv
    old = json.load(open(FIXTURE)) if os.path.exists(FIXTURE) else {"programs": {}}
    fixture = {"generated_by": "oracle/build_shimtests.py: reference runtime (oracle/_ref/libocr_x86_dbg.a), 1 HC worker",
               "programs": {}}
    if not shim_only:
        from oracle import mkcfg
        cfg = os.path.join(OUT_REF, "hc-1w.cfg")
        open(cfg, "w").write(mkcfg.cfg(1))
        cfg_lab = os.path.join(OUT_REF, "hc-1w-labeled.cfg")     # guidReserve needs the LABELED provider (counted-map-guid.c:207)
        open(cfg_lab, "w").write(mkcfg.cfg(1, guid="LABELED"))
    n_shim = 0
    for src in sources():
        name = os.path.splitext(os.path.basename(src))[0]
        rel = os.path.relpath(src, REF)
        has_shim = build_shim(src, os.path.join(OUT_SHIM, name))
        n_shim += has_shim
        if shim_only:
            if name in old["programs"]:
                fixture["programs"][name] = dict(old["programs"][name], shim_builds=has_shim)
            continue
        ent = {"source": rel, "shim_builds": has_shim, "args": args_of(src, name),
               "compare": "verdict-lines" if os.path.basename(os.path.dirname(src)) in LOOSE_DIRS else "all-lines"}
        if build_ref(src, os.path.join(OUT_REF, name)):
            try:
                use = cfg_lab if os.path.basename(os.path.dirname(src)) == "guidlabel" else cfg
                r = subprocess.run([os.path.join(OUT_REF, name), "-ocr:cfg", use] + args_of(src, name), capture_output=True, text=True, timeout=30)
                ent["rc"], ent["stdout"] = r.returncode, r.stdout.splitlines()
            except subprocess.TimeoutExpired:
                ent["rc"], ent["stdout"] = "timeout", []
        else:
            ent["rc"] = "no-ref-build"
        fixture["programs"][name] = ent
        print(f"{name:45s} ref rc={ent['rc']!s:12s} shim_builds={has_shim}")
    json.dump(fixture, open(FIXTURE, "w"), indent=1, sort_keys=True)
    print(f"{len(fixture['programs'])} programs, {n_shim} build against the shim -> {FIXTURE}")


if __name__ == "__main__":
    main()
