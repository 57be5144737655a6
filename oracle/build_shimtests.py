#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference's own non-regression programs as known-answer tests for libocrshim.

Each program under /root/reference/ocr/tests/non-regression-tests/{DIRS below} is compiled twice from
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
DIRS = ["events", "finish", "edt", "datablock", "guidlabel", "dist", "lifecycle", "rtapi", "legacy"]
ARGS = {"testGuidLabelFib": ["12"]}          # command-line arguments a program needs
# programs whose console lines depend on which of two concurrent EDTs wins a labeled create: only exit code and
# PASSED/FAILURE lines are compared
LOOSE_DIRS = {"guidlabel"}
OUT_REF = os.path.join(HERE, "_ref", "reftests")
OUT_SHIM = os.path.join(HERE, "_ref", "shimtests")
FIXTURE = os.path.join(ROOT, "tests", "golden", "reference_programs.json")
APP_DEFS = ["-DENABLE_EXTENSION_PARAMS_EVT", "-DENABLE_EXTENSION_COUNTED_EVT", "-DENABLE_EXTENSION_CHANNEL_EVT",
            "-DENABLE_EXTENSION_RTITF", "-DENABLE_EXTENSION_LEGACY", "-DENABLE_EXTENSION_LABELING", "-DENABLE_EXTENSION_AFFINITY", "-DOCR_ASSERT"]


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
           "-DENABLE_EXTENSION_LABELING", "-DENABLE_EXTENSION_AFFINITY", "-DENABLE_EXTENSION_CHANNEL_EVT", "-DENABLE_EXTENSION_RTITF", "-DENABLE_EXTENSION_LEGACY", "-I", os.path.join(ROOT, "include", "ocr_shim"), "-I", os.path.join(ROOT, "include"),
           src, os.path.join(ROOT, "ocr_amd", "libocrshim.a"), "-L", os.path.join(ROOT, "ocr_amd"), "-locrgpu",
           "-Wl,-rpath,$ORIGIN/../../../ocr_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out]
    return subprocess.run(cmd, capture_output=True, text=True).returncode == 0


# ---- the reference's own end-to-end micro-benchmarks (performance-tests/ocr/edtExecute*Sync.c: "TIME: completion of
#      all tasks"), built against both runtimes; tests/test_gpu_reference_perf_programs.py runs them side by side ----
PERF_DIR = os.path.join(REF, "tests", "performance-tests")
PERF = {  # program -> -D overrides of performance-tests/defaults.mk
    "edtExecuteHierFinishSync": {"TREE_DEPTH": 16, "NODE_FANOUT": 2, "LEAF_FANOUT": 10},
    "edtExecuteFlatFinishSync": {"NB_INSTANCES": 200000},
    "edtExecuteLatchSync": {"NB_INSTANCES": 200000},
    "edtExecuteStickySync": {"NB_INSTANCES": 200000},
}
PERF_DEFAULTS = {"DB_NBS": 32, "NB_EVT_COUNTED_DEPS": 4, "NB_ITERS": 1000, "NB_INSTANCES": 1000, "DEPV_SZ": 10, "PARAMC_SZ": 2,
                 "FAN_OUT": 100, "DB_SZ": 4, "NODE_FANOUT": 2, "LEAF_FANOUT": 10, "TREE_DEPTH": 8, "DB_NB_ELT": 10, "DB_TYPE": "u64",
                 "NB_WORKERS": 1, "NB_NODES": 1}
OUT_PERF = os.path.join(HERE, "_ref", "perfprogs")


def build_perf():
    os.makedirs(OUT_PERF, exist_ok=True)
    helper = os.path.join(PERF_DIR, "platform", "x86.c")        # get_time / summary_throughput_timer (gettimeofday)
    built = []
    for name, over in PERF.items():
        src = os.path.join(PERF_DIR, "ocr", name + ".c")
        defs = [f"-D{k}={v}" for k, v in dict(PERF_DEFAULTS, **over).items()]
        inc = ["-I", os.path.join(PERF_DIR, "inc")]
        ref = ["gcc", "-O2", "-w", "-std=gnu11"] + APP_DEFS[:-1] + defs + inc + ["-I", os.path.join(REF, "inc"),
               "-I", os.path.join(REF, "build", "x86"), src, helper, os.path.join(HERE, "_ref", "libocr_x86_perf.a"), "-lpthread", "-lm",
               "-o", os.path.join(OUT_PERF, name + "_ref")]
        shim = ["gcc", "-O2", "-w", "-std=gnu11", "-DENABLE_EXTENSION_PARAMS_EVT", "-DENABLE_EXTENSION_COUNTED_EVT",
                "-DENABLE_EXTENSION_AFFINITY"] + defs + inc + ["-I", os.path.join(ROOT, "include", "ocr_shim"),
                "-I", os.path.join(ROOT, "include"), src, helper, os.path.join(ROOT, "ocr_amd", "libocrshim.a"),
                "-L", os.path.join(ROOT, "ocr_amd"), "-locrgpu", "-lm", "-Wl,-rpath,$ORIGIN/../../../ocr_amd",
                "-Wl,-rpath-link,/opt/rocm/lib", "-o", os.path.join(OUT_PERF, name + "_shim")]
        ok = [subprocess.run(c, capture_output=True, text=True).returncode == 0 for c in (ref, shim)]
        built.append((name, ok))
    from oracle import mkcfg
    for w in (1, 8):
        open(os.path.join(OUT_PERF, f"hc-{w}w.cfg"), "w").write(mkcfg.cfg(w, guid="PTR"))
    return built


OUT_MI355X = os.path.join(HERE, "_ref", "mi355xtests")


def build_mi355x():
    """The same programs against the reference runtime BOOTED ON THE GPU FACTORIES (oracle/Makefile.mi355x ->
    oracle/_ref/libocr_x86-mi355x.a: the reference's sources + integration/*.c, cfg names GPUHASH / GPU / GPU), into
    oracle/_ref/mi355xtests/ with the machine configs beside them; tests/test_gpu_reference_boot.py runs them on the MI355X and
    expects what tests/golden/reference_programs.json recorded from the stock runtime."""
    from oracle import mkcfg
    os.makedirs(OUT_MI355X, exist_ok=True)
    for w in (1, 4):
        open(os.path.join(OUT_MI355X, f"gpu-{w}w.cfg"), "w").write(mkcfg.cfg(w, guid="GPUHASH", heuristic="GPU", workpile="GPU"))
    subprocess.check_call(["make", "-s", "-j8", "-f", "Makefile.mi355x"], cwd=HERE)
    libdir = os.path.join(HERE, "_ref", "mi355x")     # the runtime as one shared object: the programs stay a few KB each
    n = 0
    for src in sources():
        name = os.path.splitext(os.path.basename(src))[0]
        cmd = ["gcc", "-O1", "-w", "-std=gnu11"] + APP_DEFS + ["-I", os.path.join(REF, "inc"), "-I", os.path.join(REF, "build", "x86"),
               src, "-L", libdir, "-locr_x86-mi355x", "-Wl,-rpath,$ORIGIN/../mi355x", "-Wl,-rpath-link," + os.path.join(ROOT, "ocr_amd"),
               "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-lm", "-s", "-o", os.path.join(OUT_MI355X, name)]
        n += subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    print(f"{n} of {len(sources())} programs built against libocr_x86-mi355x.so -> {OUT_MI355X}")
    return n


def main():
    sys.path.insert(0, ROOT)
    if "--mi355x" in sys.argv:
        from ocr_amd import build as B
        B.build_lib()
        return build_mi355x()
    from ocr_amd import build as B
    B.build_lib()
    B.build_shim()
    os.makedirs(OUT_REF, exist_ok=True)
    os.makedirs(OUT_SHIM, exist_ok=True)
    shim_only = "--shim" in sys.argv
    for name, ok in build_perf():
        print(f"perf program {name:32s} ref/shim built: {ok}")
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
