#!/usr/bin/env python3
"""Host-side sanitizer pass over libocrshim (ocr_amd/csrc/ocrshim.c): the reference's non-regression programs built with
the shim compiled under -fsanitize=address,undefined (host gcc; GPU code is not instrumented), run on the MI355X.

    python tools/shim_asan.py build     # in the container that has /root/reference -> oracle/_ref/shimtests_asan/
    python tools/shim_asan.py run       # on the GPU box: runs every program, prints any sanitizer report
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_ref", "shimtests_asan")
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]


def build():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, ROOT)
    import build_shimtests as B
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "ocrshim_asan.o")
    subprocess.check_call(["gcc", "-O1", "-std=gnu11", "-fPIC"] + SAN + ["-I", os.path.join(ROOT, "include", "ocr_shim"), "-I", os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "ocr_amd", "csrc", "ocrshim.c"), "-o", obj])
    n = 0
    for src in B.sources():
        name = os.path.splitext(os.path.basename(src))[0]
        cmd = ["gcc", "-O1", "-w", "-std=gnu11"] + SAN + [d for d in B.APP_DEFS if d != "-DOCR_ASSERT"] + \
              ["-I", os.path.join(ROOT, "include", "ocr_shim"), "-I", os.path.join(ROOT, "include"), src, obj,
               "-L", os.path.join(ROOT, "ocr_amd"), "-locrgpu", "-Wl,-rpath,$ORIGIN/../../../ocr_amd", "-Wl,-rpath-link,/opt/rocm/lib",
               "-o", os.path.join(OUT, name)]
        n += subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    print(f"{n} programs built under ASan+UBSan -> {OUT}")


def run():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_programs.json")))["programs"]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1")
    bad, ran = [], 0
    for name in sorted(fx):
        exe = os.path.join(OUT, name)
        if not os.path.exists(exe):
            continue
        try:
            r = subprocess.run([exe] + fx[name].get("args", []), capture_output=True, text=True, timeout=120, env=env)
        except subprocess.TimeoutExpired:
            bad.append((name, "timeout")); continue
        ran += 1
        rep = [l for l in r.stderr.splitlines() if "AddressSanitizer" in l or "runtime error" in l]
        if rep or r.returncode == 99:
            bad.append((name, rep[:3] or r.stderr[-300:]))
        if ran % 25 == 0:
            print(f"[shim_asan] {ran} programs run, {len(bad)} with reports", flush=True)
    print(f"[shim_asan] {ran} programs run, {len(bad)} with sanitizer reports")
    for b in bad:
        print("  ", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(build() if sys.argv[1:] == ["build"] else run())
