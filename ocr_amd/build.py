"""Build libocrgpu.so (hipcc, gfx950 only) in-tree. No CPU fallback exists: without the library the engine raises."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libocrgpu.so")
SRC = [os.path.join(HERE, "csrc", "ocrgpu.hip")]
HDRS = [os.path.join(ROOT, "include", "ocrgpu.h"), os.path.join(ROOT, "include", "ocr_opstream.h")]


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def build_lib(force=False):
    newest = max(os.path.getmtime(p) for p in SRC + HDRS)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= newest:
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", os.path.join(ROOT, "include")] + SRC + ["-o", LIB]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True))
