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


SHIM_SRC = os.path.join(HERE, "csrc", "ocrshim.c")
SHIM_LIB = os.path.join(HERE, "libocrshim.a")
SHIM_INC = os.path.join(ROOT, "include", "ocr_shim")
PROG_DIR = os.path.join(HERE, "bin")


def build_shim(force=False):
    """libocrshim.a: the OCR API (include/ocr_shim/ocr.h) over the C-ABI, plain C host code."""
    deps = [SHIM_SRC, os.path.join(SHIM_INC, "ocr.h")] + HDRS
    if not force and os.path.exists(SHIM_LIB) and os.path.getmtime(SHIM_LIB) >= max(os.path.getmtime(p) for p in deps):
        return SHIM_LIB
    obj = os.path.join(HERE, "ocrshim.o")
    subprocess.check_call(["gcc", "-O2", "-g", "-std=gnu11", "-Wall", "-fPIC", "-I", SHIM_INC, "-I", os.path.join(ROOT, "include"),
                           "-c", SHIM_SRC, "-o", obj])
    if os.path.exists(SHIM_LIB):
        os.remove(SHIM_LIB)
    subprocess.check_call(["ar", "crs", SHIM_LIB, obj])
    return SHIM_LIB


def build_program(src, name=None, force=False, defines=()):
    """Compile an OCR program (a C file defining mainEdt) against the shim: ocr_amd/bin/<name>."""
    build_lib()
    build_shim()
    os.makedirs(PROG_DIR, exist_ok=True)
    out = os.path.join(PROG_DIR, name or os.path.splitext(os.path.basename(src))[0])
    deps = [src, SHIM_LIB, LIB]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in deps):
        return out
    subprocess.check_call(["gcc", "-O2", "-g", "-std=gnu11", "-w", "-I", SHIM_INC, "-I", os.path.join(ROOT, "include")]
                          + [f"-D{d}" for d in defines] + [src, SHIM_LIB, "-L", HERE, "-locrgpu",
                           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out])
    return out


def build_cabi_host(src, name=None, force=False):
    """Compile a plain C host of libocrgpu (no shim, no OCR headers): ocr_amd/bin/<name>."""
    build_lib()
    os.makedirs(PROG_DIR, exist_ok=True)
    out = os.path.join(PROG_DIR, name or os.path.splitext(os.path.basename(src))[0])
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in [src, LIB] + HDRS):
        return out
    subprocess.check_call(["gcc", "-O2", "-g", "-std=gnu11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", HERE, "-locrgpu",
                           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link,/opt/rocm/lib", "-o", out])
    return out


if __name__ == "__main__":
    print(build_lib(force=True))
    print(build_shim(force=True))
