#!/usr/bin/env python3
"""Read the two PMC passes of tools/pmc_calib.hip and print reported / expected bytes per access pattern.
usage: python tools/pmc_calib.py gpurun_out   (expects gpurun_out/calib_fetch and gpurun_out/calib_write)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_summary import pmc_per_kernel  # noqa: E402

LINES = 1 << 24
EXPECT = {  # kernel -> (bytes that must be fetched, bytes that must be written back) through the memory side
    "calib_line64_read": (LINES * 64, 0), "calib_word8_read": (LINES * 64, 0),
    "calib_atomic4_returning": (LINES * 64, LINES * 64), "calib_atomic4_noreturn": (LINES * 64, LINES * 64),
    "calib_store4": (0, LINES * 64), "calib_store4_stream": (0, LINES * 4),
}


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
    got = {}
    for sub, ctr in (("calib_fetch", "FETCH_SIZE"), ("calib_write", "WRITE_SIZE")):
        p = os.path.join(root, sub, "pmc_results.db")
        if not os.path.exists(p):
            continue
        for k, ctrs in pmc_per_kernel(p).items():
            for name in EXPECT:
                if name in k and ctr in ctrs:
                    got.setdefault(name, {})[ctr] = ctrs[ctr][1] * 1024.0     # the counters are in KiB
    print(f"{'pattern':28s} {'FETCH_SIZE':>14s} {'expected':>14s} {'ratio':>6s} | {'WRITE_SIZE':>14s} {'expected':>14s} {'ratio':>6s}")
    for name, (ef, ew) in EXPECT.items():
        f, w = got.get(name, {}).get("FETCH_SIZE"), got.get(name, {}).get("WRITE_SIZE")
        fr = f"{f / ef:6.2f}" if f is not None and ef else "     -"
        wr = f"{w / ew:6.2f}" if w is not None and ew else "     -"
        print(f"{name:28s} {f if f is not None else float('nan'):14.0f} {ef:14d} {fr} | {w if w is not None else float('nan'):14.0f} {ew:14d} {wr}")


if __name__ == "__main__":
    main()
