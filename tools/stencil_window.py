#!/usr/bin/env python3
"""BASELINE config 4 at its stated length on one GPU's share of the tile grid (4096 x 4096 tiles block-
partitioned 2 x 4 over 8 GPUs = 2048 x 1024 tiles per GPU, 1000 time steps = 2.1e9 tile-steps), as a sliding window over
time (ocr_amd/stencil_window.py).  One progress line per window on stderr, one JSON line at the end.

    python tools/stencil_window.py 2048 1024 1000 24
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    gx, gy, steps, window = (int(x) for x in sys.argv[1:5])
    from ocr_amd.stencil_window import StencilWindow

    def progress(msg):
        print(f"[stencil_window {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    w = StencilWindow(gx, gy, steps, window, progress=progress)
    progress(f"engine for {window + 2} steps of {gx * gy} tiles: max_edt {int(w.eng.cfg.max_edt)}, max_edges {int(w.eng.cfg.max_edges)}")
    w.run()
    s = w.summary()
    s["parity"] = ("every tile-step at its closed-form level, every value equal to the vectorised HASH recurrence"
                   if s["level_errors"] == 0 and s["value_errors"] == 0 and sum(s["err"]) == 0 else "MISMATCH")
    print(json.dumps(s), flush=True)
    return 0 if s["parity"] != "MISMATCH" else 1


if __name__ == "__main__":
    sys.exit(main())
