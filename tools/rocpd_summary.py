#!/usr/bin/env python3
"""Turn the rocpd sqlite files rocprofv3 writes (tools/profile_config2.sh) into the small text summaries kept under profiles/.

usage: python tools/rocpd_summary.py gpurun_out/<tag> profiles/<tag>
"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """What bench.py compares before it quotes a profile: the engine source the profiled library was built from."""
    return hashlib.sha256(open(os.path.join(ROOT, "ocr_amd", "csrc", "ocrgpu.hip"), "rb").read()).hexdigest()[:16]


def kernel_stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
        for n, c, t, a, mn, mx in rows:
            f.write(f'"{n}",{c},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}\n')
    return rows


def pmc_per_kernel(db_path):
    db = sqlite3.connect(db_path)
    out = {}
    for name, ctr, n, s in db.execute(
        "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"
    ):
        out.setdefault(name, {})[ctr] = (n, s)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    rows = kernel_stats(os.path.join(src, "trace", "trace_results.db"), dst + "_kernel_stats.csv")
    lines = ["source: ocr_amd/csrc/ocrgpu.hip sha256[:16]=" + kernel_source_hash()]
    bench = json.loads(open(os.path.join(src, "trace.json")).read().strip().splitlines()[-1])
    lines.append("bench line of the traced run: " + json.dumps({k: bench[k] for k in ("value", "ms_per_step", "steps", "warmup")}))
    for n, c, t, a, mn, mx in rows[:6]:
        lines.append(f"kernel-trace: {n}: {c} launches, avg {a / 1000:.3f} us, min {mn / 1000:.3f}, max {mx / 1000:.3f}, total {t / 1e6:.3f} ms")
    for sub in ("pmc_fetch", "pmc_write", "pmc_atom"):
        p = os.path.join(src, sub, "pmc_results.db")
        if not os.path.exists(p):
            continue
        per = pmc_per_kernel(p)
        for k, ctrs in per.items():
            if "k_step" not in k and "k_async(" not in k and not k.startswith("k_async"):
                continue
            for ctr, (n, s) in sorted(ctrs.items()):
                lines.append(f"{sub}: {k}: {ctr} sum={s:.0f} over {n} dispatches (per dispatch {s / n:.1f})")
    open(dst + "_pmc_summary.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
