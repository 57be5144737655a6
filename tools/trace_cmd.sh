#!/bin/bash
# Kernel trace of one python tool, summarised on the box (the rocpd files are too big to bring back):
#   bash tools/trace_cmd.sh <tag> tools/dist_timing.py windows2   ->  gpurun_out/<tag>/{kernel_stats.txt,out.txt}
set -o pipefail
tag=$1; shift
script=$1; shift
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trc && mkdir -p /tmp/trc
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/trc/trace -o t -- python3 $root/$script "$@" > $out/out.txt 2> $out/err.txt || exit 1
python3 - $out <<'PY'
import glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob('/tmp/trc/trace/**/*.db', recursive=True)[0])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out + '/kernel_stats.txt', 'w') as f:
    f.write(f"total kernel ms {tot / 1e6:.2f}\n")
    for n, c, t, a, mn, mx in rows[:40]:
        f.write(f"{t / 1e6:9.2f} ms {100 * t / tot:5.1f}% calls {c:5d} avg {a / 1e3:9.1f} us min {mn / 1e3:9.1f} max {mx / 1e3:9.1f}  {n[:100]}\n")
PY
python3 - $out <<'PY'
# do kernels of different streams overlap?  sum of durations against the union of the busy intervals, per queue and overall
import glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob('/tmp/trc/trace/**/*.db', recursive=True)[0])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute(f"select start, end, {qcol or 0} from kernels order by start").fetchall()
tot = sum(e - s for s, e, _ in rows); union = 0; cur_s = cur_e = None
for s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
if cur_e is not None: union += cur_e - cur_s
qs = {}
for s, e, q in rows: qs[q] = qs.get(q, 0) + (e - s)
with open(out + '/kernel_stats.txt', 'a') as f:
    f.write(f"overlap: sum of kernel durations {tot / 1e6:.2f} ms, union of busy intervals {union / 1e6:.2f} ms ({qcol}: " + ", ".join(f"{q}: {t / 1e6:.2f} ms" for q, t in sorted(qs.items())) + ")\n")
PY
cat $out/out.txt | tail -5; tail -1 $out/kernel_stats.txt; head -14 $out/kernel_stats.txt
