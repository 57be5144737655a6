#!/bin/bash
# Kernel trace + request-count PMC pass of a short windowed-stencil run, summarised on the box (the rocpd files are too big to bring back):
#   bash tools/sw_profile.sh <tag> [gx gy steps window]   ->  gpurun_out/<tag>/{kernel_stats.txt,pmc_requests.txt,trace.json}
set -o pipefail
tag=${1:-sw}; shift
args=${@:-2048 1024 50 24}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/swp && mkdir -p /tmp/swp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/swp/trace -o t -- python3 $root/tools/stencil_window.py $args > $out/trace.json 2> $out/trace.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d /tmp/swp/pmc -o p -- python3 $root/tools/stencil_window.py $args > $out/pmc.json 2> $out/pmc.err || exit 1
python3 - $out <<'PY'
import glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob('/tmp/swp/trace/**/*.db', recursive=True)[0])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out + '/kernel_stats.txt', 'w') as f:
    f.write(f"total kernel ms {tot / 1e6:.2f}\n")
    for n, c, t, a, mn, mx in rows[:40]:
        f.write(f"{t / 1e6:9.2f} ms {100 * t / tot:5.1f}% calls {c:5d} avg {a / 1e3:9.1f} us min {mn / 1e3:9.1f} max {mx / 1e3:9.1f}  {n[:100]}\n")
db = sqlite3.connect(glob.glob('/tmp/swp/pmc/**/*.db', recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
with open(out + '/pmc_requests.txt', 'w') as f:
    try:
        q = "select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name order by 1, 2"
        for n, cn, cnt, v in db.execute(q):
            f.write(f"{n[:60]:60s} {cn:24s} dispatches {cnt:6d} sum {v:16.0f} per dispatch {v / cnt:14.0f}\n")
    except Exception as ex:
        f.write(f"query failed: {ex}\ntables: {tabs}\n")
        for t in tabs:
            if 'counter' in t.lower() or 'pmc' in t.lower():
                f.write(t + ': ' + str([r[1] for r in db.execute(f'pragma table_info({t})')]) + '\n')
PY
cat $out/kernel_stats.txt | head -30; cat $out/pmc_requests.txt | head -60
