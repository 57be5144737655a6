#!/bin/bash
# Wide levels (layered 100 x 100 000): kernel trace + request / byte PMC passes of tools/bench_configs.py, summarised on the box.
#   bash tools/wide_profile.sh <tag> [spec]   ->  gpurun_out/<tag>/{run.json,kernel_stats.txt,pmc.txt}
set -o pipefail
tag=${1:-wide}; spec=${2:-layered:10000000:100000}
root=$PWD; out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/wp && mkdir -p /tmp/wp
timeout -k 10 250 rocprofv3 --kernel-trace --stats -d /tmp/wp/trace -o t -- python3 $root/tools/bench_configs.py $spec --no-parity > $out/run.json 2> $out/run.err || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 250 rocprofv3 --pmc $pass -d /tmp/wp/pmc_$n -o p -- python3 $root/tools/bench_configs.py $spec --no-parity > $out/pmc_$n.json 2> $out/pmc_$n.err || exit 1
done
python3 - $out <<'PY'
import glob, sqlite3, sys
out = sys.argv[1]
db = sqlite3.connect(glob.glob('/tmp/wp/trace/**/*.db', recursive=True)[0])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out + '/kernel_stats.txt', 'w') as f:
    f.write(f"total kernel ms {tot / 1e6:.2f}\n")
    for n, c, t, a, mn, mx in rows[:12]:
        f.write(f"{t / 1e6:9.2f} ms {100 * t / tot:5.1f}% calls {c:5d} avg {a / 1e3:9.1f} us min {mn / 1e3:9.1f} max {mx / 1e3:9.1f}  {n[:100]}\n")
    ks = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like '%k_step%' order by start")]
    big = [x for x in ks if x > 15]
    f.write(f"k_step launches {len(ks)}; above 15 us: {len(big)}, mean {sum(big) / max(len(big), 1):.1f} us\n")
with open(out + '/pmc.txt', 'w') as f:
    for p in glob.glob('/tmp/wp/pmc_*/**/*.db', recursive=True):
        d2 = sqlite3.connect(p)
        for n, cn, cnt, v in d2.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection where kernel_name like '%k_step%' group by kernel_name, counter_name"):
            f.write(f"{n[:50]:50s} {cn:22s} dispatches {cnt:6d} sum {v:16.0f} per dispatch {v / cnt:14.0f}\n")
PY
cat $out/run.json | cut -c1-600; cat $out/kernel_stats.txt | head -8; cat $out/pmc.txt
