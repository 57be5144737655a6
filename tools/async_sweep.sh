# k_async: waves x entries per ticket block, ms per step of bench.py (config 2); pairs from the command line or a default grid
pairs=${@:-"1024:16 1024:24 1024:32 1280:16 1280:24 1280:32 1536:16 1536:24 1536:32 2048:16 2048:24 2048:32 2560:16 2560:24 2560:32"}
for p in $pairs; do w=${p%%:*}; b=${p##*:}
  r=$(OCRGPU_ASYNC_WAVES=$w OCRGPU_ASYNC_BLOCK=$b timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config']['parity'][:9])") || exit 1
  echo "waves $w block $b: $r"
done
