for w in 1024 1280 1536 2048 2560; do for b in 16 24 32; do
  r=$(OCRGPU_ASYNC_WAVES=$w OCRGPU_ASYNC_BLOCK=$b timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['config']['parity'][:9])") || exit 1
  echo "waves $w block $b: $r"
done; done
