# two k_async settings on several DAG shapes: median ms per resolve (tools/soak_async.py), digest checked on every resolve
for shape in "10000000 10000" "10000000 50000" "2000000 1000" "200000 100" "5000000 25000"; do
  for p in 1280:24 1536:16; do w=${p%%:*}; b=${p##*:}
    r=$(OCRGPU_ASYNC_WAVES=$w OCRGPU_ASYNC_BLOCK=$b timeout -k 10 200 python tools/soak_async.py 60 $shape 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_median'],3), round(d['ms_min'],3), d['mismatches'])") || exit 1
    echo "shape $shape waves $w block $b: median/min/mismatches $r"
  done
done
