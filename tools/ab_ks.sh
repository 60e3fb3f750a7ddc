export GDL_TUNING=1
for r in 1 2 3; do for s in GDL_C64_WIDE=0 X=1; do
out=$(env $s python3 bench.py --workload ks --steps 60 --warmup 10 --no-cpu-baseline --no-f32 --no-prof 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])"); echo "round $r $s $out"; done; done
