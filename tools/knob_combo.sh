#!/bin/bash
export GDL_TUNING=1
run() { local label=$1; shift; out=$(env "$@" python3 bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-f32 --no-prof 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])"); echo "$label $out"; }
for r in 1 2 3; do
run "base      " X=1
run "minst16   " GDL_WGRAD9_MINST=16
run "wg128     " GDL_WGRAD_BLOCKS=128
done
