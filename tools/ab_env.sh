#!/bin/bash
# A/B of tuning-knob settings on one box: alternates the default bench workload over the given "VAR=VALUE[,VAR=VALUE]" settings
# (GDL_TUNING=1 is set; "X=1" is the baseline).  usage (GPU box, repo root): bash tools/ab_env.sh <rounds> <steps> SETTING ...
set -u
ROUNDS=$1; STEPS=$2; shift 2
export GDL_TUNING=1
for r in $(seq 1 $ROUNDS); do
  for s in "$@"; do
    out=$(env ${s//,/ } python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
    echo "round $r  $s  $out"
  done
done
