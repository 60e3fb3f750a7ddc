#!/bin/bash
# A/B of library builds on one box: alternates the default bench workload over the given libgdl_hip.so paths.
# usage (GPU box, repo root): [AB_ARGS="--workload vggsound_swin"] bash tools/ab.sh <rounds> <steps> libA.so libB.so ...
set -u
ROUNDS=$1; STEPS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    out=$(GDL_LIB=$PWD/$lib python3 bench.py ${AB_ARGS:-} --steps $STEPS --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
    echo "round $r  $lib  $out"
  done
done
