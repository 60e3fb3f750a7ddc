#!/bin/bash
# One-at-a-time sweep of the library's tuning knobs on the default bench workload (GDL_TUNING=1); prints ms/step per setting.
# usage (GPU box, repo root): bash tools/knob_sweep.sh [steps]
set -u
STEPS=${1:-40}
export GDL_TUNING=1
run() {  # run <label> VAR=VALUE ...
  local label=$1; shift
  local out
  out=$(env "$@" python3 bench.py --steps $STEPS --warmup 8 --no-cpu-baseline --no-f32 --no-prof 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
  echo "$label  $out"
}
run "base                 " X=1
run "base (again)         " X=1
for v in 0 2 3; do run "SLAB_CFG=$v            " GDL_SLAB_CFG=$v; done
for v in 4 12 16; do run "WGRAD9_MINST=$v        " GDL_WGRAD9_MINST=$v; done
for v in 128 256 512; do run "WGRAD_BLOCKS=$v        " GDL_WGRAD_BLOCKS=$v; done
for v in 1 4; do run "CONV_CFG=$v             " GDL_CONV_CFG=$v; done
run "SLAB_BM=256          " GDL_SLAB_BM=256
run "EW_V4=0              " GDL_EW_V4=0
run "STEM_ROWS_BLOCKS=512 " GDL_STEM_ROWS_BLOCKS=512
run "base (end)           " X=1
