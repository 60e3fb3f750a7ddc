#!/bin/bash
# "Skip bounds" (timing experiments; run on the GPU box from the repo root): the default bench over a -DGDL_EXPERIMENT build of the
# library (make -C iccv2025-gdl_amd/csrc BUILD=build_exp EXTRA=-DGDL_EXPERIMENT) with one class of launches left out per setting
# (GDL_SKIP bit mask, csrc/encoder.cpp).  The results of such a step are WRONG; the step-time difference to GDL_SKIP=0 is the most
# that removing / fusing that pass could return -- measured before a fusion is built, not after.
set -u
export GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_exp/libgdl_hip.so
ROUNDS=${1:-2}
for r in $(seq 1 $ROUNDS); do
  for m in ${MASKS:-0 2 1 4 8 12 16 32 48 64 128 256 512}; do
    out=$(GDL_SKIP=$m python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
    echo "round $r  GDL_SKIP=$m  $out"
  done
done
