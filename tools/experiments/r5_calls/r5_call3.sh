#!/bin/bash
mkdir -p gpurun_out/r5c
export GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_exp/libgdl_hip.so
run() { env "$@" python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])"; }
for r in 1 2; do
  echo "round $r base $(run X=1)"
  echo "round $r skip2=1 (wgrad folds) $(run GDL_SKIP2=1)"
  echo "round $r skip=2097152 (1x1 shortcut fwd) $(run GDL_SKIP=2097152)"
  echo "round $r no_opt $(run GDL_TUNING=1 GDL_NO_OPT=1)"
  echo "round $r skip=65536 (all s1 fwd) $(run GDL_SKIP=65536)"
  echo "round $r skip=4 (bn1 bwd apply) $(run GDL_SKIP=4)"
  echo "round $r skip=8 (bn2 bwd apply) $(run GDL_SKIP=8)"
  echo "round $r skip=1 (bn1 fwd apply) $(run GDL_SKIP=1)"
  echo "round $r skip=192 (stem pool fwd+bwd) $(run GDL_SKIP=192)"
done > gpurun_out/r5c/bounds.txt 2>&1
GDL_SKIP=65536 python3 bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator > gpurun_out/r5c/s65536.log 2>&1
