#!/bin/bash
O=gpurun_out/r5v; mkdir -p $O
bash tools/ab_env.sh 2 100 X=1 GDL_WGRAD_BLOCKS=96 GDL_WGRAD_BLOCKS=192 GDL_WGRAD_BLOCKS=256 > $O/ab_default.txt 2>&1
export GDL_TUNING=1
for r in 1 2; do for s in X=1 GDL_WGRAD_GEMM_BLOCKS=256 GDL_WGRAD_GEMM_BLOCKS=512 GDL_WGRAD_GEMM_BLOCKS=768; do
out=$(env $s python3 bench.py --workload vggsound_swin --steps 40 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
echo "round $r  $s  $out"; done; done > $O/ab_swin.txt 2>&1
