#!/bin/bash
O=gpurun_out/r5w; mkdir -p $O
export GDL_TUNING=1
for r in 1 2; do for s in X=1 GDL_WGRAD_GEMM_BLOCKS=320 GDL_WGRAD_GEMM_BLOCKS=352 GDL_WGRAD_GEMM_BLOCKS=416 GDL_WGRAD_GEMM_BLOCKS=448; do
out=$(env $s python3 bench.py --workload vggsound_swin --steps 40 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
echo "round $r  $s  $out"; done; done > $O/ab_swin.txt 2>&1
