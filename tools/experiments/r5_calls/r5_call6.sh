#!/bin/bash
mkdir -p gpurun_out/r5f
python3 -m pytest tests/test_ops_gpu.py -x -q -k "stem" > gpurun_out/r5f/pytest_stem.log 2>&1
python3 bench.py --no-cpu-baseline --no-extra --no-comparator --no-f32 --steps 40 > gpurun_out/r5f/bench_fused.json 2> gpurun_out/r5f/bench_fused.err
bash tools/ab_env.sh 3 100 X=1 GDL_STEM_FUSED=0 > gpurun_out/r5f/ab_stem_fused.txt 2>&1
