#!/bin/bash
mkdir -p gpurun_out/r5d
python3 -m pytest tests/test_ops_gpu.py -x -q -k "stem" > gpurun_out/r5d/pytest_stem.log 2>&1
python3 -m pytest tests/test_step_gpu.py -x -q > gpurun_out/r5d/pytest_step.log 2>&1
bash tools/ab_env.sh 3 100 X=1 GDL_STEM_FUSED=0 > gpurun_out/r5d/ab_stem_fused.txt 2>&1
