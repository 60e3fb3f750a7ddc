#!/bin/bash
mkdir -p gpurun_out/r5j
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -k "dgrad or relu_bits" > gpurun_out/r5j/pytest_dgrad.log 2>&1
python3 tools/bench_conv.py > gpurun_out/r5j/bench_conv_s2d.txt 2>&1
GDL_TUNING=1 GDL_S2D=0 python3 tools/bench_conv.py > gpurun_out/r5j/bench_conv_flat.txt 2>&1
timeout 900 python3 -m pytest tests/test_step_gpu.py -x -q -k "golden or full_size" > gpurun_out/r5j/pytest_step.log 2>&1
bash tools/ab_env.sh 3 100 X=1 GDL_S2D=0 > gpurun_out/r5j/ab_s2d.txt 2>&1
