#!/bin/bash
# round 5, call 15: the XCD-linear block mapping (common.h xcd_linear) against the per-slice / per-M-tile cut: operator parity,
# every conv shape alone with both libraries, the step A/B
O=gpurun_out/r5o; mkdir -p $O
B=iccv2025-gdl_amd/csrc/build_base/libgdl_hip.so; N=iccv2025-gdl_amd/csrc/build/libgdl_hip.so
python3 -m pytest tests/test_ops_gpu.py tests/test_encoder_gpu.py -m gpu -x -q > $O/pytest_ops.log 2>&1
GDL_LIB=$PWD/$B python3 tools/bench_conv.py > $O/conv_base.txt 2>&1
GDL_LIB=$PWD/$N python3 tools/bench_conv.py > $O/conv_new.txt 2>&1
python3 tools/wgrad9_table.py > $O/w9_new.txt 2>&1
bash tools/ab.sh 3 100 $B $N > $O/ab.txt 2>&1
