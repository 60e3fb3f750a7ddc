#!/bin/bash
mkdir -p gpurun_out/r5i
python3 -m pytest tests/test_swin_ops_gpu.py -x -q > gpurun_out/r5i/pytest_swin_ops.log 2>&1
python3 -m pytest tests/test_swin_gpu.py -x -q -k "not config5" > gpurun_out/r5i/pytest_swin.log 2>&1
python3 tools/wgrad9_table.py > gpurun_out/r5i/wgrad9_table.txt 2> gpurun_out/r5i/wgrad9_table.err
GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so python3 tools/wgrad9_table.py --cycles > gpurun_out/r5i/wgrad9_table_cycles.txt 2>> gpurun_out/r5i/wgrad9_table.err
bash tools/ab_env.sh 2 100 X=1 > gpurun_out/r5i/base.txt 2>&1
