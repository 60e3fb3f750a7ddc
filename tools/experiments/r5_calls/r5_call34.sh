#!/bin/bash
O=gpurun_out/r5ag; mkdir -p $O
B=iccv2025-gdl_amd/csrc/build_base/libgdl_hip.so; N=iccv2025-gdl_amd/csrc/build/libgdl_hip.so
python3 -m pytest tests/test_swin_ops_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
echo "## before" > $O/ln.txt; GDL_LIB=$PWD/$B python3 tools/bench_swin_ln.py 2>/dev/null >> $O/ln.txt
echo "## after" >> $O/ln.txt; GDL_LIB=$PWD/$N python3 tools/bench_swin_ln.py 2>/dev/null >> $O/ln.txt
AB_ARGS="--workload vggsound_swin" bash tools/ab.sh 3 40 $B $N > $O/ab_swin.txt 2>&1
