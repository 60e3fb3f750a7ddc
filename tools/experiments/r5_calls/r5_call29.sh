#!/bin/bash
O=gpurun_out/r5ab; mkdir -p $O
B=iccv2025-gdl_amd/csrc/build_base/libgdl_hip.so; N=iccv2025-gdl_amd/csrc/build/libgdl_hip.so
python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "stem" > $O/pytest.log 2>&1
bash tools/ab.sh 3 100 $B $N > $O/ab.txt 2>&1
python3 tools/utilisation_timeline.py --launches --out $O/timeline.txt > /dev/null 2> $O/timeline.err
GDL_LIB=$PWD/$B python3 tools/utilisation_timeline.py --launches --out $O/timeline_base.txt > /dev/null 2>> $O/timeline.err
