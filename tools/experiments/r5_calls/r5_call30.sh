#!/bin/bash
O=gpurun_out/r5ac; mkdir -p $O
python3 tools/bench_stem_bwd.py > $O/stem_bwd.txt 2>&1
GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so python3 tools/bench_stem_bwd.py --cycles >> $O/stem_bwd.txt 2>&1
GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_base/libgdl_hip.so python3 tools/bench_stem_bwd.py >> $O/stem_bwd.txt 2>&1
