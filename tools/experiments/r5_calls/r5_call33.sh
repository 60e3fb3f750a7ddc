#!/bin/bash
O=gpurun_out/r5af; mkdir -p $O
for b in build build_a4 build_a5; do
  echo "## $b" >> $O/w9.txt
  GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/$b/libgdl_hip.so python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -1 >> $O/w9.txt
  GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/$b/libgdl_hip.so python3 tools/wgrad9_table.py 2>/dev/null | grep -v "^#" | cut -c1-120 >> $O/w9.txt
done
bash tools/ab.sh 2 100 iccv2025-gdl_amd/csrc/build/libgdl_hip.so iccv2025-gdl_amd/csrc/build_a4/libgdl_hip.so iccv2025-gdl_amd/csrc/build_a5/libgdl_hip.so > $O/ab.txt 2>&1
