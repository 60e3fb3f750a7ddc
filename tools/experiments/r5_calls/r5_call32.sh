#!/bin/bash
O=gpurun_out/r5ae; mkdir -p $O
python3 -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" > $O/pytest.log 2>&1
export GDL_TUNING=1
for s in 0 1; do
  echo "## GDL_WGRAD9_SPREAD=$s" >> $O/w9.txt
  GDL_WGRAD9_SPREAD=$s python3 tools/wgrad9_table.py 2>/dev/null | grep -v "^#" >> $O/w9.txt
  GDL_WGRAD9_SPREAD=$s GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so python3 tools/wgrad9_table.py --cycles 2>/dev/null | grep -v "^#" | cut -c150- >> $O/w9.txt
done
bash tools/ab_env.sh 3 100 GDL_WGRAD9_SPREAD=0 X=1 > $O/ab.txt 2>&1
