#!/bin/bash
# round-5 profile collection (GPU box, repo root): PMC passes, kernel-trace statistics, bench lines, utilisation timeline, wgrad9 table
export GDL_ROUND=05
bash tools/collect_profiles.sh
python3 tools/utilisation_timeline.py --launches --out gpurun_out/fin/utilisation_timeline.txt > /dev/null 2> gpurun_out/fin/timeline.err
python3 tools/wgrad9_table.py > gpurun_out/fin/wgrad9_table.txt 2> gpurun_out/fin/wgrad9_table.err
GDL_LIB=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so python3 tools/wgrad9_table.py --cycles > gpurun_out/fin/wgrad9_table_cycles.txt 2>> gpurun_out/fin/wgrad9_table.err
# the GPU suite and the smoke entry on the same box (profiles/r05_pytest_gpu.log)
mkdir -p gpurun_out/r5m
python3 -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r5m/pytest_gpu.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5m/smoke.log 2>&1
