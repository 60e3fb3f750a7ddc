#!/bin/bash
# round-6 profile collection (GPU box, repo root): PMC passes, kernel-trace statistics, bench lines, utilisation timeline, wgrad9 table,
# the slab kernels' timing probe before (GDL_PSLAB=0: round 5's kernels) / after, the data gradients alone, the K-step micro-benchmark
export GDL_ROUND=06
bash tools/collect_profiles.sh
python3 tools/utilisation_timeline.py --launches --out gpurun_out/fin/utilisation_timeline.txt > /dev/null 2> gpurun_out/fin/timeline.err
python3 tools/wgrad9_table.py > gpurun_out/fin/wgrad9_table.txt 2> gpurun_out/fin/wgrad9_table.err
TL=$PWD/iccv2025-gdl_amd/csrc/build_timing/libgdl_hip.so
GDL_LIB=$TL python3 tools/wgrad9_table.py --cycles > gpurun_out/fin/wgrad9_table_cycles.txt 2>> gpurun_out/fin/wgrad9_table.err
for mode in before after; do
  for shp in 192,128,28,28,128,3,1,1 192,256,14,14,256,3,1,1 192,512,7,7,512,3,1,1 64,128,33,24,128,3,1,1 64,256,17,12,256,3,1,1 64,512,9,6,512,3,1,1; do
    for op in fwd dgrad; do
      echo "=== $op $shp"
      if [ $mode = before ]; then
        GDL_TUNING=1 GDL_PSLAB=0 GDL_LIB=$TL timeout 120 python3 tools/timing_probe.py --op $op --shape $shp 2>&1 | grep -v "Warning\|amdgpu.ids\|_methods\|fromnumeric\|ret = "
      else
        GDL_LIB=$TL timeout 120 python3 tools/timing_probe.py --op $op --shape $shp --names entry,issued,landed,kloop_end,kloop_end2,end,rounds,fold 2>&1 | grep -v "Warning\|amdgpu.ids"
      fi
    done
  done > gpurun_out/fin/timing_probe_$mode.txt 2>&1
done
python3 tools/bench_dgrad_bn.py > gpurun_out/fin/bench_dgrad_bn.txt 2>&1
GDL_TUNING=1 GDL_PSLAB=0 python3 tools/bench_dgrad_bn.py > gpurun_out/fin/bench_dgrad_bn_round5_kernels.txt 2>&1
python3 tools/bench_conv.py > gpurun_out/fin/bench_conv.txt 2>&1
(cd tools/micro && ./kstep) > gpurun_out/fin/micro_kstep.txt 2>&1
# the GPU suite and the smoke entry on the same box
python3 -m pytest tests -m gpu -q --durations=12 > gpurun_out/fin/pytest_gpu.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/fin/smoke.log 2>&1
tail -3 gpurun_out/fin/pytest_gpu.log
