#!/bin/bash
# round 5, call 19: head / tail kernels of the Swin composition (token mean, unimodal head, cross-entropy, concat head forward)
O=gpurun_out/r5r; mkdir -p $O
B=iccv2025-gdl_amd/csrc/build_base/libgdl_hip.so; N=iccv2025-gdl_amd/csrc/build/libgdl_hip.so
python3 -m pytest tests/test_ops_gpu.py tests/test_swin_ops_gpu.py tests/test_step_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
AB_ARGS="--workload vggsound_swin" bash tools/ab.sh 2 40 $B $N > $O/ab_swin.txt 2>&1
bash tools/ab.sh 2 100 $B $N > $O/ab_default.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --workload vggsound_swin --steps 4 --warmup 2 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator > $O/kt_stdout.log 2>&1
python3 tools/trace_window.py $O/kt --grep "head_|softmax|token_mean" > $O/window_swin.txt 2>&1
rm -rf $O/kt
