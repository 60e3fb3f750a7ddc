#!/bin/bash
O=gpurun_out/r5p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator > $O/kt_stdout.log 2>&1
python3 tools/trace_window.py $O/kt --grep rocclr > $O/window_rocclr.txt 2>&1
python3 tools/trace_window.py $O/kt > $O/window_all.txt 2>&1
rm -rf $O/kt
