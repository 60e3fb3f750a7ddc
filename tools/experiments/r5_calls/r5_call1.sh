#!/bin/bash
# round-5 GPU call 1: baseline bench line, utilisation timeline, skip bounds of the convolution classes
mkdir -p gpurun_out/r5a
python3 bench.py --no-cpu-baseline --no-extra --no-comparator --no-f32 > gpurun_out/r5a/bench_default.json 2> gpurun_out/r5a/bench_default.err
python3 tools/utilisation_timeline.py --launches --out gpurun_out/r5a/timeline.txt > /dev/null 2> gpurun_out/r5a/timeline.err
MASKS="0 1024 2048 4096 6144 8192 16384 32768 65536 131072 262144 524288 1048576 0" bash tools/skip_bounds.sh 1 > gpurun_out/r5a/skip.txt 2>&1
