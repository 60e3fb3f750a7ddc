#!/bin/bash
mkdir -p gpurun_out/r5l
python3 tools/bench_swin_split.py > gpurun_out/r5l/swin_split.txt 2>&1
