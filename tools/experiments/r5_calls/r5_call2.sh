#!/bin/bash
mkdir -p gpurun_out/r5b
python3 tools/utilisation_timeline.py --launches --out gpurun_out/r5b/timeline.txt > /dev/null 2> gpurun_out/r5b/timeline.err
