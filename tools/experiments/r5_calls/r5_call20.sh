#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
python3 tools/utilisation_timeline.py --launches --out $O/timeline.txt > /dev/null 2> $O/timeline.err
