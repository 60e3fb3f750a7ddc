#!/bin/bash
O=gpurun_out/r5z; mkdir -p $O
python3 tools/utilisation_timeline.py --workload vggsound_swin --launches --window 500 --out $O/timeline_swin.txt > /dev/null 2> $O/timeline_swin.err
