#!/bin/bash
O=gpurun_out/r5ad; mkdir -p $O
MASKS="0 4194304 4194560" bash tools/skip_bounds.sh 3 > $O/skip_stage.txt 2>&1
