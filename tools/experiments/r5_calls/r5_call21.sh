#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
bash tools/ab_env.sh 3 100 X=1 GDL_SPEC_BOUND=1 GDL_NO_OPT=1 > $O/ab_spec_bound.txt 2>&1
