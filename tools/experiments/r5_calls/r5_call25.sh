#!/bin/bash
O=gpurun_out/r5x; mkdir -p $O
bash tools/ab_env.sh 2 100 X=1 GDL_SPLITK=1 GDL_SPLITK=1,GDL_SPLITK_BLOCKS=150 > $O/ab_splitk.txt 2>&1
