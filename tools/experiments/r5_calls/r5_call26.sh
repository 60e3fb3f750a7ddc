#!/bin/bash
O=gpurun_out/r5y; mkdir -p $O
bash tools/ab_env.sh 2 100 X=1 GDL_C64_SMALL_GRID=256 GDL_C64_SMALL_GRID=384 GDL_C64_SMALL_GRID=128 > $O/ab_c64_grid.txt 2>&1
