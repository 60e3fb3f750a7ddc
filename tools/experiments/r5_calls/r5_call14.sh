#!/bin/bash
mkdir -p gpurun_out/r5n
bash tools/pmc_one.sh w9_l4 wgrad 192,512,7,7,512,3,1,1 > gpurun_out/r5n/pmc_w9_l4.txt 2>&1
bash tools/pmc_one.sh w9_l3 wgrad 192,256,14,14,256,3,1,1 > gpurun_out/r5n/pmc_w9_l3.txt 2>&1
