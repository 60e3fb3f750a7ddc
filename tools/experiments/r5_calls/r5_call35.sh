#!/bin/bash
O=gpurun_out/r5ah; mkdir -p $O
bash tools/ab_env.sh 2 100 X=1 GDL_AUDIO_DELAY=-1 GDL_AUDIO_DELAY=1 GDL_AUDIO_DELAY=2 GDL_AUDIO_DELAY=4 > $O/ab.txt 2>&1
