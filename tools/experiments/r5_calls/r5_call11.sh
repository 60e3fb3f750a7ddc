#!/bin/bash
mkdir -p gpurun_out/r5k
python3 -m pytest tests/test_step_gpu.py -x -q -k "optim or golden" > gpurun_out/r5k/pytest_optim.log 2>&1
python3 tools/pg_variants.py --steps 60 --emulate-traffic 16 > gpurun_out/r5k/pg_own_stream.txt 2>&1
python3 tools/pg_variants.py --steps 60 --emulate-traffic 16 --traffic-on-caller > gpurun_out/r5k/pg_on_caller.txt 2>&1
bash tools/ab_env.sh 2 100 X=1 > gpurun_out/r5k/base.txt 2>&1
