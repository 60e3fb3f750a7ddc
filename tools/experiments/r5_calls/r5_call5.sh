#!/bin/bash
mkdir -p gpurun_out/r5e
python3 bench.py --no-cpu-baseline --no-extra --no-comparator --no-f32 --steps 40 > gpurun_out/r5e/bench_fused.json 2> gpurun_out/r5e/bench_fused.err
GDL_TUNING=1 GDL_STEM_FUSED=0 python3 bench.py --no-cpu-baseline --no-extra --no-comparator --no-f32 --steps 40 > gpurun_out/r5e/bench_unfused.json 2> gpurun_out/r5e/bench_unfused.err
