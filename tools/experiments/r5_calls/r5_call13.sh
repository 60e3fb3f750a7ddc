#!/bin/bash
mkdir -p gpurun_out/r5m
python3 -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r5m/pytest_gpu.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5m/smoke.log 2>&1
