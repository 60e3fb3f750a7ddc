#!/bin/bash
mkdir -p gpurun_out/r5g
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5g/pytest_gpu.log 2>&1
