#!/bin/bash
mkdir -p gpurun_out/r5h
python3 -m pytest tests/test_swin_ops_gpu.py -x -q -s -k "config5_size" > gpurun_out/r5h/pytest_attn.log 2>&1
python3 -m pytest tests/test_step_gpu.py -x -q -s -k "film_mirror" > gpurun_out/r5h/pytest_film.log 2>&1
python3 -m pytest tests/test_swin_gpu.py -x -q -s -k "config5" > gpurun_out/r5h/pytest_swin5.log 2>&1
