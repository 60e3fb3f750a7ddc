#!/bin/bash
O=gpurun_out/r5aa; mkdir -p $O
bash tools/ab_env.sh 3 100 GDL_TAIL_GATE=0 X=1 > $O/ab_tail_gate.txt 2>&1
python3 tools/utilisation_timeline.py --launches --out $O/timeline.txt > /dev/null 2> $O/timeline.err
python3 -m pytest tests/test_step_gpu.py tests/test_bf16_parity_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
