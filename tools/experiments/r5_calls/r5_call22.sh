#!/bin/bash
O=gpurun_out/r5u; mkdir -p $O
bash tools/ab_env.sh 2 100 X=1 GDL_WGRAD_ROUND8=0 GDL_WGRAD9_MINST=24 GDL_WGRAD9_MINST=12 > $O/ab_default.txt 2>&1
cat > /tmp/ab_swin.sh <<'XX'
export GDL_TUNING=1
for r in 1 2; do for s in X=1 GDL_WGRAD_ROUND8=0; do
out=$(env $s python3 bench.py --workload vggsound_swin --steps 40 --warmup 10 --no-cpu-baseline --no-f32 --no-prof --no-extra --no-comparator 2>/dev/null | python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
echo "round $r  $s  $out"; done; done
XX
bash /tmp/ab_swin.sh > $O/ab_swin.txt 2>&1
