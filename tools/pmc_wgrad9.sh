#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of conv_wgrad9_kernel at the eight 3x3
# stride-1 shapes of the CREMA-D B=64 step -> gpurun_out/pmc_wgrad9.txt
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_w9
mkdir -p "$OUT"
: > $PWD/gpurun_out/pmc_wgrad9.txt
for shape in 64,64,65,47,64,3,1,1 64,128,33,24,128,3,1,1 64,256,17,12,256,3,1,1 64,512,9,6,512,3,1,1 \
             192,64,56,56,64,3,1,1 192,128,28,28,128,3,1,1 192,256,14,14,256,3,1,1 192,512,7,7,512,3,1,1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/p"
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/p" -o pmc -- python3 tools/bench_one.py --op wgrad --shape $shape --iters 3 > "$OUT/log.txt" 2>&1
    python3 - "$OUT/p" $shape $c >> $PWD/gpurun_out/pmc_wgrad9.txt <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wgrad9_kernel" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[3]:
            vals.append(float(r["Counter_Value"]))
print(sys.argv[2], sys.argv[3], sum(vals) / max(len(vals), 1), len(vals))
PY
  done
done
rm -rf "$OUT"
cat $PWD/gpurun_out/pmc_wgrad9.txt
