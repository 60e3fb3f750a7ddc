#!/bin/bash
# Collects the rocprofv3 kernel-trace statistics of the default bench.py run (on the GPU box).
# usage: tools/rocprof_bench.sh <tag>   -> gpurun_out/prof_<tag>/ ; copies *_kernel_stats.csv next to it
set -u
TAG=${1:-r01}
shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32 --no-extra --no-comparator "$@" > "$OUT/bench_stdout.log" 2>&1
for f in $(find "$OUT" -name "*kernel_stats.csv"); do cp "$f" "$PWD/gpurun_out/prof_${TAG}_kernel_stats.csv"; done
# keep the per-launch trace small enough to merge back: gzip it
for f in $(find "$OUT" -name "*kernel_trace.csv"); do gzip -f "$f"; done
ls -la $(find "$OUT" -type f) | head
tail -2 "$OUT/bench_stdout.log"
