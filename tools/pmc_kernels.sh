#!/bin/bash
# Per-kernel PMC evidence of the default bench.py command (the B=64 bf16 CREMA-D step): separate --pmc passes
# (FETCH_SIZE; WRITE_SIZE; MFMA busy + SQ busy) over `python3 bench.py`, folded by tools/pmc_kernels.py into
# profiles/rNN_pmc_kernels.json (NN = $GDL_ROUND, default 03) (per kernel: launches/step, duration, HBM read / write bytes per launch, MFMA busy
# fraction; whole-step totals; hash of the kernel sources).  Run on the GPU box from the repo root:
#     bash tools/pmc_kernels.sh [extra bench.py flags]
# (rocprofv3 gets the python program directly behind `--`; --pmc is never combined with other trace domains)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_kernels
rm -rf "$OUT"; mkdir -p "$OUT"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$tag" -o pmc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof --no-f32 --no-extra --no-comparator "$@" > "$OUT/$tag.log" 2>&1
done
python3 tools/pmc_kernels.py "$OUT" "$@"
rm -rf "$OUT"/*/
