# everything under profiles/ for a round in one GPU call (run from the repo root on the GPU box): PMC passes, kernel-trace
# statistics, the default / ks / vggsound_swin bench lines.  R = round tag (default 03); results land in gpurun_out/fin and
# gpurun_out/ -- copy what is to be judged into profiles/ afterwards.
R=${GDL_ROUND:-03}
export GDL_ROUND=$R
mkdir -p gpurun_out/fin
bash tools/pmc_kernels.sh > gpurun_out/fin/pmc_kernels.log 2>&1
cp gpurun_out/r${R}_pmc_kernels.json profiles/r${R}_pmc_kernels.json 2>/dev/null
bash tools/rocprof_bench.sh r$R > gpurun_out/fin/rocprof.log 2>&1
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
python bench.py --workload ks --no-cpu-baseline --no-comparator > gpurun_out/fin/bench_ks.json 2> gpurun_out/fin/bench_ks.err
python bench.py --workload vggsound_swin --no-cpu-baseline > gpurun_out/fin/bench_swin.json 2> gpurun_out/fin/bench_swin.err
