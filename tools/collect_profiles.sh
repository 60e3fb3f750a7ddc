mkdir -p gpurun_out/fin
bash tools/pmc_kernels.sh > gpurun_out/fin/pmc_kernels.log 2>&1
cp gpurun_out/r02_pmc_kernels.json profiles/r02_pmc_kernels.json 2>/dev/null
bash tools/rocprof_bench.sh r02 > gpurun_out/fin/rocprof.log 2>&1
python bench.py > gpurun_out/fin/bench_default.json 2> gpurun_out/fin/bench_default.err
python bench.py --workload ks --no-cpu-baseline > gpurun_out/fin/bench_ks.json 2> gpurun_out/fin/bench_ks.err
python bench.py --workload vggsound_swin --no-cpu-baseline > gpurun_out/fin/bench_swin.json 2> gpurun_out/fin/bench_swin.err
