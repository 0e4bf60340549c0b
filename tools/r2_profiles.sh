# Round-2 evidence: default bench, kernel trace of the same command, PMC passes for the two MFMA-bound products,
# cfg5 kernel trace.  Summaries are written to gpurun_out/r2_profiles/ (the only directory that travels back from the
# GPU box); copy them into profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2_profiles
P=gpurun_out/r2_profiles
timeout -k 10 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
grep '^{' gpurun_out/bench_final.json > $P/r02_bench_line.json
bash tools/trace.sh --steps 100 --warmup 20 > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/ktrace 120 > $P/r02_bench_kernel_stats.txt
bash tools/pmc_gemm.sh r2_k4fused k4fused 0 > /dev/null
cp gpurun_out/pmc_r2_k4fused/summary.txt $P/r02_pmc_projection_gemm.txt
bash tools/pmc_gemm.sh r2_wgrad wgrad_swapped 0 > /dev/null
cp gpurun_out/pmc_r2_wgrad/summary.txt $P/r02_pmc_wgrad_gemm.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktrace_c5
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktrace_c5 -- python3 $R/bench.py --no-cpu-baseline --no-unfolded --batch 512 --regions 576 --hops 2 --steps 20 --warmup 5 > $R/gpurun_out/bench_cfg5.json 2>/dev/null
cd $R
python tools/prof_summary.py gpurun_out/ktrace_c5 25 > $P/r02_cfg5_kernel_stats.txt
grep '^{' gpurun_out/bench_cfg5.json > $P/r02_cfg5_bench_line.json
