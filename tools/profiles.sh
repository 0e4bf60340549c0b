# Profile evidence for profiles/ (file prefix: first argument, e.g. r02): default bench, kernel trace of the same command, PMC passes for the two MFMA-bound products,
# cfg5 kernel trace.  Summaries are written to gpurun_out/profiles_out/ (the only directory that travels back from the
# GPU box); copy them into profiles/ afterwards.
set -e
TAG=${1:?usage: profiles.sh <prefix>}
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/profiles_out
P=gpurun_out/profiles_out
timeout -k 10 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
grep '^{' gpurun_out/bench_final.json > $P/${TAG}_bench_line.json
bash tools/trace.sh --steps 100 --warmup 20 > /dev/null 2>&1
python tools/prof_summary.py gpurun_out/ktrace 120 > $P/${TAG}_bench_kernel_stats.txt
python tools/step_timeline.py gpurun_out/ktrace > $P/${TAG}_step_timeline.txt
bash tools/pmc_gemm.sh ${TAG}_k4fused k4fused 0 > /dev/null
cp gpurun_out/pmc_${TAG}_k4fused/summary.txt $P/${TAG}_pmc_projection_gemm.txt
bash tools/pmc_gemm.sh ${TAG}_wgrad wgrad_swapped 0 > /dev/null
cp gpurun_out/pmc_${TAG}_wgrad/summary.txt $P/${TAG}_pmc_wgrad_gemm.txt
bash tools/pmc_attn.sh > /dev/null 2>&1
cd $R
python tools/attn_traffic_json.py $P/attn_fwd_traffic.json > /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktrace_c5
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktrace_c5 -- python3 $R/bench.py --no-cpu-baseline --no-unfolded --batch 512 --regions 576 --hops 2 --steps 20 --warmup 5 > $R/gpurun_out/bench_cfg5.json 2>/dev/null
cd $R
python tools/prof_summary.py gpurun_out/ktrace_c5 25 > $P/${TAG}_cfg5_kernel_stats.txt
grep '^{' gpurun_out/bench_cfg5.json > $P/${TAG}_cfg5_bench_line.json
