# Final numbers of a round: the default bench (with the CPU baseline), a kernel trace of the same command,
# encoder-mode and cfg5 step times.  Outputs under gpurun_out/.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktrace
cd $R
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktrace -- python3 $R/bench.py --no-cpu-baseline --no-unfolded > $R/gpurun_out/bench_under_rocprof.json
timeout -k 10 200 python bench.py --no-cpu-baseline --no-unfolded --batch 512 --regions 576 --hops 2 --steps 20 --warmup 5 > gpurun_out/bench_cfg5.json
timeout -k 10 300 python tools/encoder_bench.py > gpurun_out/encoder_bench.txt 2>&1
