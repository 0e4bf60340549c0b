# Kernel trace of the bench (extra bench arguments pass through); output under gpurun_out/ktrace.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/ktrace
cd $R
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ktrace -- python3 $R/bench.py --no-cpu-baseline --no-unfolded "$@" > $R/gpurun_out/bench_under_rocprof.json
