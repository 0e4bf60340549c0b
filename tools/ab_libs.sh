# Same-box A/B of library builds: bash tools/ab_libs.sh <rounds> <name> [<name> ...]  (name "prod" = vqa_amd/libvqa_hip.so,
# otherwise vqa_amd/libvqa_hip_<name>.so); every round runs the default bench once per library, interleaved.
cd $GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for name in "$@"; do
    lib=$GRAFT_REPO_ROOT/vqa_amd/libvqa_hip_$name.so
    [ "$name" = prod ] && lib=$GRAFT_REPO_ROOT/vqa_amd/libvqa_hip.so
    out=$(VQA_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-unfolded $BENCH_FLAGS 2>/dev/null | tail -1) || exit 1
    echo "$name round $r: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); g=d["roofline_gemm"]; print(round(d["value"]), "samples/s", round(d["ms_per_step"]*1000,1), "us/step  proj", round(g["proj"]["avg_launch_ms"]*1000,1), " wgrad", round(g["wgrad"]["avg_launch_ms"]*1000,1), " attn_fwd", round(d["roofline"]["avg_launch_ms"]*1000,1), " loss", d["loss"])')"
  done
done
