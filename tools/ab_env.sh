# same-box A/B of bench.py under different environments, with the timed kernels' launch times:
# bash tools/ab_env.sh <file>; each line of the file is "<label>|<env assignments>|<bench flags>"; two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  while IFS='|' read -r label envs flags; do
    [ -z "$label" ] && continue
    out=$(env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-unfolded --steps 100 --warmup 20 $flags 2>/dev/null | tail -1) || exit 1
    echo "$label round $round: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); g=d["roofline_gemm"]; print(round(d["value"]), "samples/s", round(d["ms_per_step"]*1000,1), "us/step  proj", round(g["proj"]["avg_launch_ms"]*1000,1), " wgrad", round(g["wgrad"]["avg_launch_ms"]*1000,1), " attn_fwd", round(d["roofline"]["avg_launch_ms"]*1000,1), " loss", d["loss"])')"
  done < $1
done
