# Same-box A/B at cfg5 (B = 512, R = 576, 2 hops): current library against vqa_amd/libvqa_hip_prev.so.
set -e
cd $GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-unfolded --batch 512 --regions 576 --hops 2 --steps 20 --warmup 5"
for i in 1 2 3; do
  timeout -k 10 120 python bench.py $ARGS > gpurun_out/ab5_new_$i.json
  VQA_HIP_LIB=$GRAFT_REPO_ROOT/vqa_amd/libvqa_hip_prev.so timeout -k 10 120 python bench.py $ARGS > gpurun_out/ab5_old_$i.json
done
python - <<'PY'
import json
for k in ('new','old'):
    for i in (1,2,3):
        d=json.load(open(f'gpurun_out/ab5_{k}_{i}.json'))
        print(k,i,round(d['value']),round(d['ms_per_step'],4),round(d['roofline_gemm']['wgrad']['avg_launch_ms']*1e3,1))
PY
