# Same-box A/B: current library against vqa_amd/libvqa_hip_prev.so (three alternations of the default bench).
set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/ab_new_$i.json
  VQA_HIP_LIB=$GRAFT_REPO_ROOT/vqa_amd/libvqa_hip_prev.so timeout -k 10 120 python bench.py --no-cpu-baseline > gpurun_out/ab_old_$i.json
done
python - <<'PY'
import json
for k in ('new','old'):
    for i in (1,2,3):
        d=json.load(open(f'gpurun_out/ab_{k}_{i}.json'))
        print(k,i,round(d['value']),round(d['ms_per_step'],4),round(d['roofline']['frac'],4),round(d['roofline']['avg_launch_ms']*1e3,2))
PY
