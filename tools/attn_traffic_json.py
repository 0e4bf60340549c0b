"""Turns the two PMC passes of tools/pmc_attn.sh (gpurun_out/pmc_fetch, gpurun_out/pmc_write) into
profiles/attn_fwd_traffic.json, the HBM traffic per launch that bench.py reports as roofline.traffic.
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16-byte-per-lane streams
as 64 bytes, so it is doubled; WRITE_SIZE is taken as read.
usage: python tools/attn_traffic_json.py [out.json]"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, R, A, Dv = 256, 196, 1200, 2048


def avg(d, counter, kernel):
    vals = []
    csv.field_size_limit(1 << 30)
    for f in glob.glob(os.path.join(ROOT, 'gpurun_out', d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r['Kernel_Name'] and r['Counter_Name'] == counter:
                vals.append(float(r['Counter_Value']))
    if not vals:
        raise SystemExit(f'no {counter} rows for {kernel} under gpurun_out/{d}')
    return sum(vals) / len(vals), len(vals)


def entry(kernel, label, alg):
    f, nf = avg('pmc_fetch', 'FETCH_SIZE', kernel)
    w, nw = avg('pmc_write', 'WRITE_SIZE', kernel)
    return {'kernel': label, 'launches_averaged': [nf, nw], 'FETCH_SIZE_KiB_avg': f, 'WRITE_SIZE_KiB_avg': w,
            'hbm_bytes_per_launch': int(round((2 * f + w) * 1024)), 'algorithmic_bytes_per_launch': alg}


fwd = entry('k_attn_fwd_online', 'k_attn_fwd_online<3> (bf16)', B * (2 * R * A + 2 * R * Dv + 2 * A + 8 * R + 2 * Dv))
bwd = entry('k_attn_bwd_pipe', 'k_attn_bwd_pipe<4> (bf16)', B * (2 * R * Dv + 4 * R * A + 8 * R + 2 * (Dv + 2 * A)))
out = {'shape': {'B': B, 'R': R, 'A': A, 'Dv': Dv},
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of tools/attn_only.py (9 launches, 3 input '
                 'sets in rotation, the fused dropout output on); counters are in KiB; FETCH_SIZE doubled (gfx950 counts '
                 '128-B requests of 16-B-per-lane streams at 64 B), WRITE_SIZE as read'}
out.update(fwd)
out['backward'] = bwd
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'attn_fwd_traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out))
