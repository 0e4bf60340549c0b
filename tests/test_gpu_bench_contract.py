"""bench.py's result line: one JSON object on stdout with the fields the driver reads (metric, value, unit, n_gpus,
steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload), the roofline of the
fused attention forward measured inside the timed region, the two MFMA-bound products, and the CPU baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '8', '--warmup', '2'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['metric'].startswith('train samples/sec') and d['unit'] == 'samples/s'
    assert d['n_gpus'] == 1 and d['steps'] == 8 and d['warmup'] == 2
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] > 0 and abs(d['value'] - 256 / (d['ms_per_step'] * 1e-3)) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and r['timed_launches'] >= 2
    assert abs(r['frac'] - r['achieved'] / r['peak']) <= 1e-9 and 0.3 < r['frac'] < 1.0
    assert r['algorithmic_bytes_per_launch'] == 256 * (2 * 196 * 1200 + 2 * 196 * 2048 + 2 * 1200 + 8 * 196 + 2 * 2048)
    for key in ('proj', 'wgrad'):
        g = d['roofline_gemm'][key]
        assert g['unit'] == 'TFLOP/s' and g['peak'] == 2500 and g['timed_launches'] >= 2 and 0.1 < g['frac'] < 1.0
        assert g['flop_per_launch'] == 2.0 * 256 * 196 * 2048 * 1200
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == 'samples/s' and c['value'] > 0 and c['cores'] >= 1 and c['sample']
    assert 7.5 < d['loss'] < 8.5                      # ln 3000 = 8.006 at initialisation
    assert d['value_unfolded'] > 0


def test_bench_multi_gpu_block_over_one_rank_rccl():
    """The diagnostics the first multi-GPU run will carry (ranks seen, per-rank kernel timings, collectives and bytes
    per step, exposed wait, RCCL version), produced here by the one-rank RCCL mode."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '8', '--warmup', '2',
                          '--rccl-single', '--no-cpu-baseline', '--no-unfolded'],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    m = d['multi_gpu']
    assert 'diagnostics_error' not in m, m
    assert len(m['rccl_ranks']) == 1 and m['rccl_ranks'][0]['rank'] == 0
    k = m['rccl_ranks'][0]['kernel_ms']
    assert k['attn_fwd'] > 0 and k['proj'] > 0 and k['wgrad'] > 0
    assert m['kernel_ms_max_over_ranks']['wgrad'] == k['wgrad']
    assert m['rccl']['version'].count('.') >= 1
    c = m['communication']
    assert c['collectives_per_step'] == 2 and c['allreduce_bytes_per_step'] > 40e6 and c['exposed_wait_ms_per_step'] >= 0
