"""Per-kernel timings of the step at the benchmark shapes (development aid, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd.model import VQAAttnModel
from vqa_amd import _lib
from vqa_amd._lib import call, ptr

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us

def main():
    B = int(os.environ.get('B', 256)); R = int(os.environ.get('R', 196)); L = 14; C = 3000
    dt = torch.bfloat16
    m = VQAAttnModel(['w'] * 10000, ['a'] * C, None, dtype=dt).cuda()
    g = torch.Generator(device='cuda').manual_seed(0)
    feats = torch.randn(B, R, 2048, device='cuda', generator=g).clamp_min(0).to(dt)
    tok = torch.randint(1, 10001, (B, L), device='cuda', generator=g)
    lens = torch.randint(3, L + 1, (B,), device='cuda', generator=g)
    tok = torch.where(torch.arange(L, device='cuda')[None] < (lens[:, None] - 1), tok, torch.zeros_like(tok))
    ans = torch.randint(0, C, (B,), device='cuda', generator=g)
    m.train()
    for _ in range(2): m.train_step(feats, tok, ans)
    torch.cuda.synchronize()
    ws = m._ctx[0]
    A, Dv, H = m.A, m.Dv, m.Hq
    res = {}
    res['train_step'] = timeit(lambda: m.train_step(feats, tok, ans), n=10)
    res['encode_question'] = timeit(lambda: m.encode_question(tok, ws), n=10)
    res['inv_norm'] = timeit(lambda: call('vqa_inv_norm', ptr(feats), ptr(ws['inv_norm']), B * R, Dv, Dv, m.dt))
    pre = 'attention.0.'
    res['gemm_P(K4)'] = timeit(lambda: call('vqa_gemm_fused', ptr(feats), Dv, 1, ptr(m._w(pre + 'linear_va.weight')), Dv, 1, ptr(ws['P0']), A, B * R, A, Dv, ptr(m._f32(pre + 'linear_va.bias')), ptr(ws['inv_norm']), 0, m.dt, m.dt, 0))
    res['attn_fwd'] = timeit(lambda: call('vqa_attn_fwd', ptr(ws['P0']), A, ptr(feats), Dv, ptr(ws['qa0']), A, ptr(m._f32(pre + 'linear_a.weight')), ptr(m._f32(pre + 'linear_a.bias')), ptr(ws['inv_norm']), ptr(ws['p0']), ptr(ws['pooled0']), None, Dv, B, R, A, Dv, m.dt, None, 0, 0.0))
    res['attn_bwd'] = timeit(lambda: call('vqa_attn_bwd', ptr(ws['P0']), A, ptr(feats), Dv, ptr(ws['qa0']), A, ptr(m._f32(pre + 'linear_a.weight')), ptr(ws['inv_norm']), ptr(ws['p0']), ptr(ws['d_pooled']), Dv, ptr(ws['dP']), A, ptr(ws['dqa32']), ptr(ws['dqaT']), A, ptr(ws['dwa_part']), ptr(ws['dc_part']), B, R, A, Dv, m.dt))
    res['wgrad_Wva'] = timeit(lambda: m._wgrad(ws['dP'], A, feats, Dv, pre + 'linear_va.weight', A, Dv, B * R))
    res['lstm_step_gemm'] = timeit(lambda: m._gemm_slabs(ws['h'], H, 1, m._whh_t, H, 1, B, 4 * H, H))
    res['linear_v'] = timeit(lambda: m._linear(ws['pooled_d'], Dv, 'linear_v.weight', 'linear_v.bias', B, m.F, Dv, act=1, out_pre=ws['xv'], ld_pre=m.F))
    res['adam'] = timeit(lambda: m.adam_step(1e-5))
    def fwd():
        with torch.no_grad(): m._forward_impl(feats, tok, None)
    res['forward'] = timeit(fwd, n=10)
    def bwd():
        with torch.no_grad(): m._backward_impl()
    res['backward'] = timeit(bwd, n=10)
    for k, v in res.items(): print(f'{k:20s} {v:10.1f} us')
    fl = 2.0 * B * R * Dv * A
    print('K4 TFLOP/s', fl / res['gemm_P(K4)'] / 1e6, ' wgrad TFLOP/s', fl / res['wgrad_Wva'] / 1e6)
    bytes_f = B * (2 * R * A + 2 * R * Dv + 2 * A + 8 * R + 2 * Dv)
    bytes_b = B * (2 * R * Dv + 4 * R * A + 8 * R + 2 * (Dv + 2 * A))
    print('attn fwd GB/s', bytes_f / res['attn_fwd'] / 1e3, ' bwd GB/s', bytes_b / res['attn_bwd'] / 1e3)
    print('samples/s', B / res['train_step'] * 1e6)

if __name__ == '__main__':
    main()
