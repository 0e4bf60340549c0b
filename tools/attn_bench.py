"""Times the fused attention kernels on cold, rotating inputs at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr, BF16
B, R, A, Dv = int(os.environ.get('B', 256)), int(os.environ.get('R', 196)), int(os.environ.get('A', 1200)), 2048
n_rot = 3
dt = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
Ps = [(torch.randn(B * R, A, device='cuda', generator=g)).to(dt) for _ in range(n_rot)]
Vs = [torch.randn(B * R, Dv, device='cuda', generator=g).clamp_min(0).to(dt) for _ in range(n_rot)]
qa = torch.randn(B, A, device='cuda', generator=g); wa = torch.randn(A, device='cuda', generator=g) * 0.03
c = torch.zeros(1, device='cuda'); inv = torch.rand(B * R, device='cuda', generator=g) * 0.05
p = torch.zeros(B, R, device='cuda'); pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
dpool = torch.randn(B, Dv, device='cuda', generator=g); dP = torch.zeros(B * R, A, device='cuda', dtype=dt)
dqa = torch.zeros(B, A, device='cuda'); dqaT = torch.zeros(B, A, device='cuda', dtype=dt)
dwa = torch.zeros(B, A, device='cuda'); dc = torch.zeros(B, device='cuda')
def fwd(i):
    call('vqa_attn_fwd', ptr(Ps[i % n_rot]), A, ptr(Vs[i % n_rot]), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled), None, Dv, B, R, A, Dv, BF16, None, 0, 0.0)
def bwd(i):
    call('vqa_attn_bwd', ptr(Ps[i % n_rot]), A, ptr(Vs[i % n_rot]), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dpool), Dv, ptr(dP), A, ptr(dqa), ptr(dqaT), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
NS = int(os.environ.get('NS', 0)) or call('vqa_attn_splits', B, R, A, Dv, BF16)
if NS > 1:
    wsf = torch.zeros(call('vqa_attn_split_workspace_floats', B, R, Dv, NS), device='cuda')
    slabs = torch.zeros(NS, B, A, device='cuda'); dwa_s = torch.zeros(B * NS, A, device='cuda'); dc_s = torch.zeros(B * NS, device='cuda')
def fwd_split(i):
    call('vqa_attn_fwd_split', ptr(Ps[i % n_rot]), A, ptr(Vs[i % n_rot]), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled), None, Dv, B, R, A, Dv, BF16, None, 0, 0.0, NS, ptr(wsf))
def bwd_split(i):
    call('vqa_attn_bwd_split', ptr(Ps[i % n_rot]), A, ptr(Vs[i % n_rot]), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dpool), Dv, ptr(dP), A, ptr(slabs), A, ptr(dwa_s), ptr(dc_s), B, R, A, Dv, BF16, NS, ptr(wsf))
    call('vqa_epilogue', ptr(slabs), NS, B * A, B, A, None, None, None, 0, 0, ptr(dqa), A, ptr(dqaT), A, BF16, 0, 0.0)
def inv_norm(i):
    call('vqa_inv_norm', ptr(Vs[i % n_rot]), ptr(inv), B * R, Dv, Dv, BF16)
FB, BB = B * (2 * R * A + 2 * R * Dv + 2 * A + 8 * R + 2 * Dv), B * (2 * R * Dv + 4 * R * A + 8 * R + 2 * (Dv + 2 * A))
cases = [('attn_fwd', fwd, FB), ('attn_bwd', bwd, BB)]
if NS > 1:
    cases += [(f'attn_fwd_split x{NS} (+ combine launch)', fwd_split, FB), (f'attn_bwd_split x{NS} (2 launches + dqa reduce)', bwd_split, BB)]
cases += [('inv_norm', inv_norm, B * R * (2 * Dv + 4))]
print(f'B = {B}, R = {R}, A = {A}')
for name, fn, nbytes in cases:
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    evs = []
    for i in range(12):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(i); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) for a, b_ in evs)
    med = ts[len(ts) // 2] * 1e3
    print(f'{name}: median {med:.1f} us  min {ts[0]*1e3:.1f} us  -> {nbytes / med / 1e3:.0f} GB/s ({nbytes / med / 1e3 / 80:.1f}% of 8 TB/s)')
