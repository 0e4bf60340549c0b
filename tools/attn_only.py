"""Runs only the fused attention forward / backward kernels at the bench shape (for PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr, BF16

B, R, A, Dv = 256, 196, 1200, 2048
n_rot = 3
dt = torch.bfloat16
g = torch.Generator(device='cuda').manual_seed(0)
Ps = [(torch.randn(B * R, A, device='cuda', generator=g)).to(dt) for _ in range(n_rot)]
Vs = [torch.randn(B * R, Dv, device='cuda', generator=g).clamp_min(0).to(dt) for _ in range(n_rot)]
qa = torch.randn(B, A, device='cuda', generator=g)
wa = torch.randn(A, device='cuda', generator=g) * 0.03
c = torch.zeros(1, device='cuda')
inv = torch.rand(B * R, device='cuda', generator=g) * 0.05
p = torch.zeros(B, R, device='cuda'); pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
pooled_d = torch.zeros(B, Dv, device='cuda', dtype=dt)
dpool = torch.randn(B, Dv, device='cuda', generator=g)
dP = torch.zeros(B * R, A, device='cuda', dtype=dt)
dqa = torch.zeros(B, A, device='cuda'); dqaT = torch.zeros(B, A, device='cuda', dtype=dt)
dwa = torch.zeros(B, A, device='cuda'); dc = torch.zeros(B, device='cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for i in range(n):
    P, V = Ps[i % n_rot], Vs[i % n_rot]
    call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled), None, Dv, B, R, A, Dv, BF16, ptr(pooled_d), 1234, 0.5)
    call('vqa_attn_bwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dpool), Dv, ptr(dP), A, ptr(dqa), ptr(dqaT), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
torch.cuda.synchronize()
print('done')
