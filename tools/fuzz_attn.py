"""Random shapes through the fused attention entry points against fp64 torch (development aid, GPU box): the same
checks as tests/test_gpu_attention.py on shapes drawn around the kernels' chunk and lane-slot boundaries."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr, BF16
from tests.test_gpu_attention import _ref

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dt = torch.bfloat16
for case in range(n_cases):
    B = rnd.choice([1, 2, 3, 5])
    R = rnd.choice([1, 2, 15, 16, 17, 31, 32, 33, 196, 197, 576, rnd.randint(1, 700)])
    A = 8 * rnd.choice([1, 3, 8, 63, 64, 65, 127, 128, 129, 150, 192, 193, 256, rnd.randint(1, 256)])
    Dv = 8 * rnd.choice([1, 9, 64, 128, 255, 256, 257, 512, rnd.randint(1, 520)])
    g = torch.Generator(device='cuda').manual_seed(case)
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    qa = torch.randn(B, A, device='cuda', generator=g)
    wa = torch.randn(A, device='cuda', generator=g) * (2.0 / A ** 0.5)
    c = torch.full((1,), 0.25, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    dpool = torch.randn(B, Dv, device='cuda', generator=g)
    p = torch.full((B, R), -1.0, device='cuda')
    pooled32 = torch.full((B, Dv), -1.0, device='cuda')
    pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
    call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled),
         ptr(pooled32), Dv, B, R, A, Dv, BF16, None, 0, 0.0)
    torch.cuda.synchronize()
    Pd = P.double().requires_grad_(True); qad = qa.double().requires_grad_(True)
    wad = wa.double().requires_grad_(True); cd = c.double().requires_grad_(True)
    p_ref, pooled_ref = _ref(Pd, V, qad, wad, cd, inv)
    tag = f'B={B} R={R} A={A} Dv={Dv}'
    e1 = float((p.double() - p_ref).abs().max())
    e2 = float((pooled32.double() - pooled_ref).abs().max() / pooled_ref.abs().max().clamp_min(1e-30))
    if e1 > 2e-5 * max(1.0, float(p_ref.max()) * 10) or e2 > 2e-5 or not torch.equal(pooled, pooled32.to(dt)):
        print('MISMATCH fwd', tag, e1, e2, flush=True); break
    (pooled_ref * dpool.double()).sum().backward()
    pin = p_ref.detach().float().contiguous()
    dP = torch.zeros(B, R, A, device='cuda', dtype=dt)
    dqa = torch.zeros(B, A, device='cuda'); dqa_t = torch.zeros(B, A, device='cuda', dtype=dt)
    dwa = torch.zeros(B, A, device='cuda'); dc = torch.zeros(B, device='cuda')
    call('vqa_attn_bwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(pin), ptr(dpool), Dv, ptr(dP), A,
         ptr(dqa), ptr(dqa_t), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
    torch.cuda.synchronize()
    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    if R > 1:
        errs = (rel(dP, Pd.grad * inv.double()[:, :, None]), rel(dqa, qad.grad), rel(dwa.sum(0), wad.grad))
        if errs[0] > 1e-2 or errs[1] > 1e-4 or errs[2] > 1e-4 or not torch.equal(dqa_t, dqa.to(dt)):
            print('MISMATCH bwd', tag, errs, flush=True); break
    # the same sample's rows shared by nsplit workgroups (the small-batch form): against the one-workgroup results
    max_ns = min(8, (R + 15) // 16, R)
    if max_ns >= 2 and Dv // 8 <= 256 and call('vqa_attn_splits', 1, max(R, 64), A, Dv, BF16) >= 1 and (A // 8 + 63) // 64 <= 4:
        ns = rnd.randint(2, max_ns)
        ws = torch.full((call('vqa_attn_split_workspace_floats', B, R, Dv, ns),), float('nan'), device='cuda')
        p2 = torch.full((B, R), -1.0, device='cuda'); pooled32_2 = torch.full((B, Dv), -1.0, device='cuda')
        pooled_2 = torch.zeros(B, Dv, device='cuda', dtype=dt)
        call('vqa_attn_fwd_split', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p2), ptr(pooled_2),
             ptr(pooled32_2), Dv, B, R, A, Dv, BF16, None, 0, 0.0, ns, ptr(ws))
        torch.cuda.synchronize()
        f1 = float((p2.double() - p_ref).abs().max())
        f2 = float((pooled32_2.double() - pooled_ref).abs().max() / pooled_ref.abs().max().clamp_min(1e-30))
        if f1 > 2e-5 * max(1.0, float(p_ref.max()) * 10) or f2 > 2e-5 or not torch.equal(pooled_2, pooled32_2.to(dt)):
            print('MISMATCH fwd_split', tag, ns, f1, f2, flush=True); break
        dP2 = torch.zeros(B, R, A, device='cuda', dtype=dt)
        slabs = torch.full((ns, B, A), float('nan'), device='cuda')
        dwa2 = torch.full((B * ns, A), float('nan'), device='cuda'); dc2 = torch.full((B * ns,), float('nan'), device='cuda')
        call('vqa_attn_bwd_split', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(pin), ptr(dpool), Dv, ptr(dP2), A,
             ptr(slabs), A, ptr(dwa2), ptr(dc2), B, R, A, Dv, BF16, ns, ptr(ws))
        torch.cuda.synchronize()
        if R > 1:
            errs = (rel(dP2, Pd.grad * inv.double()[:, :, None]), rel(slabs.sum(0), qad.grad), rel(dwa2.sum(0), wad.grad),
                    float((dc2.sum() - dc.sum()).abs()))
            if errs[0] > 1e-2 or errs[1] > 1e-4 or errs[2] > 1e-4 or errs[3] > 1e-4 * max(1.0, float(dc.abs().sum())):
                print('MISMATCH bwd_split', tag, ns, errs, flush=True); break
    if case % 20 == 0:
        print('ok', case, tag, f'{e1:.1e} {e2:.1e}', flush=True)
else:
    print('done', n_cases, 'cases, no mismatch')
