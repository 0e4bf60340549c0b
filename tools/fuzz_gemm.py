"""Random shapes through the GEMM entry points against fp64 torch (development aid, GPU box).  Output buffers are
sized from the same M, N the calls receive; every result is checked, the first mismatch stops the run."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr, BF16, F32
from tests.test_gpu_gemm import make

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for case in range(n_cases):
    M = rnd.choice([1, 7, 16, 255, 256, 257, 511, 513, 1000, 2049, rnd.randint(1, 3000)])
    N = rnd.choice([1, 8, 15, 64, 176, 255, 256, 257, 1200, 1281, rnd.randint(1, 1600)])
    K = rnd.choice([8, 64, 72, 128, 192, 320, 1000, 2048, 8 * rnd.randint(1, 300)])
    akm, bkm = rnd.choice([(1, 1), (1, 0), (0, 1), (0, 0)])
    eng = rnd.choice([0, 4, 6, 9, 10, 11])
    sk = rnd.choice([1, 1, 2, 3, 5])
    As, lda, Bs, ldb, ref = make(M, N, K, akm, bkm, torch.bfloat16, case)
    scale = float(ref.abs().max()) or 1.0
    slab = torch.full((sk, M, N), float('nan'), device='cuda')
    call('vqa_gemm_slabs', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(slab), M * N, N, M, N, K, sk, BF16, eng)
    torch.cuda.synchronize()
    err = float((slab.double().sum(0) - ref).abs().max()) / scale
    ok = err <= 3e-5
    tag = f'slabs M={M} N={N} K={K} akm={akm} bkm={bkm} eng={eng} sk={sk}'
    if ok and akm and bkm:
        g = torch.Generator(device='cuda').manual_seed(case)
        bias = torch.randn(N, device='cuda', generator=g)
        rs = torch.rand(M, device='cuda', generator=g) * 0.2
        ldc = (N + 7) // 8 * 8
        for odt in (torch.bfloat16, torch.float32):
            C = torch.zeros(M, ldc, device='cuda', dtype=odt)
            call('vqa_gemm_fused', ptr(As), lda, 1, ptr(Bs), ldb, 1, ptr(C), ldc, M, N, K, ptr(bias), ptr(rs), 0,
                 BF16 if odt == torch.bfloat16 else F32, BF16, eng)
            torch.cuda.synchronize()
            want = ref * rs.double()[:, None] + bias.double()[None, :]
            e2 = float((C[:, :N].double() - want).abs().max() / want.abs().max())
            if e2 > (1e-2 if odt == torch.bfloat16 else 3e-5) or (ldc > N and float(C[:, N:].abs().max()) != 0.0):
                ok = False; tag = f'fused {odt} M={M} N={N} K={K} eng={eng}'; err = e2
        if ok:
            C = torch.zeros(M, ldc, device='cuda', dtype=torch.bfloat16)
            inv = torch.full((M,), float('nan'), device='cuda')
            call('vqa_gemm_fused_rownorm', ptr(As), lda, ptr(Bs), ldb, ptr(C), ldc, M, N, K, ptr(bias), ptr(inv), 0, BF16, BF16, 0)
            torch.cuda.synchronize()
            inv_ref = 1.0 / As[:, :K].double().norm(dim=1).clamp_min(1e-12)
            want = ref * inv_ref[:, None] + bias.double()[None, :]
            e3 = float((C[:, :N].double() - want).abs().max() / want.abs().max())
            e4 = float(((inv.double() - inv_ref).abs() / inv_ref).max())
            if e3 > 1e-2 or e4 > 1e-5:
                ok = False; tag = f'rownorm M={M} N={N} K={K}'; err = max(e3, e4)
    if not ok:
        bad += 1
        print('MISMATCH', tag, err, flush=True)
        break
    if case % 25 == 0:
        print('ok', case, tag, f'{err:.2e}', flush=True)
print('done', n_cases, 'cases, mismatches', bad)
