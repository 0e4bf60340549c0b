"""GEMM variant sweep at the step's shapes (development aid, GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd._lib import call, ptr

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def run_fused(name, M, N, K, engines, ldc=None):
    """the projection as the step calls it: bf16 output, bias and row scale in the epilogue"""
    dt = torch.bfloat16
    A = (torch.randn((M, K), device='cuda') * 0.5).clamp_min_(0).to(dt)
    Bm = (torch.randn((N, K), device='cuda') * 0.05).to(dt)
    bias = torch.randn(N, device='cuda'); rs = torch.rand(M, device='cuda')
    ldc = ldc or N
    C = torch.empty((M, ldc), device='cuda', dtype=dt)
    for eng in engines:
        us = timeit(lambda: call('vqa_gemm_fused', ptr(A), K, 1, ptr(Bm), K, 1, ptr(C), ldc, M, N, K, ptr(bias), ptr(rs), 0, 1, 1, eng))
        print(f'{name:12s} fused bf16 out ldc={ldc} M={M:6d} N={N:5d} K={K:6d} engine={eng:2d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF')

def run_rownorm(name, M, N, K, ldc, engines=(0,)):
    """projection with the row norms fused (the step's call) against row-scale from a separate pass"""
    dt = torch.bfloat16
    A = (torch.randn((M, K), device='cuda')).clamp_min_(0).to(dt)
    Bm = (torch.randn((N, K), device='cuda') * 0.02).to(dt)
    bias = torch.randn(N, device='cuda'); inv = torch.empty(M, device='cuda')
    C = torch.empty((M, ldc), device='cuda', dtype=dt)
    for eng in engines:
        us = timeit(lambda: call('vqa_gemm_fused_rownorm', ptr(A), K, ptr(Bm), K, ptr(C), ldc, M, N, K, ptr(bias), ptr(inv), 0, 1, 1, eng))
        print(f'{name:10s} fused row norms engine={eng} ldc={ldc}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF')
    for label, fn in (('fused row norms', lambda: call('vqa_gemm_fused_rownorm', ptr(A), K, ptr(Bm), K, ptr(C), ldc, M, N, K, ptr(bias), ptr(inv), 0, 1, 1, 0)),
                      ('row scale given', lambda: call('vqa_gemm_fused', ptr(A), K, 1, ptr(Bm), K, 1, ptr(C), ldc, M, N, K, ptr(bias), ptr(inv), 0, 1, 1, 0)),
                      ('inv_norm pass  ', lambda: call('vqa_inv_norm', ptr(A), ptr(inv), M, K, K, 1))):
        us = timeit(fn)
        print(f'{name:10s} {label} ldc={ldc}: {us:8.1f} us')

def run(name, M, N, K, akm, bkm, variants, ldb=None):
    dt = torch.bfloat16
    A = (torch.randn((M, K) if akm else (K, M), device='cuda') * 0.5).to(dt)
    if ldb and not bkm:
        Bm = (torch.randn((K, ldb), device='cuda') * 0.5).to(dt)
    else:
        Bm = (torch.randn((N, K) if bkm else (K, N), device='cuda') * 0.5).to(dt)
        ldb = K if bkm else N
    lda = K if akm else M
    slab = torch.empty(max(v[1] for v in variants) * M * N, device='cuda', dtype=torch.float32)
    for eng, sk in variants:
        us = timeit(lambda: call('vqa_gemm_slabs', ptr(A), lda, int(akm), ptr(Bm), ldb, int(bkm), ptr(slab), M * N, N, M, N, K, sk, 1, eng))
        print(f'{name:12s} M={M:6d} N={N:5d} K={K:6d} tile={ {2: "64", 3: "128", 4: "256x128", 6: "256x256", 8: "256x256r10", 9: "256x256r10pf", 10: "256x256pp", 11: "256x256sp"}[eng] } splitk={sk:2d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF')

def run_pitch():
    """Does the operand pitch matter (power-of-two row pitch = every row of a tile column on one memory channel)?"""
    dt = torch.bfloat16
    M, N, K = 50176, 1200, 2048
    Bm = (torch.randn((N, K), device='cuda') * 0.05).to(dt)
    slab = torch.empty(max(M * N, 6 * 2048 * 1200), device='cuda', dtype=torch.float32)   # fp32 output of either product
    for _ in range(3):
        for lda in (2048, 2048 + 64, 2048 + 128, 2048 + 512):
            A = (torch.randn((M, lda), device='cuda') * 0.5).clamp_min_(0).to(dt)
            us = timeit(lambda: call('vqa_gemm_slabs', ptr(A), lda, 1, ptr(Bm), K, 1, ptr(slab), M * N, N, M, N, K, 1, 1, 10))
            print(f'proj k-major lda={lda}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF')
            # weight gradient: A = V as [K=50176][lda] (m-major), B = dP as [K][1280]
            dP = (torch.randn((M, 1280), device='cuda') * 0.5).to(dt)
            us = timeit(lambda: call('vqa_gemm_slabs', ptr(A), lda, 0, ptr(dP), 1280, 0, ptr(slab), 2048 * 1200, 1200, 2048, 1200, M, 6, 1, 10))
            print(f'wgrad m-major lda={lda}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF')
            del A, dP


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'big'
    if which == 'big':
        for _ in range(2):
            run('K4 fwd NT', 50176, 1200, 2048, True, True, [(6, 1), (8, 1), (9, 1)])
            run('square', 8192, 8192, 8192, True, True, [(6, 1), (8, 1), (9, 1)])
            run('wgrad TN', 1200, 2048, 50176, False, False, [(6, 6), (8, 6), (9, 6), (4, 3), (6, 5), (6, 7)])
            run('wgrad TN swapped', 2048, 1200, 50176, False, False, [(6, 6), (8, 6), (4, 3)])
    elif which == 'pp':
        for _ in range(3):
            run('K4 fwd NT', 50176, 1200, 2048, True, True, [(9, 1), (10, 1)])
            run('sq4k', 4096, 4096, 4096, True, True, [(9, 1), (10, 1)])
            run('square', 8192, 8192, 8192, True, True, [(9, 1), (10, 1)])
            run('wgrad TN swapped', 2048, 1200, 50176, False, False, [(6, 6), (10, 6)])
    elif which == 'pp10':
        for _ in range(3):
            engs = [int(e) for e in (sys.argv[2] if len(sys.argv) > 2 else '10').split(',')]
            run_fused('K4 fwd', 50176, 1200, 2048, engs)
            run_fused('K4 fwd', 50176, 1200, 2048, engs, ldc=1280)
            run('K4 fwd NT', 50176, 1200, 2048, True, True, [(10, 1)])
            run('sq4k', 4096, 4096, 4096, True, True, [(10, 1)])
            run('square', 8192, 8192, 8192, True, True, [(10, 1)])
    elif which == 'sp':
        for _ in range(3):
            run('K4 fwd NT', 50176, 1200, 2048, True, True, [(10, 1), (11, 1)])
            run_fused('K4 fwd', 50176, 1200, 2048, [10, 11], ldc=1280)
            run('sq4k', 4096, 4096, 4096, True, True, [(10, 1), (11, 1)])
            run('square', 8192, 8192, 8192, True, True, [(10, 1), (11, 1)])
            run('wgrad TN sw ld1280', 2048, 1200, 50176, False, False, [(10, 6), (11, 6)], ldb=1280)
    elif which == 'spn':
        for _ in range(3):
            run_rownorm('K4 fwd', 50176, 1200, 2048, 1280, engines=(10, 11))
            run('wgrad TN sw ld1280', 2048, 1200, 50176, False, False, [(10, 6), (11, 5), (11, 6), (11, 7), (11, 8)], ldb=1280)
    elif which == 'pitch':
        run_pitch()
    elif which == 'wg':
        for _ in range(3):
            run('wgrad TN swapped', 2048, 1200, 50176, False, False, [(6, 6), (10, 6), (10, 5), (10, 7)])
            run('wgrad TN sw ld1280', 2048, 1200, 50176, False, False, [(6, 6), (10, 6)], ldb=1280)
    elif which == 'norm':
        for _ in range(3):
            run_rownorm('K4 fwd', 50176, 1200, 2048, 1280)
    elif which == 'probe':
        for _ in range(2):
            run('K4 fwd NT', 50176, 1200, 2048, True, True, [(9, 1)])
            run('square', 8192, 8192, 8192, True, True, [(9, 1)])
    elif which == 'xg':
        for _ in range(2):
            for M in (1920, 2048, 3584):
                run('xg proj', M, 9600, 624, True, True, [(2, 1), (3, 1), (4, 1), (6, 1), (9, 1)])
    else:
        for _ in range(2):
            run('classif2 fwd', 256, 3000, 256, True, True, [(2, 1), (3, 1)])
            run('classif2 wgr', 3000, 256, 256, False, False, [(2, 1), (3, 1)])
            run('classif2 dgr', 256, 256, 3000, True, False, [(2, 4), (3, 4), (2, 8)])
            run('lin_v wgrad', 1200, 2048, 256, False, False, [(2, 1), (3, 1), (4, 1)])
            run('lin_q wgrad', 1200, 2400, 256, False, False, [(2, 1), (3, 1), (4, 1)])
            run('qa fwd', 256, 1200, 2400, True, True, [(2, 6), (2, 4), (2, 3), (3, 6), (3, 9)])
            run('lin_v dgrad', 256, 2048, 1200, True, False, [(2, 4), (2, 2), (3, 4), (3, 8)])
