"""8192^3 and 4096 x 16384 x 4096 bf16 products with bf16 output through vqa_gemm_fused, timed (store policy A/B with a
development build: VQA_GEMM_NT=0 / 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr, BF16
for M, N, K in ((8192, 8192, 8192), (16384, 4096, 4096), (50176, 1200, 2048), (294912, 1200, 2048)):
    A = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    B = (torch.randn(N, K, device='cuda') * 0.5).to(torch.bfloat16)
    ldc = (N + 63) // 64 * 64
    C = torch.empty(M, ldc, device='cuda', dtype=torch.bfloat16)
    f = lambda: call('vqa_gemm_fused', ptr(A), K, 1, ptr(B), K, 1, ptr(C), ldc, M, N, K, None, None, 0, BF16, BF16, 11)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f'{M} x {N} x {K}: {us:.1f} us  {2.0 * M * N * K / us / 1e6:.0f} TF  (VQA_GEMM_NT={os.environ.get("VQA_GEMM_NT", "unset")})', flush=True)
    del A, B, C
