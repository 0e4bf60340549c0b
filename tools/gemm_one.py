"""Runs one GEMM shape a few times (for counter passes).  args: name"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd._lib import call, ptr
which = sys.argv[1] if len(sys.argv) > 1 else 'k4'
dt = torch.bfloat16
if which in ('k4', 'k4fused'):
    M, N, K, akm, bkm, sk = 50176, 1200, 2048, 1, 1, 1
elif which == 'wgrad':
    M, N, K, akm, bkm, sk = 1200, 2048, 50176, 0, 0, 3
elif which == 'wgrad_swapped':
    M, N, K, akm, bkm, sk = 2048, 1200, 50176, 0, 0, 6
elif which == 'sq4k':
    M, N, K, akm, bkm, sk = 4096, 4096, 4096, 1, 1, 1
elif which == 'sq8k':
    M, N, K, akm, bkm, sk = 8192, 8192, 8192, 1, 1, 1
else:
    M, N, K, akm, bkm, sk = 256, 9600, 2400, 1, 1, 3
if which == 'k4fused':
    # the projection as the step calls it: post-ReLU features, bf16 output with a 128-byte-aligned pitch, bias, row
    # norms computed by the same launch
    M, N, K = 50176, 1200, 2048
    A = (torch.randn((M, K), device='cuda')).clamp_min_(0).to(dt)
    W = (torch.randn((N, K), device='cuda') * 0.02).to(dt)
    bias = torch.randn(N, device='cuda')
    inv = torch.empty(M, device='cuda')
    C = torch.empty((M, 1280), device='cuda', dtype=dt)
    eng = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for _ in range(4):
        call('vqa_gemm_fused_rownorm', ptr(A), K, ptr(W), K, ptr(C), 1280, M, N, K, ptr(bias), ptr(inv), 0, 1, 1, eng)
    torch.cuda.synchronize()
    print('ok')
    sys.exit(0)
A = (torch.randn((M, K) if akm else (K, M), device='cuda') * 0.5).to(dt)
Bm = (torch.randn((N, K) if bkm else (K, N), device='cuda') * 0.5).to(dt)
eng = int(sys.argv[2]) if len(sys.argv) > 2 else 0
slab = torch.empty(sk * M * N, device='cuda', dtype=torch.float32)
for _ in range(4):
    call('vqa_gemm_slabs', ptr(A), K if akm else M, akm, ptr(Bm), K if bkm else N, bkm, ptr(slab), M * N, N, M, N, K, sk, 1, eng)
torch.cuda.synchronize()
print('ok')
