"""Platform calibration on the GPU box: HBM stream copy / read rates, the vendor bf16 GEMM
(torch.matmul -> hipBLASLt) and this library's GEMM at the image-projection shape, and the CPU
oracle at 1 and all host threads.  Writes one JSON object to stdout.

    python tools/calibrate.py [--cpu]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def gpu_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gpu_part():
    from vqa_amd._lib import call, ptr, bind_stream
    bind_stream()
    out = {'device': torch.cuda.get_device_name(0)}
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device='cuda').random_(0, 255)
    b = torch.empty_like(a)
    t = gpu_time(lambda: b.copy_(a))
    out['stream_copy_GBs'] = 2 * n / t / 1e9            # read + write
    # read-only stream through this library's row-norm kernel (bf16, 2048 columns, nontemporal loads)
    rows = n // (2 * 2048)
    inv = torch.empty(rows, dtype=torch.float32, device='cuda')
    t = gpu_time(lambda: call('vqa_inv_norm', ptr(a), ptr(inv), rows, 2048, 2048, 1))
    out['read_stream_GBs'] = n / t / 1e9
    del a, b
    dt = torch.bfloat16
    for name, (M, N, K) in {'k4_fwd_50176x1200x2048': (50176, 1200, 2048), 'square_8192': (8192, 8192, 8192)}.items():
        A = (torch.randn((M, K), device='cuda') * 0.5).to(dt)
        W = (torch.randn((N, K), device='cuda') * 0.5).to(dt)
        C = torch.empty((M, N), device='cuda', dtype=dt)
        t = gpu_time(lambda: torch.matmul(A, W.t(), out=C))
        out[f'hipblaslt_{name}_TFs'] = 2.0 * M * N * K / t / 1e12
        t = gpu_time(lambda: call('vqa_gemm_fused', ptr(A), K, 1, ptr(W), K, 1, ptr(C), N, M, N, K, None, None, 0, 1, 1, 0))
        out[f'vqa_gemm_{name}_TFs'] = 2.0 * M * N * K / t / 1e12
    # weight-gradient shape: dW (1200 x 2048) = dP^T (50176 x 1200) . V (50176 x 2048)
    M, N, K = 1200, 2048, 50176
    dP = (torch.randn((K, M), device='cuda') * 0.5).to(dt)
    V = (torch.randn((K, N), device='cuda') * 0.5).to(dt)
    t = gpu_time(lambda: torch.matmul(dP.t(), V))
    out['hipblaslt_k4_wgrad_1200x2048x50176_TFs'] = 2.0 * M * N * K / t / 1e12
    sk = call('vqa_gemm_splitk_hint', M, N, K, 0, 0)
    slab = torch.empty(sk * M * N, device='cuda', dtype=torch.float32)
    t = gpu_time(lambda: call('vqa_gemm_slabs', ptr(dP), M, 0, ptr(V), N, 0, ptr(slab), M * N, N, M, N, K, sk, 1, 0))
    out['vqa_gemm_k4_wgrad_1200x2048x50176_TFs'] = 2.0 * M * N * K / t / 1e12
    out['vqa_gemm_k4_wgrad_splitk'] = sk
    return out


def cpu_part():
    from oracle import vqa_ref
    res = {'host_logical_cpus': os.cpu_count()}
    try:
        with open('/proc/cpuinfo') as f:
            res['cpu_model'] = next(l.split(':', 1)[1].strip() for l in f if l.startswith('model name'))
    except Exception:
        pass
    Vq, C, R, Dv, L = 10000, 3000, 196, 2048, 14
    p0 = vqa_ref.init_params(vq=Vq, C=C, E=620, Hq=2400, Dv=Dv, A=1200, F=1200, Hf=256, seed=0)
    for B in (32, 256):
        b = vqa_ref.synthetic_batch(B, R, Dv, L, Vq, C, seed=42)
        for threads in (16, 1):
            torch.set_num_threads(threads)
            p = {k: v.clone() for k, v in p0.items()}
            names = vqa_ref.trainable_names(p)
            m = {k: torch.zeros_like(p[k]) for k in names}
            v = {k: torch.zeros_like(p[k]) for k in names}
            seeds = [vqa_ref.site_seed(0, 1, s) for s in range(4)]
            warm, n = (3, 10) if threads > 1 or B == 32 else (1, 3)
            ts = []
            for i in range(warm + n):
                t0 = time.perf_counter()
                _, _, g = vqa_ref.loss_and_grads(p, b['feats'], b['tokens'], b['answers'], None, 1, drop_seeds=seeds)
                for k in names:
                    vqa_ref.adam_step(p[k], g[k], m[k], v[k], i + 1, 1e-5)
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[warm:])
            res[f'cpu_oracle_B{B}_threads{threads}_samples_per_s'] = B / ts[len(ts) // 2]
            print(f'# B={B} threads={threads} median {ts[len(ts) // 2]:.3f} s/step', file=sys.stderr, flush=True)
    return res


if __name__ == '__main__':
    out = gpu_part() if '--cpu' not in sys.argv else {}
    if '--cpu' in sys.argv or '--all' in sys.argv:
        out.update(cpu_part())
    print(json.dumps(out, indent=1))
