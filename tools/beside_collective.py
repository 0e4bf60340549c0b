"""What the kernels whose workgroups all run for the whole launch (the big weight gradient, the weight-stationary
encoder) cost beside a collective's kernel: a stand-in that holds N compute units (one 256-thread workgroup each, as an
RCCL kernel holds one per channel) runs on a second stream.  Development build: make dev;
VQA_HIP_LIB=vqa_amd/libvqa_hip_dev.so python tools/beside_collective.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd._lib import call, ptr
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

lib = _lib.load()
lib.vqa_dev_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda', 0)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
dt = torch.bfloat16
M, N, K = 2048, 1200, 50176                       # the weight gradient as the step runs it: V^T dP
A = (torch.randn((K, M), device=dev) * 0.5).to(dt)
Bm = (torch.randn((K, N), device=dev) * 0.5).to(dt)
slab = torch.empty(8 * M * N, device=dev)
B, L, Vq, C = 256, 14, 10000, 3000
batches = synth_batches(3, B, 4, 2048, L, Vq, C, dev, 42, dt)
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=dt).to(dev).eval()


def timed(fn, occupy):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        if occupy:
            with torch.cuda.stream(side):
                lib.vqa_dev_occupy(occupy, 1500, ptr(sink), ctypes.c_void_p(side.cuda_stream))
            torch.cuda._sleep(200000)             # the stand-in is resident before the measured launch is enqueued
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) * 1e3)
    return sorted(out)[len(out) // 2]


for occupy in (0, 16, 32, 48, 56):
    row = [f'{occupy:2d} CUs held']
    for sk in (6, 5):
        t = timed(lambda: call('vqa_gemm_slabs', ptr(A), M, 0, ptr(Bm), N, 0, ptr(slab), M * N, N, M, N, K, sk, 1, 0), occupy)
        row.append(f'weight gradient, {sk} K ranges {t:6.1f} us')
    t = timed(lambda: m.encode_question(batches[0][1]), occupy)
    row.append(f'question encoder (200 workgroups) {t:6.1f} us')
    print(' | '.join(row), flush=True)
m.check_status()
