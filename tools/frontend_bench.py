"""HBM rate of the grid-mean kernel on ResNet-152 stage-map shapes (both layouts) and the step
time of the gated-pooling model.  python tools/frontend_bench.py [B] [input_size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd._lib import call, ptr

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
stages = [(256, S // 4), (512, S // 8), (1024, S // 16), (2048, S // 32)]
dt = torch.bfloat16


def gpu_time(fn, iters=10, warm=2):
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


_lib.bind_stream()
for C, g in stages:
    HW = g * g
    x = torch.randn((B, C, HW), device='cuda').clamp_min_(0).to(dt)
    out = torch.empty((B, C), device='cuda')
    t = gpu_time(lambda: call('vqa_grid_mean', ptr(x), 0, B, C, HW, ptr(out), C, 0, None, 1, 1))
    ref = x.float().mean(2)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'rows  C={C:5d} HW={HW:6d}: {x.numel() * 2 / t / 1e9:8.1f} GB/s  {t * 1e6:8.1f} us  err {err:.1e}')
    xc = x.permute(0, 2, 1).contiguous()
    nsp = call('vqa_grid_mean_splits', 1, B, C, HW)
    part = torch.empty(max(nsp * B * C, 1), device='cuda')
    t = gpu_time(lambda: call('vqa_grid_mean', ptr(xc), 1, B, C, HW, ptr(out), C, 0, ptr(part), nsp, 1))
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f'chlast C={C:5d} HW={HW:6d}: {x.numel() * 2 / t / 1e9:8.1f} GB/s  {t * 1e6:8.1f} us  splits {nsp} err {err:.1e}')
    del x, xc

from vqa_amd.model import VQAGAPModel
m = VQAGAPModel(['w'] * 10000, ['a'] * 3000, None, dtype=dt).cuda().train()
maps = [torch.randn((B, C, g, g), device='cuda').clamp_min_(0).to(dt) for C, g in stages]
tok = torch.randint(1, 10000, (B, 14), device='cuda')
tok[:, 9:] = 0
ans = torch.randint(0, 3000, (B,), device='cuda')
t = gpu_time(lambda: m.train_step(maps, tok, ans))
nbytes = sum(x.numel() * 2 for x in maps)
print(f'VQAGAPModel train_step B={B}: {t * 1e3:.3f} ms  ({B / t:.0f} samples/s; maps {nbytes / 1e6:.0f} MB -> {nbytes / t / 1e9:.0f} GB/s if only the maps were read)')
