"""HBM bandwidth probe with stock torch ops (calibration of the achievable roofline on this box)."""
import torch
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    dt = t(lambda: y.copy_(x)); print(f'copy  {mb:5d} MiB: {2 * n * 4 / dt / 1e12:.2f} TB/s (read+write)')
    dt = t(lambda: x.sum()); print(f'sum   {mb:5d} MiB: {n * 4 / dt / 1e12:.2f} TB/s (read)')
    dt = t(lambda: y.fill_(1.0)); print(f'fill  {mb:5d} MiB: {n * 4 / dt / 1e12:.2f} TB/s (write)')
    xb = x.view(torch.bfloat16) if False else x.to(torch.bfloat16)
    dt = t(lambda: xb.float().sum() if False else xb.sum(dtype=torch.float32)); print(f'sumbf {mb//2:5d} MiB: {n * 2 / dt / 1e12:.2f} TB/s (read bf16)')
    del x, y, xb
