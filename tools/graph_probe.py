"""Does a HIP graph shorten the kernel-to-kernel time of dependent tiny kernels?  Eager stream launches vs
a captured graph of the same chain (development probe)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd._lib import call, ptr, BF16

x = torch.randn(256, 1200, device='cuda').to(torch.bfloat16)
y = torch.empty_like(x)
N = 400

def chain():
    for i in range(N // 2):
        call('vqa_dropout', ptr(x), 1200, ptr(y), 1200, 256, 1200, 1 + i, 0.5, BF16)
        call('vqa_dropout', ptr(y), 1200, ptr(x), 1200, 256, 1200, 2 + i, 0.5, BF16)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / N)
    return best

_lib.bind_stream()
print(f'eager: {timed(chain):.2f} us per kernel')
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    _lib.bind_stream()
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        _lib.bind_stream()
        chain()
torch.cuda.synchronize()
print(f'graph: {timed(g.replay):.2f} us per kernel')
