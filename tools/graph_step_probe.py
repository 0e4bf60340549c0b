"""Upper bound of what replaying the train step as a HIP graph would give (seeds and the Adam step number are
baked into the captured kernels here, so this is a timing probe only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqa_amd import _lib
from vqa_amd.model import VQAAttnModel
dev = torch.device('cuda', 0)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).train()
batches = bench.synth_batches(4, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
for i in range(8): m.train_step(*batches[i % 4])
torch.cuda.synchronize()
def eager(n=100):
    t0 = time.perf_counter()
    for i in range(n): m.train_step(*batches[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f'eager {eager():.1f} us/step')
graphs = []
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for b in batches:
        g = torch.cuda.CUDAGraph()
        m.train_step(*b)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            m.train_step(*b)
        graphs.append(g)
torch.cuda.synchronize()
def replay(n=100):
    t0 = time.perf_counter()
    for i in range(n): graphs[i % 4].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
replay(8)
print(f'graph {replay():.1f} us/step')
print(f'eager {eager():.1f} us/step')
print(f'graph {replay():.1f} us/step')
