"""Per-step GPU time of the first steps after a short warm-up (one event per step): does the step time settle?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqa_amd.model import VQAAttnModel
dev = torch.device('cuda', 0)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).train()
batches = bench.synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for i in range(W): m.train_step(*batches[i % 3])
m.sync(); torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for i in range(40):
    m.train_step(*batches[(W + i) % 3])
    evs[i + 1].record()
torch.cuda.synchronize()
print('warmup', W, ' '.join(f'{evs[i].elapsed_time(evs[i + 1]) * 1e3:.0f}' for i in range(40)))
