"""Soak of the train step with the one-launch encoder: many steps while a second stream keeps streaming a buffer
larger than the caches (uneven load on the memory system and the compute units), then the status words and a
comparison of the final weights with a run without the side stream (the step is deterministic).
python tools/soak_resident.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device('cuda', 0)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
batches = synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
big = torch.zeros(96 << 20, dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
finals = []
for load in (False, True):
    m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, seed=0).to(dev).train()
    for i in range(steps):
        if load and i % 7 == 0:
            with torch.cuda.stream(side):
                big.add_(1.0)
        loss = m.train_step(*batches[i % 3], lr=1e-5)
        if i % 5000 == 4999:
            print(f'load={load} step {i + 1}: loss {float(loss):.6f}', flush=True)
    m.sync()
    finals.append((float(loss), m._flat.clone()))
    del m
print('final losses', finals[0][0], finals[1][0], 'weights identical:', bool(torch.equal(finals[0][1], finals[1][1])))
