"""Runs the question encoder at the bench workload a few times (for counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd.model import VQAAttnModel
from bench import synth_batches
dev = torch.device('cuda', 0)
batches = synth_batches(3, 256, 4, 2048, 14, 10000, 3000, dev, 42, torch.bfloat16)
m = VQAAttnModel(['w'] * 10000, ['a'] * 3000, None, dtype=torch.bfloat16).to(dev).eval()
for i in range(6):
    m.encode_question(batches[i % 3][1])
torch.cuda.synchronize()
m.check_status()
print('ok')
