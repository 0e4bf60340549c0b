"""Step time at the bench workload with the question encoder frozen / trainable, lstm / 2-lstm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
dev = torch.device('cuda', 0)
batches = synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
for arch in ('lstm', '2-lstm'):
    for tr in (False, True):
        m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, seq2vec_arch=arch, train_encoder=tr).to(dev).train()
        for i in range(5):
            m.train_step(*batches[i % 3])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(30):
            m.train_step(*batches[i % 3])
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 30
        print(f'{arch:7s} train_encoder={tr!s:5s}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} samples/s', flush=True)
        del m
