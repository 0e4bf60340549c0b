"""Question encoder alone at the bench workload: the per-step kernels against vqa_lstm_resident (same box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

B, R, Dv, L, Vq, C = int(os.environ.get('B', 256)), 4, 2048, 14, 10000, 3000
dev = torch.device('cuda', 0)
batches = synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).eval()
for resident in (False, True, False, True):
    m.resident_encoder = resident
    for i in range(5):
        m.encode_question(batches[i % 3][1])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 50
    for i in range(n):
        m.encode_question(batches[i % 3][1])
    e.record(); torch.cuda.synchronize()
    m.check_status()
    print(f'B={B} resident={resident!s:5s}: {s.elapsed_time(e) / n * 1e3:.1f} us per encode_question', flush=True)
