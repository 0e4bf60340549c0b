"""Random shapes through vqa_lstm_resident (both forms) against the per-step kernels and the oracle.
python tools/fuzz_resident.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import random
import torch
from oracle import vqa_ref
from vqa_amd.model import VQAAttnModel

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for case in range(n_cases):
    H = 8 * rng.randint(1, 40)
    B = rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 400, 512])
    L = rng.randint(3, 24)
    vq, E = rng.randint(5, 80), 8 * rng.randint(1, 6)
    kw = dict(vq=vq, C=7, E=E, Hq=H, Dv=32, A=16, F=16, Hf=8)
    p = vqa_ref.init_params(seed=case, **kw)
    m = VQAAttnModel(['w'] * vq, ['a'] * 7, None, dtype=torch.bfloat16, emb_size=E, question_out=H, visual_out=32,
                     attention_dim=16, fusion_in=16, fusion_hidden=8, seed=1)
    m.load_state_dict(p)
    m = m.to('cuda').eval()
    g = torch.Generator().manual_seed(case)
    tok = torch.randint(1, vq + 1, (B, L), generator=g)
    mode = rng.choice(['ragged', 'full', 'short', 'holes'])
    if mode != 'full':
        for b in range(B):
            n = int(torch.randint(0 if mode == 'short' else 1, (3 if mode == 'short' else L) + 1, (1,), generator=g))
            tok[b, n:] = 0
        if mode == 'holes':                       # zeros in the middle count as padding too (seq2vec.py:11-14)
            tok[torch.rand(tok.shape, generator=g) < 0.1] = 0
    out = {}
    for form in ('whole', 'steps', 'per-step'):
        m.resident_encoder, m.resident_whole = form != 'per-step', form == 'whole'
        out[form] = m.encode_question(tok.cuda()).clone()
        m.check_status()
    ref = vqa_ref.encode_question(tok, {k: v.to(torch.bfloat16).float() for k, v in p.items()})
    scale = float(ref.abs().max()) + 1e-6
    e1 = float((out['whole'] - out['per-step']).abs().max()) / scale
    e2 = float((out['steps'] - out['per-step']).abs().max()) / scale
    e3 = float((out['whole'].cpu() - ref).abs().max()) / scale
    worst = max(worst, e1, e2, e3)
    flag = '' if max(e1, e2) <= 1e-2 and e3 <= 3e-2 else '   <-- FLAG'
    print(f'case {case:3d} B {B:3d} H {H:3d} L {L:2d} {mode:6s}: whole {e1:.1e} steps {e2:.1e} vs oracle {e3:.1e}{flag}', flush=True)
    del m
print('worst relative difference', worst)
