"""Random widths through the attention model with a trainable question encoder (both encoder architectures, eval and
training mode with the oracle's dropout masks) against the oracle's autograd (development aid, GPU box)."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_encoder_train import _model, _batch, _compare
from vqa_amd.model import VQAAttnModel

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m8 = lambda lo, hi: 8 * rnd.randint(lo, hi)
for case in range(n_cases):
    two = rnd.random() < 0.5
    kw = dict(vq=rnd.randint(8, 120), C=rnd.randint(2, 150), E=rnd.randint(1, 70), Hq=16 * rnd.randint(1, 20), Dv=m8(1, 48),
              A=m8(1, 30), F=m8(1, 30), Hf=m8(1, 16))
    hops = rnd.choice([1, 1, 2])
    dtype = rnd.choice([torch.float32, torch.bfloat16])
    B, R, L = rnd.choice([6, 9, 19, 40]), rnd.choice([1, 7, 25, 50]), rnd.randint(3, 15)
    tag = f'case {case}: {kw} two={two} hops={hops} B={B} R={R} L={L} {dtype}'
    m, p = _model(VQAAttnModel, kw, dtype, two, visual_out=kw['Dv'], attention_dim=kw['A'], hops=hops)
    b = _batch(kw, B, R, L, seed=case)
    feats = b['feats'].to(dtype)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    try:
        _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=False, hops=hops)
        _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=True, hops=hops)
    except AssertionError as e:
        a = e.args[0] if e.args else None
        if isinstance(a, tuple) and len(a) == 3 and isinstance(a[2], float) and a[2] <= (5e-4 if dtype == torch.float32 else 3e-2):
            print('noise', tag, a, flush=True)
            continue
        print('MISMATCH', tag, e, flush=True)
        sys.exit(1)
    if case % 5 == 0:
        print('ok', tag, flush=True)
print('done', n_cases, 'cases')
