"""Detail of one fuzz_encoder case: per-parameter gradient error of the kernels and of the fp32 oracle against fp64."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vqa_ref
from tests.test_gpu_encoder_train import _model, _batch, rel_err
from vqa_amd.model import VQAAttnModel
seed, want = int(sys.argv[1]), int(sys.argv[2])
rnd = random.Random(seed)
m8 = lambda lo, hi: 8 * rnd.randint(lo, hi)
for case in range(want + 1):
    two = rnd.random() < 0.5
    kw = dict(vq=rnd.randint(8, 120), C=rnd.randint(2, 150), E=rnd.randint(1, 70), Hq=16 * rnd.randint(1, 20), Dv=m8(1, 48),
              A=m8(1, 30), F=m8(1, 30), Hf=m8(1, 16))
    hops = rnd.choice([1, 1, 2]); dtype = rnd.choice([torch.float32, torch.bfloat16])
    B, R, L = rnd.choice([6, 9, 19, 40]), rnd.choice([1, 7, 25, 50]), rnd.randint(3, 15)
print(kw, two, hops, B, R, L, dtype)
m, p = _model(VQAAttnModel, kw, dtype, two, visual_out=kw['Dv'], attention_dim=kw['A'], hops=hops)
b = _batch(kw, B, R, L, seed=want)
feats = b['feats'].to(dtype)
po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
for train in (False, True):
    m.train(train)
    seeds = [vqa_ref.site_seed(1234, m.drop_step + 1, s) for s in range(6)] if train else None
    ref = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], None, hops, drop_seeds=seeds, train_encoder=True)
    ref64 = vqa_ref.loss_and_grads({k: v.double() for k, v in po.items()}, feats.double(), b['tokens'], b['answers'], None, hops,
                                   drop_seeds=seeds, train_encoder=True)
    logits = m(feats.cuda(), b['tokens'].cuda())
    m.zero_grad()
    torch.nn.functional.cross_entropy(logits.float(), b['answers'].cuda()).backward()
    named = dict(m.named_parameters())
    for k, gref in ref[2].items():
        e, n = rel_err(named[k].grad, ref64[2][k]), rel_err(gref, ref64[2][k])
        if e > 2e-5:
            print(train, k, f'kernels {e:.2e}  fp32 oracle {n:.2e}  max|g| {float(ref64[2][k].abs().max()):.3e}  '
                  f'max|g| of the matching weight {float(ref64[2].get(k.replace(".bias", ".weight"), ref64[2][k]).abs().max()):.3e}')
