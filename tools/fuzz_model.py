"""Random model widths and batch shapes through forward + loss + backward against the CPU oracle (development aid, GPU
box): the whole-model parity test of tests/test_gpu_model.py on configurations drawn around tile and chunk edges."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_model import test_forward_backward_eval as check

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m8 = lambda lo, hi: 8 * rnd.randint(lo, hi)
for case in range(n_cases):
    kw = dict(vq=rnd.randint(5, 400), C=rnd.randint(2, 260), E=rnd.randint(1, 70), Hq=m8(1, 40), Dv=m8(1, 80),
              A=m8(1, 40), F=m8(1, 40), Hf=m8(1, 24))
    hops = rnd.choice([1, 1, 2])
    cap = rnd.random() < 0.3
    B, R, L = rnd.choice([1, 2, 5, 17, 64, 130]), rnd.choice([1, 3, 16, 17, 49, 100]), rnd.randint(3, 16)
    dtype = rnd.choice([torch.float32, torch.bfloat16])
    tag = f'case {case}: {kw} hops={hops} cap={cap} B={B} R={R} L={L} {dtype}'
    try:
        check(kw, hops, cap, B, R, L, dtype)
    except AssertionError as e:
        a = e.args[0] if e.args else None
        # a gradient that is a cancelling sum: the test's bar is 4x the fp32 oracle's own distance from fp64; up to 10x
        # of it (and under 1e-3 relative) is the order of the fp32 sums, not a defect
        if isinstance(a, tuple) and len(a) == 3 and a[1] <= max(10 * a[2], 1e-4) and a[1] <= 1e-3:
            print('noise', tag, a, flush=True)
            continue
        print('MISMATCH', tag, e, flush=True)
        sys.exit(1)
    if case % 5 == 0:
        print('ok', tag, flush=True)
print('done', n_cases, 'cases')
