"""GPU parity: the HIP module (through the C ABI) against the CPU oracle.

Tolerances (relative to the largest magnitude of the oracle tensor):
fp32 storage 1e-5, bf16 storage 1e-2 -- the bars the north-star states for the
logits; gradients are held to the same bars scaled by 2 (they chain more ops).
Integer results (gather, lengths, argmax, counts) must be bit-exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu

TINY = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)
MID = dict(vq=500, C=101, E=44, Hq=200, Dv=256, A=136, F=104, Hf=72)


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def build(kw, hops=1, cap=False, dtype=torch.float32, seed=3, drop=0.5):
    from vqa_amd.model import VQAAttnModel
    kw = dict(kw)
    vc, Ec = (20, 8) if cap else (0, 0)
    p = vqa_ref.init_params(hops=hops, seed=seed, vc=vc, Ec=Ec, **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                     hops=hops, dropout=drop, caption_vocab=vc, caption_dim=Ec if cap else 384, seed=1234)
    m.load_state_dict(p)
    return m.to('cuda'), p


def batch(kw, B, R, L, seed=5, cap=False):
    b = vqa_ref.synthetic_batch(B, R, kw['Dv'], L, kw['vq'], kw['C'], seed=seed, Tc=6 if cap else 0, vc=20 if cap else 0)
    if B > 3:
        b['tokens'][3] = 0          # all-pad question row
    return b


def oracle_params_as(p, dtype):
    """The oracle sees the weights the kernels see: bf16 mode rounds weights (and inputs) to bf16."""
    if dtype == torch.float32:
        return p
    return {k: v.to(torch.bfloat16).float() for k, v in p.items()}


@pytest.mark.parametrize('kw,hops,cap,B,R,L', [
    (TINY, 1, False, 4, 9, 7), (TINY, 2, False, 4, 9, 7), (TINY, 1, True, 4, 9, 7),
    (MID, 1, False, 37, 49, 14), (MID, 2, True, 16, 196, 14),
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_forward_backward_eval(kw, hops, cap, B, R, L, dtype):
    m, p = build(kw, hops, cap, dtype)
    m.eval()
    b = batch(kw, B, R, L, cap=cap)
    feats = b['feats'].to(dtype)
    po = oracle_params_as(p, dtype)
    logits_ref, loss_ref, grads_ref = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'],
                                                             b.get('captions'), hops)
    # fp64 run of the same oracle: tells how much of a gradient's error is cancellation noise that the
    # fp32 oracle has too (bias gradients of the attention are sums that cancel to ~1e-9)
    _, _, grads64 = vqa_ref.loss_and_grads({k: v.double() for k, v in po.items()}, feats.double(), b['tokens'],
                                           b['answers'], b.get('captions'), hops)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    dev = 'cuda'
    cap_t = b['captions'].to(dev) if cap else None
    logits = m(feats.to(dev), b['tokens'].to(dev), cap_t)
    assert logits.shape == (B, kw['C'])
    assert rel_err(logits, logits_ref) <= tol
    loss = torch.nn.functional.cross_entropy(logits.float(), b['answers'].to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) <= tol * max(1.0, abs(float(loss_ref)))
    named = dict(m.named_parameters())
    for k, gref in grads_ref.items():
        g = named[k].grad
        assert g is not None, k
        if k.endswith('linear_a.bias'):
            # the score offset cancels in the softmax: its gradient is rounding noise around zero
            wref = grads_ref[k.replace('.bias', '.weight')].abs().max()
            assert float((g.cpu() - gref).abs().max()) <= 2 * tol * max(float(wref), 1e-3), k
            continue
        noise = rel_err(gref, grads64[k])
        assert rel_err(g, grads64[k]) <= max(2 * tol, 4 * noise), (k, rel_err(g, grads64[k]), noise)
    for k, v in named.items():
        if k.startswith('seq2vec.') or k.startswith('caption_embedding.'):
            assert v.grad is None and not v.requires_grad


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_golden_fixture_h1(golden_dir, dtype):
    """Committed fixture (oracle-generated): parameters, inputs, logits, loss, every gradient, with and
    without dropout."""
    from vqa_amd.model import VQAAttnModel
    z = np.load(os.path.join(golden_dir, 'g3_model_h1.npz'))
    p = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('param:')}
    kw = TINY
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=1234)
    m.load_state_dict(p)
    m = m.to('cuda')
    tol = 1e-5 if dtype == torch.float32 else 2e-2      # fixture is fp32 arithmetic on fp32 weights
    feats = torch.from_numpy(z['in:feats']).to('cuda')
    tok = torch.from_numpy(z['in:tokens']).to('cuda')
    ans = torch.from_numpy(z['in:answers']).to('cuda')
    m.eval()
    logits = m(feats, tok)
    assert rel_err(logits, torch.from_numpy(z['logits'])) <= tol
    # training mode: the dropout mask is an integer hash, the oracle draws the same one
    m.train()
    m.drop_step = 6                                       # next forward is dropout step 7, as in the fixture
    logits_d = m(feats, tok)
    assert [int(s) for s in z['drop_seeds']] == [vqa_ref.site_seed(1234, 7, s) for s in range(4)]
    assert rel_err(logits_d, torch.from_numpy(z['logits_drop'])) <= tol
    torch.nn.functional.cross_entropy(logits_d.float(), ans).backward()
    named = dict(m.named_parameters())
    for k in z.files:
        if k.startswith('grad_drop:'):
            gref = torch.from_numpy(z[k])
            if k.endswith('linear_a.bias'):
                continue
            assert rel_err(named[k[10:]].grad, gref) <= 2 * tol, k


def test_uniform_attention_is_mean_pooling():
    """w_a = 0 => p = 1/R and pooled = mean over regions of the normalised features
    (the reference's torch.mean(x, axis=(2,3)), model.py:161-164)."""
    m, p = build(TINY, dtype=torch.float32)
    m.eval()
    with torch.no_grad():
        m.attention[0].linear_a.weight.zero_()
    b = batch(TINY, 4, 9, 7)
    m(b['feats'].cuda(), b['tokens'].cuda())
    ws = m._ctx[0]
    torch.cuda.synchronize()
    assert torch.allclose(ws['p0'].cpu(), torch.full((4, 9), 1.0 / 9), atol=1e-7)
    ref = vqa_ref.l2_normalise(b['feats']).mean(1)
    assert rel_err(ws['pooled0'].float(), ref) <= 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_question_encoder_against_reference_fixture(golden_dir, dtype):
    """Embedding gather (bit-exact), lengths (bit-exact), LSTM hidden states and last-step selection against
    outputs of the reference's own seq2vec.LSTM sub-modules (fixture g2)."""
    from vqa_amd.model import VQAAttnModel
    from vqa_amd import _lib
    z = np.load(os.path.join(golden_dir, 'g2_seq2vec.npz'))
    E, H = z['embedding'].shape[1], z['w_hh'].shape[1]
    m = VQAAttnModel(['w'] * (z['embedding'].shape[0] - 1), ['a'] * 5, None, dtype=dtype, emb_size=E, question_out=H,
                     visual_out=32, attention_dim=16, fusion_in=16, fusion_hidden=8)
    sd = m.state_dict()
    sd['seq2vec.embedding.weight'] = torch.from_numpy(z['embedding'])
    sd['seq2vec.rnn.weight_ih_l0'] = torch.from_numpy(z['w_ih'])
    sd['seq2vec.rnn.weight_hh_l0'] = torch.from_numpy(z['w_hh'])
    sd['seq2vec.rnn.bias_ih_l0'] = torch.from_numpy(z['b_ih'])
    sd['seq2vec.rnn.bias_hh_l0'] = torch.from_numpy(z['b_hh'])
    m.load_state_dict(sd)
    m = m.to('cuda')
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for name in ('main', 'rand'):
        ids = torch.from_numpy(z[f'{name}_ids']).cuda()
        B, L = ids.shape
        m.fold_embedding, m.fold_first_step = True, True     # both per-token tables (frozen encoder)
        q_both = m.encode_question(ids).clone()
        assert m._xg_table is not None and m._hh_table is not None
        assert rel_err(q_both, torch.from_numpy(z[f'{name}_last'])) <= tol
        m.fold_first_step = False       # table of input projections only: bit-identical to the unfolded path
        q_table = m.encode_question(ids).clone()
        assert m._xg_table is not None and m._hh_table is None
        assert rel_err(q_table, torch.from_numpy(z[f'{name}_last'])) <= tol
        m.fold_embedding = False        # gather + per-step projection: the intermediate gather is checked below
        q = m.encode_question(ids)
        ws = m._enc_workspace(B, L, ids.device)
        torch.cuda.synchronize()
        assert m._xg_table is None and torch.equal(q, q_table)
        assert np.array_equal(ws['len'].cpu().numpy(), z[f'{name}_lengths'].astype(np.int32))
        # packed gather: row offs[t] + j holds the embedding of token t of sorted row j (= original row
        # perm[j]) for the m_t[t] rows still running; bit-exact against the table in the storage dtype
        ids_np = z[f'{name}_ids']
        perm, m_t, offs = (ws[k].cpu().numpy() for k in ('perm', 'm_t', 'offs'))
        table = torch.from_numpy(z['embedding']).to(dtype).float()
        x_emb = ws['x_emb'][:, :E].float().cpu()
        steps = np.where(z[f'{name}_lengths'] == 0, L, z[f'{name}_lengths'])
        assert offs[L] == steps.sum()
        for t in range(L):
            rows = perm[:m_t[t]]
            assert np.all(steps[rows] > t) and m_t[t] == (steps > t).sum()
            got = x_emb[offs[t]:offs[t] + m_t[t]]
            assert torch.equal(got, table[torch.from_numpy(ids_np[rows, t])])
            if dtype == torch.float32:
                assert torch.equal(got, torch.from_numpy(z[f'{name}_emb'])[torch.from_numpy(rows), t])
        assert rel_err(q, torch.from_numpy(z[f'{name}_last'])) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_two_lstm_encoder_against_reference_fixture(golden_dir, dtype):
    """TwoLSTM (seq2vec.py:48-76) against outputs of the reference's own module run along the token axis:
    tanh on the embeddings, two stacked LSTMs, select_last of each, concatenation."""
    from vqa_amd.model import VQAAttnModel
    z = np.load(os.path.join(golden_dir, 'g2b_two_lstm.npz'))
    Vq1, E = z['embedding'].shape
    Hh = z['rnn_0.weight_hh_l0'].shape[1]
    m = VQAAttnModel(['w'] * (Vq1 - 1), ['a'] * 5, None, dtype=dtype, emb_size=E, question_out=2 * Hh, visual_out=32,
                     attention_dim=16, fusion_in=16, fusion_hidden=8, seq2vec_arch='2-lstm')
    sd = m.state_dict()
    sd['seq2vec.embedding.weight'] = torch.from_numpy(z['embedding'])
    for k in z.files:
        if k.startswith('rnn_'):
            sd['seq2vec.' + k] = torch.from_numpy(z[k])
    m.load_state_dict(sd)
    m = m.to('cuda').eval()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for name in ('main', 'rand'):
        q = m.encode_question(torch.from_numpy(z[f'{name}_ids']).cuda())
        assert rel_err(q, torch.from_numpy(z[f'{name}_last'])) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_two_lstm_model_forward_backward(dtype):
    """Whole step with the 2-lstm question encoder, eval and training mode (dropout 0.3 on the two
    question vectors uses dropout sites 4 and 5)."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(MID, Hq=208)
    p = vqa_ref.init_params(seed=3, two_lstm=True, **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                     seed=1234, seq2vec_arch='2-lstm')
    m.load_state_dict(p)
    m = m.to('cuda')
    b = batch(kw, 21, 25, 14)
    feats = b['feats'].to(dtype)
    po = oracle_params_as(p, dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for train in (False, True):
        m.train(train)
        seeds = [vqa_ref.site_seed(1234, m.drop_step + 1, s) for s in range(6)] if train else None
        logits_ref, loss_ref, grads_ref = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], None, 1,
                                                                 drop_seeds=seeds)
        _, _, grads64 = vqa_ref.loss_and_grads({k: v.double() for k, v in po.items()}, feats.double(), b['tokens'],
                                               b['answers'], None, 1, drop_seeds=seeds)
        logits = m(feats.cuda(), b['tokens'].cuda())
        assert rel_err(logits, logits_ref) <= tol
        m.zero_grad()
        torch.nn.functional.cross_entropy(logits.float(), b['answers'].cuda()).backward()
        named = dict(m.named_parameters())
        for k, gref in grads_ref.items():
            if k.endswith('linear_a.bias'):
                continue
            noise = rel_err(gref, grads64[k])
            assert rel_err(named[k].grad, grads64[k]) <= max(2 * tol, 4 * noise), (k, train)


def test_train_step_matches_oracle_adam():
    """Three fused steps (fwd + CE + bwd + Adam, dropout on) against the oracle's autograd + Adam."""
    kw, dtype = MID, torch.float32
    m, p = build(kw, 1, False, dtype)
    m.train()
    po = {k: v.clone() for k, v in p.items()}
    names = vqa_ref.trainable_names(po)
    mom = {k: torch.zeros_like(po[k]) for k in names}
    var = {k: torch.zeros_like(po[k]) for k in names}
    lr = 1e-3
    for step in range(1, 4):
        b = batch(kw, 24, 25, 14, seed=100 + step)
        seeds = [vqa_ref.site_seed(1234, step, s) for s in range(4)]
        _, loss_ref, grads = vqa_ref.loss_and_grads(po, b['feats'], b['tokens'], b['answers'], None, 1, drop_seeds=seeds)
        for k in names:
            vqa_ref.adam_step(po[k], grads[k], mom[k], var[k], step, lr)
        loss = m.train_step(b['feats'].cuda(), b['tokens'].cuda(), b['answers'].cuda(), lr=lr)
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref)))
    sd = m.state_dict()
    for k in names:
        if k.endswith('linear_a.bias'):
            continue     # zero-mean rounding noise as gradient: Adam turns its sign into +-lr steps
        assert rel_err(sd[k], po[k]) <= 2e-5, k


class _LoopbackReducer:
    """One-rank stand-in for GradReducer with the deferral switched on: the collective is the
    identity, so parameters must come out bit-identical to the reducer-less step."""
    enabled, defer_last, world = True, True, 1

    def __init__(self, defer_all=False):
        self.fired, self.last_waited, self.defer_all = [], 0, defer_all

    def defer_everything(self):
        self.deferred = True
        self.fired = list(range(self.nb))
        return 1.0

    def begin(self, model):
        self.fired, self.nb = [], len(model.buckets)

    def bucket_ready(self, index):
        if index not in self.fired:
            self.fired.append(index)

    def finish(self, defer_last=False):
        self.deferred = defer_last
        return 1.0

    def finish_last(self):
        self.last_waited += 1


@pytest.mark.parametrize('defer_all', [False, True])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_deferred_last_bucket_is_exact(dtype, defer_all):
    """The last gradient bucket's Adam update is applied inside the next step (or by sync /
    state_dict); every parameter ends bit-identical to the undeferred schedule."""
    kw = MID
    ma, _ = build(kw, 2, False, dtype)
    mb, _ = build(kw, 2, False, dtype)
    ma.train(), mb.train()
    red = _LoopbackReducer(defer_all)
    lo, hi = None, None
    for step in range(1, 4):
        b = batch(kw, 16, 25, 14, seed=200 + step)
        args = (b['feats'].cuda().to(dtype), b['tokens'].cuda(), b['answers'].cuda())
        la = ma.train_step(*args, lr=1e-3)
        lb = mb.train_step(*args, lr=1e-3, reducer=red)
        assert torch.equal(la, lb)
        lo, hi = mb.buckets[-1]
        if defer_all:                # every bucket's update is one step behind
            lo = 0
        assert red.deferred and red.fired == list(range(len(mb.buckets)))
        assert mb._pending is not None and red.last_waited == step - 1
        torch.cuda.synchronize()    # the deferred slice is one update behind, the rest is current
        assert torch.equal(mb._flat[:lo], ma._flat[:lo])
        assert not torch.equal(mb._flat[lo:hi], ma._flat[lo:hi])
    sd_a, sd_b = ma.state_dict(), mb.state_dict()        # state_dict() flushes the deferred update
    assert mb._pending is None and red.last_waited == 3
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    if dtype == torch.bfloat16:
        assert torch.equal(ma._shadow, mb._shadow)
    # eval forward after a deferred step sees the flushed parameters
    b = batch(kw, 16, 25, 14, seed=300)
    args = (b['feats'].cuda().to(dtype), b['tokens'].cuda(), b['answers'].cuda())
    ma.train_step(*args, lr=1e-3)
    mb.train_step(*args, lr=1e-3, reducer=red)
    ma.eval(), mb.eval()
    with torch.no_grad():
        assert torch.equal(ma(args[0], args[1]), mb(args[0], args[1]))


@pytest.mark.parametrize('B,L', [(1, 5), (37, 14), (256, 14), (1000, 20)])
def test_sequence_pack_bit_exact(B, L):
    """Lengths, the stable descending sort by step count, per-step live counts and their prefix sums
    are integer work: exact against numpy.  Includes all-pad rows (they take all L steps) and
    interior zeros (process_lengths counts zeros anywhere, seq2vec.py:11-14)."""
    from vqa_amd import _lib
    from vqa_amd._lib import call, ptr
    rng = np.random.default_rng(B + L)
    tok = rng.integers(1, 50, size=(B, L))
    lens = rng.integers(0, L + 1, size=B)
    tok[np.arange(L)[None, :] >= lens[:, None]] = 0
    if B > 4:
        tok[2, 1] = 0                                       # a zero in the middle of a question
    t = torch.from_numpy(tok).cuda()
    out = {k: torch.full((n,), -7, dtype=torch.int32, device='cuda') for k, n in
           (('len', B), ('perm', B), ('m_t', L), ('offs', L + 1))}
    _lib.bind_stream()
    call('vqa_sequence_pack', ptr(t), B, L, ptr(out['len']), ptr(out['perm']), ptr(out['m_t']), ptr(out['offs']), 0, None)
    ln = L - (tok == 0).sum(1)
    assert np.array_equal(out['len'].cpu().numpy(), vqa_ref.process_lengths(torch.from_numpy(tok)).numpy())
    steps = np.where(ln == 0, L, ln)
    perm = np.argsort(-steps, kind='stable')
    m_t = np.array([(steps > s).sum() for s in range(L)])
    assert np.array_equal(out['perm'].cpu().numpy(), perm)
    assert np.array_equal(out['m_t'].cpu().numpy(), m_t)
    assert np.array_equal(out['offs'].cpu().numpy(), np.concatenate([[0], np.cumsum(m_t)]))


def test_argmax_and_counts_bit_exact(golden_dir):
    from vqa_amd import _lib
    from vqa_amd._lib import call, ptr
    z = np.load(os.path.join(golden_dir, 'g5_ce_g6_metrics.npz'))
    for key, exp in (('logits', 'argmax'), ('logits_nan', 'argmax_nan')):
        lg = torch.from_numpy(z[key]).cuda().contiguous()
        B, C = lg.shape
        pred = torch.zeros(B, dtype=torch.int64, device='cuda')
        counters = torch.zeros(8, dtype=torch.int64, device='cuda')
        ans = torch.from_numpy(z['target']).cuda()
        types = torch.from_numpy(z['types'].astype(np.int32)).cuda()
        call('vqa_argmax_accuracy', ptr(lg), C, B, C, ptr(ans), ptr(types), 4, ptr(pred), ptr(counters))
        torch.cuda.synchronize()
        assert np.array_equal(pred.cpu().numpy(), z[exp])
        assert np.array_equal(pred.cpu().numpy(), np.argmax(z[key], axis=1))
        if key == 'logits':
            assert np.array_equal(counters.cpu().numpy()[:4], z['count'])
            assert np.array_equal(counters.cpu().numpy()[4:], z['right'])
    # big random case with forced ties, against numpy
    g = torch.Generator().manual_seed(0)
    lg = torch.randint(-3, 4, (257, 3000), generator=g).float()
    pred = torch.zeros(257, dtype=torch.int64, device='cuda')
    lgd = lg.cuda()
    call('vqa_argmax_accuracy', ptr(lgd), 3000, 257, 3000, None, None, 4, ptr(pred), None)
    torch.cuda.synchronize()
    assert np.array_equal(pred.cpu().numpy(), np.argmax(lg.numpy(), axis=1))


def test_softmax_ce_fixture(golden_dir):
    from vqa_amd._lib import call, ptr, F32
    z = np.load(os.path.join(golden_dir, 'g5_ce_g6_metrics.npz'))
    lg = torch.from_numpy(z['logits']).cuda()
    B, C = lg.shape
    y = torch.from_numpy(z['target']).cuda()
    rows = torch.zeros(B, device='cuda')
    dl = torch.zeros(B, 16, device='cuda')
    out = torch.zeros(1, device='cuda')
    call('vqa_softmax_ce', ptr(lg), C, ptr(y), ptr(rows), ptr(dl), 16, B, C, 1.0 / B, F32, None)
    call('vqa_sum_scale', ptr(rows), ptr(out), B, 1.0 / B)
    torch.cuda.synchronize()
    assert abs(float(out) - float(z['loss'])) <= 1e-6 * abs(float(z['loss']))
    assert rel_err(dl[:, :C], torch.from_numpy(z['dlogits'])) <= 1e-6
    assert float(dl[:, C:].abs().max()) == 0.0


def test_train_driver_synthetic(tmp_path):
    """The reference-CLI driver on synthetic data: loss falls, checkpoints carry the reference key names and
    reload into a fresh module giving identical logits."""
    from vqa_amd import train as drv
    from vqa_amd.model import VQAAttnModel
    import argparse
    args = argparse.Namespace(model='VQAAttnModel', gpu=0)
    dev = torch.device('cuda:0')
    kw = dict(emb_size=20, question_out=48, visual_out=64, attention_dim=40, fusion_in=32, fusion_hidden=16)
    tr = drv.SynthVQADataset(4, 32, 16, 64, 8, 50, 21, dev, seed=1)
    va = drv.SynthVQADataset(1, 32, 16, 64, 8, 50, 21, dev, seed=2)
    m = VQAAttnModel(['w'] * 50, ['a'] * 21, args, dtype=torch.float32, dropout=0.0, **kw).to(dev)
    hist = drv.train(m, tr, va, 6, 3e-3, args, out_dir=str(tmp_path))
    assert hist[-1]['train_loss'] < hist[0]['train_loss']
    sd = torch.load(tmp_path / 'ModelRunningRSVQA_VQAAttnModel.pth', weights_only=True)
    assert 'linear_classif2.weight' in sd and 'seq2vec.rnn.weight_hh_l0' in sd
    m2 = VQAAttnModel(['w'] * 50, ['a'] * 21, args, dtype=torch.float32, dropout=0.0, seed=99, **kw)
    m2.load_state_dict(sd)
    m2 = m2.to(dev).eval()
    m.eval()
    q, a, f, t = va.batches[0]
    with torch.no_grad():
        assert torch.equal(m(f, q), m2(f, q))
    counts = hist[-1]['val']['count']
    assert sum(counts.values()) == 32
