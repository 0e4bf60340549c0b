"""Trainable question encoder: backward through time of the packed LSTM layer(s), weight / bias
gradients, deterministic embedding gradient (pad row masked), against the oracle's autograd.  The
reference freezes its encoder (model.py:42-43); this is the autograd of seq2vec.py:40-45 / 61-76 that
un-freezing it would run."""
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu

MID = dict(vq=60, C=101, E=44, Hq=208, Dv=256, A=136, F=104, Hf=72)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _batch(kw, B, R, L, seed):
    b = vqa_ref.synthetic_batch(B, R, kw['Dv'], L, kw['vq'], kw['C'], seed=seed)
    b['tokens'][3] = 0                 # all-pad row: runs all L steps on the pad embedding
    b['tokens'][5, 1] = 0              # a pad id in the middle of a question: no embedding gradient from it
    return b


def _model(cls, kw, dtype, two, **extra):
    p = vqa_ref.init_params(seed=3, two_lstm=two, hops=extra.get('hops', 1), **kw, **extra.pop('init', {}))
    m = cls(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
            fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=1234, seq2vec_arch='2-lstm' if two else 'lstm',
            train_encoder=True, **extra)
    m.load_state_dict(p)
    return m.to('cuda'), p


def _compare(m, po, vis_ref, vis_gpu, b, dtype, train, hops=1, mode='attn'):
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    m.train(train)
    seeds = [vqa_ref.site_seed(1234, m.drop_step + 1, s) for s in range(6)] if train else None
    ref = vqa_ref.loss_and_grads(po, vis_ref, b['tokens'], b['answers'], None, hops, drop_seeds=seeds, mode=mode,
                                 train_encoder=True)
    to64 = (lambda v: [x.double() for x in v]) if isinstance(vis_ref, list) else (lambda v: v.double())
    ref64 = vqa_ref.loss_and_grads({k: v.double() for k, v in po.items()}, to64(vis_ref), b['tokens'], b['answers'],
                                   None, hops, drop_seeds=seeds, mode=mode, train_encoder=True)
    logits = m(vis_gpu, b['tokens'].cuda())
    assert rel_err(logits, ref[0]) <= tol
    m.zero_grad()
    torch.nn.functional.cross_entropy(logits.float(), b['answers'].cuda()).backward()
    named = dict(m.named_parameters())
    assert any(k.startswith('seq2vec.') for k in ref[2])
    for k, gref in ref[2].items():
        if k.endswith('linear_a.bias'):
            continue
        assert named[k].grad is not None, k
        noise = rel_err(gref, ref64[2][k])
        assert rel_err(named[k].grad, ref64[2][k]) <= max(2 * tol, 4 * noise), (k, train, rel_err(named[k].grad, ref64[2][k]))
    emb_grad = named['seq2vec.embedding.weight'].grad
    assert float(emb_grad[0].abs().max()) == 0.0             # padding_idx row


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('two,hops', [(False, 1), (True, 1), (False, 2), (True, 2)])
def test_attention_model_with_trainable_encoder(dtype, two, hops):
    from vqa_amd.model import VQAAttnModel
    kw = MID
    m, p = _model(VQAAttnModel, kw, dtype, two, visual_out=kw['Dv'], attention_dim=kw['A'], hops=hops)
    b = _batch(kw, 19, 25, 14, seed=5)
    feats = b['feats'].to(dtype)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=False, hops=hops)
    _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=True, hops=hops)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gap_model_with_trainable_encoder(dtype):
    from vqa_amd.model import VQAGAPModel
    kw, channels, grids = MID, (32, 64, 128, 256), (8, 4, 4, 2)
    m, p = _model(VQAGAPModel, kw, dtype, False, gap_channels=channels, init=dict(mode='gap', gap_channels=channels))
    b = _batch(kw, 17, 4, 14, seed=6)
    maps = [x.to(dtype) for x in vqa_ref.synthetic_maps(17, channels, grids, seed=9)]
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    _compare(m, po, [x.float() for x in maps], [x.cuda() for x in maps], b, dtype, train=True, mode='gap')


def test_train_steps_update_the_encoder_like_the_oracle():
    """Three fused steps with the encoder in the optimiser: every parameter, the embedding table and the
    LSTM weights included, follows the oracle's autograd + Adam."""
    from vqa_amd.model import VQAAttnModel
    kw = MID
    m, p = _model(VQAAttnModel, kw, torch.float32, False, visual_out=kw['Dv'], attention_dim=kw['A'])
    m.train()
    po = {k: v.clone() for k, v in p.items()}
    names = vqa_ref.trainable_names(po, train_encoder=True)
    mom = {k: torch.zeros_like(po[k]) for k in names}
    var = {k: torch.zeros_like(po[k]) for k in names}
    lr = 1e-3
    for step in range(1, 4):
        b = _batch(kw, 24, 25, 14, seed=100 + step)
        seeds = [vqa_ref.site_seed(1234, step, s) for s in range(6)]
        _, loss_ref, grads = vqa_ref.loss_and_grads(po, b['feats'], b['tokens'], b['answers'], None, 1, drop_seeds=seeds,
                                                    train_encoder=True)
        for k in names:
            vqa_ref.adam_step(po[k], grads[k], mom[k], var[k], step, lr)
        loss = m.train_step(b['feats'].cuda(), b['tokens'].cuda(), b['answers'].cuda(), lr=lr)
        assert abs(float(loss) - float(loss_ref)) <= 2e-5 * max(1.0, abs(float(loss_ref)))
    sd = m.state_dict()
    assert float((sd['seq2vec.rnn.weight_hh_l0'].cpu() - p['seq2vec.rnn.weight_hh_l0']).abs().max()) > 0
    for k in names:
        if k.endswith('linear_a.bias'):
            continue
        # Adam turns rounding-level gradients into +-lr steps: compare where the oracle's gradient is not noise
        assert rel_err(sd[k], po[k]) <= 5e-4, k


@pytest.mark.parametrize('arch', ['lstm', '2-lstm'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_folded_embedding_table_matches_per_step_projection(arch, dtype):
    """Frozen encoder: the per-token table of input projections (embedding folded into W_ih) gives the question
    vector of the gather + per-step projection path bit for bit, for ragged lengths, pad rows and ids outside the table."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(emb_size=20, question_out=48, visual_out=72, attention_dim=24, fusion_in=40, fusion_hidden=16)
    Vq, C, B, L = 37, 9, 29, 11
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(1, Vq + 1, (B, L), generator=g)
    lens = torch.randint(0, L + 1, (B,), generator=g)
    tok[torch.arange(L)[None, :] >= lens[:, None]] = 0
    tok[3, 0] = Vq + 5                                    # outside the table: reads the pad row
    outs = []
    for fold, first in ((True, False), (False, False), (True, True)):
        m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=dtype, seq2vec_arch=arch, seed=9, **kw).to('cuda').eval()
        m.fold_embedding, m.fold_first_step = fold, first
        q = m.encode_question(tok.cuda()).clone()
        torch.cuda.synchronize()
        assert (m._xg_table is not None) == fold and (m._hh_table is not None) == (fold and first)
        outs.append(q)
    assert torch.equal(outs[0], outs[1])
    # steps 0 and 1 in one launch against the same two steps launched separately: bit for bit
    m.fuse_first_two_cells = False
    q_sep = m.encode_question(tok.cuda()).clone()
    assert torch.equal(q_sep, outs[2])
    # the tabulated first recurrent product is one fp32 sum per element instead of a split-K sum: equal up to rounding
    err = float((outs[2] - outs[1]).abs().max() / outs[1].abs().max())
    assert err <= (1e-5 if dtype == torch.float32 else 1e-2), err
