"""Odd shapes through the whole step (fp32, against the oracle): a single sample, a single token, a single
region, every question all-pad, maximum-length questions, tiny class counts, an empty batch, region counts
and widths that are not multiples of the kernels' tile sizes."""
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _run(kw, B, R, L, hops=1, tokens=None, train=True):
    from vqa_amd.model import VQAAttnModel
    p = vqa_ref.init_params(seed=11, hops=hops, **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.float32, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], hops=hops,
                     seed=77)
    m.load_state_dict(p)
    m = m.to('cuda').train(train)
    b = vqa_ref.synthetic_batch(B, R, kw['Dv'], max(L, 3), kw['vq'], kw['C'], seed=B * 100 + R + L)
    if L < 3:                           # the synthetic generator draws lengths from {3..L}
        b['tokens'] = torch.randint(0, kw['vq'] + 1, (B, L), generator=torch.Generator().manual_seed(B + L))
    if tokens is not None:
        b['tokens'] = tokens
    seeds = [vqa_ref.site_seed(77, 1, s) for s in range(6)] if train else None
    lref, loss_ref, gref = vqa_ref.loss_and_grads(p, b['feats'], b['tokens'], b['answers'], None, hops, drop_seeds=seeds)
    _, _, g64 = vqa_ref.loss_and_grads({k: v.double() for k, v in p.items()}, b['feats'].double(), b['tokens'],
                                       b['answers'], None, hops, drop_seeds=seeds)
    logits = m(b['feats'].cuda(), b['tokens'].cuda())
    assert logits.shape == (B, kw['C'])
    assert rel_err(logits, lref) <= 1e-5
    torch.nn.functional.cross_entropy(logits, b['answers'].cuda()).backward()
    named = dict(m.named_parameters())
    for k in gref:
        if k.endswith('linear_a.bias') or float(g64[k].abs().max()) < 1e-12:
            continue                    # R = 1: the softmax is constant and the attention parameters get zero gradient
        noise = rel_err(gref[k], g64[k])
        assert rel_err(named[k].grad, g64[k]) <= max(2e-5, 4 * noise), (k, rel_err(named[k].grad, g64[k]), noise)
    return m


KW = dict(vq=40, C=13, E=20, Hq=40, Dv=72, A=24, F=40, Hf=16)


@pytest.mark.parametrize('B,R,L,hops', [(1, 9, 7, 1), (2, 1, 5, 1), (3, 7, 1, 1), (1, 1, 1, 2), (5, 197, 3, 1),
                                        (67, 3, 20, 2), (130, 50, 14, 1)])
def test_odd_shapes(B, R, L, hops):
    _run(KW, B, R, L, hops)


def test_every_question_all_pad_or_full_length():
    L = 9
    tok = torch.zeros((6, L), dtype=torch.long)
    tok[1::2] = torch.randint(1, KW['vq'] + 1, (3, L), generator=torch.Generator().manual_seed(1))   # no trailing pad at all
    _run(KW, 6, 11, L, tokens=tok)
    _run(KW, 4, 5, L, tokens=torch.zeros((4, L), dtype=torch.long), train=False)


def test_minimal_widths():
    _run(dict(vq=3, C=2, E=8, Hq=8, Dv=8, A=8, F=8, Hf=8), 4, 4, 4)


def test_empty_batch_returns_empty_logits():
    from vqa_amd.model import VQAAttnModel
    kw = KW
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.float32, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf']).to('cuda').eval()
    out = m(torch.zeros((0, 5, kw['Dv']), device='cuda'), torch.zeros((0, 7), dtype=torch.long, device='cuda'))
    assert out.shape == (0, kw['C'])


def test_rejected_inputs_raise():
    from vqa_amd import _lib
    from vqa_amd.model import VQAAttnModel
    kw = KW
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.float32, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf']).to('cuda')
    with pytest.raises(_lib.VqaHipError):                       # no CPU path
        m(torch.zeros((2, 5, kw['Dv'])), torch.zeros((2, 7), dtype=torch.long))
    with pytest.raises(ValueError):                             # wrong feature width
        m(torch.zeros((2, 5, kw['Dv'] + 8), device='cuda'), torch.zeros((2, 7), dtype=torch.long, device='cuda'))
    with pytest.raises(ValueError):                             # batch mismatch
        m(torch.zeros((2, 5, kw['Dv']), device='cuda'), torch.zeros((3, 7), dtype=torch.long, device='cuda'))


def test_out_of_range_ids_raise_at_the_next_sync():
    """nn.Embedding raises for a token id outside its table (seq2vec.py:42) and CrossEntropyLoss for a target outside
    [0, C) (train.py:201).  The kernels cannot raise: they set a device-side status word (a bad target also makes its
    row's loss NaN) and the module raises IndexError at the next sync() / state_dict()."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.float32, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=77).to('cuda')
    b = vqa_ref.synthetic_batch(4, 9, kw['Dv'], 7, kw['vq'], kw['C'], seed=1)
    f, t, a = b['feats'].cuda(), b['tokens'].cuda(), b['answers'].cuda()
    m.train_step(f, t, a)
    m.sync()                                           # clean inputs: nothing raised
    bad_t = t.clone()
    bad_t[1, 0] = kw['vq'] + 1                         # the table has vq + 1 rows: ids 0 .. vq
    m.eval()
    with torch.no_grad():
        m(f, bad_t)
    with pytest.raises(IndexError, match='question token'):
        m.sync()
    m.sync()                                           # the words were cleared by the check
    bad_a = a.clone()
    bad_a[2] = kw['C']
    m.train()
    loss = m.train_step(f, t, bad_a)
    assert torch.isnan(loss).item()
    with pytest.raises(IndexError, match='Target out of bounds'):
        m.state_dict()
    bad_a[2] = -1
    m2 = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.bfloat16, emb_size=kw['E'], question_out=kw['Hq'],
                      visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=77).to('cuda')
    m2.train_step(f, t, bad_a)
    with pytest.raises(IndexError, match='Target out of bounds'):
        m2.sync()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bad_target_leaves_weights_and_moments_untouched(dtype):
    """CrossEntropyLoss raises for a target outside [0, C) at the loss call, before backward() and optimizer.step()
    (train.py:201-204): the reference's weights survive the bad batch.  Here the error surfaces only at the next
    check, so the fused Adam kernel must skip every update from the bad step on (device-side status word); after
    the IndexError the weights, the bf16 copies, both moments and the step counter are those of the last good step,
    and training continues from there exactly like a run that never saw the bad batch.  -100 (torch's
    ignore_index) is out of range like any other negative id."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)

    def make():
        return VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                            visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                            seed=5).to('cuda').train()
    b = vqa_ref.synthetic_batch(6, 9, kw['Dv'], 7, kw['vq'], kw['C'], seed=3)
    f, t, a = b['feats'].cuda().to(dtype), b['tokens'].cuda(), b['answers'].cuda()
    for bad_id in (kw['C'] + 5, -100):
        m, ref = make(), make()
        for mm in (m, ref):
            mm.train_step(f, t, a, lr=1e-2)
            mm.train_step(f, t, a, lr=1e-2)
            mm.sync()
        before = {k: v.clone() for k, v in m.state_dict().items()}
        opt_before = m.optimizer_state_dict()
        shadow_before = None if m._shadow is None else m._shadow.clone()
        bad_a = a.clone()
        bad_a[4] = bad_id
        loss = m.train_step(f, t, bad_a, lr=1e-2).clone()     # (the returned tensor lives in the workspace)
        m.train_step(f, t, a, lr=1e-2)            # a clean batch before the host has looked: skipped too
        assert torch.isnan(loss).item()
        with pytest.raises(IndexError, match='Target out of bounds'):
            m.sync()
        after = m.state_dict()
        for k in before:
            assert torch.isfinite(after[k]).all(), k
            assert torch.equal(after[k], before[k]), k
        opt_after = m.optimizer_state_dict()
        assert opt_after['step_count'] == opt_before['step_count'] == 2
        for k in opt_before['exp_avg']:
            assert torch.equal(opt_after['exp_avg'][k], opt_before['exp_avg'][k]), k
            assert torch.equal(opt_after['exp_avg_sq'][k], opt_before['exp_avg_sq'][k]), k
        if shadow_before is not None:
            assert torch.equal(m._shadow, shadow_before)
        # the run goes on from the last good step: same weights as the run that never saw the bad batch
        m.drop_step = ref.drop_step             # (the two skipped steps advanced the dropout stream)
        m.train_step(f, t, a, lr=1e-2)
        ref.train_step(f, t, a, lr=1e-2)
        m.sync(); ref.sync()
        sa, sb = m.state_dict(), ref.state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize('arch', ['lstm', '2-lstm'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_one_and_two_token_questions(arch, dtype):
    """Questions of one, two and three token slots (the tabulated first two steps of the frozen encoder have nothing
    or one step after them), with an all-pad row and a row of length 1, through the encoder and the whole forward,
    with and without the per-token tables (seq2vec.py:11-25, 40-45)."""
    from oracle import vqa_ref
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=50, C=11, E=20, Hq=48, Dv=64, A=16, F=16, Hf=8)
    p = vqa_ref.init_params(seed=3, two_lstm=(arch == '2-lstm'), **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=1,
                     seq2vec_arch=arch)
    m.load_state_dict(p)
    m = m.to('cuda').eval()
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for L in (1, 2, 3):
        g = torch.Generator().manual_seed(L)
        tok = torch.randint(1, kw['vq'], (9, L), generator=g)
        tok[2] = 0
        if L > 1:
            tok[4, 1:] = 0
        ref = vqa_ref.encode_question(tok, po)
        for fold in (True, False):
            m.fold_embedding = m.fold_first_step = fold
            q = m.encode_question(tok.cuda()).cpu()
            assert float((q - ref).abs().max()) <= tol * float(ref.abs().max()), (L, fold)
        feats = torch.randn(9, 5, kw['Dv'], generator=g).clamp_min(0).to(dtype)
        with torch.no_grad():
            out = m(feats.cuda(), tok.cuda()).float().cpu()
        want, _, _ = vqa_ref.loss_and_grads(po, feats.float(), tok, torch.zeros(9, dtype=torch.long), None, 1)
        assert float((out - want).abs().max()) <= tol * float(want.abs().max()), L
