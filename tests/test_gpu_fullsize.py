"""Full-size (BASELINE config) GPU checks through size-independent properties, plus oracle parity on
sub-batches at the real model widths (Dv=2048, A=F=1200, Hq=2400, E=620, Hf=256, C=3000, L=14)."""
import numpy as np
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu

FULL = dict(vq=10000, C=3000, E=620, Hq=2400, Dv=2048, A=1200, F=1200, Hf=256)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


@pytest.fixture(scope='module')
def full_model():
    from vqa_amd.model import VQAAttnModel
    p = vqa_ref.init_params(hops=1, seed=11, **FULL)
    m = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, seed=5)
    m.load_state_dict(p)
    return m.to('cuda'), p


def test_cfg2_rows_match_oracle_and_are_independent(full_model):
    """B=256, R=196 (config 2).  Rows are independent in every kernel (what makes data-parallel sharding
    exact): the logits of rows 0..3 inside the 256-batch equal the oracle's logits for those 4 rows alone,
    and equal the module's own logits when the same rows are fed as a batch of 4."""
    m, p = full_model
    m.eval()
    b = vqa_ref.synthetic_batch(256, 196, 2048, 14, FULL['vq'], FULL['C'], seed=42)
    feats = b['feats'].to(torch.bfloat16)
    logits = m(feats.cuda(), b['tokens'].cuda()).float().cpu()
    assert logits.shape == (256, 3000) and bool(torch.isfinite(logits).all())
    po = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref = vqa_ref.forward(po, feats[:4].float(), b['tokens'][:4])
    assert rel_err(logits[:4], ref) <= 1e-2
    small = m(feats[:4].cuda(), b['tokens'][:4].cuda()).float().cpu()
    assert rel_err(small, logits[:4]) <= 2e-3          # same rows, different tiling of the batch
    # attention probabilities are a distribution over the 196 regions
    ps = m._ctx[0]['p0']
    ws4 = m._workspace(4, 196, 14, 0, torch.device('cuda', 0))['p0']
    torch.cuda.synchronize()
    assert torch.allclose(ws4.sum(1).cpu(), torch.ones(4), atol=1e-5)
    # prediction: bit-exact first-max argmax of the logits the module produced
    pred = m.predict().cpu().numpy()
    lg = m._ctx[0]['logits'][:, :3000].float().cpu().numpy()
    assert np.array_equal(pred, np.argmax(lg, axis=1))


def test_cfg2_region_permutation_invariance(full_model):
    """Attention pooling is a sum over regions: permuting the 196 regions of every sample leaves the logits
    unchanged up to summation order."""
    m, _ = full_model
    m.eval()
    b = vqa_ref.synthetic_batch(64, 196, 2048, 14, FULL['vq'], FULL['C'], seed=7)
    feats = b['feats'].to(torch.bfloat16).cuda()
    tok = b['tokens'].cuda()
    a = m(feats, tok).float()
    perm = torch.randperm(196, generator=torch.Generator().manual_seed(0)).cuda()
    c = m(feats[:, perm].contiguous(), tok).float()
    assert rel_err(c, a) <= 2e-3


def _check_step_against_oracle(m, p, b, hops=1, captions=False):
    """fwd + CE + bwd of the module (eval mode, bf16 storage) against the oracle's autograd on the same batch: loss
    to 1e-2, every gradient to max(2e-2, 4 x the fp32 oracle's own distance from its fp64 run) of its largest
    magnitude -- the rule of the small-shape tests."""
    m.eval()
    feats = b['feats'].to(torch.bfloat16)
    cap = b['captions'] if captions else None
    po = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    logits_ref, loss_ref, grads = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], cap, hops)
    _, _, grads64 = vqa_ref.loss_and_grads({k: v.double() for k, v in po.items()}, feats.double(), b['tokens'],
                                           b['answers'], cap, hops)
    with torch.no_grad():
        m._bptt = False
        m._forward_impl(*m._prepare_inputs(feats.cuda(), b['tokens'].cuda(), cap.cuda() if captions else None))
        loss = m.loss_and_grad_logits(b['answers'].cuda())
        m._backward_impl()
    C = logits_ref.shape[1]
    assert rel_err(m._ctx[0]['logits'][:, :C], logits_ref) <= 1e-2
    assert abs(float(loss) - float(loss_ref)) <= 1e-2 * float(loss_ref)
    named = dict(m.named_parameters())
    checked = []
    for k, g in grads.items():
        if k.endswith('linear_a.bias'):
            continue                      # the score offset cancels in the softmax: rounding noise around zero
        got = m._g(k).view(named[k].shape)
        noise = rel_err(g, grads64[k])
        err = rel_err(got, grads64[k])
        assert err <= max(2e-2, 4 * noise), (k, err, noise)
        checked.append(k)
    return checked


def test_cfg2_train_step_gradients_subbatch(full_model):
    """fwd + CE + bwd at the real widths on a batch of 8 against the oracle's autograd."""
    m, p = full_model
    b = vqa_ref.synthetic_batch(8, 196, 2048, 14, FULL['vq'], FULL['C'], seed=3)
    checked = _check_step_against_oracle(m, p, b)
    assert 'attention.0.linear_va.weight' in checked and 'linear_classif2.weight' in checked


def test_cfg2_full_batch_gradients_at_the_benched_shape(full_model):
    """The exact shape bench.py times (B = 256, R = 196: the 256 x 256-tile projection over 50 176 rows and its
    weight gradient over a 50 176-long contraction, split over K ranges and transposed back by the reduce) against
    the oracle's autograd on the same 256 samples: logits, loss, and the gradients of the attention weights."""
    m, p = full_model
    m.eval()
    b = vqa_ref.synthetic_batch(256, 196, 2048, 14, FULL['vq'], FULL['C'], seed=42)
    feats = b['feats'].to(torch.bfloat16)
    po = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    logits_ref, loss_ref, grads = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], None, 1)
    with torch.no_grad():
        m._bptt = False
        m._forward_impl(*m._prepare_inputs(feats.cuda(), b['tokens'].cuda(), None))
        loss = m.loss_and_grad_logits(b['answers'].cuda())
        m._backward_impl()
    assert rel_err(m._ctx[0]['logits'][:, :3000], logits_ref) <= 1e-2
    assert abs(float(loss) - float(loss_ref)) <= 1e-2 * float(loss_ref)
    named = dict(m.named_parameters())
    for k in ('attention.0.linear_va.weight', 'attention.0.linear_va.bias', 'attention.0.linear_qa.weight',
              'attention.0.linear_a.weight', 'linear_v.weight', 'linear_classif2.weight'):
        got = m._g(k).view(named[k].shape)
        assert rel_err(got, grads[k]) <= 2e-2, (k, rel_err(got, grads[k]))


def test_cfg4_caption_stream_gradients_full_width():
    """Config 4 at the real widths, forward AND backward on a batch of 8: masked mean + L2 of the caption embeddings,
    concatenation, the widened linear_classif1 (model.py:1571-1574, 1678, 1734-1741)."""
    from vqa_amd.model import VQAAttnModel
    p = vqa_ref.init_params(hops=1, seed=13, vc=5000, Ec=384, **FULL)
    m = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, caption_vocab=5000,
                     caption_dim=384, seed=5)
    m.load_state_dict(p)
    m = m.to('cuda')
    b = vqa_ref.synthetic_batch(8, 196, 2048, 14, FULL['vq'], FULL['C'], seed=4, Tc=16, vc=5000)
    b['captions'][5] = 0                                  # one sample without a caption (all padding)
    checked = _check_step_against_oracle(m, p, b, captions=True)
    assert 'linear_classif1.weight' in checked and p['linear_classif1.weight'].shape == (256, 1200 + 384)


def test_cfg5_two_hops_gradients_full_width():
    """Config 5 geometry (24 x 24 regions, 2 stacked hops) at the real widths on a batch of 3: every gradient,
    linear_u of the first hop included."""
    from vqa_amd.model import VQAAttnModel
    p = vqa_ref.init_params(hops=2, seed=12, **FULL)
    m = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, hops=2, seed=5)
    m.load_state_dict(p)
    m = m.to('cuda')
    b = vqa_ref.synthetic_batch(3, 576, 2048, 14, FULL['vq'], FULL['C'], seed=21)
    checked = _check_step_against_oracle(m, p, b, hops=2)
    assert 'attention.0.linear_u.weight' in checked and 'attention.1.linear_va.weight' in checked


def test_cfg5_large_grid_two_hops_runs_and_matches_subbatch():
    """Config 5 shape: 24x24 regions, 2 stacked hops, batch 512 per GPU; rows 0..1 against the oracle."""
    from vqa_amd.model import VQAAttnModel
    p = vqa_ref.init_params(hops=2, seed=12, **FULL)
    m = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, hops=2, seed=5)
    m.load_state_dict(p)
    m = m.to('cuda').train()
    g = torch.Generator(device='cuda').manual_seed(1)
    B, R = 512, 576
    feats = torch.randn((B, R, 2048), device='cuda', generator=g).clamp_min_(0).to(torch.bfloat16)
    b = vqa_ref.synthetic_batch(B, 1, 8, 14, FULL['vq'], FULL['C'], seed=9)
    loss0 = float(m.train_step(feats, b['tokens'].cuda(), b['answers'].cuda(), lr=1e-4))
    for _ in range(3):
        loss = float(m.train_step(feats, b['tokens'].cuda(), b['answers'].cuda(), lr=1e-4))
    assert np.isfinite(loss0) and np.isfinite(loss) and loss < loss0      # same batch: loss falls
    m.eval()
    logits = m(feats, b['tokens'].cuda()).float().cpu()
    po = {k: v.detach().float().cpu().to(torch.bfloat16).float() for k, v in m.state_dict().items()}
    ref = vqa_ref.forward(po, feats[:2].float().cpu(), b['tokens'][:2], hops=2)
    assert rel_err(logits[:2], ref) <= 1e-2


def test_cfg4_caption_stream_full_width():
    """Config 4: caption stream (16-token captions, 384-d caption vectors) at the real widths."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(FULL)
    p = vqa_ref.init_params(hops=1, seed=13, vc=5000, Ec=384, **kw)
    m = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, caption_vocab=5000,
                     caption_dim=384, seed=5)
    m.load_state_dict(p)
    m = m.to('cuda').eval()
    b = vqa_ref.synthetic_batch(32, 196, 2048, 14, FULL['vq'], FULL['C'], seed=4, Tc=16, vc=5000)
    feats = b['feats'].to(torch.bfloat16)
    logits = m(feats.cuda(), b['tokens'].cuda(), b['captions'].cuda()).float().cpu()
    po = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    ref = vqa_ref.forward(po, feats[:4].float(), b['tokens'][:4], b['captions'][:4])
    assert rel_err(logits[:4], ref) <= 1e-2


def test_cfg2_fast_paths_agree_with_plain_paths(full_model):
    """Config 2, one train-mode forward + backward with every fused fast path on (row norms inside the projection
    GEMM, the big weight gradient in the swapped orientation, grouped small products, partial sums reduced in the
    consumers' prologues, the per-token table of input projections) against the same step with the separate-kernel paths: same loss and gradients to bf16
    rounding of the intermediate activations (the fused norm sums its squares in a different order)."""
    from vqa_amd.model import VQAAttnModel
    m, p = full_model
    b = vqa_ref.synthetic_batch(256, 196, 2048, 14, FULL['vq'], FULL['C'], seed=77)
    feats, tok, ans = b['feats'].to(torch.bfloat16).cuda(), b['tokens'].cuda(), b['answers'].cuda()
    grads = []
    for fast in (True, False):
        mm = VQAAttnModel(['w'] * FULL['vq'], ['a'] * FULL['C'], None, dtype=torch.bfloat16, seed=5)
        mm.load_state_dict(p)
        mm = mm.to('cuda').train()
        mm.fuse_row_norm = fast
        mm.swap_big_wgrad = fast
        mm._dpool_from_slabs = fast
        mm.fold_embedding = mm.fold_first_step = fast
        if not fast:
            mm._launch_early = lambda ws, B: ws.__setitem__('early', {})

            def plain_wgrads(items, colsums=None, mm=mm):
                from vqa_amd import _lib
                if colsums:
                    _lib.colsum_multi(colsums)
                for it in items:
                    mm._wgrad(*it, defer=True)
            mm._wgrad_grouped = plain_wgrads
        logits = mm(feats, tok)
        loss = torch.nn.functional.cross_entropy(logits.float(), ans)
        loss.backward()
        torch.cuda.synchronize()
        grads.append((float(loss.detach()), {k: v.grad.detach().float().cpu().clone() for k, v in mm.named_parameters()
                                     if v.grad is not None}))
    (la, ga), (lb, gb) = grads
    assert abs(la - lb) <= 1e-3 * abs(lb)
    assert set(ga) == set(gb) and len(ga) >= 14
    for k in ga:
        if k.endswith('linear_a.bias'):
            continue                                  # a sum of zero-mean rounding noise
        num = float((ga[k] - gb[k]).norm())
        den = float(gb[k].norm())
        assert num <= 2e-2 * den + 1e-12, (k, num, den)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gap_model_reference_widths(dtype):
    """VQAGAPModel at the reference's own widths (model.py:81-184: ResNet-152 stage maps of a 512-pixel image,
    256 / 512 / 1024 / 2048 channels on 128 / 64 / 32 / 16 grids, VISUAL_OUT_GAP = 3840, QUESTION_OUT = 2400,
    FUSION_IN = 1200, FUSION_HIDDEN = 256), channels-last maps, evaluation and training mode, against the oracle."""
    from tests.test_gpu_frontends import _gap_model, _check
    kw = dict(vq=300, C=95, E=620, Hq=2400, Dv=3840, A=8, F=1200, Hf=256)
    channels, grids, B = (256, 512, 1024, 2048), (128, 64, 32, 16), 5
    m, p = _gap_model(kw, channels, dtype)
    maps = [x.to(dtype) for x in vqa_ref.synthetic_maps(B, channels, grids, seed=4)]
    b = vqa_ref.synthetic_batch(B, 4, 8, 14, kw['vq'], kw['C'], seed=6)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    gpu_maps = [x.cuda().contiguous(memory_format=torch.channels_last) for x in maps]
    ref_maps = [x.float() for x in maps]
    _check(m, po, ref_maps, gpu_maps, b, 'gap', dtype, train=False)
    _check(m, po, ref_maps, gpu_maps, b, 'gap', dtype, train=True)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_conv_model_reference_widths(dtype):
    """VQAModel at the reference's widths (model.py:29-79): the last ResNet map of a 512-pixel image (2048 channels on a
    16 x 16 grid), the 1 x 1 convolution down to VISUAL_OUT / 256 = 8 channels, NCHW flatten to VISUAL_OUT = 2048."""
    from tests.test_gpu_frontends import _check
    from vqa_amd.model import VQAModel
    kw = dict(vq=300, C=95, E=620, Hq=2400, Dv=2048, A=8, F=1200, Hf=256)
    input_size, B, Cin = 512, 6, 2048
    g = input_size // 32
    c_out = kw['Dv'] // (g * g)
    p = vqa_ref.init_params(seed=3, mode='conv', conv_out=c_out, **{**kw, 'Dv': Cin})
    p['linear_v.weight'] = vqa_ref.init_params(seed=4, **kw)['linear_v.weight']
    m = VQAModel(['w'] * kw['vq'], ['a'] * kw['C'], None, input_size, dtype=dtype, emb_size=kw['E'],
                 question_out=kw['Hq'], visual_out=kw['Dv'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                 map_channels=Cin, seed=1234)
    m.load_state_dict(p)
    m = m.to('cuda')
    fmap = vqa_ref.synthetic_maps(B, (Cin,), (g,), seed=11)[0].to(dtype)
    b = vqa_ref.synthetic_batch(B, 4, 8, 14, kw['vq'], kw['C'], seed=6)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    _check(m, po, fmap.float(), fmap.cuda(), b, 'conv', dtype, train=False)
    _check(m, po, fmap.float(), fmap.cuda(), b, 'conv', dtype, train=True)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_stack_model_reference_widths(dtype):
    """The max / mean stack model (model.py:1234-1241) on the four ResNet-152 stage maps of a 512-pixel image with a
    2048-wide stack."""
    from tests.test_gpu_frontends import _check
    from vqa_amd.model import VQAGAPStackModel
    kw = dict(vq=300, C=95, E=620, Hq=2400, Dv=2048, A=8, F=1200, Hf=256)
    channels, grids, B = (256, 512, 1024, 2048), (128, 64, 32, 16), 5
    p = vqa_ref.init_params(seed=3, mode='stack', gap_channels=channels, **kw)
    m = VQAGAPStackModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                         fusion_in=kw['F'], fusion_hidden=kw['Hf'], gap_channels=channels, stack_dim=kw['Dv'], seed=1234)
    m.load_state_dict(p)
    m = m.to('cuda')
    maps = [x.to(dtype) for x in vqa_ref.synthetic_maps(B, channels, grids, seed=9)]
    b = vqa_ref.synthetic_batch(B, 4, 8, 14, kw['vq'], kw['C'], seed=5)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    _check(m, po, [x.float() for x in maps], [x.cuda() for x in maps], b, 'stack', dtype, train=False)
    _check(m, po, [x.float() for x in maps], [x.cuda() for x in maps], b, 'stack', dtype, train=True)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_two_lstm_encoder_reference_widths(dtype):
    """The attention model with the 2-lstm question encoder (seq2vec.py:48-76) at full width (two stacked layers of
    1200, dropout 0.3 on each vector), evaluation and training mode, logits and every gradient against the oracle."""
    from tests.test_gpu_model import rel_err, oracle_params_as
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=500, C=300, E=620, Hq=2400, Dv=2048, A=1200, F=1200, Hf=256)
    p = vqa_ref.init_params(seed=3, two_lstm=True, **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                     seed=1234, seq2vec_arch='2-lstm')
    m.load_state_dict(p)
    m = m.to('cuda')
    b = vqa_ref.synthetic_batch(12, 49, kw['Dv'], 14, kw['vq'], kw['C'], seed=5)
    b['tokens'][3] = 0
    feats = b['feats'].to(dtype)
    po = oracle_params_as(p, dtype)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for train in (False, True):
        m.train(train)
        seeds = [vqa_ref.site_seed(1234, m.drop_step + 1, s) for s in range(6)] if train else None
        logits_ref, _, grads_ref = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], None, 1, drop_seeds=seeds)
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
            assert rel_err(named[k].grad, grads64[k]) <= max(2 * tol, 4 * noise), (k, train, rel_err(named[k].grad, grads64[k]), noise)


@pytest.mark.parametrize('dtype,two', [(torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)])
def test_trainable_encoder_reference_widths(dtype, two):
    """Backward through time at full width (E = 620, 2400 hidden units; the 2-lstm encoder as two layers of 1200):
    every gradient, the embedding table's included, against the oracle's autograd, evaluation and training mode."""
    from tests.test_gpu_encoder_train import _model, _batch, _compare
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=200, C=95, E=620, Hq=2400, Dv=2048, A=1200, F=1200, Hf=256)
    m, p = _model(VQAAttnModel, kw, dtype, two, visual_out=kw['Dv'], attention_dim=kw['A'], hops=1)
    b = _batch(kw, 10, 16, 14, seed=5)
    feats = b['feats'].to(dtype)
    po = p if dtype == torch.float32 else {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=False)
    _compare(m, po, feats.float(), feats.cuda(), b, dtype, train=True)
