"""The CPU oracle against the committed fixtures: G2 comes from the reference's own
seq2vec sub-modules, G3/G5/G6 pin the oracle against silent edits and against torch's
stock ops (the ones the reference calls)."""
import os

import numpy as np
import torch

from oracle import vqa_ref


def test_question_encoder_matches_reference_modules(golden_dir):
    z = np.load(os.path.join(golden_dir, 'g2_seq2vec.npz'))
    emb = torch.from_numpy(z['embedding'])
    assert float(emb[0].abs().max()) == 0.0                       # padding_idx row
    w = [torch.from_numpy(z[k]) for k in ('w_ih', 'w_hh', 'b_ih', 'b_hh')]
    for name in ('main', 'rand'):
        ids = torch.from_numpy(z[f'{name}_ids'])
        lengths = vqa_ref.process_lengths(ids)
        assert np.array_equal(lengths.numpy(), z[f'{name}_lengths'])
        e = vqa_ref.embedding_gather(ids, emb)
        assert np.array_equal(e.numpy(), z[f'{name}_emb'])         # gather is exact
        hs = vqa_ref.lstm_time_major(e, *w)
        assert np.allclose(hs.numpy(), z[f'{name}_hs'], rtol=0, atol=2e-6)
        last = vqa_ref.select_last(hs, lengths)
        assert np.allclose(last.numpy(), z[f'{name}_last'], rtol=0, atol=2e-6)
    # an all-pad row has length 0 and selects the last step (negative index upstream)
    assert int(z['rand_lengths'][5]) == 0
    assert np.array_equal(z['rand_last'][5], z['rand_hs'][5, -1])


def test_reference_forward_as_written_recurs_over_the_batch(golden_dir):
    """Documents the quirk: the committed forward feeds (B, L, E) to a time-major LSTM, so rows 1 and 2
    of the repo's own smoke input (identical questions) come out different."""
    z = np.load(os.path.join(golden_dir, 'g2_seq2vec.npz'))
    assert np.array_equal(z['main_ids'][1], z['main_ids'][2])
    assert not np.allclose(z['main_as_written'][1], z['main_as_written'][2])
    assert np.allclose(z['main_last'][1], z['main_last'][2])


def test_model_fixtures_reproduce(golden_dir):
    for tag, hops in (('h1', 1), ('h2', 2), ('h1cap', 1)):
        z = np.load(os.path.join(golden_dir, f'g3_model_{tag}.npz'))
        p = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('param:')}
        cap = torch.from_numpy(z['in:captions']) if 'in:captions' in z.files else None
        args = (p, torch.from_numpy(z['in:feats']), torch.from_numpy(z['in:tokens']), torch.from_numpy(z['in:answers']), cap, hops)
        logits, loss, grads = vqa_ref.loss_and_grads(*args)
        assert np.allclose(logits.numpy(), z['logits'], rtol=1e-5, atol=1e-6)
        assert np.allclose(loss.numpy(), z['loss'], rtol=1e-6)
        for k, g in grads.items():
            assert np.allclose(g.numpy(), z[f'grad:{k}'], rtol=1e-4, atol=1e-7), k
        logits_d, loss_d, _ = vqa_ref.loss_and_grads(*args, drop_seeds=[int(s) for s in z['drop_seeds']])
        assert np.allclose(logits_d.numpy(), z['logits_drop'], rtol=1e-5, atol=1e-6)


def test_uniform_attention_is_mean_pooling():
    """w_a = 0 => softmax is uniform => pooled == torch.mean over the grid (model.py:161-164)."""
    p = vqa_ref.init_params(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8, seed=0)
    p['attention.0.linear_a.weight'].zero_()
    b = vqa_ref.synthetic_batch(5, 12, 32, 7, 30, 11, seed=1)
    _, aux = vqa_ref.forward(p, b['feats'], b['tokens'], return_aux=True)
    nchw = aux['v_hat'].permute(0, 2, 1).reshape(5, 32, 3, 4)
    assert torch.allclose(aux['pooled'], torch.mean(nchw, axis=(2, 3)), atol=1e-7)
    assert torch.allclose(aux['att'][0], torch.full((5, 12), 1 / 12.0), atol=1e-7)


def test_head_matches_torch_modules():
    """The head restatement against torch.nn.Linear / Tanh composed as model.py:62-77 composes them."""
    kw = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)
    p = vqa_ref.init_params(seed=2, **kw)
    b = vqa_ref.synthetic_batch(6, 9, 32, 7, 30, 11, seed=3)
    logits, aux = vqa_ref.forward(p, b['feats'], b['tokens'], return_aux=True)
    lin = {}
    for n, (i, o) in (('linear_v', (32, 16)), ('linear_q', (24, 16)), ('linear_classif1', (16, 8)), ('linear_classif2', (8, 11))):
        m = torch.nn.Linear(i, o)
        m.load_state_dict({'weight': p[n + '.weight'], 'bias': p[n + '.bias']})
        lin[n] = m
    x_v = torch.nn.Tanh()(lin['linear_v'](aux['pooled']))
    x_q = torch.nn.Tanh()(lin['linear_q'](aux['q']))
    x = torch.nn.Tanh()(torch.mul(x_v, x_q))
    x = torch.nn.Tanh()(lin['linear_classif1'](x))
    assert torch.allclose(lin['linear_classif2'](x), logits, atol=1e-6)


def test_caption_vector_matches_torch_ops():
    g = torch.Generator().manual_seed(0)
    table = torch.randn(21, 8, generator=g); table[0] = 0
    cap = torch.tensor([[3, 4, 5, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0], [7, 8, 9, 10, 11, 12]])
    emb = table[cap]
    mask = cap.ne(0)
    me = mask.unsqueeze(-1).expand(emb.size()).float()                       # model.py:1573-1574
    ref = torch.nn.functional.normalize(torch.sum(emb * me, 1) / torch.clamp(me.sum(1), min=1e-9), p=2, dim=1)
    assert torch.allclose(vqa_ref.caption_vector(cap, table), ref, atol=1e-7)


def test_loss_metrics_and_dropout_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, 'g5_ce_g6_metrics.npz'))
    lg, y = torch.from_numpy(z['logits']), torch.from_numpy(z['target'])
    assert np.isfinite(z['loss'])
    assert abs(float(vqa_ref.cross_entropy(lg, y)) - float(torch.nn.CrossEntropyLoss()(lg, y))) <= 1e-6 * float(z['loss'])
    lg2 = lg.clone().requires_grad_(True)
    torch.nn.CrossEntropyLoss()(lg2, y).backward()
    assert np.allclose(vqa_ref.cross_entropy_grad(lg, y).numpy(), lg2.grad.numpy(), atol=1e-7)
    assert np.allclose(vqa_ref.cross_entropy_grad(lg, y).numpy(), z['dlogits'], atol=1e-7)
    assert np.array_equal(vqa_ref.argmax_first(z['logits']), z['argmax'])
    assert z['argmax'][2] == 4                                      # exact tie: first maximum wins
    assert np.array_equal(vqa_ref.argmax_first(z['logits_nan']), z['argmax_nan']) and z['argmax_nan'][4] == 7
    count, right = vqa_ref.accuracy_counts(z['argmax'], z['target'], z['types'])
    assert np.array_equal(count, z['count']) and np.array_equal(right, z['right']) and count[3] == 0
    oa, aa = vqa_ref.oa_aa(count, right)
    assert oa == float(z['oa']) and aa == float(z['aa'])
    assert np.array_equal(vqa_ref.dropout_keep_mask(77, 64, 0.5), z['dropout_mask_seed77_p05'])
    assert np.array_equal(vqa_ref.dropout_keep_mask(5, 64, 0.3), z['dropout_mask_seed5_p03'])
    keep = vqa_ref.dropout_keep_mask(123, 200000, 0.5)
    assert abs(keep.mean() - 0.5) < 5e-3


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(50, generator=g)
    wt = w.clone().requires_grad_(True)
    opt = torch.optim.Adam([wt], lr=1e-3)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    for step in range(1, 6):
        grad = torch.randn(50, generator=g)
        wt.grad = grad.clone()
        opt.step()
        vqa_ref.adam_step(w, grad, m, v, step, 1e-3)
    assert torch.allclose(w, wt.detach(), atol=1e-7)
