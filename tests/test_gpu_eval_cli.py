"""Evaluation driver (the counterpart of vqa_model/test.py) and caption strings through the module, on the GPU."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu

KW = dict(emb_size=20, question_out=48, visual_out=64, attention_dim=40, fusion_in=32, fusion_hidden=16)


def test_eval_driver_prints_the_reference_lines(tmp_path):
    """test(): loads ModelBestRSVQA_<model>.pth, one eval pass, prints 'test loss', per-category accuracy, OA, AA
    (test.py:92-110); the figures equal numpy's on the module's own logits (argmax first-max, AA = sum / 4)."""
    from vqa_amd import test as ev, train as drv
    from vqa_amd.model import VQAAttnModel
    args = argparse.Namespace(model='VQAAttnModel', gpu=0)
    dev = torch.device('cuda:0')
    tr = drv.SynthVQADataset(4, 32, 16, 64, 8, 50, 21, dev, seed=1)
    va = drv.SynthVQADataset(1, 32, 16, 64, 8, 50, 21, dev, seed=2)
    m = VQAAttnModel(['w'] * 50, ['a'] * 21, args, dtype=torch.float32, dropout=0.0, **KW).to(dev)
    drv.train(m, tr, va, 3, 3e-3, args, out_dir=str(tmp_path))
    m2 = VQAAttnModel(['w'] * 50, ['a'] * 21, args, dtype=torch.float32, dropout=0.0, seed=5, **KW).to(dev)
    m2.load_state_dict(torch.load(tmp_path / 'ModelBestRSVQA_VQAAttnModel.pth', weights_only=True))
    te = drv.SynthVQADataset(3, 32, 16, 64, 8, 50, 21, dev, seed=3)
    te.batches[1][3][:] = 2                                   # one batch of a single question type
    lines = []
    res = ev.test(m2, te, args, 'HR', out=lines.append)
    assert lines[0].startswith('test loss: ') and lines[1] == 'Per category accuracy'
    assert lines[-2] == 'OA: %.3f' % res['OA'] and lines[-1] == 'AA: %.3f' % res['AA']
    # independent bookkeeping on the host (train.py:105-110, 147-158)
    m2.eval()
    count, right, loss_sum = np.zeros(4, np.int64), np.zeros(4, np.int64), 0.0
    with torch.no_grad():
        for q, a, f, t in te:
            lg = m2(f, q)
            loss_sum += float(torch.nn.functional.cross_entropy(lg, a)) * q.shape[0]
            c, r = vqa_ref.accuracy_counts(vqa_ref.argmax_first(lg.cpu().numpy()), a.cpu().numpy(), t.cpu().numpy())
            count += c
            right += r
    oa, aa = vqa_ref.oa_aa(count, right)
    assert [res['count'][k] for k in ('presence', 'count', 'comp', 'area')] == count.tolist()
    assert [res['right'][k] for k in ('presence', 'count', 'comp', 'area')] == right.tolist()
    assert res['OA'] == oa and res['AA'] == aa
    assert abs(res['loss'] - loss_sum / len(te)) <= 1e-5 * max(1.0, abs(res['loss']))
    per_type = [l for l in lines[2:-2]]
    assert len(per_type) == int((count > 0).sum())


def test_eval_cli_main(tmp_path, capsys):
    from vqa_amd import test as ev, train as drv
    args = argparse.Namespace(model='VQAGAPModel', gpu=0)
    model, geo = drv.model_and_geometry(args, torch.device('cuda:0'))
    torch.save(model.state_dict(), tmp_path / 'ModelBestRSVQA_VQAGAPModel.pth')
    del model
    res = ev.main(['--model', 'VQAGAPModel', '--dataset', 'SYNTH', '--bs', '8', '--gpu', '0', '--batches', '2',
                   '--models-dir', str(tmp_path)])
    out = capsys.readouterr().out
    assert 'test loss:' in out and 'Per category accuracy' in out and 'OA:' in out and 'AA:' in out
    assert sum(res['count'].values()) == 16


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_caption_strings_equal_caption_ids(dtype):
    """forward(image, question, captions) with the reference's collated caption strings gives the logits of the same
    captions passed as ids, and the oracle's."""
    from vqa_amd.captions import CaptionTokenizer
    from vqa_amd.model import VQAAttnModel
    words = [f'w{i}' for i in range(20)]
    kw = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)
    p = vqa_ref.init_params(hops=1, seed=3, vc=20, Ec=8, **kw)
    m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=dtype, emb_size=kw['E'], question_out=kw['Hq'],
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'],
                     caption_vocab=CaptionTokenizer(words, max_tokens=6), caption_dim=8, seed=1234)
    m.load_state_dict(p)
    m = m.to('cuda').eval()
    b = vqa_ref.synthetic_batch(4, 9, kw['Dv'], 7, kw['vq'], kw['C'], seed=5, Tc=6, vc=20)
    strings = tuple(' '.join(words[i - 1] for i in row if i > 0) + '.' for row in b['captions'].tolist())
    feats = b['feats'].to(dtype).cuda()
    with torch.no_grad():
        by_ids = m(feats, b['tokens'].cuda(), b['captions'].cuda())
        by_strings = m(feats, b['tokens'].cuda(), [strings])          # collated: element 0 is the tuple of B strings
    assert torch.equal(by_ids, by_strings)
    po = p if dtype == torch.float32 else {k: v.to(dtype).float() for k, v in p.items()}
    ref, _, _ = vqa_ref.loss_and_grads(po, feats.float().cpu(), b['tokens'], b['answers'], b['captions'], 1)
    err = float((by_strings.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err <= (1e-5 if dtype == torch.float32 else 1e-2)
