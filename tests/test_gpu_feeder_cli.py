"""JSON files -> VocabEncoder -> FeatureVQALoader -> DataLoader -> collate -> HIP kernels -> AccuracyMeter, end to end
through the two command lines (vqa_amd.train / vqa_amd.test with --dataset LR), on a tiny RSVQA-LR-shaped dataset
written here under the reference's file names (train.py:260-271, 286-305; test.py:136-170).  The oracle is fed the
same encoded batches: logits, test loss and the per-type counters must agree."""
import numpy as np
import pytest
import torch

from _tiny_dataset import write_lr_dataset
from oracle import vqa_ref

pytestmark = pytest.mark.gpu

R, DV, LQ = 9, 32, 12
WIDTHS = dict(emb_size=12, question_out=24, visual_out=DV, attention_dim=16, fusion_in=16, fusion_hidden=8, dropout=0.0)


def _encoded_split(info, split):
    """The split as the reference's loader flattens it (VQALoader.py:128-139): encoded questions / answers, features, types."""
    from vqa_amd.loader import FeatureVQALoader
    from vqa_amd.metrics import LR_TYPES
    from vqa_amd.train import dataset_files
    from vqa_amd.vocab_encoder import VocabEncoder
    files = dataset_files('LR', info['data_path'])
    enc_q = VocabEncoder(files['all_questions'], questions=True, len_question=LQ)
    enc_a = VocabEncoder(files['all_answers'], questions=False, range_numbers=True)
    ds = FeatureVQALoader(info['features_dir'], files[f'{split}_images'], files[f'{split}_questions'],
                          files[f'{split}_answers'], enc_q, enc_a, train=False)
    items = [ds[i] for i in range(len(ds))]
    tokens = torch.from_numpy(np.stack([it[0] for it in items])).long()
    answers = torch.from_numpy(np.stack([it[1][:1] for it in items])).long().reshape(-1)
    feats = torch.stack([it[2] for it in items])
    types = np.array([LR_TYPES.index(it[3]) for it in items])
    return enc_q, enc_a, tokens, answers, feats, types


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_cli_end_to_end_on_json_files(tmp_path, capsys, dtype):
    from vqa_amd import test as ev, train as drv
    from vqa_amd.model import VQAAttnModel
    info = write_lr_dataset(tmp_path / 'data', n_images=30, R=R, Dv=DV, seed=2)
    out = str(tmp_path / 'ckpt')
    common = ['--model', 'VQAAttnModel', '--dataset', 'LR', '--data-path', info['data_path'], '--features-dir',
              info['features_dir'], '--bs', '8', '--gpu', '0', '--len-question', str(LQ)]
    hist = drv.main(common + ['--epochs', '2', '--lr', '0.003', '--out', out, '--workers', '2'], dtype=dtype, **WIDTHS)
    printed = capsys.readouterr().out
    assert 'epoch #1 val loss:' in printed and 'Best epoch #' in printed
    assert len(hist) == 2 and hist[0]['val']['samples'] == info['n_questions']['val']
    assert sum(hist[0]['val']['count'].values()) == info['n_questions']['val']
    assert hist[1]['train_loss'] < hist[0]['train_loss']
    assert np.isfinite(hist[1]['val']['loss'])

    res = ev.main(common + ['--models-dir', out, '--workers', '0'], dtype=dtype, **WIDTHS)
    printed = capsys.readouterr().out
    assert 'test loss:' in printed and 'OA:' in printed and 'AA:' in printed

    # the oracle on the same encoded test split, with the weights of the checkpoint the evaluation loaded
    enc_q, enc_a, tokens, answers, feats, types = _encoded_split(info, 'test')
    sd = torch.load(tmp_path / 'ckpt' / 'ModelBestRSVQA_VQAAttnModel.pth', weights_only=True)
    assert sd['seq2vec.embedding.weight'].shape[0] == len(enc_q.getVocab()) + 1          # seq2vec.py:35
    assert sd['linear_classif2.weight'].shape[0] == len(enc_a.getVocab())                # model.py:36
    p = {k: v.float().cpu() for k, v in sd.items()}
    if dtype == torch.bfloat16:
        p = {k: v.to(dtype).float() for k, v in p.items()}
        feats = feats.to(dtype).float()
    ref = vqa_ref.forward(p, feats, tokens)
    m = VQAAttnModel(enc_q.getVocab(), enc_a.getVocab(), None, 256, dtype=dtype, **WIDTHS)
    m.load_state_dict(sd)
    m = m.to('cuda').eval()
    with torch.no_grad():
        got = m(feats.to(dtype).cuda(), tokens.cuda()).float().cpu()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((got - ref).abs().max() / ref.abs().max()) <= tol
    assert abs(res['loss'] - float(vqa_ref.cross_entropy(ref, answers))) <= tol * max(1.0, res['loss'])
    assert sum(res['count'].values()) == info['n_questions']['test'] == len(tokens)
    from vqa_amd.metrics import LR_TYPES
    count, right = vqa_ref.accuracy_counts(vqa_ref.argmax_first(got.numpy()), answers.numpy(), types)
    assert [res['count'][t] for t in LR_TYPES] == count.tolist()
    assert [res['right'][t] for t in LR_TYPES] == right.tolist()
    oa, aa = vqa_ref.oa_aa(count, right)
    assert res['OA'] == oa and res['AA'] == aa
    if dtype == torch.float32:           # (bf16 logits can flip a near-tie against the fp32 oracle's argmax)
        c2, r2 = vqa_ref.accuracy_counts(vqa_ref.argmax_first(ref.numpy()), answers.numpy(), types)
        assert c2.tolist() == count.tolist() and r2.tolist() == right.tolist()


def test_cli_gap_model_on_stage_map_files(tmp_path):
    """--model VQAGAPModel (the class the reference's train.py trains) fed from .npz stage-map files."""
    from vqa_amd import train as drv
    maps = ((8, 8), (16, 4), (24, 4), (32, 2))
    info = write_lr_dataset(tmp_path / 'data', n_images=20, seed=4, stage_maps=maps)
    hist = drv.main(['--model', 'VQAGAPModel', '--dataset', 'LR', '--data-path', info['data_path'], '--features-dir',
                     info['features_dir'], '--bs', '8', '--gpu', '0', '--len-question', str(LQ), '--epochs', '3', '--lr', '0.003',
                     '--workers', '0'],
                    dtype=torch.float32, emb_size=12, question_out=24, fusion_in=16, fusion_hidden=8, dropout=0.0,
                    gap_channels=(8, 16, 24, 32))
    assert hist[-1]['train_loss'] < hist[0]['train_loss']
    assert sum(hist[0]['val']['count'].values()) == info['n_questions']['val']
