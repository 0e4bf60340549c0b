"""Host logic of the data-parallel train driver on CPU: 2 gloo ranks run vqa_amd.train.train() over the tiny
LR-shaped dataset (VocabEncoder -> FeatureVQALoader -> FeatureBatches) with a small CPU stand-in for the module
(the kernels need a GPU; the driver's rank handling does not).  Checked: validation loss and accuracy counters
are those of one process on the whole validation set, both ranks end on identical weights and take the same
best-epoch decisions, only rank 0 writes checkpoints, and the evaluation shards partition the set."""
import argparse
import json
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _tiny_dataset import write_lr_dataset

R, DV, BS, EPOCHS = 9, 32, 4, 3


class _StubModel(torch.nn.Module):
    """Softmax regression on [mean region feature, bag of question tokens] with the driver-facing interface of the
    real module: train_step / eval_step / predict / sync / check_status / (optimizer_)state_dict, a flat gradient
    buffer with two buckets for GradReducer."""

    def __init__(self, n_words, n_classes, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.n_words, self.num_classes = n_words, n_classes
        self.w = torch.nn.Parameter(0.1 * torch.randn(DV + n_words, n_classes, generator=g))
        self._flat_grad = torch.zeros(self.w.numel())
        half = (DV * n_classes)
        self.buckets = [(0, half), (half, self.w.numel())]
        self.dropout_rank = 0
        self._logits = None

    def _x(self, image, question):
        bag = torch.zeros(question.shape[0], self.n_words)
        bag.scatter_add_(1, question, torch.ones_like(question, dtype=torch.float32))
        return torch.cat([image.float().mean(1), bag], 1)

    def eval_step(self, image, question, answer):
        with torch.no_grad():
            self._logits = self._x(image, question) @ self.w
            return torch.nn.functional.cross_entropy(self._logits, answer).reshape(1)

    def train_step(self, image, question, answer, lr=1e-3, reducer=None):
        with torch.no_grad():
            x = self._x(image, question)
            z = x @ self.w
            loss = torch.nn.functional.cross_entropy(z, answer).reshape(1)
            d = torch.softmax(z, 1)
            d[torch.arange(len(answer)), answer] -= 1
            self._flat_grad.copy_((x.t() @ d / len(answer)).reshape(-1))
            reducer.begin(self)
            reducer.bucket_ready(0)
            reducer.bucket_ready(1)
            scale = reducer.finish()
            self.w -= lr * scale * self._flat_grad.view_as(self.w)
        return loss

    def predict(self, answers=None, type_ids=None, counters=None, n_types=4):
        pred = torch.from_numpy(np.argmax(self._logits.numpy(), axis=1))
        for p, a, t in zip(pred.tolist(), answers.tolist(), type_ids.tolist()):
            counters[t] += 1
            counters[n_types + t] += int(p == a)
        return pred

    def sync(self, check=True):
        pass

    def check_status(self):
        pass

    def optimizer_state_dict(self):
        return {'step_count': 0}


def _build(info, rank, world):
    from vqa_amd.loader import FeatureBatches, FeatureVQALoader
    from vqa_amd.metrics import LR_TYPES
    from vqa_amd.train import dataset_files
    from vqa_amd.vocab_encoder import VocabEncoder
    files = dataset_files('LR', info['data_path'])
    enc_q = VocabEncoder(files['all_questions'], questions=True, len_question=12)
    enc_a = VocabEncoder(files['all_answers'], questions=False, range_numbers=True)
    sets = {}
    for split in ('train', 'val'):
        ds = FeatureVQALoader(info['features_dir'], files[f'{split}_images'], files[f'{split}_questions'],
                              files[f'{split}_answers'], enc_q, enc_a, train=split == 'train')
        sets[split] = FeatureBatches(ds, BS, 'cpu', LR_TYPES, shuffle=split == 'train', feature_dtype=torch.float32,
                                     num_workers=0, rank=rank, world=world, drop_uneven=split == 'train')
    model = _StubModel(len(enc_q.getVocab()), len(enc_a.getVocab()), seed=3)
    return model, sets


def _worker(rank, world, port, info, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vqa_amd import train as drv
    saves = []
    real_save = torch.save
    torch.save = lambda obj, path, *a, **k: (saves.append(os.path.basename(str(path))), real_save(obj, path, *a, **k))[1]
    model, sets = _build(info, rank, world)
    assert len(sets['train']) == info['n_questions']['train'] // world
    lines = []
    hist = drv.train(model, sets['train'], sets['val'], EPOCHS, 0.5, argparse.Namespace(model='Stub'), 'LR',
                     out_dir=os.path.join(out_dir, 'ckpt'), out=lambda *a: lines.append(' '.join(str(x) for x in a)))
    torch.save = real_save
    torch.save(model.w.detach().clone(), os.path.join(out_dir, f'w{rank}.pt'))
    json.dump({'saves': saves, 'lines': lines,
               'hist': [{'loss': h['val']['loss'], 'count': h['val']['count'], 'right': h['val']['right'],
                         'samples': h['val']['samples'], 'train_loss': h['train_loss']} for h in hist]},
              open(os.path.join(out_dir, f'r{rank}.json'), 'w'))
    dist.destroy_process_group()


def test_two_rank_train_driver(tmp_path):
    info = write_lr_dataset(tmp_path / 'data', n_images=30, R=R, Dv=DV, seed=1)
    port = 23000 + (os.getpid() % 2000)
    out_dir = str(tmp_path / 'out')
    os.makedirs(out_dir)
    mp.spawn(_worker, args=(2, port, info, out_dir), nprocs=2, join=True)
    r0, r1 = (json.load(open(os.path.join(out_dir, f'r{r}.json'))) for r in (0, 1))
    # one writer: rank 0 saved best (at least once) + running + train state every epoch, rank 1 nothing
    assert r1['saves'] == [] and r1['lines'] == []
    assert r0['saves'].count('ModelRunningRSVQA_Stub.pth') == EPOCHS and r0['saves'].count('TrainState_Stub.pth') == EPOCHS
    assert 1 <= r0['saves'].count('ModelBestRSVQA_Stub.pth') <= EPOCHS
    assert any(l.startswith('epoch #0 val loss:') for l in r0['lines']) and any(l.startswith('Best epoch #') for l in r0['lines'])
    # reduced figures: identical on both ranks
    assert r0['hist'] == r1['hist']
    w0, w1 = torch.load(os.path.join(out_dir, 'w0.pt')), torch.load(os.path.join(out_dir, 'w1.pt'))
    assert torch.equal(w0, w1)
    sd = torch.load(os.path.join(out_dir, 'ckpt', 'ModelRunningRSVQA_Stub.pth'), weights_only=True)
    assert torch.equal(sd['w'], w0)
    # epoch 0's validation ran on the initial weights: one process over the whole validation set gives the same figures
    from vqa_amd import train as drv
    from vqa_amd.metrics import AccuracyMeter, LR_TYPES
    model, sets = _build(info, 0, 1)
    res = drv.evaluate(model, sets['val'], AccuracyMeter(LR_TYPES, device='cpu'))
    assert res['samples'] == r0['hist'][0]['samples'] == info['n_questions']['val']
    assert res['count'] == r0['hist'][0]['count'] and res['right'] == r0['hist'][0]['right']
    assert abs(res['loss'] - r0['hist'][0]['loss']) <= 1e-6 * abs(res['loss'])
    assert r0['hist'][-1]['train_loss'] < r0['hist'][0]['train_loss']


def test_feature_batches_shards(tmp_path):
    """Evaluation shards partition the set (nothing dropped or repeated); training shards are equally long and
    reshuffled every epoch, the same permutation on every rank."""
    info = write_lr_dataset(tmp_path, n_images=11, R=4, Dv=8, seed=5)
    _, s0 = _build(info, 0, 3)
    _, s1 = _build(info, 1, 3)
    _, s2 = _build(info, 2, 3)
    nv, nt = info['n_questions']['val'], info['n_questions']['train']
    idx = [s['val']._indices() for s in (s0, s1, s2)]
    assert sorted(sum(idx, [])) == list(range(nv)) and [len(s['val']) for s in (s0, s1, s2)] == [len(i) for i in idx]
    tr = [s['train']._indices() for s in (s0, s1, s2)]
    assert len({len(t) for t in tr}) == 1 and len(tr[0]) == nt // 3 == len(s0['train'])
    assert len(set(sum(tr, []))) == 3 * (nt // 3)
    first = list(s0['train']._indices())
    for _ in s0['train']:
        pass
    assert s0['train']._indices() != first            # next epoch, next permutation
    q, a, f, t = next(iter(s0['val']))
    assert q.dtype == torch.int64 and q.shape[1] == 12 and a.dtype == torch.int64 and f.shape[1:] == (4, 8) and t.dtype == torch.int32
