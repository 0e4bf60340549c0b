"""FeatureVQALoader keeps the reference loader's item contract (VQALoader.py:62-63, 128-138, 176) and
collate() the batch preparation of train.py:187-189; checked on small JSON files written here."""
import json

import numpy as np
import torch

from vqa_amd.loader import FeatureVQALoader, collate
from vqa_amd.vocab_encoder import VocabEncoder


def _write(tmp_path):
    questions = {'questions': [
        {'id': 0, 'question': 'Is there a road at the top?', 'type': 'presence', 'answers_ids': [0], 'active': True},
        {'id': 1, 'question': 'How many buildings are there?', 'type': 'count', 'answers_ids': [1], 'active': True},
        {'id': 2, 'question': 'Is it rural or urban?', 'type': 'rural_urban', 'answers_ids': [2], 'active': True},
        {'id': 3, 'question': 'Is there a road?', 'type': 'presence', 'answers_ids': [3], 'active': True},
        {'id': 4, 'question': 'Unused question here?', 'type': 'comp', 'answers_ids': [4], 'active': False},
    ]}
    answers = {'answers': [
        {'id': 0, 'answer': 'yes', 'active': True}, {'id': 1, 'answer': '12', 'active': True},
        {'id': 2, 'answer': 'urban', 'active': True}, {'id': 3, 'answer': 'no', 'active': True},
        {'id': 4, 'answer': 'yes', 'active': False},
    ]}
    images = {'images': [
        {'id': 0, 'active': True, 'questions_ids': [0, 1]},
        {'id': 1, 'active': False, 'questions_ids': [4]},
        {'id': 2, 'active': True, 'questions_ids': [2]},
        {'id': 3, 'active': True, 'questions_ids': [3]},
    ]}
    for name, obj in (('q.json', questions), ('a.json', answers), ('i.json', images)):
        (tmp_path / name).write_text(json.dumps(obj))
    rng = np.random.default_rng(0)
    for i in (0, 2):
        np.save(tmp_path / f'{i}.npy', rng.standard_normal((9, 16)).astype(np.float32))
    np.savez(tmp_path / '3.npz', s1=rng.standard_normal((4, 4, 4)).astype(np.float32),
             s2=rng.standard_normal((8, 2, 2)).astype(np.float32))
    return str(tmp_path / 'i.json'), str(tmp_path / 'q.json'), str(tmp_path / 'a.json')


def test_item_contract_and_collate(tmp_path):
    ifile, qfile, afile = _write(tmp_path)
    eq = VocabEncoder(qfile, questions=True, len_question=8)
    ea = VocabEncoder(afile, questions=False, range_numbers=True, max_answers=3000)
    ds = FeatureVQALoader(str(tmp_path), ifile, qfile, afile, eq, ea, train=True, ratio_images_to_use=0.7)
    # 3 active images, int(3 * 0.7) = 2 kept: ids 0 and 2 -> questions 0, 1, 2
    assert ds.image_ids == [0, 2] and len(ds) == 3
    q, a, f, t = ds[1]
    assert q.dtype == np.int16 and q.shape == (8,) and a.dtype == np.int16
    assert list(q) == eq.encode('How many buildings are there?')
    assert list(a) == ea.encode('12') == [ea.words['between 10 and 100']]
    assert t == 'count' and tuple(f.shape) == (9, 16)
    assert torch.equal(f, torch.from_numpy(np.load(tmp_path / '0.npy')))
    assert ds[2][3] == 'rural_urban' and torch.equal(ds[2][2], torch.from_numpy(np.load(tmp_path / '2.npy')))
    qb, ab, fb, tb = collate([ds[i] for i in range(3)], 'cpu', ['count', 'presence', 'comp', 'rural_urban'], torch.float32)
    assert qb.dtype == torch.int64 and qb.shape == (3, 8) and ab.dtype == torch.int64 and ab.shape == (3,)
    assert fb.shape == (3, 9, 16) and tb.tolist() == [1, 0, 3]
    assert eq.decode(qb[0].tolist()) == 'is there a road at the top?'


def test_stage_map_items(tmp_path):
    ifile, qfile, afile = _write(tmp_path)
    eq = VocabEncoder(qfile, questions=True, len_question=8)
    ea = VocabEncoder(afile, questions=False)
    ds = FeatureVQALoader(str(tmp_path), ifile, qfile, afile, eq, ea, train=False)
    assert ds.image_ids == [0, 2, 3] and len(ds) == 4
    q, a, f, t = ds[3]
    assert isinstance(f, list) and [tuple(x.shape) for x in f] == [(4, 4, 4), (8, 2, 2)]
    qb, ab, fb, tb = collate([ds[3], ds[3]], 'cpu', ['presence'], torch.float32)
    assert [tuple(x.shape) for x in fb] == [(2, 4, 4, 4), (2, 8, 2, 2)] and tb.tolist() == [0, 0]
