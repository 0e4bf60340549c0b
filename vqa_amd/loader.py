"""Dataset over precomputed image features with the reference loader's item contract.

Mirrors vqa_model/VQALoader.py:33-178 for the non-BERT models: the same three JSON files
(``images`` with ``id / active / questions_ids``, ``questions`` with ``question / type /
answers_ids``, ``answers`` with ``answer``), the same ``ratio_images_to_use`` cut (VQALoader.py:62-63),
the same per-question item ``(question int16[L], answer int16[n], image, type_str)``
(VQALoader.py:138, 176).  The image element is the precomputed feature array of the image
(``<feature_dir>/<id>.npy``: (R, Dv) region features, or a ``.npz`` of stage maps for the gated
pooling model) instead of the transformed .tif, because the ResNet lives outside this package.

The reference's random flips / rotations of the image (VQALoader.py:147-155) have no counterpart:
on precomputed features they would only permute the regions, which none of the front-ends here is
sensitive to (mean and attention pooling are permutation invariant).
"""
from __future__ import annotations

import json
import os
from typing import List, Sequence

import numpy as np
import torch


class FeatureVQALoader(torch.utils.data.Dataset):
    def __init__(self, feature_dir: str, images_file: str, questions_file: str, answers_file: str,
                 encoder_questions, encoder_answers, train: bool = True, ratio_images_to_use: float = 1,
                 cache_features: bool = True):
        self.encoder_questions = encoder_questions
        self.encoder_answers = encoder_answers
        self.train = train
        self.feature_dir = feature_dir
        with open(questions_file) as fh:
            self.questionsJSON = json.load(fh)
        with open(answers_file) as fh:
            self.answersJSON = json.load(fh)
        with open(images_file) as fh:
            self.imagesJSON = json.load(fh)

        images = [img['id'] for img in self.imagesJSON['images'] if img['active']]
        images = images[:int(len(images) * ratio_images_to_use)]
        self.image_ids = images
        self._cache = {} if cache_features else None

        self.images_questions_answers: List[list] = []
        for i, image in enumerate(images):
            # the reference indexes the image list by id (VQALoader.py:70, 128): ids are positions
            for qid in self.imagesJSON['images'][image]['questions_ids']:
                question = self.questionsJSON['questions'][qid]
                answer_str = self.answersJSON['answers'][question['answers_ids'][0]]['answer']
                self.images_questions_answers.append([self.encoder_questions.encode(question['question']),
                                                      self.encoder_answers.encode(answer_str), i, question['type']])
        self.len = len(self.images_questions_answers)

    def __len__(self):
        return self.len

    def _features(self, i: int):
        if self._cache is not None and i in self._cache:
            return self._cache[i]
        base = os.path.join(self.feature_dir, str(self.image_ids[i]))
        if os.path.exists(base + '.npy'):
            f = torch.from_numpy(np.load(base + '.npy'))
        else:
            with np.load(base + '.npz') as z:
                f = [torch.from_numpy(z[k]) for k in sorted(z.files)]
        if self._cache is not None:
            self._cache[i] = f
        return f

    def __getitem__(self, idx):
        q = self.images_questions_answers[idx]
        return np.array(q[0], dtype='int16'), np.array(q[1], dtype='int16'), self._features(q[2]), q[3]


def collate_host(items: Sequence, type_names: Sequence[str]):
    """The host half of the batch preparation of train.py:176, 187-189 (safe inside DataLoader worker
    processes: nothing here touches the GPU): question.long() (B, L), answer.long().resize_(B), the image
    element stacked, plus int32 question-type ids for the on-device accuracy counters (the reference keeps
    the strings and counts in Python)."""
    question = torch.from_numpy(np.stack([it[0] for it in items])).long()
    answer = torch.from_numpy(np.stack([it[1][:1] for it in items])).long().reshape(len(items))
    first = items[0][2]
    if isinstance(first, (list, tuple)):
        image = [torch.stack([it[2][k] for it in items]) for k in range(len(first))]
    else:
        image = torch.stack([it[2] for it in items])
    types = torch.tensor([type_names.index(it[3]) for it in items], dtype=torch.int32)
    return question, answer, image, types


def to_device(batch, device, feature_dtype=torch.bfloat16):
    """The device half (the .to(cuda) copies of train.py:187-189): features are cast to the compute dtype."""
    question, answer, image, types = batch
    if isinstance(image, (list, tuple)):
        image = [m.to(device=device, dtype=feature_dtype, non_blocking=True) for m in image]
    else:
        image = image.to(device=device, dtype=feature_dtype, non_blocking=True)
    return (question.to(device, non_blocking=True), answer.to(device, non_blocking=True), image,
            types.to(device, non_blocking=True))


def collate(items: Sequence, device, type_names: Sequence[str], feature_dtype=torch.bfloat16):
    """Both halves in one call: (question (B, L) int64, answer (B,) int64, image, type ids int32) on `device`."""
    return to_device(collate_host(items, type_names), device, feature_dtype)


class _HostCollate:
    """Picklable collate_fn (DataLoader workers may be started with spawn)."""

    def __init__(self, type_names):
        self.type_names = tuple(type_names)

    def __call__(self, items):
        return collate_host(items, self.type_names)


class FeatureBatches:
    """The DataLoader of train.py:30-31 around a FeatureVQALoader, yielding device batches
    (question, answer, image, type ids): ``shuffle`` for the training set (reshuffled every epoch from
    ``seed`` + epoch), 2 worker processes as upstream by default.  With ``world`` > 1 the samples are dealt
    to the ranks: training drops the remainder so that every rank runs the same number of equally sized
    steps (each step ends in a gradient all-reduce); evaluation takes samples rank, rank + world, ... with
    nothing dropped or repeated, so counters summed over ranks are those of one rank on the whole set.
    ``len()`` is the number of samples this rank iterates over."""

    def __init__(self, dataset, batch_size: int, device, type_names: Sequence[str], shuffle: bool = False,
                 feature_dtype=torch.bfloat16, num_workers: int = 2, rank: int = 0, world: int = 1, seed: int = 42,
                 drop_uneven: bool = False):
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), device
        self.shuffle, self.feature_dtype, self.num_workers = shuffle, feature_dtype, num_workers
        self.rank, self.world, self.seed, self.drop_uneven = rank, world, seed, drop_uneven
        self.collate_fn = _HostCollate(type_names)
        self.epoch = 0

    def _indices(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        if self.world > 1:
            if self.drop_uneven:
                order = order[:n - n % self.world]
            order = order[self.rank::self.world]
        return order

    def __len__(self):
        n = len(self.dataset)
        if self.world == 1:
            return n
        if self.drop_uneven:
            return n // self.world
        return len(range(self.rank, n, self.world))

    def __iter__(self):
        order = self._indices()
        self.epoch += 1
        loader = torch.utils.data.DataLoader(self.dataset, batch_size=self.batch_size, sampler=order,
                                             num_workers=self.num_workers, collate_fn=self.collate_fn,
                                             pin_memory=self.num_workers > 0)
        for batch in loader:
            yield to_device(batch, self.device, self.feature_dtype)
