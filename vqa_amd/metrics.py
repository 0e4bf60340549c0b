"""Per-question-type accuracy bookkeeping (reference: vqa_model/train.py:105-110, 147-158;
vqa_model/test.py:85-108).

The argmax and the count / right counters are produced on the device by
vqa_argmax_accuracy (integer atomics, so the totals are exact and independent
of batch order or rank count); this class only turns them into OA / AA.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from .parallel import reduce_counters

HR_TYPES = ('presence', 'count', 'comp', 'area')            # train.py:54
LR_TYPES = ('presence', 'count', 'comp', 'rural_urban')     # train.py:57


class AccuracyMeter:
    def __init__(self, type_names: Sequence[str] = HR_TYPES, device='cuda'):
        self.type_names = tuple(type_names)
        self.n = len(self.type_names)
        self.counters = torch.zeros(2 * self.n, dtype=torch.int64, device=device)

    def type_ids(self, type_strs: Sequence[str]) -> torch.Tensor:
        """int32 device tensor of indices into type_names for a batch of type strings."""
        idx = {t: i for i, t in enumerate(self.type_names)}
        return torch.tensor([idx[t] for t in type_strs], dtype=torch.int32, device=self.counters.device)

    def update(self, model, answers: torch.Tensor, type_ids: torch.Tensor) -> torch.Tensor:
        """Argmax of the model's last forward + counter update; returns the predictions."""
        return model.predict(answers=answers.long().contiguous(), type_ids=type_ids.to(torch.int32).contiguous(),
                             counters=self.counters, n_types=self.n)

    def result(self, process_group=None) -> Dict[str, object]:
        """count / right per type, per-type accuracy for the non-empty types,
        OA = sum right / sum count, AA = (sum of per-type accuracies) / 4 -- the
        constant 4 of train.py:158, whatever the number of types."""
        c = reduce_counters(self.counters.clone(), process_group).cpu().tolist()
        count, right = c[:self.n], c[self.n:]
        acc = {t: right[i] * 1.0 / count[i] for i, t in enumerate(self.type_names) if count[i] > 0}
        total = sum(count)
        oa = sum(right) * 1.0 / total if total else float('nan')
        aa = sum(acc.values()) * 1.0 / 4
        return {'count': dict(zip(self.type_names, count)), 'right': dict(zip(self.type_names, right)),
                'acc': acc, 'OA': oa, 'AA': aa}

    def reset(self):
        self.counters.zero_()
