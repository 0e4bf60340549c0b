"""Caption strings -> token ids for the caption stream.

The reference hands its caption model the default-collated caption item: a list whose element 0 is the tuple of B
caption strings (VQALoader.py:163 returns ``self.all_captions[str(img_idx)]``, a one-element list per sample, as
written by trial.py:43-51; model.py:1673 undoes the collate with ``captions = list(captions[0])``) and tokenises
them with the MiniLM word-piece tokenizer fetched by name (model.py:1583, 1674).  That tokenizer and its weights are
not part of this package: the caption stream here embeds whole words from a vocabulary built from the caption file
itself (``from_captions_json``), ids 1..V in order of first appearance, 0 = padding; a word outside the vocabulary maps
to 0 and therefore drops out of the masked mean (model.py:1571-1574), as a padded position does.
"""
from __future__ import annotations

import json
import re
from typing import Dict, Iterable, List, Sequence

import torch

MAX_CAPTION_TOKENS = 16         # trial.py:16 generates captions with max_length=16

_WORD = re.compile(r"[a-z0-9]+(?:'[a-z]+)?")


def caption_words(sentence: str) -> List[str]:
    """Lower-cased words of a caption; punctuation separates words and is dropped."""
    return _WORD.findall(sentence.lower())


def uncollate(captions) -> List[str]:
    """The B caption strings of a batch in any of the forms the reference's loop produces: the collated item (a list
    whose element 0 is the tuple of B strings, model.py:1673), a plain sequence of B strings, or a sequence of
    one-element lists (un-collated dataset items, VQALoader.py:163)."""
    if isinstance(captions, str):
        return [captions]
    captions = list(captions)
    if len(captions) == 1 and isinstance(captions[0], (tuple, list)):
        return [str(c) for c in captions[0]]
    out = []
    for c in captions:
        if isinstance(c, (tuple, list)):
            if len(c) != 1:
                raise ValueError('a caption item must be one string or a one-element list (captions.json holds one caption per image)')
            c = c[0]
        out.append(str(c))
    return out


class CaptionTokenizer:
    def __init__(self, vocab: Sequence[str], max_tokens: int = MAX_CAPTION_TOKENS):
        self.vocab = list(vocab)
        self.max_tokens = int(max_tokens)
        self.index: Dict[str, int] = {w: i + 1 for i, w in enumerate(self.vocab)}
        if len(self.index) != len(self.vocab):
            raise ValueError('caption vocabulary holds a word twice')

    @classmethod
    def from_sentences(cls, sentences: Iterable[str], max_tokens: int = MAX_CAPTION_TOKENS) -> 'CaptionTokenizer':
        seen: Dict[str, None] = {}
        for s in sentences:
            for w in caption_words(s):
                seen.setdefault(w)
        return cls(list(seen), max_tokens)

    @classmethod
    def from_captions_json(cls, path: str, max_tokens: int = MAX_CAPTION_TOKENS) -> 'CaptionTokenizer':
        """captions.json as trial.py:43-51 writes it: {str(image id): [caption]}; image ids in numeric order."""
        with open(path) as fh:
            data = json.load(fh)
        keys = sorted(data, key=lambda k: (len(k), k))
        return cls.from_sentences((c for k in keys for c in data[k]), max_tokens)

    def __len__(self) -> int:
        return len(self.vocab)

    def encode(self, captions, device=None) -> torch.Tensor:
        """(B, max_tokens) int64: word ids, truncated to max_tokens and right-padded with 0."""
        rows = []
        for s in uncollate(captions):
            ids = [self.index.get(w, 0) for w in caption_words(s)][:self.max_tokens]
            rows.append(ids + [0] * (self.max_tokens - len(ids)))
        t = torch.tensor(rows, dtype=torch.int64).reshape(len(rows), self.max_tokens)
        return t.to(device) if device is not None else t
