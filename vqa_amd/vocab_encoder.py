"""Word <-> id dictionaries for questions and answers.

Host-side mirror of the reference's vocabulary encoder
(vqa_model/VocabEncoder.py:15-137): same class name, constructor arguments,
attributes (``words``, ``list_words``, ``questions``, ``range_numbers``,
``encoder_type``) and ``encode`` / ``decode`` / ``getVocab`` behaviour, with the
two module constants of the reference (MAX_ANSWERS = 100, LEN_QUESTION = 20,
VocabEncoder.py:11-12) turned into constructor parameters so that the
3000-answer / 14-token configuration can be expressed.
"""
from __future__ import annotations

import json
from collections import OrderedDict
from typing import Dict, List, Optional

MAX_ANSWERS = 100
LEN_QUESTION = 20

_EOS = '<EOS>'

# (upper bound inclusive, label) pairs; anything above the last bound falls through
_COUNT_BUCKETS = ((10, 'between 0 and 10'), (100, 'between 10 and 100'), (1000, 'between 100 and 1000'))
_COUNT_OVER = 'more than 1000'
_AREA_BUCKETS = ((10, 'between 0m2 and 10m2'), (100, 'between 10m2 and 100m2'), (1000, 'between 100m2 and 1000m2'))
_AREA_OVER = 'more than 1000m2'


def _bucket(num: int, buckets, over: str, keep: str) -> str:
    """Bucket label of a positive number; zero and negatives keep their token
    (the reference only rewrites ``num > 0``, VocabEncoder.py:47-54)."""
    if num <= 0:
        return keep
    for bound, label in buckets:
        if num <= bound:
            return label
    return over


def _strip_final_punct(sentence: str) -> str:
    # exactly one trailing '?' or '.' is removed (VocabEncoder.py:39-40, 84-85);
    # an empty sentence raises IndexError as upstream does
    if sentence[-1] in '?.':
        return sentence[:-1]
    return sentence


class VocabEncoder:
    def __init__(self, JSONFile: Optional[str], string: Optional[str] = None, questions: bool = True,
                 range_numbers: bool = False, max_answers: int = MAX_ANSWERS,
                 len_question: int = LEN_QUESTION):
        self.encoder_type = 'question' if questions else 'answer'
        self.questions = questions
        self.range_numbers = range_numbers
        self.max_answers = max_answers
        self.len_question = len_question

        if JSONFile is not None:
            with open(JSONFile) as fh:
                self.data = json.load(fh)[self.encoder_type + 's']
        else:
            # upstream builds an entry without an "active" key here and then
            # fails on it (VocabEncoder.py:30-37); the single sentence is
            # treated as active instead
            self.data = [{self.encoder_type: string, 'active': True}]

        freq: "OrderedDict[str, int]" = OrderedDict()
        for entry in self.data:
            if not entry['active']:
                continue
            for token in self._tokens(entry[self.encoder_type]):
                freq[token] = freq.get(token, 0) + 1

        # descending frequency; equal counts keep first-seen order (stable sort,
        # VocabEncoder.py:71)
        ranked = sorted(freq.items(), key=lambda kv: -kv[1])
        if not questions:
            ranked = ranked[:max_answers]
        self.list_words: List[str] = [_EOS] + [w for w, _ in ranked]
        self.words: Dict[str, int] = {w: i for i, w in enumerate(self.list_words)}

    def _tokens(self, sentence: str) -> List[str]:
        out = []
        for token in _strip_final_punct(sentence).split():
            token = token.lower()
            if not self.questions:
                if self.range_numbers and token.isdigit():
                    token = _bucket(int(token), _COUNT_BUCKETS, _COUNT_OVER, token)
                # the area rule is independent of range_numbers (VocabEncoder.py:56, 101)
                if token[-2:] == 'm2':
                    token = _bucket(int(token[:-2]), _AREA_BUCKETS, _AREA_OVER, token)
            out.append(token)
        return out

    def encode(self, sentence: str) -> List[int]:
        """ids of the sentence; an unknown word raises KeyError.  Questions get
        one <EOS> appended and are padded / truncated to ``len_question``."""
        ids = [self.words[t] for t in self._tokens(sentence)]
        if self.questions:
            ids.append(self.words[_EOS])
            ids += [self.words[_EOS]] * (self.len_question - len(ids))
            ids = ids[:self.len_question]
        return ids

    def getVocab(self) -> List[str]:
        return self.list_words

    def decode(self, sentence) -> str:
        words = []
        for i in sentence:
            if i == 0:
                break
            words.append(self.list_words[i])
        text = ' '.join(words)
        return text + '?' if self.questions else text
