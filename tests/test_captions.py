"""Caption strings -> ids (host logic): the collated form the reference's model receives (model.py:1673), dataset-item
form (VQALoader.py:163), vocabulary from a captions.json as trial.py:43-51 writes it."""
import json

import pytest
import torch

from vqa_amd.captions import CaptionTokenizer, MAX_CAPTION_TOKENS, caption_words, uncollate


def test_words_and_uncollate_forms():
    assert caption_words("A large Building, next to 2 roads.") == ['a', 'large', 'building', 'next', 'to', '2', 'roads']
    assert caption_words("it's a river") == ["it's", 'a', 'river']
    strings = ['a road', 'two houses']
    assert uncollate([tuple(strings)]) == strings            # default_collate of one-element lists
    assert uncollate(strings) == strings
    assert uncollate([[s] for s in strings]) == strings       # un-collated dataset items
    assert uncollate('a road') == ['a road']
    with pytest.raises(ValueError):
        uncollate([['a', 'b'], ['c', 'd']])


def test_encode_pads_truncates_and_drops_unknown_words(tmp_path):
    data = {'10': ['a road next to a house'], '2': ['a river'], '1': ['green field with trees']}
    p = tmp_path / 'captions.json'
    p.write_text(json.dumps(data))
    tok = CaptionTokenizer.from_captions_json(str(p))
    # ids follow first appearance with the image ids in numeric order: 1, 2, 10
    assert tok.vocab == ['green', 'field', 'with', 'trees', 'a', 'river', 'road', 'next', 'to', 'house']
    ids = tok.encode([('a river', 'a purple road!', ' '.join(['trees'] * 40))])
    assert ids.shape == (3, MAX_CAPTION_TOKENS) and ids.dtype == torch.int64
    assert ids[0].tolist() == [5, 6] + [0] * 14
    assert ids[1].tolist() == [5, 0, 7] + [0] * 13           # a word outside the vocabulary is a padded position
    assert ids[2].tolist() == [4] * MAX_CAPTION_TOKENS       # truncated to the captioner's max_length
    with pytest.raises(ValueError):
        CaptionTokenizer(['a', 'a'])
