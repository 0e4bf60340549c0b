"""Host-side vocabulary encoder against fixtures produced by the reference's own
VocabEncoder (oracle/make_golden.py, G1): vocab order incl. frequency ties and inactive
entries, answer cap, '?' / '.' stripping, number and m2 bucketing, padding / truncation,
OOV error, decode."""
import json
import os

import pytest

from vqa_amd.vocab_encoder import VocabEncoder, MAX_ANSWERS, LEN_QUESTION


@pytest.fixture(scope='module')
def g1(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'g1_vocab_expected.json')))


def test_constants(g1):
    assert (MAX_ANSWERS, LEN_QUESTION) == (g1['max_answers'], g1['len_question'])


def test_question_vocab_and_encodings(golden_dir, g1):
    enc = VocabEncoder(os.path.join(golden_dir, 'g1_all_questions.json'), questions=True)
    assert enc.getVocab() == g1['question_vocab']
    assert enc.words == g1['question_words']
    assert enc.getVocab()[0] == '<EOS>'
    for sentence, ids in g1['question_encodings']:
        got = enc.encode(sentence)
        assert got == ids, sentence
        assert len(got) == LEN_QUESTION
    for ids, text in g1['question_decodings']:
        assert enc.decode(ids) == text
    with pytest.raises(KeyError) as ei:
        enc.encode('is there a unicorn?')
    assert ['KeyError', ei.value.args[0]] == g1['question_oov']


@pytest.mark.parametrize('rn,key', [(False, 'answers_plain'), (True, 'answers_range')])
def test_answer_vocab_and_encodings(golden_dir, g1, rn, key):
    enc = VocabEncoder(os.path.join(golden_dir, 'g1_all_answers.json'), questions=False, range_numbers=rn)
    exp = g1[key]
    assert enc.getVocab() == exp['vocab']
    assert len(enc.getVocab()) == MAX_ANSWERS + 1          # capped, plus <EOS>
    assert enc.words == exp['words']
    for sentence, ids in exp['encodings']:
        if ids and ids[0] == 'KeyError':
            with pytest.raises(KeyError) as ei:
                enc.encode(sentence)
            assert ei.value.args[0] == ids[1]
        else:
            assert enc.encode(sentence) == ids, sentence
    for ids, text in exp['decodings']:
        assert enc.decode(ids) == text


def test_configurable_lengths(golden_dir):
    """The 3000-answer / 14-token configuration: the reference's two constants as parameters."""
    enc = VocabEncoder(os.path.join(golden_dir, 'g1_all_questions.json'), questions=True, len_question=14)
    ids = enc.encode('is there a road?')
    assert len(ids) == 14 and ids[4:] == [0] * 10 and 0 not in ids[:4]
    assert len(enc.encode(' '.join(['road'] * 30))) == 14
    enc_a = VocabEncoder(os.path.join(golden_dir, 'g1_all_answers.json'), questions=False, max_answers=3000)
    assert len(enc_a.getVocab()) > MAX_ANSWERS + 1


def test_single_sentence_constructor():
    enc = VocabEncoder(None, string='What is the area?', questions=True)
    assert enc.getVocab() == ['<EOS>', 'what', 'is', 'the', 'area']
    with pytest.raises(IndexError):
        enc.encode('')
