"""The C-ABI library loads on a machine without a GPU and exports every symbol the public
header declares (no compute calls here)."""
import ctypes
import os
import re

from vqa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'vqa_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vqa_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/vqa_hip.h but not exported'
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, 'ctypes signatures and header disagree'


def test_version_and_argument_errors_without_gpu():
    lib = _lib.load()
    assert lib.vqa_version() == 220
    # argument validation happens before any HIP call
    rc = lib.vqa_inv_norm(None, None, 4, 12, 12, 0, None)
    assert rc == -22 and b'vqa_inv_norm' in lib.vqa_last_error_string()
    assert lib.vqa_gemm_splitk_hint(256, 1200, 2048, 1, 1) >= 1
    assert lib.vqa_attn_lds_bytes(196, 1200, 2048) <= 160 * 1024


def test_missing_library_is_loud(monkeypatch, tmp_path):
    import pytest
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.VqaHipError):
        _lib.load()


def test_model_refuses_cpu_tensors():
    import pytest
    import torch
    from vqa_amd.model import VQAAttnModel
    m = VQAAttnModel(['w'] * 10, ['a'] * 5, None, emb_size=8, question_out=8, visual_out=8, attention_dim=8,
                     fusion_in=8, fusion_hidden=8)
    with pytest.raises(_lib.VqaHipError):
        m(torch.zeros(2, 4, 8), torch.zeros(2, 5, dtype=torch.long))
    with pytest.raises(ValueError):
        VQAAttnModel(['w'] * 10, ['a'] * 5, None, attention_dim=12)


def test_state_dict_keys_follow_the_reference_names():
    from vqa_amd.model import VQAAttnModel
    m = VQAAttnModel(['w'] * 10, ['a'] * 5, None, emb_size=8, question_out=8, visual_out=8, attention_dim=8,
                     fusion_in=8, fusion_hidden=8)
    keys = set(m.state_dict())
    for k in ('linear_q.weight', 'linear_q.bias', 'linear_v.weight', 'linear_classif1.weight', 'linear_classif2.bias',
              'seq2vec.embedding.weight', 'seq2vec.rnn.weight_ih_l0', 'seq2vec.rnn.weight_hh_l0',
              'seq2vec.rnn.bias_ih_l0', 'seq2vec.rnn.bias_hh_l0'):
        assert k in keys
    assert m.seq2vec.embedding.weight.shape[0] == 10 + 1            # len(vocab)+1 rows (seq2vec.py:35)
    assert m.num_classes == 5 and m.linear_classif2.weight.shape == (5, 8)
    assert not any(p.requires_grad for p in m.seq2vec.parameters())  # frozen (model.py:42-43)
