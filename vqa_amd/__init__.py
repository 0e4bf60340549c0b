"""MI355X-native VQA train step: question encoder, visual front-ends, fusion head, loss, Adam.

HIP kernels (csrc/ -> libvqa_hip.so, C ABI in include/vqa_hip.h) driven from
PyTorch-ROCm through ctypes.  There is no CPU fallback.
"""
from .vocab_encoder import VocabEncoder  # noqa: F401

_MODELS = ('VQAAttnModel', 'VQAGAPModel', 'VQAGAPStackModel', 'VQAModel')
__all__ = ['VocabEncoder', *_MODELS]


def __getattr__(name):
    if name in _MODELS:
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
