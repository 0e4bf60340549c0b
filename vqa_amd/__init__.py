"""MI355X-native VQA attention / fusion / classifier train step.

HIP kernels (csrc/ -> libvqa_hip.so, C ABI in include/vqa_hip.h) driven from
PyTorch-ROCm through ctypes.  There is no CPU fallback.
"""
from .vocab_encoder import VocabEncoder  # noqa: F401

__all__ = ['VocabEncoder', 'VQAAttnModel']


def __getattr__(name):
    if name == 'VQAAttnModel':
        from .model import VQAAttnModel
        return VQAAttnModel
    raise AttributeError(name)
