"""ctypes binding of libvqa_hip.so (the C ABI declared in include/vqa_hip.h).

There is no fallback: if the library is missing or a call fails, an exception
is raised.  Pointers are raw device addresses taken from torch tensors; the
stream is torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16 = 0, 1
ABI_VERSION = 220          # VQA_HIP_VERSION of include/vqa_hip.h these signatures were written against
STATUS_WORDS = 4           # VQA_STATUS_BAD_TOKEN, VQA_STATUS_BAD_TARGET, VQA_STATUS_SKIPPED_STEP, VQA_STATUS_RESIDENT_TIMEOUT
RESIDENT_PITCH, RESIDENT_MAX_ROWS, RESIDENT_SYNC_WORDS, RESIDENT_MAX_STEPS = 2432, 512, 176, 64     # VQA_LSTM_RESIDENT_* of the header


def resident_ring_rows(B: int) -> int:
    """VQA_LSTM_RESIDENT_RING_ROWS(B)."""
    return ((B + 15) & ~15) + 16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VQA_HIP_LIB', os.path.join(_HERE, 'libvqa_hip.so'))   # (override: A/B of kernel builds)

_p, _i, _l, _f, _u, _z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint, C.c_size_t

# name -> argtypes (stream last); every function returns int unless listed in _RESTYPE
_SIGNATURES = {
    'vqa_version': [],
    'vqa_last_error_string': [],
    'vqa_process_lengths': [_p, _p, _i, _i, _p],
    'vqa_embedding_gather': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'vqa_lstm_cell': [_p, _i, _z, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'vqa_sequence_pack': [_p, _i, _i, _p, _p, _p, _p, _l, _p, _p],
    'vqa_embedding_gather_packed': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p],
    'vqa_lstm_cell_packed': [_p, _i, _z, _p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _u, _f, _i, _p],
    'vqa_lstm_cell_packed_table': [_p, _i, _z, _p, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _u, _f, _p, _i, _p],
    'vqa_lstm_cell01_table': [_p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _u, _f, _i, _i, _p],
    'vqa_lstm_resident_workgroups': [_i],
    'vqa_lstm_resident': [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _u, _f, _p, _p, _p],
    'vqa_lstm_cell_bwd_packed': [_p, _i, _z, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'vqa_embedding_grad': [_p, _i, _z, _i, _p, _p, _p, _i, _p, _i, _i, _i, _p],
    'vqa_colsum_multi': [_p, _i, _p],
    'vqa_slab_sum_multi': [_p, _i, _p],
    'vqa_colsum_rows': [_p, _i, _p, _p, _i, _i, _p],
    'vqa_caption_pool': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'vqa_inv_norm': [_p, _p, _l, _i, _i, _i, _p],
    'vqa_grid_mean_splits': [_i, _i, _i, _i],
    'vqa_grid_mean': [_p, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _p],
    'vqa_grid_mean_static': [_p, _i, _i, _i, _i, _p, _i, _i, _i, _p],
    'vqa_gate_fwd': [_p, _i, _z, _p, _p, _i, _p, _i, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_gate_bwd': [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _p],
    'vqa_stack_maxavg_fwd': [_p, _p, _i, _p, _i, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_stack_maxavg_bwd': [_p, _i, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    'vqa_flatten_cm': [_p, _i, _p, _i, _i, _i, _i, _u, _f, _i, _p],
    'vqa_unflatten_cm': [_p, _i, _p, _i, _i, _i, _i, _i, _p],
    'vqa_attn_lds_bytes': [_i, _i, _i],
    'vqa_attn_fwd': [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _u, _f, _p],
    'vqa_attn_bwd': [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    'vqa_attn_fwd_qslabs': [_p, _i, _p, _i, _p, _i, _z, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _u, _f, _p],
    'vqa_attn_bwd_dslabs': [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _z, _u, _f, _p, _i, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    'vqa_attn_splits': [_i, _i, _i, _i, _i],
    'vqa_attn_split_workspace_floats': [_i, _i, _i, _i],
    'vqa_attn_fwd_split': [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _u, _f, _i, _p, _p],
    'vqa_attn_bwd_split': [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    'vqa_gemm_splitk_hint': [_i, _i, _i, _i, _i],
    'vqa_gemm_slabs': [_p, _i, _i, _p, _i, _i, _p, _z, _i, _i, _i, _i, _i, _i, _i, _p],
    'vqa_gemm_slabs_rows': [_p, _i, _i, _p, _i, _i, _p, _z, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    'vqa_gemm_slabs_group': [_p, _i, _i, _i, _i, _p],
    'vqa_gemm_slabs_group_colsum': [_p, _i, _i, _i, _i, _p, _i, _p],
    'vqa_slab_sum_transpose': [_p, _i, _z, _i, _i, _p, _i, _p],
    'vqa_fusion_fwd_slabs': [_p, _i, _z, _p, _p, _i, _z, _p, _p, _p, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_epilogue_fusion': [_p, _i, _z, _i, _i, _p, _p, _p, _p, _i, _u, _f, _i, _p],
    'vqa_gemm_fused_rownorm': [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    'vqa_gemm_fused_rows': [_p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p],
    'vqa_gemm_fused': [_p, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    'vqa_epilogue': [_p, _i, _z, _i, _i, _p, _p, _p, _i, _i, _p, _i, _p, _i, _i, _u, _f, _p],
    'vqa_dropout': [_p, _i, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_fusion_fwd': [_p, _p, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_fusion_bwd': [_p, _i, _z, _i, _p, _p, _p, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_tanh_dropout_bwd': [_p, _i, _z, _p, _i, _p, _i, _i, _i, _u, _f, _i, _p],
    'vqa_colsum': [_p, _i, _p, _i, _i, _i, _p],
    'vqa_cast_pad': [_p, _i, _p, _i, _i, _i, _i, _p],
    'vqa_softmax_ce': [_p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _p, _p],
    'vqa_sum_scale': [_p, _p, _i, _f, _p],
    'vqa_argmax_accuracy': [_p, _i, _i, _i, _p, _p, _i, _p, _p, _p],
    'vqa_adam_step': [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _i, _f, _p, _p],
}
_RESTYPE = {'vqa_last_error_string': C.c_char_p, 'vqa_attn_lds_bytes': C.c_size_t,
            'vqa_attn_split_workspace_floats': C.c_size_t}
_NO_STATUS = {'vqa_version', 'vqa_lstm_resident_workgroups', 'vqa_last_error_string', 'vqa_attn_lds_bytes', 'vqa_gemm_splitk_hint',
              'vqa_grid_mean_splits', 'vqa_attn_splits', 'vqa_attn_split_workspace_floats'}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class VqaHipError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the kernel library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VqaHipError(f'{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950); '
                              'there is no CPU or PyTorch fallback for the kernels')
        lib = C.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, C.c_int)
        if lib.vqa_version() != ABI_VERSION:
            raise VqaHipError(f'{LIB_PATH} is ABI version {lib.vqa_version()}, this binding needs {ABI_VERSION}: '
                              'rebuild the library with `make`')
        _lib = lib
    return _lib


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_stream_handle = None


def stream() -> int:
    """HIP stream the kernels are enqueued on: torch's current stream, looked up once per
    public entry of the module (bind_stream) rather than once per kernel."""
    return _stream_handle if _stream_handle is not None else torch.cuda.current_stream().cuda_stream


def bind_stream() -> None:
    global _stream_handle
    _stream_handle = torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Invoke an entry point on the current stream; raise on a non-zero status."""
    lib = load()
    fn = getattr(lib, name)
    if name in _NO_STATUS:
        return fn(*args)
    rc = fn(*args, stream())
    if rc != 0:
        raise VqaHipError(f'{name} failed ({rc}): {lib.vqa_last_error_string().decode()}')
    return rc


class ColsumDesc(C.Structure):
    """VqaColsumDesc of include/vqa_hip.h."""
    _fields_ = [('X', C.c_void_p), ('ld', C.c_int), ('M', C.c_int), ('N', C.c_int), ('dtype', C.c_int),
                ('scale', C.c_float), ('out', C.c_void_p), ('out2', C.c_void_p)]


def colsum_multi(problems):
    """problems: iterable of (X, ld, M, N, dtype, out, out2-or-None[, scale]) with tensors for X / out / out2."""
    return call('vqa_colsum_multi', _colsum_array(problems), len(problems))


class SlabSumDesc(C.Structure):
    """VqaSlabSumDesc of include/vqa_hip.h."""
    _fields_ = [('slabs', C.c_void_p), ('nslab', C.c_int), ('slab_stride', C.c_size_t), ('count', C.c_long),
                ('out', C.c_void_p)]


def slab_sum_multi(problems):
    """problems: iterable of (slab address, nslab, stride, count, out tensor)."""
    arr = (SlabSumDesc * len(problems))()
    for d, (slabs, nslab, stride, count, out) in zip(arr, problems):
        d.slabs, d.nslab, d.slab_stride, d.count, d.out = slabs, nslab, stride, count, ptr(out)
    return call('vqa_slab_sum_multi', arr, len(problems))


class GemmDesc(C.Structure):
    """VqaGemmDesc of include/vqa_hip.h."""
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int), ('B', C.c_void_p), ('ldb', C.c_int),
                ('slabs', C.c_void_p), ('slab_stride', C.c_size_t), ('ld_slab', C.c_int),
                ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('splitk', C.c_int)]


def _colsum_array(problems):
    arr = (ColsumDesc * len(problems))()
    for d, pr in zip(arr, problems):
        X, ld, M, N, dt, out, out2 = pr[:7]
        d.X, d.ld, d.M, d.N, d.dtype, d.out, d.out2 = ptr(X), ld, M, N, dt, ptr(out), ptr(out2)
        d.scale = pr[7] if len(pr) > 7 else 1.0
    return arr


def gemm_slabs_group(problems, akm: int, bkm: int, dtype: int, colsums=None):
    """problems: iterable of (A address, lda, B address, ldb, slab address, slab_stride, ld_slab, M, N, K, splitk);
    colsums: optional problems of colsum_multi that ride in the same launch."""
    arr = (GemmDesc * len(problems))()
    for d, (A, lda, B, ldb, slabs, stride, ld_slab, M, N, K, sk) in zip(arr, problems):
        d.A, d.lda, d.B, d.ldb, d.slabs, d.slab_stride, d.ld_slab = A, lda, B, ldb, slabs, stride, ld_slab
        d.M, d.N, d.K, d.splitk = M, N, K, sk
    if colsums:
        return call('vqa_gemm_slabs_group_colsum', arr, len(problems), akm, bkm, dtype, _colsum_array(colsums), len(colsums))
    return call('vqa_gemm_slabs_group', arr, len(problems), akm, bkm, dtype)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise VqaHipError(f'unsupported storage dtype {dt}; use torch.float32 or torch.bfloat16')
