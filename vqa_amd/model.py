"""VQA attention / fusion / classifier module driven by the HIP kernels.

Keeps the reference module interface (vqa_model/models/model.py:29-79):
``Cls(vocab_questions, vocab_answers, args, input_size=512)``, attributes
``args / vocab_questions / vocab_answers / num_classes``, ``forward(input_v,
input_q[, captions]) -> (B, num_classes)`` un-normalised logits, ``train() /
eval()`` toggling dropout, ``parameters() / state_dict() / load_state_dict() /
to()``, with the reference's parameter names (``linear_q``, ``linear_v``,
``linear_classif1``, ``linear_classif2``, ``seq2vec.embedding``,
``seq2vec.rnn``).  The reference's image branch (a frozen ResNet-152 followed
by an unweighted mean over the grid, model.py:156-164) is replaced by
precomputed region features and question-conditioned attention over them.

All arithmetic runs in libvqa_hip.so; torch only owns the memory, the stream
and autograd's bookkeeping.  ``forward`` is differentiable through one
``torch.autograd.Function`` (so ``loss.backward(); optimizer.step()`` of the
reference's train.py:201-204 work unchanged); ``train_step`` is the fused path
(forward + softmax-CE + backward + gradient all-reduce + Adam) used for
throughput.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr

# reference constants (model.py:1585-1592)
VISUAL_OUT = 2048
QUESTION_OUT = 2400
FUSION_IN = 1200
FUSION_HIDDEN = 256
DROPOUT_V = DROPOUT_Q = DROPOUT_F = 0.5


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def site_seed(base_seed: int, step: int, site: int) -> int:
    """Seed of dropout site `site` (0: visual, 1: question, 2: fused, 3: hidden) at `step`."""
    x = (base_seed * 0x01000193 + step * 0x9E3779B1 + site * 0x7F4A7C15 + 0x165667B1) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x & 0xFFFFFFFF


class _Seq2Vec(nn.Module):
    """Parameter holder with the reference's names: ``embedding`` has len(vocab)+1 rows with the pad
    row 0; arch 'lstm' has ``rnn`` (seq2vec.py:27-38), arch '2-lstm' has ``rnn_0`` / ``rnn_1`` of half
    the output width each (TwoLSTM, seq2vec.py:48-59)."""

    def __init__(self, n_vocab: int, emb_size: int, out_size: int, gen: torch.Generator, arch: str = 'lstm'):
        super().__init__()
        self.arch = arch
        self.embedding = nn.Embedding(n_vocab + 1, emb_size, padding_idx=0)
        if arch == 'lstm':
            self.rnn = nn.LSTM(input_size=emb_size, hidden_size=out_size, num_layers=1)
            self.layers = [self.rnn]
        elif arch == '2-lstm':
            self.rnn_0 = nn.LSTM(input_size=emb_size, hidden_size=out_size // 2, num_layers=1)
            self.rnn_1 = nn.LSTM(input_size=out_size // 2, hidden_size=out_size // 2, num_layers=1)
            self.layers = [self.rnn_0, self.rnn_1]
        else:
            raise NotImplementedError(f"seq2vec arch {arch!r}: 'lstm' and '2-lstm' are built")
        with torch.no_grad():
            self.embedding.weight.copy_(torch.randn(self.embedding.weight.shape, generator=gen))
            self.embedding.weight[0].zero_()
            for rnn in self.layers:
                k = 1.0 / math.sqrt(rnn.hidden_size)
                for p in rnn.parameters():
                    p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * k)


class _Hop(nn.Module):
    def __init__(self, Dv: int, Hq: int, A: int, with_u: bool):
        super().__init__()
        self.linear_va = nn.Linear(Dv, A)
        self.linear_qa = nn.Linear(Hq, A)
        self.linear_a = nn.Linear(A, 1)
        if with_u:
            self.linear_u = nn.Linear(Dv, Hq)


class _VQAFunction(torch.autograd.Function):
    """The whole module forward as one autograd node; backward hands out the
    gradients the kernels wrote into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, feats, tokens, captions, *params):
        logits = model._forward_impl(feats, tokens, captions).clone()
        ctx.model = model
        ctx.n_params = len(params)
        ctx.generation = model._generation
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.generation != model._generation:
            # the activations this node needs live in the module's workspace and the forward that ran in
            # between has overwritten them (one forward is in flight per module)
            raise RuntimeError(
                f'{type(model).__name__}: backward() of forward #{ctx.generation} called after forward '
                f'#{model._generation} of the same module; the module keeps the activations of its latest forward '
                'only. Call backward() before the next forward (an evaluation forward included), as '
                'train.py:200-204 does, or use a second module instance.')
        model._backward_from_dense(dlogits)
        grads = [v.clone() for v in model._grad_views]
        return (None, None, None, None, *grads)


class _VQAHeadModel(nn.Module):
    """Question encoder + fusion head + classifier + loss / optimiser shared by the three model
    classes; a subclass supplies the visual front-end that turns its image input into the
    (B, Dv) vector entering ``linear_v`` (hooks ``_build_visual`` ... ``_visual_backward``)."""

    _dpool_from_slabs = False   # the front end's backward takes d(visual vector) as split-K partial sums

    def __init__(self, vocab_questions: Sequence[str], vocab_answers: Sequence[str], args=None,
                 input_size: int = 512, *, dtype: torch.dtype = torch.bfloat16, emb_size: int = 620,
                 question_out: int = QUESTION_OUT, visual_out: int = VISUAL_OUT,
                 attention_dim: int = FUSION_IN, fusion_in: int = FUSION_IN,
                 fusion_hidden: int = FUSION_HIDDEN, hops: int = 1, dropout: float = DROPOUT_F,
                 caption_vocab=0, caption_dim: int = 384, seed: int = 0, gemm_engine: int = 0,
                 seq2vec_arch: str = 'lstm', seq2vec_dropout: float = 0.3, train_encoder: bool = False):
        super().__init__()
        self.args = args
        self.vocab_questions = vocab_questions
        self.vocab_answers = vocab_answers
        self.num_classes = len(self.vocab_answers)
        self.input_size = input_size
        self.compute_dtype = dtype
        self.dt = _lib.dtype_code(dtype)
        self.E, self.Hq, self.Dv = emb_size, question_out, visual_out
        self.A, self.F, self.Hf = attention_dim, fusion_in, fusion_hidden
        self.hops, self.drop_p = hops, float(dropout)
        # caption_vocab: the number of caption words, or the word list itself (captions may then be passed to forward
        # as the reference passes them: the collated strings of model.py:1673)
        self.caption_tokenizer = None
        if not isinstance(caption_vocab, int):
            from .captions import CaptionTokenizer
            self.caption_tokenizer = caption_vocab if isinstance(caption_vocab, CaptionTokenizer) else CaptionTokenizer(caption_vocab)
            caption_vocab = len(self.caption_tokenizer)
        self.Ec = caption_dim if caption_vocab > 0 else 0
        self.caption_vocab = caption_vocab
        self.gemm_engine = gemm_engine
        for name, v in (('question_out', self.Hq), ('visual_out', self.Dv), ('attention_dim', self.A),
                        ('fusion_in', self.F), ('fusion_hidden', self.Hf), ('caption_dim', self.Ec)):
            if v % 8:
                raise ValueError(f'{name}={v} must be a multiple of 8 (16-byte bf16 chunks)')
        if hops < 1:
            raise ValueError('hops must be >= 1')

        gen = torch.Generator().manual_seed(seed)
        if seq2vec_arch == '2-lstm' and question_out % 16:
            raise ValueError('question_out must be a multiple of 16 for the 2-lstm encoder (two halves of 8-element chunks)')
        self.seq2vec = _Seq2Vec(len(vocab_questions), emb_size, question_out, gen, seq2vec_arch)
        self.seq2vec_dropout = float(seq2vec_dropout)      # F.dropout(p=0.3) on each TwoLSTM vector, seq2vec.py:73-74
        self.train_encoder = bool(train_encoder)
        if self.train_encoder and emb_size > 1024:
            raise ValueError('a trainable question encoder needs emb_size <= 1024')
        for p in self.seq2vec.parameters():          # frozen as model.py:42-43 unless train_encoder
            p.requires_grad = self.train_encoder
        self._build_visual(gen)
        self.linear_q = nn.Linear(self.Hq, self.F)
        self.linear_v = nn.Linear(self.Dv, self.F)
        self.linear_classif1 = nn.Linear(self.F + self.Ec, self.Hf)
        self.linear_classif2 = nn.Linear(self.Hf, self.num_classes)
        if caption_vocab > 0:
            self.caption_embedding = nn.Embedding(caption_vocab + 1, caption_dim, padding_idx=0)
            self.caption_embedding.weight.requires_grad = False
            with torch.no_grad():
                self.caption_embedding.weight.copy_(torch.randn(self.caption_embedding.weight.shape, generator=gen))
                self.caption_embedding.weight[0].zero_()
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    b = 1.0 / math.sqrt(m.in_features)
                    m.weight.copy_((torch.rand(m.weight.shape, generator=gen) * 2 - 1) * b)
                    m.bias.copy_((torch.rand(m.bias.shape, generator=gen) * 2 - 1) * b)

        self.base_seed = seed
        self.dropout_rank = 0         # data-parallel rank: decorrelates the dropout masks of the ranks (dropout_seed)
        self.step_count = 0           # optimiser steps taken by train_step
        self.drop_step = 0            # dropout stream position
        self._flat = None
        self._ws: Dict[Tuple, Dict[str, torch.Tensor]] = {}
        self._slab = None
        self._frozen_key = None
        self._shadow_fresh = False
        self._ctx = None
        self._generation = 0          # forwards run so far; an autograd node belongs to exactly one of them
        self.fold_first_step = True   # ... and the first recurrent product (h_0 depends on one token)
        self.fuse_first_two_cells = True   # steps 0 and 1 then run as one launch
        self.fold_embedding = True    # frozen encoder: embedding x first input projection tabulated per token id
        self.collective_cus = 0       # compute units a gradient collective's kernel may hold beside backward (train_step sets it)
        self.resident_encoder = True  # ... and steps 2.. of a single bf16 layer in one launch, W_hh resident in registers
        self.resident_whole = True    # ... which then also packs the sequences and runs the two tabulated steps itself
        self._xg_table_u = self._resident_sync = None
        self.grad_ready_hook = None   # callable(bucket_index) fired as gradient buckets complete
        self.kernel_events = None     # set to {} to collect per-kernel HIP event pairs
        self.kernel_event_stride = 1  # time every n-th launch only (an event pair costs the stream ~10 us of gaps)
        self._event_calls: Dict[str, int] = {}
        self._pending = None          # last gradient bucket of the previous train_step still in flight
        self._bptt = False            # the running forward keeps the encoder state for backward through time
        self._loss_pending = False    # the mean of the per-row losses is still to be reduced (by backward)
        self._wslab, self._wslab_off, self._wgrad_pending = None, 0, []
        self._held_group = None       # the head's weight / bias gradient problems waiting for the front end's launch

    # ------------------------------------------------------------------ flat parameter storage
    def _trainable_order(self) -> List[Tuple[str, nn.Parameter]]:
        """Trainable parameters in the order backward completes them (classifier
        first, image projection last); gradient buckets follow this order."""
        out = [('linear_classif2.weight', self.linear_classif2.weight), ('linear_classif2.bias', self.linear_classif2.bias),
               ('linear_classif1.weight', self.linear_classif1.weight), ('linear_classif1.bias', self.linear_classif1.bias),
               ('linear_v.weight', self.linear_v.weight), ('linear_v.bias', self.linear_v.bias),
               ('linear_q.weight', self.linear_q.weight), ('linear_q.bias', self.linear_q.bias)]
        return out + self._visual_trainable_order() + self._encoder_trainable_order()

    def _enc_names(self):
        return ['rnn'] if self.seq2vec.arch == 'lstm' else ['rnn_0', 'rnn_1']

    def _encoder_trainable_order(self):
        """Question-encoder parameters (when trainable) in the order backward through time completes
        them: last layer first, the embedding table last."""
        if not self.train_encoder:
            return []
        out = []
        for nm, rnn in reversed(list(zip(self._enc_names(), self.seq2vec.layers))):
            pre = f'seq2vec.{nm}.'
            out += [(pre + 'bias_ih_l0', rnn.bias_ih_l0), (pre + 'bias_hh_l0', rnn.bias_hh_l0),
                    (pre + 'weight_hh_l0', rnn.weight_hh_l0), (pre + 'weight_ih_l0', rnn.weight_ih_l0)]
        return out + [('seq2vec.embedding.weight', self.seq2vec.embedding.weight)]

    def _materialize(self, device: torch.device):
        """Move trainable parameters into one flat fp32 buffer (parameters become
        views of it) with matching flat gradient / bf16 shadow buffers."""
        order = self._trainable_order()
        ok = self._flat is not None and self._flat.device == device
        if ok:
            for (name, p), (o, n) in zip(order, self._offsets):
                if p.data_ptr() != self._flat.data_ptr() + 4 * o or p.numel() != n:
                    ok = False
                    break
        if ok:
            return
        # Adam moments of an earlier layout (the module was moved after training had started) follow by name
        moments = None
        if getattr(self, '_exp_avg', None) is not None and getattr(self, '_index', None):
            moments = {name: (self._exp_avg[o:o + n].to(device), self._exp_avg_sq[o:o + n].to(device))
                       for name, (o, n, _) in self._index.items()}
        offsets, total = [], 0
        for _, p in order:
            offsets.append((total, p.numel()))
            total += _pad8(p.numel())
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grad = torch.zeros(total, dtype=torch.float32, device=device)
        self._grad_views = []
        self._index = {}
        for (name, p), (o, n) in zip(order, offsets):
            flat[o:o + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[o:o + n].view(p.shape)
            self._grad_views.append(grad[o:o + n].view(p.shape))
            self._index[name] = (o, n, tuple(p.shape))
        self._flat, self._flat_grad, self._offsets = flat, grad, offsets
        self._shadow = torch.zeros(total, dtype=torch.bfloat16, device=device) if self.dt == _lib.BF16 else None
        self._exp_avg = self._exp_avg_sq = None
        if moments is not None and all(name in moments and moments[name][0].numel() == n
                                       for name, (o, n, _) in self._index.items()):
            self._exp_avg, self._exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
            for name, (o, n, _) in self._index.items():
                self._exp_avg[o:o + n].copy_(moments[name][0])
                self._exp_avg_sq[o:o + n].copy_(moments[name][1])
        self._shadow_fresh = False
        self._shadow_version = None
        # bucket boundaries (element ranges of the flat buffer) in completion order
        names = [n for n, _ in order]
        enc = [self._encoder_trainable_order()[0][0]] if self.train_encoder else []
        cuts = sorted({names.index(n) for n in self._visual_bucket_starts() + enc})
        starts = [0] + [offsets[c][0] for c in cuts]
        ends = starts[1:] + [total]
        self.buckets = [(s, e) for s, e in zip(starts, ends) if e > s]
        self._bucket_of = {}
        for (name, _), (o, n) in zip(order, offsets):
            for bi, (s, e) in enumerate(self.buckets):
                if s <= o < e:
                    self._bucket_of[name] = bi

    def _w(self, name: str):
        """Device pointer + tensor of the compute-dtype copy of a trainable weight."""
        o, n, shape = self._index[name]
        src = self._shadow if self.dt == _lib.BF16 else self._flat
        return src[o:o + n]

    def _f32(self, name: str):
        o, n, _ = self._index[name]
        return self._flat[o:o + n]

    def _g(self, name: str):
        o, n, _ = self._index[name]
        return self._flat_grad[o:o + n]

    def _param_versions(self):
        """Version counters of the flat buffer and of every parameter view of it.  The parameters are attached
        with ``p.data = view``, which gives each its own counter: torch.optim.Adam.step() (train.py:204),
        ``p.add_()`` or ``p.copy_()`` move the parameter's counter and leave the flat buffer's alone."""
        return (self._flat._version,) + tuple(p._version for _, p in self._trainable_order())

    def _mark_shadow_fresh(self):
        """The fused Adam kernel has just written the bf16 copies itself (through raw pointers: no counter moved)."""
        self._shadow_fresh = self.dt == _lib.BF16
        self._shadow_version = self._param_versions()

    def _refresh_shadow(self):
        """bf16 copies of the trainable weights.  The fused Adam kernel writes them itself; any torch-side in-place
        update of a parameter (the reference's optimizer.step(), train.py:204) moves that parameter's version
        counter and triggers a re-cast here."""
        if self.dt != _lib.BF16:
            return
        versions = self._param_versions()
        if self._shadow_fresh and self._shadow_version == versions:
            return
        n = self._flat.numel()
        call('vqa_cast_pad', ptr(self._flat), n, ptr(self._shadow), n, 1, n, _lib.BF16)
        self._shadow_fresh = True
        self._shadow_version = versions

    def _refresh_frozen(self, device):
        """Compute-dtype copies of the frozen question / caption encoder weights: embedding table and
        every W_ih zero-padded to a multiple of 8 columns, b_ih + b_hh summed, per LSTM layer."""
        ps = [self.seq2vec.embedding.weight]
        for rnn in self.seq2vec.layers:
            ps += [rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0]
        if self.Ec:
            ps.append(self.caption_embedding.weight)
        key = tuple((p.data_ptr(), p._version) for p in ps) + (str(device), self.fold_embedding, self.fold_first_step, self.train_encoder, self.resident_encoder)
        if key == self._frozen_key:
            return
        tdt = self.compute_dtype

        def cast_pad(src, ld_dst):
            src = src.detach().to(device=device, dtype=torch.float32).contiguous()
            dst = torch.empty((src.shape[0], ld_dst), dtype=tdt, device=device)
            call('vqa_cast_pad', ptr(src), src.shape[1], ptr(dst), ld_dst, src.shape[0], src.shape[1], self.dt)
            return dst

        self._emb_t = cast_pad(ps[0], _pad8(self.E))
        self._enc_layers = []
        for rnn in self.seq2vec.layers:
            self._enc_layers.append((cast_pad(rnn.weight_ih_l0, _pad8(rnn.input_size)),
                                     cast_pad(rnn.weight_hh_l0, rnn.hidden_size),
                                     (rnn.bias_ih_l0.detach().float() + rnn.bias_hh_l0.detach().float()).to(device).contiguous()))
        if self.Ec:
            self._cap_t = cast_pad(ps[-1], self.Ec)
        # Frozen encoder: embedding followed by the first layer's input projection is one fixed map per token id, so
        # it is tabulated here, once per weight version: xg_table[v] = act(emb[v]) W_ih^T + b_ih + b_hh, computed by
        # the same GEMM kernel on the same rows as the per-step projection it replaces (bit-identical values).
        self._xg_table = self._hh_table = self._xg_table_u = self._resident_sync = None
        V1, H4 = self._emb_t.shape[0], 4 * self.seq2vec.layers[0].hidden_size
        if self.fold_embedding and not self.train_encoder and V1 * H4 * self._emb_t.element_size() <= (1 << 30):
            wih, _, bias = self._enc_layers[0]
            Ep = _pad8(self.E)
            emb = self._emb_t
            if self.seq2vec.arch == '2-lstm':             # TwoLSTM feeds tanh(embedding) (seq2vec.py:63-64)
                emb = torch.empty_like(self._emb_t)
                ident = torch.arange(V1, device=device, dtype=torch.int64).view(V1, 1).contiguous()
                one = torch.zeros(1, dtype=torch.int32, device=device)
                offs = torch.tensor([0, V1], dtype=torch.int32, device=device)
                call('vqa_embedding_gather_packed', ptr(ident), ptr(self._emb_t), ptr(emb), V1, 1, Ep, Ep, Ep, V1,
                     ptr(torch.arange(V1, device=device, dtype=torch.int32)), ptr(offs), 1, None, self.dt)
            table = torch.empty((V1, H4), dtype=tdt, device=device)
            call('vqa_gemm_fused', ptr(emb), Ep, 1, ptr(wih), Ep, 1, ptr(table), H4, V1, H4, Ep, ptr(bias), None, 0,
                 self.dt, self.dt, self.gemm_engine)
            self._xg_table = table
            # ... and so is the first recurrent product: h_0 is a function of the first token alone, so
            # hh_table[v] = h_0(v) W_hh^T replaces the recurrent GEMM of step 1 (the most expensive one: all rows live,
            # W_hh cold).  fp32, one K range: equal to the per-step split-K sum up to fp32 rounding of the summation order.
            self._hh_table = None
            Hh = H4 // 4
            if self.fold_first_step and 3 * V1 * H4 * 4 <= (2 << 30):
                _, whh, _ = self._enc_layers[0]
                h0 = torch.empty((V1, Hh), dtype=tdt, device=device)
                c0 = torch.empty((V1, Hh), dtype=torch.float32, device=device)
                lens = torch.full((V1,), 2, dtype=torch.int32, device=device)     # (no row ends at step 0: q is not written)
                dq32 = torch.empty((1, Hh), dtype=torch.float32, device=device)
                dqt = torch.empty((1, Hh), dtype=tdt, device=device)
                call('vqa_lstm_cell', None, 0, 0, ptr(table), H4, ptr(c0), ptr(h0), Hh, ptr(lens), ptr(dq32), ptr(dqt), Hh,
                     V1, Hh, 0, 2, self.dt)
                hh = torch.empty((V1, H4), dtype=torch.float32, device=device)
                call('vqa_gemm_fused', ptr(h0), Hh, 1, ptr(whh), Hh, 1, ptr(hh), H4, V1, H4, Hh, None, None, 0,
                     _lib.F32, self.dt, self.gemm_engine)
                self._hh_table = hh
                # All later steps in one launch with W_hh resident in registers (vqa_lstm_resident): single layer,
                # bf16, as many free compute units as the layer has groups of 12 hidden units.  Its table has the four
                # gates of a unit adjacent (one 8-byte gather per cell).
                nwg = call('vqa_lstm_resident_workgroups', Hh)
                if (self.resident_encoder and self.dt == _lib.BF16 and len(self._enc_layers) == 1 and nwg > 0 and
                        nwg <= torch.cuda.get_device_properties(device).multi_processor_count):
                    self._xg_table_u = table.view(V1, 4, Hh).transpose(1, 2).contiguous()
                    self._resident_sync = torch.zeros(_lib.RESIDENT_SYNC_WORDS, dtype=torch.int64, device=device)
        self._frozen_key = key

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B: int, geom, L: int, Tc: int, device) -> Dict[str, torch.Tensor]:
        """Per-shape buffers; `geom` is the front-end's geometry (region count / grid sizes)."""
        key = (B, geom, L, Tc, str(device))
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        tdt, f32 = self.compute_dtype, torch.float32
        Ep, H, A, Dv, F, Hf, Ec = _pad8(self.E), self.Hq, self.A, self.Dv, self.F, self.Hf, self.Ec
        Cp = _pad8(self.num_classes)

        def z(shape, dt):
            return torch.zeros(shape, dtype=dt, device=device)

        ws = {}
        self._encoder_buffers(ws, z, B, L)
        ws.update({
            'pooled_d': z((B, Dv), tdt), 'q_d': z((B, H), tdt),
            'xv': z((B, F), f32), 'xq': z((B, F), f32), 'x': z((B, F + Ec), tdt),
            'h1t': z((B, Hf), f32), 'h1d': z((B, Hf), tdt), 'logits': z((B, Cp), f32),
            'loss_rows': z((B,), f32), 'loss': z((1,), f32), 'dlogits': z((B, Cp), tdt),
            'dz1': z((B, Hf), tdt), 'dzv': z((B, F), tdt), 'dzq': z((B, F), tdt),
            'd_pooled': z((B, Dv), f32),
            'pred': z((B,), torch.int64),
        })
        self._visual_workspace(ws, z, B, geom)
        self._ws[key] = ws
        return ws

    def _slabs(self, nfloats: int, device) -> torch.Tensor:
        if self._slab is None or self._slab.numel() < nfloats or self._slab.device != device:
            self._slab = torch.empty(max(nfloats, 1 << 20), dtype=torch.float32, device=device)
        return self._slab

    # ------------------------------------------------------------------ kernel helpers
    def _gemm_slabs(self, A, lda, akm, Bm, ldb, bkm, M, N, K, active_rows=None, active_k=None, sk_max=None):
        """fp32 partial sums of A B^T into the slab workspace -> (slab tensor, nslab, stride).
        active_rows: device address of an int32 row count past which output tiles are skipped."""
        sk = call('vqa_gemm_splitk_hint', M, N, K, akm, bkm)
        if sk_max is not None:
            sk = max(1, min(sk, sk_max))
        stride = M * N
        slab = self._slabs(sk * stride, A.device)
        call('vqa_gemm_slabs_rows', ptr(A), lda, akm, ptr(Bm), ldb, bkm, ptr(slab), stride, N, M, N, K, sk,
             self.dt, self.gemm_engine, active_rows, active_k)
        return slab, sk, stride

    def _linear(self, x, ldx, wname, bname, M, N, K, *, act=0, out_pre=None, ld_pre=0, out_post=None, ld_post=0,
                post_dtype=None, addend=None, ld_add=0, seed=0, p=0.0):
        """epilogue(x W^T + b) for a small-M product (split-K slabs + fused reduce)."""
        if (act == 0 and addend is None and out_post is None and out_pre is not None
                and call('vqa_gemm_splitk_hint', M, N, K, 1, 1) == 1):
            # one K range and a plain bias: the product's own epilogue writes the result (no reduce launch)
            call('vqa_gemm_fused', ptr(x), ldx, 1, ptr(self._w(wname)), K, 1, ptr(out_pre), ld_pre, M, N, K,
                 ptr(self._f32(bname)) if bname else None, None, 0, _lib.F32, self.dt, self.gemm_engine)
            return
        slab, sk, stride = self._gemm_slabs(x, ldx, 1, self._w(wname), K, 1, M, N, K)
        call('vqa_epilogue', ptr(slab), sk, stride, M, N, ptr(self._f32(bname)) if bname else None, None,
             ptr(addend), ld_add, act, ptr(out_pre), ld_pre, ptr(out_post), ld_post,
             self.dt if post_dtype is None else post_dtype, seed, p)

    def _wgrad(self, dy, ld_dy, x, ldx, wname, M, N, K, active_k=None, defer=False, swapped=False):
        """grad(W)[M,N] = dy^T x with dy stored [K][M], x stored [K][N].  defer: the split-K partial sums go to
        an arena of their own and are reduced later, together with the other deferred ones, by _flush_wgrads().
        swapped: the product runs as x^T dy (N x M) and the split-K reduction transposes it back."""
        if swapped:
            ev = self._event_pair('wgrad:' + wname)
            # every workgroup of this product runs for the whole launch, one per CU: while a collective's kernel holds
            # compute units (collective_cus, set by train_step), one workgroup too many would wait for a second round
            # and double the launch, so the K ranges are cut to what fits beside it (6 -> 5 ranges: +3 %)
            sk_max = None
            if self.collective_cus:
                cus = torch.cuda.get_device_properties(x.device).multi_processor_count
                sk_max = max(1, (cus - self.collective_cus) // (((N + 255) // 256) * ((M + 255) // 256)))
            slab, sk, stride = self._gemm_slabs(x, ldx, 0, dy, ld_dy, 0, N, M, K, active_k=active_k, sk_max=sk_max)
            if ev is not None:
                ev[1].record()
            call('vqa_slab_sum_transpose', ptr(slab), sk, stride, N, M, ptr(self._g(wname)), N)
            return
        sk = call('vqa_gemm_splitk_hint', M, N, K, 0, 0)
        if sk == 1:                 # a single K range: the product is the gradient, written in place
            call('vqa_gemm_slabs_rows', ptr(dy), ld_dy, 0, ptr(x), ldx, 0, ptr(self._g(wname)), M * N, N, M, N, K, 1,
                 self.dt, self.gemm_engine, None, active_k)
            return
        if defer:
            stride = M * N
            addr = self._arena(sk * stride, dy.device)
            call('vqa_gemm_slabs_rows', ptr(dy), ld_dy, 0, ptr(x), ldx, 0, addr, stride, N, M, N, K, sk,
                 self.dt, self.gemm_engine, None, active_k)
            self._wgrad_pending.append((addr, sk, stride, stride, self._g(wname)))
            return
        slab, sk, stride = self._gemm_slabs(dy, ld_dy, 0, x, ldx, 0, M, N, K, active_k=active_k)
        call('vqa_epilogue', ptr(slab), sk, stride, M, N, None, None, None, 0, 0, ptr(self._g(wname)), N,
             None, 0, _lib.F32, 0, 0.0)

    def _wgrad_grouped(self, items, colsums=None):
        """Several weight gradients (dy, ld_dy, x, ldx, wname, M, N, K), each grad(W)[M,N] = dy^T x, as one launch
        when their contraction is short; a product with one K range is written straight into the gradient buffer,
        the others leave partial sums for _flush_wgrads()."""
        if self.dt != _lib.BF16 or any(K > 512 for *_, K in items) or len(items) > 8:
            if colsums:
                _lib.colsum_multi(colsums)
            for dy, ld_dy, x, ldx, wname, M, N, K in items:
                self._wgrad(dy, ld_dy, x, ldx, wname, M, N, K, defer=True)
            return
        # the group's whole arena need is reserved in one piece: a flush in the middle of the list (the arena
        # running full) would reduce partial sums of products that have not been launched yet
        sks = [call('vqa_gemm_splitk_hint', M, N, K, 0, 0) for *_, M, N, K in items]
        need = sum((sk * M * N + 63) // 64 * 64 for sk, (*_, M, N, K) in zip(sks, items) if sk > 1)
        base = self._arena(need, items[0][0].device) if need else 0
        problems, off = [], 0
        for sk, (dy, ld_dy, x, ldx, wname, M, N, K) in zip(sks, items):
            if sk == 1:
                addr = ptr(self._g(wname))
            else:
                addr = base + 4 * off
                off += (sk * M * N + 63) // 64 * 64
                self._wgrad_pending.append((addr, sk, M * N, M * N, self._g(wname)))
            problems.append((ptr(dy), ld_dy, ptr(x), ldx, addr, M * N, N, M, N, K, sk))
        _lib.gemm_slabs_group(problems, 0, 0, self.dt, colsums)       # (bias gradients ride in the same launch)

    def _q_side_products(self, ws):
        """Forward products whose input is ready as soon as the question is encoded: (key, x, ldx, weight name, N, K).
        They run as one launch before the image side; their partial sums wait in ws['early'] for the epilogue."""
        return [('linear_q', ws['q_d'], self.Hq, 'linear_q.weight', self.F, self.Hq)]

    def _launch_early(self, ws, B):
        items = self._q_side_products(ws)
        ws['early'] = {}
        if self.dt != _lib.BF16 or not items:
            return
        problems, off = [], 0
        sizes = []
        for key, x, ldx, wname, N, K in items:
            sk = call('vqa_gemm_splitk_hint', B, N, K, 1, 1)
            sizes.append((sk, sk * B * N))
        total = sum(n for _, n in sizes)
        buf = ws.get('early_buf')
        if buf is None or buf.numel() < total:
            buf = ws['early_buf'] = torch.empty(total, dtype=torch.float32, device=x.device)
        for (key, x, ldx, wname, N, K), (sk, n) in zip(items, sizes):
            addr = buf.data_ptr() + 4 * off
            off += n
            problems.append((ptr(x), ldx, ptr(self._w(wname)), K, addr, B * N, N, B, N, K, sk))
            ws['early'][key] = (addr, sk, B * N)
        _lib.gemm_slabs_group(problems, 1, 1, self.dt)

    def _arena(self, nfloats: int, device) -> int:
        """Address of nfloats floats that stay untouched until _flush_wgrads()."""
        if self._wslab is None or self._wslab.device != device or self._wslab_off + nfloats > self._wslab.numel():
            self._flush_wgrads()
            if self._wslab is None or self._wslab.device != device or nfloats > self._wslab.numel():
                self._wslab = torch.empty(max(2 * nfloats, 16 << 20), dtype=torch.float32, device=device)
        addr = self._wslab.data_ptr() + 4 * self._wslab_off
        self._wslab_off += (nfloats + 63) // 64 * 64
        return addr

    def _flush_wgrads(self):
        """Reduce the deferred weight-gradient partial sums into the flat gradient buffer, 8 per launch."""
        pend, self._wgrad_pending, self._wslab_off = self._wgrad_pending, [], 0
        for i in range(0, len(pend), 8):
            _lib.slab_sum_multi(pend[i:i + 8])

    def _bucket_done(self, name: str):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(self._bucket_of[name])

    def _event_pair(self, name: str):
        """When kernel timing is switched on (self.kernel_events = {}), a (start, end) pair of HIP
        events on the launch stream around one kernel; the start event is recorded here."""
        if self.kernel_events is None:
            return None
        if name not in self._event_calls:      # kernels are sampled on different steps: an event pair costs the stream ~10 us
            self._event_calls[name] = len(self._event_calls)
        n = self._event_calls[name]
        self._event_calls[name] = n + 1
        if n % self.kernel_event_stride:
            return None
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        self.kernel_events.setdefault(name, []).append(pair)
        pair[0].record()
        return pair

    # ------------------------------------------------------------------ forward
    def _prepare_inputs(self, input_v, input_q, captions):
        dev = input_q.device
        if dev.type != 'cuda':
            raise _lib.VqaHipError(f'{type(self).__name__} runs on a HIP device only (inputs must be cuda tensors); '
                                   'there is no CPU path')
        feats, nb = self._prepare_visual(input_v)
        tokens = input_q.long().contiguous()
        if tokens.dim() != 2 or tokens.shape[0] != nb:
            raise ValueError('question_tokens must be (B, L) with the batch of the image input')
        if self.Ec:
            if captions is None:
                raise ValueError('this model was built with a caption stream: pass captions (B, Tc) int64 or the '
                                 'collated caption strings')
            if not torch.is_tensor(captions):            # the reference's form: strings (model.py:1673)
                if self.caption_tokenizer is None:
                    raise ValueError('caption strings need the model built with the caption word list '
                                     '(caption_vocab=[...] or a CaptionTokenizer)')
                captions = self.caption_tokenizer.encode(captions, device=dev)
            captions = captions.long().contiguous()
            if captions.dim() != 2 or captions.shape[0] != nb:
                raise ValueError('captions must be (B, Tc) with the batch of the image input')
        else:
            captions = None
        return feats, tokens, captions

    def forward(self, input_v, input_q, captions=None):
        feats, tokens, captions = self._prepare_inputs(input_v, input_q, captions)
        if tokens.shape[0] == 0:                     # empty batch: nothing to launch
            return torch.zeros((0, self.num_classes), dtype=torch.float32, device=tokens.device)
        self._materialize(tokens.device)
        params = [p for _, p in self._trainable_order()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._bptt = True
            return _VQAFunction.apply(self, feats, tokens, captions, *params)
        # the returned tensor is a copy: the workspace is reused by the next call
        self._bptt = False
        return self._forward_impl(feats, tokens, captions).clone()

    def _drop_seeds(self):
        """Seeds of the dropout sites of one training forward: 0 visual, 1 question, 2 fused, 3 hidden,
        4 / 5 the two vectors of the 2-lstm encoder."""
        if not self.training:
            return (0,) * 6, 0.0
        self.drop_step += 1
        return tuple(site_seed(self.dropout_seed(), self.drop_step, s) for s in range(6)), self.drop_p

    def dropout_seed(self) -> int:
        """Base seed of the dropout stream.  The masks hash (seed, local row, column), so data-parallel ranks that
        share a seed would drop the same elements of their local samples: each rank offsets the seed by its
        ``dropout_rank`` (set it to the process rank; weights still initialise from the shared ``base_seed``)."""
        return (self.base_seed + 1000003 * int(self.dropout_rank)) & 0x7FFFFFFF

    def encode_question(self, tokens: torch.Tensor, ws=None, seeds=None, q_drop=None) -> torch.Tensor:
        """Embedding gather + LSTM layer(s) + last-valid-step selection -> q (B, Hq) fp32
        (seq2vec.py:40-45; with the '2-lstm' encoder seq2vec.py:61-76).  `seeds`: the two dropout seeds
        of the 2-lstm vectors in training mode."""
        B, L = tokens.shape
        dev = tokens.device
        _lib.bind_stream()
        self._refresh_frozen(dev)
        if ws is None:
            ws = self._enc_workspace(B, L, dev)
        Ep, Hq, dt = _pad8(self.E), self.Hq, self.dt
        two = self.seq2vec.arch == '2-lstm'
        fuse_drop = q_drop is not None and not two     # (buffer, seed, p): the head's dropout of q written by the cell kernel
        table = self._xg_table
        Hh0 = self._enc_layers[0][1].shape[1]
        if (self.resident_whole and self._xg_table_u is not None and self._hh_table is not None and 2 < L <= _lib.RESIDENT_MAX_STEPS and
                B <= _lib.RESIDENT_MAX_ROWS and not (self.train_encoder and self._bptt)):
            # the whole frozen encoder as ONE launch: packing, the two tabulated steps and the recurrent steps
            if 'h_ring' not in ws:
                ws['h_ring'] = torch.zeros((2, _lib.RESIDENT_PITCH // 8, _lib.resident_ring_rows(B), 8), dtype=self.compute_dtype, device=dev)
            call('vqa_lstm_resident', ptr(self._enc_layers[0][1]), Hh0, ptr(self._xg_table_u), ptr(self._hh_table), table.shape[0],
                 ptr(tokens), None, ptr(ws['h_ring']), ptr(ws['len']), ptr(ws['q32']), ptr(ws['qT']), Hq, B, Hh0, 0, L,
                 ptr(ws['perm']), ptr(ws['m_t']), ptr(q_drop[0]) if fuse_drop else None, q_drop[1] if fuse_drop else 0,
                 q_drop[2] if fuse_drop else 0.0, ptr(self._resident_sync), ptr(self._status_words(dev)))
            return ws['q32']
        # packed sequences: rows sorted by length, so the rows still running at step t are the first
        # m_t[t] and their input projections are rows offs[t].. of xg; padding steps cost nothing
        call('vqa_sequence_pack', ptr(tokens), B, L, ptr(ws['len']), ptr(ws['perm']), ptr(ws['m_t']), ptr(ws['offs']),
             self._emb_t.shape[0], ptr(self._status_words(dev)))
        table = self._xg_table                  # frozen encoder: gather + first input projection come from one table
        if table is None:
            call('vqa_embedding_gather_packed', ptr(tokens), ptr(self._emb_t), ptr(ws['x_emb']), B, L, Ep, Ep, Ep,
                 self._emb_t.shape[0], ptr(ws['perm']), ptr(ws['offs']), 1 if two else 0, ptr(ws.get('ptok')), dt)
        total_rows = ws['offs'].data_ptr() + 4 * L
        x, kin = ws['x_emb'], Ep
        esz = ws['qT'].element_size()
        for li, (wih, whh, bias) in enumerate(self._enc_layers):
            Hh = whh.shape[1]
            last_layer = li + 1 == len(self._enc_layers)
            save = self.train_encoder and self._bptt
            hs = ws[f'hs{li}'] if save else (None if last_layer else ws['hs'])
            gates, cs = (ws[f'gates{li}'], ws[f'cs{li}']) if save else (None, None)
            from_table = li == 0 and table is not None
            if not from_table:
                call('vqa_gemm_fused_rows', ptr(x), kin, 1, ptr(wih), kin, 1, ptr(ws['xg']), 4 * Hh,
                     L * B, 4 * Hh, kin, ptr(bias), None, 0, dt, dt, self.gemm_engine, total_rows)
            q32 = ws['q32'].data_ptr() + 4 * li * Hh            # this layer's columns of q
            qT = ws['qT'].data_ptr() + esz * li * Hh
            t_first = 0
            if from_table and self._hh_table is not None and L >= 2 and self.fuse_first_two_cells:   # no GEMM between them
                resident = (self._xg_table_u is not None and hs is None and 2 < L <= _lib.RESIDENT_MAX_STEPS and B <= _lib.RESIDENT_MAX_ROWS)
                if resident and 'h_ring' not in ws:      # two slots of h, chunk-major, K padded with zeros to the kernel's contraction length
                    ws['h_ring'] = torch.zeros((2, _lib.RESIDENT_PITCH // 8, _lib.resident_ring_rows(B), 8), dtype=self.compute_dtype, device=dev)
                call('vqa_lstm_cell01_table', ptr(table), 4 * Hh, ptr(self._hh_table), table.shape[0], ptr(tokens),
                     ptr(ws['c']), ptr(ws['h_ring'] if resident else ws['h']), Hh, ptr(ws['len']), q32, qT, Hq, B, Hh, L,
                     ptr(ws['perm']), ptr(ws['m_t']), ptr(ws['offs']), ptr(hs), Hh,
                     ptr(q_drop[0]) if fuse_drop else None, q_drop[1] if fuse_drop else 0,
                     q_drop[2] if fuse_drop else 0.0, _lib.resident_ring_rows(B) if resident else 0, dt)
                t_first = 2
                if resident:                             # steps 2 .. L-1 in one launch, W_hh resident in registers
                    call('vqa_lstm_resident', ptr(whh), Hh, ptr(self._xg_table_u), None, table.shape[0], ptr(tokens),
                         ptr(ws['c']), ptr(ws['h_ring']), ptr(ws['len']), q32, qT, Hq, B, Hh, 2, L, ptr(ws['perm']),
                         ptr(ws['m_t']), ptr(q_drop[0]) if fuse_drop else None, q_drop[1] if fuse_drop else 0,
                         q_drop[2] if fuse_drop else 0.0, ptr(self._resident_sync), ptr(self._status_words(dev)))
                    t_first = L
            h_cur = ws['h']
            for t in range(t_first, L):
                hh_tab = self._hh_table if (from_table and t == 1) else None     # step 1's recurrent product, tabulated
                if t == 0 or hh_tab is not None:
                    slab, sk, stride = None, 0, 0
                else:
                    slab, sk, stride = self._gemm_slabs(h_cur, Hh, 1, whh, Hh, 1, B, 4 * Hh, Hh,
                                                        active_rows=ws['m_t'].data_ptr() + 4 * t)
                if from_table:
                    call('vqa_lstm_cell_packed_table', ptr(slab), sk, stride, ptr(table), 4 * Hh, table.shape[0], ptr(tokens),
                         ptr(ws['c']), ptr(h_cur), Hh, ptr(ws['len']), q32, qT, Hq, B, Hh, t, L, ptr(ws['perm']),
                         ptr(ws['m_t']), ptr(ws['offs']), ptr(hs), Hh,
                         ptr(q_drop[0]) if fuse_drop else None, q_drop[1] if fuse_drop else 0,
                         q_drop[2] if fuse_drop else 0.0, ptr(hh_tab), dt)
                    continue
                call('vqa_lstm_cell_packed', ptr(slab), sk, stride, ptr(ws['xg']), 4 * Hh, ptr(ws['c']), ptr(h_cur), Hh,
                     ptr(ws['len']), q32, qT, Hq, B, Hh, t, L, ptr(ws['perm']), ptr(ws['m_t']),
                     ptr(ws['offs']), ptr(hs), Hh, ptr(gates), ptr(cs),
                     ptr(q_drop[0]) if fuse_drop else None, q_drop[1] if fuse_drop else 0, q_drop[2] if fuse_drop else 0.0, dt)
            x, kin = hs, Hh
        if two and self.training and self.seq2vec_dropout > 0.0 and seeds is not None:
            Hh = Hq // 2
            for li in range(2):                                  # F.dropout(vec_li, 0.3), seq2vec.py:73-74
                call('vqa_dropout', ws['q32'].data_ptr() + 4 * li * Hh, Hq, ws['q32'].data_ptr() + 4 * li * Hh, Hq, B, Hh,
                     seeds[li], self.seq2vec_dropout, _lib.F32)
                call('vqa_dropout', ws['qT'].data_ptr() + esz * li * Hh, Hq, ws['qT'].data_ptr() + esz * li * Hh, Hq, B, Hh,
                     seeds[li], self.seq2vec_dropout, dt)
        return ws['q32']

    def _visual_early(self, ws, feats, B):
        """Question-independent part of the visual forward, launched right after the question encoder; the default is none."""
        ws['visual_early_done'] = False

    def _enc_workspace(self, B, L, device):
        """Question-encoder buffers alone (encode_question called outside a forward)."""
        key = ('enc', B, L, str(device))
        ws = self._ws.get(key)
        if ws is None:
            ws = {}
            self._encoder_buffers(ws, lambda shape, dt: torch.zeros(shape, dtype=dt, device=device), B, L)
            self._ws[key] = ws
        return ws

    def _encoder_buffers(self, ws, z, B, L):
        tdt, f32 = self.compute_dtype, torch.float32
        Ep, H = _pad8(self.E), self.Hq
        Hh = max(rnn.hidden_size for rnn in self.seq2vec.layers)
        ws.update({'len': z((B,), torch.int32), 'perm': z((B,), torch.int32), 'm_t': z((L,), torch.int32),
                   'offs': z((L + 1,), torch.int32), 'x_emb': z((L * B, Ep), tdt), 'xg': z((L * B, 4 * Hh), tdt),
                   'c': z((B, Hh), f32), 'h': z((B, Hh), tdt), 'q32': z((B, H), f32), 'qT': z((B, H), tdt)})
        if len(self.seq2vec.layers) > 1:
            ws['hs'] = z((L * B, Hh), tdt)
        if self.train_encoder:                        # state kept for backward through time
            ws['ptok'] = z((L * B,), torch.int32)
            for li, rnn in enumerate(self.seq2vec.layers):
                h = rnn.hidden_size
                ws[f'gates{li}'] = z((L * B, 4 * h), tdt)
                ws[f'cs{li}'] = z((L * B, h), f32)
                ws[f'hs{li}'] = z((L * B, h), tdt)
            ws.update({'dG': z((L * B, 4 * Hh), tdt), 'dgs': z((B, 4 * Hh), tdt), 'hprev': z((L * B, Hh), tdt),
                       'dc': z((B, Hh), f32), 'dq32': z((B, H), f32), 'dh_ext': z((L * B, Hh), f32)})

    def _forward_impl(self, feats, tokens, captions):
        B, L = tokens.shape
        dev = tokens.device
        Tc = captions.shape[1] if captions is not None else 0
        _lib.bind_stream()
        self._materialize(dev)
        ws = self._workspace(B, self._geometry(feats), L, Tc, dev)
        H, F, Hf, Ec, C, Cp, dt = self.Hq, self.F, self.Hf, self.Ec, self.num_classes, _pad8(self.num_classes), self.dt
        Dv = self.Dv
        seeds, p = self._drop_seeds()

        # parameter-independent prologue first: a gradient bucket still being all-reduced from the
        # previous train_step (and its Adam update) is only needed from here on
        if self.train_encoder:
            self._finish_pending()               # the deferred bucket holds the encoder's own weights
            self._frozen_key = None              # they change every step: refresh the operand copies
        self.encode_question(tokens, ws, seeds[4:6], q_drop=(ws['q_d'], seeds[1], p))
        self._visual_prologue(ws, feats, B)
        self._finish_pending()
        self._refresh_shadow()
        self._visual_early(ws, feats, B)
        if self.seq2vec.arch == '2-lstm':       # (the single-layer encoder's cell kernel wrote q_d already)
            call('vqa_dropout', ptr(ws['qT']), H, ptr(ws['q_d']), H, B, H, seeds[1], p, dt)
        self._launch_early(ws, B)                                  # question-side products of the head, one launch
        self._visual_forward(ws, feats, B, seeds[0], p)           # -> ws['pooled_d'] = dropout(visual vector)
        # xv = tanh(linear_v(v)), xq = tanh(linear_q(q)) and the fusion x = dropout(tanh(xv * xq)): one launch reduces
        # both products' partial sums (linear_q's wait in ws['early'] or, failing that, share the slab workspace)
        skv = call('vqa_gemm_splitk_hint', B, F, Dv, 1, 1)
        if 'linear_q' in ws['early']:
            qslab, skq, stq = ws['early']['linear_q']
            nfl = skv * B * F
        else:
            skq = call('vqa_gemm_splitk_hint', B, F, H, 1, 1)
            nfl = (skv + skq) * B * F
        buf = self._slabs(nfl, ws['pooled_d'].device)
        call('vqa_gemm_slabs_rows', ptr(ws['pooled_d']), Dv, 1, ptr(self._w('linear_v.weight')), Dv, 1, ptr(buf), B * F, F,
             B, F, Dv, skv, self.dt, self.gemm_engine, None, None)
        if 'linear_q' not in ws['early']:
            qslab, stq = buf.data_ptr() + 4 * skv * B * F, B * F
            call('vqa_gemm_slabs_rows', ptr(ws['q_d']), H, 1, ptr(self._w('linear_q.weight')), H, 1, qslab, stq, F,
                 B, F, H, skq, self.dt, self.gemm_engine, None, None)
        call('vqa_fusion_fwd_slabs', ptr(buf), skv, B * F, ptr(self._f32('linear_v.bias')), qslab, skq, stq,
             ptr(self._f32('linear_q.bias')), ptr(ws['xv']), ptr(ws['xq']), ptr(ws['x']), F + Ec, B, F, seeds[2], p, dt)
        if Ec:
            call('vqa_caption_pool', ptr(captions), ptr(self._cap_t), ptr(ws['x']), B, Tc, Ec, Ec,
                 self._cap_t.shape[0], F + Ec, F, dt)
        self._linear(ws['x'], F + Ec, 'linear_classif1.weight', 'linear_classif1.bias', B, Hf, F + Ec, act=1,
                     out_pre=ws['h1t'], ld_pre=Hf, out_post=ws['h1d'], ld_post=Hf, seed=seeds[3], p=p)
        self._linear(ws['h1d'], Hf, 'linear_classif2.weight', 'linear_classif2.bias', B, C, Hf,
                     out_pre=ws['logits'], ld_pre=Cp)
        self._ctx = (ws, feats, B, L, seeds, p)
        self._generation += 1
        return ws['logits'][:, :C]

    # ------------------------------------------------------------------ backward
    def _backward_from_dense(self, dlogits: torch.Tensor):
        """Backward from an arbitrary d loss / d logits (B, C) tensor (autograd path)."""
        ws, feats, B, L, seeds, p = self._ctx
        C, Cp = self.num_classes, _pad8(self.num_classes)
        d = dlogits.to(torch.float32).contiguous()
        _lib.bind_stream()
        call('vqa_cast_pad', ptr(d), C, ptr(ws['dlogits']), Cp, B, C, self.dt)
        self._backward_impl()

    def _backward_impl(self):
        """Gradients of every trainable parameter from ws['dlogits'] into the flat gradient buffer."""
        ws, feats, B, L, seeds, p = self._ctx
        H, Dv, F, Hf, Ec, C, Cp, dt = self.Hq, self.Dv, self.F, self.Hf, self.Ec, self.num_classes, _pad8(self.num_classes), self.dt
        _lib.bind_stream()
        g = self._g
        # classifier
        slab, sk, st = self._gemm_slabs(ws['dlogits'], Cp, 1, self._w('linear_classif2.weight'), Hf, 0, B, Hf, C)
        call('vqa_tanh_dropout_bwd', ptr(slab), sk, st, ptr(ws['h1t']), Hf, ptr(ws['dz1']), Hf, B, Hf, seeds[3], p, dt)
        # fusion
        slab, sk, st = self._gemm_slabs(ws['dz1'], Hf, 1, self._w('linear_classif1.weight'), F + Ec, 0, B, F + Ec, Hf)
        call('vqa_fusion_bwd', ptr(slab), sk, st, F + Ec, ptr(ws['xv']), ptr(ws['xq']), ptr(ws['dzv']), ptr(ws['dzq']),
             F, B, F, seeds[2], p, dt)
        # the four bias gradients of the head (and the mean of the per-row losses) in one launch
        extra = [(ws['loss_rows'], 1, B, 1, _lib.F32, ws['loss'], None, 1.0 / B)] if self._loss_pending else []
        self._loss_pending = False
        head_bias = extra + [(ws['dlogits'], Cp, B, C, dt, g('linear_classif2.bias'), None),
                             (ws['dz1'], Hf, B, Hf, dt, g('linear_classif1.bias'), None),
                             (ws['dzv'], F, B, F, dt, g('linear_v.bias'), None),
                             (ws['dzq'], F, B, F, dt, g('linear_q.bias'), None)]
        head_items = [(ws['dlogits'], Cp, ws['h1d'], Hf, 'linear_classif2.weight', C, Hf, B),
                      (ws['dz1'], Hf, ws['x'], F + Ec, 'linear_classif1.weight', Hf, F + Ec, B),
                      (ws['dzv'], F, ws['pooled_d'], Dv, 'linear_v.weight', F, Dv, B),
                      (ws['dzq'], F, ws['q_d'], H, 'linear_q.weight', F, H, B)]
        if self._takes_head_group(B):
            # nothing downstream reads these gradients: the front end's backward launches them together with its own
            # small ones (one dependent launch less on the way to its first big kernel)
            self._held_group = (head_items, head_bias)
        else:
            self._held_group = None
            self._wgrad_grouped(head_items, colsums=head_bias)     # the head's four weight and bias gradients: one launch
            self._flush_wgrads()                  # (those with split K: partial sums reduced in one launch)
            self._bucket_done('linear_q.bias')
        slab, sk, st = self._gemm_slabs(ws['dzv'], F, 1, self._w('linear_v.weight'), Dv, 0, B, Dv, F)
        if self._dpool_from_slabs and not self.train_encoder and ws.get('attn_nsplit', 1) == 1:
            # the attention backward kernel reduces these partial sums (and applies dropoutV's mask) in its prologue;
            # nothing else touches the slab workspace before it runs
            ws['dpool_src'] = (ptr(slab), sk, st, seeds[0], p)
        else:
            ws['dpool_src'] = None
            call('vqa_epilogue', ptr(slab), sk, st, B, Dv, None, None, None, 0, 0, None, 0, ptr(ws['d_pooled']), Dv,
                 _lib.F32, seeds[0], p)
        if self.train_encoder:                  # d loss / d q through linear_q and its dropout (model.py:67-68)
            slab, sk, st = self._gemm_slabs(ws['dzq'], F, 1, self._w('linear_q.weight'), H, 0, B, H, F)
            call('vqa_epilogue', ptr(slab), sk, st, B, H, None, None, None, 0, 0, None, 0, ptr(ws['dq32']), H,
                 _lib.F32, seeds[1], p)
        self._visual_backward(ws, feats, B)
        if self.train_encoder:
            self._encoder_backward(ws, B, L, seeds)

    def _takes_head_group(self, B) -> bool:
        """True when _visual_backward() launches the head's held weight / bias gradients (self._held_group) with its
        own first group of small gradients and then signals the head's bucket."""
        return False

    def _add_dq(self, ws, B, slab, sk, st):
        """dq += sum of the slabs (a front-end's contribution to d loss / d q)."""
        H = self.Hq
        call('vqa_epilogue', ptr(slab), sk, st, B, H, None, None, ptr(ws['dq32']), H, 0, ptr(ws['dq32']), H,
             None, 0, _lib.F32, 0, 0.0)

    def _encoder_backward(self, ws, B, L, seeds):
        """Backward through time of the packed LSTM layer(s) from ws['dq32'], weight / bias / embedding
        gradients into the flat gradient buffer (autograd of seq2vec.py:42-43, 63-75 restated)."""
        Hq, dt, Ep = self.Hq, self.dt, _pad8(self.E)
        two = self.seq2vec.arch == '2-lstm'
        total = ws['offs'].data_ptr() + 4 * L
        if two and self.training and self.seq2vec_dropout > 0.0:
            Hh = Hq // 2
            for li in range(2):
                a = ws['dq32'].data_ptr() + 4 * li * Hh
                call('vqa_dropout', a, Hq, a, Hq, B, Hh, seeds[4 + li], self.seq2vec_dropout, _lib.F32)
        names = self._enc_names()
        dh_ext = None
        for li in reversed(range(len(self._enc_layers))):
            wih, whh, _ = self._enc_layers[li]
            Hh, kin = whh.shape[1], wih.shape[1]
            pre = f'seq2vec.{names[li]}.'
            rnn = self.seq2vec.layers[li]
            x_in = ws['x_emb'] if li == 0 else ws[f'hs{li - 1}']
            dq_l = ws['dq32'].data_ptr() + 4 * li * Hh
            for t in reversed(range(L)):
                if t == L - 1:
                    slab, sk, st = None, 0, 0
                else:                             # dh_t from the step after: dG_{t+1} W_hh
                    slab, sk, st = self._gemm_slabs(ws['dgs'], 4 * Hh, 1, whh, Hh, 0, B, Hh, 4 * Hh,
                                                    active_rows=ws['m_t'].data_ptr() + 4 * (t + 1))
                call('vqa_lstm_cell_bwd_packed', ptr(slab), sk, st, dq_l, Hq, ptr(dh_ext), ptr(ws[f'gates{li}']),
                     ptr(ws[f'cs{li}']), ptr(ws[f'hs{li}']), ptr(ws['dc']), ptr(ws['dG']), ptr(ws['dgs']), ptr(ws['hprev']),
                     ptr(ws['len']), ptr(ws['perm']), ptr(ws['m_t']), ptr(ws['offs']), B, Hh, t, L, dt)
            g = self._g
            call('vqa_colsum_rows', ptr(ws['dG']), 4 * Hh, ptr(g(pre + 'bias_ih_l0')), total, 4 * Hh, dt)
            call('vqa_colsum_rows', ptr(ws['dG']), 4 * Hh, ptr(g(pre + 'bias_hh_l0')), total, 4 * Hh, dt)
            self._wgrad(ws['dG'], 4 * Hh, ws['hprev'], Hh, pre + 'weight_hh_l0', 4 * Hh, Hh, L * B, active_k=total)
            self._wgrad(ws['dG'], 4 * Hh, x_in, kin, pre + 'weight_ih_l0', 4 * Hh, rnn.input_size, L * B, active_k=total)
            # gradient at the layer's input sequence: dX = dG W_ih
            slab, sk, st = self._gemm_slabs(ws['dG'], 4 * Hh, 1, wih, kin, 0, L * B, kin, 4 * Hh, active_rows=total)
            if li > 0:
                call('vqa_epilogue', ptr(slab), sk, st, L * B, kin, None, None, None, 0, 0, ptr(ws['dh_ext']), kin,
                     None, 0, _lib.F32, 0, 0.0)
                dh_ext = ws['dh_ext']
            else:
                emb = self.seq2vec.embedding.weight
                call('vqa_embedding_grad', ptr(slab), sk, st, kin, ptr(ws['ptok']), total,
                     ptr(ws['x_emb']) if two else None, Ep, ptr(g('seq2vec.embedding.weight')), emb.shape[0], self.E, dt)
        self._bucket_done('seq2vec.embedding.weight')

    # ------------------------------------------------------------------ loss / prediction / step
    def loss_and_grad_logits(self, answers: torch.Tensor, reduce: bool = True) -> torch.Tensor:
        """Fused softmax cross-entropy (mean) of the last forward + its gradient into the
        backward workspace.  Returns the 1-element device tensor holding the loss."""
        ws, feats, B, L, seeds, p = self._ctx
        C, Cp = self.num_classes, _pad8(self.num_classes)
        ans = answers.long().contiguous()
        _lib.bind_stream()
        call('vqa_softmax_ce', ptr(ws['logits']), Cp, ptr(ans), ptr(ws['loss_rows']), ptr(ws['dlogits']), Cp, B, C,
             1.0 / B, self.dt, ptr(self._status_words(ans.device)))
        if reduce:                      # (train_step folds this mean into the backward pass's batched column sums)
            call('vqa_sum_scale', ptr(ws['loss_rows']), ptr(ws['loss']), B, 1.0 / B)
        self._loss_pending = not reduce
        return ws['loss']

    def predict(self, logits: Optional[torch.Tensor] = None, answers=None, type_ids=None, counters=None, n_types: int = 4):
        """First-max argmax of the last forward's logits (train.py:106); optionally
        accumulates per-question-type count / right counters on the device."""
        ws, feats, B, L, seeds, p = self._ctx
        C, Cp = self.num_classes, _pad8(self.num_classes)
        _lib.bind_stream()
        call('vqa_argmax_accuracy', ptr(ws['logits']), Cp, B, C, ptr(answers), ptr(type_ids), n_types, ptr(ws['pred']),
             ptr(counters))
        return ws['pred']

    def _adam_range(self, lo: int, hi: int, lr, betas, eps, step: int, grad_scale: float):
        if self._exp_avg is None:
            self._exp_avg = torch.zeros_like(self._flat)
            self._exp_avg_sq = torch.zeros_like(self._flat)
        if hi <= lo:
            return
        _lib.bind_stream()
        sh = None if self._shadow is None else self._shadow.data_ptr() + 2 * lo
        call('vqa_adam_step', self._flat.data_ptr() + 4 * lo, self._flat_grad.data_ptr() + 4 * lo,
             self._exp_avg.data_ptr() + 4 * lo, self._exp_avg_sq.data_ptr() + 4 * lo, sh, hi - lo, lr, betas[0], betas[1],
             eps, step, grad_scale, ptr(self._status_words(self._flat.device)))

    def adam_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        """One fused Adam update of every trainable parameter from the flat gradient buffer."""
        self._finish_pending()
        self.step_count += 1
        self._adam_range(0, self._flat.numel(), lr, betas, eps, self.step_count, grad_scale)
        self._mark_shadow_fresh()

    def _finish_pending(self):
        """Complete a train_step whose last gradient bucket was left in flight: wait for its
        all-reduce and apply its Adam update."""
        pend = self._pending
        if pend is None:
            return
        self._pending = None
        reducer, lo, hi, lr, betas, eps, step, scale = pend
        reducer.finish_last()
        self._adam_range(lo, hi, lr, betas, eps, step, scale)

    def sync(self, check: bool = True):
        """Make every parameter current (applies a deferred last-bucket update, if any) and raise for inputs the
        kernels flagged since the last check.  check=False only enqueues the deferred update (no read-back of the
        status words, which waits for the stream): call check_status() later."""
        self._finish_pending()
        if check:
            self.check_status()

    def _status_words(self, device) -> torch.Tensor:
        st = getattr(self, '_status', None)
        if st is None or st.device != device:
            self._status = st = torch.zeros(_lib.STATUS_WORDS, dtype=torch.int32, device=device)
        return st

    def check_status(self) -> None:
        """Read the device-side status words (one small copy, which synchronises with the stream) and raise what the
        reference's stock modules would have raised at the offending call: IndexError for a question token outside
        the embedding table (nn.Embedding, seq2vec.py:42) or an answer id outside [0, C) (CrossEntropyLoss,
        train.py:201).  The kernels do not stop: a bad token reads the pad row; a bad target makes its row's loss NaN
        and, as the reference raises before optimizer.step() (train.py:201-204), the fused Adam kernel leaves weights
        and moments untouched from that step on until this check has raised (the optimiser step counter is rewound to
        the last update that was applied).  ignore_index is not implemented: -100 is out of range like any negative
        id.  Called by sync(), state_dict() and the train / evaluation drivers."""
        st = getattr(self, '_status', None)
        if st is None:
            return
        tok, tgt, skipped, resident_timeout = st.tolist()
        if resident_timeout:
            st.zero_()
            if self._resident_sync is not None:
                self._resident_sync.zero_()
            raise _lib.VqaHipError('vqa_lstm_resident: its grid barrier timed out (the workgroups were not all resident: is '
                              'another process using the GPU?); the question vectors of that step are not valid -- '
                              'construct the module with resident_encoder=False to use the per-step kernels')
        if tok or tgt:
            st.zero_()
            if skipped:
                self.step_count = skipped - 1
            what = []
            if tok:
                what.append(f'a question token id outside [0, {self.seq2vec.embedding.num_embeddings})')
            if tgt:
                what.append(f'an answer id outside [0, {self.num_classes}) (Target out of bounds)')
            raise IndexError(f'{type(self).__name__}: ' + ' and '.join(what) + ' reached the kernels since the last check')

    def train_step(self, input_v, input_q, answers, captions=None, lr: float = 1e-5, reducer=None) -> torch.Tensor:
        """forward + loss + backward (+ gradient all-reduce through `reducer`) + Adam
        (train.py:200-204 in one call).  Returns the loss as a 1-element device tensor.
        With a reducer whose `defer_last` is set, the all-reduce and Adam update of the last
        gradient bucket (of every bucket with `defer_all`) are completed inside the next call (after its
        question-encoder / inv-norm prologue, which they do not affect) or by sync(); results are
        identical either way."""
        feats, tokens, captions = self._prepare_inputs(input_v, input_q, captions)
        self._materialize(tokens.device)
        betas, eps = (0.9, 0.999), 1e-8
        with torch.no_grad():
            self._bptt = True
            self._forward_impl(feats, tokens, captions)
            loss = self.loss_and_grad_logits(answers, reduce=False)
            self.collective_cus = getattr(reducer, 'collective_cus', 0) if (reducer is not None and reducer.enabled) else 0
            if reducer is not None:
                reducer.begin(self)
                self.grad_ready_hook = reducer.bucket_ready
            try:
                self._backward_impl()
            finally:
                self.grad_ready_hook = None
            if reducer is not None and reducer.enabled and getattr(reducer, 'defer_all', False):
                scale = reducer.defer_everything()       # nothing is waited for inside the step
                self.step_count += 1
                self._pending = (reducer, 0, self._flat.numel(), lr, betas, eps, self.step_count, scale)
                self._mark_shadow_fresh()
            elif reducer is not None and reducer.enabled and reducer.defer_last and len(self.buckets) > 1:
                scale = reducer.finish(defer_last=True)
                lo, hi = self.buckets[-1]
                self.step_count += 1
                self._adam_range(0, lo, lr, betas, eps, self.step_count, scale)
                self._pending = (reducer, lo, hi, lr, betas, eps, self.step_count, scale)
                self._mark_shadow_fresh()
            else:
                scale = reducer.finish() if reducer is not None else 1.0
                self.adam_step(lr, betas, eps, grad_scale=scale)
        return loss

    def eval_step(self, input_v, input_q, answers, captions=None) -> torch.Tensor:
        """forward + mean cross-entropy with no gradient bookkeeping (the validation / test passes of train.py:96-97,
        test.py:78-79).  Returns the loss as a 1-element device tensor; the logits stay in the workspace for
        predict()."""
        feats, tokens, captions = self._prepare_inputs(input_v, input_q, captions)
        with torch.no_grad():
            self._bptt = False
            self._forward_impl(feats, tokens, captions)
            return self.loss_and_grad_logits(answers)

    def optimizer_state_dict(self) -> Dict[str, object]:
        """Adam moments per parameter name plus the step / dropout counters (the reference
        checkpoints the weights only, train.py:215-228; this makes a run resumable)."""
        self._finish_pending()
        out = {'step_count': self.step_count, 'drop_step': self.drop_step, 'base_seed': self.base_seed,
               'exp_avg': {}, 'exp_avg_sq': {}}
        if self._flat is not None and self._exp_avg is not None:
            for name, (o, n, shape) in self._index.items():
                out['exp_avg'][name] = self._exp_avg[o:o + n].view(shape).clone()
                out['exp_avg_sq'][name] = self._exp_avg_sq[o:o + n].view(shape).clone()
        return out

    def load_optimizer_state_dict(self, state: Dict[str, object]) -> None:
        self._finish_pending()
        dev = next(self.parameters()).device
        self._materialize(dev)
        self.step_count, self.drop_step = int(state['step_count']), int(state['drop_step'])
        self.base_seed = int(state.get('base_seed', self.base_seed))      # the dropout stream continues
        self._exp_avg = torch.zeros_like(self._flat)
        self._exp_avg_sq = torch.zeros_like(self._flat)
        for name, (o, n, shape) in self._index.items():
            if name in state['exp_avg']:
                self._exp_avg[o:o + n].copy_(state['exp_avg'][name].reshape(-1).to(dev))
                self._exp_avg_sq[o:o + n].copy_(state['exp_avg_sq'][name].reshape(-1).to(dev))

    # ------------------------------------------------------------------ torch plumbing
    def state_dict(self, *a, **kw):
        self._finish_pending()
        self.check_status()
        return super().state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self._finish_pending()
        out = super()._apply(fn, *a, **kw)
        self._flat = None               # parameters moved: rebuild the flat views lazily
        self._frozen_key = None
        self._ws.clear()
        self._slab = None
        self._wslab, self._wslab_off, self._wgrad_pending = None, 0, []
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._finish_pending()
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._shadow_fresh = False
        self._frozen_key = None
        return out


class VQAAttnModel(_VQAHeadModel):
    """Question-conditioned attention over precomputed region features (B, R, Dv) (or the
    (B, Dv, H, W) map model.py:159 produces): K hops of tanh-add attention + softmax over the grid +
    weighted pooling replace the reference's unweighted mean (model.py:161-164)."""

    def _build_visual(self, gen):
        self.attention = nn.ModuleList([_Hop(self.Dv, self.Hq, self.A, k + 1 < self.hops) for k in range(self.hops)])

    def _visual_trainable_order(self):
        out = []
        for k in reversed(range(self.hops)):
            hop = self.attention[k]
            pre = f'attention.{k}.'
            out += [(pre + 'linear_a.weight', hop.linear_a.weight), (pre + 'linear_a.bias', hop.linear_a.bias),
                    (pre + 'linear_qa.weight', hop.linear_qa.weight), (pre + 'linear_qa.bias', hop.linear_qa.bias),
                    (pre + 'linear_va.bias', hop.linear_va.bias)]
            if k > 0:
                prev = self.attention[k - 1]
                ppre = f'attention.{k - 1}.'
                out += [(ppre + 'linear_u.weight', prev.linear_u.weight), (ppre + 'linear_u.bias', prev.linear_u.bias)]
            out += [(pre + 'linear_va.weight', hop.linear_va.weight)]
        return out

    def _visual_bucket_starts(self):
        out = []
        for k in range(self.hops):
            out += [f'attention.{k}.linear_a.weight', f'attention.{k}.linear_va.weight']
        return out

    # the head's weight / bias gradients ride in the launch of the last hop's small gradients (bf16; the grouped
    # launch takes contractions up to 512 rows and 8 column sums)
    merge_head_group = True

    def _takes_head_group(self, B) -> bool:
        return self.merge_head_group and self.dt == _lib.BF16 and B <= 512

    def _q_side_products(self, ws):
        # the first hop's query projection reads the question vector itself
        return super()._q_side_products(ws) + [('qa0', ws['qT'], self.Hq, 'attention.0.linear_qa.weight', self.A, self.Hq)]

    def _prepare_visual(self, input_v):
        if input_v.dim() == 4:                       # (B, Dv, H, W) as model.py:159 produces
            B, Dv, Hh, Ww = input_v.shape
            input_v = input_v.permute(0, 2, 3, 1).reshape(B, Hh * Ww, Dv)
        if input_v.dim() != 3 or input_v.shape[2] != self.Dv:
            raise ValueError(f'image_feats must be (B, R, {self.Dv}) or (B, {self.Dv}, H, W)')
        feats = input_v.to(self.compute_dtype).contiguous()
        return feats, feats.shape[0]

    def _geometry(self, feats):
        return feats.shape[1]

    def _ldA(self) -> int:
        """Row pitch (elements) of the projected features P and of dP: rows start on 128-byte lines, so that the
        projection's output stores and the attention kernels' row reads touch whole cache lines."""
        return (self.A + 63) // 64 * 64 if self.pad_projection_rows else self.A

    def _visual_workspace(self, ws, z, B, R):
        tdt, f32 = self.compute_dtype, torch.float32
        H, A, Dv = self.Hq, self.A, self.Dv
        ldA = self._ldA()
        # small batches: a sample's region rows are shared by several workgroups of the attention kernels
        ns = call('vqa_attn_splits', B, R, A, Dv, self.dt) if self.split_small_batches else 1
        ws['attn_nsplit'] = ns
        if ns > 1:
            ws['attn_ws'] = z((call('vqa_attn_split_workspace_floats', B, R, Dv, ns),), f32)
            ws['dqa_slabs'] = z((ns, B, A), f32)
        ws.update({'inv_norm': z((B * R,), f32), 'dP': z((B * R, ldA), tdt),
                   'dqa32': z((B, A), f32), 'dqaT': z((B, A), tdt), 'dwa_part': z((B * ns, A), f32), 'dc_part': z((B * ns,), f32),
                   'du32': z((B, H), f32), 'duT': z((B, H), tdt)})
        for k in range(self.hops):
            ws[f'P{k}'] = z((B * R, ldA), tdt)
            ws[f'qa{k}'] = z((B, A), f32)
            ws[f'p{k}'] = z((B, R), f32)
            ws[f'pooled{k}'] = z((B, Dv), tdt)
            if k > 0:
                ws[f'u32_{k}'] = z((B, H), f32)
                ws[f'uT_{k}'] = z((B, H), tdt)

    _dpool_from_slabs = True
    # the forward kernel can reduce the query projection's partial sums in its prologue (vqa_attn_fwd_qslabs), but the
    # launch that saves costs the kernel about as much as it gains (measured: 57 -> 60.7 us against a 4.8 us launch)
    attn_qa_from_slabs = False
    swap_big_wgrad = True    # dW_va is computed as its transpose (the faster operand orientation) and transposed by the reduce
    pad_projection_rows = True
    fuse_row_norm = True     # the first hop's projection computes the region norms itself (one pass over the features less)
    split_small_batches = True   # fewer samples than CUs: several workgroups per sample in the attention kernels

    def _visual_prologue(self, ws, feats, B):
        R, Dv = feats.shape[1], self.Dv
        if not self.fuse_row_norm:
            call('vqa_inv_norm', ptr(feats), ptr(ws['inv_norm']), B * R, Dv, Dv, self.dt)

    def _project(self, ws, feats, B, k):
        """P^k = normalize(V) W_va^k^T + b (B R, A); hop 0 with fuse_row_norm also writes inv_norm."""
        R = feats.shape[1]
        A, Dv, dt = self.A, self.Dv, self.dt
        ldA = self._ldA()
        pre = f'attention.{k}.'
        ev = self._event_pair(f'proj:{k}')
        if k == 0 and self.fuse_row_norm:       # inv_norm written by the same launch
            call('vqa_gemm_fused_rownorm', ptr(feats), Dv, ptr(self._w(pre + 'linear_va.weight')), Dv, ptr(ws[f'P{k}']), ldA,
                 B * R, A, Dv, ptr(self._f32(pre + 'linear_va.bias')), ptr(ws['inv_norm']), 0, dt, dt, self.gemm_engine)
        else:
            call('vqa_gemm_fused', ptr(feats), Dv, 1, ptr(self._w(pre + 'linear_va.weight')), Dv, 1, ptr(ws[f'P{k}']), ldA,
                 B * R, A, Dv, ptr(self._f32(pre + 'linear_va.bias')), ptr(ws['inv_norm']), 0, dt, dt, self.gemm_engine)
        if ev is not None:
            ev[1].record()

    def _visual_early(self, ws, feats, B):
        # every hop's region projection reads the features and its own weights only: none waits for the question
        for k in range(self.hops):
            self._project(ws, feats, B, k)
        ws['visual_early_done'] = True

    def _visual_forward(self, ws, feats, B, seed, p):
        R = feats.shape[1]
        H, A, Dv, dt = self.Hq, self.A, self.Dv, self.dt
        ldA = self._ldA()
        u32, uT = ws['q32'], ws['qT']
        for k in range(self.hops):
            pre = f'attention.{k}.'
            if not ws.get('visual_early_done'):
                self._project(ws, feats, B, k)
            early = k == 0 and 'qa0' in ws['early']
            if early and not self.attn_qa_from_slabs:
                slab, sk, st = ws['early']['qa0']
                call('vqa_epilogue', slab, sk, st, B, A, ptr(self._f32(pre + 'linear_qa.bias')), None, None, 0, 0,
                     ptr(ws['qa0']), A, None, 0, dt, 0, 0.0)
                early = False
            elif not early:
                self._linear(uT, H, pre + 'linear_qa.weight', pre + 'linear_qa.bias', B, A, H, out_pre=ws[f'qa{k}'], ld_pre=A)
            ev = self._event_pair('attn_fwd')
            ns = ws['attn_nsplit']
            if ns > 1:
                if early:
                    slab, sk, st = ws['early']['qa0']
                    call('vqa_epilogue', slab, sk, st, B, A, ptr(self._f32(pre + 'linear_qa.bias')), None, None, 0, 0,
                         ptr(ws['qa0']), A, None, 0, dt, 0, 0.0)
                call('vqa_attn_fwd_split', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, ptr(ws[f'qa{k}']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(self._f32(pre + 'linear_a.bias')), ptr(ws['inv_norm']),
                     ptr(ws[f'p{k}']), ptr(ws[f'pooled{k}']), None, Dv, B, R, A, Dv, dt,
                     ptr(ws['pooled_d']) if k + 1 == self.hops else None, seed, p, ns, ptr(ws['attn_ws']))
            elif early:             # the kernel's prologue reduces the query projection's partial sums (+ bias) into qa0
                slab, sk, st = ws['early']['qa0']
                call('vqa_attn_fwd_qslabs', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, slab, sk, st,
                     ptr(self._f32(pre + 'linear_qa.bias')), ptr(ws['qa0']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(self._f32(pre + 'linear_a.bias')), ptr(ws['inv_norm']),
                     ptr(ws[f'p{k}']), ptr(ws[f'pooled{k}']), None, Dv, B, R, A, Dv, dt,
                     ptr(ws['pooled_d']) if k + 1 == self.hops else None, seed, p)
            else:
                call('vqa_attn_fwd', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, ptr(ws[f'qa{k}']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(self._f32(pre + 'linear_a.bias')), ptr(ws['inv_norm']),
                     ptr(ws[f'p{k}']), ptr(ws[f'pooled{k}']), None, Dv, B, R, A, Dv, dt,
                     ptr(ws['pooled_d']) if k + 1 == self.hops else None, seed, p)   # dropoutV fused into the last hop
            if ev is not None:
                ev[1].record()
            if k + 1 < self.hops:
                self._linear(ws[f'pooled{k}'], Dv, pre + 'linear_u.weight', pre + 'linear_u.bias', B, H, Dv,
                             addend=u32, ld_add=H, out_pre=ws[f'u32_{k + 1}'], ld_pre=H, out_post=ws[f'uT_{k + 1}'], ld_post=H)
                u32, uT = ws[f'u32_{k + 1}'], ws[f'uT_{k + 1}']

    def _visual_backward(self, ws, feats, B):
        R = feats.shape[1]
        H, A, Dv, dt = self.Hq, self.A, self.Dv, self.dt
        ldA = self._ldA()
        g = self._g
        # attention hops, last to first
        du_carry = None
        for k in reversed(range(self.hops)):
            pre = f'attention.{k}.'
            uT = ws['qT'] if k == 0 else ws[f'uT_{k}']
            src = ws.get('dpool_src') if k + 1 == self.hops else None
            ns = ws['attn_nsplit']
            if ns > 1:              # row ranges: per-range sums of dqa come back as slabs, reduced (and cast) right after
                call('vqa_attn_bwd_split', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, ptr(ws[f'qa{k}']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(ws['inv_norm']), ptr(ws[f'p{k}']), ptr(ws['d_pooled']), Dv,
                     ptr(ws['dP']), ldA, ptr(ws['dqa_slabs']), A, ptr(ws['dwa_part']), ptr(ws['dc_part']),
                     B, R, A, Dv, dt, ns, ptr(ws['attn_ws']))
                call('vqa_epilogue', ptr(ws['dqa_slabs']), ns, B * A, B, A, None, None, None, 0, 0, ptr(ws['dqa32']), A,
                     ptr(ws['dqaT']), A, dt, 0, 0.0)
            elif src is not None:
                slab, sk, st, dseed, dp = src
                call('vqa_attn_bwd_dslabs', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, ptr(ws[f'qa{k}']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(ws['inv_norm']), ptr(ws[f'p{k}']), slab, sk, st, dseed, dp,
                     ptr(ws['d_pooled']), Dv, ptr(ws['dP']), ldA, ptr(ws['dqa32']), ptr(ws['dqaT']), A, ptr(ws['dwa_part']),
                     ptr(ws['dc_part']), B, R, A, Dv, dt)
            else:
                call('vqa_attn_bwd', ptr(ws[f'P{k}']), ldA, ptr(feats), Dv, ptr(ws[f'qa{k}']), A,
                     ptr(self._f32(pre + 'linear_a.weight')), ptr(ws['inv_norm']), ptr(ws[f'p{k}']), ptr(ws['d_pooled']), Dv,
                     ptr(ws['dP']), ldA, ptr(ws['dqa32']), ptr(ws['dqaT']), A, ptr(ws['dwa_part']), ptr(ws['dc_part']),
                     B, R, A, Dv, dt)
            # the hop's small gradients in one launch: w_a, its offset c, and the two biases that share d(P + qa);
            # the last hop's launch (the first one here) also carries the head's held weight and bias gradients
            items = [(ws['dqaT'], A, uT, H, pre + 'linear_qa.weight', A, H, B)]
            colsums = [(ws['dwa_part'], A, B * ns, A, _lib.F32, g(pre + 'linear_a.weight'), None),
                       (ws['dc_part'], 1, B * ns, 1, _lib.F32, g(pre + 'linear_a.bias'), None),
                       (ws['dqa32'], A, B, A, _lib.F32, g(pre + 'linear_qa.bias'), g(pre + 'linear_va.bias'))]
            held, self._held_group = self._held_group, None
            if held is not None:
                items, colsums = held[0] + items, held[1] + colsums
            self._wgrad_grouped(items, colsums=colsums)
            self._flush_wgrads()
            if held is not None:
                self._bucket_done('linear_q.bias')
            if k > 0:
                ppre = f'attention.{k - 1}.'
                slab, sk, st = self._gemm_slabs(ws['dqaT'], A, 1, self._w(pre + 'linear_qa.weight'), H, 0, B, H, A)
                call('vqa_epilogue', ptr(slab), sk, st, B, H, None, None, ptr(du_carry), H, 0, ptr(ws['du32']), H,
                     ptr(ws['duT']), H, dt, 0, 0.0)
                du_carry = ws['du32']
                self._wgrad(ws['duT'], H, ws[f'pooled{k - 1}'], Dv, ppre + 'linear_u.weight', H, Dv, B)
                call('vqa_colsum', ptr(ws['du32']), H, ptr(g(ppre + 'linear_u.bias')), B, H, _lib.F32)
            elif self.train_encoder:              # d q = dqa W_qa (+ the carry of the later hops: u^1 = q + ...)
                slab, sk, st = self._gemm_slabs(ws['dqaT'], A, 1, self._w(pre + 'linear_qa.weight'), H, 0, B, H, A)
                self._add_dq(ws, B, slab, sk, st)
                if du_carry is not None:
                    self._add_dq(ws, B, du_carry, 1, 0)
            self._bucket_done(pre + 'linear_va.bias')
            self._wgrad(ws['dP'], ldA, feats, Dv, pre + 'linear_va.weight', A, Dv, B * R, swapped=self.swap_big_wgrad)
            self._bucket_done(pre + 'linear_va.weight')
            if k > 0:
                slab, sk, st = self._gemm_slabs(ws['duT'], H, 1, self._w(ppre + 'linear_u.weight'), Dv, 0, B, Dv, H)
                call('vqa_epilogue', ptr(slab), sk, st, B, Dv, None, None, None, 0, 0, ptr(ws['d_pooled']), Dv,
                     None, 0, _lib.F32, 0, 0.0)


class VQAGAPModel(_VQAHeadModel):
    """The reference's VQAGAPModel (model.py:81-184) on precomputed ResNet stage maps: per stage
    the mean over the grid times a question gate ``gapK = Sigmoid(Linear(QUESTION_OUT, C_k))``,
    concatenated into the VISUAL_OUT_GAP = 3840-wide vector entering ``linear_v``.  The frozen
    ResNet-152 itself (model.py:98-118, 156-159) is outside this module: ``input_v`` is the list of
    its four stage outputs, each (B, C_k, H_k, W_k) in NCHW or channels_last memory format."""

    def __init__(self, vocab_questions, vocab_answers, args=None, input_size: int = 512, *,
                 gap_channels: Sequence[int] = (256, 512, 1024, 2048), **kw):
        self.gap_channels = tuple(int(c) for c in gap_channels)
        for c in self.gap_channels:
            if c % 8:
                raise ValueError('stage channel counts must be multiples of 8')
        kw['visual_out'] = sum(self.gap_channels)          # VISUAL_OUT_GAP (model.py:144)
        super().__init__(vocab_questions, vocab_answers, args, input_size, **kw)

    def _build_visual(self, gen):
        for k, c in enumerate(self.gap_channels, 1):        # names as model.py:125-140
            setattr(self, f'gap{k}', nn.Sequential(nn.Linear(self.Hq, c), nn.Sigmoid()))

    def _gates(self):
        return [getattr(self, f'gap{k}')[0] for k in range(1, len(self.gap_channels) + 1)]

    def _visual_trainable_order(self):
        # the gate weights (and the biases) sit back to back in the flat buffer, so that the
        # gates of all stages are one (sum C_k) x Hq product
        gs = self._gates()
        return ([(f'gap{k}.0.bias', g.bias) for k, g in enumerate(gs, 1)] +
                [(f'gap{k}.0.weight', g.weight) for k, g in enumerate(gs, 1)])

    def _visual_bucket_starts(self):
        return ['gap1.0.bias']

    def _prepare_visual(self, input_v):
        if not isinstance(input_v, (list, tuple)) or len(input_v) != len(self.gap_channels):
            raise ValueError(f'input_v must be the list of {len(self.gap_channels)} stage maps (B, C_k, H_k, W_k)')
        maps = []
        for m, c in zip(input_v, self.gap_channels):
            if m.dim() != 4 or m.shape[1] != c or m.shape[0] != input_v[0].shape[0]:
                raise ValueError(f'stage map must be (B, {c}, H, W), got {tuple(m.shape)}')
            m = m.to(self.compute_dtype)
            if m.is_contiguous():
                layout = 0
            elif m.is_contiguous(memory_format=torch.channels_last):
                layout = 1
            else:
                m, layout = m.contiguous(), 0
            maps.append((m, layout))
        return tuple(maps), input_v[0].shape[0]

    def _geometry(self, maps):
        return tuple((m.shape[2] * m.shape[3], layout) for m, layout in maps)

    def _visual_workspace(self, ws, z, B, geom):
        f32 = torch.float32
        Dv = self.Dv
        ws.update({'gmean': z((B, Dv), f32), 'gate': z((B, Dv), f32), 'dzg': z((B, Dv), self.compute_dtype)})
        nsp = [call('vqa_grid_mean_splits', layout, B, c, hw) for (hw, layout), c in zip(geom, self.gap_channels)]
        ws['gm_splits'] = nsp
        need = max([n * B * c for n, c in zip(nsp, self.gap_channels) if n > 1], default=0)
        ws['gm_part'] = z((need,), f32) if need else None

    def _visual_prologue(self, ws, maps, B):
        col = 0
        for (m, layout), c, nsp in zip(maps, self.gap_channels, ws['gm_splits']):
            call('vqa_grid_mean', ptr(m), layout, B, c, m.shape[2] * m.shape[3], ptr(ws['gmean']), self.Dv, col,
                 ptr(ws['gm_part']), nsp, self.dt)
            col += c

    def _visual_forward(self, ws, maps, B, seed, p):
        H, Dv = self.Hq, self.Dv
        slab, sk, stride = self._gemm_slabs(ws['qT'], H, 1, self._w('gap1.0.weight'), H, 1, B, Dv, H)
        call('vqa_gate_fwd', ptr(slab), sk, stride, ptr(self._f32('gap1.0.bias')), ptr(ws['gmean']), Dv,
             ptr(ws['gate']), Dv, ptr(ws['pooled_d']), Dv, B, Dv, seed, p, self.dt)

    def _visual_backward(self, ws, maps, B):
        H, Dv = self.Hq, self.Dv
        call('vqa_gate_bwd', ptr(ws['d_pooled']), Dv, ptr(ws['gmean']), Dv, ptr(ws['gate']), Dv, ptr(ws['dzg']), Dv,
             B, Dv, self.dt)
        call('vqa_colsum', ptr(ws['dzg']), Dv, ptr(self._g('gap1.0.bias')), B, Dv, self.dt)
        self._wgrad(ws['dzg'], Dv, ws['qT'], H, 'gap1.0.weight', Dv, H, B)
        self._bucket_done('gap1.0.weight')
        if self.train_encoder:                    # the gates read the question vector (model.py:166-169)
            slab, sk, st = self._gemm_slabs(ws['dzg'], Dv, 1, self._w('gap1.0.weight'), H, 0, B, H, Dv)
            self._add_dq(ws, B, slab, sk, st)

    def _w(self, name):
        if name == 'gap1.0.weight':           # all gate weights as one (Dv, Hq) matrix
            o = self._index[name][0]
            src = self._shadow if self.dt == _lib.BF16 else self._flat
            return src[o:o + self.Dv * self.Hq]
        return super()._w(name)

    def _f32(self, name):
        if name == 'gap1.0.bias':
            o = self._index[name][0]
            return self._flat[o:o + self.Dv]
        return super()._f32(name)

    def _g(self, name):
        if name in ('gap1.0.bias', 'gap1.0.weight'):
            o = self._index[name][0]
            return self._flat_grad[o:o + (self.Dv if name.endswith('bias') else self.Dv * self.Hq)]
        return super()._g(name)


class VQAModel(_VQAHeadModel):
    """The reference's VQAModel (model.py:29-79) on the precomputed last ResNet map: a trainable
    1x1 convolution to VISUAL_OUT / (H W) channels (model.py:53-54) whose NCHW output is viewed as
    (B, VISUAL_OUT) (model.py:61).  ``input_v`` is the (B, 2048, H, W) map (or (B, H W, 2048)
    channels-last rows); H W must divide VISUAL_OUT, as the reference's .view requires."""

    def __init__(self, vocab_questions, vocab_answers, args=None, input_size: int = 512, *,
                 map_channels: int = 2048, **kw):
        self.Cin = int(map_channels)
        self.R = int((input_size / 32) ** 2)               # output_size, model.py:53
        kw.setdefault('visual_out', VISUAL_OUT)
        if self.R < 1 or kw['visual_out'] % self.R:
            raise ValueError('(input_size / 32)^2 must divide visual_out')
        self.c_out = kw['visual_out'] // self.R
        super().__init__(vocab_questions, vocab_answers, args, input_size, **kw)

    def _build_visual(self, gen):
        # index 0 stands where the reference keeps the frozen ResNet; index 1 is its Conv2d
        self.visual = nn.Sequential(nn.Identity(), nn.Conv2d(self.Cin, self.c_out, 1))
        conv = self.visual[1]
        with torch.no_grad():
            b = 1.0 / math.sqrt(self.Cin)
            conv.weight.copy_((torch.rand(conv.weight.shape, generator=gen) * 2 - 1) * b)
            conv.bias.copy_((torch.rand(conv.bias.shape, generator=gen) * 2 - 1) * b)

    def _visual_trainable_order(self):
        return [('visual.1.bias', self.visual[1].bias), ('visual.1.weight', self.visual[1].weight)]

    def _visual_bucket_starts(self):
        return ['visual.1.bias']

    def _prepare_visual(self, input_v):
        if input_v.dim() == 4:
            B, Cin, Hh, Ww = input_v.shape
            input_v = input_v.permute(0, 2, 3, 1).reshape(B, Hh * Ww, Cin)
        if input_v.dim() != 3 or input_v.shape[2] != self.Cin or input_v.shape[1] != self.R:
            raise ValueError(f'input_v must be (B, {self.Cin}, H, W) with H W = {self.R}, or (B, {self.R}, {self.Cin})')
        feats = input_v.to(self.compute_dtype).contiguous()
        return feats, feats.shape[0]

    def _geometry(self, feats):
        return feats.shape[1]

    def _visual_workspace(self, ws, z, B, R):
        cp = _pad8(self.c_out)
        ws.update({'y': z((B * R, cp), self.compute_dtype), 'dy': z((B * R, cp), self.compute_dtype)})

    def _visual_prologue(self, ws, feats, B):
        pass

    def _visual_forward(self, ws, feats, B, seed, p):
        R, Cin, c, cp, dt = self.R, self.Cin, self.c_out, _pad8(self.c_out), self.dt
        call('vqa_gemm_fused', ptr(feats), Cin, 1, ptr(self._w('visual.1.weight')), Cin, 1, ptr(ws['y']), cp,
             B * R, c, Cin, ptr(self._f32('visual.1.bias')), None, 0, dt, dt, self.gemm_engine)
        call('vqa_flatten_cm', ptr(ws['y']), cp, ptr(ws['pooled_d']), self.Dv, B, R, c, seed, p, dt)

    def _visual_backward(self, ws, feats, B):
        R, Cin, c, cp, dt = self.R, self.Cin, self.c_out, _pad8(self.c_out), self.dt
        call('vqa_unflatten_cm', ptr(ws['d_pooled']), self.Dv, ptr(ws['dy']), cp, B, R, c, dt)
        call('vqa_colsum', ptr(ws['dy']), cp, ptr(self._g('visual.1.bias')), B * R, c, dt)
        self._wgrad(ws['dy'], cp, feats, Cin, 'visual.1.weight', c, Cin, B * R)
        self._bucket_done('visual.1.weight')


class VQAGAPStackModel(_VQAHeadModel):
    """Stage-mean pooling with the max / mean stack of the reference's BERT-side models
    (model.py:1187-1207, 1225-1241) in front of the fusion head: each stage mean goes through
    ``dim_chg_gapK = Sigmoid(Linear(C_k, D))``, the four D-wide vectors are weighted by the scalars
    ``w1..w4``, stacked (with the reference's cross-wiring: slots hold stages 1, 3, 4, 2), and the
    element-wise max and mean over the stack are concatenated into the 2 D-wide visual vector.
    (Upstream that vector is a VisualBERT visual embedding; BERT is outside this package, so here it
    enters ``linear_v`` of the head.)  ``input_v`` as for VQAGAPModel."""

    WIRING = (1, 3, 4, 2)

    def __init__(self, vocab_questions, vocab_answers, args=None, input_size: int = 512, *,
                 gap_channels: Sequence[int] = (256, 512, 1024, 2048), stack_dim: int = 2048, **kw):
        self.gap_channels = tuple(int(c) for c in gap_channels)
        self.D = int(stack_dim)
        if len(self.gap_channels) != 4 or any(c % 8 for c in self.gap_channels) or self.D % 8:
            raise ValueError('four stage channel counts and stack_dim, all multiples of 8')
        kw['visual_out'] = 2 * self.D
        super().__init__(vocab_questions, vocab_answers, args, input_size, **kw)

    def _build_visual(self, gen):
        for k, c in enumerate(self.gap_channels, 1):        # names as model.py:1187-1202
            setattr(self, f'dim_chg_gap{k}', nn.Sequential(nn.Linear(c, self.D), nn.Sigmoid()))
        for i in range(1, 5):                               # model.py:1204-1207
            setattr(self, f'w{i}', nn.Parameter(torch.tensor(0.25)))

    def _visual_trainable_order(self):
        out = [(f'w{i}', getattr(self, f'w{i}')) for i in range(1, 5)]     # adjacent: 8 floats apart in the flat buffer
        for k in range(1, 5):
            lin = getattr(self, f'dim_chg_gap{k}')[0]
            out += [(f'dim_chg_gap{k}.0.bias', lin.bias), (f'dim_chg_gap{k}.0.weight', lin.weight)]
        return out

    def _visual_bucket_starts(self):
        return ['w1']

    _prepare_visual = VQAGAPModel._prepare_visual
    _geometry = VQAGAPModel._geometry

    def _visual_workspace(self, ws, z, B, geom):
        f32, tdt = torch.float32, self.compute_dtype
        Ct, D = sum(self.gap_channels), self.D
        ws.update({'gmean': z((B, Ct), f32), 'gmeanT': z((B, Ct), tdt), 'sig': z((B, 4, D), f32),
                   'arg': z((B, D), torch.uint8), 'dzs': z((B, 4, D), tdt), 'dw_part': z((B, 4), f32)})
        nsp = [call('vqa_grid_mean_splits', layout, B, c, hw) for (hw, layout), c in zip(geom, self.gap_channels)]
        ws['gm_splits'] = nsp
        need = max([n * B * c for n, c in zip(nsp, self.gap_channels) if n > 1], default=0)
        ws['gm_part'] = z((need,), f32) if need else None

    def _visual_prologue(self, ws, maps, B):
        Ct = sum(self.gap_channels)
        col = 0
        for (m, layout), c, nsp in zip(maps, self.gap_channels, ws['gm_splits']):
            call('vqa_grid_mean', ptr(m), layout, B, c, m.shape[2] * m.shape[3], ptr(ws['gmean']), Ct, col,
                 ptr(ws['gm_part']), nsp, self.dt)
            col += c
        call('vqa_cast_pad', ptr(ws['gmean']), Ct, ptr(ws['gmeanT']), Ct, B, Ct, self.dt)

    def _cols(self):
        out, col = {}, 0
        for k, c in enumerate(self.gap_channels, 1):
            out[k] = (col, c)
            col += c
        return out

    def _visual_forward(self, ws, maps, B, seed, p):
        Ct, D = sum(self.gap_channels), self.D
        cols = self._cols()
        for i, k in enumerate(self.WIRING):
            col, c = cols[k]
            slab, sk, st = self._gemm_slabs(ws['gmeanT'][:, col:], Ct, 1, self._w(f'dim_chg_gap{k}.0.weight'), c, 1, B, D, c)
            call('vqa_epilogue', ptr(slab), sk, st, B, D, ptr(self._f32(f'dim_chg_gap{k}.0.bias')), None, None, 0, 2,
                 ptr(ws['sig'][:, i]), 4 * D, None, 0, _lib.F32, 0, 0.0)
        call('vqa_stack_maxavg_fwd', ptr(ws['sig']), ptr(self._f32('w1')), 8, ptr(ws['pooled_d']), 2 * D, ptr(ws['arg']),
             B, D, 4, seed, p, self.dt)

    def _visual_backward(self, ws, maps, B):
        Ct, D = sum(self.gap_channels), self.D
        cols = self._cols()
        call('vqa_stack_maxavg_bwd', ptr(ws['d_pooled']), 2 * D, ptr(ws['sig']), ptr(self._f32('w1')), 8, ptr(ws['arg']),
             ptr(ws['dzs']), ptr(ws['dw_part']), B, D, 4, self.dt)
        for i, k in enumerate(self.WIRING):
            col, c = cols[k]
            call('vqa_colsum', ptr(ws['dw_part'][:, i]), 4, ptr(self._g(f'w{i + 1}')), B, 1, _lib.F32)
            call('vqa_colsum', ptr(ws['dzs'][:, i]), 4 * D, ptr(self._g(f'dim_chg_gap{k}.0.bias')), B, D, self.dt)
            self._wgrad(ws['dzs'][:, i], 4 * D, ws['gmeanT'][:, col:], Ct, f'dim_chg_gap{k}.0.weight', D, c, B)
        self._bucket_done('dim_chg_gap4.0.weight')
