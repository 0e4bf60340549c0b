"""CPU oracle for the VQA attention / fusion / classifier train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vqa_amd/`` may import this module; it
is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / the timed CPU baseline,
never as the thing shipped.

It is a plain restatement, in explicit PyTorch tensor equations on the CPU, of
the arithmetic the reference performs on this path.  Each function cites the
reference lines it follows (paths relative to the reference checkout):

* question encoder   vqa_model/models/seq2vec.py:11-45
* region pooling     vqa_model/models/model.py:161-164   (uniform special case)
* gated GAP model    vqa_model/models/model.py:125-140, 149-184
* conv-flatten model vqa_model/models/model.py:53-54, 60-79
* fusion head        vqa_model/models/model.py:60-79
* caption stream     vqa_model/models/model.py:1571-1574, 1678, 1734-1741
* loss / optimiser   vqa_model/train.py:34-35, 201-204
* argmax / accuracy  vqa_model/train.py:105-110, 147-158

Pinning: the reference holds no golden vectors or tests for this path.  The
integer parts (vocabulary encoding, lengths, last-step selection, embedding
gather, LSTM via the reference's own ``seq2vec.LSTM`` sub-modules) are pinned by
fixtures generated from the reference's importable modules
(``oracle/make_golden.py`` -> ``tests/golden``).  The question-conditioned
spatial attention does not exist in the reference (it pools with an unweighted
mean); for it the oracle is anchored only through the ``w_a = 0 => mean over
regions`` identity, and the skip-thoughts encoder the reference actually calls
is an un-vendored third-party module: parity for those two is unpinned.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

# ----------------------------------------------------------------------------
# dropout mask: a counter-based integer hash so that the GPU kernels and this
# oracle draw the SAME mask from (seed, element index).  The reference uses
# torch.nn.Dropout(0.5) (model.py:38-40, 1590-1592) whose Philox stream cannot
# be reproduced elsewhere; only the distribution (Bernoulli keep, 1/(1-p)
# scale) is the contract.
# ----------------------------------------------------------------------------

def dropout_keep_mask(seed: int, n: int, p: float) -> np.ndarray:
    """keep[i] for i in [0, n): murmur3 finaliser of (i * golden + seed)."""
    idx = np.arange(n, dtype=np.uint64)
    x = (idx * np.uint64(0x9E3779B1) + np.uint64(seed & 0xFFFFFFFF)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    # p is taken as a float32 (the C ABI passes a float) and widened, as the kernels do
    thr = np.uint64(min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF))
    return x >= thr


def site_seed(base_seed: int, step: int, site: int) -> int:
    """Seed of one dropout site at one step (same mixing as the host code)."""
    x = (base_seed * 0x01000193 + step * 0x9E3779B1 + site * 0x7F4A7C15 + 0x165667B1) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x & 0xFFFFFFFF


def apply_dropout(x: torch.Tensor, seed: Optional[int], p: float) -> torch.Tensor:
    if seed is None or p <= 0.0:
        return x
    keep = torch.from_numpy(dropout_keep_mask(seed, x.numel(), p)).reshape(x.shape)
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


# ----------------------------------------------------------------------------
# question encoder
# ----------------------------------------------------------------------------

def process_lengths(tokens: torch.Tensor) -> torch.Tensor:
    """len_b = L - #(tok == 0)            (seq2vec.py:11-14)."""
    return tokens.size(1) - tokens.eq(0).sum(1)


def select_last(x: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """x[b, len_b - 1, :]; len_b == 0 picks the last step, as the Python
    negative index ``mask[i][lengths[i]-1]`` does (seq2vec.py:16-25)."""
    B, L, _ = x.shape
    idx = (lengths.long() - 1) % L
    return x[torch.arange(B), idx]


def embedding_gather(tokens: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """nn.Embedding lookup with padding_idx=0 (seq2vec.py:35-37, 42): row 0 is the pad row and
    receives no gradient."""
    return torch.nn.functional.embedding(tokens.long(), table, padding_idx=0)


def lstm_time_major(e: torch.Tensor, w_ih, w_hh, b_ih, b_hh) -> torch.Tensor:
    """Single-layer LSTM run along the token axis of e (B, L, E) -> (B, L, H).

    Equations of torch.nn.LSTM (the op seq2vec.py:38 constructs), gate order
    i, f, g, o, zero initial state.  The reference feeds (B, L, E) to a
    batch_first=False LSTM and so recurs along the batch axis (seq2vec.py:42-43);
    the build recurs along tokens, which is the reference's own modules applied
    to ``x.transpose(0, 1)``.
    """
    B, L, _ = e.shape
    H = w_hh.shape[1]
    h = e.new_zeros(B, H)
    c = e.new_zeros(B, H)
    outs = []
    for t in range(L):
        g = e[:, t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, gg, o = g.split(H, dim=1)
        i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
        c = f * c + i * gg
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1)


def encode_question(tokens, p: Dict[str, torch.Tensor], drop_seeds=(None, None), drop_p: float = 0.3) -> torch.Tensor:
    """seq2vec.LSTM (seq2vec.py:40-45) or, when the parameters hold ``rnn_0`` / ``rnn_1``, TwoLSTM
    (seq2vec.py:61-76): tanh on the embeddings, two stacked LSTMs, select_last of each layer's
    outputs, dropout(0.3) on each vector in training mode, concatenation."""
    if 'seq2vec.rnn_0.weight_ih_l0' in p:
        e = torch.tanh(embedding_gather(tokens, p['seq2vec.embedding.weight']))
        x = e
        vecs = []
        lengths = process_lengths(tokens)
        for li in range(2):
            pre = f'seq2vec.rnn_{li}.'
            x = lstm_time_major(x, p[pre + 'weight_ih_l0'], p[pre + 'weight_hh_l0'], p[pre + 'bias_ih_l0'], p[pre + 'bias_hh_l0'])
            vecs.append(apply_dropout(select_last(x, lengths), drop_seeds[li], drop_p))
        return torch.cat(vecs, dim=1)
    e = embedding_gather(tokens, p['seq2vec.embedding.weight'])
    hs = lstm_time_major(e, p['seq2vec.rnn.weight_ih_l0'], p['seq2vec.rnn.weight_hh_l0'],
                         p['seq2vec.rnn.bias_ih_l0'], p['seq2vec.rnn.bias_hh_l0'])
    return select_last(hs, process_lengths(tokens))


# ----------------------------------------------------------------------------
# image side
# ----------------------------------------------------------------------------

def l2_normalise(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(p=2) semantics: v / max(||v||, eps)   (model.py:1678)."""
    n = v.pow(2).sum(-1, keepdim=True).sqrt().clamp_min(eps)
    return v / n


def attention_hop(v_hat, u, w_va, b_va, w_qa, b_qa, w_a, c_a):
    """One question-conditioned attention hop over the region grid.

    P = v_hat W_va^T + b_va; qa = u W_qa^T + b_qa; s = w_a . tanh(P + qa) + c;
    p = softmax_r(s); pooled = sum_r p_r v_hat_r.  With w_a = 0 this is
    torch.mean(x, axis=(2, 3)) of model.py:161-164 on normalised features.
    """
    P = v_hat @ w_va.t() + b_va                       # (B, R, A)
    qa = u @ w_qa.t() + b_qa                          # (B, A)
    s = torch.tanh(P + qa[:, None, :]) @ w_a.reshape(-1) + c_a.reshape(())   # (B, R)
    p = torch.softmax(s, dim=1)
    pooled = (p[:, :, None] * v_hat).sum(1)           # (B, Dv)
    return pooled, p


def caption_vector(cap_tokens, table) -> torch.Tensor:
    """Masked mean over caption tokens then L2 normalise
    (mean_pooling model.py:1571-1574; F.normalize model.py:1678)."""
    e = table[cap_tokens.long()]                      # (B, Tc, Ec)
    m = cap_tokens.ne(0).to(e.dtype)[:, :, None]
    pooled = (e * m).sum(1) / m.sum(1).clamp_min(1e-9)
    return l2_normalise(pooled)


def gated_gap(maps: Sequence[torch.Tensor], q: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """VQAGAPModel's visual vector: for each ResNet stage k = 1..4 the mean of the (B, C_k, H_k, W_k)
    map over its grid (model.py:161-164) times the question gate sigmoid(gap_k(q)) (model.py:125-140,
    166-169, on the un-dropped question vector), concatenated (model.py:171)."""
    parts = []
    for k, m in enumerate(maps, 1):
        avg = m.mean(dim=(2, 3))
        gate = torch.sigmoid(q @ p[f'gap{k}.0.weight'].t() + p[f'gap{k}.0.bias'])
        parts.append(avg * gate)
    return torch.cat(parts, dim=1)


STACK_WIRING = (1, 3, 4, 2)     # stack slot i holds dim_chg_gap{k}(mean of stage k), k = STACK_WIRING[i] (model.py:1234-1237)


def stack_max_avg(maps: Sequence[torch.Tensor], p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Max / mean over the weighted stack of the four gated stage vectors (model.py:1225-1241):
    s_k = sigmoid(dim_chg_gap_k(mean_k)); stacked = [w1 s_1, w2 s_3, w3 s_4, w4 s_2]; cat(max, mean)."""
    st = []
    for i, k in enumerate(STACK_WIRING):
        avg = maps[k - 1].mean(dim=(2, 3))
        s = torch.sigmoid(avg @ p[f'dim_chg_gap{k}.0.weight'].t() + p[f'dim_chg_gap{k}.0.bias'])
        st.append(p[f'w{i + 1}'] * s)
    stacked = torch.stack(st, 2)
    return torch.cat((stacked.max(dim=2).values, stacked.mean(dim=2)), dim=1)


def conv_flatten(fmap: torch.Tensor, p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """VQAModel's visual vector: Conv2d(Dv, Dv / (H W), 1) over the last ResNet map (model.py:53-54)
    then .view(-1, VISUAL_OUT) of the NCHW result (model.py:61): channel-major flatten."""
    w = p['visual.1.weight']
    y = torch.einsum('bkhw,ck->bchw', fmap, w.reshape(w.shape[0], w.shape[1])) + p['visual.1.bias'][None, :, None, None]
    return y.reshape(fmap.shape[0], -1)


def forward(p: Dict[str, torch.Tensor], feats, tokens: torch.Tensor,
            captions: Optional[torch.Tensor] = None, hops: int = 1,
            drop_seeds: Optional[Sequence[int]] = None, drop_p: float = 0.5,
            return_aux: bool = False, mode: str = 'attn'):
    """mode 'attn': feats (B, R, Dv) region features; mode 'gap': feats = the four (B, C_k, H_k, W_k)
    stage maps of VQAGAPModel; mode 'conv': feats = the (B, Dv, H, W) map of VQAModel.
    tokens (B, L) int64 -> logits (B, C).

    Head verbatim from model.py:62-77: x_v = tanh(linear_v(drop(pooled)));
    x_q = tanh(linear_q(drop(q))); x = drop(tanh(x_v * x_q));
    x = drop(tanh(linear_classif1(x))); logits = linear_classif2(x).
    drop_seeds = (seed_v, seed_q, seed_f1, seed_f2) or None for eval mode.
    """
    ds = tuple(drop_seeds) if drop_seeds is not None else (None, None, None, None)
    ds = ds + (None,) * (6 - len(ds))                 # sites 4, 5: the two TwoLSTM vectors (seq2vec.py:73-74)
    q = encode_question(tokens, p, ds[4:6])
    u = q
    att = []
    pooled = None
    v_hat = None
    if mode == 'gap':
        pooled = gated_gap(feats, q, p)
        hops = 0
    elif mode == 'conv':
        pooled = conv_flatten(feats, p)
        hops = 0
    elif mode == 'stack':
        pooled = stack_max_avg(feats, p)
        hops = 0
    else:
        v_hat = l2_normalise(feats)
    for k in range(hops):
        pre = f'attention.{k}.'
        pooled, pk = attention_hop(v_hat, u, p[pre + 'linear_va.weight'], p[pre + 'linear_va.bias'],
                                   p[pre + 'linear_qa.weight'], p[pre + 'linear_qa.bias'],
                                   p[pre + 'linear_a.weight'], p[pre + 'linear_a.bias'])
        att.append(pk)
        if k + 1 < hops:
            u = u + pooled @ p[pre + 'linear_u.weight'].t() + p[pre + 'linear_u.bias']
    x_v = torch.tanh(apply_dropout(pooled, ds[0], drop_p) @ p['linear_v.weight'].t() + p['linear_v.bias'])
    x_q = torch.tanh(apply_dropout(q, ds[1], drop_p) @ p['linear_q.weight'].t() + p['linear_q.bias'])
    x = apply_dropout(torch.tanh(x_v * x_q), ds[2], drop_p)
    if captions is not None:
        x = torch.cat((x, caption_vector(captions, p['caption_embedding.weight'])), dim=1)
    x = torch.tanh(x @ p['linear_classif1.weight'].t() + p['linear_classif1.bias'])
    x = apply_dropout(x, ds[3], drop_p)
    logits = x @ p['linear_classif2.weight'].t() + p['linear_classif2.bias']
    if return_aux:
        return logits, {'q': q, 'v_hat': v_hat, 'att': att, 'pooled': pooled}
    return logits


# ----------------------------------------------------------------------------
# loss, optimiser, metrics
# ----------------------------------------------------------------------------

def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """CrossEntropyLoss(), mean reduction (train.py:35, 201)."""
    z = logits - logits.max(dim=1, keepdim=True).values
    lse = z.exp().sum(1).log()
    return (lse - z[torch.arange(z.shape[0]), target.long()]).mean()


def cross_entropy_grad(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """d loss / d logits = (softmax - onehot) / B."""
    g = torch.softmax(logits, dim=1)
    g[torch.arange(g.shape[0]), target.long()] -= 1.0
    return g / logits.shape[0]


def adam_step(param, grad, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults, no weight decay (train.py:34, 204).  step >= 1."""
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


def argmax_first(logits: np.ndarray) -> np.ndarray:
    """np.argmax(axis=1): first maximum, a NaN counts as the maximum (train.py:106)."""
    return np.argmax(logits, axis=1)


def accuracy_counts(pred: np.ndarray, answer: np.ndarray, type_ids: np.ndarray, n_types: int = 4):
    """count[t] += 1; right[t] += (answer == pred)      (train.py:107-110)."""
    count = np.zeros(n_types, dtype=np.int64)
    right = np.zeros(n_types, dtype=np.int64)
    for j in range(answer.shape[0]):
        count[type_ids[j]] += 1
        if answer[j] == pred[j]:
            right[type_ids[j]] += 1
    return count, right


def oa_aa(count: np.ndarray, right: np.ndarray):
    """OA = sum right / sum count; AA = sum_t acc_t / 4 with the constant 4 and
    only types with count > 0 contributing (train.py:147-158).  The reference
    raises IndexError when a type is empty (it indexes a list that was not
    appended to); the oracle returns the sum over the non-empty types / 4."""
    acc = [right[t] * 1.0 / count[t] for t in range(len(count)) if count[t] > 0]
    oa = right.sum() * 1.0 / count.sum()
    aa = sum(acc) * 1.0 / 4
    return oa, aa


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------

def init_params(vq: int, C: int, E: int, Hq: int, Dv: int, A: int, F: int, Hf: int,
                hops: int = 1, vc: int = 0, Ec: int = 0, seed: int = 0,
                dtype=torch.float32, mode: str = 'attn', gap_channels: Sequence[int] = (),
                conv_out: int = 0, two_lstm: bool = False) -> Dict[str, torch.Tensor]:
    """Random parameters with the reference's shapes and state_dict key names
    (model.py:44, 55-58; seq2vec.py:35-38).  nn.Linear / nn.LSTM / nn.Embedding
    default initialisers."""
    g = torch.Generator().manual_seed(seed)

    def uni(shape, bound):
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)

    def linear(pre, out_f, in_f, p):
        b = 1.0 / math.sqrt(in_f)
        p[pre + '.weight'] = uni((out_f, in_f), b)
        p[pre + '.bias'] = uni((out_f,), b)

    p: Dict[str, torch.Tensor] = {}
    emb = torch.randn((vq + 1, E), generator=g, dtype=torch.float64).to(dtype)
    emb[0] = 0
    p['seq2vec.embedding.weight'] = emb
    if two_lstm:                    # seq2vec.py:58-59: hidden = Hq / 2 per layer, outputs concatenated
        Hh = Hq // 2
        k = 1.0 / math.sqrt(Hh)
        for li, nin in enumerate((E, Hh)):
            p[f'seq2vec.rnn_{li}.weight_ih_l0'] = uni((4 * Hh, nin), k)
            p[f'seq2vec.rnn_{li}.weight_hh_l0'] = uni((4 * Hh, Hh), k)
            p[f'seq2vec.rnn_{li}.bias_ih_l0'] = uni((4 * Hh,), k)
            p[f'seq2vec.rnn_{li}.bias_hh_l0'] = uni((4 * Hh,), k)
    else:
        k = 1.0 / math.sqrt(Hq)
        p['seq2vec.rnn.weight_ih_l0'] = uni((4 * Hq, E), k)
        p['seq2vec.rnn.weight_hh_l0'] = uni((4 * Hq, Hq), k)
        p['seq2vec.rnn.bias_ih_l0'] = uni((4 * Hq,), k)
        p['seq2vec.rnn.bias_hh_l0'] = uni((4 * Hq,), k)
    if mode == 'gap':               # model.py:125-140; linear_v takes the concatenation (model.py:144)
        for k, ck in enumerate(gap_channels, 1):
            linear(f'gap{k}.0', ck, Hq, p)
        Dv = sum(gap_channels)
    elif mode == 'stack':           # model.py:1187-1207; Dv here = the common gate width (2048 upstream)
        for k, ck in enumerate(gap_channels, 1):
            linear(f'dim_chg_gap{k}.0', Dv, ck, p)
        for i in range(4):
            p[f'w{i + 1}'] = torch.tensor(0.25, dtype=dtype) + uni((), 0.1)      # 0.25 upstream (model.py:1204-1207), spread for the test
        Dv = 2 * Dv
    elif mode == 'conv':            # model.py:54: Conv2d(Dv, conv_out, 1); default init bound 1/sqrt(fan_in)
        b = 1.0 / math.sqrt(Dv)
        p['visual.1.weight'] = uni((conv_out, Dv, 1, 1), b)
        p['visual.1.bias'] = uni((conv_out,), b)
    for h in range(hops if mode == 'attn' else 0):
        pre = f'attention.{h}.'
        linear(pre + 'linear_va', A, Dv, p)
        linear(pre + 'linear_qa', A, Hq, p)
        linear(pre + 'linear_a', 1, A, p)
        if h + 1 < hops:
            linear(pre + 'linear_u', Hq, Dv, p)
    linear('linear_v', F, Dv, p)
    linear('linear_q', F, Hq, p)
    linear('linear_classif1', Hf, F + Ec, p)
    linear('linear_classif2', C, Hf, p)
    if vc > 0:
        cemb = torch.randn((vc + 1, Ec), generator=g, dtype=torch.float64).to(dtype)
        cemb[0] = 0
        p['caption_embedding.weight'] = cemb
    return p


FROZEN_PREFIXES = ('seq2vec.', 'caption_embedding.')


def trainable_names(p: Dict[str, torch.Tensor], train_encoder: bool = False) -> List[str]:
    """The question encoder is frozen (model.py:42-43) unless train_encoder; so is the caption
    encoder (it runs under no_grad, model.py:1675-1676)."""
    frozen = ('caption_embedding.',) if train_encoder else FROZEN_PREFIXES
    return [k for k in p if not k.startswith(frozen)]


def loss_and_grads(p, feats, tokens, answers, captions=None, hops=1, drop_seeds=None, drop_p=0.5, mode='attn',
                   train_encoder=False):
    """One fwd + loss + bwd of the step (train.py:200-203) through autograd."""
    names = trainable_names(p, train_encoder)
    q = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in p.items()}
    logits = forward(q, feats, tokens, captions, hops, drop_seeds, drop_p, mode=mode)
    loss = cross_entropy(logits, answers)
    grads = torch.autograd.grad(loss, [q[k] for k in names])
    return logits.detach(), loss.detach(), dict(zip(names, grads))


def synthetic_batch(B: int, R: int, Dv: int, L: int, vq: int, C: int, seed: int,
                    dtype=torch.float32, Tc: int = 0, vc: int = 0):
    """VQA-v2-shaped synthetic batch: feats relu(N(0,1)); question length
    U{3..L} incl. the trailing <EOS>=0 (VocabEncoder.py:113-119), ids U{1..vq};
    answers U{0..C-1}; 4 question types uniform."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn((B, R, Dv), generator=g, dtype=torch.float32).clamp_min_(0).to(dtype)
    lens = torch.randint(3, L + 1, (B,), generator=g)
    tok = torch.randint(1, vq + 1, (B, L), generator=g)
    pos = torch.arange(L)[None, :]
    tok = torch.where(pos < (lens[:, None] - 1), tok, torch.zeros_like(tok))
    ans = torch.randint(0, C, (B,), generator=g)
    types = torch.randint(0, 4, (B,), generator=g)
    out = {'feats': feats, 'tokens': tok, 'answers': ans, 'types': types}
    if Tc > 0:
        clen = torch.randint(3, Tc + 1, (B,), generator=g)
        cap = torch.randint(1, vc + 1, (B, Tc), generator=g)
        cpos = torch.arange(Tc)[None, :]
        out['captions'] = torch.where(cpos < (clen[:, None] - 1), cap, torch.zeros_like(cap))
    return out


def synthetic_maps(B: int, channels: Sequence[int], grids: Sequence[int], seed: int, dtype=torch.float32):
    """Post-ReLU-like NCHW stage maps (B, C_k, g_k, g_k), as ResNet stages feed model.py:161-164."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn((B, c, s, s), generator=g, dtype=torch.float32).clamp_min_(0).to(dtype)
            for c, s in zip(channels, grids)]
