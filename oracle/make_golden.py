"""Generate the committed golden fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs the read-only
reference checkout at /root/reference); the fixtures travel, the reference does
not.  Two kinds of vectors are written:

* G1/G2 come from the REFERENCE's own importable code:
  ``VocabEncoder.VocabEncoder`` (vqa_model/VocabEncoder.py) and the
  ``embedding`` / ``rnn`` sub-modules, ``process_lengths`` and ``select_last``
  of ``models.seq2vec`` (vqa_model/models/seq2vec.py:11-45).  ``seq2vec`` does
  ``import skipthoughts`` at module top (seq2vec.py:8); that third-party module
  is not vendored, is only touched by ``factory(arch='skipthoughts')`` and is
  not used here, so an empty module object is registered under that name for
  the import to succeed.
* G3..G6 come from the oracle restatement (oracle/vqa_ref.py) at tiny sizes and
  pin the oracle itself against silent edits.

    python oracle/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference/vqa_model'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'models'))

from oracle import vqa_ref  # noqa: E402


def make_vocab_inputs():
    rng = np.random.RandomState(42)
    q_words = ['is', 'there', 'a', 'the', 'road', 'building', 'water', 'area', 'number', 'of',
               'more', 'than', 'left', 'right', 'top', 'bottom', 'Urban', 'rural', 'what', 'how',
               'many', 'are', 'in', 'image', 'Large', 'small', '12', 'at', 'next', 'to']
    questions = []
    for i in range(60):
        n = int(rng.randint(3, 12))
        words = [q_words[int(j)] for j in rng.randint(0, len(q_words), n)]
        end = ['?', '.', ''][i % 3]
        questions.append({'id': i, 'active': (i % 7 != 3), 'question': ' '.join(words) + end,
                          'type': ['presence', 'count', 'comp', 'area'][i % 4]})
    # exact frequency ties: two words that each appear exactly once, in this order
    questions.append({'id': 60, 'active': True, 'question': 'zebra yak?', 'type': 'presence'})
    # a long question (truncation past LEN_QUESTION) built from known words
    questions.append({'id': 61, 'active': True, 'question': ' '.join(['road'] * 25) + '?', 'type': 'count'})
    answers = []
    base = ['yes', 'no', 'rural', 'urban', '0', '3', '10', '11', '100', '101', '1000', '1001', '25000',
            '0m2', '7m2', '10m2', '55m2', '100m2', '999m2', '1000m2', '5000m2']
    k = 0
    for rep, w in enumerate(base):
        for _ in range(1 + (rep * 5) % 9):
            answers.append({'id': k, 'active': True, 'answer': w}); k += 1
    for j in range(130):      # > MAX_ANSWERS distinct answers, descending counts then ties
        for _ in range(1 + (130 - j) // 40):
            answers.append({'id': k, 'active': (k % 11 != 5), 'answer': f'w{j}'}); k += 1
    return {'questions': questions}, {'answers': answers}


def g1_vocab():
    import VocabEncoder as RefVE
    qj, aj = make_vocab_inputs()
    qpath = os.path.join(OUT, 'g1_all_questions.json')
    apath = os.path.join(OUT, 'g1_all_answers.json')
    json.dump(qj, open(qpath, 'w'), indent=0)
    json.dump(aj, open(apath, 'w'), indent=0)
    enc_q = RefVE.VocabEncoder(qpath, questions=True)
    out = {'max_answers': RefVE.MAX_ANSWERS, 'len_question': RefVE.LEN_QUESTION,
           'question_vocab': enc_q.getVocab(), 'question_words': enc_q.words}
    probes = [q['question'] for q in qj['questions'] if q['active']][:25]
    probes += ['Is there a road?', 'how many building are in the image.', 'road', ' '.join(['water'] * 19),
               ' '.join(['water'] * 20) + '?', ' '.join(['water'] * 40)]
    out['question_encodings'] = [[s, enc_q.encode(s)] for s in probes]
    out['question_decodings'] = [[enc_q.encode(s), enc_q.decode(enc_q.encode(s))] for s in probes[:10]]
    try:
        enc_q.encode('is there a unicorn?')
        out['question_oov'] = 'no error'
    except KeyError as e:
        out['question_oov'] = ['KeyError', e.args[0]]
    for rn in (False, True):
        enc_a = RefVE.VocabEncoder(apath, questions=False, range_numbers=rn)
        key = 'answers_range' if rn else 'answers_plain'
        aprobes = ['yes', 'No', '3', '10', '11', '100', '101', '1000', '1001', '25000', '7m2', '10m2',
                   '55m2', '100m2', '999m2', '1000m2', '5000m2', 'w0', 'w5', 'rural.', 'urban?']
        encs = []
        for s in aprobes:
            try:
                encs.append([s, enc_a.encode(s)])
            except KeyError as e:
                encs.append([s, ['KeyError', e.args[0]]])
        out[key] = {'vocab': enc_a.getVocab(), 'words': enc_a.words, 'encodings': encs,
                    'decodings': [[[i], enc_a.decode([i])] for i in (1, 2, 5, len(enc_a.getVocab()) - 1)]}
    json.dump(out, open(os.path.join(OUT, 'g1_vocab_expected.json'), 'w'), indent=0)
    print('G1 vocab:', len(out['question_vocab']), 'question words;',
          len(out['answers_plain']['vocab']), 'answer classes')


def g2_seq2vec():
    sys.modules.setdefault('skipthoughts', types.ModuleType('skipthoughts'))
    import seq2vec as ref_s2v
    torch.manual_seed(42)
    vocab = ['robots', 'are', 'very', 'cool', '<eos>', 'BiDiBu']          # seq2vec.py:102
    E, H = 20, 24
    m = ref_s2v.LSTM(vocab, E, H, 1).eval()
    ids_main = torch.LongTensor([[1, 2, 3, 4, 5, 0, 0], [6, 1, 2, 3, 3, 4, 5], [6, 1, 2, 3, 3, 4, 5]])  # seq2vec.py:105-109
    g = torch.Generator().manual_seed(7)
    ids_rand = torch.randint(1, 7, (8, 14), generator=g)
    lens = torch.tensor([14, 13, 9, 3, 1, 0, 7, 14])
    ids_rand = torch.where(torch.arange(14)[None, :] < lens[:, None], ids_rand, torch.zeros_like(ids_rand))
    out = {'embedding': m.embedding.weight.detach().numpy(),
           'w_ih': m.rnn.weight_ih_l0.detach().numpy(), 'w_hh': m.rnn.weight_hh_l0.detach().numpy(),
           'b_ih': m.rnn.bias_ih_l0.detach().numpy(), 'b_hh': m.rnn.bias_hh_l0.detach().numpy()}
    for name, ids in (('main', ids_main), ('rand', ids_rand)):
        with torch.no_grad():
            lengths = torch.stack(ref_s2v.process_lengths(ids)) if ids.shape[0] > 1 else None
            e = m.embedding(ids)                                   # (B, L, E)
            hs, _ = m.rnn(e.transpose(0, 1))                       # time-major recurrence: (L, B, H)
            hs_bt = hs.transpose(0, 1).contiguous()                # (B, L, H)
            last = ref_s2v.select_last(hs_bt.clone(), [int(x) for x in lengths])
            as_written = m(ids)                                    # the reference forward as committed
        out[f'{name}_ids'] = ids.numpy()
        out[f'{name}_lengths'] = lengths.numpy()
        out[f'{name}_emb'] = e.numpy()
        out[f'{name}_hs'] = hs_bt.numpy()
        out[f'{name}_last'] = last.numpy()
        out[f'{name}_as_written'] = as_written.numpy()
    np.savez(os.path.join(OUT, 'g2_seq2vec.npz'), **out)
    print('G2 seq2vec: embedding', out['embedding'].shape, 'row0 zero:', bool((out['embedding'][0] == 0).all()))


def g2b_two_lstm():
    """The reference's TwoLSTM (seq2vec.py:48-76): its own embedding / rnn_0 / rnn_1 sub-modules, tanh on
    the embeddings, select_last of both layers, concatenation (eval mode: no dropout), run along the
    token axis, plus the forward as committed (which recurs along the batch axis)."""
    sys.modules.setdefault('skipthoughts', types.ModuleType('skipthoughts'))
    import seq2vec as ref_s2v
    torch.manual_seed(43)
    vocab = ['robots', 'are', 'very', 'cool', '<eos>', 'BiDiBu']          # seq2vec.py:102
    E, H = 20, 16
    m = ref_s2v.TwoLSTM(vocab, E, H).eval()
    ids_main = torch.LongTensor([[1, 2, 3, 4, 5, 0, 0], [6, 1, 2, 3, 3, 4, 5], [6, 1, 2, 3, 3, 4, 5]])  # seq2vec.py:105-109
    g = torch.Generator().manual_seed(8)
    ids_rand = torch.randint(1, 7, (9, 14), generator=g)
    lens = torch.tensor([14, 13, 9, 3, 1, 0, 7, 14, 2])
    ids_rand = torch.where(torch.arange(14)[None, :] < lens[:, None], ids_rand, torch.zeros_like(ids_rand))
    out = {'embedding': m.embedding.weight.detach().numpy()}
    for li, rnn in enumerate((m.rnn_0, m.rnn_1)):
        for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'):
            out[f'rnn_{li}.{k}'] = getattr(rnn, k).detach().numpy()
    for name, ids in (('main', ids_main), ('rand', ids_rand)):
        with torch.no_grad():
            lengths = [int(x) for x in ref_s2v.process_lengths(ids)]
            e = torch.tanh(m.embedding(ids))                       # seq2vec.py:63-64
            x0, _ = m.rnn_0(e.transpose(0, 1))                     # time-major recurrence
            x1, _ = m.rnn_1(x0)
            v0 = ref_s2v.select_last(x0.transpose(0, 1).contiguous().clone(), lengths)
            v1 = ref_s2v.select_last(x1.transpose(0, 1).contiguous().clone(), lengths)
            as_written = m(ids)
        out[f'{name}_ids'] = ids.numpy()
        out[f'{name}_lengths'] = np.array(lengths)
        out[f'{name}_last'] = torch.cat((v0, v1), 1).numpy()       # seq2vec.py:75
        out[f'{name}_as_written'] = as_written.numpy()
    np.savez(os.path.join(OUT, 'g2b_two_lstm.npz'), **out)
    print('G2b TwoLSTM:', out['rand_last'].shape)


TINY = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)


def g3_model():
    for tag, hops, cap in (('h1', 1, False), ('h2', 2, False), ('h1cap', 1, True)):
        kw = dict(TINY)
        if cap:
            kw.update(vc=20, Ec=8)
        p = vqa_ref.init_params(hops=hops, seed=3, **kw)
        b = vqa_ref.synthetic_batch(4, 9, kw['Dv'], 7, kw['vq'], kw['C'], seed=5, Tc=6 if cap else 0, vc=20 if cap else 0)
        b['tokens'][3] = 0      # an all-pad question row
        logits, loss, grads = vqa_ref.loss_and_grads(p, b['feats'], b['tokens'], b['answers'],
                                                     b.get('captions'), hops)
        seeds = [vqa_ref.site_seed(1234, 7, s) for s in range(4)]
        logits_d, loss_d, grads_d = vqa_ref.loss_and_grads(p, b['feats'], b['tokens'], b['answers'],
                                                           b.get('captions'), hops, drop_seeds=seeds)
        out = {f'param:{k}': v.numpy() for k, v in p.items()}
        out.update({f'in:{k}': v.numpy() for k, v in b.items()})
        out.update({'logits': logits.numpy(), 'loss': loss.numpy(), 'logits_drop': logits_d.numpy(),
                    'loss_drop': loss_d.numpy(), 'drop_seeds': np.array(seeds, dtype=np.int64)})
        out.update({f'grad:{k}': v.numpy() for k, v in grads.items()})
        out.update({f'grad_drop:{k}': v.numpy() for k, v in grads_d.items()})
        np.savez(os.path.join(OUT, f'g3_model_{tag}.npz'), **out)
        print('G3', tag, 'loss', float(loss), 'loss(dropout)', float(loss_d))


def g5_ce_g6_metrics():
    g = torch.Generator().manual_seed(11)
    z = torch.randn((6, 13), generator=g)
    z[1] *= 1e4                      # huge logits: stability
    z[2, 4] = z[2, 9] = z[2].max() + 1.0   # exact tie for the argmax
    y = torch.tensor([10, 3, 4, 12, 0, 5])
    loss = vqa_ref.cross_entropy(z, y)
    ref_loss = torch.nn.functional.cross_entropy(z, y)
    assert abs(float(loss) - float(ref_loss)) <= 1e-6 * abs(float(ref_loss))
    dz = vqa_ref.cross_entropy_grad(z, y)
    za = z.numpy().copy()
    za_nan = za.copy(); za_nan[4, 7] = np.nan
    types = np.array([0, 1, 2, 0, 1, 2])          # type 3 stays empty
    pred = vqa_ref.argmax_first(za)
    count, right = vqa_ref.accuracy_counts(pred, y.numpy(), types)
    oa, aa = vqa_ref.oa_aa(count, right)
    np.savez(os.path.join(OUT, 'g5_ce_g6_metrics.npz'), logits=za, target=y.numpy(), loss=loss.numpy(),
             dlogits=dz.numpy(), logits_nan=za_nan, argmax=pred, argmax_nan=vqa_ref.argmax_first(za_nan),
             types=types, count=count, right=right, oa=oa, aa=aa,
             dropout_mask_seed77_p05=vqa_ref.dropout_keep_mask(77, 64, 0.5),
             dropout_mask_seed5_p03=vqa_ref.dropout_keep_mask(5, 64, 0.3))
    print('G5/G6 loss', float(loss), 'argmax', pred, 'OA', oa, 'AA', aa)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    g1_vocab()
    g2_seq2vec()
    g2b_two_lstm()
    g3_model()
    g5_ce_g6_metrics()
