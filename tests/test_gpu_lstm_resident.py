"""vqa_lstm_resident (all recurrent steps after the two tabulated ones in one launch, W_hh resident in registers)
against the per-step kernels it replaces and against the oracle's LSTM (seq2vec.py:11-45 restated in
oracle/vqa_ref.py), through the module's encode_question, i.e. through the C ABI."""
import pytest
import torch

from oracle import vqa_ref

pytestmark = pytest.mark.gpu


def _model(H, E=24, vq=60, seed=5):
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=vq, C=11, E=E, Hq=H, Dv=64, A=16, F=16, Hf=8)
    p = vqa_ref.init_params(seed=seed, **kw)
    m = VQAAttnModel(['w'] * vq, ['a'] * kw['C'], None, dtype=torch.bfloat16, emb_size=E, question_out=H,
                     visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=1)
    m.load_state_dict(p)
    return m.to('cuda').eval(), {k: v.to(torch.bfloat16).float() for k, v in p.items()}


def _tokens(B, L, vq, seed, full=False):
    g = torch.Generator().manual_seed(seed)
    tok = torch.randint(1, vq, (B, L), generator=g)
    if not full:
        lens = torch.randint(1, L + 1, (B,), generator=g)
        for b in range(B):
            tok[b, int(lens[b]):] = 0
    if B > 3 and not full:
        tok[2] = 0                    # an all-pad row takes all L steps (seq2vec.py:21: index -1)
        tok[3, 1:] = 0                # a one-token question ends inside the tabulated steps
    return tok


def _encode(m, tok, resident, drop=None, whole=True):
    m.resident_encoder = resident
    m.resident_whole = whole       # True: packing and the two tabulated steps run inside the same launch
    B = tok.shape[0]
    qd = None
    if drop is not None:
        qd = torch.zeros((B, m.Hq), dtype=torch.bfloat16, device='cuda')
    q = m.encode_question(tok.cuda(), q_drop=None if drop is None else (qd, drop[0], drop[1])).clone()
    ws = m._enc_workspace(B, tok.shape[1], torch.device('cuda', torch.cuda.current_device()))
    qt = ws['qT'].clone()
    used = m._xg_table_u is not None
    m.check_status()
    return q, qt, qd, used


@pytest.mark.parametrize('whole', [True, False])
@pytest.mark.parametrize('B,H,L', [(5, 48, 7), (70, 104, 14), (64, 16, 3), (300, 200, 9), (512, 72, 5)])
def test_resident_steps_match_per_step_kernels_and_oracle(B, H, L, whole):
    m, po = _model(H)
    tok = _tokens(B, L, 60, seed=B + H + L)
    q_res, qt_res, qd_res, used = _encode(m, tok, True, drop=(1234, 0.5), whole=whole)
    if whole:        # what vqa_sequence_pack would have left in the workspace
        ws = m._enc_workspace(B, L, torch.device('cuda', torch.cuda.current_device()))
        packed = {k: ws[k].clone() for k in ('len', 'perm', 'm_t')}
    assert used, 'the resident kernel was not selected'
    q_ref, qt_ref, qd_ref, used_ref = _encode(m, tok, False, drop=(1234, 0.5))
    assert not used_ref
    if whole:
        for k, v in packed.items():
            assert torch.equal(v, ws[k]), k          # bit-exact integer bookkeeping
    scale = float(q_ref.abs().max())
    # same arithmetic up to the order of the fp32 sums of the recurrent product; h is rounded to bf16 every step
    assert float((q_res - q_ref).abs().max()) <= 1e-2 * scale
    assert float((qt_res.float() - qt_ref.float()).abs().max()) <= 2e-2 * scale
    # the dropped copy: same mask (hash of seed, original row, column), values of the stored bf16 vector x 2
    keep_res, keep_ref = qd_res != 0, qd_ref != 0
    both = (qt_res != 0) & (qt_ref != 0)
    assert torch.equal(keep_res & both, keep_ref & both)
    assert torch.equal(qd_res[keep_res].float(), (qt_res[keep_res].float() * 2).to(torch.bfloat16).float())
    ref = vqa_ref.encode_question(tok, po)
    assert float((q_res.cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_resident_steps_full_width_and_repeatable():
    """The benchmarked width (E = 620, H = 2400: 200 workgroups, one per CU, 19 K sub-steps per wave) at B = 256,
    L = 14; three launches in a row share the barrier words (their counters only ever grow) and give the same bits."""
    m, po = _model(2400, E=620, vq=300)
    tok = _tokens(256, 14, 300, seed=9)
    q1, qt1, _, used = _encode(m, tok, True)
    assert used
    q_steps, _, _, _ = _encode(m, tok, True, whole=False)      # the form that starts after the two tabulated steps
    assert float((q_steps - q1).abs().max()) <= 1e-2 * float(q1.abs().max())      # (its first two steps: another kernel)
    q2, qt2, _, _ = _encode(m, tok, True)
    tok_b = _tokens(256, 14, 300, seed=10, full=True)      # every row runs all 14 steps
    qb, _, _, _ = _encode(m, tok_b, True)
    q3, qt3, _, _ = _encode(m, tok, True)
    assert torch.equal(q1, q2) and torch.equal(q1, q3) and torch.equal(qt1, qt3)
    q_ref, _, _, _ = _encode(m, tok, False)
    qb_ref, _, _, _ = _encode(m, tok_b, False)
    assert float((q1 - q_ref).abs().max()) <= 1e-2 * float(q_ref.abs().max())
    assert float((qb - qb_ref).abs().max()) <= 1e-2 * float(qb_ref.abs().max())
    ref = vqa_ref.encode_question(tok, po)
    assert float((q1.cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_resident_entry_point_rejects_what_it_cannot_run():
    from vqa_amd import _lib
    lib = _lib.load()
    assert lib.vqa_lstm_resident_workgroups(2400) == 200
    assert lib.vqa_lstm_resident_workgroups(2440) == 0 and lib.vqa_lstm_resident_workgroups(20) == 0
    x = torch.zeros(4096, dtype=torch.int64, device='cuda')
    p = x.data_ptr()
    args = lambda H, B, t0, L, hh=None: (p, H, p, hh, 10, p, p, p, p, p, p, H, B, H, t0, L, p, p, None, 0, 0.0, p, None, None)
    for bad in (args(2440, 8, 2, 5), args(48, 600, 2, 5), args(48, 8, 0, 5), args(44, 8, 2, 5), args(48, 8, 0, 1, p),
                args(48, 8, 2, 65)):
        assert lib.vqa_lstm_resident(*bad) == -22
        assert b'vqa_lstm_resident' in lib.vqa_last_error_string()


def test_resident_barrier_abort_raises_and_recovers():
    """A barrier that cannot complete (here: the abort word is set before the launch) makes every workgroup leave, sets
    the status word, and the module raises at its next check, clears the barrier words and works again."""
    from vqa_amd import _lib
    m, _ = _model(104)
    tok = _tokens(40, 9, 60, seed=4)
    q_good, _, _, used = _encode(m, tok, True)
    assert used
    m._resident_sync[16 * 8] = 1                     # the abort word: one 128-byte line per word, ninth line
    m.resident_encoder = True
    m.encode_question(tok.cuda())
    with pytest.raises(_lib.VqaHipError, match='grid barrier'):
        m.check_status()
    assert int(m._resident_sync.abs().sum()) == 0
    q_again, _, _, _ = _encode(m, tok, True)
    assert torch.equal(q_again, q_good)


def test_resident_handoff_is_repeatable_under_load():
    """The hidden state crosses the chip once per step through global memory (write-through stores, arrival counters,
    one acquire).  A stale read would depend on timing: 150 launches on the same questions must give the same bits,
    idle and beside a second stream that keeps HBM, the L2s and some compute units busy."""
    m, _ = _model(2400, E=620, vq=300)
    tok = _tokens(256, 14, 300, seed=21).cuda()
    m.resident_encoder = True
    first = m.encode_question(tok).clone()
    assert m._xg_table_u is not None
    big = torch.empty(64 << 20, dtype=torch.float32, device='cuda')       # 256 MiB: larger than the L2s together
    side = torch.cuda.Stream()
    stop = 150
    for load in (False, True):
        for i in range(stop):
            if load:
                with torch.cuda.stream(side):
                    big.add_(1.0)
            q = m.encode_question(tok)
            if i % 10 == 9 or i == stop - 1:
                assert torch.equal(q, first), (load, i)
    torch.cuda.synchronize()
    m.check_status()


@pytest.mark.parametrize('B,L', [(600, 6), (9, 70)])
def test_shapes_beyond_the_resident_limits_use_the_per_step_kernels(B, L):
    m, po = _model(48)
    tok = _tokens(B, L, 60, seed=B + L)
    q, _, _, _ = _encode(m, tok, True)
    ref = vqa_ref.encode_question(tok, po)
    assert float((q.cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_train_steps_agree_between_the_encoder_forms():
    """Three fused train steps (dropout on) with the one-launch encoder, with the three-launch form and with the
    per-step kernels: the same losses up to the bf16 noise of the question vector, and the same loss as the oracle."""
    from vqa_amd.model import VQAAttnModel
    kw = dict(vq=60, C=13, E=24, Hq=104, Dv=64, A=24, F=40, Hf=16)
    p = vqa_ref.init_params(seed=8, **kw)
    b = vqa_ref.synthetic_batch(37, 9, kw['Dv'], 11, kw['vq'], kw['C'], seed=5)
    feats = b['feats'].to(torch.bfloat16)
    losses = {}
    for form in ('whole', 'steps', 'per-step'):
        m = VQAAttnModel(['w'] * kw['vq'], ['a'] * kw['C'], None, dtype=torch.bfloat16, emb_size=kw['E'], question_out=kw['Hq'],
                         visual_out=kw['Dv'], attention_dim=kw['A'], fusion_in=kw['F'], fusion_hidden=kw['Hf'], seed=3)
        m.load_state_dict(p)
        m = m.to('cuda').train()
        m.resident_encoder, m.resident_whole = form != 'per-step', form == 'whole'
        losses[form] = [float(m.train_step(feats.cuda(), b['tokens'].cuda(), b['answers'].cuda(), lr=1e-3)) for _ in range(3)]
        m.sync()
        assert (m._xg_table_u is not None) == (form != 'per-step')
    for form in ('whole', 'steps'):
        for got, want in zip(losses[form], losses['per-step']):
            assert abs(got - want) <= 1e-2 * abs(want), (form, losses)
    po = {k: v.to(torch.bfloat16).float() for k, v in p.items()}
    seeds = [vqa_ref.site_seed(3, 1, s) for s in range(4)]
    _, loss_ref, _ = vqa_ref.loss_and_grads(po, feats.float(), b['tokens'], b['answers'], None, 1, drop_seeds=seeds)
    assert abs(losses['whole'][0] - float(loss_ref)) <= 1e-2 * float(loss_ref)
