"""The fused attention entry points called directly through the C ABI in bf16, at region counts and widths
that are not multiples of the kernels' chunk sizes, against the same arithmetic in fp64 from the same bf16
inputs (scores w_a . tanh(P_r + qa) + c, softmax over regions, pooled = sum_r p_r inv_norm_r V_r; with
w_a = 0 this is the mean over regions that the reference pools with, model.py:161-164)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(P, V, qa, wa, c, inv):
    B, R, A = P.shape
    s = (torch.tanh(P.double() + qa.double()[:, None, :]) * wa.double()).sum(-1) + c.double()
    p = torch.softmax(s, dim=1)
    pooled = ((p * inv.double())[:, :, None] * V.double()).sum(1)
    return p, pooled


@pytest.mark.parametrize('B,R,A,Dv', [
    (3, 1, 8, 8), (2, 15, 24, 72), (2, 16, 64, 64), (2, 17, 1200, 2048), (1, 197, 520, 1600),
    (2, 33, 2048, 512), (1, 1000, 64, 128), (2, 50, 1200, 4096), (4, 196, 1200, 2048)])
def test_attn_fwd_bf16_shapes(B, R, A, Dv):
    from vqa_amd._lib import call, ptr, BF16
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + R)
    dt = torch.bfloat16
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    qa = torch.randn(B, A, device='cuda', generator=g)
    wa = torch.randn(A, device='cuda', generator=g) * (2.0 / A ** 0.5)
    c = torch.full((1,), 0.25, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    p = torch.full((B, R), -1.0, device='cuda')
    pooled32 = torch.full((B, Dv), -1.0, device='cuda')
    pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
    call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled),
         ptr(pooled32), Dv, B, R, A, Dv, BF16, None, 0, 0.0)
    torch.cuda.synchronize()
    p_ref, pooled_ref = _ref(P, V, qa, wa, c, inv)
    assert torch.allclose(p.double().sum(1), torch.ones(B, device='cuda', dtype=torch.float64), atol=1e-5)
    assert float((p.double() - p_ref).abs().max()) <= 2e-5 * max(1.0, float(p_ref.max()) * 10)
    err = float((pooled32.double() - pooled_ref).abs().max() / pooled_ref.abs().max())
    assert err <= 2e-5, err
    assert torch.equal(pooled, pooled32.to(dt))


def test_attn_fwd_dropout_output_matches_separate_dropout():
    """The fused dropout output equals the dropout kernel applied to the pooled vector with the same seed."""
    from vqa_amd._lib import call, ptr, BF16
    B, R, A, Dv = 5, 37, 1200, 2048
    g = torch.Generator(device='cuda').manual_seed(5)
    dt = torch.bfloat16
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    qa = torch.randn(B, A, device='cuda', generator=g)
    wa = torch.randn(A, device='cuda', generator=g) * 0.05
    c = torch.zeros(1, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    p = torch.zeros(B, R, device='cuda')
    pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
    fused = torch.zeros(B, Dv, device='cuda', dtype=dt)
    call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled),
         None, Dv, B, R, A, Dv, BF16, ptr(fused), 4242, 0.5)
    sep = torch.zeros_like(pooled)
    call('vqa_dropout', ptr(pooled), Dv, ptr(sep), Dv, B, Dv, 4242, 0.5, BF16)
    torch.cuda.synchronize()
    assert torch.equal(fused, sep)
    kept = (fused != 0).float().mean()
    assert 0.3 < float(kept) < 0.7


@pytest.mark.parametrize('B,R,A,Dv', [(3, 1, 8, 8), (2, 15, 24, 72), (2, 17, 1200, 2048), (1, 197, 520, 1600),
                                      (2, 33, 2048, 512), (2, 50, 1200, 4096)])
def test_attn_bwd_bf16_shapes(B, R, A, Dv):
    """dP, d qa, per-sample partials of d w_a and d c against autograd in fp64 on the same bf16 inputs."""
    from vqa_amd._lib import call, ptr, BF16
    g = torch.Generator(device='cuda').manual_seed(B * 77 + R)
    dt = torch.bfloat16
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    qa = torch.randn(B, A, device='cuda', generator=g)
    wa = torch.randn(A, device='cuda', generator=g) * (2.0 / A ** 0.5)
    c = torch.zeros(1, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    dpool = torch.randn(B, Dv, device='cuda', generator=g)

    Pd = P.double().requires_grad_(True)
    qad = qa.double().requires_grad_(True)
    wad = wa.double().requires_grad_(True)
    cd = c.double().requires_grad_(True)
    p_ref, pooled_ref = _ref(Pd, V, qad, wad, cd, inv)
    (pooled_ref * dpool.double()).sum().backward()

    p = p_ref.detach().float().contiguous()
    dP = torch.zeros(B, R, A, device='cuda', dtype=dt)
    dqa = torch.zeros(B, A, device='cuda')
    dqa_t = torch.zeros(B, A, device='cuda', dtype=dt)
    dwa = torch.zeros(B, A, device='cuda')
    dc = torch.zeros(B, device='cuda')
    call('vqa_attn_bwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dpool), Dv, ptr(dP), A,
         ptr(dqa), ptr(dqa_t), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
    torch.cuda.synchronize()

    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    if R > 1:                                            # R = 1: the softmax is constant, every gradient is zero
        # dP is written pre-multiplied by inv_norm (the weight gradient then needs no row scale); stored in bf16
        assert rel(dP, Pd.grad * inv.double()[:, :, None]) <= 1e-2
        assert rel(dqa, qad.grad) <= 1e-4
        assert rel(dwa.sum(0), wad.grad) <= 1e-4
        assert torch.equal(dqa_t, dqa.to(dt))
    else:
        assert float(dP.float().abs().max()) <= 1e-6 and float(dqa.abs().max()) <= 1e-6
    assert float(dc.double().sum().abs()) <= 1e-4 * max(1.0, float(dpool.abs().max()))


@pytest.mark.parametrize('B,R,A,Dv', [(4, 196, 1200, 2048), (2, 17, 24, 72), (2, 50, 1200, 4096)])
def test_attention_fed_with_partial_sums_matches_reduced_operands(B, R, A, Dv):
    """vqa_attn_fwd_qslabs / vqa_attn_bwd_dslabs = vqa_epilogue + vqa_attn_fwd / vqa_attn_bwd, bit for bit
    (streaming kernels: the prologue reduces; Dv = 4096: the entry reduces first)."""
    from vqa_amd._lib import call, ptr, BF16, F32
    g = torch.Generator(device='cuda').manual_seed(B + R)
    dt = torch.bfloat16
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    wa = torch.randn(A, device='cuda', generator=g) * (2.0 / A ** 0.5)
    c = torch.zeros(1, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    qs = torch.randn(3, B, A, device='cuda', generator=g) * 0.5
    qb = torch.randn(A, device='cuda', generator=g)
    dsl = torch.randn(2, B, Dv, device='cuda', generator=g)
    outs = []
    for fused in (False, True):
        qa = torch.zeros(B, A, device='cuda')
        p = torch.zeros(B, R, device='cuda')
        pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
        pd = torch.zeros(B, Dv, device='cuda', dtype=dt)
        dpool = torch.zeros(B, Dv, device='cuda')
        dP = torch.zeros(B, R, A, device='cuda', dtype=dt)
        dqa = torch.zeros(B, A, device='cuda'); dqa_t = torch.zeros(B, A, device='cuda', dtype=dt)
        dwa = torch.zeros(B, A, device='cuda'); dc = torch.zeros(B, device='cuda')
        if fused:
            call('vqa_attn_fwd_qslabs', ptr(P), A, ptr(V), Dv, ptr(qs), 3, B * A, ptr(qb), ptr(qa), A, ptr(wa), ptr(c),
                 ptr(inv), ptr(p), ptr(pooled), None, Dv, B, R, A, Dv, BF16, ptr(pd), 99, 0.5)
            call('vqa_attn_bwd_dslabs', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dsl), 2, B * Dv,
                 321, 0.5, ptr(dpool), Dv, ptr(dP), A, ptr(dqa), ptr(dqa_t), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
        else:
            call('vqa_epilogue', ptr(qs), 3, B * A, B, A, ptr(qb), None, None, 0, 0, ptr(qa), A, None, 0, BF16, 0, 0.0)
            call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled), None,
                 Dv, B, R, A, Dv, BF16, ptr(pd), 99, 0.5)
            call('vqa_epilogue', ptr(dsl), 2, B * Dv, B, Dv, None, None, None, 0, 0, None, 0, ptr(dpool), Dv, F32, 321, 0.5)
            call('vqa_attn_bwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p), ptr(dpool), Dv, ptr(dP), A,
                 ptr(dqa), ptr(dqa_t), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16)
        torch.cuda.synchronize()
        outs.append((qa, p, pooled, pd, dP, dqa, dqa_t, dwa, dc))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('B,R,A,Dv', [(40, 196, 1200, 2048), (70, 196, 1200, 2048), (3, 37, 520, 1600), (1, 1000, 64, 128),
                                      (5, 33, 2048, 512), (17, 576, 1200, 2048)])
def test_small_batch_row_split_matches_one_workgroup_per_sample(B, R, A, Dv):
    """vqa_attn_fwd_split / vqa_attn_bwd_split (a sample's rows shared by several workgroups; the reference's own batch
    sizes are 40 and 70, train.py:242, test.py:120) against fp64 and against the one-workgroup-per-sample kernels, for
    every split count the shape admits, dropout output included."""
    from vqa_amd._lib import call, ptr, BF16, F32
    g = torch.Generator(device='cuda').manual_seed(B * 31 + R)
    dt = torch.bfloat16
    P = torch.randn(B, R, A, device='cuda', generator=g).to(dt)
    V = torch.randn(B, R, Dv, device='cuda', generator=g).clamp_min(0).to(dt)
    qa = torch.randn(B, A, device='cuda', generator=g)
    wa = torch.randn(A, device='cuda', generator=g) * (4.0 / A ** 0.5)        # (peaked attention: the maxima of the ranges differ)
    c = torch.full((1,), 0.25, device='cuda')
    inv = torch.rand(B, R, device='cuda', generator=g) + 0.5
    dpool = torch.randn(B, Dv, device='cuda', generator=g)
    auto = call('vqa_attn_splits', B, R, A, Dv, BF16)
    assert 1 <= auto <= 8 and (B * auto <= 256 or auto == 1)
    if (B, R) in ((40, 196), (70, 196)):
        assert auto == {40: 6, 70: 3}[B]
    assert call('vqa_attn_splits', 256, R, A, Dv, BF16) == 1 and call('vqa_attn_splits', B, R, A, Dv, F32) == 1

    Pd = P.double().requires_grad_(True)
    qad = qa.double().requires_grad_(True)
    wad = wa.double().requires_grad_(True)
    cd = c.double().requires_grad_(True)
    p_ref, pooled_ref = _ref(Pd, V, qad, wad, cd, inv)
    (pooled_ref * dpool.double()).sum().backward()

    def rel(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))

    # one workgroup per sample
    p1 = torch.zeros(B, R, device='cuda'); pooled1 = torch.zeros(B, Dv, device='cuda', dtype=dt); drop1 = torch.zeros_like(pooled1)
    call('vqa_attn_fwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p1), ptr(pooled1), None,
         Dv, B, R, A, Dv, BF16, ptr(drop1), 77, 0.5)
    dP1 = torch.zeros(B, R, A, device='cuda', dtype=dt)
    dqa1 = torch.zeros(B, A, device='cuda'); dqat1 = torch.zeros(B, A, device='cuda', dtype=dt)
    dwa1 = torch.zeros(B, A, device='cuda'); dc1 = torch.zeros(B, device='cuda')
    call('vqa_attn_bwd', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p1), ptr(dpool), Dv, ptr(dP1), A,
         ptr(dqa1), ptr(dqat1), A, ptr(dwa1), ptr(dc1), B, R, A, Dv, BF16)
    torch.cuda.synchronize()

    nch = (R + 15) // 16
    for ns in sorted({2, min(3, nch), min(8, nch), auto} - {1}):
        if ns > nch:
            continue
        wsf = torch.full((call('vqa_attn_split_workspace_floats', B, R, Dv, ns),), float('nan'), device='cuda')
        p = torch.full((B, R), -1.0, device='cuda')
        pooled32 = torch.zeros(B, Dv, device='cuda'); pooled = torch.zeros(B, Dv, device='cuda', dtype=dt)
        drop = torch.zeros_like(pooled)
        call('vqa_attn_fwd_split', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(c), ptr(inv), ptr(p), ptr(pooled),
             ptr(pooled32), Dv, B, R, A, Dv, BF16, ptr(drop), 77, 0.5, ns, ptr(wsf))
        torch.cuda.synchronize()
        assert torch.allclose(p.double().sum(1), torch.ones(B, device='cuda', dtype=torch.float64), atol=1e-5)
        assert float((p.double() - p_ref).abs().max()) <= 2e-5 * max(1.0, float(p_ref.max()) * 10), ns
        assert rel(pooled32, pooled_ref.detach()) <= 2e-5, ns
        assert torch.equal(pooled, pooled32.to(dt))
        assert float((pooled.float() - pooled1.float()).abs().max()) <= 1e-2 * float(pooled1.float().abs().max())
        sep = torch.zeros_like(pooled)
        call('vqa_dropout', ptr(pooled), Dv, ptr(sep), Dv, B, Dv, 77, 0.5, BF16)
        torch.cuda.synchronize()
        assert torch.equal(drop, sep)
        # backward, from the same probabilities as the one-workgroup run
        dP = torch.zeros(B, R, A, device='cuda', dtype=dt)
        slabs = torch.full((ns, B, A), float('nan'), device='cuda')
        dwa = torch.full((B * ns, A), float('nan'), device='cuda'); dc = torch.full((B * ns,), float('nan'), device='cuda')
        call('vqa_attn_bwd_split', ptr(P), A, ptr(V), Dv, ptr(qa), A, ptr(wa), ptr(inv), ptr(p1), ptr(dpool), Dv, ptr(dP), A,
             ptr(slabs), A, ptr(dwa), ptr(dc), B, R, A, Dv, BF16, ns, ptr(wsf))
        dqa = torch.zeros(B, A, device='cuda'); dqat = torch.zeros(B, A, device='cuda', dtype=dt)
        call('vqa_epilogue', ptr(slabs), ns, B * A, B, A, None, None, None, 0, 0, ptr(dqa), A, ptr(dqat), A, BF16, 0, 0.0)
        torch.cuda.synchronize()
        assert torch.equal(dP, dP1), ns                 # row by row the same arithmetic from the same sums
        assert rel(dqa, qad.grad) <= 1e-4 and rel(dqa, dqa1.double()) <= 1e-5
        assert torch.equal(dqat, dqa.to(dt))
        assert rel(dwa.sum(0), wad.grad) <= 1e-4 and rel(dwa.sum(0), dwa1.double().sum(0)) <= 1e-5
        assert abs(float(dc.double().sum()) - float(dc1.double().sum())) <= 1e-4 * max(1.0, float(dpool.abs().max()))
