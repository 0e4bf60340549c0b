"""The launch-merging entry point against the two-launch sequence it replaces (bit-exact, fp32 and bf16):
vqa_epilogue_fusion = vqa_epilogue + vqa_fusion_fwd."""
import pytest
import torch

from vqa_amd._lib import call, ptr, BF16, F32

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,F,pad,sk', [(256, 1200, 0, 6), (7, 40, 8, 1), (33, 1208, 304, 3)])
def test_epilogue_fusion_matches_two_launches(dtype, B, F, pad, sk):
    dt = F32 if dtype == torch.float32 else BF16
    g = torch.Generator(device='cuda').manual_seed(B + F)
    slabs = torch.randn(sk, B, F, device='cuda', generator=g) * 0.5
    bias = torch.randn(F, device='cuda', generator=g)
    xv = torch.tanh(torch.randn(B, F, device='cuda', generator=g))
    ldx = F + pad
    xq_a = torch.zeros(B, F, device='cuda'); x_a = torch.zeros(B, ldx, device='cuda', dtype=dtype)
    xq_b = torch.zeros(B, F, device='cuda'); x_b = torch.zeros(B, ldx, device='cuda', dtype=dtype)
    call('vqa_epilogue', ptr(slabs), sk, B * F, B, F, ptr(bias), None, None, 0, 1, ptr(xq_a), F, None, 0, dt, 0, 0.0)
    call('vqa_fusion_fwd', ptr(xv), ptr(xq_a), ptr(x_a), ldx, B, F, 777, 0.5, dt)
    call('vqa_epilogue_fusion', ptr(slabs), sk, B * F, B, F, ptr(bias), ptr(xv), ptr(xq_b), ptr(x_b), ldx, 777, 0.5, dt)
    torch.cuda.synchronize()
    assert torch.equal(xq_a, xq_b)
    assert torch.equal(x_a, x_b)
    assert 0.3 < float((x_b[:, :F] != 0).float().mean()) < 0.7


@pytest.mark.parametrize('M,N,sk', [(2048, 1200, 6), (33, 65, 1), (1, 7, 3), (100, 31, 2), (100, 36, 2), (4, 4, 1), (68, 132, 3)])
def test_slab_sum_transpose(M, N, sk):
    g = torch.Generator(device='cuda').manual_seed(M + N)
    slabs = torch.randn(sk, M, N, device='cuda', generator=g)
    ld = M + 8
    out = torch.full((N, ld), 7.0, device='cuda')
    call('vqa_slab_sum_transpose', ptr(slabs), sk, M * N, M, N, ptr(out), ld)
    torch.cuda.synchronize()
    ref = slabs[0].clone()
    for s in range(1, sk):
        ref += slabs[s]                       # the kernel's summation order
    assert torch.equal(out[:, :M], ref.t())
    assert bool((out[:, M:] == 7.0).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,F,pad,skv,skq', [(256, 1200, 0, 4, 6), (7, 40, 8, 1, 1), (33, 1208, 304, 3, 2)])
def test_fusion_from_both_slab_sets_matches_three_launches(dtype, B, F, pad, skv, skq):
    dt = F32 if dtype == torch.float32 else BF16
    g = torch.Generator(device='cuda').manual_seed(B + F + 1)
    sv = torch.randn(skv, B, F, device='cuda', generator=g) * 0.5
    sq = torch.randn(skq, B, F, device='cuda', generator=g) * 0.5
    bv = torch.randn(F, device='cuda', generator=g)
    bq = torch.randn(F, device='cuda', generator=g)
    ldx = F + pad
    xv_a = torch.zeros(B, F, device='cuda'); xq_a = torch.zeros(B, F, device='cuda')
    x_a = torch.zeros(B, ldx, device='cuda', dtype=dtype)
    xv_b = torch.zeros(B, F, device='cuda'); xq_b = torch.zeros(B, F, device='cuda')
    x_b = torch.zeros(B, ldx, device='cuda', dtype=dtype)
    call('vqa_epilogue', ptr(sv), skv, B * F, B, F, ptr(bv), None, None, 0, 1, ptr(xv_a), F, None, 0, dt, 0, 0.0)
    call('vqa_epilogue', ptr(sq), skq, B * F, B, F, ptr(bq), None, None, 0, 1, ptr(xq_a), F, None, 0, dt, 0, 0.0)
    call('vqa_fusion_fwd', ptr(xv_a), ptr(xq_a), ptr(x_a), ldx, B, F, 31, 0.5, dt)
    call('vqa_fusion_fwd_slabs', ptr(sv), skv, B * F, ptr(bv), ptr(sq), skq, B * F, ptr(bq), ptr(xv_b), ptr(xq_b),
         ptr(x_b), ldx, B, F, 31, 0.5, dt)
    torch.cuda.synchronize()
    assert torch.equal(xv_a, xv_b) and torch.equal(xq_a, xq_b) and torch.equal(x_a, x_b)
