"""GEMM kernels through the C ABI: every MFMA tile configuration (64x64, 128x128, 256x128 ring) and operand
layout against fp64 torch on ragged shapes (M, N not multiples of the tile, K tails, split-K, empty splits),
plus the fused epilogue (row scale, bias, tanh, bf16 / fp32 output)."""
import pytest
import torch

from vqa_amd._lib import call, ptr, BF16, F32

pytestmark = pytest.mark.gpu


def make(M, N, K, akm, bkm, dtype, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    pad8 = lambda n: (n + 7) // 8 * 8
    # logical operands
    A = torch.randn(M, K, device='cuda', generator=g)
    B = torch.randn(N, K, device='cuda', generator=g)
    A = A.to(dtype).float()
    B = B.to(dtype).float()
    def store(X, kmajor):          # (rows, K) logical -> storage with a padded, zero-filled leading dimension
        if kmajor:
            ld = pad8(X.shape[1])
            buf = torch.zeros(X.shape[0], ld, device='cuda', dtype=dtype)
            buf[:, :X.shape[1]] = X.to(dtype)
        else:
            ld = pad8(X.shape[0])
            buf = torch.zeros(X.shape[1], ld, device='cuda', dtype=dtype)
            buf[:, :X.shape[0]] = X.t().to(dtype)
        return buf, ld
    As, lda = store(A, akm)
    Bs, ldb = store(B, bkm)
    ref = A.double() @ B.double().t()
    return As, lda, Bs, ldb, ref


SHAPES = [(37, 50, 72), (130, 70, 200), (300, 129, 64), (257, 384, 1000), (512, 136, 40), (64, 64, 8), (1, 9, 640)]


@pytest.mark.parametrize('engine', [2, 3, 4, 6, 8, 9, 10, 11, 1])
@pytest.mark.parametrize('akm,bkm', [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_slab_gemm_all_layouts(engine, akm, bkm):
    for si, (M, N, K) in enumerate(SHAPES):
        for sk in (1, 3):
            As, lda, Bs, ldb, ref = make(M, N, K, akm, bkm, torch.bfloat16, 10 * si + sk)
            slab = torch.full((sk, M, N), float('nan'), device='cuda')
            call('vqa_gemm_slabs', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(slab), M * N, N, M, N, K, sk, BF16, engine)
            torch.cuda.synchronize()
            got = slab.double().sum(0)
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err <= 2e-5, (engine, akm, bkm, M, N, K, sk, err)


@pytest.mark.parametrize('engine', [10, 11])
@pytest.mark.parametrize('akm,bkm', [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_eight_phase_engine_whole_k_tiles(akm, bkm, engine):
    """The eight-phase 256 x 256 engine takes contractions made of whole 64-wide K tiles (others fall back to the
    ring engine): one K tile, two, an odd count, many; ragged M / N; split-K with an uneven last split."""
    for si, (M, N, K) in enumerate([(300, 129, 64), (515, 300, 128), (257, 520, 192), (100, 1200, 640), (1000, 264, 2048)]):
        for sk in (1, 3):
            As, lda, Bs, ldb, ref = make(M, N, K, akm, bkm, torch.bfloat16, 300 + 10 * si + sk)
            slab = torch.full((sk, M, N), float('nan'), device='cuda')
            call('vqa_gemm_slabs', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(slab), M * N, N, M, N, K, sk, BF16, engine)
            torch.cuda.synchronize()
            got = slab.double().sum(0)
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err <= 2e-5, (engine, akm, bkm, M, N, K, sk, err)


@pytest.mark.parametrize('engine', [2, 3, 4, 6, 8, 9, 10, 11])
@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
def test_fused_epilogue(engine, out_dtype):
    for si, (M, N, K) in enumerate([(300, 136, 256), (515, 1200, 192), (96, 8, 72)]):
        As, lda, Bs, ldb, ref = make(M, N, K, 1, 1, torch.bfloat16, 77 + si)
        g = torch.Generator(device='cuda').manual_seed(si)
        bias = torch.randn(N, device='cuda', generator=g)
        rs = torch.rand(M, device='cuda', generator=g) * 0.2
        ldc = (N + 7) // 8 * 8
        for act in (0, 1):
            C = torch.zeros(M, ldc, device='cuda', dtype=out_dtype)
            call('vqa_gemm_fused', ptr(As), lda, 1, ptr(Bs), ldb, 1, ptr(C), ldc, M, N, K, ptr(bias), ptr(rs), act,
                 BF16 if out_dtype == torch.bfloat16 else F32, BF16, engine)
            torch.cuda.synchronize()
            want = ref * rs.double()[:, None] + bias.double()[None, :]
            if act:
                want = torch.tanh(want)
            tol = 1e-2 if out_dtype == torch.bfloat16 else 3e-5
            err = float((C[:, :N].double() - want).abs().max() / want.abs().max())
            assert err <= tol, (engine, out_dtype, M, N, K, act, err)
            assert float(C[:, N:].abs().max() if ldc > N else 0.0) == 0.0      # padding untouched


@pytest.mark.parametrize('engine', [0, 10, 11])
@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
def test_streaming_stores_large_output(engine, out_dtype):
    """An output larger than the Infinity Cache (256 MB) with 128-byte-aligned rows is written with streaming stores (a
    kernel instance of its own in the 256 x 256 engines)."""
    M, N, K = 106000, 1280, 192
    As, lda, Bs, ldb, ref = make(M, N, K, 1, 1, torch.bfloat16, 21)
    g = torch.Generator(device='cuda').manual_seed(4)
    bias = torch.randn(N, device='cuda', generator=g)
    rs = torch.rand(M, device='cuda', generator=g) * 0.2
    C = torch.zeros(M, N, device='cuda', dtype=out_dtype)
    assert C.numel() * C.element_size() >= 256 << 20 and C.data_ptr() % 128 == 0
    call('vqa_gemm_fused', ptr(As), lda, 1, ptr(Bs), ldb, 1, ptr(C), N, M, N, K, ptr(bias), ptr(rs), 0,
         BF16 if out_dtype == torch.bfloat16 else F32, BF16, engine)
    torch.cuda.synchronize()
    want = ref * rs.double()[:, None] + bias.double()[None, :]
    tol = 1e-2 if out_dtype == torch.bfloat16 else 3e-5
    err = float((C.double() - want).abs().max() / want.abs().max())
    assert err <= tol, (engine, out_dtype, err)


def test_fp32_gemm():
    for akm, bkm in ((1, 1), (1, 0), (0, 0)):
        As, lda, Bs, ldb, ref = make(77, 45, 131, akm, bkm, torch.float32, 5)
        slab = torch.zeros(2, 77, 45, device='cuda')
        call('vqa_gemm_slabs', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(slab), 77 * 45, 45, 77, 45, 131, 2, F32, 0)
        torch.cuda.synchronize()
        err = float((slab.double().sum(0) - ref).abs().max() / ref.abs().max())
        assert err <= 1e-6


@pytest.mark.parametrize('engine', [4, 3, 2, 0])
@pytest.mark.parametrize('akm,bkm', [(1, 1), (0, 1)])
def test_device_side_row_limit(engine, akm, bkm):
    """A device-side live-row count (sequences sorted by length: step t has m_t live rows).  Rows below the
    limit are exact for every limit; the 256 x 128 engine leaves the rows past it untouched."""
    M, N, K, sk = 700, 392, 2400, 3
    As, lda, Bs, ldb, ref = make(M, N, K, akm, bkm, torch.bfloat16, 3)
    for live in (1, 16, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 320, 333, 512, 699, 700, 900):
        lim = torch.tensor([live], device='cuda', dtype=torch.int32)
        slab = torch.full((sk, M, N), float('nan'), device='cuda')
        call('vqa_gemm_slabs_rows', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(slab), M * N, N, M, N, K, sk, BF16, engine,
             ptr(lim), None)
        torch.cuda.synchronize()
        n = min(live, M)
        got = slab[:, :n].double().sum(0)
        err = float((got - ref[:n]).abs().max() / ref.abs().max())
        assert err <= 2e-5, (engine, akm, bkm, live, err)
        if engine == 4 and akm and bkm and n < M:
            assert bool(torch.isnan(slab[:, n:]).all()), (live, 'rows past the limit were written')


@pytest.mark.parametrize('M,N,K', [(12547, 1200, 2072), (300, 136, 256), (4099, 520, 64), (50176, 1200, 2048)])
@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('engine', [0, 10, 11])
def test_projection_with_fused_row_norm(M, N, K, out_dtype, engine):
    """C = normalize(A) B^T + bias in one call (ring kernel: norms from the A tiles in LDS; small products: norm
    kernel + product), and the inverse norms it stores, against fp64 from the same bf16 operands."""
    As, lda, Bs, ldb, ref = make(M, N, K, 1, 1, torch.bfloat16, M % 97)
    As[5].zero_()                                            # an all-zero row: scale 1e12 times 0
    ref[5].zero_()
    g = torch.Generator(device='cuda').manual_seed(9)
    bias = torch.randn(N, device='cuda', generator=g)
    ldc = (N + 7) // 8 * 8
    C = torch.zeros(M, ldc, device='cuda', dtype=out_dtype)
    inv = torch.full((M,), float('nan'), device='cuda')
    call('vqa_gemm_fused_rownorm', ptr(As), lda, ptr(Bs), ldb, ptr(C), ldc, M, N, K, ptr(bias), ptr(inv), 0,
         BF16 if out_dtype == torch.bfloat16 else F32, BF16, engine)
    torch.cuda.synchronize()
    norm = As[:, :K].double().norm(dim=1)
    inv_ref = 1.0 / norm.clamp_min(1e-12)
    assert float(((inv.double() - inv_ref).abs() / inv_ref).max()) <= 1e-5
    want = ref * inv_ref[:, None] + bias.double()[None, :]
    tol = 1e-2 if out_dtype == torch.bfloat16 else 3e-5
    err = float((C[:, :N].double() - want).abs().max() / want.abs().max())
    assert err <= tol, (M, N, K, err)
    # the separate norm kernel gives the same scales to fp32 rounding
    inv2 = torch.empty_like(inv)
    call('vqa_inv_norm', ptr(As), ptr(inv2), M, K, lda, BF16)
    torch.cuda.synchronize()
    assert float(((inv - inv2).abs() / inv2).max()) <= 1e-5


@pytest.mark.parametrize('akm,bkm', [(1, 1), (0, 0), (1, 0)])
def test_grouped_small_products(akm, bkm):
    """Up to eight independent products in one launch give exactly what the single launches give (64 x 64 tiles)."""
    from vqa_amd import _lib
    shapes = [(256, 1200, 2400, 6), (256, 1208, 2400, 6), (37, 50, 72, 1), (300, 129, 64, 3),
              (3000, 256, 256, 1), (256, 1200, 256, 1), (1200, 2048, 256, 1), (64, 8, 512, 2)]
    ops, problems, singles = [], [], []
    for i, (M, N, K, sk) in enumerate(shapes):
        As, lda, Bs, ldb, ref = make(M, N, K, akm, bkm, torch.bfloat16, 40 + i)
        grouped = torch.full((sk, M, N), float('nan'), device='cuda')
        single = torch.full((sk, M, N), float('nan'), device='cuda')
        call('vqa_gemm_slabs', ptr(As), lda, akm, ptr(Bs), ldb, bkm, ptr(single), M * N, N, M, N, K, sk, BF16, 2)
        problems.append((ptr(As), lda, ptr(Bs), ldb, ptr(grouped), M * N, N, M, N, K, sk))
        ops.append((As, Bs, ref, grouped, single))
    for n in (8, 4, 1, 3):
        for _, _, _, grouped, _ in ops:
            grouped.fill_(float('nan'))
        _lib.gemm_slabs_group(problems[:n], akm, bkm, BF16)
        torch.cuda.synchronize()
        for _, _, ref, grouped, single in ops[:n]:
            assert torch.equal(grouped, single)
            assert float((grouped.double().sum(0) - ref).abs().max() / ref.abs().max()) <= 2e-5
    with pytest.raises(_lib.VqaHipError):
        _lib.gemm_slabs_group(problems + problems[:1], akm, bkm, BF16)


def test_grouped_products_with_column_sums_in_one_launch():
    from vqa_amd import _lib
    M, N, K = 300, 200, 256
    As, lda, Bs, ldb, ref = make(M, N, K, 0, 0, torch.bfloat16, 5)
    slab = torch.full((1, M, N), float('nan'), device='cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    X1 = torch.randn(77, 130, device='cuda', generator=g)
    X2 = torch.randn(256, 40, device='cuda', generator=g).to(torch.bfloat16)
    o1 = torch.zeros(130, device='cuda'); o1b = torch.zeros(130, device='cuda'); o2 = torch.zeros(40, device='cuda')
    _lib.gemm_slabs_group([(ptr(As), lda, ptr(Bs), ldb, ptr(slab), M * N, N, M, N, K, 1)], 0, 0, BF16,
                          colsums=[(X1, 130, 77, 130, F32, o1, o1b, 0.5), (X2, 40, 256, 40, BF16, o2, None)])
    torch.cuda.synchronize()
    assert float((slab[0].double() - ref).abs().max() / ref.abs().max()) <= 2e-5
    r1 = torch.zeros_like(o1); r2 = torch.zeros_like(o2)
    _lib.colsum_multi([(X1, 130, 77, 130, F32, r1, None, 0.5), (X2, 40, 256, 40, BF16, r2, None)])
    torch.cuda.synchronize()
    assert torch.equal(o1, r1) and torch.equal(o1b, r1) and torch.equal(o2, r2)
    assert float((o1.double() - 0.5 * X1.double().sum(0)).abs().max()) <= 1e-4
