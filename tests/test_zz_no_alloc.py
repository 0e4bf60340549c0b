"""vqa_grid_mean_static (the entry point that allocates nothing: its split partial sums live in a static device
array) against vqa_grid_mean with a caller-provided workspace, through the C ABI: bit-identical means on two calls
in a row, the second on a smaller input that reuses the same static array, plus the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

STATIC_FLOATS = 1 << 18          # VQA_GRID_MEAN_STATIC_FLOATS of include/vqa_hip.h


def _both(x, layout, B, C, HW, dt):
    from vqa_amd import _lib
    from vqa_amd._lib import call, ptr
    _lib.bind_stream()
    nsplit = call('vqa_grid_mean_splits', layout, B, C, HW)
    ld, col0 = C + 24, 8
    ref = torch.full((B, ld), -3.0, device='cuda')
    got = torch.full((B, ld), -3.0, device='cuda')
    part = torch.empty(max(nsplit * B * C, 1), device='cuda')
    call('vqa_grid_mean', ptr(x), layout, B, C, HW, ptr(ref), ld, col0, ptr(part) if nsplit > 1 else None, nsplit, dt)
    call('vqa_grid_mean_static', ptr(x), layout, B, C, HW, ptr(got), ld, col0, dt)
    torch.cuda.synchronize()
    return ref, got, nsplit, col0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_static_scratch_grid_mean_matches_workspace_form(dtype):
    from vqa_amd import _lib
    dt = _lib.dtype_code(dtype)
    g = torch.Generator().manual_seed(7)
    # (B, HW, C) channels-last maps whose grid is split (partial sums go through the scratch): the second call is
    # smaller and runs on the array the first one left full; a third, larger again, follows
    for B, C, HW in ((8, 512, 4096), (3, 256, 1024), (5, 1024, 2304)):
        x = torch.relu(torch.randn(B, HW, C, generator=g)).to(dtype).cuda()
        ref, got, nsplit, col0 = _both(x, 1, B, C, HW, dt)
        assert nsplit > 1 and nsplit * B * C <= STATIC_FLOATS
        assert torch.equal(ref, got)
        # and against the plain mean (fp32 accumulation of the stored values)
        want = x.float().mean(1).cpu()
        assert float((got[:, col0:col0 + C].cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())
        assert float(got[:, :col0].min()) == -3.0 and float(got[:, col0 + C:].max()) == -3.0
    # layouts / sizes that need no partial sums take the same entry point
    x = torch.relu(torch.randn(4, 64, 49, generator=g)).to(dtype).cuda()
    ref, got, nsplit, _ = _both(x, 0, 4, 64, 49, dt)
    assert nsplit == 1 and torch.equal(ref, got)


def test_static_entry_point_allocates_nothing():
    """Device memory in use (hipMemGetInfo through torch) is the same before and after calls of the static form:
    the first call on the device (which looks the scratch address up) and a later one."""
    from vqa_amd import _lib
    from vqa_amd._lib import call, ptr
    B, C, HW = 6, 512, 4096
    x = torch.relu(torch.randn(B, HW, C)).to(torch.bfloat16).cuda()
    out = torch.zeros(B, C, device='cuda')
    _lib.load()
    _lib.bind_stream()
    # (the library's code object, static array included, is placed on the GPU when the process first uses the
    #  library: run another of its entry points first so that this one-time load is not counted)
    part = torch.empty(call('vqa_grid_mean_splits', 1, B, C, HW) * B * C, device='cuda')
    call('vqa_grid_mean', ptr(x), 1, B, C, HW, ptr(out), C, 0, ptr(part), call('vqa_grid_mean_splits', 1, B, C, HW), _lib.BF16)
    torch.cuda.synchronize()
    for _ in range(2):
        free0, _total = torch.cuda.mem_get_info()
        call('vqa_grid_mean_static', ptr(x), 1, B, C, HW, ptr(out), C, 0, _lib.BF16)
        torch.cuda.synchronize()
        free1, _total = torch.cuda.mem_get_info()
        assert free1 == free0, (free0, free1)
    # a stream of torch's on the current device is accepted
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        _lib.bind_stream()
        call('vqa_grid_mean_static', ptr(x), 1, B, C, HW, ptr(out), C, 0, _lib.BF16)
    side.synchronize()
    _lib.bind_stream()


def test_static_scratch_limit_is_an_error():
    from vqa_amd import _lib
    from vqa_amd._lib import call, ptr
    B, C, HW = 48, 512, 4096                  # 16 splits x 48 x 512 floats = 384 Ki > 256 Ki
    nsplit = call('vqa_grid_mean_splits', 1, B, C, HW)
    assert nsplit * B * C > STATIC_FLOATS
    x = torch.zeros(8, device='cuda')         # never read: the call is rejected before any launch
    out = torch.zeros(B, C, device='cuda')
    _lib.bind_stream()
    with pytest.raises(_lib.VqaHipError, match='static scratch'):
        call('vqa_grid_mean_static', ptr(x), 1, B, C, HW, ptr(out), C, 0, _lib.F32)
