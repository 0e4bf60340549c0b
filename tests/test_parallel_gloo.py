"""Data-parallel path on CPU: 2 gloo ranks.  (i) bucketed gradient all-reduce through GradReducer
equals the gradient of the global-batch mean loss; (ii) summed integer accuracy counters give the
same OA / AA as one rank on the concatenated data."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vqa_ref

KW = dict(vq=30, C=11, E=12, Hq=24, Dv=32, A=16, F=16, Hf=8)


class _FakeModel:
    """Stands in for the module's flat gradient storage (the kernels need a GPU; the reducer does not)."""

    def __init__(self, grads, order, cuts):
        sizes = [grads[k].numel() for k in order]
        self._flat_grad = torch.cat([grads[k].reshape(-1) for k in order])
        offs = np.cumsum([0] + sizes)
        self.buckets = [(int(offs[a]), int(offs[b])) for a, b in zip([0] + cuts, cuts + [len(order)])]
        self.views = {k: self._flat_grad[offs[i]:offs[i + 1]].view(grads[k].shape) for i, k in enumerate(order)}


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vqa_amd.parallel import GradReducer, connect_collectives, reduce_counters
    connect_collectives('cpu', nbytes=4096)      # the warm-up collectives every rank issues after joining the group
    p = vqa_ref.init_params(seed=0, **KW)
    full = vqa_ref.synthetic_batch(8, 9, KW['Dv'], 7, KW['vq'], KW['C'], seed=4)
    sl = slice(rank * 4, rank * 4 + 4)
    _, _, g = vqa_ref.loss_and_grads(p, full['feats'][sl], full['tokens'][sl], full['answers'][sl])
    order = list(g)
    fm = _FakeModel(g, order, cuts=[4, 8, len(order) - 1])
    red = GradReducer()
    assert red.enabled and red.world == world
    red.begin(fm)
    for i in range(len(fm.buckets) - 1):       # the last bucket is left to finish()
        red.bucket_ready(i)
    scale = red.finish()
    _, _, gfull = vqa_ref.loss_and_grads(p, full['feats'], full['tokens'], full['answers'])
    for k in order:
        assert torch.allclose(fm.views[k] * scale, gfull[k], rtol=1e-4, atol=1e-7), k
    # deferred last bucket: finish(defer_last=True) leaves exactly that all-reduce in flight and
    # finish_last() completes it; the reduced values are the same
    fm2 = _FakeModel(g, order, cuts=[4, 8, len(order) - 1])
    red.begin(fm2)
    for i in range(len(fm2.buckets)):
        red.bucket_ready(i)
    assert red.finish(defer_last=True) == scale
    assert red._handles == [] and red._last_handle is not None
    red.finish_last()
    assert red._last_handle is None
    assert torch.equal(fm2._flat_grad, fm._flat_grad)
    # the small buckets of this toy model were held back and reduced in one collective; the last went alone
    assert red.launched == [(fm2.buckets[0][0], fm2.buckets[-2][1]), fm2.buckets[-1]]
    # a byte threshold of zero reduces every bucket on its own; signalling out of order changes nothing
    fm3 = _FakeModel(g, order, cuts=[4, 8, len(order) - 1])
    red0 = GradReducer(bucket_bytes=0)
    red0.begin(fm3)
    for i in (2, 0, 3, 1):
        red0.bucket_ready(i)
    red0.finish()
    assert red0.launched == list(fm3.buckets)
    assert torch.equal(fm3._flat_grad, fm._flat_grad)
    # a model with several attention hops: everything but the last bucket goes out as one collective even with a
    # small byte threshold (nothing is launched while backward still has its big one-workgroup-per-sample kernels ahead)
    fm4 = _FakeModel(g, order, cuts=[4, 8, len(order) - 1])
    fm4.hops = 2
    red1 = GradReducer(bucket_bytes=1)
    red1.begin(fm4)
    for i in range(len(fm4.buckets)):
        red1.bucket_ready(i)
        if i < len(fm4.buckets) - 2:
            assert red1.launched == []
    red1.finish()
    assert red1.launched == [(fm4.buckets[0][0], fm4.buckets[-2][1]), fm4.buckets[-1]]
    assert torch.equal(fm4._flat_grad, fm._flat_grad)
    # accuracy counters: integer sums
    logits = vqa_ref.forward(p, full['feats'][sl], full['tokens'][sl])
    pred = vqa_ref.argmax_first(logits.numpy())
    c, r = vqa_ref.accuracy_counts(pred, full['answers'][sl].numpy(), full['types'][sl].numpy())
    counters = reduce_counters(torch.from_numpy(np.concatenate([c, r])))
    lf = vqa_ref.forward(p, full['feats'], full['tokens'])
    cf, rf = vqa_ref.accuracy_counts(vqa_ref.argmax_first(lf.numpy()), full['answers'].numpy(), full['types'].numpy())
    assert np.array_equal(counters.numpy(), np.concatenate([cf, rf]))
    open(os.path.join(out_dir, f'ok{rank}'), 'w').write('ok')
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_counters(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_collective_cus_follow_the_channel_cap(monkeypatch):
    """The compute units left to the collectives' kernels = RCCL's channel cap + a margin of 8: the drivers' cap of 48 by
    default, the user's NCCL_MAX_NCHANNELS when set, at most half of the chip."""
    from vqa_amd.parallel import GradReducer, reserve_compute_units_for_resident_kernels
    monkeypatch.delenv('NCCL_MAX_NCHANNELS', raising=False)
    assert GradReducer().collective_cus == 56
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '32')
    reserve_compute_units_for_resident_kernels()             # a user's setting wins
    assert os.environ['NCCL_MAX_NCHANNELS'] == '32'
    assert GradReducer().collective_cus == 40
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '200')
    assert GradReducer().collective_cus == 128
