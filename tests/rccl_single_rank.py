"""Child process of test_gpu_rccl.py: a one-rank RCCL ("nccl" backend) process group on cuda:0.  Trains two
equal models for a few steps, one through the bucketed GradReducer with the collectives forced on (sum over one
rank = identity, 1/world = 1), one without a reducer; parameters must stay bit-identical, with and without the
deferred last bucket.  Exits 0 on success."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    port = sys.argv[1]
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
    from vqa_amd.model import VQAAttnModel
    from vqa_amd.parallel import GradReducer, reduce_counters
    kw = dict(emb_size=40, question_out=64, visual_out=128, attention_dim=48, fusion_in=64, fusion_hidden=32)
    g = torch.Generator(device='cuda').manual_seed(1)
    B, R, L, Vq, C = 24, 9, 7, 50, 11
    batches = []
    for _ in range(3):
        feats = torch.randn(B, R, 128, device='cuda', generator=g).clamp_min_(0).to(torch.bfloat16)
        tok = torch.randint(0, Vq + 1, (B, L), device='cuda', generator=g)
        ans = torch.randint(0, C, (B,), device='cuda', generator=g)
        batches.append((feats, tok, ans))
    for defer, whole in ((True, True), (True, False), (False, False)):
        ma = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, seed=3, **kw).to(dev).train()
        mb = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, seed=3, **kw).to(dev).train()
        red = GradReducer(defer_last=defer, defer_all=whole, force=True)
        assert red.enabled and red.world == 1
        for i in range(6):
            f, t, a = batches[i % 3]
            la = ma.train_step(f, t, a, lr=1e-3, reducer=red)
            lb = mb.train_step(f, t, a, lr=1e-3, reducer=None)
            assert float(la) == float(lb), (defer, i, float(la), float(lb))
        ma.sync(); mb.sync()
        torch.cuda.synchronize()
        assert torch.equal(ma._flat, mb._flat), f'parameters diverged (defer_last={defer}, defer_all={whole})'
    c = torch.tensor([3, 4, 5], device='cuda')
    assert reduce_counters(c.clone()).tolist() == [3, 4, 5]
    t = torch.ones(1 << 20, device='cuda')
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t.sum()) == float(1 << 20)
    dist.destroy_process_group()
    # the train driver as torch.distributed.run starts it (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment):
    # it creates the RCCL group itself, trains one epoch with the collectives forced on, writes the checkpoints from
    # rank 0 and tears the group down again
    import tempfile
    from vqa_amd import train as drv
    os.environ.update({'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1',
                       'MASTER_PORT': str(int(port) + 1)})
    with tempfile.TemporaryDirectory() as out:
        hist = drv.main(['--model', 'VQAAttnModel', '--dataset', 'SYNTH', '--epochs', '2', '--bs', '16', '--lr', '0.001',
                         '--synth-batches', '3', '--force-collectives', '--out', out],
                        emb_size=40, question_out=64, visual_out=2048, attention_dim=48, fusion_in=64, fusion_hidden=32)
        assert len(hist) == 2 and hist[1]['train_loss'] == hist[1]['train_loss']
        assert sorted(os.listdir(out)) == ['ModelBestRSVQA_VQAAttnModel.pth', 'ModelRunningRSVQA_VQAAttnModel.pth',
                                           'TrainState_VQAAttnModel.pth']
    assert not dist.is_initialized()
    print('rccl single rank ok')


if __name__ == '__main__':
    main()
