"""Child process of test_gpu_two_ranks.py: rank `argv[2]` of a two-rank gloo group whose ranks both compute on cuda:0
(RCCL refuses two ranks on one device).  The collectives run on host copies: gloo's own staging of device tensors
hung in about one run out of three when two processes shared the device (both ranks inside the same all_gather).  Each rank trains on its
half of a batch through GradReducer; the reduced gradient must equal the gradient of the whole batch computed by one
model without a reducer, the mean of the rank losses the whole-batch loss, and after a few steps the weights of the
two ranks must be bit-identical to each other.  Dropout is off (its masks are rank-dependent by design) and the
weight-stationary encoder is off (two processes' launches of it could each hold half of the compute units and wait
for the other half).  Exits 0 on success."""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    port, rank, out_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    faulthandler.dump_traceback_later(45, exit=True)       # a rank that waits for a lost peer says where, and ends
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=2)
    from vqa_amd.model import VQAAttnModel
    from vqa_amd.parallel import GradReducer, connect_collectives
    connect_collectives('cpu', nbytes=1 << 16)

    class Done:
        def wait(self):
            pass

    class HostReducer(GradReducer):
        """GradReducer with each collective done on a host copy of its gradient range (synchronously)."""

        def _launch(self, s, e):
            self.launched.append((s, e))
            host = self._flat_grad[s:e].cpu()               # (waits for the kernels that produced the range)
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self._flat_grad[s:e].copy_(host)
            return Done()

    kw = dict(emb_size=40, question_out=64, visual_out=128, attention_dim=48, fusion_in=64, fusion_hidden=32, dropout=0.0)
    B, R, L, Vq, C = 32, 9, 7, 50, 11
    g = torch.Generator(device='cuda').manual_seed(1)            # the same whole batches on both ranks
    batches = []
    for _ in range(3):
        feats = torch.randn(B, R, 128, device='cuda', generator=g).clamp_min_(0)
        tok = torch.randint(0, Vq + 1, (B, L), device='cuda', generator=g)
        tok[:, 0].clamp_(min=1)
        ans = torch.randint(0, C, (B,), device='cuda', generator=g)
        batches.append((feats, tok, ans))
    half = slice(rank * (B // 2), (rank + 1) * (B // 2))
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        for defer_all in (True, False):
            dp = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=dtype, seed=3, **kw).to(dev).train()
            one = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=dtype, seed=3, **kw).to(dev).train()
            dp.resident_encoder = one.resident_encoder = False
            dp.dropout_rank = rank
            red = HostReducer(defer_all=defer_all)
            assert red.enabled and red.world == 2
            for i in range(4):
                f, t, a = batches[i % 3]
                f = f.to(dtype)
                l_dp = dp.train_step(f[half], t[half], a[half], lr=1e-3, reducer=red)
                dp.sync()
                l_one = one.train_step(f, t, a, lr=1e-3, reducer=None)
                one.sync()
                torch.cuda.synchronize()
                if i == 0:
                    # same weights going in: the all-reduced gradient / world is the whole-batch gradient
                    g_dp = dp._flat_grad.double() / 2
                    g_one = one._flat_grad.double()
                    err = float((g_dp - g_one).abs().max() / g_one.abs().max())
                    assert err <= tol, (str(dtype), defer_all, 'gradient', err)
                    lm = l_dp.detach().double().cpu()
                    dist.all_reduce(lm)
                    assert abs(float(lm) / 2 - float(l_one)) <= tol * max(1.0, abs(float(l_one))), (float(lm) / 2, float(l_one))
            # both ranks applied the same reduced gradients: their weights are bit-identical
            mine = dp._flat.detach().cpu()
            other = [torch.empty_like(mine), torch.empty_like(mine)]
            dist.all_gather(other, mine)
            assert torch.equal(other[0], other[1]), (str(dtype), defer_all, 'ranks diverged')
            # ... and stay close to the one-process run (Adam turns gradient noise at zero crossings into +-lr steps)
            d = (dp._flat - one._flat).abs()
            assert float(d.max()) <= 2 * 4 * 1e-3 + 1e-6
            assert float((d > 1e-4).float().mean()) <= 0.05, float((d > 1e-4).float().mean())
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f'ok{rank}'), 'w').write('ok')
    print(f'two ranks on one device: rank {rank} ok')


if __name__ == '__main__':
    main()
