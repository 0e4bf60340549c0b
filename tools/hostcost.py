"""Host-side cost of one train_step (launch path only) vs device time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqa_amd.model import VQAAttnModel
dev = torch.device('cuda', 0)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).train()
batches = bench.synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
red = None
if len(sys.argv) > 1 and sys.argv[1] == 'rccl':
    import torch.distributed as dist
    from vqa_amd.parallel import GradReducer
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29534', rank=0, world_size=1, device_id=dev)
    red = GradReducer(force=True)
_ts = m.train_step
m.train_step = lambda f, t, a: _ts(f, t, a, reducer=red)
for i in range(5): m.train_step(*batches[i % 3])
torch.cuda.synchronize()
# host time: enqueue 10 steps without waiting
t0 = time.perf_counter()
for i in range(10): m.train_step(*batches[i % 3])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue per step {(t1 - t0) / 10 * 1e3:.3f} ms; total per step {(t2 - t0) / 10 * 1e3:.3f} ms')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(10): m.train_step(*batches[i % 3])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
