"""Cost of the bucketed all-reduce path on one GPU (one-rank RCCL group): step time without a reducer, with a
reducer whose collectives are off (hooks only), and with the collectives on (deferred last bucket or not)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from vqa_amd.model import VQAAttnModel
from vqa_amd.parallel import GradReducer
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29535', rank=0, world_size=1, device_id=dev)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).train()
batches = bench.synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
variants = {'none': None, 'hooks_only': GradReducer(force=False), 'rccl_defer_all': GradReducer(force=True),
            'rccl_defer_last': GradReducer(force=True, defer_all=False),
            'rccl_nodefer': GradReducer(force=True, defer_last=False)}
for rep in range(2):
    for name, red in variants.items():
        for i in range(10): m.train_step(*batches[i % 3], reducer=red)
        m.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100): m.train_step(*batches[i % 3], reducer=red)
        m.sync(); torch.cuda.synchronize()
        print(f'{name:14s} {(time.perf_counter() - t0) * 1e4:.1f} us/step', flush=True)
dist.destroy_process_group()
