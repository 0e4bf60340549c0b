"""One-GPU rehearsal of a data-parallel step's compute-unit contention: the gradient collectives are replaced by a
stand-in kernel that HOLDS compute units (one 256-thread workgroup of 96 registers and 16 KB of LDS per 'channel') for
as long as an all-reduce of that size would take at an assumed bus bandwidth, on a second stream with the same
cross-stream ordering as the RCCL path.  It moves no data and says nothing about xGMI: it shows what the step's own
kernels cost beside a resident collective, and how much of the collective's duration the step hides.
Development build: make dev; VQA_HIP_LIB=$PWD/vqa_amd/libvqa_hip_dev.so python tools/dp_rehearsal.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from vqa_amd import _lib, parallel
from vqa_amd._lib import ptr
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29577', rank=0, world_size=1)
lib = _lib.load()
lib.vqa_dev_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sink = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
cfg = {'cus': 48, 'busbw': 250.0, 'ranks': 8}


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def fake_all_reduce(tensor, op=None, group=None, async_op=False):
    nbytes = tensor.numel() * tensor.element_size()
    n = cfg['ranks']
    usec = int(2.0 * (n - 1) / n * nbytes / (cfg['busbw'] * 1e3))          # ring-equivalent bytes / bus bandwidth
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        lib.vqa_dev_occupy(cfg['cus'], max(usec, 1), ptr(sink), ctypes.c_void_p(side.cuda_stream))
        ev = torch.cuda.Event()
        ev.record(side)
    return _Work(ev)


parallel.dist.all_reduce = fake_all_reduce
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
batches = synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)


def run(reducer_on, collective_cus, steps=150):
    model = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16, seed=0).to(dev).train()
    red = None
    if reducer_on:
        red = parallel.GradReducer(force=True)
        red._in_order = True                       # one side stream: the youngest handle covers the older ones
        red.collective_cus = collective_cus
    for i in range(20):
        model.train_step(*batches[i % 3], lr=1e-5, reducer=red)
    model.sync(check=False)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(steps):
        model.train_step(*batches[i % 3], lr=1e-5, reducer=red)
    model.sync(check=False)
    e.record(); torch.cuda.synchronize()
    model.check_status()
    return s.elapsed_time(e) / steps * 1e3


base = run(False, 0)
print(f'no reducer: {base:.1f} us per step', flush=True)
for cus in [int(a) for a in sys.argv[1:]] or [32, 48]:
    for busbw in (150.0, 250.0, 350.0):
        cfg.update(cus=cus, busbw=busbw)
        t5 = run(True, 56)
        t6 = run(True, 0)
        c1 = 2 * 7 / 8 * 37.2e6 / (busbw * 1e3)
        c2 = 2 * 7 / 8 * 9.8e6 / (busbw * 1e3)
        print(f'stand-in holds {cus} CUs, {busbw:.0f} GB/s bus bandwidth (collectives {c1:.0f} + {c2:.0f} us): '
              f'{t5:.1f} us per step with the weight gradient sized to fit ({base / t5:.3f} of the one-GPU rate), '
              f'{t6:.1f} us with 6 K ranges', flush=True)
dist.destroy_process_group()
