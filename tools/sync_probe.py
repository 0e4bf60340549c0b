"""What does a cross-stream edge cost the compute stream?  The bucketed reducer's dependency pattern with the
collective replaced by (a) nothing, (b) a tiny kernel on a side stream, against the one-rank RCCL all-reduce."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from vqa_amd.model import VQAAttnModel
from vqa_amd.parallel import GradReducer
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29536', rank=0, world_size=1, device_id=dev)
B, R, Dv, L, Vq, C = 256, 196, 2048, 14, 10000, 3000
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).train()
batches = bench.synth_batches(3, B, R, Dv, L, Vq, C, dev, 42, torch.bfloat16)
side = torch.cuda.Stream()
scratch = torch.zeros(64, device=dev)


class Handle:
    def __init__(self, ev): self.ev = ev
    def wait(self): torch.cuda.current_stream().wait_event(self.ev)


class FakeReducer(GradReducer):
    """same bucketing and deferral; the collective is an event edge to a side stream and back"""
    def __init__(self, kernel, **kw):
        super().__init__(force=True, **kw)
        self.kernel = kernel
    def _launch(self, s, e):
        self.launched.append((s, e))
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        if self.kernel:
            with torch.cuda.stream(side):
                scratch.add_(1.0)
        ev2 = torch.cuda.Event(); ev2.record(side)
        return Handle(ev2)


import ctypes
hip = ctypes.CDLL('libamdhip64.so')
hipEventDisableTiming, hipEventDisableSystemFence = 0x2, 0x20000000


class NoFenceEdge:
    """record on one stream / wait on another through an event created without the system-scope fence"""
    def __init__(self, n=8):
        self.evs = []
        for _ in range(n):
            ev = ctypes.c_void_p()
            rc = hip.hipEventCreateWithFlags(ctypes.byref(ev), hipEventDisableTiming | hipEventDisableSystemFence)
            assert rc == 0, rc
            self.evs.append(ev)
        self.i = 0
    def edge(self, src, dst):
        ev = self.evs[self.i % len(self.evs)]; self.i += 1
        assert hip.hipEventRecord(ev, ctypes.c_void_p(src.cuda_stream)) == 0
        assert hip.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), ev, 0) == 0


class NFHandle:
    def __init__(self, edges): self.edges = edges
    def wait(self): self.edges.edge(side, torch.cuda.current_stream())


class NoFenceReducer(GradReducer):
    def __init__(self, rccl, **kw):
        super().__init__(force=True, **kw)
        self.edges = NoFenceEdge()
        self.rccl = rccl
    def _launch(self, s, e):
        self.launched.append((s, e))
        self.edges.edge(torch.cuda.current_stream(), side)
        if self.rccl:
            with torch.cuda.stream(side):        # c10d's own stream bookkeeping now hangs off the side stream
                self.h = dist.all_reduce(self._flat_grad[s:e], op=dist.ReduceOp.SUM, async_op=True)
                self.h.wait()                    # side stream waits for the collective; the compute stream does not
        return NFHandle(self.edges)


variants = {'edge_nofence': NoFenceReducer(False), 'rccl_nofence': NoFenceReducer(True), 'none': None, 'rccl': GradReducer(force=True), 'edge_only': FakeReducer(False), 'edge_kernel': FakeReducer(True)}
for rep in range(2):
    for name, red in variants.items():
        for i in range(10): m.train_step(*batches[i % 3], reducer=red)
        m.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100): m.train_step(*batches[i % 3], reducer=red)
        m.sync(); torch.cuda.synchronize()
        print(f'{name:12s} {(time.perf_counter() - t0) * 1e4:.1f} us/step', flush=True)
dist.destroy_process_group()
