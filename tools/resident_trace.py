"""Where one launch of vqa_lstm_resident spends its time: wall-clock stamps of two workgroups from a development
build (make dev; VQA_HIP_LIB=vqa_amd/libvqa_hip_dev.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqa_amd import _lib
from vqa_amd.model import VQAAttnModel
from bench import synth_batches

B, L, Vq, C = int(os.environ.get('B', 256)), 14, 10000, 3000
dev = torch.device('cuda', 0)
batches = synth_batches(3, B, 4, 2048, L, Vq, C, dev, 42, torch.bfloat16)
m = VQAAttnModel(['w'] * Vq, ['a'] * C, None, dtype=torch.bfloat16).to(dev).eval()
for i in range(4):
    m.encode_question(batches[i % 3][1])
torch.cuda.synchronize()
ws = m._enc_workspace(B, L, dev)
print('live rows per step', ws['m_t'].tolist())
lib = _lib.load()
buf = (ctypes.c_ulonglong * (2 * 16 * 12))()
lib.vqa_dev_resident_trace.argtypes = [ctypes.c_void_p]
assert lib.vqa_dev_resident_trace(buf) == 0
for wg in range(2):
    print('workgroup', 'first' if wg == 0 else 'middle')
    e = [buf[(wg * 16 + 15) * 12 + k] for k in range(4)]
    print('  entry -> %.2f us -> sequences packed %.2f us -> steps 0/1 stored %.2f us (then the weight slice is requested)' % tuple(0.01 * (e[k + 1] - e[k]) for k in range(3)))
    base = e[0]
    for st in range(14):
        row = [buf[(wg * 16 + st) * 12 + k] for k in range(12)]
        if row[0] == 0:
            continue
        t0 = row[0]
        txt = f'  step {st:2d} starts {0.01 * (t0 - base):7.2f} us: barrier {0.01 * (row[1] - t0):6.2f}'
        prev = row[1]
        for ps in range(5):
            a, b = row[2 + 2 * ps], row[3 + 2 * ps]
            if a < prev:
                break
            txt += f' | mfma {0.01 * (a - prev):5.2f} cells {0.01 * (b - a):5.2f}'
            prev = b
        print(txt)
