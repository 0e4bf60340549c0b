"""Data parallelism across two processes on the real kernels: two gloo ranks that both compute on the one GPU of the
test machine (tests/gloo_two_ranks_gpu.py).  The reduced gradient equals the whole-batch gradient, the ranks' weights
stay bit-identical, with and without the deferred update, fp32 and bf16."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_gloo_ranks_on_one_device(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gloo_two_ranks_gpu.py')
    procs = [subprocess.Popen([sys.executable, '-u', script, str(port), str(r), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=70)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append('(killed after 70 s)\n' + p.communicate()[0])
    report = '\n'.join(f'--- rank {r} (exit {p.returncode}):\n{o[-3000:]}' for r, (p, o) in enumerate(zip(procs, outs)))
    assert all(p.returncode == 0 for p in procs), report
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
