"""The bucketed gradient all-reduce over the real RCCL backend, in a one-rank process group on the one GPU of
the test machine (two ranks cannot share a device): stream ordering between the compute stream and RCCL's
stream, the deferred last bucket, and tear-down.  The multi-rank arithmetic is covered on CPU with gloo."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_reducer_over_rccl_single_rank():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rccl_single_rank.py')
    r = subprocess.run([sys.executable, '-u', script, str(port)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'rccl single rank ok' in r.stdout
