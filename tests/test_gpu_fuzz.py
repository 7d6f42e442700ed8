"""Random ragged shapes through both fp32-class kernel sets (tools/shape_fuzz.py): the split-precision kernels and the exact-fp32
kernels share no GEMM, attention or positional-conv code, so agreement of every stage tap and hidden state to 5e-6 on shapes nobody
picked by hand (T = 1 ... 6 250, B = 1 ... 24, lengths on and off every tile boundary) is a check on both."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_agree_between_the_two_kernel_sets():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "shape_fuzz.py"), "16", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst over 16 cases" in r.stdout


def test_random_packs_agree_with_the_one_batch_forwards():
    """tools/pack_fuzz.py: random packs (1-12 batches of 1-4 clips, 400 samples to 20 s, masks / none / mixed, host- and
    device-packed, both fp32-class modes) against the forwards of their batches alone: 5e-6 per clip, padded frames included."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_fuzz.py"), "24", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "worst over 24 cases" in r.stdout
