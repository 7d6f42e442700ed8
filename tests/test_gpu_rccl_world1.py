"""The RCCL branch on the one-GPU box: bench.py under `python -m torch.distributed.run
--nproc-per-node 1` with --force-collective creates the "nccl" (= RCCL) process group of one rank and issues the per-step
`all_gather_into_tensor` on the device embeddings through dp.OverlappedGather -- asynchronously, on RCCL's stream, next to the
encoder's own two streams -- exactly what every rank does at N > 1; bench.py itself asserts that the gathered tensor equals
the local one.  The launcher is a child process of the test runner and only ITS child touches the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_with_a_real_rccl_all_gather_at_world_size_one():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--clip-seconds", "10", "--batch", "8", "--force-collective", "--no-alt", "--no-two-streams", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert "forced in a process group of one rank and checked" in d["config"]["collective"]
    assert d["config"]["workload"].startswith("SpeechT5-base speech encoder, synthetic 16 kHz 10 s clips, batch 8")
