"""The N > 1 path on CPU: two gloo ranks shard a ragged list of clips, encode their shards and gather.
The encode function here is the CPU oracle (tests may use it as the checker's stand-in for the GPU forward);
what is under test is the sharding, the padding to the global T_max, the ordering and the single large
all_gather_into_tensor -- the code bench.py and extract.py run over RCCL."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

LENGTHS = [16000, 9000, 12000, 4000, 16000]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    la = importlib.import_module("loco-asr_amd")
    dp = importlib.import_module("loco-asr_amd.dp")
    import speecht5_oracle as oracle
    sd = la.synth.encoder_state_dict(0, layers=1)
    clips = [la.synth.clip(i, n) for i, n in enumerate(LENGTHS)]
    fe = la.SpeechT5FeatureExtractorMI355X()

    def make_batch(cs):
        b = fe(audio=cs, sampling_rate=16000)
        return b["input_values"], b["attention_mask"]

    def encode(x, m):
        return oracle.encode(x, m, sd)

    res = dp.encode_sharded(encode, clips, make_batch, torch.device("cpu"), max_batch=2)
    assert all(r is not None for r in res)
    # equal-shape fast path used by bench.py
    loc = torch.full((2, 3, 4), float(rank))
    g = dp.all_gather_embeddings(loc)
    assert g.shape == (2 * world, 3, 4) and g[0, 0, 0] == 0 and g[-1, 0, 0] == world - 1
    torch.save([r.clone() for r in res], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_encode_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert len(r0) == len(r1) == len(LENGTHS)
    for a, b in zip(r0, r1):  # every rank ends with the same gathered result
        assert torch.equal(a, b)
    # and each clip equals what its owner batch produces in a single process
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    la = importlib.import_module("loco-asr_amd")
    dp = importlib.import_module("loco-asr_amd.dp")
    import speecht5_oracle as oracle
    sd = la.synth.encoder_state_dict(0, layers=1)
    clips = [la.synth.clip(i, n) for i, n in enumerate(LENGTHS)]
    fe = la.SpeechT5FeatureExtractorMI355X()
    for r in range(world):
        mine = dp.shard_units(LENGTHS, world, r)
        for c0 in range(0, len(mine), 2):
            ids = mine[c0:c0 + 2]
            b = fe(audio=[clips[i] for i in ids], sampling_rate=16000)
            ref = oracle.encode(b["input_values"], b["attention_mask"], sd)
            for row, gid in enumerate(ids):
                assert r0[gid].shape == ref[row].shape
                assert torch.allclose(r0[gid], ref[row], atol=1e-5)


def _overlap_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dp = importlib.import_module("loco-asr_amd.dp")
    g = dp.OverlappedGather()
    for step in range(3):  # what bench.py does: submit per step, finish at the end
        g.submit(torch.full((2, 3, 4), float(10 * step + rank)))
    out = g.finish()
    assert out.shape == (2 * world, 3, 4)
    for r in range(world):
        assert float(out[2 * r, 0, 0]) == 20 + r
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_two_ranks():
    mp.spawn(_overlap_worker, args=(2, _free_port()), nprocs=2, join=True)
