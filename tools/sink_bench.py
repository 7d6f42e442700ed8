#!/usr/bin/env python3
"""Can the writer ("next" row f-2) keep up with the encoder?  Feeds the sink batches of [32, 1499, 768] embeddings that are
already on the GPU (D2H on a side stream into pinned memory, per-utterance pickle or .npy by worker threads) and reports
GB/s and the frames/s that corresponds to; the encoder produces 0.86 M frames/s = 2.65 GB/s per GPU."""
import importlib, os, shutil, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sink_mod = importlib.import_module("loco-asr_amd.sink")

B, T = 32, 1499
emb = torch.randn(B, T, 768, device="cuda")
tg = [np.eye(101, dtype=np.int64)[i % 101] for i in range(B)]
for fmt in ("pickle", "npy"):
    for workers in (4, 12):
        root = tempfile.mkdtemp(prefix="sink_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
        nb = 12
        t0 = time.perf_counter()
        with sink_mod.EmbeddingSink(root, "devel", "audio", fmt, workers=workers, max_pending=4) as sink:
            for k in range(nb):
                sink.submit([f"u{k:03d}_{i:02d}" for i in range(B)], emb, tg)
        dt = time.perf_counter() - t0
        gb = nb * emb.numel() * 4 / 1e9
        print(f"sink {fmt:6s} workers={workers:2d}: {gb:.2f} GB in {dt:.2f} s = {gb/dt:.2f} GB/s = {nb*B*T/dt:,.0f} frames/s "
              f"({nb*B/dt:.0f} utterance files/s)", flush=True)
        shutil.rmtree(root, ignore_errors=True)
