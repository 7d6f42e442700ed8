#!/usr/bin/env python3
"""Measure the intent head ("next" row f-1, BASELINE configs[4]/SURVEY config 5): forward and one training step
(loss + gradients + Adam) on [16, T, 768] embeddings already in HBM, for the three pooling methods.  The head is HBM-bound:
the compulsory traffic is one read of x for the forward and two for a training step (pooling forward + its backward)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
la = importlib.import_module("loco-asr_amd")

for B, T in ((16, 281), (16, 1499), (64, 1499), (4, 29999)):
    x = torch.randn(B, T, 768, device="cuda")
    tgt = torch.zeros(B, 101, device="cuda")
    tgt[torch.arange(B), torch.arange(B) % 101] = 1
    xb = x.numel() * 4
    for method in ("average", "max", "attention"):
        head = la.IntentClassifierMI355X(method).to("cuda")
        for _ in range(3):
            head(x); head.train_step(x, tgt)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reps = 20
        e0.record()
        for _ in range(reps):
            head(x)
        e1.record()
        for _ in range(reps):
            head.train_step(x, tgt)
        e2.record(); torch.cuda.synchronize()
        tf, tt = e0.elapsed_time(e1) / reps, e1.elapsed_time(e2) / reps
        print(f"head [{B:3d},{T:5d},768] {method:9s}: forward {tf*1e3:7.1f} us ({xb/tf/1e6:6.0f} GB/s of x)   "
              f"train step {tt*1e3:7.1f} us ({2*xb/tt/1e6:6.0f} GB/s of 2x)   {B*T/tt*1e3:12.0f} frames/s trained", flush=True)
