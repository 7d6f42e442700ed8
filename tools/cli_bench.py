#!/usr/bin/env python3
"""End-to-end rate of the extraction CLI (synthetic 30 s clips -> host batching -> encoder -> asynchronous sink) with the
host work inline (as the reference does it) and on the loader threads."""
import importlib, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
extract = importlib.import_module("loco-asr_amd.extract")
N = 384
for th in (0, 8):
    out = tempfile.mkdtemp(prefix="cli_bench_")
    t0 = time.perf_counter()
    extract.main(["-m", "audio", "-s", "devel", "--synthetic", str(N), "--synthetic-seconds", "30", "--batch-size", "32", "--random-init",
                  "--format", "npy", "--out", out, "--loader-threads", str(th)])
    dt = time.perf_counter() - t0
    print(f"CLI loader-threads={th}: {N} clips x <=30 s in {dt:.2f} s wall (incl. weight generation + load) = {N/dt:.1f} clips/s", flush=True)
    shutil.rmtree(out, ignore_errors=True)
