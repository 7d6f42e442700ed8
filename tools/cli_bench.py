#!/usr/bin/env python3
"""End-to-end rate of the extraction CLI at the REFERENCE'S operating point: batch_size = 2, corpus order
(/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:67-68), on a SLURP-like ragged synthetic corpus
(2-6 s utterances), through host batching -> encoder -> asynchronous sink -> per-utterance pickles, with K = 1, 2, 4, 8 batches in
flight (extract.py --inflight).  The batches -- and therefore the results -- are identical at every K; this script checks that
byte for byte on every file before it prints a rate.

    python3 tools/cli_bench.py [N utterances, default 2000] [--big]     (--big adds the 30 s x batch 32 throughput case)
"""
import hashlib, importlib, json, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
extract = importlib.import_module("loco-asr_amd.extract")
la = importlib.import_module("loco-asr_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
lens = la.synth.mixed_lengths(N, 6 * 16000, min_fraction=2.0 / 6.0)
frames = sum(la.synth.conv_out_length(max(lens[i:i + 2])) * len(lens[i:i + 2]) for i in range(0, N, 2))  # padded frames, as pickled
valid = sum(la.synth.conv_out_length(n) for n in lens)
print(f"corpus: {N} synthetic utterances of 2-6 s (mean {sum(lens) / N / 16000:.2f} s), reference batches of 2 in corpus order: "
      f"{frames} frames encoded ({valid} valid)", flush=True)
base = ["-m", "audio", "-s", "devel", "--synthetic", str(N), "--synthetic-seconds", "6", "--synthetic-min-seconds", "2", "--random-init"]
digests, results = {}, []
extract.main(base[:4] + ["--synthetic", "16", "--synthetic-seconds", "6", "--random-init", "--out", tempfile.mkdtemp(prefix="cli_warm_")])  # warm-up: library, allocator
for k in (1, 2, 4, 8):
    out = tempfile.mkdtemp(prefix=f"cli_bench_k{k}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    st = extract.main(base + ["--out", out, "--inflight", str(k)])
    dt = st["seconds"]  # the loop alone: synthesis -> batching -> H2D -> encoder -> D2H -> pickles closed (no model load)
    assert st["frames"] == frames and st["utterances"] == N
    folder = os.path.join(out, "devel", "audio")
    h = hashlib.sha256()
    names = sorted(os.listdir(folder))
    for n in names:
        h.update(n.encode()); h.update(open(os.path.join(folder, n), "rb").read())
    digests[k] = (len(names), h.hexdigest())
    shutil.rmtree(out, ignore_errors=True)
    r = dict(inflight=k, seconds=round(dt, 3), utterances_per_s=round(N / dt, 1), frames_per_s=round(frames / dt, 1), files=len(names), sha256=h.hexdigest()[:16])
    results.append(r)
    print(f"--inflight {k}: {N} utterances in {dt:.2f} s (encode loop: host batching, H2D, encoder, D2H, pickles written) = {N / dt:.1f} utterances/s, "
          f"{frames / dt:,.0f} frames/s; {len(names)} pickles, sha256 {h.hexdigest()[:16]}", flush=True)
assert len({d for d in digests.values()}) == 1, f"pickles differ between --inflight values: {digests}"
print("all --inflight values wrote byte-identical pickles")
print(json.dumps({"corpus": {"utterances": N, "frames_padded": frames, "frames_valid": valid}, "runs": results}))
if "--big" in sys.argv:
    for th in (0, 8):
        out = tempfile.mkdtemp(prefix="cli_bench_")
        t0 = time.perf_counter()
        extract.main(["-m", "audio", "-s", "devel", "--synthetic", "384", "--synthetic-seconds", "30", "--batch-size", "32", "--random-init",
                      "--format", "npy", "--out", out, "--loader-threads", str(th), "--inflight", "1"])
        dt = time.perf_counter() - t0
        print(f"CLI 30 s x batch 32, loader-threads={th}: 384 clips in {dt:.2f} s wall = {384 / dt:.1f} clips/s", flush=True)
        shutil.rmtree(out, ignore_errors=True)
