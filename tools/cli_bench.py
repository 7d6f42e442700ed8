#!/usr/bin/env python3
"""End-to-end rate of the extraction CLI at the REFERENCE'S operating point: batch_size = 2, corpus order
(/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:67-68), on a SLURP-LAYOUT corpus written to /dev/shm first
(dataset/slurp/devel.jsonl + audio/slurp_real/*.wav: 2-6 s utterances, 16-bit PCM at 16 kHz) -- so the measured loop is the real
one: read + decode the files on loader threads -> pad / mask in pinned memory -> H2D -> encoder -> D2H -> one pickle per utterance
on writer threads -- with K = 1, 2, 4, 8 batches in flight (extract.py --inflight).  The batches, and therefore the results, are
identical at every K; this script checks that byte for byte on every file before it prints a rate.

    python3 tools/cli_bench.py [N utterances, default 2000] [--big]     (--big adds the 30 s x batch 32 throughput case)
"""
import hashlib, importlib, json, os, shutil, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from scipy.io import wavfile  # noqa: E402
extract = importlib.import_module("loco-asr_amd.extract")
la = importlib.import_module("loco-asr_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
root = tempfile.mkdtemp(prefix="cli_bench_corpus_", dir=shm)
os.makedirs(os.path.join(root, "dataset", "slurp"))
os.makedirs(os.path.join(root, "audio", "slurp_real"))
lens = la.synth.mixed_lengths(N, 6 * 16000, min_fraction=2.0 / 6.0)
classes = sorted(extract.load_classes())


def write_clip(i):
    x = la.synth.clip(i, lens[i])
    wavfile.write(os.path.join(root, "audio", "slurp_real", f"audio-{i:06d}.wav"), 16000, np.clip(x * 32768.0, -32768, 32767).astype(np.int16))


t0 = time.perf_counter()
with ThreadPoolExecutor(16) as ex:
    list(ex.map(write_clip, range(N)))
with open(os.path.join(root, "dataset", "slurp", "devel.jsonl"), "w") as fh:
    for i in range(N):
        fh.write(json.dumps({"slurp_id": i, "sentence": "", "intent": classes[i % 101], "recordings": [{"file": f"audio-{i:06d}.wav"}]}) + "\n")
frames = sum(la.synth.conv_out_length(max(lens[i:i + 2])) * len(lens[i:i + 2]) for i in range(0, N, 2))  # padded frames, as pickled
valid = sum(la.synth.conv_out_length(n) for n in lens)
print(f"corpus: {N} WAV files of 2-6 s (mean {sum(lens) / N / 16000:.2f} s) in SLURP layout under {root} ({time.perf_counter() - t0:.1f} s to write); "
      f"reference batches of 2 in corpus order: {frames} frames encoded ({valid} valid)", flush=True)
base = ["-m", "audio", "-s", "devel", "--data-path", root, "--random-init", "--loader-threads", "12", "--sink-threads", "8"]
digests, results = {}, []
extract.main(["-m", "audio", "-s", "devel", "--synthetic", "32", "--synthetic-seconds", "6", "--random-init", "--out", tempfile.mkdtemp(prefix="cli_warm_", dir=shm)])
ref_dir = None
KS = (1,) if "--pack-only" in sys.argv else (1, 2, 4, 8)
for k in KS:
    out = tempfile.mkdtemp(prefix=f"cli_bench_k{k}_", dir=shm)
    st = extract.main(base + ["--out", out, "--inflight", str(k)])
    dt = st["seconds"]  # the loop alone: decode -> batching -> H2D -> encoder -> D2H -> pickles closed (no model load)
    assert st["frames"] == frames and st["utterances"] == N, st
    folder = os.path.join(out, "devel", "audio")
    h = hashlib.sha256()
    names = sorted(os.listdir(folder))
    for n in names:
        h.update(n.encode()); h.update(open(os.path.join(folder, n), "rb").read())
    digests[k] = (len(names), h.hexdigest())
    if k == 1 and "--no-pack" not in sys.argv:
        ref_dir = out  # kept: the --pack runs below are compared with it file by file
    else:
        shutil.rmtree(out, ignore_errors=True)
    r = dict(inflight=k, seconds=round(dt, 3), utterances_per_s=round(N / dt, 1), frames_per_s=round(frames / dt, 1), files=len(names), sha256=h.hexdigest()[:16])
    results.append(r)
    print(f"--inflight {k}: {N} utterances in {dt:.2f} s (decode, batching, H2D, encoder, D2H, pickles written) = {N / dt:.1f} utterances/s, "
          f"{frames / dt:,.0f} frames/s; {len(names)} pickles, sha256 {h.hexdigest()[:16]}", flush=True)
assert len({d for d in digests.values()}) == 1, f"pickles differ between --inflight values: {digests}"
print("all --inflight values wrote byte-identical pickles")
# ---- --pack G: G of the SAME reference batches per launch sequence (loco_forward_packed); equal to the runs above up to the fp32
# summation order of the GEMMs, checked file by file against the --inflight 1 pickles
if ref_dir is not None:
    import pickle
    ref_folder = os.path.join(ref_dir, "devel", "audio")
    extract.main(["-m", "audio", "-s", "devel", "--synthetic", "256", "--synthetic-seconds", "6", "--synthetic-min-seconds", "2", "--random-init", "--pack", "32",
                  "--out", tempfile.mkdtemp(prefix="cli_warm_", dir=shm)])
    configs = [(8, 3, 12, 8, 0.5), (16, 3, 12, 8, 0.5), (32, 3, 12, 8, 0.5), (64, 3, 12, 8, 0.5), (32, 2, 12, 8, 0.5), (32, 3, 12, 8, 5)]
    if "--packs" in sys.argv:  # G:inflight:loader threads:sink threads:gil switch ms, comma separated
        configs = [tuple(float(v) if "." in v else int(v) for v in c.split(":")) for c in sys.argv[sys.argv.index("--packs") + 1].split(",")]
    for cfg in configs:
        g, k, th, sk, gil = cfg[:5]
        win = int(cfg[5]) if len(cfg) > 5 else 8  # optional sixth field: --pack-window
        out = tempfile.mkdtemp(prefix=f"cli_bench_pack{g}_", dir=shm)
        st = extract.main(["-m", "audio", "-s", "devel", "--data-path", root, "--random-init", "--loader-threads", str(th), "--sink-threads", str(sk),
                           "--out", out, "--pack", str(g), "--inflight", str(k), "--gil-switch-ms", str(gil), "--pack-window", str(win)])
        dt = st["seconds"]
        assert st["frames"] == frames and st["utterances"] == N, st
        folder = os.path.join(out, "devel", "audio")
        names = sorted(os.listdir(folder))
        assert names == sorted(os.listdir(ref_folder))
        worst = 0.0
        for n in names:
            with open(os.path.join(folder, n), "rb") as fa, open(os.path.join(ref_folder, n), "rb") as fb:
                a, b = pickle.load(fa), pickle.load(fb)
            assert a["id"] == b["id"] and a["embedding"].shape == b["embedding"].shape and (a["target"] == b["target"]).all(), n
            e64 = b["embedding"].astype(np.float64)
            worst = max(worst, float(np.linalg.norm(a["embedding"] - e64) / np.linalg.norm(e64)))
        assert worst < 5e-6, worst
        shutil.rmtree(out, ignore_errors=True)
        r = dict(pack=g, inflight=k, loader_threads=th, sink_threads=sk, gil_switch_ms=gil, pack_window=win, seconds=round(dt, 3), utterances_per_s=round(N / dt, 1), frames_per_s=round(frames / dt, 1),
                 files=len(names), worst_rel_l2_vs_inflight1=worst)
        results.append(r)
        print(f"--pack {g} --inflight {k} ({th} loader / {sk} sink threads, switch interval {gil} ms, window {win}): {N} utterances in {dt:.2f} s = {N / dt:.1f} utterances/s, {frames / dt:,.0f} frames/s; "
              f"{len(names)} pickles, worst relative L2 against the --inflight 1 pickles {worst:.2e}", flush=True)
    shutil.rmtree(ref_dir, ignore_errors=True)
    if "--cold" in sys.argv:
        # the same corpus through a FRESH process (python loco-asr_amd/extract.py ...): nothing cached -- the slots' multi-GB workspaces, the
        # pinned staging buffers and the allocator's blocks are all first-time allocations inside the timed loop
        import re, subprocess
        for g in (32,):
            out = tempfile.mkdtemp(prefix=f"cli_bench_cold{g}_", dir=shm)
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "loco-asr_amd", "extract.py"), "-m", "audio", "-s", "devel",
                                "--data-path", root, "--random-init", "--loader-threads", "12", "--sink-threads", "8", "--out", out, "--pack", str(g)],
                               capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("Encoded")]
            print(f"cold process, --pack {g}: {line[-1] if line else r.stdout[-500:] + r.stderr[-500:]}", flush=True)
            shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(root, ignore_errors=True)
print(json.dumps({"corpus": {"utterances": N, "frames_padded": frames, "frames_valid": valid}, "runs": results}))
if "--big" in sys.argv:
    # the headline shape through the CLI: 768 WAV files of 30 s (SLURP layout, /dev/shm), --batch-size 32 (NOT the reference's batching:
    # throughput mode), one and two batches in flight; the GPU alone encodes 32 x 30 s in ~52 ms = 615 clips/s
    NB = 768
    root = tempfile.mkdtemp(prefix="cli_bench_big_", dir=shm)
    os.makedirs(os.path.join(root, "dataset", "slurp")); os.makedirs(os.path.join(root, "audio", "slurp_real"))

    def write_big(i):
        x = la.synth.clip(1000 + i, 480000)
        wavfile.write(os.path.join(root, "audio", "slurp_real", f"audio-{i:06d}.wav"), 16000, np.clip(x * 32768.0, -32768, 32767).astype(np.int16))
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(write_big, range(NB)))
    with open(os.path.join(root, "dataset", "slurp", "devel.jsonl"), "w") as fh:
        for i in range(NB):
            fh.write(json.dumps({"slurp_id": i, "sentence": "", "intent": classes[i % 101], "recordings": [{"file": f"audio-{i:06d}.wav"}]}) + "\n")
    for k, th, sk in ((1, 12, 8), (2, 12, 8), (2, 16, 12)):
        out = tempfile.mkdtemp(prefix="cli_bench_big_out_", dir=shm)
        st = extract.main(["-m", "audio", "-s", "devel", "--data-path", root, "--random-init", "--batch-size", "32", "--out", out,
                           "--loader-threads", str(th), "--sink-threads", str(sk), "--inflight", str(k)])
        print(f"CLI 30 s x batch 32 from WAV files, --inflight {k}, {th} loader / {sk} sink threads: {NB} clips in {st['seconds']:.2f} s = "
              f"{NB / st['seconds']:.1f} clips/s, {st['frames'] / st['seconds']:,.0f} frames/s", flush=True)
        shutil.rmtree(out, ignore_errors=True)
    shutil.rmtree(root, ignore_errors=True)
