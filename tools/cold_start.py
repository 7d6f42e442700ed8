#!/usr/bin/env python3
"""What a FRESH process pays before its GPU is busy: the extraction CLI (--pack 32) on a 2 000-WAV SLURP-layout corpus in /dev/shm, run as
`python loco-asr_amd/extract.py ...` R times, with the loop's own profile (LOCO_EXTRACT_PROFILE=1) and the process's wall time from spawn to
exit (imports, model load and the HIP runtime's start included).

    python3 tools/cold_start.py [N utterances, default 2000] [R runs, default 3] [extra extract.py arguments ...]"""
import importlib, json, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
from scipy.io import wavfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
la = importlib.import_module("loco-asr_amd")
nums = [a for a in sys.argv[1:] if a.isdigit()]
N = int(nums[0]) if nums else 2000
R = int(nums[1]) if len(nums) > 1 else 3
extra = sys.argv[1 + len(nums):]
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
root = tempfile.mkdtemp(prefix="cold_corpus_", dir=shm)
os.makedirs(os.path.join(root, "dataset", "slurp")); os.makedirs(os.path.join(root, "audio", "slurp_real"))
lens = la.synth.mixed_lengths(N, 6 * 16000, min_fraction=2.0 / 6.0)
with open(os.path.join(ROOT, "loco-asr_amd", "data", "slurp_intent_classes.txt")) as fh:
    classes = sorted(l.strip() for l in fh if l.strip())


FLAC = "--flac" in extra  # SLURP's own format: files written by the test suite's plain-Python FLAC writer (16-bit mono, 4 096-sample blocks, an
if FLAC:                  # eighth-order LPC subframe with Rice partitions) -- slow to write, so 256 distinct files are written and the corpus cycles them
    extra.remove("--flac")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
DISTINCT = min(N, 256) if FLAC else N
EXT = "flac" if FLAC else "wav"


def write_clip(i):
    pcm = np.clip(la.synth.clip(i, lens[i]) * 32768.0, -32768, 32767).astype(np.int16)
    path = os.path.join(root, "audio", "slurp_real", f"audio-{i:06d}.{EXT}")
    if not FLAC:
        wavfile.write(path, 16000, pcm)
        return
    import flac_writer as fw
    frames, at = [], 0
    while at < len(pcm):
        size = min(4096, len(pcm) - at)
        spec = dict(kind="lpc", order=8, precision=12, shift=10, coefs=[900, -300, 120, -60, 30, -10, 5, -2], porder=3 if size == 4096 else 0, method=0)
        if size < 16:  # a last block shorter than the predictor
            spec = dict(kind="verbatim")
        frames.append(dict(size=size, specs=[spec]))
        at += size
    with open(path, "wb") as fh:
        fh.write(fw.write_stream(pcm.astype(np.int64)[:, None], 16, 16000, frames))


if FLAC:
    import multiprocessing as mp
    with mp.Pool(16) as pool_:
        pool_.map(write_clip, range(DISTINCT))
    lens = [lens[i % DISTINCT] for i in range(N)]
else:
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(write_clip, range(N)))
with open(os.path.join(root, "dataset", "slurp", "devel.jsonl"), "w") as fh:
    for i in range(N):
        fh.write(json.dumps({"slurp_id": i, "sentence": "", "intent": classes[i % 101], "recordings": [{"file": f"audio-{i % DISTINCT:06d}.{EXT}"}]}) + "\n")
print(f"corpus: {N} {EXT.upper()} files of 2-6 s ({DISTINCT} distinct) under {root}; extract.py arguments: --pack 32 {' '.join(extra)}", flush=True)
env = dict(os.environ, LOCO_EXTRACT_PROFILE="1")
for run in range(R):
    out = tempfile.mkdtemp(prefix="cold_out_", dir=shm)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "loco-asr_amd", "extract.py"), "-m", "audio", "-s", "devel", "--data-path", root, "--random-init",
                        "--loader-threads", "12", "--sink-threads", "8", "--out", out, "--pack", "32"] + extra, capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t0
    print(f"--- fresh process {run}: {wall:.2f} s from spawn to exit (rc {r.returncode})")
    for l in r.stdout.splitlines():
        if l.startswith(("Encoded", "consumer", "inside", "main thread", "start-up")):
            print("   ", l[:400])
    if r.returncode:
        print(r.stderr[-1500:])
    shutil.rmtree(out, ignore_errors=True)
shutil.rmtree(root, ignore_errors=True)
