#!/usr/bin/env python3
"""Drop-in for the audio branch of the reference's extraction scripts
(/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:96-113 and
extract_speecht5_finetuned_embeddings_slurp.py:93-113) on MI355X.

    python loco-asr_amd/extract.py -m audio -s train                       # SLURP layout under ./slurp
    python loco-asr_amd/extract.py -m audio -s devel --synthetic 64        # no corpus: seeded synthetic clips
    torchrun --standalone --nproc-per-node 8 loco-asr_amd/extract.py -m audio -s train     # data parallel

Same flags (-m/--modality, -s/--split), same folder layout (extracted/<model>/<split>/<modality>/), the same
per-utterance pickle and THE SAME BATCHES as the reference: ``batch_size = 2``, ``shuffle=False``, corpus order
(…base…py:67-68).  Batch composition is part of the function -- GroupNorm statistics run over the padded axis, so an
utterance encoded next to a longer one differs from the same utterance encoded alone -- and the
reference's pairs (0,1), (2,3), ... are therefore kept at every world size: with WORLD_SIZE > 1 whole batches are dealt
round-robin to the ranks (dp.shard_batches); each rank writes the files of its own batches, so no embedding crosses the
fabric unless --gather is given.  ``--batch-size N`` / ``--bucket-by-length`` trade that identity for throughput
(bigger batches; units sorted by length so that padding is minimal) and say so in their help text.

What differs by necessity: weights come from ``--prenet-state-dict`` / ``--encoder-state-dict`` /
``--text-prenet-state-dict`` (the pickled dicts the base script loads from extracted/speecht5/mapping/, …base…py:40-49)
or ``--random-init``, or from a checkpoint directory on disk (``--pretrained DIR``: the fine-tuned script's ``from_pretrained``
without the hub download); audio decoding needs no libsndfile -- 16 / 32-bit PCM and float WAV files and FLAC files (SLURP's format)
are decoded here (read_pcm_wav; the library's loco_flac_decode, CRC- and MD5-verified), anything else through soundfile when it is
installed -- followed by the device polyphase resampler to 16 kHz; ``-m text`` (…base…py:79-93) needs the tokenizer files in a
local directory.  Targets are one-hot over the reference's fixed 101 intent labels (``ALL_CLASSES``, …base…py:32-36),
shipped as loco-asr_amd/data/slurp_intent_classes.txt, for every split alike.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
la = importlib.import_module("loco-asr_amd")
dp = importlib.import_module("loco-asr_amd.dp")
sink_mod = importlib.import_module("loco-asr_amd.sink")


def read_slurp_split(data_path: str, split: str):
    """(slurp_id, sentence, wav_path, 16000, intent) per utterance, the tuple contract of the reference's
    SLURPDataset (slurp_data.py:19-66): headset recording if there is one, else the first."""
    text_path = os.path.join(data_path, "dataset/slurp", split + ".jsonl")
    audio_dir = os.path.join(data_path, "audio", "slurp_synth" if split == "train_synthetic" else "slurp_real")
    items = []
    with open(text_path) as fh:
        for line in fh:
            if not line.strip():
                continue
            item = json.loads(line)
            rec = next((r["file"] for r in item["recordings"] if "headset" in r), item["recordings"][0]["file"])
            items.append((item["slurp_id"], item["sentence"], os.path.join(audio_dir, rec), 16000, item["intent"]))
    return items


_soundfile = None  # module, or False once its import has failed (a failed import is re-tried -- and re-paid -- on every call otherwise)


def read_pcm_wav(path: str):
    """(rate, float32 mono samples in [-1, 1)) of an uncompressed RIFF/WAVE file -- 16 / 32-bit PCM or 32-bit float, the formats
    SLURP's .wav files and this repo's synthetic corpora use -- or None for anything else (the caller falls back to scipy).
    One read, one frombuffer, one scaling pass: the loader threads share the interpreter lock with the thread that feeds the GPU,
    so the Python spent per file matters more than the bytes."""
    import struct
    with open(path, "rb") as fh:
        raw = fh.read()
    if len(raw) < 44 or raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        return None
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
        if cid == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", raw, pos + 8)
        elif cid == b"data":
            data = (pos + 8, min(size, len(raw) - pos - 8))
            break
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        return None
    tag, channels, rate, _, _, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2", count=data[1] // 2, offset=data[0]).astype(np.float32)
        x *= np.float32(1.0 / 32768.0)
    elif tag == 1 and bits == 32:
        x = np.frombuffer(raw, dtype="<i4", count=data[1] // 4, offset=data[0]).astype(np.float32)
        x *= np.float32(1.0 / 2147483648.0)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4", count=data[1] // 4, offset=data[0]).astype(np.float32)
    else:
        return None
    if channels > 1:
        x = x[:x.size // channels * channels].reshape(-1, channels).mean(axis=1)
    return int(rate), x


def read_flac(path: str):
    """(rate, float32 mono samples) of a FLAC file through the library's host-side decoder -- bit-exact integer decoding with the
    frame CRCs and the STREAMINFO MD5 signature verified, scaled and mixed down as soundfile.read(dtype='float32').mean(axis=1) would."""
    import ctypes as C
    lib = importlib.import_module("loco-asr_amd._lib").load()
    with open(path, "rb") as fh:
        raw = fh.read()
    sr, ch, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    if lib.loco_flac_info(raw, len(raw), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total)):
        raise RuntimeError(f"{path}: {lib.loco_flac_last_error().decode()}")
    cap = int(total.value) if total.value > 0 else 8 * len(raw)  # unknown length: FLAC rarely beats 8 samples per byte... retried below
    while True:
        x = np.empty(cap, dtype=np.float32)
        n = C.c_int64()
        rc = lib.loco_flac_decode(raw, len(raw), x.ctypes.data_as(C.c_void_p), None, cap, C.byref(n), 1)
        if rc == -3 and total.value <= 0:
            cap *= 4
            continue
        if rc:
            raise RuntimeError(f"{path}: {lib.loco_flac_last_error().decode()}")
        return int(sr.value), x[:n.value]


def load_audio_16k(path: str, device=None):
    """mono float32 at 16 kHz (the reference uses librosa.load(path, sr=16000), …base…py:56): files at another rate (Fisher:
    8 kHz, podcasts: 44.1 kHz) are converted ON THE DEVICE by loco_op_resample (resample.py) and come back as CUDA tensors,
    which the feature extractor pads on the device; 16 kHz files stay numpy arrays on the host.  `device` = the rank's GPU: this
    runs on loader threads, where torch.cuda.current_device() is 0 whatever the main thread selected."""
    global _soundfile
    if _soundfile is None:
        try:
            import soundfile
            _soundfile = soundfile
        except ImportError:
            _soundfile = False
    try:
        got = read_pcm_wav(path) if path.lower().endswith(".wav") else None
        if got is None and path.lower().endswith(".flac") and not _soundfile:
            got = read_flac(path)  # SLURP's own format, decoded by the library (include/loco_asr.h, loco_flac_decode): no libsndfile needed
        if got is not None:
            sr, x = got
        elif _soundfile:
            x, sr = _soundfile.read(path, dtype="float32", always_2d=True)
            x = x.mean(axis=1)
        else:
            from scipy.io import wavfile
            if not path.lower().endswith(".wav"):
                raise RuntimeError("install soundfile for this format, or convert to WAV / FLAC")
            sr, x = wavfile.read(path)
            if x.dtype.kind == "i":
                x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
            elif x.dtype.kind == "u":  # 8-bit PCM is unsigned
                x = (x.astype(np.float32) - 128.0) / 128.0
            x = x.astype(np.float32)
            if x.ndim == 2:
                x = x.mean(axis=1)
    except Exception as e:  # a corpus of 100 000 files: the message must say WHICH one
        raise RuntimeError(str(e) if path in str(e) else f"cannot decode {path}: {e}") from e
    if sr != 16000:
        return importlib.import_module("loco-asr_amd.resample").resample_to_16k(x, int(sr), device=device)
    return x


CLASSES_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "slurp_intent_classes.txt")


def load_classes(path=None):
    """The reference's ALL_CLASSES (intent_classes.py, used at …base…py:32): the fixed 101 SLURP intent labels, the same for
    every split -- never the labels that happen to occur in the split being extracted (train has 91 of them, train_synthetic
    98: targets of different widths and index maps would result)."""
    with open(path or CLASSES_FILE) as fh:
        classes = [l.strip() for l in fh if l.strip()]
    if len(classes) != 101 or len(set(classes)) != 101:
        raise SystemExit(f"{path or CLASSES_FILE}: expected 101 distinct intent labels, found {len(classes)} "
                         f"({len(set(classes))} distinct): IntentClassifier's output layer is Linear(768, 101)")
    return classes


def one_hot_encoder(classes):
    """LabelEncoder + LabelBinarizer of the reference (…base…py:32-36): classes sorted, one-hot int64 rows."""
    order = sorted(set(classes))
    index = {c: i for i, c in enumerate(order)}
    eye = np.eye(len(order), dtype=np.int64)

    def encode(labels):
        unknown = [l for l in labels if l not in index]
        if unknown:  # sklearn's LabelEncoder.transform raises ValueError("y contains previously unseen labels") here
            raise ValueError(f"y contains previously unseen labels: {unknown}")
        return np.stack([eye[index[l]] for l in labels])
    return encode


def load_state_dict_file(path):
    with open(path, "rb") as fh:
        sd = pickle.load(fh)
    return {k: (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))) for k, v in sd.items()}


def extract_text(args, items, classes, encode_labels, device, world, rank):
    """The reference's text branch (…base…py:79-93): SpeechT5ForTextToSpeech(...).speecht5.encoder(texts.input_ids) -- ids
    padded to the longest transcript of the batch with <pad> = 1 and NO attention mask -- one pickle per utterance."""
    if args.random_init:
        _, enc_sd = la.synth.split_state_dict(la.synth.encoder_state_dict(0))
        enc_sd = {k: torch.from_numpy(v) for k, v in enc_sd.items()}
        tpre = {k[len("text_prenet."):]: torch.from_numpy(np.asarray(v)) for k, v in la.synth.text_prenet_state_dict(0).items()}
    else:
        tpre, enc_sd = load_state_dict_file(args.text_prenet_state_dict), load_state_dict_file(args.encoder_state_dict)
    model = la.SpeechT5ForTextToSpeechMI355X.from_state_dicts(tpre, enc_sd).to(device)
    print("Loaded model")
    model.eval()
    if args.synthetic:
        lens = [8 + (37 * i) % 90 for i in range(len(items))]
        tokenize = lambda idx: [la.synth.token_ids(1, lens[i], seed=100 + i)[0][0] for i in idx]
    else:
        try:
            from transformers import SpeechT5Tokenizer
            tok = SpeechT5Tokenizer.from_pretrained(args.tokenizer)
        except Exception as e:  # no hub access / no local files: say what is needed instead of a stack trace
            raise SystemExit(f"-m text needs the SpeechT5 tokenizer files (--tokenizer DIR with spm_char.model): {e}")
        tokenize = lambda idx: [np.asarray(tok(items[i][1])["input_ids"], dtype=np.int64) for i in idx]
    from collections import deque
    enc = model.speecht5.encoder
    enc.precision, enc.range_policy = getattr(args, "precision", enc.precision), getattr(args, "range_policy", enc.range_policy)
    inflight = args.inflight if args.inflight > 0 else 4
    if inflight > 1:
        enc.set_inflight(inflight)  # batches of transcripts are tiny (<= 2 x ~100 tokens): several in flight, results untouched
    tickets = deque()
    pack = max(0, getattr(args, "pack", 0))
    with torch.no_grad(), sink_mod.EmbeddingSink(args.out, args.split, args.modality, args.format) as sink:
        def finish():
            i0, t0 = tickets.popleft()
            res = t0.result() if hasattr(t0, "result") else t0
            if pack:  # i0 = the index lists of the pack's batches, res = one output per batch
                for idx_b, o in zip(i0, res):
                    sink.submit([items[i][0] for i in idx_b], o.last_hidden_state, encode_labels([items[i][4] for i in idx_b]))
            else:
                sink.submit([items[i][0] for i in i0], res.last_hidden_state, encode_labels([items[i][4] for i in i0]))

        def padded_ids(idx):
            seqs = tokenize(idx)
            T = max(len(q) for q in seqs)
            ids = np.full((len(seqs), T), 1, dtype=np.int64)  # padding="longest" with pad_token_id = 1
            for r, q in enumerate(seqs):
                ids[r, :len(q)] = q
            return torch.from_numpy(ids)

        my = dp.shard_batches(len(items), args.batch_size, world, rank)  # the reference's batches, dealt whole
        if pack:  # --pack G: G of those batches per forward, every row's keys ending where its own batch ends (text_encoder.forward_packed)
            enc.set_inflight(max(1, inflight))
            for g0 in range(0, len(my), pack):
                group = my[g0:g0 + pack]
                tickets.append((group, enc.forward_packed_async([padded_ids(idx) for idx in group])))
                while len(tickets) >= max(1, inflight):
                    finish()
        else:
            for idx in my:
                dev_ids = padded_ids(idx).to(device)
                tickets.append((idx, enc.forward_async(dev_ids) if inflight > 1 else enc(dev_ids)))
                while len(tickets) >= inflight:
                    finish()
        while tickets:
            finish()
    print("Done!")


def main(argv=None):
    ap = argparse.ArgumentParser(description="Extract SpeechT5 speech-encoder embeddings on MI355X")
    ap.add_argument("--modality", "-m", choices=["text", "audio"], required=True)
    ap.add_argument("--split", "-s", choices=["train", "devel", "test", "train_synthetic"], required=True)
    ap.add_argument("--data-path", default="slurp")
    ap.add_argument("--out", default=None,
                    help="output root; default extracted/speecht5_base (the base script's, …base…py:70) or, with --pretrained, "
                         "extracted/speecht5 (the fine-tuned script's, …finetuned…py:63-64)")
    ap.add_argument("--pretrained", default=None, metavar="DIR",
                    help="the fine-tuned script's weights (…finetuned…py:95: SpeechT5ForSpeechToText.from_pretrained('microsoft/speecht5_asr')) "
                         "from a checkpoint ON DISK: a directory with model.safetensors / pytorch_model.bin, such a file, or a hub name "
                         "already in the local HuggingFace cache (never downloaded); replaces the two --*-state-dict pickles")
    ap.add_argument("--resume", action="store_true",
                    help="skip every reference batch whose utterances all have their output file already (whole batches only: batch "
                         "composition is part of the function, a batch is never re-formed around a missing member)")
    ap.add_argument("--batch-size", type=int, default=2,
                    help="default 2 = the reference's batch_size (…base…py:67); batch composition changes the embeddings of padded "
                         "utterances (GroupNorm over the padded axis), so other values are not bit-comparable with the reference")
    ap.add_argument("--bucket-by-length", action="store_true",
                    help="sort units by length and deal them to ranks longest first (minimal padding, balanced T^2 work) instead of "
                         "the reference's corpus-order batches; changes which utterances share a batch")
    ap.add_argument("--prenet-state-dict", default="extracted/speecht5/mapping/speech_prenet_state_dict.pickle")
    ap.add_argument("--encoder-state-dict", default="extracted/speecht5/mapping/encoder_state_dict.pickle")
    ap.add_argument("--random-init", action="store_true", help="deterministic synthetic weights (no checkpoint available)")
    ap.add_argument("--synthetic", type=int, default=0, help="encode N seeded synthetic clips instead of a corpus")
    ap.add_argument("--synthetic-seconds", type=float, default=5.0)
    ap.add_argument("--synthetic-min-seconds", type=float, default=None,
                    help="shortest synthetic clip (default: half of --synthetic-seconds); 2 with --synthetic-seconds 6 = a SLURP-like "
                         "ragged corpus of 2-6 s utterances")
    ap.add_argument("--synthetic-exact", action="store_true",
                    help="every synthetic clip lasts exactly --synthetic-seconds (default: lengths drawn U[0.5, 1] x that, SURVEY.md 8d)")
    ap.add_argument("--classes-file", default=None,
                    help="one intent label per line; default: the reference's 101 ALL_CLASSES shipped with the package")
    ap.add_argument("--do-normalize", action="store_true")
    ap.add_argument("--normalize-on-device", action="store_true",
                    help="with --do-normalize: run zero_mean_unit_var_norm on the GPU right after the H2D copy instead of in numpy")
    ap.add_argument("--loader-threads", type=int, default=8,
                    help="decode / synthesise the NEXT batches on this many host threads while the GPU encodes the current one "
                         "(the reference loads inside collate_fn with num_workers=0, …base…py:53-57,67); 0 = inline")
    ap.add_argument("--sink-threads", type=int, default=4, help="threads that wait for the D2H copies and write the per-utterance files")
    ap.add_argument("--text-prenet-state-dict", default="extracted/speecht5/mapping/text_prenet_state_dict.pickle")
    ap.add_argument("--tokenizer", default="microsoft/speecht5_asr",
                    help="-m text: name or local directory of the SpeechT5 tokenizer (the reference's processor, …base…py:38)")
    ap.add_argument("--format", choices=["pickle", "npy"], default="pickle")
    ap.add_argument("--gather", action="store_true", help="all-gather embeddings so that rank 0 writes everything")
    ap.add_argument("--inflight", type=int, default=0,
                    help="batches in flight on the GPU at once (each on its own stream / workspace / status block).  The batches "
                         "themselves are untouched -- the reference's pairs in corpus order -- and so are the results, bit for bit; "
                         "a pair of 5 s clips alone cannot fill 256 CUs.  1 = one batch at a time, as the reference runs; 0 (default) = "
                         "4 for utterances, 1 for --window-seconds >= 60 (a pair of 10-minute windows fills the chip by itself and "
                         "every slot would hold a 15 GB workspace)")
    ap.add_argument("--pack", type=int, default=0,
                    help="encode G of the reference's batches as ONE launch sequence (loco_forward_packed): the batches themselves are "
                         "untouched -- corpus-order pairs, every clip keeps the padded length of its own batch for GroupNorm, the "
                         "positional conv, positions and the key mask -- so the embeddings equal those of --inflight to the fp32 "
                         "summation order of the GEMMs (<= 5e-6 relative L2), not bit for bit.  0 (default) = off.  32 is a good value "
                         "for 2-6 s utterances; --inflight then counts packs (default 3, each on one stream)")
    ap.add_argument("--pack-window", type=int, default=8,
                    help="with --pack: batches are sorted by padded length inside windows of this many packs before they are packed "
                         "(which batches share a pack does not change any embedding; short batches just do not idle in long packs).  "
                         "8: rows computed per frame kept 1.05; larger windows pad less (~1.02 at 32) for the price of more decoded audio "
                         "held on the host -- on 10 000 files 8 / 16 / 32 measured 784-818 k / 813 k / 827 k frames/s: within the run-to-run spread")
    ap.add_argument("--gil-switch-ms", type=float, default=0.5,
                    help="sys.setswitchinterval for the run, in ms (CPython's default is 5): the thread that enqueues forwards gives the "
                         "interpreter lock up at every library call and, with a dozen loader / writer threads runnable, waits a switch "
                         "interval per thread to get it back; 0 = leave the interpreter's setting alone")
    ap.add_argument("--precision", choices=["f16x3", "f32", "f16x2"], default="f16x3",
                    help="arithmetic of the contractions (include/loco_asr.h, loco_set_precision): f16x3 (default) = three fp16 MFMAs per "
                         "fp32-class product, ~1e-6 of an fp64 evaluation; f32 = every contraction on the exact fp32 MFMA (2.7x slower); "
                         "f16x2 = weights rounded to fp16, ~9e-4: at the 1e-3 bar, not inside it")
    ap.add_argument("--range-policy", choices=["fp32", "raise", "off"], default="fp32",
                    help="a batch whose activations leave the range of the fp16 planes (f16x3 / f16x2) is run again on the exact-fp32 "
                         "kernels (fp32, default), refused with an error (raise), or not checked at all (off)")
    ap.add_argument("--window-seconds", type=float, default=0.0,
                    help="cut every recording into windows of this many seconds (10-minute windows for hour-long podcasts, "
                         "BASELINE.json configs[3]); each window is an independent unit written as <id>_w<k>")
    args = ap.parse_args(argv)
    if args.out is None:
        args.out = os.path.join("extracted", "speecht5" if args.pretrained else "speecht5_base")
    if args.modality == "text" and (args.window_seconds > 0 or args.gather or args.resume or args.pretrained):
        raise SystemExit("-m text: --window-seconds / --gather / --resume / --pretrained apply to audio only")

    if args.gil_switch_ms > 0:
        sys.setswitchinterval(args.gil_switch_ms / 1000.0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    print(f"Extracting {args.modality} embeddings from SLURP {args.split} set using SpeechT5 (rank {rank}/{world})")
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device: this extractor has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    print("Running on", device)
    # LOCO_FORCE_COLLECTIVE=1 under a launcher: create the process group and issue every collective even at world size 1, so that
    # a one-GPU box runs the real RCCL gather of --gather on device tensors (tests/test_gpu_rccl_world1.py)
    collective = world > 1 or (os.environ.get("LOCO_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ)
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        dp.FORCE_COLLECTIVE = world == 1

    # ---- utterances
    if args.synthetic:
        n = int(args.synthetic_seconds * 16000)
        frac = 0.5 if args.synthetic_min_seconds is None else min(1.0, max(0.0, args.synthetic_min_seconds / args.synthetic_seconds))
        lens = [n] * args.synthetic if args.synthetic_exact else la.synth.mixed_lengths(args.synthetic, n, min_fraction=frac)
        classes = load_classes(args.classes_file)
        order = sorted(classes)
        items = [(f"synthetic-{i:06d}", "", None, 16000, order[i % 101]) for i in range(args.synthetic)]
        fetch = lambda i: la.synth.clip(i, lens[i])
        lengths = lens
    else:
        items = read_slurp_split(args.data_path, args.split)
        classes = load_classes(args.classes_file)
        fetch = lambda i: load_audio_16k(items[i][2], device)
        lengths = [0] * len(items)  # unknown until decoded
    if args.window_seconds > 0:
        # windows become the units: (id_wk, text, path, sr, label) with a fetch that slices the parent recording
        win = int(args.window_seconds * 16000)
        if not any(lengths):
            lengths = [len(fetch(i)) for i in range(len(items))]
        units = dp.window_units(lengths, win)
        parent_fetch, parent_items = fetch, items
        cache = {}

        def fetch(u, _units=units):  # noqa: E731 -- one decode per recording, sliced per window
            i, a, b = _units[u]
            if i not in cache:
                cache.clear()
                cache[i] = parent_fetch(i)
            return cache[i][a:b]
        items = [(f"{parent_items[i][0]}_w{a // win:03d}", parent_items[i][1], parent_items[i][2], 16000, parent_items[i][4])
                 for (i, a, b) in units]
        lengths = [b - a for (_, a, b) in units]
    print(f"{args.split} set size: {len(items)}")
    encode_labels = one_hot_encoder(classes)

    if args.modality == "text":
        return extract_text(args, items, classes, encode_labels, device, world, rank)

    # ---- model
    if args.pretrained:
        model = la.SpeechT5ForSpeechToTextMI355X.from_pretrained(args.pretrained).to(device)
    else:
        if args.random_init:
            pre, enc_sd = la.synth.split_state_dict(la.synth.encoder_state_dict(0))
            pre = {k: torch.from_numpy(v) for k, v in pre.items()}
            enc_sd = {k: torch.from_numpy(v) for k, v in enc_sd.items()}
        else:
            pre, enc_sd = load_state_dict_file(args.prenet_state_dict), load_state_dict_file(args.encoder_state_dict)
        model = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts(pre, enc_sd).to(device)
    print("Loaded model")
    model.eval()
    processor = la.SpeechT5FeatureExtractorMI355X(do_normalize=args.do_normalize, pin_memory=True,
                                                  normalize_on_device=args.do_normalize and args.normalize_on_device)

    if args.bucket_by_length:
        if not any(lengths):
            lengths = [len(fetch(i)) for i in range(len(items))]  # one decoding pass to learn the lengths
        shards = [dp.shard_units(lengths, world, r) for r in range(world)]
        my_batches = [shards[rank][a:a + args.batch_size] for a in range(0, len(shards[rank]), args.batch_size)]
        n_rounds = (max(len(s) for s in shards) + args.batch_size - 1) // args.batch_size
    else:  # the reference's batches: corpus order, whole batches dealt round-robin
        my_batches = dp.shard_batches(len(items), args.batch_size, world, rank)
        n_rounds = dp.rounds(len(items), args.batch_size, world)  # equal on all ranks: collectives line up
    if args.resume:
        if args.gather and collective:
            raise SystemExit("--resume with --gather: rank 0 writes other ranks' files, so a rank cannot tell which of ITS batches are done")
        folder = os.path.join(os.path.join(args.out, args.split), args.modality)
        done = lambda b: all(os.path.exists(sink_mod.embedding_path(folder, items[i][0], args.format)) for i in b)  # noqa: E731
        kept = [b for b in my_batches if not done(b)]
        print(f"--resume: {len(my_batches) - len(kept)} of {len(my_batches)} batches already written, {len(kept)} to encode")
        my_batches, n_rounds = kept, len(kept)

    pack = max(0, args.pack)
    pack_window = max(1, args.pack_window)
    if pack:
        processor.pin_memory = False  # batches are copied into the pack's pinned buffer: their own staging need not be pinned
    if pack * args.batch_size > 512:
        raise SystemExit(f"--pack {pack} x --batch-size {args.batch_size} = {pack * args.batch_size} clips per launch sequence; the library takes "
                         "at most 512 (loco_max_pack_clips) -- and gains nothing beyond ~128 clips of utterance length")
    if pack and args.bucket_by_length:
        raise SystemExit("--pack keeps the reference's batches (it packs WHOLE batches); --bucket-by-length re-forms them: use one or the other")
    # --pack G: a "round" is one pack of up to G of this rank's batches; equal on all ranks, so that collectives line up
    n_packs = (n_rounds + pack - 1) // pack if pack else 0

    def host_features(idx):
        """decode / synthesise the clips of ONE reference batch, pad and mask them (SpeechT5FeatureExtractor, …base…py:51-65)"""
        # a reference batch is two clips: decoded one after the other, the parallelism is across batches; a throughput batch
        # (--batch-size 32: 32 files of 30 s) is decoded by its own pool, or the GPU waits for one thread to read 30 MB of audio
        clips = list(clip_pool.map(fetch, idx)) if clip_pool is not None and len(idx) >= 8 else [fetch(i) for i in idx]
        if any(torch.is_tensor(c) for c in clips):  # some files were resampled on the device: the whole batch is padded there
            clips = [c if torch.is_tensor(c) else torch.from_numpy(np.ascontiguousarray(c)).to(device) for c in clips]
        return processor(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")

    def host_batch(rnd):
        """Everything the host does for one batch: decode, pad, mask (pinned memory) -- run ahead of the GPU on worker threads,
        several BATCHES at a time (a batch of the reference is two clips: parallelism inside one would be two threads wide)."""
        idx = my_batches[rnd] if rnd < len(my_batches) else []
        if not idx:
            return [(idx, None)]
        # the H2D copies are enqueued HERE, by the loader thread (pinned source, non-blocking, the default stream): the thread that
        # enqueues forwards only waits for the future; forward_async orders its stream behind the default stream
        return [(idx, host_features(idx).to(device))]

    fused_pack = not (args.do_normalize and args.normalize_on_device)

    def host_feats_for_pack(idx):
        """one reference batch for the pack stage: its decoded clips (host arrays: padded and packed in one go by
        processor.pack_clips), or -- some clip was resampled on the device, or the normaliser is deferred to the device -- the
        batch as the processor forms it"""
        clips = [fetch(i) for i in idx]
        if fused_pack and not any(torch.is_tensor(c) for c in clips):
            return clips
        if any(torch.is_tensor(c) for c in clips):
            clips = [c if torch.is_tensor(c) else torch.from_numpy(np.ascontiguousarray(c)).to(device) for c in clips]
        f = processor(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
        # the deferred normaliser (--normalize-on-device) runs per BATCH, before batches are packed
        return f.to(device) if getattr(processor, "normalize_on_device", False) else f

    def batch_len(f):
        return max(len(c) for c in f) if isinstance(f, list) else int(f["input_values"].shape[1])

    def batch_clips(f):
        return len(f) if isinstance(f, list) else int(f["input_values"].shape[0])

    def make_pack(fs):
        if all(isinstance(f, list) for f in fs):
            return processor.pack_clips(fs, device, encoder._lib.loco_output_frames)
        fs = [processor(audio=f, sampling_rate=16000, return_tensors="pt", padding="longest") if isinstance(f, list) else f for f in fs]
        return encoder.pack_batches(fs, device)

    def staged_packs():
        """--pack: (list of unit-index lists, packed tensors) per pack, in order.  Three host stages run ahead of the GPU:
        (1) every reference batch is decoded, padded and masked BY ITSELF on the loader pool (it stays the batch the reference
        forms), two windows ahead; (2) when a window of pack_window packs has all its batches, they are sorted by padded length
        and cut into packs of G -- any set of batches may share a pack (include/loco_asr.h), sorting only removes the rows a
        short batch would idle in a long pack; (3) each pack is laid out in pinned memory and crosses PCIe as one copy
        (encoder.pack_batches) on a small pool of its own, so that it does not queue behind the decoding of later windows."""
        # windows of 1, 2, 4, ... pack_window packs: the first pack can leave as soon as ITS batches are decoded (a sorted window
        # cannot be cut before its last batch is in), the steady state sorts over pack_window packs
        bounds, p0, size = [], 0, 1
        while p0 < n_packs:
            bounds.append((p0, min(n_packs, p0 + size)))
            p0 += size
            size = min(pack_window, 2 * size)
        from collections import deque
        futs, submitted = {}, 0
        for w, (p0, p1) in enumerate(bounds):
            ahead_to = bounds[min(len(bounds) - 1, w + 2)][1] * pack  # decoding runs two windows ahead
            while pool and submitted < min(ahead_to, len(my_batches)):
                futs[submitted] = pool.submit(host_feats_for_pack, my_batches[submitted])
                submitted += 1
            lo, hi = min(len(my_batches), p0 * pack), min(len(my_batches), p1 * pack)
            feats = {i: (futs.pop(i).result() if pool else host_feats_for_pack(my_batches[i])) for i in range(lo, hi)}
            order = sorted(range(lo, hi), key=lambda i: (batch_len(feats[i]), i))
            if order:  # the window's largest pack sizes the slots' workspaces at their next (re)allocation
                encoder.reserve_workspace(sum(batch_clips(feats[i]) for i in order[-pack:]), batch_len(feats[order[-1]]) + 8)
            groups = [order[g0:g0 + pack] for g0 in range(0, len(order), pack)]
            jobs, nxt = deque(), 0
            for k in range(len(groups)):
                # at most a handful of packs are being laid out at a time: each is two pinned buffers of tens of MB, and a whole
                # window's worth at once sends the pinned allocator to the driver (hipHostMalloc stalls every other HIP call)
                while nxt < len(groups) and len(jobs) < 5:
                    sel = groups[nxt]
                    meta, fs = [my_batches[i] for i in sel], [feats[i] for i in sel]
                    jobs.append((meta, pack_pool.submit(make_pack, fs) if pack_pool else fs))
                    nxt += 1
                meta, job = jobs.popleft()
                yield meta, (job.result() if pack_pool else make_pack(job))
            feats = None
            for _ in range((p1 - p0) - len(groups)):
                yield [], None  # this rank has run out of batches: empty contributions keep the collectives lined up

    def staged_batches():
        ahead = max(2, 2 * inflight, n_loaders)  # batches being prepared while others are on the GPU
        pending = [pool.submit(host_batch, r) for r in range(min(ahead, n_rounds))] if pool else []
        for rnd in range(n_rounds):
            if pool:
                work = pending.pop(0).result()
                if rnd + ahead < n_rounds:
                    pending.append(pool.submit(host_batch, rnd + ahead))
            else:
                work = host_batch(rnd)
            yield from work

    def on_device(_device=device):  # loader threads start on GPU 0: select the rank's GPU for anything they do there
        torch.cuda.set_device(_device)

    from concurrent.futures import ThreadPoolExecutor
    inflight = args.inflight if args.inflight > 0 else (1 if args.window_seconds >= 60 else (3 if pack else 4))
    # the window cache of one decoded recording is not thread-safe: windows are prepared by ONE thread, in order
    n_loaders = 0 if args.loader_threads <= 0 else (1 if args.window_seconds > 0 else args.loader_threads)
    pool = ThreadPoolExecutor(n_loaders, initializer=on_device) if n_loaders > 0 else None
    clip_pool = ThreadPoolExecutor(n_loaders, initializer=on_device) if n_loaders > 1 and args.batch_size >= 8 and args.window_seconds <= 0 else None
    pack_pool = ThreadPoolExecutor(min(4, n_loaders), initializer=on_device) if pack and n_loaders > 0 else None
    encoder = model.speecht5.encoder
    encoder.precision, encoder.range_policy = args.precision, args.range_policy
    if inflight > 1 or pack:
        encoder.set_inflight(max(1, inflight))
    if pack and inflight > 1:
        # packs in flight fill each other's tails: the two half-batch schedule INSIDE a forward (loco_set_streams) then only adds
        # launches -- measured at 32 pairs per pack: 820 k frames/s with two streams per pack and two packs in flight, 924 k with one
        # stream per pack, 944 k with three packs in flight (profiles/r04_packed_bench_streams.txt)
        encoder.streams = 1
    gathers = 0
    gatherer = dp.RaggedGatherPipeline(cap=args.batch_size * max(1, pack)) if args.gather and collective else None

    def write_gathered(done):
        if rank == 0:
            for gid, e in done:
                sink.submit([items[gid][0]], e[None], encode_labels([items[gid][4]]))

    def finish(meta, res):
        """What follows a forward: the optional gather, then the sink -- in order, whatever finished first.  meta = the unit
        indices of one batch, or (--pack) the list of such lists of one pack; res = BaseModelOutput / ticket / None."""
        nonlocal gathers, frames_done
        t_f0 = time.perf_counter()
        if pack:
            if res is not None:
                res.result()
                emb, spans = res.packed_output()
                idx = [i for b in meta for i in b]
                rows = [t for (_, nb, t) in spans for _ in range(nb)]
                frames_done += sum(nb * t for (_, nb, t) in spans)
            else:
                emb, idx, rows = None, [], None
        else:
            out = res.result() if hasattr(res, "result") else res
            emb, idx, rows = (out.last_hidden_state if out is not None else None), meta, None
            if emb is not None:
                frames_done += int(emb.shape[0]) * int(emb.shape[1])
        t_f1 = time.perf_counter()
        if gatherer is not None:
            write_gathered(gatherer.submit(emb, idx, rows, device))  # two collectives per round, both overlapped (dp.py)
            gathers += 1
        elif idx:
            sink.submit([items[i][0] for i in idx], emb, encode_labels([items[i][4] for i in idx]), rows=rows, chunk=8 if pack else 0)
        cprof[0] += t_f1 - t_f0
        cprof[1] += time.perf_counter() - t_f1

    import queue
    import threading
    import time
    frames_done = 0
    cprof = [0.0, 0.0]  # consumer side, seconds: waiting for forwards (+ range check), handing results to the gather / the sink
    torch.cuda.synchronize(device)
    t_loop = time.perf_counter()
    with torch.no_grad(), sink_mod.EmbeddingSink(args.out, args.split, args.modality, args.format, workers=args.sink_threads,
                                                 max_pending=max(4, 4 * inflight, 16 * (pack > 0))) as sink:
        # Two host threads share the loop when batches are in flight: this one stages (H2D) and enqueues forwards, the consumer
        # waits for each batch IN ORDER (an event, not a device-wide synchronisation), checks its range status and hands it to the
        # gather / the sink.  The queue is bounded: at most `inflight` batches wait behind the one being finished.
        todo = queue.Queue(maxsize=inflight)
        failure = []

        def consume():
            try:
                torch.cuda.set_device(device)
                with torch.no_grad():
                    while True:
                        item = todo.get()
                        if item is None:
                            return
                        finish(*item)
            except BaseException as e:  # noqa: BLE001 -- re-raised on the producing thread
                failure.append(e)
                while todo.get() is not None:  # keep draining so that the producer never blocks on a dead consumer
                    pass

        consumer = threading.Thread(target=consume, name="loco-finish") if (inflight > 1 or pack) else None
        if consumer:
            consumer.start()
        try:
            prof = [0.0] * 4 if os.environ.get("LOCO_EXTRACT_PROFILE") == "1" else None  # seconds: wait for the staged batch, H2D, enqueue, hand-over
            tick = time.perf_counter
            stream_ = staged_packs() if pack else staged_batches()
            while not failure:
                t_a = tick()
                try:
                    meta, feats = next(stream_)
                except StopIteration:
                    break
                t_b = tick()
                if feats is None:  # this rank has run out of batches: an empty contribution keeps the collectives lined up
                    item = (meta, None)
                    t_c = t_b
                elif pack:
                    t_c = t_b
                    item = (meta, encoder.forward_packed_async(packed=feats))
                else:
                    on_dev = feats if feats["input_values"].is_cuda else feats.to(device)
                    t_c = tick()
                    item = (meta, encoder.forward_async(**on_dev) if inflight > 1 else encoder(**on_dev))
                t_d = tick()
                if consumer:
                    todo.put(item)
                else:
                    finish(*item)
                if prof is not None:
                    t_e = tick()
                    for k_, v_ in enumerate((t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d)):
                        prof[k_] += v_
            if prof is not None:
                t_close = tick()
                print("consumer thread so far, s: waiting for forwards %.3f, gather / sink hand-over %.3f; loop so far %.3f s" % (cprof[0], cprof[1], t_close - t_loop))
                if getattr(encoder, "submit_profile", None):
                    print("inside forward_async / forward_packed_async, s:", {k_: round(v_, 4) for k_, v_ in encoder.submit_profile.items()})
                print("main thread, ms per %s: wait for the staged batch %.3f, H2D %.3f, enqueue %.3f, hand-over / finish %.3f"
                      % (("pack" if pack else "batch",) + tuple(1e3 * v_ / max(1, n_packs if pack else n_rounds) for v_ in prof)))
        except BaseException as e:  # noqa: BLE001 -- a failure of THIS thread (an unreadable file, a refused forward): handled below like the consumer's
            failure.insert(0, e)
        finally:
            if consumer:
                todo.put(None)
                consumer.join()
        if failure:
            if collective:
                # the other ranks are (or will be) blocked in this round's collectives: a rank that stops contributing must take the
                # job down, not leave it hanging -- the launcher (torchrun) tears the other ranks down when one exits non-zero
                import traceback
                traceback.print_exception(type(failure[0]), failure[0], failure[0].__traceback__)
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(13)
            raise failure[0]
        if gatherer is not None:
            write_gathered(gatherer.flush())
    torch.cuda.synchronize(device)
    t_loop = time.perf_counter() - t_loop  # decode / synthesis -> batching -> encoder -> sink, files closed
    n_mine = sum(len(b) for b in my_batches)
    print(f"Encoded {n_mine} utterances ({frames_done} frames, padded frames included) in {t_loop:.3f} s: "
          f"{n_mine / max(t_loop, 1e-9):.1f} utterances/s, {frames_done / max(t_loop, 1e-9):,.0f} frames/s "
          f"(--inflight {inflight}{f', --pack {pack}' if pack else ''})")
    stats = {"utterances": n_mine, "frames": frames_done, "seconds": t_loop, "inflight": inflight, "pack": pack,
             "gpu_gib_allocated_peak": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
             "gpu_gib_reserved_peak": round(torch.cuda.max_memory_reserved(device) / 2 ** 30, 2)}
    print(f"GPU memory: {stats['gpu_gib_allocated_peak']} GiB allocated at peak, {stats['gpu_gib_reserved_peak']} GiB reserved by the allocator")
    if args.gather and collective:
        import torch.distributed as dist
        print(f"Embedding gathers issued: {gathers} rounds, {gatherer.collectives} collectives (backend {dist.get_backend()}, world size {world})")
        stats["gather_rounds"], stats["collectives"] = gathers, gatherer.collectives
    for ex in (pool, clip_pool, pack_pool):
        if ex is not None:
            ex.shutdown()
    print("Done!")
    if collective:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return stats


if __name__ == "__main__":
    main()
