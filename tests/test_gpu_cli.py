"""The drop-in extraction CLI end to end on one MI355X: synthetic clips -> encoder -> asynchronous sink -> the
reference's per-utterance pickle, read back with the reader that restates SLURPEmbeddingsTargets and checked
against the CPU oracle."""
import importlib
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_extract_cli_synthetic(tmp_path, oracle):
    """Default flags = the reference's loop: batch_size 2, shuffle=False (…base…py:67-68).  A ragged 7-clip corpus must give,
    per utterance, what the oracle gives on the reference's pairs (0,1), (2,3), (4,5), (6) -- built here from the corpus
    order itself, not from the package's sharding helpers."""
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    sink = importlib.import_module("loco-asr_amd.sink")
    out = str(tmp_path / "extracted" / "speecht5_base")
    extract.main(["-m", "audio", "-s", "devel", "--synthetic", "7", "--synthetic-seconds", "1.5", "--random-init", "--out", out])
    ds = sink.EmbeddingsTargets(out, modality="audio", split="devel")
    assert len(ds) == 7
    n = int(1.5 * 16000)
    lens = la.synth.mixed_lengths(7, n)
    assert len(set(lens)) == 7  # ragged: every pair pads its shorter member
    sd = la.synth.encoder_state_dict(0)
    fe = la.SpeechT5FeatureExtractorMI355X()
    for idx in ([0, 1], [2, 3], [4, 5], [6]):
        b = fe(audio=[la.synth.clip(i, lens[i]) for i in idx], sampling_rate=16000)
        ref = oracle.encode(b["input_values"], b["attention_mask"], sd)
        for row, i in enumerate(idx):
            with open(os.path.join(out, "devel", "audio", f"synthetic-{i:06d}_embedding_and_target.pickle"), "rb") as fh:
                d = pickle.load(fh)
            assert d["embedding"].shape == tuple(ref[row].shape) and d["embedding"].dtype == np.float32
            rel = np.linalg.norm(d["embedding"] - ref[row].numpy()) / np.linalg.norm(ref[row].numpy())
            assert rel < 1e-4
            assert d["target"].shape == (101,) and d["target"].sum() == 1 and d["target"][i % 101] == 1
    # --bucket-by-length regroups the clips (longest first): same files, but the padded clips see other batch mates
    out2 = str(tmp_path / "bucketed")
    extract.main(["-m", "audio", "-s", "devel", "--synthetic", "7", "--synthetic-seconds", "1.5", "--random-init", "--bucket-by-length",
                  "--out", out2])
    assert sorted(os.listdir(os.path.join(out2, "devel", "audio"))) == sorted(os.listdir(os.path.join(out, "devel", "audio")))


def test_extract_cli_text_modality(tmp_path, oracle):
    """-m text (…base…py:79-93): synthetic token ids -> text prenet + encoder -> the same per-utterance pickles; ids are
    padded with <pad> = 1 and NO mask is passed, exactly like `model.speecht5.encoder(texts.input_ids)`."""
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    out = str(tmp_path / "extracted" / "speecht5_base")
    extract.main(["-m", "text", "-s", "devel", "--synthetic", "5", "--random-init", "--batch-size", "3", "--out", out])
    sd = la.synth.encoder_state_dict(0)
    sd.update(la.synth.text_prenet_state_dict(0))
    lens = [8 + (37 * i) % 90 for i in range(5)]
    for idx in ([0, 1, 2], [3, 4]):
        seqs = [la.synth.token_ids(1, lens[i], seed=100 + i)[0][0] for i in idx]
        T = max(len(q) for q in seqs)
        ids = np.full((len(seqs), T), 1, dtype=np.int64)
        for r, q in enumerate(seqs):
            ids[r, :len(q)] = q
        ref = oracle.encode_text(ids, None, sd)
        for row, i in enumerate(idx):
            with open(os.path.join(out, "devel", "text", f"synthetic-{i:06d}_embedding_and_target.pickle"), "rb") as fh:
                d = pickle.load(fh)
            assert d["embedding"].shape == (T, 768) and d["embedding"].dtype == np.float32
            assert np.linalg.norm(d["embedding"] - ref[row].numpy()) / np.linalg.norm(ref[row].numpy()) < 1e-4
    # a real corpus needs the tokenizer files, which are not in the container: a clear message, not a stack trace
    import json
    os.makedirs(tmp_path / "slurp" / "dataset" / "slurp")
    with open(tmp_path / "slurp" / "dataset" / "slurp" / "devel.jsonl", "w") as fh:
        fh.write(json.dumps({"slurp_id": 7, "sentence": "wake me up at nine", "intent": "alarm_set", "recordings": [{"file": "a.flac"}]}) + "\n")
    with pytest.raises(SystemExit, match="tokenizer"):
        extract.main(["-m", "text", "-s", "devel", "--data-path", str(tmp_path / "slurp"), "--random-init", "--out", out,
                      "--tokenizer", str(tmp_path / "no_tokenizer_here")])


def test_extract_cli_normalize_on_device_equals_host(tmp_path):
    extract = importlib.import_module("loco-asr_amd.extract")
    a, b = str(tmp_path / "host"), str(tmp_path / "dev")
    common = ["-m", "audio", "-s", "devel", "--synthetic", "3", "--synthetic-seconds", "1", "--random-init", "--batch-size", "3",
              "--do-normalize"]
    extract.main(common + ["--out", a])
    extract.main(common + ["--normalize-on-device", "--out", b])
    for i in range(3):
        name = os.path.join("devel", "audio", f"synthetic-{i:06d}_embedding_and_target.pickle")
        ea = pickle.load(open(os.path.join(a, name), "rb"))["embedding"]
        eb = pickle.load(open(os.path.join(b, name), "rb"))["embedding"]
        assert np.linalg.norm(ea - eb) / np.linalg.norm(ea) < 2e-5


def test_extract_then_train_head_pipeline(tmp_path):
    """configs[4] in miniature: synthetic clips -> encoder -> pickles -> train_head.py (the train_classifier.py
    drop-in) for two epochs; the head must fit the 8-class toy labels better than chance and write the reference's
    checkpoint / log files."""
    extract = importlib.import_module("loco-asr_amd.extract")
    train_head = importlib.import_module("loco-asr_amd.train_head")
    root = str(tmp_path / "extracted" / "speecht5_base")
    for split, n in (("train", 24), ("devel", 8)):
        extract.main(["-m", "audio", "-s", split, "--synthetic", str(n), "--synthetic-seconds", "1", "--random-init",
                      "--batch-size", "8", "--out", root])
    tl, ta = train_head.main(["-m", "audio", "-p", "attention", "-v", "base", "--folder", root, "--epochs", "2",
                              "--out-root", str(tmp_path)])
    ck = tmp_path / "checkpoints" / "base" / "audio" / "attention"
    assert (ck / "speecht5_attention_audio_best.pth").exists() and (ck / "speecht5_attention_audio_last.pth").exists()
    assert (ck / "speecht5_attention_audio_epoch_2.pth").exists()
    assert (tmp_path / "results" / "base" / "audio" / "attention" / "logs" / "results.txt").exists()
    sd = torch.load(ck / "speecht5_attention_audio_best.pth")
    assert set(sd) == {"q", "classifier.0.weight", "classifier.0.bias"}
    assert np.isfinite(tl)


def test_extract_cli_windows(tmp_path, oracle):
    """--window-seconds: a 2.5 s synthetic clip cut into 1 s windows -> 3 units, each equal to encoding that slice alone."""
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    out = str(tmp_path / "w")
    extract.main(["-m", "audio", "-s", "test", "--synthetic", "1", "--synthetic-seconds", "5", "--random-init", "--window-seconds", "1",
                  "--batch-size", "4", "--out", out])
    n = la.synth.mixed_lengths(1, 80000)[0]
    clip = la.synth.clip(0, n)
    files = sorted(os.listdir(os.path.join(out, "test", "audio")))
    nwin = (n + 15999) // 16000 if n % 16000 >= 400 or n % 16000 == 0 else n // 16000
    assert len(files) == nwin and files[0] == "synthetic-000000_w000_embedding_and_target.pickle"
    sd = la.synth.encoder_state_dict(0)
    with open(os.path.join(out, "test", "audio", files[1]), "rb") as fh:
        d = pickle.load(fh)
    ref = oracle.encode(clip[None, 16000:32000], None, sd)[0].numpy()
    # windows of equal length batched together are independent units
    assert d["embedding"].shape == ref.shape
    assert np.linalg.norm(d["embedding"] - ref) / np.linalg.norm(ref) < 1e-4


def test_extract_cli_one_hour_recording_in_ten_minute_windows(tmp_path, oracle):
    """BASELINE.json configs[3] on one GPU at its real size: one 60-minute recording through --window-seconds 600 -> six units of
    T = 29 999 frames, <id>_w000 ... _w005, encoded in the reference's batches of two.  Chain of evidence for the last pair:
    the pickles equal a direct forward of the same two windows bit for bit, and that forward's last layer reproduces the fp64
    row oracle on window 5 (keys over all 29 999 frames)."""
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    from gpu_util import model, rel_l2
    out = str(tmp_path / "podcast")
    extract.main(["-m", "audio", "-s", "test", "--synthetic", "1", "--synthetic-seconds", "3600", "--synthetic-exact", "--random-init",
                  "--window-seconds", "600", "--out", out])
    files = sorted(os.listdir(os.path.join(out, "test", "audio")))
    assert files == [f"synthetic-000000_w{k:03d}_embedding_and_target.pickle" for k in range(6)]
    emb = {}
    for k in (0, 4, 5):
        with open(os.path.join(out, "test", "audio", files[k]), "rb") as fh:
            d = pickle.load(fh)
        assert d["id"] == f"synthetic-000000_w{k:03d}" and d["embedding"].shape == (29999, 768) and d["embedding"].dtype == np.float32
        assert np.isfinite(d["embedding"]).all() and d["target"].shape == (101,)
        emb[k] = d["embedding"]
    assert not np.array_equal(emb[0], emb[5])
    win = 600 * 16000
    rec = la.synth.clip(0, 6 * win)
    m, sd = model()
    pair = torch.from_numpy(np.stack([rec[4 * win:5 * win], rec[5 * win:6 * win]])).cuda()
    res = m.speecht5.encoder(input_values=pair, attention_mask=torch.ones_like(pair, dtype=torch.int32), output_hidden_states=True)
    y = res.last_hidden_state.cpu().numpy()
    assert np.array_equal(y[0], emb[4]) and np.array_equal(y[1], emb[5])
    rows = [0, 1, 159, 160, 161, 15000, 29838, 29998]
    pe_k = torch.from_numpy(sd["wrapped_encoder.embed_positions.pe_k.weight"])
    ref = oracle.encoder_layer_rows(res.hidden_states[11][1].cpu(), rows, 29999, sd, "wrapped_encoder.layers.11.", pe_k)
    assert rel_l2(emb[5][rows], ref) < 1e-5


def test_extract_cli_pretrained_directory_and_resume(tmp_path):
    """The fine-tuned script's flow (…finetuned…py:95,104-113) with its weights in a checkpoint directory on disk (--pretrained), and
    --resume: after half the files have been removed, only the reference batches with a missing member are encoded again -- whole
    batches, so every re-written file is byte-identical to the first run's."""
    import importlib
    import pickle
    from safetensors.torch import save_file
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    sd = la.synth.encoder_state_dict(0)
    ckpt = tmp_path / "speecht5_asr"
    os.makedirs(ckpt)
    save_file({"speecht5.encoder." + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, str(ckpt / "model.safetensors"))
    common = ["-m", "audio", "-s", "devel", "--synthetic", "9", "--synthetic-seconds", "1.5"]
    a, b = str(tmp_path / "from_dir"), str(tmp_path / "random_init")
    extract.main(common + ["--pretrained", str(ckpt), "--out", a])
    extract.main(common + ["--random-init", "--out", b])          # synth.encoder_state_dict(0): the same weights
    fa, fb = os.path.join(a, "devel", "audio"), os.path.join(b, "devel", "audio")
    names = sorted(os.listdir(fa))
    assert len(names) == 9 and names == sorted(os.listdir(fb))
    blobs = {n: open(os.path.join(fa, n), "rb").read() for n in names}
    for n in names:
        assert blobs[n] == open(os.path.join(fb, n), "rb").read(), n
    # resume: remove members of batches (2,3) and (6,7) and the single (8): three batches are re-encoded, two are skipped
    for n in (names[2], names[7], names[8]):
        os.remove(os.path.join(fa, n))
    mt = {n: os.path.getmtime(os.path.join(fa, n)) for n in (names[0], names[1], names[4], names[5])}
    st = extract.main(common + ["--pretrained", str(ckpt), "--out", a, "--resume"])
    assert st["utterances"] == 5
    assert sorted(os.listdir(fa)) == names
    for n in names:
        assert open(os.path.join(fa, n), "rb").read() == blobs[n], n
    for n, t in mt.items():
        assert os.path.getmtime(os.path.join(fa, n)) == t, n      # untouched


def test_extract_cli_precision_and_range_policy_flags(tmp_path):
    """--precision f32 runs the exact-fp32 kernel set (bit-identical to the module with precision="f32"), the default is f16x3, and
    the two agree to 5e-6; --range-policy is handed to the encoder."""
    la = importlib.import_module("loco-asr_amd")
    extract = importlib.import_module("loco-asr_amd.extract")
    import torch
    common = ["-m", "audio", "-s", "devel", "--synthetic", "4", "--synthetic-seconds", "1.5", "--random-init", "--inflight", "1"]
    outs = {}
    for name, extra in (("default", []), ("f32", ["--precision", "f32", "--range-policy", "raise"])):
        out = str(tmp_path / name)
        extract.main(common + ["--out", out] + extra)
        outs[name] = out
    sd = la.synth.encoder_state_dict(0)
    pre, enc_sd = la.synth.split_state_dict(sd)
    m = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                         {k: torch.from_numpy(v) for k, v in enc_sd.items()}).cuda()
    m.speecht5.encoder.precision = "f32"
    lens = la.synth.mixed_lengths(4, int(1.5 * 16000))
    fe = la.SpeechT5FeatureExtractorMI355X()
    for idx in ([0, 1], [2, 3]):
        b = fe(audio=[la.synth.clip(i, lens[i]) for i in idx], sampling_rate=16000, return_tensors="pt")
        ref = m.speecht5.encoder(input_values=b["input_values"].cuda(), attention_mask=b["attention_mask"].cuda()).last_hidden_state.cpu().numpy()
        for row, i in enumerate(idx):
            emb = {}
            for name in outs:
                with open(os.path.join(outs[name], "devel", "audio", f"synthetic-{i:06d}_embedding_and_target.pickle"), "rb") as fh:
                    emb[name] = pickle.load(fh)["embedding"]
            assert np.array_equal(emb["f32"], ref[row])
            assert not np.array_equal(emb["default"], emb["f32"])
            assert np.linalg.norm(emb["default"].astype(np.float64) - emb["f32"]) / np.linalg.norm(emb["f32"].astype(np.float64)) < 5e-6
