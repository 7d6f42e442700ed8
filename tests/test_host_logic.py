"""Host-side pieces that need no GPU: synthetic data determinism, the feature-extractor mirror, the pickle sink
and its reader, unit sharding."""
import importlib
import os
import pickle
import sys

import numpy as np
import pytest
import torch

la = importlib.import_module("loco-asr_amd")
dp = importlib.import_module("loco-asr_amd.dp")
sink = importlib.import_module("loco-asr_amd.sink")
extract = importlib.import_module("loco-asr_amd.extract")


def test_synth_is_pinned():
    """The hash generator must never drift: goldens depend on it."""
    u = la.synth.hashed_uniform("pin", (4,), 0)
    assert u.dtype == np.float32
    np.testing.assert_array_equal(u, la.synth.hashed_uniform("pin", (4,), 0))
    assert not np.array_equal(u, la.synth.hashed_uniform("pin", (4,), 1))
    sd = la.synth.encoder_state_dict(0, layers=1)
    w = sd["wrapped_encoder.layers.0.attention.q_proj.weight"]
    assert w.shape == (768, 768) and abs(float(w.std()) - 1.5 / np.sqrt(768)) < 2e-3
    c = la.synth.clip(0, 1000)
    assert c.dtype == np.float32 and c.shape == (1000,)
    np.testing.assert_array_equal(c, la.synth.clip(0, 1000))
    assert la.synth.mixed_lengths(3, 480000) == la.synth.mixed_lengths(3, 480000)
    # first values pinned (a change here invalidates every fixture in tests/golden/)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g1_1s.npz"))
    assert int(g["lengths"][0]) == 16000


def test_feature_extractor_pads_and_masks():
    fe = la.SpeechT5FeatureExtractorMI355X()
    clips = [la.synth.clip(0, 1000), la.synth.clip(1, 640)]
    out = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
    assert out["input_values"].shape == (2, 1000) and out["input_values"].dtype == torch.float32
    assert out["attention_mask"].dtype == torch.int32
    assert out.attention_mask.sum(1).tolist() == [1000, 640]
    assert float(out.input_values[1, 640:].abs().max()) == 0.0
    np.testing.assert_array_equal(out.input_values[1, :640].numpy(), clips[1])
    with pytest.raises(ValueError):
        fe(audio=clips, sampling_rate=8000)
    with pytest.raises(ValueError):
        fe(audio=None)
    kw = dict(**out)  # usable as encoder(**audios)
    assert set(kw) == {"input_values", "attention_mask"}


def test_feature_extractor_matches_hf_when_available():
    tr = pytest.importorskip("transformers")
    clips = [la.synth.clip(0, 2000), la.synth.clip(1, 1234), la.synth.clip(2, 777)]
    for norm in (False, True):
        hf = tr.SpeechT5FeatureExtractor(do_normalize=norm)(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
        ours = la.SpeechT5FeatureExtractorMI355X(do_normalize=norm)(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
        assert torch.equal(hf["attention_mask"].to(torch.int32), ours["attention_mask"])
        torch.testing.assert_close(hf["input_values"], ours["input_values"], rtol=1e-6, atol=1e-6)


def test_sink_writes_the_reference_format(tmp_path):
    emb = torch.randn(3, 7, 768)
    tg = np.eye(101, dtype=np.int64)[[5, 0, 100]]
    with sink.EmbeddingSink(str(tmp_path), "devel", "audio") as s:
        s.submit(["a-1", "b-2", 33], emb, tg)
    folder = tmp_path / "devel" / "audio"
    assert sorted(os.listdir(folder)) == ["33_embedding_and_target.pickle", "a-1_embedding_and_target.pickle", "b-2_embedding_and_target.pickle"]
    with open(folder / "b-2_embedding_and_target.pickle", "rb") as fh:
        d = pickle.load(fh)
    assert set(d) == {"id", "embedding", "target"} and d["id"] == "b-2"
    assert d["embedding"].dtype == np.float32 and d["embedding"].shape == (7, 768)
    np.testing.assert_array_equal(d["embedding"], emb[1].numpy())
    assert d["target"].shape == (101,) and d["target"][0] == 1
    # the reader restating SLURPEmbeddingsTargets (slurp_embeddings_and_targets.py:19-28) consumes it
    ds = sink.EmbeddingsTargets(str(tmp_path), modality="audio", split="devel")
    assert len(ds) == 3
    sid, e, t = ds[0]
    assert torch.is_tensor(e) and e.shape == (7, 768) and t.shape == (101,)


def test_sink_npy_format(tmp_path):
    with sink.EmbeddingSink(str(tmp_path), "test", "audio", fmt="npy") as s:
        s.submit(["x"], torch.ones(1, 2, 768), [np.zeros(101, np.int64)])
    assert np.load(tmp_path / "test" / "audio" / "x.embedding.npy").shape == (2, 768)


def test_one_hot_encoder_matches_sklearn():
    sk = pytest.importorskip("sklearn.preprocessing")
    classes = ["b", "a", "d", "c", "e"]
    le = sk.LabelEncoder()
    lb = sk.LabelBinarizer()
    lb.fit_transform(le.fit_transform(classes))
    labels = ["c", "a", "e"]
    np.testing.assert_array_equal(extract.one_hot_encoder(classes)(labels), lb.transform(le.transform(labels)))


def test_shard_units_balances_and_partitions():
    lengths = [480000, 16000, 300000, 80000, 80000, 9600000, 1000, 48000, 123456]
    for world in (1, 2, 3, 8):
        shards = [dp.shard_units(lengths, world, r) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    # longest first, dealt round-robin: the two longest land on different ranks
    s0, s1 = dp.shard_units(lengths, 2, 0), dp.shard_units(lengths, 2, 1)
    assert (5 in s0) != (0 in s0)


def test_shard_batches_keeps_the_reference_pairs():
    """DataLoader(batch_size=2, shuffle=False) forms (0,1), (2,3), ...; under DP whole pairs are dealt, never split or re-paired."""
    for n in (1, 2, 7, 8, 13):
        ref_pairs = [list(range(a, min(n, a + 2))) for a in range(0, n, 2)]
        assert dp.corpus_batches(n, 2) == ref_pairs
        for world in (1, 2, 3, 8):
            dealt = [dp.shard_batches(n, 2, world, r) for r in range(world)]
            assert sorted(b for d in dealt for b in d) == ref_pairs
            assert all(b in ref_pairs for d in dealt for b in d)
            assert max(len(d) for d in dealt) == dp.rounds(n, 2, world)
            assert max(map(len, dealt)) - min(map(len, dealt)) <= 1


def test_intent_targets_use_the_fixed_101_classes_for_every_split(tmp_path):
    """targets are one-hot over the reference's ALL_CLASSES (…base…py:32-36), whatever labels a split contains."""
    classes = extract.load_classes()
    assert len(classes) == 101 and len(set(classes)) == 101 and "alarm_set" in classes
    enc = extract.one_hot_encoder(classes)
    train_labels, devel_labels = ["alarm_set", "weather_query"], ["weather_query", "qa_factoid", "alarm_set"]
    a, b = enc(train_labels), enc(devel_labels)
    assert a.shape == (2, 101) and b.shape == (3, 101) and a.dtype == np.int64
    assert (a[0] == b[2]).all() and (a[1] == b[0]).all()  # same label -> same column in every split
    assert int(a[0].argmax()) == sorted(classes).index("alarm_set")  # LabelEncoder sorts
    with pytest.raises(ValueError, match="unseen"):
        enc(["not_an_intent"])
    bad = tmp_path / "classes.txt"
    bad.write_text("a\nb\n")
    with pytest.raises(SystemExit, match="101"):
        extract.load_classes(str(bad))


def test_train_head_rank_strided_batches():
    """Under DP each rank reads only its own batches of the epoch's permutation; together they are the first floor(nb/W)*W."""
    th = importlib.import_module("loco-asr_amd.train_head")
    n, bs = 103, 16
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0)).tolist()
    full = [perm[a:a + bs] for a in range(0, n, bs)]  # 7 batches, the last one short
    ids, b = th.epoch_batches(n, bs, 1, 0, torch.Generator().manual_seed(0))
    assert ids == list(range(7)) and b == full
    for world in (2, 3, 8):
        got = [th.epoch_batches(n, bs, world, r, torch.Generator().manual_seed(0)) for r in range(world)]
        assert len({len(i) for i, _ in got}) == 1  # same number of steps on every rank
        merged = sorted((i, tuple(x)) for ids_, bs_ in got for i, x in zip(ids_, bs_))
        keep = (7 // world) * world
        assert merged == [(i, tuple(full[i])) for i in range(keep)]


def test_slurp_reader_contract(tmp_path):
    root = tmp_path / "slurp"
    (root / "dataset" / "slurp").mkdir(parents=True)
    (root / "audio" / "slurp_real").mkdir(parents=True)
    rows = [{"slurp_id": 7, "sentence": "wake me at five", "intent": "alarm_set",
             "recordings": [{"file": "audio-1.flac"}, {"file": "audio-1-headset.flac", "headset": True}]},
            {"slurp_id": 9, "sentence": "stop", "intent": "audio_volume_mute", "recordings": [{"file": "audio-2.flac"}]}]
    import json
    (root / "dataset" / "slurp" / "devel.jsonl").write_text("\n".join(json.dumps(r) for r in rows) + "\n")
    items = extract.read_slurp_split(str(root), "devel")
    assert [i[0] for i in items] == [7, 9]
    assert items[0][2].endswith("audio-1-headset.flac") and items[1][2].endswith("audio-2.flac")
    assert items[0][3] == 16000 and items[0][4] == "alarm_set"


def test_fairseq_checkpoint_mapping_round_trip():
    """map_speecht5_hf.py equivalent: fairseq key names -> the two HF-named dicts, nothing lost, nothing invented."""
    cm = importlib.import_module("loco-asr_amd.checkpoint_map")
    sd = la.synth.encoder_state_dict(0, layers=2)
    pre, enc = la.synth.split_state_dict(sd)
    ckpt = cm.to_fairseq_names(pre, enc)
    ckpt["decoder.layers.0.fc1.weight"] = np.zeros(1)  # other sub-modules are ignored
    ckpt["text_encoder_prenet.encoder_prenet.0.weight"] = np.zeros(1)
    assert "encoder.layers.1.self_attn.q_proj.weight" in ckpt and "encoder.layers.0.fc2.bias" in ckpt
    assert "speech_encoder_prenet.feature_extractor.conv_layers.0.2.weight" in ckpt
    assert "speech_encoder_prenet.pos_conv.0.weight_g" in ckpt and "speech_encoder_prenet.mask_emb" in ckpt
    enc2, pre2, unmapped = cm.map_fairseq_speecht5(ckpt)
    assert unmapped == []
    assert set(enc2) == set(enc)
    legacy = {k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v") for k in pre}
    assert set(pre2) == legacy
    for k in enc:
        assert enc2[k] is enc[k]
    # the mapped dicts load through the same two load_state_dict calls the reference makes
    m = la.SpeechT5ForSpeechToTextMI355X(2)
    r1 = m.speecht5.encoder.wrapped_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc2.items()})
    r2 = m.speecht5.encoder.prenet.load_state_dict({k: torch.from_numpy(v) for k, v in pre2.items()})
    assert not r1.missing_keys and not r1.unexpected_keys and not r2.missing_keys and not r2.unexpected_keys
    _, _, unm = cm.map_fairseq_speecht5({"encoder.layers.0.mystery.weight": 1})
    assert unm == ["encoder.layers.0.mystery.weight"]


def test_checkpoint_mapping_matches_the_reference_mapping_class():
    """f-3 pinned: fixture g9 holds the key tables the reference's own `Mapping` (map_speecht5_hf.py) produced for a synthetic
    fairseq-named checkpoint (tests/golden/make_mapping_goldens.py); the rename tables here must produce the same three dicts."""
    import json
    cm = importlib.import_module("loco-asr_amd.checkpoint_map")
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g9_mapping.json")))
    ckpt = {k: ("ckpt", k) for k in g["fairseq_keys"]}  # value = where it came from
    enc, pre, unmapped = cm.map_fairseq_speecht5(ckpt)
    ref = g["hf_4_30_2"]  # the reference's pinned transformers spelling
    assert {k: v[1] for k, v in enc.items()} == ref["encoder_state_dict"]
    assert {k: v[1] for k, v in pre.items()} == ref["speech_prenet_state_dict"]
    # keys the reference has no rule for are dropped there silently (encoder.version: `else: continue`, map_speecht5_hf.py:77;
    # the fairseq sinusoid buffer matches no branch of map_speech_prenet); here they are reported instead
    assert sorted(unmapped) == ["encoder.version", "speech_encoder_prenet.pos_sinusoidal_embed._float_tensor"]
    assert ref["encoder_unmatched"] == []
    # against transformers 5.x the reference's weight-norm rule finds no partner: exactly those two keys are missing there
    missing = set(ref["speech_prenet_state_dict"]) - set(g["hf_installed"]["speech_prenet_state_dict"])
    assert missing == {"pos_conv_embed.conv.weight_g", "pos_conv_embed.conv.weight_v"}
    assert g["hf_installed"]["encoder_state_dict"] == ref["encoder_state_dict"]
    # text prenet dict: embed_tokens from the checkpoint, alpha and pe from the HF TTS model (map_speecht5_hf.py:168-181)
    tts = {"embed_tokens.weight": ("model", "embed_tokens.weight"), "encode_positions.alpha": ("model", "alpha"),
           "encode_positions.pe": ("model", "pe")}
    txt = cm.map_text_prenet(ckpt, tts)
    want = ref["text_prenet_state_dict"]
    assert set(txt) == set(want)
    for k, src in want.items():
        assert (txt[k][0] == "ckpt" and txt[k][1] == src) if src.startswith("text_encoder_prenet") else txt[k][0] == "model", k
    assert cm.map_text_prenet(ckpt)["encode_positions.alpha"] == ("ckpt", "text_encoder_prenet.encoder_prenet.1.alpha")


def test_window_units_for_long_recordings():
    """configs[3]: 60-minute recordings -> 10-minute windows as independent units."""
    hour, ten = 60 * 60 * 16000, 10 * 60 * 16000
    u = dp.window_units([hour, ten + 5, 399, 2 * ten + 400], ten)
    assert [x for x in u if x[0] == 0] == [(0, k * ten, (k + 1) * ten) for k in range(6)]
    assert [x for x in u if x[0] == 1] == [(1, 0, ten)]  # the 5-sample tail is shorter than a frame: dropped
    assert [x for x in u if x[0] == 2] == []
    assert [x for x in u if x[0] == 3] == [(3, 0, ten), (3, ten, 2 * ten), (3, 2 * ten, 2 * ten + 400)]
    shards = [dp.shard_units([b - a for _, a, b in u], 8, r) for r in range(8)]
    assert sorted(i for s in shards for i in s) == list(range(len(u)))


def _g7_clips():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_normalize.npz"))
    clips = [(la.synth.clip(i, int(n)) * np.float32(0.5 + i) + np.float32(0.1 * i - 0.15)).astype(np.float32) for i, n in enumerate(g["lengths"])]
    return g, clips


def test_feature_extractor_do_normalize_matches_hf_golden():
    """do_normalize=True on the host against HF's SpeechT5FeatureExtractor output (fixture g7)."""
    g, clips = _g7_clips()
    fe = la.SpeechT5FeatureExtractorMI355X(do_normalize=True)
    out = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
    x = out["input_values"].numpy()
    assert x.shape == (4, 16000) and out["attention_mask"].sum(1).tolist() == g["mask_sums"].tolist()
    np.testing.assert_allclose(x[:, g["cols"]], g["values"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(x[0], g["first_clip"], rtol=2e-6, atol=2e-6)


def test_feature_extractor_defers_normalisation_to_the_device():
    """normalize_on_device: the host only pads; the debt travels with the BatchFeature until a GPU pays it."""
    g, clips = _g7_clips()
    fe = la.SpeechT5FeatureExtractorMI355X(do_normalize=True, normalize_on_device=True)
    out = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
    assert out._pending_normalize == 0.0
    np.testing.assert_array_equal(out["input_values"][1, :12345].numpy(), clips[1])  # untouched on the host
    assert out.to("cpu")._pending_normalize == 0.0
    with pytest.raises(ValueError):
        fe(audio=clips, sampling_rate=16000, return_tensors="np")


def test_bench_starts_its_own_ranks_when_no_launcher_is_present(monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset must start `python -m torch.distributed.run --nproc-per-node N ... bench.py
    --gpus N ...` as a CHILD process (the shape of the driver's own command) and exit with the child's code --
    before anything in the parent touches the GPU."""
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the parent must not touch the GPU")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    # under a launcher whose world size disagrees, it is an error, not a second launch
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="must agree"):
        bench.main()


def test_loader_threads_resample_on_the_ranks_own_gpu(monkeypatch, tmp_path):
    """fetch() runs on ThreadPoolExecutor threads, where torch.cuda.current_device() is 0 whatever the main thread
    selected -- every rank > 0 would resample on GPU 0.  The rank's device must be handed down explicitly."""
    from concurrent.futures import ThreadPoolExecutor
    from scipy.io import wavfile
    extract = importlib.import_module("loco-asr_amd.extract")
    resample = importlib.import_module("loco-asr_amd.resample")
    path = str(tmp_path / "a.wav")
    wavfile.write(path, 8000, (np.sin(np.arange(8000) * 0.01) * 2000).astype(np.int16))
    seen = []

    def fake(x, sr, device=None):
        seen.append((int(sr), device, len(x)))
        return torch.zeros(2 * len(x))

    monkeypatch.setattr(resample, "resample_to_16k", fake)
    rank_device = torch.device("cuda", 5)
    with ThreadPoolExecutor(2) as pool:
        y = list(pool.map(lambda _: extract.load_audio_16k(path, rank_device), range(3)))
    assert len(y) == 3 and seen == [(8000, rank_device, 8000)] * 3


def test_train_head_batch_dealing_partitions_every_epoch_and_the_validation_set():
    """Data-parallel head training (configs[4]): every rank draws the SAME epoch permutation and takes batches rank, rank + W, ...
    (an incomplete last round is dropped so that the gradient all-reduces line up); validation batches are dealt the same way and
    the two sums meet in one all-reduce.  Together the ranks' shares must tile the data exactly once."""
    th = importlib.import_module("loco-asr_amd.train_head")
    n, bs, W = 103, 16, 4
    seen = []
    for rank in range(W):
        g = torch.Generator().manual_seed(0)
        ids, batches = th.epoch_batches(n, bs, W, rank, g)
        assert ids == list(range(rank, (7 // W) * W, W))  # 7 batches -> one full round of 4
        seen += [i for b in batches for i in b]
    assert len(seen) == len(set(seen)) == 4 * 16
    val = [i for rank in range(W) for b in th.strided_batches(n, bs, W, rank) for i in b]
    assert sorted(val) == list(range(n))
    assert th.strided_batches(n, bs, 1, 0) == [list(range(a, min(n, a + bs))) for a in range(0, n, bs)]


@pytest.mark.parametrize("normalize", [False, True])
def test_pack_clips_is_the_processor_batch_by_batch(normalize):
    """feature_extractor.pack_clips (extract.py --pack: every decoded clip written once into the pack's [B, L] buffer) must hold, for
    every reference batch, exactly what SpeechT5FeatureExtractor's __call__ gives for that batch alone (…base…py:60: padding="longest",
    zeros, optional zero-mean / unit-variance over the unpadded samples) -- the pack only adds the bookkeeping: per clip the padded
    length of ITS batch, its own sample count (what HF reduces the attention mask to), per batch the span of rows and output frames."""
    la = importlib.import_module("loco-asr_amd")
    fe = la.SpeechT5FeatureExtractorMI355X(do_normalize=normalize)
    rng = np.random.default_rng(3)
    lens = [[40000, 23000], [400, 31999], [16000], [9000, 9000, 12000], [52001, 800]]
    batches = [[(la.synth.clip(10 * i + j, n) * np.float32(0.5 + j) + np.float32(0.1 * i)).astype(np.float32) for j, n in enumerate(b)] for i, b in enumerate(lens)]
    pack = fe.pack_clips(batches, "cpu", la.synth.conv_out_length)
    assert pack.mask is None and pack.wav.dtype == torch.float32 and pack.wav.shape[1] % 8 == 0
    assert pack.wav.shape == (sum(len(b) for b in lens), (max(max(b) for b in lens) + 7) // 8 * 8)
    b0 = 0
    for b, clips, span in zip(lens, batches, pack.spans):
        ref = fe(audio=clips, sampling_rate=16000, return_tensors="pt", padding="longest")
        nb, li = ref["input_values"].shape
        assert span == (b0, nb, la.synth.conv_out_length(li)) and li == max(b)
        assert torch.equal(pack.wav[b0:b0 + nb, :li], ref["input_values"])
        assert pack.pad_len[b0:b0 + nb] == [li] * nb
        assert pack.valid_len[b0:b0 + nb] == ref["attention_mask"].sum(1).tolist() == list(b)
        b0 += nb
    with pytest.raises(ValueError, match="shorter than one encoder frame"):
        fe.pack_clips([[np.zeros(399, np.float32)]], "cpu", la.synth.conv_out_length)
    with pytest.raises(ValueError, match="empty batch"):
        fe.pack_clips([[]], "cpu", la.synth.conv_out_length)


def test_attention_band_scratch_layout_is_a_bijection_without_bank_conflicts():
    """attention_f16x3.hip / attention_f32.hip, AX_BAND_ADD: the per-wave transpose scratch of the relative-position band stores
    element (query row, key column) at 16 row + row // 2 + column.  The layout must be a bijection inside the 32 x 17 floats the
    kernels allot, and -- ds_write_b32 / ds_read_b32 being served 32 lanes at a time over 32 banks -- every write group (two
    neighbouring rows x 16 columns; the kernel's address is 66 u + 16 lq + lq // 2 + lj for row 4 u + lq) and every read group
    (one column of the 32 rows) must touch 32 different banks."""
    addr = lambda row, col: 16 * row + (row >> 1) + col
    cells = {addr(r, c) for r in range(32) for c in range(16)}
    assert len(cells) == 512 and max(cells) < 32 * 17
    for u in range(8):
        for lq in range(4):
            for lj in range(16):
                assert 66 * u + 16 * lq + (lq >> 1) + lj == addr(4 * u + lq, lj)
        for pair in (0, 2):
            assert len({addr(4 * u + pair + i, lj) % 32 for i in range(2) for lj in range(16)}) == 32
    for col in range(16):
        assert len({addr(r, col) % 32 for r in range(32)}) == 32
    # the layout of rounds 1-3 (rows padded to 17 floats), for the record: reads conflict-free, every write group one bank short
    old = lambda row, col: 17 * row + col
    assert len({old(r, 0) % 32 for r in range(32)}) == 32
    assert len({old(i, lj) % 32 for i in range(2) for lj in range(16)}) == 31


def test_every_tool_script_still_parses():
    """tools/ holds the probes DESIGN.md and profiles/ quote; most need a GPU to RUN, but a syntax error or a stale shell construct in
    one of them should not wait for the next time somebody needs the measurement."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "**", "*.py"), recursive=True)) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(scripts) > 30
    for path in scripts:
        with open(path) as fh:
            compile(fh.read(), path, "exec")  # syntax only: nothing is written, nothing runs
    for path in sorted(glob.glob(os.path.join(root, "tools", "**", "*.sh"), recursive=True)):
        subprocess.run(["bash", "-n", path], check=True)
