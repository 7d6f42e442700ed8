""""Next" row f-4 on the GPU: the text branch of the reference (`model.speecht5.encoder(texts.input_ids)`,
/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:80-86) through loco_forward_text, against the
HF-generated golden g6 and the oracle.  Same bars as the speech path: golden rows within 2e-5 (f16x3 and f32 modes)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import check, la, lib, ptr, rel_l2, stream

TOL = 2e-5
_cache = {}


def text_model(layers=12, precision="f16x3"):
    if layers not in _cache:
        sd = la.synth.encoder_state_dict(0, layers)
        _, enc = la.synth.split_state_dict(sd)
        tsd = la.synth.text_prenet_state_dict(0)
        pre = {k[len("text_prenet."):]: torch.from_numpy(np.asarray(v)) for k, v in tsd.items()}
        pre["encode_positions.pe"] = la.scaled_positional_table(450)[None]  # the reference's pickled dict carries it
        m = la.SpeechT5ForTextToSpeechMI355X.from_state_dicts(pre, {k: torch.from_numpy(v) for k, v in enc.items()}, layers=layers)
        full = dict(sd)
        full.update(tsd)
        _cache[layers] = (m.to("cuda"), full)
    m, sd = _cache[layers]
    m.speecht5.encoder.precision = precision
    return m, sd


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_golden_g6_without_and_with_mask(precision):
    g = golden("g6_text.npz")
    m, _ = text_model(precision=precision)
    enc = m.speecht5.encoder
    ids, mask = la.synth.token_ids(3, 57, lengths=list(g["lengths"]))
    idt, mt = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    a = enc(idt, output_hidden_states=True)  # positional, ids only: exactly the reference's call
    assert tuple(a.last_hidden_state.shape) == (3, 57, 768) and len(a.hidden_states) == 13
    rows = g["short_rows"]
    assert rel_l2(a.last_hidden_state[:, rows], g["nomask_last"]) < TOL
    for i, h in enumerate(a.hidden_states):
        assert abs(float(h.double().norm()) / g["nomask_hidden_stats"][i, 0] - 1) < 2e-5, i
    assert torch.equal(a.hidden_states[-1], a.last_hidden_state)
    b = enc(input_values=idt, attention_mask=mt)
    assert rel_l2(b.last_hidden_state, g["masked_last"]) < TOL
    assert enc.last_frames.tolist() == list(g["lengths"])
    long_ids, _ = la.synth.token_ids(1, 450, seed=11)
    c = enc(torch.from_numpy(long_ids).cuda()).last_hidden_state
    assert rel_l2(c[:, g["long_rows"]], g["long_last"]) < TOL
    assert abs(float(c.double().norm()) / g["long_stats"][0] - 1) < 2e-5


def test_against_fp64_oracle_and_edge_shapes(oracle):
    m, sd = text_model()
    enc = m.speecht5.encoder
    for B, T, lengths in ((1, 1, None), (2, 64, [64, 1]), (5, 129, [129, 128, 65, 64, 3])):
        ids, mask = la.synth.token_ids(B, T, seed=3, lengths=lengths)
        kw = {} if lengths is None else {"attention_mask": torch.from_numpy(mask).cuda()}
        y = enc(torch.from_numpy(ids).cuda(), **kw).last_hidden_state
        ref = oracle.encode_text(ids, None if lengths is None else mask, sd, dtype=torch.float64)
        assert rel_l2(y, ref) < 1e-5, (B, T)
    # text prenet alone is exact: gather + one multiply + one add (no FMA contraction)
    ids, _ = la.synth.token_ids(2, 33, seed=5)
    m1, sd1 = text_model(layers=0)
    y0 = m1.speecht5.encoder(torch.from_numpy(ids).cuda(), output_hidden_states=True)
    pre = oracle.text_prenet(ids, sd1)
    ln = torch.nn.functional.layer_norm(pre, (768,), torch.from_numpy(sd1["wrapped_encoder.layer_norm.weight"]),
                                        torch.from_numpy(sd1["wrapped_encoder.layer_norm.bias"]), 1e-5)
    assert rel_l2(y0.last_hidden_state, ln) < 2e-6


def test_error_contract():
    m, _ = text_model()
    enc = m.speecht5.encoder
    with pytest.raises(IndexError):
        enc(torch.full((1, 4), 81, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 451, dtype=torch.long, device="cuda") + 5)
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 4, device="cuda"))  # float "ids"
    with pytest.raises(NotImplementedError):
        enc(torch.full((1, 4), 5, dtype=torch.long, device="cuda"), attention_mask=torch.tensor([[1, 0, 1, 1]], device="cuda"))
    with pytest.raises(RuntimeError):
        enc(torch.full((1, 4), 5, dtype=torch.long))  # CPU tensor: there is no CPU path
    # a text-only handle has no speech prenet: the speech entry point must refuse, not crash
    h = enc._handle
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    x = torch.zeros(1, 16000, device="cuda")
    out = torch.empty(1, 49, 768, device="cuda")
    rc = lib().loco_forward(h, ptr(x), None, 1, 16000, ptr(out), None, None, ptr(ws), ws.numel(), stream())
    assert rc != 0 and b"speech prenet" in lib().loco_last_error()
    # and a speech handle refuses text
    from gpu_util import model
    sm, _ = model(layers=1)
    sm.speecht5.encoder(input_values=torch.zeros(1, 16000, device="cuda"))
    ids = torch.zeros(1, 4, dtype=torch.int32, device="cuda")
    rc = lib().loco_forward_text(sm.speecht5.encoder._handle, ptr(ids), None, 1, 4, ptr(out), None, None, ptr(ws), ws.numel(), stream())
    assert rc != 0 and b"text prenet" in lib().loco_last_error()


def test_text_forwards_in_flight_equal_one_at_a_time_bitwise(tmp_path):
    """The text branch of the reference's loop is batch_size = 2 as well (…base…py:67-68, 79-93): loco_forward_text_async /
    forward_async keep several batches of transcripts in flight; results equal the synchronous forwards bit for bit, with and without a
    padding mask, and the CLI writes byte-identical pickles at --inflight 1 and 4."""
    import importlib
    import os
    la = importlib.import_module("loco-asr_amd")
    sd = la.synth.encoder_state_dict(0, layers=2)
    _, enc_sd = la.synth.split_state_dict(sd)
    tpre = {k[len("text_prenet."):]: torch.from_numpy(np.asarray(v)) for k, v in la.synth.text_prenet_state_dict(0).items()}
    model = la.SpeechT5ForTextToSpeechMI355X.from_state_dicts(tpre, {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=2).cuda()
    enc = model.speecht5.encoder
    batches = []
    for i, (n, lens) in enumerate(((57, [57, 31]), (20, [20, 20]), (90, [44, 90]), (9, [9, 3]), (33, [33, 30]))):
        ids, mask = la.synth.token_ids(2, n, seed=20 + i, lengths=lens)
        batches.append((torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda() if i % 2 else None))
    ref = [enc(x, attention_mask=m).last_hidden_state.clone() for x, m in batches]
    for k in (2, 4):
        enc.set_inflight(k)
        tickets = [enc.forward_async(x, attention_mask=m) for x, m in batches]
        for t, r in zip(tickets, ref):
            assert torch.equal(t.result().last_hidden_state, r) and not t.used_fp32
    extract = importlib.import_module("loco-asr_amd.extract")
    folders = {}
    for k in (1, 4):
        out = str(tmp_path / f"k{k}")
        extract.main(["-m", "text", "-s", "devel", "--synthetic", "9", "--random-init", "--out", out, "--inflight", str(k)])
        folders[k] = os.path.join(out, "devel", "text")
    names = sorted(os.listdir(folders[1]))
    assert len(names) == 9 and names == sorted(os.listdir(folders[4]))
    for n in names:
        assert open(os.path.join(folders[1], n), "rb").read() == open(os.path.join(folders[4], n), "rb").read(), n


def test_text_batches_packed_into_one_forward(tmp_path):
    """Several of the reference's text batches (ids padded to the batch's longest, NO mask: the pads of a batch attend like tokens,
    …base…py:79-93) as ONE forward whose key mask ends every row where its own batch ends (text_encoder.forward_packed): equal to the
    batches' own forwards up to the fp32 summation order of the GEMMs; the CLI's --pack writes the same files."""
    import importlib
    import os
    import pickle
    sd = la.synth.encoder_state_dict(0, layers=2)
    _, enc_sd = la.synth.split_state_dict(sd)
    tpre = {k[len("text_prenet."):]: torch.from_numpy(np.asarray(v)) for k, v in la.synth.text_prenet_state_dict(0).items()}
    model = la.SpeechT5ForTextToSpeechMI355X.from_state_dicts(tpre, {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=2).cuda()
    enc = model.speecht5.encoder
    batches = []
    for i, (n, lens) in enumerate(((57, [57, 31]), (20, [20, 20]), (90, [44, 90]), (9, [9, 3]), (33, [33, 30]), (64, [64, 1]))):
        ids, mask = la.synth.token_ids(2, n, seed=40 + i, lengths=lens)
        # the reference's way (no mask) for most batches, a right-padding mask for two of them
        batches.append(dict(input_values=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)) if i in (2, 4) else torch.from_numpy(ids))
    ref = []
    for b in batches:
        if isinstance(b, dict):
            ref.append(enc(b["input_values"].cuda(), attention_mask=b["attention_mask"].cuda()).last_hidden_state.clone())
        else:
            ref.append(enc(b.cuda()).last_hidden_state.clone())
    t = enc.forward_packed_async(batches)
    outs = t.result()
    assert len(outs) == len(batches) and not t.used_fp32
    for o, r in zip(outs, ref):
        assert tuple(o.last_hidden_state.shape) == tuple(r.shape)
        assert rel_l2(o.last_hidden_state, r) < 5e-6
    extract = importlib.import_module("loco-asr_amd.extract")
    folders = {}
    for name, extra in (("one", ["--inflight", "1"]), ("packed", ["--pack", "3"])):
        out = str(tmp_path / name)
        extract.main(["-m", "text", "-s", "devel", "--synthetic", "11", "--random-init", "--out", out] + extra)
        folders[name] = os.path.join(out, "devel", "text")
    names = sorted(os.listdir(folders["one"]))
    assert len(names) == 11 and names == sorted(os.listdir(folders["packed"]))
    for n in names:
        with open(os.path.join(folders["one"], n), "rb") as f1, open(os.path.join(folders["packed"], n), "rb") as f2:
            a, b = pickle.load(f1), pickle.load(f2)
        assert a["id"] == b["id"] and a["embedding"].shape == b["embedding"].shape and (a["target"] == b["target"]).all()
        assert rel_l2(b["embedding"], a["embedding"]) < 5e-6
