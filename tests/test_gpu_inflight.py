"""Several forwards of one handle in flight (include/loco_asr.h, loco_forward_async; encoder.forward_async; extract.py
--inflight): the reference's batches of two utterances (…base…py:67-68) stay what they are -- batch composition is part of the
function -- but nothing orders batch k+1 behind batch k.  Each forward in flight owns a stream, a workspace and a status block;
results must equal the one-at-a-time forwards BIT FOR BIT, and an out-of-range batch must still be reported (and re-run in
fp32) when later batches were enqueued behind it without any synchronisation."""
import importlib
import os
import pickle
import threading

import numpy as np
import pytest
import torch

from gpu_util import la, model, rel_l2

pytestmark = pytest.mark.gpu


def _pairs(n_pairs, seconds=2.0, seed_base=0):
    n = int(seconds * 16000)
    lens = la.synth.mixed_lengths(2 * n_pairs, n)
    fe = la.SpeechT5FeatureExtractorMI355X()
    out = []
    for p in range(n_pairs):
        b = fe(audio=[la.synth.clip(seed_base + 2 * p + j, lens[2 * p + j]) for j in (0, 1)], sampling_rate=16000, return_tensors="pt")
        out.append((b["input_values"].cuda(), b["attention_mask"].cuda()))
    return out


def test_forwards_in_flight_equal_one_at_a_time_bitwise():
    m, _ = model()
    enc = m.speecht5.encoder
    batches = _pairs(10)
    ref = [enc(input_values=x, attention_mask=a).last_hidden_state.clone() for x, a in batches]
    for k in (2, 4, 8):
        enc.set_inflight(k)
        tickets = [enc.forward_async(input_values=x, attention_mask=a) for x, a in batches]  # slots are reused: 10 batches on k slots
        outs = [t.result().last_hidden_state for t in tickets]
        assert not any(t.used_fp32 for t in tickets)
        for i, (o, r) in enumerate(zip(outs, ref)):
            assert torch.equal(o, r), (k, i, rel_l2(o, r))
    enc.set_inflight(1)


def test_an_out_of_range_batch_is_still_reported_with_later_batches_behind_it(oracle):
    """with one status per handle, forward B's range_begin could wipe the maxima forward A had folded, and both
    copied into one pinned buffer.  A model whose FFN intermediate exceeds fp16's maximum (bias + 1e5) makes every f16x3 forward
    leave the range.  (a) three batches enqueued back to back on three streams, no synchronisation in between: every ticket
    reports ITS overflow and hands back the fp32 re-run, equal to the oracle; (b) through the C ABI, forwards at precision 1
    (overflows), 0 (exact fp32: clean) and 1 again, in flight together on one handle: each status block tells its own story."""
    import ctypes as C
    sd = la.synth.encoder_state_dict(0, layers=2)
    ovf = dict(sd)
    k1, k2 = "wrapped_encoder.layers.0.feed_forward.intermediate_dense.bias", "wrapped_encoder.layers.0.feed_forward.output_dense.weight"
    ovf[k1] = sd[k1] + np.float32(1.0e5)
    ovf[k2] = sd[k2] * np.float32(1e-5)
    pre, enc_sd = la.synth.split_state_dict(ovf)
    mo = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                          {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=2).cuda()
    eo = mo.speecht5.encoder
    batches = _pairs(3, seconds=1.0)
    eo.set_inflight(3)
    tickets = [eo.forward_async(input_values=x, attention_mask=a) for x, a in batches]
    outs = [t.result().last_hidden_state for t in tickets]
    assert all(t.used_fp32 for t in tickets)
    for (x, a), o in zip(batches, outs):
        ref = oracle.encode(x.cpu().numpy(), a.cpu().numpy(), ovf)
        assert torch.isfinite(o).all() and rel_l2(o, ref) < 2e-5
    eo.range_policy = "raise"
    t1 = eo.forward_async(input_values=batches[0][0], attention_mask=batches[0][1])
    with pytest.raises(la.LocoError, match="feed_forward intermediate"):
        t1.result()
    eo.range_policy = "fp32"
    eo.drain()
    # (b) raw C ABI
    lib = eo._lib
    nst = int(lib.loco_status_bytes())
    work = []
    for (x, a), prec in zip(batches, (1, 0, 1)):
        a32 = a.to(torch.int32).contiguous()
        B, L = x.shape
        st = torch.cuda.Stream()
        ws = torch.empty(int(lib.loco_workspace_bytes(eo._handle, B, L)), dtype=torch.uint8, device="cuda")
        status = torch.zeros(nst, dtype=torch.uint8).pin_memory()
        out = torch.empty((B, int(lib.loco_output_frames(L)), 768), dtype=torch.float32, device="cuda")
        work.append((x, a32, B, L, st, ws, status, out, prec))
    torch.cuda.synchronize()
    for x, a32, B, L, st, ws, status, out, prec in work:  # enqueue all three before waiting for any
        rc = lib.loco_forward_async(eo._handle, prec, C.c_void_p(x.data_ptr()), C.c_void_p(a32.data_ptr()), B, L, C.c_void_p(out.data_ptr()),
                                    None, None, C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(st.cuda_stream), C.c_void_p(status.data_ptr()))
        assert rc == 0, lib.loco_last_error()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(400)
    codes = [lib.loco_status_check(C.c_void_p(w[6].data_ptr()), buf, 400) for w in work]
    assert codes == [-5, 0, -5], codes
    assert torch.isfinite(work[1][7]).all() and rel_l2(work[1][7], oracle.encode(batches[1][0].cpu().numpy(), batches[1][1].cpu().numpy(), ovf)) < 2e-5
    amax, layer, name = C.c_float(), C.c_int32(), C.create_string_buffer(160)
    n = lib.loco_status_range(C.c_void_p(work[0][6].data_ptr()), 0, C.byref(amax), C.byref(layer), name, 160)
    assert n > 5 and b"conv_layers.0" in name.value and 0.1 < amax.value < 1e4  # conv0's output is a tracked stage now


def test_enqueue_from_two_host_threads():
    """The handle is only read while a forward is enqueued (per-call state lives in the workspace and the status block), so two
    host threads may enqueue forwards of ONE handle at the same time -- here straight through the C ABI, each thread with its own
    stream / workspace / status / output, 6 forwards each -- and get the bits of the single-threaded forwards."""
    import ctypes as C
    m, _ = model()
    enc = m.speecht5.encoder
    batches = _pairs(2, seconds=3.0)
    ref = [enc(input_values=x, attention_mask=a).last_hidden_state.clone() for x, a in batches]
    lib = enc._lib
    nst = int(lib.loco_status_bytes())
    results, errors = {}, []

    def work(tid):
        try:
            torch.cuda.set_device(0)
            x, a = batches[tid]
            a32 = a.to(torch.int32).contiguous()
            B, L = x.shape
            T = int(lib.loco_output_frames(L))
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())  # a32 was produced on this thread's current (default) stream
            ws = torch.empty(int(lib.loco_workspace_bytes(enc._handle, B, L)), dtype=torch.uint8, device="cuda")
            status = torch.zeros(nst, dtype=torch.uint8).pin_memory()
            outs = []
            for _ in range(6):
                out = torch.empty((B, T, 768), dtype=torch.float32, device="cuda")
                rc = lib.loco_forward_async(enc._handle, 1, C.c_void_p(x.data_ptr()), C.c_void_p(a32.data_ptr()), B, L,
                                            C.c_void_p(out.data_ptr()), None, None, C.c_void_p(ws.data_ptr()), ws.numel(),
                                            C.c_void_p(st.cuda_stream), C.c_void_p(status.data_ptr()))
                assert rc == 0, lib.loco_last_error()
                st.synchronize()
                assert lib.loco_status_check(C.c_void_p(status.data_ptr()), None, 0) == 0
                outs.append(out)
            results[tid] = outs
        except BaseException as e:  # noqa: BLE001 -- surfaced below
            errors.append(e)

    torch.cuda.synchronize()
    threads = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for tid in (0, 1):
        for o in results[tid]:
            assert torch.equal(o, ref[tid])


def test_extract_cli_inflight_writes_identical_pickles(tmp_path):
    extract = importlib.import_module("loco-asr_amd.extract")
    outs = {}
    for k in (1, 4):
        out = str(tmp_path / f"k{k}")
        extract.main(["-m", "audio", "-s", "devel", "--synthetic", "13", "--synthetic-seconds", "2.0", "--random-init", "--out", out,
                      "--inflight", str(k)])
        outs[k] = os.path.join(out, "devel", "audio")
    names = sorted(os.listdir(outs[1]))
    assert len(names) == 13 and names == sorted(os.listdir(outs[4]))
    for n in names:
        assert open(os.path.join(outs[1], n), "rb").read() == open(os.path.join(outs[4], n), "rb").read(), n
    with open(os.path.join(outs[4], names[0]), "rb") as fh:
        d = pickle.load(fh)
    assert d["embedding"].dtype == np.float32 and d["embedding"].shape[1] == 768


def test_the_sinusoid_table_may_grow_while_other_forwards_are_in_flight():
    """include/loco_asr.h: a clip longer than any before makes the library grow its sinusoid table (HF does the same on demand,
    modeling:331-333).  With forwards of the handle in flight on other streams the old table must stay alive for them (it is retired,
    not freed) and the new one must be complete before it is published.  Raw C ABI on a private one-layer model: eight forwards of a
    short batch queued on stream A, then -- no synchronisation -- a 4 062-frame clip on stream B whose enqueue grows the table."""
    import ctypes as C
    sd = la.synth.encoder_state_dict(0, layers=1)
    pre, enc_sd = la.synth.split_state_dict(sd)
    mm = la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                          {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=1).cuda()
    enc = mm.speecht5.encoder
    lib = enc._lib
    xs = torch.from_numpy(la.synth.batch([40000, 31000])[0]).cuda()
    enc(input_values=xs)  # loads the weights; the table has its initial ~4 000 rows
    xl = torch.from_numpy(la.synth.batch([1_300_000], first_index=3)[0]).cuda()
    assert int(lib.loco_output_frames(1_300_000)) == 4062
    nst = int(lib.loco_status_bytes())

    def slot(x):
        B, L = x.shape
        return dict(x=x, B=B, L=L, st=torch.cuda.Stream(), ws=torch.empty(int(lib.loco_workspace_bytes(enc._handle, B, L)), dtype=torch.uint8, device="cuda"),
                    status=torch.zeros(nst, dtype=torch.uint8).pin_memory(),
                    outs=[torch.empty((B, int(lib.loco_output_frames(L)), 768), device="cuda") for _ in range(8)])

    a, b = slot(xs), slot(xl)
    torch.cuda.synchronize()

    def enqueue(s, out):
        rc = lib.loco_forward_async(enc._handle, 1, C.c_void_p(s["x"].data_ptr()), None, s["B"], s["L"], C.c_void_p(out.data_ptr()), None, None,
                                    C.c_void_p(s["ws"].data_ptr()), s["ws"].numel(), C.c_void_p(s["st"].cuda_stream), C.c_void_p(s["status"].data_ptr()))
        assert rc == 0, lib.loco_last_error()

    for o in a["outs"]:
        enqueue(a, o)          # in flight on stream A, reading the small table
    enqueue(b, b["outs"][0])   # grows the table inside the call
    torch.cuda.synchronize()
    assert lib.loco_status_check(C.c_void_p(a["status"].data_ptr()), None, 0) == 0
    assert lib.loco_status_check(C.c_void_p(b["status"].data_ptr()), None, 0) == 0
    for o in a["outs"][1:]:
        assert torch.equal(o, a["outs"][0])
    # again with the grown table in place.  The long clip must reproduce itself bit for bit.  The short batch was first encoded with the
    # table the PYTHON module had uploaded (torch's sin / cos, bit-identical to HF's buffer) and now reads the library's own generator:
    # the same positions to the last bit or two of sinf / cosf -- equal to rounding, which is all a caller of the raw ABI is promised
    # across a table growth (the Python module uploads the larger HF-identical table itself before a longer clip, encoder.py).
    enqueue(a, a["outs"][1]); enqueue(b, b["outs"][1])
    torch.cuda.synchronize()
    assert torch.equal(b["outs"][1], b["outs"][0]) and bool(torch.isfinite(b["outs"][0]).all())
    assert rel_l2(a["outs"][1], a["outs"][0]) < 1e-6


def test_load_state_dict_with_forwards_in_flight_finishes_them_on_the_old_weights():
    """A second load_state_dict rebuilds the split planes in place.  Forwards enqueued before it must have finished on the OLD weights
    by then (the module drains its slots before it touches the handle), and the forwards after it use the new ones."""
    layers = 2
    sd_a = la.synth.encoder_state_dict(0, layers=layers)
    sd_b = la.synth.encoder_state_dict(1, layers=layers)

    def fresh(sd):
        pre, enc_sd = la.synth.split_state_dict(sd)
        return la.SpeechT5ForSpeechToTextMI355X.from_state_dicts({k: torch.from_numpy(v) for k, v in pre.items()},
                                                                 {k: torch.from_numpy(v) for k, v in enc_sd.items()}, layers=layers).cuda().speecht5.encoder

    batches = _pairs(6, seconds=4.0, seed_base=300)
    ref_a = [fresh(sd_a)(input_values=x, attention_mask=a).last_hidden_state.clone() for x, a in batches]
    ref_b = [fresh(sd_b)(input_values=x, attention_mask=a).last_hidden_state.clone() for x, a in batches]
    assert not torch.equal(ref_a[0], ref_b[0])
    enc = fresh(sd_a)
    enc.set_inflight(4)
    first = [enc.forward_async(input_values=x, attention_mask=a) for x, a in batches[:4]]   # in flight on weights A
    pre_b, enc_b = la.synth.split_state_dict(sd_b)
    enc.prenet.load_state_dict({k: torch.from_numpy(v) for k, v in pre_b.items()})
    enc.wrapped_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc_b.items()})
    second = [enc.forward_async(input_values=x, attention_mask=a) for x, a in batches]       # weights B
    for t, r in zip(first, ref_a):
        assert torch.equal(t.result().last_hidden_state, r)
    for t, r in zip(second, ref_b):
        assert torch.equal(t.result().last_hidden_state, r)
