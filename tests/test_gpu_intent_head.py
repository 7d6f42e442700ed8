"""Intent head ("next" row f-1) on the GPU: against fixture g8 -- written by the reference's own IntentClassifier driven as
train_classifier.py drives it (tests/golden/make_head_goldens.py) -- and against the torch-autograd oracle on more shapes:
forward logits, loss, every gradient, and three Adam steps, for the three pooling methods, with zero-padded ragged batches
as the reference's collate_fn builds them.  fp32 tolerances: 2e-5 relative on logits/loss/grads, 1e-5 on updated parameters."""
import importlib
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpu_util import la, rel_l2
    import intent_head_oracle as iho


def head_with_g8_params(method):
    q, w, b = la.synth.head_params(method)
    head = la.IntentClassifierMI355X(method)
    head.load_state_dict({"q": torch.from_numpy(q), "classifier.0.weight": torch.from_numpy(w), "classifier.0.bias": torch.from_numpy(b)})
    return head.to("cuda")


@pytest.mark.parametrize("method", ["average", "max", "attention"])
def test_against_the_reference_fixture(method):
    """The HIP head against what /root/reference/speech_text/intent_classifier.py + train_classifier.py:104-116 computed."""
    from test_intent_head_oracle import G8, check_against_g8
    head = head_with_g8_params(method)
    x, target, _ = la.synth.head_batch(5, 129, "fwd")
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(target).cuda()
    assert rel_l2(head(xd), G8[f"{method}/fwd_logits"]) < 2e-5
    loss, _, grads = head.loss_and_grads(xd, td)
    assert abs(float(loss) - float(G8[f"{method}/fwd_loss"])) < 2e-5 * float(G8[f"{method}/fwd_loss"])
    g = grads.cpu().numpy()
    check_against_g8(f"{method}/grad_", g[:768], g[768:768 + 101 * 768].reshape(101, 768), g[768 + 101 * 768:], 5e-5)
    for step in range(3):
        x, target, _ = la.synth.head_batch(16, 180, f"adam{step}")
        lg, _ = head.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(target).cuda())
        assert abs(float(lg) - float(G8[f"{method}/adam_losses"][step])) < 5e-5 * float(G8[f"{method}/adam_losses"][step]), step
    sd = {k: v.cpu().numpy() for k, v in head.state_dict().items()}
    check_against_g8(f"{method}/adam_", sd["q"], sd["classifier.0.weight"], sd["classifier.0.bias"], 1e-5)


def make_batch(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 768, generator=g) * 0.8
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g)
    lens[0] = T
    for b in range(B):
        x[b, lens[b]:] = 0  # pad_sequence zeros
    cls = torch.randint(0, 101, (B,), generator=g)
    target = torch.eye(101, dtype=torch.int64)[cls]
    return x, target


def paired(method, seed=0):
    torch.manual_seed(seed)
    ref = iho.IntentClassifierOracle(method)
    with torch.no_grad():
        ref.q.mul_(300.0)  # make the attention weights non-uniform (q starts at 1e-3 scale)
    head = la.IntentClassifierMI355X(method)
    head.load_state_dict(ref.state_dict())
    return ref, head.to("cuda")


@pytest.mark.parametrize("method", ["average", "max", "attention"])
@pytest.mark.parametrize("B,T", [(16, 250), (3, 1), (5, 129), (2, 700)])
def test_forward_loss_and_gradients(method, B, T):
    ref, head = paired(method)
    x, target = make_batch(B, T, 1)
    logits = head(x.cuda())
    assert tuple(logits.shape) == (B, 1, 101)
    pred = ref(x)
    assert rel_l2(logits, pred.detach()) < 2e-5
    loss = torch.nn.CrossEntropyLoss()(pred.squeeze(1), target.float())
    loss.backward()
    gl, glogits, grads = head.loss_and_grads(x.cuda(), target.cuda())
    assert abs(float(gl) - float(loss.detach())) < 2e-5 * max(1.0, abs(float(loss.detach())))
    gq, gw, gb = grads[:768].cpu(), grads[768:768 + 101 * 768].view(101, 768).cpu(), grads[768 + 101 * 768:].cpu()
    assert rel_l2(gw, ref.classifier[0].weight.grad) < 2e-5
    assert rel_l2(gb, ref.classifier[0].bias.grad) < 2e-5
    if method == "attention":
        rq = ref.q.grad.reshape(-1)
        if float(rq.norm()) == 0.0:  # T == 1: softmax over one frame has no gradient
            assert float(gq.abs().max()) < 1e-7
        else:
            assert rel_l2(gq, rq) < 5e-5
    else:
        assert ref.q.grad is None and float(gq.abs().max()) == 0.0


@pytest.mark.parametrize("method", ["average", "max", "attention"])
def test_three_adam_steps_match_torch(method):
    ref, head = paired(method, seed=3)
    opt = torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=0.0001)
    for step in range(3):
        x, target = make_batch(16, 180, 10 + step)
        lr_, _ = iho.train_step(ref, opt, x, target)
        lg, _ = head.train_step(x.cuda(), target.cuda())
        assert abs(float(lg) - float(lr_)) < 5e-5 * max(1.0, abs(float(lr_))), step
    sd = head.state_dict()
    assert rel_l2(sd["classifier.0.weight"], ref.classifier[0].weight.detach()) < 1e-5
    assert rel_l2(sd["classifier.0.bias"], ref.classifier[0].bias.detach()) < 1e-5
    assert rel_l2(sd["q"], ref.q.detach()) < 1e-5


def test_head_on_real_encoder_output():
    """extract -> classify without the pickle round trip: the encoder's last_hidden_state feeds the head directly."""
    from gpu_util import model
    m, _ = model(layers=2)
    x, msk = la.synth.batch([16000, 9000])
    emb = m.speecht5.encoder(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
    ref, head = paired("attention")
    assert rel_l2(head(emb), ref(emb.cpu()).detach()) < 2e-5


@pytest.mark.parametrize("q_scale", [30.0, 1.0])
def test_configs4_head_training_step_on_real_encoder_output(q_scale):
    """BASELINE.json configs[4] at the head's real operating point: [16, T, 768] embeddings produced by the full 12-layer
    encoder for 16 ragged 30 s clips, encoded in the reference's pairs (…base…py:67-68) so that every embedding carries the
    padded frames of its pair, zero-padded to the longest by the training collate_fn (train_classifier.py:47-51); then logits,
    loss, every gradient and three Adam steps of the HIP head against the torch oracle (itself pinned to the reference's
    IntentClassifier by fixture g8), for the three pooling modes.  q_scale 30: attention weights far from uniform (logits x.q of a
    few units on |x_t| ~ 28); q_scale 1: the reference's own initial scale (q ~ 1e-3 N(0,1), intent_classifier.py:17), near-uniform
    weights -- the regime every training run starts in."""
    from gpu_util import model
    from torch.nn.utils.rnn import pad_sequence
    m, _ = model()
    enc = m.speecht5.encoder
    lens = la.synth.mixed_lengths(16, 480000, seed=5)
    embs = []
    for a in range(0, 16, 2):
        x, msk = la.synth.batch(lens[a:a + 2], first_index=200 + a)
        y = enc(input_values=torch.from_numpy(x).cuda(), attention_mask=torch.from_numpy(msk).cuda()).last_hidden_state
        embs += [y[0].clone(), y[1].clone()]
    assert not enc.last_range_fallback
    data = pad_sequence(embs, batch_first=True)  # [16, Tmax, 768] on the GPU, zero rows beyond each utterance's file length
    assert data.shape[0] == 16 and data.shape[2] == 768 and data.shape[1] == la.synth.conv_out_length(max(lens))
    cls = torch.arange(16) * 6 % 101
    target = torch.eye(101, dtype=torch.int64)[cls]
    xc = data.cpu()
    for method in ("average", "max", "attention"):
        torch.manual_seed(11)
        ref = iho.IntentClassifierOracle(method)
        with torch.no_grad():
            ref.q.mul_(q_scale)
        head = la.IntentClassifierMI355X(method)
        head.load_state_dict(ref.state_dict())
        head = head.to("cuda")
        # truth = the oracle in fp64; the fp32 oracle beside it shows how much fp32 itself can lose on T = 1499 frames (the query
        # gradient of attention pooling is a sum with heavy cancellation), and bounds what is asked of the HIP head
        ref64 = iho.IntentClassifierOracle(method).double()
        ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
        g32, g64 = {}, {}
        for model_, x_, store in ((ref, xc, g32), (ref64, xc.double(), g64)):
            pred = model_(x_)
            loss = torch.nn.CrossEntropyLoss()(pred.squeeze(1), target.to(x_.dtype))
            loss.backward()
            store.update(logits=pred.detach(), loss=float(loss), gw=model_.classifier[0].weight.grad, gb=model_.classifier[0].bias.grad,
                         gq=model_.q.grad.reshape(-1) if model_.q.grad is not None else None)

        # FIXED bars against the fp64 truth (no bar that floats with the comparator).  torch's fp32 autograd is
        # shown beside each figure for orientation only.  The query gradient is the delicate one: as a one-pass covariance
        # (sum alpha dalpha x - (sum alpha dalpha) p) fp32 keeps two digits of it at T = 1499 (the first version of this head: 2.6e-2,
        # torch fp32: ~1e-2); intent_head.hip accumulates the centred form Cov_alpha(x) dpooled instead, which has no cancellation,
        # so 1e-5 is asked of it (measured 6e-7; torch's fp32 autograd: 6e-5 / 9e-6 in the two parametrisations).
        fig = {}
        fig["logits"] = (rel_l2(head(data), g64["logits"]), rel_l2(g32["logits"], g64["logits"]))
        gl, _, grads = head.loss_and_grads(data, target.cuda())
        fig["gw"] = (rel_l2(grads[768:768 + 101 * 768].view(101, 768).cpu(), g64["gw"]), rel_l2(g32["gw"], g64["gw"]))
        fig["gb"] = (rel_l2(grads[768 + 101 * 768:].cpu(), g64["gb"]), rel_l2(g32["gb"], g64["gb"]))
        if method == "attention":
            fig["gq"] = (rel_l2(grads[:768].cpu(), g64["gq"]), rel_l2(g32["gq"], g64["gq"]))
        msg = f"{method}, q x {q_scale:g}, [16, {data.shape[1]}, 768] vs fp64 (HIP head, torch fp32): " + ", ".join(f"{k} ({a:.1e}, {b:.1e})" for k, (a, b) in fig.items())
        print(msg)
        from conftest import record_figure
        record_figure("configs4_head_vs_fp64", method=method, q_scale=q_scale, figures={k: [float(a), float(b)] for k, (a, b) in fig.items()})
        assert fig["logits"][0] < 2e-5, msg
        assert abs(float(gl) - g64["loss"]) < 2e-5 * max(1.0, abs(g64["loss"])), msg
        assert fig["gw"][0] < 5e-5 and fig["gb"][0] < 5e-5, msg
        if method == "attention":
            assert fig["gq"][0] < 1e-5, msg
        else:
            assert float(grads[:768].abs().max()) == 0.0
        # three optimisation steps against the fp32 oracle stepped the same way (train_classifier.py:104-116)
        ref.zero_grad(set_to_none=True)
        opt = torch.optim.Adam(ref.parameters(), lr=0.001, weight_decay=0.0001)
        for step in range(3):
            lr_, _ = iho.train_step(ref, opt, xc, target)
            lg, _ = head.train_step(data, target.cuda())
            assert abs(float(lg) - float(lr_)) < 1e-4 * max(1.0, abs(float(lr_))), (method, step)
        sd = head.state_dict()
        assert rel_l2(sd["classifier.0.weight"], ref.classifier[0].weight.detach()) < 2e-5
        assert rel_l2(sd["q"], ref.q.detach()) < 2e-5


def test_errors():
    _, head = paired("max")
    with pytest.raises(ValueError):
        head(torch.zeros(2, 5, 512, device="cuda"))
    with pytest.raises(RuntimeError):
        head(torch.zeros(2, 5, 768))
