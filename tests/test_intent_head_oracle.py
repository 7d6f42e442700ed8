"""Pins the intent-head oracle to THE REFERENCE'S OWN CODE: fixture g8 was written by tests/golden/make_head_goldens.py from
/root/reference/speech_text/intent_classifier.py driven as train_classifier.py:59-116 drives it (forward logits, loss, every
gradient, parameters after three Adam steps, for the three pooling modes).  The older self-checks stay as extra: autograd
against central finite differences, and the forward against the closed forms of the three pooling branches."""
import importlib
import os

import numpy as np
import pytest
import torch

import intent_head_oracle as iho

synth = importlib.import_module("loco-asr_amd.synth")
G8 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_intent_head.npz"))


def check_against_g8(prefix, q, w, b, tol):
    """(q, W, b)-shaped results against the fixture: q and b in full, 8 rows of W in full, every row of W by its norm."""
    rows = list(G8["rows"])
    q, w, b = (np.asarray(t, np.float64) for t in (q, w, b))
    for got, key in ((q.reshape(-1), "q"), (b, "b"), (w[rows], "w_rows"), (np.linalg.norm(w, axis=1), "w_row_norms")):
        want = G8[prefix + key].astype(np.float64)
        scale = max(float(np.linalg.norm(want)), 1e-30)
        if float(np.linalg.norm(want)) == 0.0:
            assert float(np.abs(got).max()) == 0.0, prefix + key
        else:
            assert float(np.linalg.norm(got - want)) / scale < tol, (prefix + key, float(np.linalg.norm(got - want)) / scale)


def oracle_with_g8_params(method):
    m = iho.IntentClassifierOracle(method)
    q, w, b = synth.head_params(method)
    m.load_state_dict({"q": torch.from_numpy(q), "classifier.0.weight": torch.from_numpy(w), "classifier.0.bias": torch.from_numpy(b)})
    return m


@pytest.mark.parametrize("method", ["average", "max", "attention"])
def test_oracle_reproduces_the_reference_fixture(method):
    m = oracle_with_g8_params(method)
    x, target, _ = synth.head_batch(5, 129, "fwd")
    pred = m(torch.from_numpy(x))
    want = G8[f"{method}/fwd_logits"]
    assert pred.shape == want.shape == (5, 1, 101)
    assert float(np.linalg.norm(pred.detach().numpy() - want) / np.linalg.norm(want)) < 2e-6
    loss = torch.nn.CrossEntropyLoss()(pred.squeeze(1), torch.from_numpy(target).float())
    loss.backward()
    assert abs(float(loss) - float(G8[f"{method}/fwd_loss"])) < 2e-6 * float(G8[f"{method}/fwd_loss"])
    gq = m.q.grad if m.q.grad is not None else torch.zeros_like(m.q)
    check_against_g8(f"{method}/grad_", gq.numpy(), m.classifier[0].weight.grad.numpy(), m.classifier[0].bias.grad.numpy(), 5e-6)
    # three steps of train_classifier.py:104-116
    m.zero_grad(set_to_none=True)
    opt = torch.optim.Adam(m.parameters(), lr=0.001, weight_decay=0.0001)
    for step in range(3):
        x, target, _ = synth.head_batch(16, 180, f"adam{step}")
        loss, _ = iho.train_step(m, opt, torch.from_numpy(x), torch.from_numpy(target))
        assert abs(float(loss) - float(G8[f"{method}/adam_losses"][step])) < 5e-6 * float(G8[f"{method}/adam_losses"][step])
    sd = m.state_dict()
    check_against_g8(f"{method}/adam_", sd["q"].numpy(), sd["classifier.0.weight"].numpy(), sd["classifier.0.bias"].numpy(), 2e-6)


def test_pooling_branches_closed_form():
    torch.manual_seed(0)
    x = torch.randn(3, 7, 768)
    for method in ("average", "max", "attention"):
        m = iho.IntentClassifierOracle(method).double()
        with torch.no_grad():
            m.q.mul_(200)
        xd = x.double()
        if method == "average":
            pooled = xd.mean(1)
        elif method == "max":
            pooled = xd.max(1).values
        else:
            a = torch.softmax(xd @ m.q[0], dim=1)
            pooled = (a[:, :, None] * xd).sum(1)
        want = pooled @ m.classifier[0].weight.T + m.classifier[0].bias
        got = m(xd)
        assert got.shape == (3, 1, 101)
        assert torch.allclose(got[:, 0], want, atol=1e-12)


def test_attention_gradient_matches_finite_differences():
    torch.manual_seed(1)
    m = iho.IntentClassifierOracle("attention").double()
    with torch.no_grad():
        m.q.mul_(200)
    x = torch.randn(2, 5, 768, dtype=torch.float64)
    t = torch.eye(101, dtype=torch.float64)[[3, 77]]
    loss = torch.nn.CrossEntropyLoss()(m(x).squeeze(1), t)
    loss.backward()
    for idx in (0, 100, 767):
        eps = 1e-6
        with torch.no_grad():
            m.q[0, idx] += eps
            lp = torch.nn.CrossEntropyLoss()(m(x).squeeze(1), t)
            m.q[0, idx] -= 2 * eps
            lm = torch.nn.CrossEntropyLoss()(m(x).squeeze(1), t)
            m.q[0, idx] += eps
        fd = float((lp - lm) / (2 * eps))
        assert abs(fd - float(m.q.grad[0, idx])) < 1e-7 + 1e-5 * abs(fd)
