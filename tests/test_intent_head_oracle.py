"""Pins the intent-head oracle: its autograd gradients against central finite differences of its own loss, and
its forward against the closed forms of the reference's three pooling branches (intent_classifier.py:24-36)."""
import torch

import intent_head_oracle as iho


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
