"""CPU oracle of the intent head ("next" row f-1).  TEST INFRASTRUCTURE ONLY (see speecht5_oracle.py header).

Restates /root/reference/speech_text/intent_classifier.py:4-49 (IntentClassifier) as a plain torch module with
autograd, and one optimisation step of train_classifier.py:104-115 with torch.optim.Adam(lr 1e-3, wd 1e-4).
Pinning: the reference has no fixtures for the head, but its code is importable in the build container (torch only), so
tests/golden/make_head_goldens.py ran THAT code (IntentClassifier driven as train_classifier.py:59-116 drives it) and
committed its outputs as tests/golden/g8_intent_head.npz; tests/test_intent_head_oracle.py holds this restatement to them
(logits, loss, every gradient, parameters after three Adam steps; <= 5e-6), and keeps the finite-difference checks."""
import torch
from torch import nn


class IntentClassifierOracle(nn.Module):
    def __init__(self, method="average", embedding_size=768):
        super().__init__()
        self.method = method
        self.q = nn.Parameter(torch.randn(1, embedding_size) * 0.001)
        self.classifier = nn.Sequential(nn.Linear(embedding_size, 101))

    def forward(self, x):
        if self.method == "average":
            x = torch.mean(x, dim=1, keepdim=True)
        elif self.method == "max":
            x = torch.max(x, dim=1, keepdim=True).values
        else:  # learned-query attention: alpha = softmax_t(x q^T); pooled = alpha^T x
            z = torch.matmul(x, self.q.T)
            alpha = torch.softmax(z, dim=1)
            x = torch.matmul(alpha.permute(0, 2, 1), x)
        return self.classifier(x)


def train_step(model, optimizer, x, target):
    """train_classifier.py:104-115: zero_grad, forward, squeeze(1), CrossEntropyLoss(float targets), backward, step."""
    optimizer.zero_grad()
    pred = model(x).squeeze(1)
    loss = nn.CrossEntropyLoss()(pred, target.float())
    loss.backward()
    optimizer.step()
    return loss.detach(), pred.detach()
