#!/usr/bin/env python3
"""Golden vectors for the intent head ("next" row f-1), produced by THE REFERENCE'S OWN CODE.

Runs in the build container only (it imports /root/reference/speech_text/intent_classifier.py, which needs nothing but
torch); the reference's .py never travels -- only the .npz written here does.  What is replayed, line for line:

    model = IntentClassifier(method=pooling, embedding_size=768)                       train_classifier.py:59
    criterion = nn.CrossEntropyLoss()                                                   train_classifier.py:67
    optimizer = optim.Adam(model.parameters(), lr=0.001, weight_decay=0.0001)           train_classifier.py:69
    optimizer.zero_grad(); pred = model(data); pred = pred.squeeze(1)                   train_classifier.py:104-110
    loss = criterion(pred, target.float()); loss.backward(); optimizer.step()           train_classifier.py:112-116

for the three pooling modes (`--pooling attention` reaches IntentClassifier's else-branch = self_attention,
intent_classifier.py:44-45).  Inputs and initial parameters are pure functions of (key, seed) through
loco-asr_amd/synth.py's hash generator, so the tests rebuild them bit for bit and the fixture holds outputs only:

    fwd_*   one forward + backward on a ragged zero-padded batch [5, 129, 768] (pad_sequence semantics, :47-51)
    adam_*  three optimisation steps on batches [16, 180, 768]

To keep the file small the 101 x 768 weight gradient / updated weight is stored as 8 full rows + the L2 norm of every row.

    python tests/golden/make_head_goldens.py        # writes tests/golden/g8_intent_head.npz
"""
import importlib
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
synth = importlib.import_module("loco-asr_amd.synth")

REF = "/root/reference/speech_text/intent_classifier.py"
ROWS = (0, 3, 17, 50, 64, 77, 99, 100)  # rows of the [101, 768] classifier weight kept in full
METHODS = ("average", "max", "attention")  # train_classifier.py's --pooling choices, passed on as IntentClassifier(method=...)


head_params, head_batch = synth.head_params, synth.head_batch


def pack(prefix, q, w, b, out):
    out[prefix + "q"] = np.asarray(q, np.float32).reshape(-1)
    out[prefix + "b"] = np.asarray(b, np.float32)
    out[prefix + "w_rows"] = np.asarray(w, np.float32)[list(ROWS)]
    out[prefix + "w_row_norms"] = np.linalg.norm(np.asarray(w, np.float64), axis=1).astype(np.float32)


def main():
    spec = importlib.util.spec_from_file_location("ref_intent_classifier", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {"rows": np.asarray(ROWS)}
    for name in METHODS:
        model = ref.IntentClassifier(method=name, embedding_size=768)
        q, w, b = head_params(name)
        model.load_state_dict({"q": torch.from_numpy(q), "classifier.0.weight": torch.from_numpy(w),
                               "classifier.0.bias": torch.from_numpy(b)})
        criterion = torch.nn.CrossEntropyLoss()
        # ---- one forward / backward
        x, target, _ = head_batch(5, 129, "fwd")
        data, tgt = torch.from_numpy(x), torch.from_numpy(target)
        model.train()
        pred = model(data)
        out[f"{name}/fwd_logits"] = pred.detach().numpy().copy()  # [5,1,101]
        loss = criterion(pred.squeeze(1), tgt.float())
        loss.backward()
        out[f"{name}/fwd_loss"] = np.float32(loss.item())
        gq = model.q.grad if model.q.grad is not None else torch.zeros_like(model.q)  # q takes no part in average / max
        pack(f"{name}/grad_", gq.numpy(), model.classifier[0].weight.grad.numpy(), model.classifier[0].bias.grad.numpy(), out)
        # ---- three steps of the training loop
        model.zero_grad(set_to_none=True)
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001, weight_decay=0.0001)
        losses = []
        for step in range(3):
            x, target, _ = head_batch(16, 180, f"adam{step}")
            data, tgt = torch.from_numpy(x), torch.from_numpy(target)
            optimizer.zero_grad()
            pred = model(data)
            pred = pred.squeeze(1)
            loss = criterion(pred, tgt.float())
            loss.backward()
            optimizer.step()
            losses.append(loss.item())
        out[f"{name}/adam_losses"] = np.asarray(losses, np.float32)
        sd = model.state_dict()
        pack(f"{name}/adam_", sd["q"].numpy(), sd["classifier.0.weight"].numpy(), sd["classifier.0.bias"].numpy(), out)
    path = os.path.join(HERE, "g8_intent_head.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; torch", torch.__version__)


if __name__ == "__main__":
    main()
