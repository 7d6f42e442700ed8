#!/usr/bin/env python3
"""Drop-in for /root/reference/speech_text/train_classifier.py on MI355X: trains the intent head on
pre-extracted embeddings with the HIP head (intent_head.py), same flags (-m modality, -p pooling, -v version),
same folder conventions (extracted/speecht5[_base]/<split>/<modality>/ in; checkpoints/<version>/<modality>/
<pooling>/speecht5_<pooling>_<modality>_{epoch_N,best,last}.pth and results/.../logs/results.txt out), same
hyper-parameters (batch 16, Adam lr 1e-3 wd 1e-4, 100 epochs, patience 5; train_classifier.py:53,61-68) and
the same early-stopping rule (:158-169).  Like the reference, validation doubles as the "test" loader
(:56 wraps val_set).  With WORLD_SIZE > 1 each rank takes every W-th batch and gradients are all-reduced.

    python loco-asr_amd/train_head.py -m audio -p attention -v base
    torchrun --standalone --nproc-per-node 8 loco-asr_amd/train_head.py -m audio -p attention -v base
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import ConcatDataset, DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
la = importlib.import_module("loco-asr_amd")
sink = importlib.import_module("loco-asr_amd.sink")


def collate_fn(batch):
    slurp_ids, embeddings, targets = zip(*batch)
    return slurp_ids, pad_sequence(embeddings, batch_first=True), torch.stack(targets, dim=0)


def evaluate(model, loader, device, n_items):
    """sum-reduced soft-label CE and accuracy over a loader (train_classifier.py:136-152, 198-215)."""
    loss, acc = 0.0, 0.0
    for _, data, target in loader:
        pred = model(data.to(device)).squeeze(1)
        logp = torch.log_softmax(pred, dim=1)
        loss += float(-(target.to(device).float() * logp).sum())
        acc += float((pred.argmax(1) == target.to(device).argmax(1)).float().sum())
    return loss / n_items, acc / n_items


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train an Intent Classifier with SpeechT5 embeddings from SLURP dataset (MI355X)")
    ap.add_argument("--modality", "-m", choices=["text", "audio"], required=True)
    ap.add_argument("--pooling", "-p", choices=["average", "max", "attention"], required=True)
    ap.add_argument("--version", "-v", choices=["fine_tuned", "base"], required=True)
    ap.add_argument("--folder", default=None, help="override extracted/<...> root")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--out-root", default=".")
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device: the MI355X head has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    print("Running on", device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    folder = args.folder or ("extracted/speecht5" if args.version == "fine_tuned" else "extracted/speecht5_base")
    sets = [sink.EmbeddingsTargets(folder, args.modality, "train")]
    if os.path.isdir(os.path.join(folder, "train_synthetic", args.modality)):
        sets.append(sink.EmbeddingsTargets(folder, args.modality, "train_synthetic"))
    train_set = ConcatDataset(sets)
    val_set = sink.EmbeddingsTargets(folder, args.modality, "devel")
    print(f"Train set: {len(train_set)}, Val set: {len(val_set)}")
    g = torch.Generator().manual_seed(0)  # identical shuffles on every rank
    train_loader = DataLoader(train_set, batch_size=16, shuffle=True, collate_fn=collate_fn, generator=g)
    val_loader = DataLoader(val_set, batch_size=16, shuffle=False, collate_fn=collate_fn)

    model = la.IntentClassifierMI355X(method=args.pooling, lr=0.001, weight_decay=0.0001).to(device)
    save_folder = os.path.join(args.out_root, "checkpoints", args.version, args.modality, args.pooling)
    logs_folder = os.path.join(args.out_root, "results", args.version, args.modality, args.pooling, "logs")
    if rank == 0:
        os.makedirs(save_folder, exist_ok=True)
        os.makedirs(logs_folder, exist_ok=True)
    tag = f"speecht5_{args.pooling}_{args.modality}"
    text = "Results\n"
    best, stale = float("inf"), 0
    print("Training started...")
    for epoch in range(args.epochs):
        epoch_loss, acc_train, n_batches = 0.0, 0.0, 0
        for i, (_, data, target) in enumerate(train_loader):
            if i % world != rank:  # data parallel: every W-th batch is this rank's; ranks with no batch left stop together
                continue
            if (i // world) * world + world > len(train_loader) and world > 1:
                break
            loss, pred = model.train_step(data.to(device), target.to(device))
            epoch_loss += float(loss)
            acc_train += float((pred.argmax(1) == target.to(device).argmax(1)).float().sum())
            n_batches += 1
            if (i + 1) % 200 == 0 and rank == 0:
                print(f"Epoch [{epoch+1}/{args.epochs}], Iteration [{i+1}/{len(train_loader)}], Loss: {float(loss):.4f}")
        epoch_loss /= max(1, n_batches)
        acc_train /= max(1, n_batches * 16)
        val_loss, acc_val = evaluate(model, val_loader, device, len(val_set))
        line = (f"Epoch [{epoch+1}/{args.epochs}], Training Loss: {epoch_loss:.4f}, Training accuracy: {round(acc_train*100, 2)}, "
                f"Validation Loss: {val_loss:.4f}, Validation accuracy: {acc_val*100:.2f}")
        print(line)
        text += f"###### {line} ######\n\n"
        if rank == 0:
            torch.save(model.state_dict(), os.path.join(save_folder, f"{tag}_epoch_{epoch+1}.pth"))
        if val_loss < best:
            best, stale = val_loss, 0
            if rank == 0:
                torch.save(model.state_dict(), os.path.join(save_folder, f"{tag}_best.pth"))
        else:
            stale += 1
        if stale >= 5:
            print("Early stopping: Validation loss has not improved in the last 5 epochs.")
            break
    if rank == 0:
        torch.save(model.state_dict(), os.path.join(save_folder, f"{tag}_last.pth"))
        with open(os.path.join(logs_folder, "results.txt"), "w") as fh:
            fh.write(text)
    print("Training done!")
    model = la.IntentClassifierMI355X(method=args.pooling).to(device)
    model.load_state_dict(torch.load(os.path.join(save_folder, f"{tag}_best.pth")))
    print("Evaluating model on test set")
    tl, ta = evaluate(model, val_loader, device, len(val_set))
    print(f"Test Loss: {tl:.4f}")
    print(f"Test Accuracy: {ta*100:.2f}")
    return tl, ta


if __name__ == "__main__":
    main()
