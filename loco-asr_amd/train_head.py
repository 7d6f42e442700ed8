#!/usr/bin/env python3
"""Drop-in for /root/reference/speech_text/train_classifier.py on MI355X: trains the intent head on
pre-extracted embeddings with the HIP head (intent_head.py), same flags (-m modality, -p pooling, -v version),
same folder conventions (extracted/speecht5[_base]/<split>/<modality>/ in; checkpoints/<version>/<modality>/
<pooling>/speecht5_<pooling>_<modality>_{epoch_N,best,last}.pth and results/.../logs/results.txt out), same
hyper-parameters (batch 16, Adam lr 1e-3 wd 1e-4, 100 epochs, patience 5; train_classifier.py:53,61-68) and
the same early-stopping rule (:158-169), the two curves losses.png / accuracies.png (:177-196, when matplotlib is
importable).  Like the reference, validation doubles as the "test" loader (:56 wraps val_set) and the final figures are
divided by len(test_set) (:211-212) -- the `test` split is loaded for that (:37) and, when it does not exist, the
validation size is used and said so.  With WORLD_SIZE > 1 every epoch's shuffled batches (same seeded permutation on
every rank) are dealt round-robin and each rank LOADS ONLY ITS OWN batches (a rank-strided batch sampler: host I/O per
rank is 1/W of the epoch); gradients are all-reduced; the up to W-1 batches of an incomplete last round are dropped so
that the collectives line up.

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


def evaluate(model, loader, device, n_items, collective=False):
    """sum-reduced soft-label CE and accuracy over a loader (train_classifier.py:136-152, 198-215).  Under data parallelism
    (`collective`) `loader` holds only this rank's batches and the two sums are added over the ranks with one all-reduce."""
    sums = torch.zeros(2, dtype=torch.float64, device=device)
    for _, data, target in loader:
        pred = model(data.to(device)).squeeze(1)
        logp = torch.log_softmax(pred, dim=1)
        sums[0] += -(target.to(device).float() * logp).sum().double()
        sums[1] += (pred.argmax(1) == target.to(device).argmax(1)).double().sum()
    if collective:
        import torch.distributed as dist
        dist.all_reduce(sums)
    return float(sums[0]) / n_items, float(sums[1]) / n_items


def strided_batches(n_items, batch_size, world, rank):
    """Validation under data parallelism: the loader's batches in order (shuffle=False), batches rank, rank+W, ... to this rank."""
    return [list(range(a, min(n_items, a + batch_size))) for a in range(0, n_items, batch_size)][rank::world]


def epoch_batches(n_items, batch_size, world, rank, generator):
    """This rank's batches of one epoch: the epoch's seeded permutation (what DataLoader(shuffle=True, generator=g) draws,
    train_classifier.py:54) cut into batches of `batch_size`; batches rank, rank+W, ... of the first floor(nb/W)*W (all of
    them at W = 1).  Returns (global batch indices, index lists).  Every rank must call it once per epoch with an identically
    seeded generator."""
    perm = torch.randperm(n_items, generator=generator).tolist()
    batches = [perm[a:a + batch_size] for a in range(0, len(perm), batch_size)]
    stop = len(batches) if world == 1 else (len(batches) // world) * world
    ids = list(range(rank, stop, world))
    return ids, [batches[i] for i in ids]


def write_curves(curves, plots_folder):
    """losses.png / accuracies.png of train_classifier.py:177-196; skipped (with a note) where matplotlib is missing."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:
        print("matplotlib is not installed: curves not plotted")
        return
    os.makedirs(plots_folder, exist_ok=True)
    for name, a, b, ylabel, title in (("losses.png", "loss", "val_loss", "Loss", "Training and Validation Loss"),
                                      ("accuracies.png", "acc", "val_acc", "Accuracy", "Training and Validation Accuracy")):
        plt.figure()
        plt.plot(curves[a], label="Training " + ylabel)
        plt.plot(curves[b], label="Validation " + ylabel)
        plt.xlabel("Epoch")
        plt.ylabel(ylabel)
        plt.title(title)
        plt.legend()
        plt.savefig(os.path.join(plots_folder, name))
        plt.close()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train an Intent Classifier with SpeechT5 embeddings from SLURP dataset (MI355X)")
    ap.add_argument("--modality", "-m", choices=["text", "audio"], required=True)
    ap.add_argument("--pooling", "-p", choices=["average", "max", "attention"], required=True)
    ap.add_argument("--version", "-v", choices=["fine_tuned", "base"], required=True)
    ap.add_argument("--folder", default=None, help="override extracted/<...> root")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--out-root", default=".")
    ap.add_argument("--seed", type=int, default=None,
                    help="seed of the head's initial parameters (the reference does not seed: train_classifier.py draws them from "
                         "torch's default generator); under data parallelism rank 0's parameters are broadcast either way")
    args = ap.parse_args(argv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device: the MI355X head has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    print("Running on", device)
    # LOCO_FORCE_COLLECTIVE=1 (dp.py): create the process group and issue every collective even at world size 1, so that a
    # one-GPU box exercises the RCCL gradient all-reduce of BASELINE.json configs[4] (tests/test_gpu_rccl_world1.py)
    dp = importlib.import_module("loco-asr_amd.dp")
    collective = world > 1 or (os.environ.get("LOCO_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ)
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
        dp.FORCE_COLLECTIVE = world == 1

    folder = args.folder or ("extracted/speecht5" if args.version == "fine_tuned" else "extracted/speecht5_base")
    sets = [sink.EmbeddingsTargets(folder, args.modality, "train")]
    if os.path.isdir(os.path.join(folder, "train_synthetic", args.modality)):
        sets.append(sink.EmbeddingsTargets(folder, args.modality, "train_synthetic"))
    train_set = ConcatDataset(sets)
    val_set = sink.EmbeddingsTargets(folder, args.modality, "devel")
    print(f"Train set: {len(train_set)}, Val set: {len(val_set)}")
    n_test, test_note = len(val_set), " (no test split on disk: divided by the validation size)"
    if os.path.isdir(os.path.join(folder, "test", args.modality)):
        n_test, test_note = len(sink.EmbeddingsTargets(folder, args.modality, "test")), ""
        print(f"Test set: {n_test}")
    g = torch.Generator().manual_seed(0)  # identical shuffles on every rank
    batch_size = 16
    n_train_batches = (len(train_set) + batch_size - 1) // batch_size

    def my_epoch_loader():
        ids, mine = epoch_batches(len(train_set), batch_size, world, rank, g)
        return ids, DataLoader(train_set, batch_sampler=mine, collate_fn=collate_fn)  # only this rank's batches are read from disk

    # every rank evaluates ITS share of the validation batches; the two sums meet in one all-reduce (evaluate)
    val_loader = DataLoader(val_set, batch_sampler=strided_batches(len(val_set), batch_size, world, rank), collate_fn=collate_fn)

    if args.seed is not None:
        torch.manual_seed(args.seed)
    model = la.IntentClassifierMI355X(method=args.pooling, lr=0.001, weight_decay=0.0001).to(device)
    if collective:
        model.broadcast_parameters()  # every rank starts from rank 0's draw
    save_folder = os.path.join(args.out_root, "checkpoints", args.version, args.modality, args.pooling)
    logs_folder = os.path.join(args.out_root, "results", args.version, args.modality, args.pooling, "logs")
    if rank == 0:
        os.makedirs(save_folder, exist_ok=True)
        os.makedirs(logs_folder, exist_ok=True)
    tag = f"speecht5_{args.pooling}_{args.modality}"
    text = "Results\n"
    best, stale = float("inf"), 0
    curves = {"loss": [], "val_loss": [], "acc": [], "val_acc": []}
    print("Training started...")
    for epoch in range(args.epochs):
        epoch_loss, acc_train, n_batches, n_seen = 0.0, 0.0, 0, 0
        batch_ids, train_loader = my_epoch_loader()
        for i, (_, data, target) in zip(batch_ids, train_loader):
            loss, pred = model.train_step(data.to(device), target.to(device))
            epoch_loss += float(loss)
            acc_train += float((pred.argmax(1) == target.to(device).argmax(1)).float().sum())
            n_batches += 1
            n_seen += data.shape[0]
            if (i + 1) % 200 == 0 and rank == 0:
                print(f"Epoch [{epoch+1}/{args.epochs}], Iteration [{i+1}/{n_train_batches}], Loss: {float(loss):.4f}")
                text += f"Epoch [{epoch+1}/{args.epochs}], Iteration [{i+1}/{n_train_batches}], Loss: {float(loss):.4f}\n"
        epoch_loss /= max(1, n_batches)
        acc_train /= max(1, n_seen)
        val_loss, acc_val = evaluate(model, val_loader, device, len(val_set), collective)
        for k, v in (("loss", epoch_loss), ("val_loss", val_loss), ("acc", acc_train), ("val_acc", acc_val)):
            curves[k].append(v)
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
        write_curves(curves, os.path.join(os.path.dirname(logs_folder), "plots"))
    print("Training done!")
    if collective:
        import torch.distributed as dist
        print(f"Gradient all-reduces issued: {model.allreduces_issued} (backend {dist.get_backend()}, world size {world})")
        dist.barrier()  # rank 0 has finished writing *_best.pth before any rank reads it
    model = la.IntentClassifierMI355X(method=args.pooling).to(device)
    model.load_state_dict(torch.load(os.path.join(save_folder, f"{tag}_best.pth")))
    print("Evaluating model on test set" + test_note)
    # the reference iterates the validation loader here and divides by len(test_set) (train_classifier.py:56, 211-212)
    tl, ta = evaluate(model, val_loader, device, n_test, collective)
    print(f"Test Loss: {tl:.4f}")
    print(f"Test Accuracy: {ta*100:.2f}")
    print("Evaluation done!")
    if collective:
        dist.destroy_process_group()
    return tl, ta


if __name__ == "__main__":
    main()
