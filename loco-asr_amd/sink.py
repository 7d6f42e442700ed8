"""On-disk sink of the embedding path, in the reference's format, with I/O overlapped with compute.

Format (the contract of the downstream reader, /root/reference/speech_text/slurp_embeddings_and_targets.py:19-28,
written by /root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:111-113): one file per
utterance, ``{root}/{split}/{modality}/{slurp_id}_embedding_and_target.pickle``, a pickle
(``HIGHEST_PROTOCOL``) of ``{"id": slurp_id, "embedding": float32 ndarray [T_batchmax, 768], "target":
one-hot int ndarray [101]}``.  ``embedding`` keeps the batch's padded frames, as the reference does (it zips
the full ``[B, T_max, 768]`` array).  ``fmt="npy"`` writes ``{id}.embedding.npy`` / ``{id}.target.npy``
instead (BASELINE.json mentions .npy; the reference itself never writes it -- SURVEY.md §8f-2).

Overlap: ``submit`` enqueues an asynchronous D2H copy into pinned memory on a side stream (ordered after the
producing stream by an event) and returns; writer threads wait for the copy event and serialise.  The
reference blocks on ``.cpu()`` and on every ``pickle.dump`` inside its batch loop (…base…py:109-113).
"""
from __future__ import annotations

import os
import pickle
import queue
import threading
from typing import Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset


def embedding_path(folder: str, slurp_id, fmt: str = "pickle") -> str:
    return os.path.join(folder, f"{slurp_id}_embedding_and_target.pickle" if fmt == "pickle" else f"{slurp_id}.embedding.npy")


def write_one(folder: str, slurp_id, embedding: np.ndarray, target: np.ndarray, fmt: str = "pickle") -> str:
    path = embedding_path(folder, slurp_id, fmt)
    if fmt == "pickle":
        with open(path, "wb") as handle:
            pickle.dump({"id": slurp_id, "embedding": embedding, "target": target}, handle, protocol=pickle.HIGHEST_PROTOCOL)
    elif fmt == "npy":
        np.save(path, embedding)
        np.save(os.path.join(folder, f"{slurp_id}.target.npy"), target)
    else:
        raise ValueError(f"unknown format {fmt!r}")
    return path


class EmbeddingSink:
    def __init__(self, root: str, split: str, modality: str = "audio", fmt: str = "pickle", workers: int = 4,
                 max_pending: int = 4):
        self.folder = os.path.join(os.path.join(root, split), modality)
        os.makedirs(self.folder, exist_ok=True)
        self.fmt = fmt
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)
        self._err: Optional[BaseException] = None
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, workers))]
        for t in self._threads:
            t.start()
        self._copy_stream = None
        self.written = 0
        self._lock = threading.Lock()

    def _work(self):
        while True:
            item = self._q.get()
            if item is None:
                self._q.task_done()
                return
            try:
                ids, host, targets, event, rows = item
                if event is not None:
                    event.synchronize()
                arr = host.numpy()
                for i, sid in enumerate(ids):
                    # np.array(...) detaches each utterance from the pinned staging buffer before pickling
                    emb = arr[i] if rows is None else arr[i, :rows[i]]
                    write_one(self.folder, sid, np.array(emb), np.asarray(targets[i]), self.fmt)
                with self._lock:
                    self.written += len(ids)
            except BaseException as e:  # surfaced on the next submit()/close()
                self._err = e
            finally:
                self._q.task_done()

    def submit(self, ids: Sequence, embeddings: torch.Tensor, targets, rows: Optional[Sequence[int]] = None, chunk: int = 0):
        """embeddings [B, T, 768] on any device; returns as soon as the D2H copy is enqueued.  ``rows`` (packed forward: the
        output of several reference batches in one tensor) keeps the first rows[i] frames of utterance i -- the frames of ITS OWN
        batch, that batch's padded frames included -- instead of all T; ``chunk`` > 0 hands the utterances to the writer threads
        in groups of that many (one D2H copy, several writers)."""
        if self._err:
            raise self._err
        if len(ids) != embeddings.shape[0] or len(targets) != len(ids) or (rows is not None and len(rows) != len(ids)):
            raise ValueError("ids / embeddings / targets / rows disagree on the batch size")
        event = None
        if embeddings.is_cuda:
            dev = embeddings.device
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(dev)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            host = torch.empty(embeddings.shape, dtype=embeddings.dtype, pin_memory=True)
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(ready)
                host.copy_(embeddings, non_blocking=True)
                embeddings.record_stream(self._copy_stream)
                event = torch.cuda.Event()
                event.record(self._copy_stream)
        else:
            host = embeddings.detach().clone()
        ids, targets = list(ids), [np.asarray(t) for t in targets]
        rows = [int(r) for r in rows] if rows is not None else None
        step = chunk if chunk > 0 else max(1, len(ids))
        for a in range(0, len(ids), step):
            self._q.put((ids[a:a + step], host[a:a + step], targets[a:a + step], event, rows[a:a + step] if rows is not None else None))

    def close(self):
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join()
        if self._err:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class EmbeddingsTargets(Dataset):
    """Reader with the behaviour of the reference's SLURPEmbeddingsTargets
    (slurp_embeddings_and_targets.py:6-28): lists ``{root}/{split}/{modality}/`` and returns
    ``(id, torch embedding, torch target)`` per file."""

    def __init__(self, data_path: str, modality: str = "text", split: str = "train"):
        self.full_path = os.path.join(os.path.join(data_path, split), modality)
        self.dataset = sorted(f for f in os.listdir(self.full_path) if f.endswith(".pickle"))

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        with open(os.path.join(self.full_path, self.dataset[idx]), "rb") as fh:
            d = pickle.load(fh)
        return d["id"], torch.from_numpy(d["embedding"]), torch.from_numpy(d["target"])
