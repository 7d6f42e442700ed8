"""Data-parallel sharding of utterance batches: one process per GPU, one RCCL all-gather per step.

The reference is single-process, single-device (/root/reference/speech_text/
extract_speecht5_base_embeddings_slurp.py:23); data parallelism is new here and follows SURVEY.md §8(e):
every utterance (or 10-minute window of a podcast) is an independent unit -- GroupNorm is per
(clip, channel), LayerNorm per frame, attention per clip -- so ranks never exchange activations.  The
only collective is the gather of the finished embeddings:

    lengths  : all_gather of int32 [B_loc, 2]  (frames, samples)        -- tiny
    payload  : all_gather_into_tensor of f32 [B_loc, T_max, 768]         -- 147 MB/rank at 30 s x 32

On ROCm the "nccl" backend is RCCL; over the xGMI full mesh an all-gather of this size is a few ms
against >100 ms of compute per step.

Two ways of dealing units to ranks:

* ``shard_batches`` (the default of the CLI and of ``encode_sharded``): the corpus is cut into consecutive batches in
  corpus order -- exactly the batches ``DataLoader(dataset, batch_size=b, shuffle=False)`` forms in the reference
  (…base…py:67-68) -- and WHOLE batches are dealt round-robin (rank r encodes batches r, r+W, r+2W, ...).  Batch
  composition is part of the function (GroupNorm statistics run over the padded axis), so this is what keeps the
  per-utterance output identical to the reference's at any world size.
* ``shard_units`` (opt-in, ``--bucket-by-length``): single units sorted by length, longest first, dealt round-robin
  (rank r gets sorted positions r, r+W, ...): attention cost grows with T^2, so equal counts of similar lengths
  balance best and padding inside a batch is minimal -- but the batches differ from the reference's.

Everything here works with any torch.distributed backend: tests run it on CPU with gloo, world size 2.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_units(lengths: Sequence[int], world_size: int, rank: int) -> List[int]:
    """Indices of the units this rank encodes, bucketed by length: longest-first order dealt round-robin,
    rank r gets sorted positions r, r+W, r+2W, ...  -> equal counts (+-1) and matched lengths."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return order[rank::world_size]


def corpus_batches(n_units: int, batch_size: int) -> List[List[int]]:
    """The batches DataLoader(batch_size=b, shuffle=False) forms: consecutive runs of the corpus order, last one short."""
    return [list(range(a, min(n_units, a + batch_size))) for a in range(0, n_units, batch_size)]


def shard_batches(n_units: int, batch_size: int, world_size: int, rank: int) -> List[List[int]]:
    """This rank's share of ``corpus_batches``: whole batches dealt round-robin, so every batch any rank encodes is one the
    single-process reference loop encodes."""
    return corpus_batches(n_units, batch_size)[rank::world_size]


def rounds(n_units: int, batch_size: int, world_size: int) -> int:
    """Steps every rank takes (ranks that have run out contribute empty batches so that collectives line up)."""
    nb = (n_units + batch_size - 1) // batch_size
    return (nb + world_size - 1) // world_size


def window_units(lengths: Sequence[int], window: int, min_samples: int = 400) -> List[Tuple[int, int, int]]:
    """BASELINE.json configs[3]: long recordings (60-minute podcasts) are cut into fixed windows (10 minutes) that are
    encoded as independent units.  Returns (recording index, first sample, end sample) per window in recording order;
    a trailing piece shorter than one encoder frame (400 samples) is dropped."""
    units = []
    for i, n in enumerate(lengths):
        n = int(n)
        for a in range(0, n, window):
            b = min(n, a + window)
            if b - a >= min_samples:
                units.append((i, a, b))
    return units


# Set (or LOCO_FORCE_COLLECTIVE=1) to issue the collectives even in a process group of ONE rank: a one-GPU box can then run
# the real RCCL all_gather_into_tensor on device tensors next to the encoder's streams (tests/test_gpu_rccl_world1.py,
# bench.py --force-collective).  Without a process group there is nothing to issue and the flag is ignored.
FORCE_COLLECTIVE = False


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _skip_collective(world: int) -> bool:
    import os
    forced = FORCE_COLLECTIVE or os.environ.get("LOCO_FORCE_COLLECTIVE") == "1"
    return world == 1 and not (forced and dist.is_available() and dist.is_initialized())


def all_gather_embeddings(local: torch.Tensor, group=None) -> torch.Tensor:
    """[B_loc, T, D] on every rank (same shape everywhere) -> [W*B_loc, T, D]; ONE collective."""
    world, _ = _world(group)
    if _skip_collective(world):
        return local
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


class OverlappedGather:
    """The per-step all-gather issued asynchronously so that it rides under the NEXT step's compute (the gathered
    batch is only needed by the consumer -- sink or classifier -- one step later).  ``submit`` waits for the previous
    gather, then starts this one; ``finish`` waits for the last.  Output buffers alternate between two slots."""

    def __init__(self, group=None):
        self.group = group
        self._work = None
        self._bufs = [None, None]
        self._i = 0
        self.result = None

    def submit(self, local: torch.Tensor):
        world, _ = _world(self.group)
        if _skip_collective(world):
            self.result = local
            return
        self.finish()
        shape = (world * local.shape[0],) + tuple(local.shape[1:])
        buf = self._bufs[self._i]
        if buf is None or tuple(buf.shape) != shape or buf.device != local.device:
            buf = torch.empty(shape, dtype=local.dtype, device=local.device)
            self._bufs[self._i] = buf
        self._src = local.contiguous()  # keep the source alive until the collective has consumed it
        self._work = dist.all_gather_into_tensor(buf, self._src, group=self.group, async_op=True)
        self._pending = buf
        self._i ^= 1

    def finish(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            self.result = self._pending
        return self.result


def gather_ragged(local: torch.Tensor, local_ids: Sequence[int], n_total: int, group=None):
    """Gather per-rank results whose batch size and padded length differ between ranks.

    local [B_loc, T_loc, D]; local_ids = global unit index of each local row.  Returns a list of
    n_total tensors [T_rank(i), D] (padded rows as the owning rank computed them -- the reference keeps
    padded frames, …base…py:109-113) in global order, identical on every rank."""
    world, rank = _world(group)
    if _skip_collective(world):
        res = [None] * n_total
        for row, gid in enumerate(local_ids):
            res[gid] = local[row]
        return res
    dev = local.device
    meta = torch.tensor([local.shape[0], local.shape[1]], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    bmax = max(int(m[0]) for m in metas)
    tmax = max(int(m[1]) for m in metas)
    D = local.shape[2]
    ids = torch.full((bmax,), -1, dtype=torch.int64, device=dev)
    ids[:len(local_ids)] = torch.tensor(list(local_ids), dtype=torch.int64, device=dev)
    all_ids = torch.empty((world * bmax,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_ids, ids, group=group)
    pad = torch.zeros((bmax, tmax, D), dtype=local.dtype, device=dev)
    pad[:local.shape[0], :local.shape[1]] = local
    payload = torch.empty((world * bmax, tmax, D), dtype=local.dtype, device=dev)
    dist.all_gather_into_tensor(payload, pad, group=group)  # the one large collective
    res = [None] * n_total
    for r in range(world):
        t_r = int(metas[r][1])
        for row in range(int(metas[r][0])):
            gid = int(all_ids[r * bmax + row])
            res[gid] = payload[r * bmax + row, :t_r]
    return res


def encode_sharded(encode_fn: Callable, clips: Sequence, make_batch: Callable, device, group=None, max_batch: int = 2,
                   bucket_by_length: bool = False):
    """Encode `clips` (list of 1-D waveforms) data-parallel and return the per-clip embeddings on every rank.

    encode_fn(input_values, attention_mask) -> [B, T, 768] tensor (the MI355X encoder's forward);
    make_batch(list_of_clips) -> (input_values, attention_mask) on the encoder's device `device`.
    Batches are the reference's (corpus order, `max_batch` = its batch_size = 2) unless bucket_by_length is set."""
    world, rank = _world(group)
    lengths = [len(c) for c in clips]
    if bucket_by_length:
        mine = shard_units(lengths, world, rank)
        chunks = [mine[i:i + max_batch] for i in range(0, len(mine), max_batch)]
        n_rounds = (max(len(shard_units(lengths, world, r)) for r in range(world)) + max_batch - 1) // max_batch
    else:
        chunks = shard_batches(len(clips), max_batch, world, rank)
        n_rounds = rounds(len(clips), max_batch, world)
    results = [None] * len(clips)
    for k in range(n_rounds):
        ids = chunks[k] if k < len(chunks) else []
        if ids:
            x, m = make_batch([clips[i] for i in ids])
            local = encode_fn(x, m)
        else:  # this rank has run out of units: contribute an empty batch to the collective
            local = torch.zeros((0, 1, 768), dtype=torch.float32, device=device)
        for gid, emb in enumerate(gather_ragged(local, ids, len(clips), group)):
            if emb is not None:
                results[gid] = emb
    return results
