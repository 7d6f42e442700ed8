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

    def __init__(self, group=None, timed: bool = False):
        self.group = group
        self._work = None
        self._bufs = [None, None]
        self._i = 0
        self.result = None
        # timed: every wait is bracketed by two events on the stream that waits -- nothing else runs between them there, so their
        # distance is how long that stream stood still for the collective (0 when the gather had finished under the next forward)
        self.timed = timed
        self._waits = []
        self.gathers = 0
        self.bytes_gathered = 0

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
        self.gathers += 1
        self.bytes_gathered += buf.numel() * buf.element_size()

    def finish(self):
        if self._work is not None:
            if self.timed and self._pending.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self._work.wait()
                b.record()
                self._waits.append((a, b))
            else:
                self._work.wait()
            self._work = None
            self.result = self._pending
        return self.result

    def blocked_ms(self, reset: bool = True) -> float:
        """Sum over the waits so far of the time the waiting stream stood still (timed=True; synchronises those events)."""
        total = 0.0
        for a, b in self._waits:
            b.synchronize()
            total += a.elapsed_time(b)
        if reset:
            self._waits = []
        return total


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


def _on(stream):
    import contextlib
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


class RaggedGatherPipeline:
    """``gather_ragged`` for a LOOP of rounds, overlapped with the rounds that follow (extract.py --gather; configs[3]: windows of
    a podcast dealt to 8 ranks, every rank's embeddings gathered).  Per round TWO collectives instead of three, none of them
    waited for in the round that issues it, and no host read of device metadata on the critical path:

      round k    submit(local_k)  ->  all_gather_into_tensor of ONE int64 row per rank [B_loc, T_loc, ids[cap], rows[cap]], async
      round k+1  the metadata of round k has long arrived: read it (a round late, so the read waits for nothing), pad local_k into
                 a reused send buffer [b_max, t_max, D] and start the payload all_gather_into_tensor, async
      round k+2  wait for the payload of round k (issued a round ago) and hand its rows out

    ``submit`` returns the finished rows of the round two calls back as a list of (global unit index, tensor [rows, D]) -- views
    into that round's receive buffer, identical on every rank -- and ``flush`` drains the last two rounds.  ``cap`` = the largest
    number of units any rank contributes in one round (batch size x batches per round): it fixes the metadata row, so that no
    shape has to be agreed first.  ``rows[i]`` (default: all T_loc) are the frames of unit i worth keeping: a packed forward's
    clips end where their own reference batch ends.  Send buffers are reused from round to round; a receive buffer is allocated
    per round (torch's caching allocator: no driver call in steady state) because its consumers -- the sink's D2H copies on their
    own stream -- outlive the round and protect it with record_stream, which a fixed ring could not honour without a release
    handshake.  Works with any backend (gloo on CPU in tests/test_dp_gloo.py, RCCL on device tensors)."""

    def __init__(self, cap: int, dim: int = 768, group=None):
        self.cap, self.dim, self.group = int(cap), int(dim), group
        self.world, self.rank = _world(group)
        self.skip = _skip_collective(self.world)
        self._stages = []        # rounds in flight, oldest first
        self._send = [None, None]
        self._send_work = [None, None]
        self._k = 0
        self._stream = None      # collectives and their staging run beside the caller's stream
        self.collectives = 0

    def _side(self, device):
        if device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device)
        return self._stream

    def _start_meta(self, local, ids, rows, device):
        n = 0 if local is None else int(local.shape[0])
        if n > self.cap:
            raise ValueError(f"{n} units in one round > cap {self.cap}")
        t = 0 if local is None else int(local.shape[1])
        row = torch.full((2 + 2 * self.cap,), -1, dtype=torch.int64)
        row[0], row[1] = n, t
        if n:
            row[2:2 + n] = torch.tensor(list(ids), dtype=torch.int64)
            row[2 + self.cap:2 + self.cap + n] = torch.tensor([t] * n if rows is None else [int(r) for r in rows], dtype=torch.int64)
        st = dict(local=local, k=self._k, ready=None)
        self._k += 1
        side = self._side(device)
        if side is not None and local is not None:  # whatever produced `local` on the caller's stream comes first
            st["ready"] = torch.cuda.Event()
            st["ready"].record(torch.cuda.current_stream(device))
        with _on(side):
            meta = row.to(device) if device.type == "cuda" else row
            meta_all = torch.empty((self.world * row.numel(),), dtype=torch.int64, device=device)
            st["meta"], st["meta_all"] = meta, meta_all
            st["meta_work"] = dist.all_gather_into_tensor(meta_all, meta, group=self.group, async_op=True)
        self.collectives += 1
        return st

    def _start_payload(self, st, device):
        side = self._side(device)
        with _on(side):
            st["meta_work"].wait()
            mh = st["meta_all"].cpu().view(self.world, -1)  # issued a round ago: complete; on the side stream, so nothing else is waited for
            st["meta_host"] = mh
            bmax, tmax = max(1, int(mh[:, 0].max())), max(1, int(mh[:, 1].max()))
            slot = st["k"] & 1
            if self._send_work[slot] is not None:
                self._send_work[slot].wait()  # the collective that last read this send buffer (two rounds ago: long complete)
            buf = self._send[slot]
            need = bmax * tmax * self.dim
            if buf is None or buf.numel() < need or buf.device != device:
                buf = torch.empty(need, dtype=torch.float32, device=device)
                self._send[slot] = buf
            pad = buf[:need].view(bmax, tmax, self.dim)
            local = st["local"]
            if local is not None:
                if st["ready"] is not None:
                    side.wait_event(st["ready"])
                pad[:local.shape[0], :local.shape[1]].copy_(local)
                if side is not None:
                    local.record_stream(side)
            out = torch.empty((self.world * bmax, tmax, self.dim), dtype=torch.float32, device=device)
            st["out"], st["bmax"] = out, bmax
            st["pay_work"] = dist.all_gather_into_tensor(out, pad, group=self.group, async_op=True)
            self._send_work[slot] = st["pay_work"]
            st["local"] = None
        self.collectives += 1

    def _finish(self, st):
        st["pay_work"].wait()  # on the CALLER's current stream: what it enqueues next sees the gathered rows
        mh, bmax, out = st["meta_host"], st["bmax"], st["out"]
        if out.is_cuda:
            out.record_stream(torch.cuda.current_stream(out.device))  # allocated on the side stream, consumed on the caller's
        res = []
        for r in range(self.world):
            for row in range(int(mh[r, 0])):
                res.append((int(mh[r, 2 + row]), out[r * bmax + row, :int(mh[r, 2 + self.cap + row])]))
        return res

    def submit(self, local, ids, rows=None, device=None):
        """Start the gather of this round's [B_loc, T_loc, D] (``None`` / empty ids when this rank has run out of units) and
        return the finished rows of the round two calls back."""
        if local is not None and local.shape[0] == 0:
            local = None
        if self.skip:
            if local is None:
                return []
            return [(int(g), local[i, :(local.shape[1] if rows is None else int(rows[i]))]) for i, g in enumerate(ids)]
        device = local.device if local is not None else torch.device(device if device is not None else "cpu")
        if local is not None and local.is_cuda:
            local = local.contiguous()
        done = []
        if len(self._stages) == 2:
            done = self._finish(self._stages.pop(0))
        if self._stages:
            self._start_payload(self._stages[-1], device)
        self._stages.append(self._start_meta(local, ids if local is not None else [], rows, device))
        return done

    def flush(self, device=None):
        """Drain the rounds still in flight (call once, after the last submit, on every rank)."""
        done = []
        while self._stages:
            st = self._stages[0]
            dev = st["meta_all"].device
            if "pay_work" not in st:
                self._start_payload(st, dev)
            done += self._finish(self._stages.pop(0))
        return done


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
