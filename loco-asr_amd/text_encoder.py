"""Drop-in for the reference's TEXT branch: ``SpeechT5ForTextToSpeech(...).speecht5.encoder``.

/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:79-93 loads a ``wrapped_encoder`` and a text
``prenet`` state dict into HF's ``SpeechT5EncoderWithTextPrenet`` and calls ``model.speecht5.encoder(texts.input_ids)``
-- token ids only, no attention mask -- keeping ``out.last_hidden_state``.  HF (modeling_speecht5.py,
``SpeechT5TextEncoderPrenet`` / ``SpeechT5ScaledPositionalEncoding`` / ``SpeechT5EncoderWithTextPrenet``):
``hidden = embed_tokens(ids) + alpha * pe[:, :T]`` followed by the same 12-layer encoder the speech path uses.

The arithmetic runs in ``libloco_asr.so`` (``loco_forward_text``); this module only keeps HF's names and call contract.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
from torch import nn

from . import _lib
from .encoder import (HIDDEN, LAYERS, BaseModelOutput, SpeechT5EncoderMI355X, SpeechT5EncoderWithSpeechPrenetMI355X,
                      _Ref, _SpeechT5Core, _WeightHolder)

VOCAB_SIZE = 81           # SpeechT5Config.vocab_size
MAX_TEXT_POSITIONS = 450  # SpeechT5Config.max_text_positions
PAD_TOKEN_ID = 1


def scaled_positional_table(rows: int, dim: int = HIDDEN) -> torch.Tensor:
    """SpeechT5ScaledPositionalEncoding.__init__ -- the same torch expression, so the table is bit-identical to HF's."""
    pe = torch.zeros(rows, dim)
    position = torch.arange(0, rows).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, dim, 2, dtype=torch.int64).float() * -(math.log(10000.0) / dim))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


class SpeechT5TextEncoderPrenetMI355X(_WeightHolder):
    """Parameter names of HF SpeechT5TextEncoderPrenet: ``embed_tokens.weight`` [vocab,768], ``encode_positions.alpha`` []."""

    def __init__(self, owner_ref, vocab_size: int = VOCAB_SIZE):
        super().__init__(owner_ref)
        emb = nn.Module()
        emb.register_parameter("weight", nn.Parameter(torch.zeros(vocab_size, HIDDEN), requires_grad=False))
        self.add_module("embed_tokens", emb)
        pos = nn.Module()
        pos.register_parameter("alpha", nn.Parameter(torch.tensor(1.0), requires_grad=False))
        self.add_module("encode_positions", pos)

    def _translate(self, sd):
        # transformers 4.30.2 (the reference's pin) registers the table as a persistent buffer and the reference's pickled
        # dict carries it (map_speecht5_hf.py:168-181); it is a constant of (max_len, dim) and is rebuilt here
        sd.pop("encode_positions.pe", None)
        return sd


class SpeechT5EncoderWithTextPrenetMI355X(SpeechT5EncoderWithSpeechPrenetMI355X):
    """``forward(input_values=ids [B,T], attention_mask=None, ...) -> BaseModelOutput`` like HF's class of the same name."""

    def __init__(self, layers: int = LAYERS, precision: str = "f16x3", vocab_size: int = VOCAB_SIZE,
                 max_text_positions: int = MAX_TEXT_POSITIONS):
        nn.Module.__init__(self)
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}")
        self.precision = precision
        self._lib = _lib.load()  # raises when the HIP library is missing: no fallback
        ref = _Ref()
        self.prenet = SpeechT5TextEncoderPrenetMI355X(ref, vocab_size)
        self.wrapped_encoder = SpeechT5EncoderMI355X(ref, layers)
        ref.obj = self
        self.num_layers = layers
        self.vocab_size = vocab_size
        self.max_text_positions = max_text_positions
        self._handle = None
        self._handle_device = None
        self._weights_dirty = True
        self._workspace = None
        self._sin_rows = 0
        self._taps = None
        self.streams = 1
        self.last_frames = None
        self.range_policy = "fp32"
        self.last_range_fallback = False
        self._slots = []       # forwards in flight (forward_async), as in the speech encoder
        self._next_slot = 0
        self.eval()

    def _sync_weights(self, device: torch.device, min_sin_rows: int = 0):
        if not self._weights_dirty:
            return
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def put(key, t):
            t = t.detach().to(device=device, dtype=torch.float32).contiguous()
            if t.dim() == 0:
                t = t.reshape(1)
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self._lib.loco_set_weight(self._handle, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()), "load_state_dict")

        for name, p in self.prenet.state_dict().items():
            put("text_prenet." + name, p)
        put("text_prenet.encode_positions.pe", scaled_positional_table(self.max_text_positions))
        for name, p in self.wrapped_encoder.state_dict().items():
            put("wrapped_encoder." + name, p)
        _lib.check(self._lib.loco_finalize_weights(self._handle, stream), "finalize_weights")
        self._weights_dirty = False

    def workspace_bytes(self, batch: int, tokens: int) -> int:
        self._ensure_handle(self._device())
        return int(self._lib.loco_text_workspace_bytes(self._handle, batch, tokens))

    def _check_ids(self, ids, attention_mask):
        if ids.dim() != 2 or ids.dtype.is_floating_point or ids.dtype == torch.bool:
            raise ValueError(f"input_values must be integer token ids [batch, tokens], got {ids.dtype} {tuple(ids.shape)}")
        B, T = ids.shape
        if T < 1 or B < 1:
            raise ValueError("empty batch")
        if T > self.max_text_positions:
            raise ValueError(f"{T} tokens exceed max_text_positions = {self.max_text_positions}")
        lo, hi = int(ids.min()), int(ids.max())
        if lo < 0 or hi >= self.vocab_size:
            raise IndexError(f"token id out of range [0, {self.vocab_size}): min {lo}, max {hi}")  # nn.Embedding raises IndexError too
        m = None
        if attention_mask is not None:
            if attention_mask.shape != ids.shape:
                raise ValueError(f"attention_mask {tuple(attention_mask.shape)} does not match input_values {tuple(ids.shape)}")
            m = attention_mask.to(device=ids.device, dtype=torch.int32).contiguous()
            if T > 1 and not bool((m[:, 1:] <= m[:, :-1]).all()):
                raise NotImplementedError("attention_mask must be right padding (ones then zeros), as the tokenizer produces it")
        return ids.to(torch.int32).contiguous(), m

    # -- several batches of transcripts in flight (the reference's text loop is batch_size = 2 as well, …base…py:67-68,79-93) --------
    def _enqueue(self, slot, ids32, m, out, frames, precision, pack=None, hidden=None):  # pack / hidden: unused (a text pack is an ordinary masked forward)
        B, T = ids32.shape
        need = int(self._lib.loco_text_workspace_bytes(self._handle, B, T))
        if slot.workspace is None or slot.workspace.numel() < need:
            slot.workspace = None
            slot.workspace = torch.empty(need, dtype=torch.uint8, device=ids32.device)
        _lib.check(self._lib.loco_forward_text_async(
            self._handle, self.PRECISIONS[precision], C.c_void_p(ids32.data_ptr()), C.c_void_p(m.data_ptr()) if m is not None else None, B, T,
            C.c_void_p(out.data_ptr()), C.c_void_p(frames.data_ptr()), None, C.c_void_p(slot.workspace.data_ptr()), slot.workspace.numel(),
            C.c_void_p(slot.stream.cuda_stream), C.c_void_p(slot.status.data_ptr())), "loco_forward_text_async")

    @torch.no_grad()
    def forward_async(self, input_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **kwargs):
        """Enqueue one text forward on the next slot; ``ticket.result()`` is the BaseModelOutput (see the speech encoder's forward_async)."""
        from .encoder import ForwardTicket
        if self.training:
            raise RuntimeError("the MI355X encoder path is inference-only; call .eval()")
        device = input_values.device
        self._ensure_handle(device)
        if self._device() != device:
            raise RuntimeError(f"module parameters are on {self._device()} but input_values on {device}")
        if self._weights_dirty:
            self.drain()
        if not self._slots:
            self.set_inflight(2)
        ids32, m = self._check_ids(input_values, attention_mask)
        return self._submit_text(ids32, m)

    @torch.no_grad()
    def forward_packed_async(self, batches=None, *, packed=None):
        """Several of the reference's text batches (…base…py:79-93: ids padded to the batch's longest transcript with <pad> = 1,
        NO attention mask -- the pads of a batch attend like tokens) as ONE forward: rows of all batches side by side, padded to the
        pack's longest, and a key mask that ends every row where ITS OWN batch ends -- which is all a text batch's composition
        means here (no GroupNorm, no positional conv: the text prenet is row-wise, positions count from 0 in every row).  A batch
        given as a mapping with ``attention_mask`` (right padding) keeps that mask.  ``ticket.result()`` = one BaseModelOutput per
        batch, ``last_hidden_state`` [B_i, T_i, 768] equal to the batch's own forward up to the fp32 summation order of the GEMMs."""
        from .encoder import Pack
        if self.training:
            raise RuntimeError("the MI355X encoder path is inference-only; call .eval()")
        if not batches:
            raise ValueError("forward_packed: no batches")
        device = self._device()
        self._ensure_handle(device)
        if self._weights_dirty:
            self.drain()
        if not self._slots:
            self.set_inflight(2)
        ids_l = [b["input_values"] if hasattr(b, "keys") else b for b in batches]
        msk_l = [b.get("attention_mask") if hasattr(b, "keys") else None for b in batches]
        B, T = sum(int(i.shape[0]) for i in ids_l), max(int(i.shape[1]) for i in ids_l)
        ids = torch.full((B, T), 1, dtype=torch.int64)
        mask = torch.zeros((B, T), dtype=torch.int32)
        spans, b0 = [], 0
        for i, mk in zip(ids_l, msk_l):
            nb, t = int(i.shape[0]), int(i.shape[1])
            ids[b0:b0 + nb, :t] = i.to("cpu")
            mask[b0:b0 + nb, :t] = 1 if mk is None else mk.to("cpu", torch.int32)
            spans.append((b0, nb, t))
            b0 += nb
        ids32, m = self._check_ids(ids.to(device), mask.to(device))
        return self._submit_text(ids32, m, Pack(wav=ids32, mask=m, valid_len=None, pad_len=[], spans=spans))

    def forward_packed(self, batches=None, *, packed=None):
        return self.forward_packed_async(batches).result()

    def _submit_text(self, ids32, m, pack=None):
        from .encoder import ForwardTicket
        device = ids32.device
        B, T = ids32.shape
        slot = self._slots[self._next_slot]
        self._next_slot = (self._next_slot + 1) % len(self._slots)
        if slot.ticket is not None:
            slot.ticket.settle()
        with torch.cuda.device(device):
            self._sync_weights(device)
            cur = torch.cuda.current_stream(device)
            # outputs come from the CALLER's stream (see the speech encoder's _submit): no record_stream duty for consumers
            out = torch.empty((B, T, HIDDEN), dtype=torch.float32, device=device)
            frames = torch.empty((B,), dtype=torch.int32, device=device)
            slot.stream.wait_stream(cur)
            with torch.cuda.stream(slot.stream):
                for t_ in (ids32, m, out, frames):
                    if t_ is not None:
                        t_.record_stream(slot.stream)
                ticket = ForwardTicket(self, slot, ids32, m, out, frames, self.precision, pack)
                self._enqueue(slot, ids32, m, out, frames, self.precision)
                ticket._done.record(slot.stream)
        slot.ticket = ticket
        return ticket

    def forward(self, input_values: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None, **kwargs):
        if self.training:
            raise RuntimeError("the MI355X encoder path is inference-only; call .eval()")
        if output_attentions:
            raise NotImplementedError("output_attentions=True: the flash-style attention kernel never forms the [T,T] weights")
        ids = input_values
        device = ids.device
        self._ensure_handle(device)
        if self._device() != device:
            raise RuntimeError(f"module parameters are on {self._device()} but input_values on {device}")
        ids32, m = self._check_ids(ids, attention_mask)
        B, T = ids32.shape
        with torch.cuda.device(device):
            self._sync_weights(device)
            _lib.check(self._lib.loco_set_precision(self._handle, self.PRECISIONS[self.precision]), "set_precision")
            need = int(self._lib.loco_text_workspace_bytes(self._handle, B, T))
            if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
                self._workspace = None
                self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
            out = torch.empty((B, T, HIDDEN), dtype=torch.float32, device=device)
            frames = torch.empty((B,), dtype=torch.int32, device=device)
            hs, hs_ptrs = None, None
            if output_hidden_states:
                hs = [torch.empty_like(out) for _ in range(self.num_layers + 1)]
                hs_ptrs = (C.c_void_p * (self.num_layers + 1))(*[t.data_ptr() for t in hs])
            stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

            def launch():
                _lib.check(self._lib.loco_forward_text(self._handle, C.c_void_p(ids32.data_ptr()),
                                                       C.c_void_p(m.data_ptr()) if m is not None else None, B, T,
                                                       C.c_void_p(out.data_ptr()), C.c_void_p(frames.data_ptr()), hs_ptrs,
                                                       C.c_void_p(self._workspace.data_ptr()), self._workspace.numel(), stream),
                           "loco_forward_text")

            launch()
            # numeric range of precision "f16x3" (include/loco_asr.h): same policy as the speech encoder's
            self.last_range_fallback = False
            policy = self.range_policy
            if policy != "off" and self.precision != "f32":
                torch.cuda.current_stream(device).synchronize()
                rc = self._lib.loco_forward_status(self._handle, None, 0)
                if rc != 0:
                    if policy == "raise":
                        _lib.check(rc, "loco_forward_text")
                    _lib.check(self._lib.loco_set_precision(self._handle, self.PRECISIONS["f32"]), "set_precision")
                    launch()  # the same batch on the exact-fp32 kernels
                    self.last_range_fallback = True
        self.last_frames = frames
        hidden = tuple(hs) if hs is not None else None
        if return_dict is False:
            return tuple(v for v in (out, hidden) if v is not None)
        return BaseModelOutput(last_hidden_state=out, hidden_states=hidden, attentions=None)


class SpeechT5ForTextToSpeechMI355X(nn.Module):
    """Only as much of HF's SpeechT5ForTextToSpeech as the reference touches: ``.speecht5.encoder`` (…base…py:80-86)."""

    def __init__(self, layers: int = LAYERS, precision: str = "f16x3"):
        super().__init__()
        self.speecht5 = _SpeechT5Core(SpeechT5EncoderWithTextPrenetMI355X(layers, precision))
        self.eval()

    @classmethod
    def from_state_dicts(cls, text_prenet_state_dict, encoder_state_dict, layers: int = LAYERS, precision: str = "f16x3"):
        model = cls(layers, precision)
        model.speecht5.encoder.wrapped_encoder.load_state_dict(encoder_state_dict)
        model.speecht5.encoder.prenet.load_state_dict(text_prenet_state_dict)
        return model
