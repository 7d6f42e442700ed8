"""Host-side waveform batching with the contract of HF's SpeechT5FeatureExtractor for raw audio.

The reference's collate_fn calls ``processor(audio=audios, sampling_rate=16000, return_tensors="pt",
padding="longest").to(device)`` (/root/reference/speech_text/extract_speecht5_base_embeddings_slurp.py:60),
which dispatches to ``SpeechT5FeatureExtractor._process_audio``
(transformers/models/speecht5/feature_extraction_speecht5.py:275-360): zero-pad every clip on the right to
the longest one, return ``input_values`` f32 [B, L] and ``attention_mask`` i32 [B, L]; when
``do_normalize`` is set, each clip is first shifted/scaled to zero mean and unit variance over its
UNPADDED samples with epsilon 1e-7 (feature_extraction_speecht5.py:119-138).  The class default is
``do_normalize=False`` (:77); the hub checkpoint's value cannot be checked offline, so both are supported.

Padding stays Python/numpy on the host exactly as in the reference; it hands over pinned host tensors so the
H2D copy in ``.to(device)`` can overlap compute.  With ``normalize_on_device=True`` the zero-mean/unit-variance step is
deferred to the GPU (``loco_op_normalize_waveform``): ``BatchFeature.to("cuda")`` runs it in place
right after the copy, so ``processor(...).to(device)`` keeps the reference's shape and the host does no arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch


def normalize_waveform_(input_values: torch.Tensor, attention_mask=None, padding_value: float = 0.0) -> torch.Tensor:
    """In-place ``zero_mean_unit_var_norm`` of a [B, L] fp32 batch ON THE GPU (HF feature_extraction_speecht5.py:119-138)."""
    import ctypes as C

    from . import _lib
    if input_values.device.type != "cuda":
        raise RuntimeError("normalize_waveform_ runs on the GPU only (the host path is SpeechT5FeatureExtractorMI355X(do_normalize=True))")
    if input_values.dim() != 2 or input_values.dtype != torch.float32 or not input_values.is_contiguous():
        raise ValueError("input_values must be a contiguous fp32 [batch, samples] tensor")
    lib = _lib.load()
    B, L = input_values.shape
    m = None
    if attention_mask is not None:
        if attention_mask.shape != input_values.shape:
            raise ValueError("attention_mask does not match input_values")
        m = attention_mask.to(device=input_values.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(input_values.device):
        scratch = torch.empty(int(lib.loco_normalize_scratch_bytes(B)), dtype=torch.uint8, device=input_values.device)
        stream = C.c_void_p(torch.cuda.current_stream(input_values.device).cuda_stream)
        _lib.check(lib.loco_op_normalize_waveform(C.c_void_p(input_values.data_ptr()), C.c_void_p(m.data_ptr()) if m is not None else None,
                                                  B, L, float(padding_value), C.c_void_p(input_values.data_ptr()),
                                                  C.c_void_p(scratch.data_ptr()), scratch.numel(), stream), "normalize_waveform")
    return input_values


class BatchFeature(dict):
    """dict with ``.to(device)`` and attribute access, so ``encoder(**audios)`` works as in the reference."""

    _pending_normalize = None  # padding_value when the normalisation was deferred to the device

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device, non_blocking: bool = True):
        out = BatchFeature({k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v) for k, v in self.items()})
        if self._pending_normalize is not None:
            if torch.device(device).type == "cuda":
                normalize_waveform_(out["input_values"], out.get("attention_mask"), self._pending_normalize)
            else:
                out._pending_normalize = self._pending_normalize  # still owed; only a GPU can pay it
        return out


class SpeechT5FeatureExtractorMI355X:
    model_input_names = ["input_values", "attention_mask"]

    def __init__(self, sampling_rate: int = 16000, padding_value: float = 0.0, do_normalize: bool = False,
                 return_attention_mask: bool = True, pin_memory: bool = False, normalize_on_device: bool = False):
        self.sampling_rate = sampling_rate
        self.padding_value = padding_value
        self.do_normalize = do_normalize
        self.normalize_on_device = normalize_on_device  # defer do_normalize to BatchFeature.to("cuda")
        self.return_attention_mask = return_attention_mask
        self.pin_memory = pin_memory

    def _device_batch(self, clips):
        """padding="longest" for clips that live on the GPU: same contract (zero right-padding, int32 mask); do_normalize runs
        there too (loco_op_normalize_waveform, HF feature_extraction_speecht5.py:119-138)."""
        dev = clips[0].device
        clips = [c.to(torch.float32).reshape(-1) for c in clips]
        lmax = max(int(c.numel()) for c in clips)
        x = torch.full((len(clips), lmax), float(self.padding_value), dtype=torch.float32, device=dev)
        m = torch.zeros((len(clips), lmax), dtype=torch.int32, device=dev)
        for i, c in enumerate(clips):
            x[i, :c.numel()] = c
            m[i, :c.numel()] = 1
        if self.do_normalize:
            normalize_waveform_(x, m, self.padding_value)
        out = BatchFeature(input_values=x)
        if self.return_attention_mask:
            out["attention_mask"] = m
        return out

    def pack_clips(self, batches, device, output_frames):
        """The fused form of ``[self(audio=b, padding="longest") for b in batches]`` + ``encoder.pack_batches`` for clips that live
        on the host: every batch is padded to ITS OWN longest clip exactly as ``__call__`` pads it (zeros = padding_value 0 only;
        the attention mask of padding="longest" is "ones up to the clip's length", carried as valid_len -- HF reduces the mask to
        that number anyway, modeling_speecht5.py:569-582; do_normalize per clip over its unpadded samples), but the clips are
        written ONCE, straight into the pack's pinned [B, L] buffer, instead of into a per-batch tensor and from there into the
        pack.  ``batches`` = list of lists of 1-D float arrays; ``output_frames`` = loco_output_frames.  Returns encoder.Pack."""
        from .encoder import Pack
        if float(self.padding_value) != 0.0:
            raise ValueError("pack_clips pads with zeros (the reference's processor default); use __call__ + pack_batches otherwise")
        if self.do_normalize and self.normalize_on_device:
            raise ValueError("pack_clips normalises on the host; use __call__ + pack_batches for normalize_on_device")
        groups = [[np.asarray(c, dtype=np.float32).reshape(-1) for c in b] for b in batches]
        if not groups or any(not g for g in groups):
            raise ValueError("empty batch")
        if self.do_normalize:
            groups = [[self.zero_mean_unit_var_norm(c) for c in g] for g in groups]
        B = sum(len(g) for g in groups)
        lens = [max(len(c) for c in g) for g in groups]
        L = (max(lens) + 7) // 8 * 8
        wav = torch.empty((B, L), dtype=torch.float32, pin_memory=bool(torch.cuda.is_available()))
        w = wav.numpy()
        pad_len, valid_len, spans, b0 = [], [], [], 0
        for g, li in zip(groups, lens):
            t = int(output_frames(li))
            if t < 1:
                raise ValueError(f"input of {li} samples is shorter than one encoder frame (400 samples)")
            for c in g:
                n = len(c)
                w[b0, :n] = c
                w[b0, n:] = 0.0
                valid_len.append(n)
                pad_len.append(li)
                b0 += 1
            spans.append((b0 - len(g), len(g), t))
        return Pack(wav=wav.to(device, non_blocking=True), mask=None, valid_len=valid_len if self.return_attention_mask else None,
                    pad_len=pad_len, spans=spans)

    @staticmethod
    def zero_mean_unit_var_norm(x: np.ndarray) -> np.ndarray:
        return ((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32)

    def __call__(self, audio=None, sampling_rate=None, return_tensors="pt", padding="longest", **_):
        if audio is None:
            raise ValueError("You must provide `audio`.")  # same message class as HF's ValueError
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"The model was trained at {self.sampling_rate} Hz; got sampling_rate={sampling_rate}.")
        if isinstance(audio, np.ndarray) and audio.ndim == 1 or torch.is_tensor(audio) and audio.dim() == 1:
            audio = [audio]
        if len(audio) and all(torch.is_tensor(a) and a.is_cuda for a in audio):
            return self._device_batch(list(audio))  # clips already on the GPU (the device resampler's output): pad there
        clips = [np.asarray(a.cpu() if torch.is_tensor(a) else a, dtype=np.float32).reshape(-1) for a in audio]
        if not clips:
            raise ValueError("empty batch")
        defer = self.do_normalize and self.normalize_on_device
        if defer and (return_tensors == "np" or not self.return_attention_mask):
            raise ValueError("normalize_on_device needs torch tensors and the attention mask (the device op reads the clip lengths from it)")
        if self.do_normalize and not defer:
            clips = [self.zero_mean_unit_var_norm(c) for c in clips]
        lmax = max(len(c) for c in clips) if padding in ("longest", True) else None
        if lmax is None:
            if len({len(c) for c in clips}) != 1:
                raise ValueError("clips differ in length; use padding='longest'")
            lmax = len(clips[0])
        B = len(clips)
        pin = bool(self.pin_memory and torch.cuda.is_available())  # written straight into pinned memory (torch caches these blocks)
        x = torch.empty((B, lmax), dtype=torch.float32, pin_memory=pin)
        m = torch.empty((B, lmax), dtype=torch.int32, pin_memory=pin)
        xn, mn = x.numpy(), m.numpy()  # filled through numpy views: a fifth of the interpreter time of torch slice assignments
        for i, c in enumerate(clips):
            n = len(c)
            xn[i, :n] = c
            xn[i, n:] = self.padding_value
            mn[i, :n] = 1
            mn[i, n:] = 0
        out = BatchFeature(input_values=x)
        if self.return_attention_mask:
            out["attention_mask"] = m
        if return_tensors == "np":
            out = BatchFeature({k: v.numpy() for k, v in out.items()})
        if defer:
            out._pending_normalize = float(self.padding_value)
        return out
