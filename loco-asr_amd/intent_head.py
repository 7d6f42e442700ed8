"""Intent head on the embeddings ("next" row f-1): HIP implementation of the reference's
IntentClassifier forward (/root/reference/speech_text/intent_classifier.py:24-49) and of one
training step of train_classifier.py:104-115 (soft-label cross entropy, mean over the batch; Adam lr 1e-3,
weight_decay 1e-4, :61-68), data-parallel over RCCL when a process group exists.

    head = IntentClassifierMI355X(method="attention").to("cuda")
    logits = head(x)                         # [B, 1, 101] like the reference (it squeezes dim 1 itself)
    loss = head.train_step(x, target)        # fwd + bwd (+ all-reduce of 78 437 gradients) + Adam

``state_dict()`` uses the reference's names (``q``, ``classifier.0.weight``, ``classifier.0.bias``) so its
``.pth`` files load unchanged (train_classifier.py:132,163,171,221-222).  No CPU path: device tensors only.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist
from torch import nn

from . import _lib
from . import dp as _dp

METHODS = {"average": 0, "max": 1, "attention": 2, "self_attention": 2}
D, NCLS = 768, 101


class IntentClassifierMI355X(nn.Module):
    def __init__(self, method: str = "average", embedding_size: int = 768, lr: float = 1e-3, weight_decay: float = 1e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__()
        if embedding_size != D:
            raise ValueError("the kernels are specialised for 768-dim SpeechT5 embeddings")
        if method not in METHODS:
            method = "attention"  # the reference treats every other string as self-attention (intent_classifier.py:43-45)
        self.method = method
        self._lib = _lib.load()
        # same initialisation recipe as the reference: q ~ N(0,1)*1e-3, nn.Linear default init
        self.q = nn.Parameter(torch.randn(1, D) * 0.001, requires_grad=False)
        lin = nn.Linear(D, NCLS)
        self.classifier = nn.Sequential(lin)
        for p in self.classifier.parameters():
            p.requires_grad_(False)
        self.hyper = dict(lr=lr, weight_decay=weight_decay, beta1=betas[0], beta2=betas[1], eps=eps)
        self._h = None
        self._dirty = True
        self._ws = None
        self._grads = None
        self.allreduces_issued = 0  # gradient all-reduces of train_step (diagnostics: the DP path really ran)

    # ---- parameter plumbing -------------------------------------------------------------------------------
    def _flat(self) -> torch.Tensor:
        return torch.cat([self.q.detach().reshape(-1), self.classifier[0].weight.detach().reshape(-1),
                          self.classifier[0].bias.detach().reshape(-1)]).float().contiguous()

    def _apply(self, fn, recurse=True):
        self._dirty = True
        return super()._apply(fn, recurse)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        res = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._dirty = True
        return res

    def _ensure(self, device):
        if device.type != "cuda":
            raise RuntimeError("IntentClassifierMI355X runs only on an AMD GPU (no CPU path)")
        if self._h is None:
            with torch.cuda.device(device):
                h = self._lib.loco_head_create(METHODS[self.method])
            if not h:
                raise _lib.LocoError(self._lib.loco_head_last_error().decode())
            self._h = C.c_void_p(h)
            self._dirty = True
        if self._dirty:
            flat = self._flat().to(device)
            self._check(self._lib.loco_head_set_params(self._h, C.c_void_p(flat.data_ptr())))
            self._dirty = False

    def _check(self, rc):
        if rc < 0:
            msg = self._lib.loco_head_last_error().decode()
            raise (ValueError if rc == -1 else _lib.LocoError)(msg)

    def _pull(self, device):
        """copy the library's parameters back into the nn.Parameters (after optimisation steps)"""
        flat = torch.empty(self._lib.loco_head_num_params(), dtype=torch.float32, device=device)
        self._check(self._lib.loco_head_get_params(self._h, C.c_void_p(flat.data_ptr())))
        self.q.data.copy_(flat[:D].view(1, D))
        self.classifier[0].weight.data.copy_(flat[D:D + NCLS * D].view(NCLS, D))
        self.classifier[0].bias.data.copy_(flat[D + NCLS * D:])

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0, group=None):
        """Data parallelism starts from ONE draw of the initial parameters: rank `src`'s flat vector to every rank (one
        broadcast of 78 437 floats)."""
        dev = self.q.device
        if dev.type != "cuda":
            raise RuntimeError("IntentClassifierMI355X runs only on an AMD GPU (no CPU path)")
        flat = self._flat().to(dev)
        dist.broadcast(flat, src=src, group=group)
        self.q.data.copy_(flat[:D].view(1, D))
        self.classifier[0].weight.data.copy_(flat[D:D + NCLS * D].view(NCLS, D))
        self.classifier[0].bias.data.copy_(flat[D + NCLS * D:])
        self._dirty = True

    def state_dict(self, *a, **k):
        if self._h is not None and not self._dirty:
            self._pull(self.q.device)
        return super().state_dict(*a, **k)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.loco_head_destroy(self._h)
        except Exception:
            pass

    def _workspace(self, B, T, device):
        need = int(self._lib.loco_head_workspace_bytes(B, T))
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    @staticmethod
    def _x(x):
        if x.dim() != 3 or x.shape[-1] != D:
            raise ValueError(f"embeddings must be [batch, frames, 768], got {tuple(x.shape)}")
        return x.float().contiguous()

    # ---- forward / training step ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self._x(x)
        dev = x.device
        self._ensure(dev)
        B, T, _ = x.shape
        ws = self._workspace(B, T, dev)
        logits = torch.empty(B, NCLS, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            self._check(self._lib.loco_head_forward(self._h, C.c_void_p(x.data_ptr()), B, T, C.c_void_p(logits.data_ptr()),
                                                    C.c_void_p(ws.data_ptr()), ws.numel(),
                                                    C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return logits.unsqueeze(1)

    @torch.no_grad()
    def loss_and_grads(self, x: torch.Tensor, target: torch.Tensor):
        """(loss scalar tensor, logits [B,101], flat grads [78437]) -- no parameter update."""
        x = self._x(x)
        dev = x.device
        self._ensure(dev)
        B, T, _ = x.shape
        t = target.to(device=dev, dtype=torch.float32).contiguous()
        if t.shape != (B, NCLS):
            raise ValueError(f"target must be [batch, 101], got {tuple(t.shape)}")
        ws = self._workspace(B, T, dev)
        n = self._lib.loco_head_num_params()
        if self._grads is None or self._grads.device != dev:
            self._grads = torch.empty(n, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        logits = torch.empty(B, NCLS, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            self._check(self._lib.loco_head_loss_grad(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(t.data_ptr()), B, T,
                                                      C.c_void_p(loss.data_ptr()), C.c_void_p(logits.data_ptr()),
                                                      C.c_void_p(self._grads.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
                                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss, logits, self._grads

    @torch.no_grad()
    def train_step(self, x: torch.Tensor, target: torch.Tensor, group=None):
        """One step of train_classifier.py:104-115.  With an initialised process group the gradients (and the
        reported loss) are averaged over ranks with one all-reduce before Adam -- data-parallel SGD on the
        global batch."""
        loss, logits, grads = self.loss_and_grads(x, target)
        if dist.is_available() and dist.is_initialized() and not _dp._skip_collective(dist.get_world_size(group)):
            # one all-reduce of the flat gradient buffer (+ the loss); also issued in a process group of ONE rank when
            # dp.FORCE_COLLECTIVE / LOCO_FORCE_COLLECTIVE=1 asks for it (the RCCL path on a one-GPU box): sum / 1 is exact
            w = dist.get_world_size(group)
            buf = torch.cat([grads, loss.reshape(1)])
            dist.all_reduce(buf, group=group)
            buf /= w
            grads.copy_(buf[:-1])
            loss = buf[-1]
            self.allreduces_issued += 1
        hp = self.hyper
        dev = x.device
        with torch.cuda.device(dev):
            self._check(self._lib.loco_head_adam_step(self._h, C.c_void_p(grads.data_ptr()), hp["lr"], hp["beta1"], hp["beta2"],
                                                      hp["eps"], hp["weight_decay"],
                                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss, logits
