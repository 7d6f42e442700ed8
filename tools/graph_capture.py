"""Stream capture of the C-ABI forward into a hipGraph -- a TEST / PROBE utility, not a product feature.

The property it exercises is the boundary's: loco_forward_async enqueues on the caller's stream (plus, for large batches, a second
stream forked from and joined back to it with events), allocates nothing and never synchronises, so it can be captured and the
replay is bit-identical.  What it does NOT give on this path is time: one trace (profiles/r03_hipgraph_trace.txt) shows the replay of
the 5 s utterance at 176 kernels / 1 904 us of kernel time / 2.017 ms per forward against 172 / 1 862 / 2.016 eager -- the forward is
bound by its chain of small dependent kernels on the GPU, not by the host's launches; and real corpora are ragged (every batch its own
shape), so a captured graph would not be reused anyway.  The encoder module therefore has no use_graphs switch any more."""
import ctypes as C

import torch


class CapturedForward:
    def __init__(self, enc, x, m):
        """enc: SpeechT5EncoderWithSpeechPrenetMI355X (weights loaded, at least one eager forward of this shape done so that the
        sinusoid table and the weights are final); x [B, L] fp32, m [B, L] int32 or None, on the GPU."""
        lib = enc._lib
        B, L = x.shape
        T = int(lib.loco_output_frames(L))
        self.enc, self.x, self.m = enc, torch.empty_like(x), (torch.empty_like(m) if m is not None else None)
        self.x.copy_(x)
        if m is not None:
            self.m.copy_(m)
        self.out = torch.empty((B, T, 768), dtype=torch.float32, device=x.device)
        self.frames = torch.empty((B,), dtype=torch.int32, device=x.device)
        self.ws = torch.empty(int(lib.loco_workspace_bytes(enc._handle, B, L)), dtype=torch.uint8, device=x.device)
        self.status = torch.zeros(int(lib.loco_status_bytes()), dtype=torch.uint8).pin_memory()
        self._args = (B, L)
        self._launch()  # eager once
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._launch()

    def _launch(self):
        enc, (B, L) = self.enc, self._args
        rc = enc._lib.loco_forward_async(enc._handle, enc.PRECISIONS[enc.precision], C.c_void_p(self.x.data_ptr()),
                                         C.c_void_p(self.m.data_ptr()) if self.m is not None else None, B, L, C.c_void_p(self.out.data_ptr()),
                                         C.c_void_p(self.frames.data_ptr()), None, C.c_void_p(self.ws.data_ptr()), self.ws.numel(),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(self.status.data_ptr()))
        assert rc == 0, enc._lib.loco_last_error()

    def replay(self, x=None, m=None):
        if x is not None:
            self.x.copy_(x)
        if m is not None:
            self.m.copy_(m)
        self.graph.replay()
        return self.out

    def status_code(self):
        """after the stream has completed the replay: the captured forward wrote ITS OWN status block"""
        return int(self.enc._lib.loco_status_check(C.c_void_p(self.status.data_ptr()), None, 0))
