"""MI355X field query of the seq2seq Aether.

Mirrors ``Aether.predict_field`` of the reference's seq2seq model (nn/seq2seq/aether.py:86-90): random
Fourier features of the positions (nn/nn/fourier_feature_mapper.py:7-21) followed by
``field_net`` (aether.py:72-78).  The module's ``state_dict`` carries exactly the reference model's keys
for this part -- ``coordinate_embedding.B``, ``field_net.{0,2,4}.{weight,bias}`` -- so
``load_state_dict(ref_sd, strict=False)`` picks them out of a seq2seq checkpoint.  The computation
runs in libaether_hip.so (``aether_s2s_field``); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ... import _lib


class _S2SFieldParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("B", "w0", "b0", "w2", "b2", "w4", "b4")]


class _CoordinateEmbedding(nn.Module):
    """Holder of the ``B`` buffer (fourier_feature_mapper.py:12-15: default_rng(42).normal(0, std))."""

    def __init__(self, in_size, out_size, std=1.0):
        super().__init__()
        rng = np.random.default_rng(42)
        self.register_buffer("B", torch.from_numpy(rng.normal(0, std, size=(in_size, out_size))).float())


class FieldQuery(nn.Module):
    """``predict_field``: ``x[..., :D]`` -> (field ``[..., D]``, coords), as aether.py:86-90."""

    def __init__(self, num_dims=2, hidden_size=512, rff_std=1.0, device="cuda"):
        super().__init__()
        if hidden_size % 32 != 0:
            raise ValueError("hidden_size must be a multiple of 32")        # the reference requires it even (aether.py:80-81)
        if num_dims not in (2, 3):
            raise ValueError("num_dims must be 2 or 3")
        self.num_dims, self.hidden_size = num_dims, hidden_size
        self.field_net = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.SiLU(),
                                       nn.Linear(hidden_size, hidden_size), nn.SiLU(),
                                       nn.Linear(hidden_size, num_dims))
        self.coordinate_embedding = _CoordinateEmbedding(num_dims, hidden_size // 2, rff_std)
        self._ws = None
        self.to(device)

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd FieldQuery runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        D, h = self.num_dims, self.hidden_size
        if x.shape[-1] < D:
            raise ValueError(f"last dimension of x must hold at least {D} coordinates")
        coords = x[..., :D]
        pts = x.detach().to(torch.float32).reshape(-1, x.shape[-1]).contiguous()
        n = pts.shape[0]
        out = torch.empty(n, D, dtype=torch.float32, device=x.device)
        if n == 0:
            return out.reshape(*x.shape[:-1], D), coords
        need = lib.aether_s2s_field_workspace_bytes(n, h)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        fn = self.field_net
        tensors = [self.coordinate_embedding.B, fn[0].weight, fn[0].bias, fn[2].weight, fn[2].bias,
                   fn[4].weight, fn[4].bias]
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _lib.AetherHipError("FieldQuery parameters must be contiguous fp32 CUDA tensors")
        ps = _S2SFieldParams(*[t.data_ptr() for t in tensors])
        st = lib.aether_s2s_field(C.byref(ps), D, h, n, pts.data_ptr(), pts.shape[1], self._ws.data_ptr(),
                                  self._ws.numel(), out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_s2s_field")
        return out.reshape(*x.shape[:-1], D), coords
