"""MI355X augmented localizer of the seq2seq Aether.

Mirrors ``nn.utils.augmented_global_to_local.AugmentedLocalizer`` of the reference
(augmented_global_to_local.py:11-68): same constructor, ``set_edge_index`` and
``forward(x) -> (rel_feat, Rinv, edge_attr, edge_pos)`` for ``x [B, N, 3D]`` (pos | vel | force), the
fully connected per-graph edge index of ``torch.where(~eye(N))`` by default.  The computation runs in
libaether_hip.so (``aether_s2s_localize``); there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib


class AugmentedLocalizer(nn.Module):
    def __init__(self, num_objects, use_3d=False, pos_representation="polar"):
        super().__init__()
        if pos_representation not in ("cart", "polar"):
            raise ValueError
        self.use_3d = bool(use_3d)
        self.num_objects = num_objects
        self.pos_representation = pos_representation
        self.send_edges, self.recv_edges = torch.where(~torch.eye(num_objects, dtype=bool))   # :31-32
        self.num_dims = 3 if self.use_3d else 2
        self.num_orientations = self.num_dims * (self.num_dims - 1) // 2
        self.num_relative_features = 4 * self.num_dims + self.num_orientations
        self.num_pos_features = self.num_dims + self.num_orientations
        self._global_edges = {}

    def set_edge_index(self, send_edges, recv_edges):
        self.send_edges, self.recv_edges = send_edges, recv_edges
        self._global_edges = {}

    def _edges_for(self, B, N, device):
        key = (B, N, str(device))
        hit = self._global_edges.get(key)
        if hit is None:
            off = (torch.arange(B, device=device, dtype=torch.int64) * N).unsqueeze(1)
            send = (self.send_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            recv = (self.recv_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            hit = self._global_edges[key] = (send, recv)
        return hit

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd AugmentedLocalizer runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        D = self.num_dims
        if x.dim() != 3 or x.shape[-1] != 3 * D:
            raise ValueError(f"x must be [B, N, {3 * D}] (pos | vel | force)")
        lib = _lib.load()
        B, N, _ = x.shape
        E1 = self.recv_edges.shape[0]
        nf = self.num_relative_features
        xf = x.detach().to(torch.float32).reshape(B * N, 3 * D).contiguous()
        send, recv = self._edges_for(B, N, x.device)
        rel_feat = torch.empty(B, N, 3 * D + nf, dtype=torch.float32, device=x.device)
        Rinv = torch.empty(B, N, D, D, dtype=torch.float32, device=x.device)
        edge_attr = torch.empty(B, E1, 2 * nf + 3 * D, dtype=torch.float32, device=x.device)
        edge_pos = torch.empty(B, E1, self.num_pos_features, dtype=torch.float32, device=x.device)
        if B * N == 0:
            return rel_feat, Rinv, edge_attr, edge_pos
        st = lib.aether_s2s_localize(D, B * N, B * E1, xf.data_ptr(), send.data_ptr(), recv.data_ptr(),
                                     1 if self.pos_representation == "polar" else 0, rel_feat.data_ptr(),
                                     Rinv.data_ptr(), edge_attr.data_ptr(), edge_pos.data_ptr(),
                                     torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_s2s_localize")
        return rel_feat, Rinv, edge_attr, edge_pos
