"""MI355X recurrent decoder step of the seq2seq Aether (decoder half).

Mirrors ``nn.seq2seq.aether.RecurrentDecoder`` of the reference (aether.py:505-654): same ``params``
dictionary, parameters created in the same order with the same shapes (so the same torch seed gives the
same initial weights and ``state_dict`` keys / order match a reference checkpoint), same
``get_initial_hidden`` and ``forward(inputs, hidden, edges, predicted_field) -> (outputs, hidden)``.
The computation runs in libaether_hip.so (``aether_s2s_decoder_step``); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from .localizer import AugmentedLocalizer


class _DecoderParams(C.Structure):
    _fields_ = ([("msg_fc1_w", C.c_void_p * 4), ("msg_fc1_b", C.c_void_p * 4),
                 ("msg_fc2_w", C.c_void_p * 4), ("msg_fc2_b", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("hidden_r_w", "hidden_i_w", "hidden_h_w", "present_r_w", "present_r_b",
                                           "present_i_w", "present_i_b", "present_n_w", "present_n_b",
                                           "out0_w", "out0_b", "out3_w", "out3_b", "out6_w", "out6_b")] +
                [("pmsg_fc1_w", C.c_void_p * 4), ("pmsg_fc1_b", C.c_void_p * 4),
                 ("pmsg_fc2_w", C.c_void_p * 4), ("pmsg_fc2_b", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("input_r_w", "input_r_b", "input_i_w", "input_i_b", "input_n_w",
                                           "input_n_b")])


class RecurrentDecoder(nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        self.num_vars = num_vars = params["num_vars"]
        input_size = params["input_size"]
        n_hid = params["decoder_hidden"]
        edge_types = params["num_edge_types"]
        self.skip_first_edge_type = params["skip_first"]
        out_size = params["input_size"]
        self.dropout_prob = params["decoder_dropout"]
        if self.dropout_prob != 0.0:
            raise ValueError("decoder_dropout must be 0.0 (inference path; the reference zeroes it in eval)")
        if n_hid % 32 != 0:
            raise ValueError("decoder_hidden must be a multiple of 32")
        self.edge_types = edge_types
        # creation order = the reference's (aether.py:517-581): identical RNG consumption under a seed
        self.msg_fc1 = nn.ModuleList([nn.Linear(2 * n_hid, n_hid) for _ in range(edge_types)])
        self.msg_fc2 = nn.ModuleList([nn.Linear(n_hid, n_hid) for _ in range(edge_types)])
        self.msg_out_shape = n_hid
        self.hidden_r = nn.Linear(n_hid, n_hid, bias=False)
        self.hidden_i = nn.Linear(n_hid, n_hid, bias=False)
        self.hidden_h = nn.Linear(n_hid, n_hid, bias=False)
        self.present_r = nn.Linear(n_hid, n_hid, bias=True)
        self.present_i = nn.Linear(n_hid, n_hid, bias=True)
        self.present_n = nn.Linear(n_hid, n_hid, bias=True)
        self.out_mlp = nn.Sequential(nn.Linear(n_hid, n_hid), nn.ReLU(), nn.Dropout(p=self.dropout_prob),
                                     nn.Linear(n_hid, n_hid), nn.ReLU(), nn.Dropout(p=self.dropout_prob),
                                     nn.Linear(n_hid, out_size))
        self.use_3d = params.get("use_3d", False)
        self.num_dims = 3 if self.use_3d else 2
        self.num_orientations = self.num_dims * (self.num_dims - 1) // 2
        self.num_relative_features = 4 * self.num_dims + self.num_orientations
        self.num_pos_features = self.num_dims + self.num_orientations
        nrf, D = self.num_relative_features, self.num_dims
        self.present_msg_fc1 = nn.ModuleList([nn.Linear(2 * nrf + input_size + D, n_hid) for _ in range(edge_types)])
        self.present_msg_fc2 = nn.ModuleList([nn.Linear(n_hid, n_hid) for _ in range(edge_types)])
        self.input_r = nn.Linear(input_size + nrf + D, n_hid, bias=True)
        self.input_i = nn.Linear(input_size + nrf + D, n_hid, bias=True)
        self.input_n = nn.Linear(input_size + nrf + D, n_hid, bias=True)
        self.localizer = AugmentedLocalizer(num_vars, use_3d=self.use_3d, pos_representation="polar")
        self.send_edges, self.recv_edges = torch.where(~torch.eye(num_vars, dtype=bool))
        self._cache = {}
        if device is not None:
            self.to(device)

    def get_initial_hidden(self, inputs):
        return torch.zeros(inputs.size(0), inputs.size(2), self.msg_out_shape, device=inputs.device)

    # -- plumbing ----------------------------------------------------------------------
    def _param_struct(self):
        ps = _DecoderParams()
        ptr = lambda t: t.data_ptr()
        for k in range(self.edge_types):
            ps.msg_fc1_w[k], ps.msg_fc1_b[k] = ptr(self.msg_fc1[k].weight), ptr(self.msg_fc1[k].bias)
            ps.msg_fc2_w[k], ps.msg_fc2_b[k] = ptr(self.msg_fc2[k].weight), ptr(self.msg_fc2[k].bias)
            ps.pmsg_fc1_w[k], ps.pmsg_fc1_b[k] = ptr(self.present_msg_fc1[k].weight), ptr(self.present_msg_fc1[k].bias)
            ps.pmsg_fc2_w[k], ps.pmsg_fc2_b[k] = ptr(self.present_msg_fc2[k].weight), ptr(self.present_msg_fc2[k].bias)
        ps.hidden_r_w, ps.hidden_i_w, ps.hidden_h_w = (ptr(self.hidden_r.weight), ptr(self.hidden_i.weight),
                                                       ptr(self.hidden_h.weight))
        for name in ("r", "i", "n"):
            lin_p, lin_i = getattr(self, "present_" + name), getattr(self, "input_" + name)
            setattr(ps, f"present_{name}_w", ptr(lin_p.weight)); setattr(ps, f"present_{name}_b", ptr(lin_p.bias))
            setattr(ps, f"input_{name}_w", ptr(lin_i.weight)); setattr(ps, f"input_{name}_b", ptr(lin_i.bias))
        for idx in (0, 3, 6):
            setattr(ps, f"out{idx}_w", ptr(self.out_mlp[idx].weight)); setattr(ps, f"out{idx}_b", ptr(self.out_mlp[idx].bias))
        return ps

    def _graph(self, B, N, device):
        """Global edge index of the batch and its grouping by receiver (stable): cached per shape."""
        key = (B, N, str(device))
        hit = self._cache.get(key)
        if hit is None:
            off = (torch.arange(B, device=device, dtype=torch.int64) * N).unsqueeze(1)
            send = (self.send_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            recv = (self.recv_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            order = torch.argsort(recv, stable=True).contiguous()
            counts = torch.bincount(recv, minlength=B * N)
            rowptr = torch.zeros(B * N + 1, dtype=torch.int64, device=device)
            rowptr[1:] = torch.cumsum(counts, 0)
            hit = self._cache[key] = (send, recv, order, rowptr)
        return hit

    @torch.no_grad()
    def forward(self, inputs, hidden, edges, predicted_field):
        """aether.py:590-654.  inputs [B, N, 2D], hidden [B, N, h], edges [B, N(N-1), K], predicted_field
        [B, N, D] -> (outputs [B, N, 2D], hidden [B, N, h])."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd RecurrentDecoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        B, N, F_in = inputs.shape
        D, h, K = self.num_dims, self.msg_out_shape, self.edge_types
        E1 = self.recv_edges.shape[0]
        if F_in != 2 * D or hidden.shape != (B, N, h) or edges.shape != (B, E1, K) or predicted_field.shape != (B, N, D):
            raise ValueError("decoder step: input shapes do not match the module")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        inputs_f, hidden_f, edges_f, field_f = f32(inputs), f32(hidden), f32(edges), f32(predicted_field)
        send, recv, order, rowptr = self._graph(B, N, inputs.device)
        need = lib.aether_s2s_decoder_workspace_bytes(D, h, B * N, B * E1)
        ws = self._cache.get("ws")
        if ws is None or ws.numel() < need or ws.device != inputs.device:
            ws = self._cache["ws"] = torch.empty(need, dtype=torch.uint8, device=inputs.device)
        outputs = torch.empty(B, N, 2 * D, dtype=torch.float32, device=inputs.device)
        hidden_out = torch.empty(B, N, h, dtype=torch.float32, device=inputs.device)
        ps = self._param_struct()
        st = lib.aether_s2s_decoder_step(C.byref(ps), D, h, K, 1 if self.skip_first_edge_type else 0, B * N, B * E1,
                                         inputs_f.data_ptr(), hidden_f.data_ptr(), edges_f.data_ptr(),
                                         field_f.data_ptr(), send.data_ptr(), recv.data_ptr(), order.data_ptr(),
                                         rowptr.data_ptr(), ws.data_ptr(), ws.numel(), outputs.data_ptr(),
                                         hidden_out.data_ptr(), torch.cuda.current_stream(inputs.device).cuda_stream)
        _lib.check(st, "aether_s2s_decoder_step")
        return outputs, hidden_out
