"""MI355X prior step of the seq2seq Aether's encoder (prior half).

Mirrors ``nn.seq2seq.aether.Encoder`` of the reference (aether.py:250-410) for the autoregressive
prediction path: same ``params`` dictionary, sub-modules and parameters created and initialised in the
reference's order (same seed -> same weights; ``state_dict`` keys / order match a reference checkpoint,
including the bidirectional-encoder tensors this path does not use), and
``single_step_forward(inputs, prior_state, predicted_field) -> (prior_logits, prior_state)``.
``forward(inputs, predicted_field)`` is the full-sequence encoder in evaluation mode (aether.py:350-382: prior and
posterior logits).  The computation runs in libaether_hip.so (``aether_s2s_prior_step``,
``aether_s2s_encoder_features`` / ``aether_s2s_lstm_step`` / ``aether_s2s_mlp_head``); there is no CPU fallback, and
training (gradients) is not part of this library.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from .localizer import AugmentedLocalizer


class _PriorParams(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in (
        "mlp3_w0", "mlp3_b0", "mlp3_w3", "mlp3_b3", "mlp3_bn_w", "mlp3_bn_b", "mlp3_bn_mean", "mlp3_bn_var",
        "mlp4_w0", "mlp4_b0", "mlp4_w3", "mlp4_b3", "mlp4_bn_w", "mlp4_bn_b", "mlp4_bn_mean", "mlp4_bn_var",
        "lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh")] +
        [("prior_w", C.c_void_p * 4), ("prior_b", C.c_void_p * 4)] +
        [(n, C.c_void_p) for n in ("res1_w", "res1_b", "filt_w0", "filt_b0", "filt_w2", "filt_b2", "filt_image")])


def gumbel_softmax_hard(logits, uniform, tau):
    """``gumbel_softmax(logits, tau, hard=True)`` of the reference (nn/utils/model_utils.py:58-118) with the
    uniform draw supplied: logits, uniform ``[..., K]`` on the device -> edge-type weights ``[..., K]``."""
    lib = _lib.load()
    K = logits.shape[-1]
    lg = logits.detach().to(torch.float32).reshape(-1, K).contiguous()
    un = uniform.detach().to(torch.float32).reshape(-1, K).contiguous()
    out = torch.empty_like(lg)
    _lib.check(lib.aether_s2s_gumbel_hard(lg.data_ptr(), un.data_ptr(), float(tau), K, lg.shape[0], out.data_ptr(),
                                          torch.cuda.current_stream(logits.device).cuda_stream), "aether_s2s_gumbel_hard")
    return out.view(logits.shape)


def filter_image(cache, name, w, n_features, hidden):
    """Two-piece fp16 image of a filter bank ``w`` [n_features * hidden, hidden] for the matrix cores
    (``aether_s2s_filter_prepare``), kept in ``cache[name]`` and rebuilt -- into the same buffer, which captured graphs
    point at -- whenever the weight tensor moved or was written to."""
    key = (w.data_ptr(), w._version, str(w.device))
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        lib = _lib.load()
        nbytes = lib.aether_s2s_filter_image_bytes(n_features, hidden)
        buf = hit[1] if hit is not None and hit[1].numel() == nbytes and hit[1].device == w.device else \
            torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _lib.check(lib.aether_s2s_filter_prepare(w.data_ptr(), n_features, hidden, buf.data_ptr(), nbytes,
                                                 torch.cuda.current_stream(w.device).cuda_stream),
                   "aether_s2s_filter_prepare")
        hit = cache[name] = (key, buf)
    return hit[1]


class _RefNRIMLP(nn.Module):
    """Parameter holder of ``RefNRIMLP`` (nn/utils/model_utils.py:15-43), same creation / init order."""

    def __init__(self, n_in, n_hid, n_out, do_prob=0.0):
        super().__init__()
        self.model = nn.Sequential(nn.Linear(n_in, n_hid), nn.ELU(inplace=True), nn.Dropout(do_prob),
                                   nn.Linear(n_hid, n_out), nn.ELU(inplace=True))
        self.bn = nn.BatchNorm1d(n_out)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight.data)
                m.bias.data.fill_(0.1)
            elif isinstance(m, nn.BatchNorm1d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class _AnisotropicEdgeFilter(nn.Module):
    """Parameter holder of ``AnisotropicEdgeFilter`` (nn/nn/anisotropic_filter.py:12-32)."""

    def __init__(self, in_size, pos_size, hidden_size, out_size):
        super().__init__()
        self.num_relative_features, self.out_size = in_size, out_size
        self.edge_filter = nn.Sequential(nn.Linear(pos_size, hidden_size), nn.ELU(),
                                         nn.Linear(hidden_size, in_size * out_size))
        nn.init.orthogonal_(self.edge_filter[0].weight, gain=nn.init.calculate_gain("relu"))
        nn.init.orthogonal_(self.edge_filter[2].weight)


def _mlp_out(n_in, n_hidden, n_out, num_layers):
    if num_layers == 1:
        return nn.Linear(n_in, n_out)
    layers = [nn.Linear(n_in, n_hidden), nn.ELU(inplace=True)]
    for _ in range(num_layers - 2):
        layers += [nn.Linear(n_hidden, n_hidden), nn.ELU(inplace=True)]
    layers.append(nn.Linear(n_hidden, n_out))
    return nn.Sequential(*layers)


class Encoder(nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        self.num_vars = num_vars = params["num_vars"]
        self.num_edges = params["num_edge_types"]
        if params["encoder_dropout"] != 0.0:
            raise ValueError("encoder_dropout must be 0.0 (inference path)")
        hidden_size = params["encoder_hidden"]
        rnn_hidden_size = params["encoder_rnn_hidden"] or hidden_size
        if params["encoder_rnn_type"] != "lstm":
            raise ValueError("encoder_rnn_type must be 'lstm'")
        if hidden_size % 32 != 0 or rnn_hidden_size % 32 != 0:
            raise ValueError("encoder_hidden and encoder_rnn_hidden must be multiples of 32")
        inp_size = params["input_size"]
        self.hidden_size, self.rnn_hidden_size = hidden_size, rnn_hidden_size
        # creation order = the reference's (aether.py:265-330)
        self.mlp3 = _RefNRIMLP(hidden_size, hidden_size, hidden_size)
        self.mlp4 = _RefNRIMLP(hidden_size * 3, hidden_size, hidden_size)
        self.forward_rnn = nn.LSTM(hidden_size, rnn_hidden_size, batch_first=True)
        self.reverse_rnn = nn.LSTM(hidden_size, rnn_hidden_size, batch_first=True)
        self.encoder_fc_out = _mlp_out(2 * rnn_hidden_size, params.get("encoder_mlp_hidden"), self.num_edges,
                                       params["encoder_mlp_num_layers"])
        self.prior_layers = params["prior_num_layers"]
        self.prior_fc_out = _mlp_out(rnn_hidden_size, params.get("prior_hidden_size"), self.num_edges,
                                     self.prior_layers)
        self.use_3d = params.get("use_3d", False)
        self.num_dims = D = 3 if self.use_3d else 2
        self.num_orientations = D * (D - 1) // 2
        self.num_relative_features = nrf = 4 * D + self.num_orientations
        self.num_pos_features = D + self.num_orientations
        self.res1 = nn.Linear(inp_size + nrf + D, hidden_size)
        self.edge_filter = _AnisotropicEdgeFilter(2 * nrf + inp_size + D, self.num_pos_features, hidden_size,
                                                  hidden_size)
        self.pos_representation = params.get("pos_representation", "cart")
        self.localizer = AugmentedLocalizer(num_vars, use_3d=self.use_3d, pos_representation=self.pos_representation)
        for m in self.modules():                                   # Encoder.init_weights, aether.py:332-336
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight.data)
                m.bias.data.fill_(0.1)
        self.send_edges, self.recv_edges = torch.where(~torch.eye(num_vars, dtype=bool))
        self._cache = {}
        if device is not None:
            self.to(device)

    # -- plumbing ----------------------------------------------------------------------
    def _param_struct(self, with_image=True):
        ps = _PriorParams()
        ptr = lambda t: t.data_ptr()
        for name in ("mlp3", "mlp4"):
            m = getattr(self, name)
            setattr(ps, name + "_w0", ptr(m.model[0].weight)); setattr(ps, name + "_b0", ptr(m.model[0].bias))
            setattr(ps, name + "_w3", ptr(m.model[3].weight)); setattr(ps, name + "_b3", ptr(m.model[3].bias))
            setattr(ps, name + "_bn_w", ptr(m.bn.weight)); setattr(ps, name + "_bn_b", ptr(m.bn.bias))
            setattr(ps, name + "_bn_mean", ptr(m.bn.running_mean)); setattr(ps, name + "_bn_var", ptr(m.bn.running_var))
        rnn = self.forward_rnn
        ps.lstm_w_ih, ps.lstm_w_hh = ptr(rnn.weight_ih_l0), ptr(rnn.weight_hh_l0)
        ps.lstm_b_ih, ps.lstm_b_hh = ptr(rnn.bias_ih_l0), ptr(rnn.bias_hh_l0)
        layers = [self.prior_fc_out] if isinstance(self.prior_fc_out, nn.Linear) else \
            [m for m in self.prior_fc_out if isinstance(m, nn.Linear)]
        for l, lin in enumerate(layers):
            ps.prior_w[l], ps.prior_b[l] = ptr(lin.weight), ptr(lin.bias)
        ps.res1_w, ps.res1_b = ptr(self.res1.weight), ptr(self.res1.bias)
        f = self.edge_filter.edge_filter
        ps.filt_w0, ps.filt_b0, ps.filt_w2, ps.filt_b2 = ptr(f[0].weight), ptr(f[0].bias), ptr(f[2].weight), ptr(f[2].bias)
        ps.filt_image = self._filter_image(f[2].weight).data_ptr() if with_image else None   # (the fused step's plan has its own)
        return ps, len(layers), (layers[0].out_features if len(layers) > 1 else 0)

    def _filter_image(self, w):
        return filter_image(self._cache, "filt_image", w, self.edge_filter.num_relative_features, self.hidden_size)

    def _graph(self, B, N, device):
        key = (B, N, str(device))
        hit = self._cache.get(key)
        if hit is None:
            off = (torch.arange(B, device=device, dtype=torch.int64) * N).unsqueeze(1)
            send = (self.send_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            recv = (self.recv_edges.to(device=device, dtype=torch.int64).unsqueeze(0) + off).reshape(-1).contiguous()
            order = torch.argsort(recv, stable=True).contiguous()
            rowptr = torch.zeros(B * N + 1, dtype=torch.int64, device=device)
            rowptr[1:] = torch.cumsum(torch.bincount(recv, minlength=B * N), 0)
            hit = self._cache[key] = (send, recv, order, rowptr)
        return hit

    def _head(self, lib, seq, x, out_size, dev):
        """prior_fc_out / encoder_fc_out (aether.py:286-300) on the rows of x: aether_s2s_mlp_head."""
        layers = [seq] if isinstance(seq, nn.Linear) else [m for m in seq if isinstance(m, nn.Linear)]
        n = len(layers)
        w = (C.c_void_p * n)(*[l.weight.data_ptr() for l in layers])
        b = (C.c_void_p * n)(*[l.bias.data_ptr() for l in layers])
        hid = layers[0].out_features if n > 1 else 0
        rows = x.shape[0]
        scratch = torch.empty(2 * rows * max(hid, 1), dtype=torch.float32, device=dev)
        out = torch.empty(rows, out_size, dtype=torch.float32, device=dev)
        st = lib.aether_s2s_mlp_head(w, b, n, x.shape[1], hid, out_size, rows, x.data_ptr(), scratch.data_ptr(), out.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_s2s_mlp_head")
        return out

    @torch.no_grad()
    def forward(self, inputs, predicted_field, max_edges_per_call=400_000):
        """The full-sequence encoder, aether.py:350-382 (evaluation: BatchNorm running statistics).  inputs
        [B, T, N, 2D], predicted_field [B, N, T, D] -> (prior_logits [B, T, E, K], posterior_logits [B, T, E, K],
        prior_state (h, c) each [B, E, rnn]: the forward LSTM's final state, in the layout ``single_step_forward``
        takes).  The per-time-step features are independent across time: T x B graphs go through
        ``aether_s2s_encoder_features`` in chunks of time steps; the two LSTMs step through time with
        ``aether_s2s_lstm_step``; the heads run once over all (time, edge) rows."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Encoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.training:
            raise _lib.AetherHipError("the encoder uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        dev = inputs.device
        B, T, N, _ = inputs.shape
        D, h, R, K = self.num_dims, self.hidden_size, self.rnn_hidden_size, self.num_edges
        E1 = self.recv_edges.shape[0]
        if inputs.shape[-1] != 2 * D or predicted_field.shape != (B, N, T, D):
            raise ValueError("encoder: input shapes do not match the module")
        x = inputs.detach().to(torch.float32).transpose(0, 1).contiguous()                # [T, B, N, 2D]
        f = predicted_field.detach().to(torch.float32).permute(2, 0, 1, 3).contiguous()   # [T, B, N, D]
        ps, _, _ = self._param_struct()
        stream = torch.cuda.current_stream(dev).cuda_stream
        feats = torch.empty(T, B * E1, h, dtype=torch.float32, device=dev)
        chunk = max(1, int(max_edges_per_call) // (B * E1))
        for t0 in range(0, T, chunk):
            t1 = min(T, t0 + chunk)
            G = (t1 - t0) * B                                                              # independent graphs
            send, recv, order, rowptr = self._graph(G, N, dev)
            need = lib.aether_s2s_prior_workspace_bytes(D, h, R, 0, G * N, G * E1)
            ws = self._cache.get("ws")
            if ws is None or ws.numel() < need or ws.device != dev:
                ws = self._cache["ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
            st = lib.aether_s2s_encoder_features(C.byref(ps), D, h, 1 if self.pos_representation == "polar" else 0, N,
                                                 G * N, G * E1, x[t0:t1].data_ptr(), f[t0:t1].data_ptr(), send.data_ptr(),
                                                 recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(), ws.data_ptr(),
                                                 ws.numel(), feats[t0:t1].data_ptr(), stream)
            _lib.check(st, "aether_s2s_encoder_features")
        rows = B * E1
        gates = torch.empty(rows, 4 * R, dtype=torch.float32, device=dev)

        def run(rnn, order_t):
            hs = torch.empty(T, rows, R, dtype=torch.float32, device=dev)
            hprev = torch.zeros(rows, R, dtype=torch.float32, device=dev)
            cprev = torch.zeros(rows, R, dtype=torch.float32, device=dev)
            cnext = torch.empty_like(cprev)
            for t in order_t:
                st = lib.aether_s2s_lstm_step(rnn.weight_ih_l0.data_ptr(), rnn.weight_hh_l0.data_ptr(),
                                              rnn.bias_ih_l0.data_ptr(), rnn.bias_hh_l0.data_ptr(), h, R, rows,
                                              feats[t].data_ptr(), hprev.data_ptr(), cprev.data_ptr(), gates.data_ptr(),
                                              hs[t].data_ptr(), cnext.data_ptr(), stream)
                _lib.check(st, "aether_s2s_lstm_step")
                hprev = hs[t]
                cprev, cnext = cnext, (cprev if cprev.data_ptr() != cnext.data_ptr() else torch.empty_like(cprev))
            return hs, hprev, cprev

        fwd, h_last, c_last = run(self.forward_rnn, range(T))
        rev, _, _ = run(self.reverse_rnn, range(T - 1, -1, -1))
        prior = self._head(lib, self.prior_fc_out, fwd.reshape(T * rows, R), K, dev)
        post = self._head(lib, self.encoder_fc_out, torch.cat([fwd, rev], -1).reshape(T * rows, 2 * R), K, dev)
        to_bt = lambda y: y.view(T, B, E1, K).transpose(0, 1).contiguous()
        return to_bt(prior), to_bt(post), (h_last.view(B, E1, R).clone(), c_last.view(B, E1, R).clone())

    @torch.no_grad()
    def single_step_forward(self, inputs, prior_state, predicted_field):
        """aether.py:384-410.  inputs [B, N, 2D], prior_state (h, c) each [B, E, rnn], predicted_field
        [B, N, D] -> (prior_logits [B, E, K], (h', c'))."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Encoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        B, N, _ = inputs.shape
        D, h, R, K = self.num_dims, self.hidden_size, self.rnn_hidden_size, self.num_edges
        E1 = self.recv_edges.shape[0]
        h0, c0 = prior_state
        if inputs.shape != (B, N, 2 * D) or predicted_field.shape != (B, N, D) or h0.shape != (B, E1, R) or c0.shape != h0.shape:
            raise ValueError("prior step: input shapes do not match the module")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        x, f, h0f, c0f = f32(inputs), f32(predicted_field), f32(h0), f32(c0)
        send, recv, order, rowptr = self._graph(B, N, inputs.device)
        ps, n_layers, prior_hidden = self._param_struct()
        need = lib.aether_s2s_prior_workspace_bytes(D, h, R, prior_hidden, B * N, B * E1)
        ws = self._cache.get("ws")
        if ws is None or ws.numel() < need or ws.device != inputs.device:
            ws = self._cache["ws"] = torch.empty(need, dtype=torch.uint8, device=inputs.device)
        logits = torch.empty(B, E1, K, dtype=torch.float32, device=inputs.device)
        h1, c1 = torch.empty_like(h0f), torch.empty_like(c0f)
        st = lib.aether_s2s_prior_step(C.byref(ps), D, h, R, n_layers, prior_hidden, K,
                                       1 if self.pos_representation == "polar" else 0, N, B * N, B * E1,
                                       x.data_ptr(), f.data_ptr(), h0f.data_ptr(), c0f.data_ptr(), send.data_ptr(),
                                       recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(), ws.data_ptr(), ws.numel(),
                                       logits.data_ptr(), h1.data_ptr(), c1.data_ptr(),
                                       torch.cuda.current_stream(inputs.device).cuda_stream)
        _lib.check(st, "aether_s2s_prior_step")
        return logits, (h1, c1)
