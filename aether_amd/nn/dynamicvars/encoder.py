"""MI355X prior step of the variable-N model's encoder:
``nn.dynamicvars.aether_dynamicvars.Encoder`` (aether_dynamicvars.py:381-699) for the prediction path.

Same ``params`` dictionary, sub-modules created and initialised in the reference's order (same seed -> same
weights; ``state_dict`` keys match a reference checkpoint, including ``mlp2`` / ``reverse_rnn`` / ``encoder_fc_out``,
which this path does not use) and ``single_step_forward(inputs, node_masks, node_inds, all_graph_info,
forward_state, predicted_field) -> (prior_logits, forward_state)``.  The kNN graph of the present objects
(``knn_edges``, :559-586), the feature transform (:505-557) and the LSTM / prior step run in libaether_hip.so
(``aether_knn_edges``, ``aether_dyn_prior_step``); the state slots per fully connected pair (:684-694) are gathered
and scattered here with torch indexing.  No CPU fallback; the full-sequence ``forward`` (posterior) is not part of
this path.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from ...knn import csr_by_receiver, knn_edges
from ..seq2seq.encoder import _AnisotropicEdgeFilter, _RefNRIMLP, _mlp_out, filter_image


class _DynPriorParams(C.Structure):
    _fields_ = ([(f"{m}_{t}", C.c_void_p) for m in ("mlp1", "mlp3", "mlp4")
                 for t in ("w0", "b0", "w3", "b3", "bn_w", "bn_b", "bn_mean", "bn_var")] +
                [(n, C.c_void_p) for n in ("lstm_w_ih", "lstm_w_hh", "lstm_b_ih", "lstm_b_hh")] +
                [("prior_w", C.c_void_p * 4), ("prior_b", C.c_void_p * 4)] +
                [(n, C.c_void_p) for n in ("filt_w0", "filt_b0", "filt_w2", "filt_b2", "filt_image")])


class Encoder(nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        self.num_edges = params["num_edge_types"]
        if params["no_encoder_bn"]:
            raise ValueError("no_encoder_bn=True is not supported by the drop-in's parameter holders")
        if params["encoder_dropout"] != 0.0:
            raise ValueError("encoder_dropout must be 0.0 (inference path)")
        hidden_size = params["encoder_hidden"]
        self.rnn_hidden_size = rnn_hidden_size = params["encoder_rnn_hidden"] or hidden_size
        if params["encoder_rnn_type"] != "lstm":
            raise ValueError("encoder_rnn_type must be 'lstm'")
        if params.get("use_3d", False) or params["input_size"] != 4:
            raise ValueError("the variable-N encoder is 2-D")
        if hidden_size % 128 != 0 or rnn_hidden_size % 16 != 0:
            raise ValueError("encoder_hidden must be a multiple of 128 and encoder_rnn_hidden of 16")
        if params["encoder_normalize_mode"] != "normalize_all":
            raise NotImplementedError                                               # as the reference (:499-500)
        self.hidden_size = hidden_size
        inp_size = params["input_size"]
        # creation order of aether_dynamicvars.py:399-461
        self.mlp1 = _RefNRIMLP(inp_size + 2, hidden_size, hidden_size)
        self.mlp2 = _RefNRIMLP(hidden_size * 2, hidden_size, hidden_size)
        self.mlp3 = _RefNRIMLP(hidden_size, hidden_size, hidden_size)
        self.mlp4 = _RefNRIMLP(hidden_size * 3, hidden_size, hidden_size)
        self.train_data_len = params.get("train_data_len", -1)
        self.forward_rnn = nn.LSTM(hidden_size, rnn_hidden_size, batch_first=True)
        self.reverse_rnn = nn.LSTM(hidden_size, rnn_hidden_size, batch_first=True)
        self.encoder_fc_out = _mlp_out(2 * rnn_hidden_size, params.get("encoder_mlp_hidden"), self.num_edges,
                                       params["encoder_mlp_num_layers"])
        self.prior_layers = params["prior_num_layers"]
        self.prior_fc_out = _mlp_out(rnn_hidden_size, params.get("prior_hidden_size"), self.num_edges, self.prior_layers)
        self.pos_representation = params["pos_representation"]
        if self.pos_representation not in ("cart", "polar"):
            raise ValueError
        self.edge_filter = _AnisotropicEdgeFilter(9 + inp_size + 2, 3, hidden_size, hidden_size)
        for m in self.modules():                                   # Encoder.init_weights, :463-467
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight.data)
                m.bias.data.fill_(0.1)
        self._ws = None
        if device is not None:
            self.to(device)

    def get_initial_hidden(self, inputs):
        batch = inputs.size(0) * inputs.size(2) * (inputs.size(2) - 1)               # :481-486
        return (torch.zeros(1, batch, self.rnn_hidden_size, device=inputs.device),
                torch.zeros(1, batch, self.rnn_hidden_size, device=inputs.device))

    def _apply(self, fn, *args, **kwargs):                 # .to() / .cuda(): the cached pointer struct is stale
        self.__dict__.pop("_plist", None)
        self.__dict__.pop("_pstruct", None)
        return super()._apply(fn, *args, **kwargs)

    def _param_struct(self):
        """Pointer struct of parameters and BatchNorm buffers, rebuilt only when one of them moved."""
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters()) + list(self.buffers())
        key = tuple(p.data_ptr() for p in plist)
        hit = self.__dict__.get("_pstruct")
        if hit is None or hit[0] != key:
            ps = _DynPriorParams()
            ptr = lambda t: t.data_ptr()
            for name in ("mlp1", "mlp3", "mlp4"):
                m = getattr(self, name)
                for tag, t in (("w0", m.model[0].weight), ("b0", m.model[0].bias), ("w3", m.model[3].weight), ("b3", m.model[3].bias),
                               ("bn_w", m.bn.weight), ("bn_b", m.bn.bias), ("bn_mean", m.bn.running_mean), ("bn_var", m.bn.running_var)):
                    setattr(ps, f"{name}_{tag}", ptr(t))
            rnn = self.forward_rnn
            ps.lstm_w_ih, ps.lstm_w_hh, ps.lstm_b_ih, ps.lstm_b_hh = ptr(rnn.weight_ih_l0), ptr(rnn.weight_hh_l0), ptr(rnn.bias_ih_l0), ptr(rnn.bias_hh_l0)
            layers = [self.prior_fc_out] if isinstance(self.prior_fc_out, nn.Linear) else \
                [m for m in self.prior_fc_out if isinstance(m, nn.Linear)]
            for l, lin in enumerate(layers):
                ps.prior_w[l], ps.prior_b[l] = ptr(lin.weight), ptr(lin.bias)
            f = self.edge_filter.edge_filter
            ps.filt_w0, ps.filt_b0, ps.filt_w2, ps.filt_b2 = ptr(f[0].weight), ptr(f[0].bias), ptr(f[2].weight), ptr(f[2].bias)
            hit = self.__dict__["_pstruct"] = (key, ps, f[2].weight, len(layers), (layers[0].out_features if len(layers) > 1 else 0))
        ps, w = hit[1], hit[2]
        # the filter image follows in-place updates by itself (version counter)
        ps.filt_image = filter_image(self.__dict__.setdefault("_img_cache", {}), "filt", w, 15, w.shape[1]).data_ptr()
        return ps, hit[3], hit[4]

    @torch.no_grad()
    def single_step_forward(self, inputs, node_masks, node_inds, all_graph_info, forward_state, predicted_field,
                            n_present=None):
        """aether_dynamicvars.py:672-699.  inputs [1, Nmax, 4], node_masks [1, Nmax], node_inds: the present objects,
        all_graph_info = (send, recv, ...) in their numbering, forward_state (h, c) each [1, Nmax (Nmax - 1), R],
        predicted_field [1, Nmax, 2] -> (prior_logits [1, E, K], forward_state)."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Encoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        dev = inputs.device
        if len(node_inds) <= 1:                                                      # :696-697
            return torch.empty(1, 0, self.num_edges, device=dev), forward_state
        Nmax, h, R, K = inputs.size(1), self.hidden_size, self.rnn_hidden_size, self.num_edges
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        x, field = f32(inputs), f32(predicted_field)
        mask = node_masks.reshape(-1).to(dev)
        # the encoder's own kNN graph of the present objects, from the current inputs (:528)
        if n_present is None:
            send, recv, _ = knn_edges(x, mask.reshape(1, -1).to(torch.float32))
            keep = mask.bool()
            cur_in, cur_f = x[0, keep].contiguous(), field[0, keep].contiguous()
        else:
            # predict_future's own calls: the number of present objects is known on the host (len(node_inds)), so the rows
            # of the mask's non-zero entries can be gathered without a device round trip
            send, recv, _ = knn_edges(x, mask.reshape(1, -1).to(torch.float32), n_present=n_present)
            idx = torch.nonzero_static(mask, size=int(n_present))[:, 0]
            cur_in, cur_f = x[0].index_select(0, idx), field[0].index_select(0, idx)
        n, E = cur_in.shape[0], send.numel()
        order, rowptr = csr_by_receiver(recv, n)
        # LSTM state rows of the caller's edges: one slot per fully connected pair (:680-686)
        gsend, grecv = all_graph_info[0].to(dev), all_graph_info[1].to(dev)
        node_inds = node_inds.to(dev)
        gs, gr = node_inds[gsend], node_inds[grecv]
        slot = gs * (Nmax - 1) + gr - (gr >= gs).long()
        if slot.numel() != E:
            raise ValueError("graph_info and the encoder's kNN graph list a different number of edges")
        h0, c0 = f32(forward_state[0])[0, slot].contiguous(), f32(forward_state[1])[0, slot].contiguous()
        ps, n_layers, prior_hidden = self._param_struct()
        need = lib.aether_dyn_prior_workspace_bytes(h, R, prior_hidden, n, E)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        logits = torch.empty(E, K, dtype=torch.float32, device=dev)
        h1, c1 = torch.empty_like(h0), torch.empty_like(c0)
        st = lib.aether_dyn_prior_step(C.byref(ps), h, R, n_layers, prior_hidden, K,
                                       1 if self.pos_representation == "polar" else 0, n, E, cur_in.data_ptr(),
                                       cur_f.data_ptr(), h0.data_ptr(), c0.data_ptr(), send.data_ptr(), recv.data_ptr(),
                                       order.data_ptr(), rowptr.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                       logits.data_ptr(), h1.data_ptr(), c1.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_prior_step")
        new_h, new_c = forward_state[0].clone(), forward_state[1].clone()
        new_h[0, slot], new_c[0, slot] = h1, c1
        return logits.unsqueeze(0), (new_h, new_c)

    @torch.no_grad()
    def single_step_forward_batched(self, inputs, node_masks, node_inds, all_graph_info, forward_state, predicted_field):
        """``single_step_forward`` for B scenes in one call of the library (the reference refuses batch > 1,
        aether_dynamicvars.py:588-591).  inputs [B, Nmax, 4], node_masks [B, Nmax], predicted_field [B, Nmax, 2];
        ``node_inds[b]`` / ``all_graph_info[b]`` as the single-scene call takes them for scene b; forward_state (h, c)
        each [1, B * Nmax * (Nmax - 1), R] (``get_initial_hidden`` of the batched inputs).  Returns (list of
        prior_logits [1, E_b, K] per scene, forward_state).  Scenes with fewer than two present objects contribute
        no edges (:696-697)."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Encoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        dev = inputs.device
        B, Nmax, h, R, K = inputs.size(0), inputs.size(1), self.hidden_size, self.rnn_hidden_size, self.num_edges
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        x, field = f32(inputs), f32(predicted_field)
        mask = node_masks.reshape(B, Nmax).to(dev) != 0
        counts = mask.sum(1)
        counts_h = counts.tolist()
        # scenes with a single present object have no edges and are left out of the graph (as the single-scene path does)
        use = mask & (counts >= 2).unsqueeze(1)
        empty = [torch.empty(1, 0, K, device=dev) for _ in range(B)]
        if not bool(use.any()):
            return empty, forward_state
        send, recv, _ = knn_edges(x, use.to(torch.float32))                # all scenes, concatenated compacted numbering
        cur_in, cur_f = x[use].contiguous(), field[use].contiguous()
        n, E = cur_in.shape[0], send.numel()
        order, rowptr = csr_by_receiver(recv, n)
        # LSTM state slots of all scenes' edges with a handful of launches, whatever B is (:680-686 per scene)
        used = [b for b in range(B) if counts_h[b] >= 2]
        gs_l = [all_graph_info[b][0].to(device=dev, dtype=torch.int64) for b in used]
        gr_l = [all_graph_info[b][1].to(device=dev, dtype=torch.int64) for b in used]
        ni_l = [node_inds[b].to(device=dev, dtype=torch.int64) for b in used]
        E_l = [g.numel() for g in gs_l]
        per_scene = [0] * B
        for b, E_b in zip(used, E_l):
            per_scene[b] = E_b
        n_l = [t.numel() for t in ni_l]
        meta = torch.tensor([E_l, [sum(n_l[:j]) for j in range(len(used))], [b * Nmax * (Nmax - 1) for b in used]],
                            dtype=torch.int64, device=dev)
        scene_e = torch.repeat_interleave(torch.arange(len(used), device=dev), meta[0], output_size=sum(E_l))
        ni_all = torch.cat(ni_l)
        ni_off = meta[1][scene_e]
        gs, gr = ni_all[torch.cat(gs_l) + ni_off], ni_all[torch.cat(gr_l) + ni_off]
        slot = meta[2][scene_e] + gs * (Nmax - 1) + gr - (gr >= gs).long()
        if slot.numel() != E:
            raise ValueError("graph_info and the encoder's kNN graphs list a different number of edges")
        h0, c0 = f32(forward_state[0])[0, slot].contiguous(), f32(forward_state[1])[0, slot].contiguous()
        ps, n_layers, prior_hidden = self._param_struct()
        need = lib.aether_dyn_prior_workspace_bytes(h, R, prior_hidden, n, E)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        logits = torch.empty(E, K, dtype=torch.float32, device=dev)
        h1, c1 = torch.empty_like(h0), torch.empty_like(c0)
        st = lib.aether_dyn_prior_step(C.byref(ps), h, R, n_layers, prior_hidden, K,
                                       1 if self.pos_representation == "polar" else 0, n, E, cur_in.data_ptr(),
                                       cur_f.data_ptr(), h0.data_ptr(), c0.data_ptr(), send.data_ptr(), recv.data_ptr(),
                                       order.data_ptr(), rowptr.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                       logits.data_ptr(), h1.data_ptr(), c1.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_prior_step")
        new_h, new_c = forward_state[0].clone(), forward_state[1].clone()
        new_h[0, slot], new_c[0, slot] = h1, c1
        parts = logits.split(per_scene)
        out = [parts[b].unsqueeze(0) if per_scene[b] else empty[b] for b in range(B)]
        return out, (new_h, new_c)
