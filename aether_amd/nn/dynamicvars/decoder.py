"""MI355X drop-in for the decoder of the reference's variable-N model (second half):
``nn.dynamicvars.aether_dynamicvars.Decoder`` (aether_dynamicvars.py:703-870), the step the inD runner calls once
per time step for the objects present in the scene.

Same constructor dictionary, parameter names, creation order (so a seeded construction gives the reference's
initial weights) and ``forward(inputs, hidden, edges, node_masks, graph_info, predicted_field)``.  The present
objects are compacted on the device and the whole step runs in ``aether_dyn_decoder_step``; there is no CPU
fallback.  Two properties of the reference are kept (see oracle/dynamicvars_oracle.py): the edge features read the
un-compacted state with compacted indices, and a scene with a single present object raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from ..seq2seq.encoder import _AnisotropicEdgeFilter, filter_image


class _DynDecoderParams(C.Structure):
    _fields_ = ([(n, C.c_void_p * 4) for n in ("msg_fc1_w", "msg_fc1_b", "msg_fc2_w", "msg_fc2_b")] +
                [(n, C.c_void_p) for n in ("hidden_r_w", "hidden_i_w", "hidden_h_w", "input_r_w", "input_r_b", "input_i_w",
                                           "input_i_b", "input_n_w", "input_n_b", "present_r_w", "present_r_b",
                                           "present_i_w", "present_i_b", "present_n_w", "present_n_b", "out1_w", "out1_b",
                                           "out2_w", "out2_b", "out3_w", "out3_b")] +
                [(n, C.c_void_p * 4) for n in ("filt_w0", "filt_b0", "filt_w2", "filt_b2", "filt_image")])


class Decoder(nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        input_size = params["input_size"]
        n_hid = params["decoder_hidden"]
        edge_types = params["num_edge_types"]
        if params.get("use_3d", False) or input_size != 4:
            raise ValueError("the variable-N decoder is 2-D (the reference rotates back with Globalizer(num_dims=2))")
        if n_hid % 128 != 0:
            raise ValueError("decoder_hidden must be a multiple of 128")
        if not 1 <= edge_types <= 4:
            raise ValueError("num_edge_types must be 1..4")
        if params["decoder_dropout"] != 0.0:
            raise ValueError("decoder_dropout must be 0.0 (inference path)")
        self.num_dims, self.msg_out_shape, self.edge_types = 2, n_hid, edge_types
        self.skip_first_edge_type = params["skip_first"]
        # creation order of aether_dynamicvars.py:716-770
        self.msg_fc1 = nn.ModuleList([nn.Linear(2 * n_hid, n_hid) for _ in range(edge_types)])
        self.msg_fc2 = nn.ModuleList([nn.Linear(n_hid, n_hid) for _ in range(edge_types)])
        self.hidden_r = nn.Linear(n_hid, n_hid, bias=False)
        self.hidden_i = nn.Linear(n_hid, n_hid, bias=False)
        self.hidden_h = nn.Linear(n_hid, n_hid, bias=False)
        self.input_r = nn.Linear(input_size + 2, n_hid, bias=True)
        self.input_i = nn.Linear(input_size + 2, n_hid, bias=True)
        self.input_n = nn.Linear(input_size + 2, n_hid, bias=True)
        self.out_fc1 = nn.Linear(n_hid, n_hid)
        self.out_fc2 = nn.Linear(n_hid, n_hid)
        self.out_fc3 = nn.Linear(n_hid, input_size)
        self.pos_representation = params["pos_representation"]
        if self.pos_representation not in ("cart", "polar"):
            raise ValueError
        self.edge_filter = nn.ModuleList([_AnisotropicEdgeFilter(9 + input_size + 2, 3, n_hid, n_hid)
                                          for _ in range(edge_types)])
        self.present_r = nn.Linear(n_hid, n_hid, bias=True)
        self.present_i = nn.Linear(n_hid, n_hid, bias=True)
        self.present_n = nn.Linear(n_hid, n_hid, bias=True)
        self._ws = None
        if device is not None:
            self.to(device)

    def get_initial_hidden(self, inputs):
        return torch.zeros(inputs.size(0), inputs.size(2), self.msg_out_shape, device=inputs.device)

    def _apply(self, fn, *args, **kwargs):                 # .to() / .cuda(): the cached pointer struct is stale
        self.__dict__.pop("_plist", None)
        self.__dict__.pop("_pstruct", None)
        return super()._apply(fn, *args, **kwargs)

    def _param_struct(self):
        """Pointer struct of the parameters, rebuilt only when a parameter moved (walking the module tree and checking 40
        tensors took 0.3 ms per step of a 1.3 ms prediction step)."""
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = self.__dict__["_plist"] = list(self.parameters())
        key = tuple(p.data_ptr() for p in plist)
        hit = self.__dict__.get("_pstruct")
        if hit is None or hit[0] != key:
            ps = _DynDecoderParams()
            ptr = lambda t: t.data_ptr()
            for k in range(self.edge_types):
                ps.msg_fc1_w[k], ps.msg_fc1_b[k] = ptr(self.msg_fc1[k].weight), ptr(self.msg_fc1[k].bias)
                ps.msg_fc2_w[k], ps.msg_fc2_b[k] = ptr(self.msg_fc2[k].weight), ptr(self.msg_fc2[k].bias)
                f = self.edge_filter[k].edge_filter
                ps.filt_w0[k], ps.filt_b0[k], ps.filt_w2[k], ps.filt_b2[k] = ptr(f[0].weight), ptr(f[0].bias), ptr(f[2].weight), ptr(f[2].bias)
            ps.hidden_r_w, ps.hidden_i_w, ps.hidden_h_w = ptr(self.hidden_r.weight), ptr(self.hidden_i.weight), ptr(self.hidden_h.weight)
            for g in ("r", "i", "n"):
                for kind in ("input", "present"):
                    lin = getattr(self, f"{kind}_{g}")
                    setattr(ps, f"{kind}_{g}_w", ptr(lin.weight)); setattr(ps, f"{kind}_{g}_b", ptr(lin.bias))
            for j, lin in enumerate((self.out_fc1, self.out_fc2, self.out_fc3), 1):
                setattr(ps, f"out{j}_w", ptr(lin.weight)); setattr(ps, f"out{j}_b", ptr(lin.bias))
            for p in plist:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.AetherHipError("Decoder parameters must be contiguous fp32 CUDA tensors")
            filt_w2 = [self.edge_filter[k].edge_filter[2].weight for k in range(self.edge_types)]
            hit = self.__dict__["_pstruct"] = (key, ps, filt_w2)
        ps = hit[1]
        for k, w in enumerate(hit[2]):       # the filter images follow in-place updates by themselves (version counters)
            ps.filt_image[k] = filter_image(self.__dict__.setdefault("_img_cache", {}), f"filt{k}", w, 15, w.shape[1]).data_ptr()
        return ps

    @torch.no_grad()
    def forward(self, inputs, hidden, edges, node_masks, graph_info, predicted_field, n_present=None):
        """aether_dynamicvars.py:775-870.  inputs [1, Nmax, 4], hidden [1, Nmax, h], edges [1, E, K], node_masks
        [1, Nmax] (or [Nmax]), graph_info = (send_edges, recv_edges, edge2node_inds) in the numbering of the present
        objects, predicted_field [1, Nmax, 2] -> (pred_all [1, Nmax, 4], hidden [1, Nmax, h])."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Decoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if inputs.size(0) != 1:
            # the reference's models stop here ("Batching during forward not currently supported",
            # aether_dynamicvars.py:588-591); several scenes go through forward_batched (one launch sequence for all)
            if isinstance(edges, (list, tuple)) and isinstance(graph_info, (list, tuple)) and len(graph_info) == inputs.size(0):
                return self.forward_batched(inputs, hidden, edges, node_masks, graph_info, predicted_field)
            raise ValueError("Batching during forward not currently supported: pass per-scene lists of edges and "
                             "graph_info (forward_batched)")
        lib = _lib.load()
        dev = inputs.device
        h, K = self.msg_out_shape, self.edge_types
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        inputs, hidden, field = f32(inputs), f32(hidden), f32(predicted_field)
        if n_present is None:
            node_inds = node_masks.reshape(-1).to(dev).nonzero()[:, -1]
        else:                       # the caller knows the number of present objects on the host: no device round trip
            node_inds = torch.nonzero_static(node_masks.reshape(-1).to(dev), size=int(n_present))[:, 0]
        nv = int(node_inds.numel())
        if nv == 0:                                                                    # :841-843
            return torch.zeros_like(inputs), hidden
        if nv == 1:
            raise _lib.AetherHipError("a scene with one present object: the reference fails here as well "
                                      "(present_agg_msgs is never assigned, aether_dynamicvars.py:843-851)")
        send, recv, e2n = (t.to(device=dev, dtype=torch.int64).contiguous() for t in graph_info)
        E = send.numel()
        if recv.numel() != E or edges.shape != (1, E, K) or e2n.ndim != 2 or e2n.shape[0] != nv:
            raise ValueError("graph_info / edges do not match the present objects")
        cur_in, cur_h, cur_f = inputs[0, node_inds].contiguous(), hidden[0, node_inds].contiguous(), field[0, node_inds].contiguous()
        ext_full = torch.cat([inputs[0], field[0]], -1).contiguous()                  # indexed with compacted ids (:823)
        rowptr = torch.arange(nv + 1, device=dev, dtype=torch.int64) * e2n.shape[1]
        order = e2n.reshape(-1).contiguous()
        ew = f32(edges)[0]
        need = lib.aether_dyn_decoder_workspace_bytes(h, nv, E)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(nv, 4, dtype=torch.float32, device=dev)
        new_h = torch.empty(nv, h, dtype=torch.float32, device=dev)
        ps = self._param_struct()
        st = lib.aether_dyn_decoder_step(C.byref(ps), h, K, 1 if self.skip_first_edge_type else 0,
                                         1 if self.pos_representation == "polar" else 0, nv, E, cur_in.data_ptr(),
                                         cur_h.data_ptr(), ew.data_ptr(), cur_f.data_ptr(), ext_full.data_ptr(),
                                         send.data_ptr(), recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(),
                                         float(nv - 1), self._ws.data_ptr(), self._ws.numel(), out.data_ptr(),
                                         new_h.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_decoder_step")
        hidden = hidden.clone()
        hidden[0, node_inds] = new_h
        pred_all = torch.zeros_like(inputs)
        pred_all[0, node_inds] = out
        return pred_all, hidden

    @torch.no_grad()
    def forward_batched(self, inputs, hidden, edges, node_masks, graph_info, predicted_field):
        """``forward`` for B scenes in ONE call of the library (true batching; BASELINE config 4 has
        64 scenes).  inputs [B, Nmax, 4], hidden [B, Nmax, h], node_masks [B, Nmax], predicted_field [B, Nmax, 2];
        ``edges[b]`` [E_b, K] (or [1, E_b, K]) and ``graph_info[b] = (send, recv, edge2node_inds)`` exactly as the
        single-scene call takes them for scene b (indices in the numbering of that scene's present objects).  Returns
        (pred_all [B, Nmax, 4], hidden [B, Nmax, h]); every scene's rows equal its single-scene ``forward``.  Scenes
        without present objects yield zeros (:841-843); a scene with exactly one raises, as the reference does."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd Decoder runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        dev = inputs.device
        B, Nmax = inputs.size(0), inputs.size(1)
        h, K = self.msg_out_shape, self.edge_types
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        inputs, hidden, field = f32(inputs), f32(hidden), f32(predicted_field)
        masks = node_masks.reshape(B, Nmax).to(dev) != 0
        counts = masks.sum(1)                                            # present objects per scene
        counts_h = counts.tolist()
        if any(c == 1 for c in counts_h):
            raise _lib.AetherHipError("a scene with one present object: the reference fails here as well "
                                      "(present_agg_msgs is never assigned, aether_dynamicvars.py:843-851)")
        flat_idx = masks.reshape(-1).nonzero()[:, 0]                     # rows of the [B * Nmax] arrays, scene-major
        nv = int(flat_idx.numel())
        pred_all = torch.zeros_like(inputs)
        if nv == 0:
            return pred_all, hidden
        base = torch.cumsum(counts, 0) - counts                          # first concatenated row of every scene
        # The scenes' index lists are concatenated and shifted with a handful of launches, whatever B is (a loop of eight
        # small tensor operations per scene made the 64-scene step host-bound: 3.9 ms for 0.7 ms of kernels).
        present = [b for b in range(B) if counts_h[b] > 0]
        gi = [tuple(t.to(device=dev, dtype=torch.int64) for t in graph_info[b]) for b in present]
        ews = [f32(edges[b]).reshape(-1, K) for b in present]
        E_l = [g[0].numel() for g in gi]
        for b, g, ew_b, E_b in zip(present, gi, ews, E_l):
            if g[1].numel() != E_b or ew_b.shape[0] != E_b or g[2].ndim != 2 or g[2].shape[0] != counts_h[b]:
                raise ValueError(f"graph_info / edges of scene {b} do not match its present objects")
        E = sum(E_l)
        L_l = [g[2].numel() for g in gi]
        e_off = [sum(E_l[:j]) for j in range(len(present))]
        meta = torch.tensor([E_l, present, [b * Nmax for b in present], L_l, e_off, [g[2].shape[1] for g in gi],
                             [counts_h[b] for b in present]], dtype=torch.int64, device=dev)
        scene_e = torch.repeat_interleave(torch.arange(len(present), device=dev), meta[0], output_size=E)
        S_all, R_all = torch.cat([g[0] for g in gi]), torch.cat([g[1] for g in gi])
        off_c, off_u = base[meta[1]][scene_e], meta[2][scene_e]
        send, recv = (S_all + off_c).contiguous(), (R_all + off_c).contiguous()
        ssend, srecv = (S_all + off_u).contiguous(), (R_all + off_u).contiguous()        # un-compacted rows, compacted ids (:823)
        order = (torch.cat([g[2].reshape(-1) for g in gi])
                 + torch.repeat_interleave(meta[4], meta[3], output_size=sum(L_l))).contiguous()
        rowptr = torch.zeros(nv + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.repeat_interleave(meta[5], meta[6], output_size=nv), 0)
        ew = torch.cat(ews).contiguous()
        E = send.numel()
        div_node = torch.repeat_interleave((counts - 1).clamp(min=1).to(torch.float32), counts, output_size=nv).contiguous()
        x2, h2, f2 = inputs.reshape(B * Nmax, -1), hidden.reshape(B * Nmax, h), field.reshape(B * Nmax, -1)
        cur_in, cur_h, cur_f = x2[flat_idx].contiguous(), h2[flat_idx].contiguous(), f2[flat_idx].contiguous()
        ext_full = torch.cat([x2, f2], -1).contiguous()
        need = lib.aether_dyn_decoder_workspace_bytes(h, nv, E)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        out = torch.empty(nv, 4, dtype=torch.float32, device=dev)
        new_h = torch.empty(nv, h, dtype=torch.float32, device=dev)
        ps = self._param_struct()
        st = lib.aether_dyn_decoder_step_batched(
            C.byref(ps), h, K, 1 if self.skip_first_edge_type else 0, 1 if self.pos_representation == "polar" else 0,
            nv, E, cur_in.data_ptr(), cur_h.data_ptr(), ew.data_ptr(), cur_f.data_ptr(), ext_full.data_ptr(),
            ssend.data_ptr(), srecv.data_ptr(), send.data_ptr(), recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(),
            1.0, div_node.data_ptr(), self._ws.data_ptr(), self._ws.numel(), out.data_ptr(), new_h.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_decoder_step_batched")
        hidden = hidden.clone()
        hidden.reshape(B * Nmax, h)[flat_idx] = new_h
        pred_all.reshape(B * Nmax, -1)[flat_idx] = out
        return pred_all, hidden
