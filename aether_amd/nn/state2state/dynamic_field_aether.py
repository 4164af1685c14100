"""MI355X drop-in for the reference's ``nn.state2state.dynamic_field_aether.DynamicFieldAether``
(the model experiments/lorentz/main.py:148-149 builds for ``--model dynamic_field_aether``).

Same constructor, ``forward(h, x, edges, vel, edge_attr_orig, charges, num_nodes)`` and ``state_dict`` keys
(dynamic_field_aether.py:51-100).  The field comes from ``aether_dynamic_field`` (attention-pooled graph
summary + FiLM field net, :11-48), everything after it from the same kernels as ``Aether``
(``aether_forward_field``).  With gradients enabled the step goes through ``_DynStep``: ``aether_backward_field``
(GNN gradients + dL/dfield) and ``aether_dynamic_field_backward`` (FiLM field net, modulators, attention pooling),
so the training loop of experiments/lorentz/main.py:200-260 works unchanged.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ... import _lib
from .aether import GraphCache, _GNN, _kernel_width, _pad_blocks


class _AttentionalAggregation(nn.Module):
    """Parameter holder with torch_geometric's sub-module names (``gate_nn``, ``nn``)."""

    def __init__(self, gate_nn, nn_):
        super().__init__()
        self.gate_nn = gate_nn
        self.nn = nn_


class _GraphSummary(nn.Module):
    def __init__(self, input_size, hidden_size):                       # graph_pool.py:8-23
        super().__init__()
        self.summary_net = _AttentionalAggregation(
            nn.Sequential(nn.Linear(input_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, 1)),
            nn.Sequential(nn.Linear(input_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size)))


class _FiLM(nn.Module):
    def __init__(self, x_size, z_size, hidden_size):                   # film.py:48-55
        super().__init__()
        self.modulator = nn.Sequential(nn.Linear(z_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size),
                                       nn.SiLU(), nn.Linear(hidden_size, 2 * x_size))


class _FilmedNetwork(nn.Module):
    def __init__(self, x_size, z_size, hidden_size, out_size):         # film.py:12-24
        super().__init__()
        self.linear_1 = nn.Linear(x_size, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)
        self.linear_3 = nn.Linear(hidden_size, out_size)
        self.film_1 = _FiLM(hidden_size, z_size, hidden_size)
        self.film_2 = _FiLM(hidden_size, z_size, hidden_size)


class _LatentFieldNetwork(nn.Module):
    def __init__(self, num_dims, hidden_size, class_embedding_dim):    # dynamic_field_aether.py:12-26
        super().__init__()
        self.summary_net = _GraphSummary(2 * num_dims, hidden_size)
        self.wrapper = _FilmedNetwork(2 * num_dims + class_embedding_dim, hidden_size, hidden_size, num_dims)
        self.class_embedding = nn.Embedding(3, class_embedding_dim)


class _DynFieldParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "gate_w0", "gate_b0", "gate_w2", "gate_b2", "nn_w0", "nn_b0", "nn_w2", "nn_b2",
        "lin1_w", "lin1_b", "lin2_w", "lin2_b", "lin3_w", "lin3_b",
        "film1_w0", "film1_b0", "film1_w2", "film1_b2", "film1_w4", "film1_b4",
        "film2_w0", "film2_b0", "film2_w2", "film2_b2", "film2_w4", "film2_b4", "emb")]


class _DynStep(torch.autograd.Function):
    """aether_dynamic_field + aether_forward_field / their backward halves behind torch.autograd (parameters only
    get gradients: the runner detaches positions and edge attributes, experiments/lorentz/main.py:243-247)."""

    N_FIXED = 8

    @staticmethod
    def forward(ctx, module, x, vel, ea, charges, graph, n_edges, num_nodes, *params):
        out, field, ws = module._launch(x, vel, ea, charges, graph, n_edges, num_nodes, train=True)
        ctx.module, ctx.saved = module, (x, vel, charges, graph, ws, n_edges, num_nodes)
        if any(ctx.needs_input_grad[1:4]):       # x / vel / edge_attr: aether_backward_inputs recovers y = R^T (out - x)
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        module = ctx.module
        x, vel, charges, (graph, ginfo), ws, n_edges, num_nodes = ctx.saved
        D, n_nodes = module.num_dims, x.shape[0]
        ps, fps = module._structs(x.device)
        names, offsets, total = module._grad_layout()
        flat = torch.zeros(total, dtype=torch.float32, device=x.device)        # one buffer, one memset
        kshapes = module._kernel_shapes()
        # kernel-side views: a GNN tensor of a model whose hidden_size is not a kernel width has the padded shape
        kgrads = {n: flat[o:o + int(np.prod(kshapes[n]))].view(kshapes[n]) for n, o in zip(names, offsets)}
        gtensors = dict(kgrads)
        gtensors.update(module._dummy)                               # field_net.net.* slots: not written in this mode
        gs = _lib.params_struct(gtensors)
        gfs = module._dyn_struct(kgrads)
        g = grad_out.to(torch.float32).contiguous()
        grad_field = torch.empty(n_nodes, D, dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        kw = module._kw
        if kw == 64:
            st = lib.aether_backward_field(C.byref(ps), C.byref(gs), D, n_nodes, n_edges, x.data_ptr(), vel.data_ptr(),
                                           charges.data_ptr(), graph.data_ptr(), C.byref(ginfo), ws.data_ptr(),
                                           ws.numel(), g.data_ptr(), grad_field.data_ptr(), stream)
        else:
            st = lib.aether_backward_h(C.byref(ps), C.byref(gs), D, kw, n_nodes, n_edges, x.data_ptr(), vel.data_ptr(),
                                       charges.data_ptr(), graph.data_ptr(), C.byref(ginfo), ws.data_ptr(),
                                       ws.numel(), g.data_ptr(), grad_field.data_ptr(), stream)
        _lib.check(st, "aether_backward_field")
        n_graphs = n_nodes // num_nodes
        need = lib.aether_dynamic_field_backward_workspace_bytes(D, n_graphs)
        dws = torch.empty(need, dtype=torch.uint8, device=x.device)
        want_in = any(ctx.needs_input_grad[1:4])
        gz = torch.empty(n_nodes, 2 * D, dtype=torch.float32, device=x.device) if want_in else None
        _lib.check(lib.aether_dynamic_field_backward_inputs(C.byref(fps), C.byref(gfs), D, n_graphs, num_nodes, x.data_ptr(),
                                                            vel.data_ptr(), charges.data_ptr(), grad_field.data_ptr(),
                                                            dws.data_ptr(), dws.numel(),
                                                            gz.data_ptr() if gz is not None else None, stream),
                   "aether_dynamic_field_backward_inputs")
        module.last_grad_field = grad_field
        gx = gv = gea = None
        if want_in:
            # gradients w.r.t. the inputs (dynamic_field_aether.py:79-100 is differentiable in them): the GNN / frame part
            # from what aether_backward_field left in the workspace, the part through the latent field from gz
            (out_saved,) = ctx.saved_tensors
            gx, gv = torch.empty_like(x), torch.empty_like(x)
            if ctx.needs_input_grad[3]:
                gea = torch.empty(n_edges, 2, dtype=torch.float32, device=x.device)
            _lib.check(lib.aether_backward_inputs_h(C.byref(ps), D, kw, n_nodes, n_edges, x.data_ptr(), vel.data_ptr(),
                                                    charges.data_ptr(), graph.data_ptr(), C.byref(ginfo), ws.data_ptr(),
                                                    ws.numel(), out_saved.data_ptr(), g.data_ptr(), gx.data_ptr(), gv.data_ptr(),
                                                    gea.data_ptr() if gea is not None else None, gz.data_ptr(), stream),
                       "aether_backward_inputs")
            if not ctx.needs_input_grad[1]:
                gx = None
            if not ctx.needs_input_grad[2]:
                gv = None
        if module.dp_group is not None:            # one all-reduce of the flat gradient buffer (RCCL), then the mean
            import torch.distributed as dist
            dist.all_reduce(flat, group=module.dp_group)
            flat.div_(dist.get_world_size(module.dp_group))
        need_g = ctx.needs_input_grad[_DynStep.N_FIXED:]
        grads = module._narrow_grads(kgrads)
        return (None, gx, gv, gea, None, None, None, None) + tuple(grads[n] if k else None for n, k in zip(names, need_g))


class DynamicFieldAether(nn.Module):
    """Drop-in for nn/state2state/dynamic_field_aether.py:51-100."""

    def __init__(self, input_size, hidden_size, dropout_prob, num_dims, device="cuda"):
        super().__init__()
        if not (1 <= hidden_size <= 4096):
            raise ValueError("hidden_size must lie in [1, 4096] (experiments/lorentz/main.py:42-43)")
        if num_dims not in (2, 3) or input_size != 2 * num_dims:
            raise ValueError("num_dims must be 2 or 3 and input_size == 2*num_dims")
        if hidden_size == 3 * num_dims:
            raise ValueError("hidden_size == 3 * num_dims is not supported (the reference then builds layer_1 without its "
                             "res Linear, locs.py:214-218)")
        if not (0.0 <= float(dropout_prob) < 1.0):
            raise ValueError("dropout_prob must lie in [0, 1)")
        # (the runner passes 0.0, main.py:149; > 0: identity in eval(), the out MLP's two masks in train() -- as Aether)
        self.dropout_prob = float(dropout_prob)
        self.gnn = _GNN(input_size, hidden_size, dropout_prob, num_dims, additional_features=num_dims)
        self.num_dims = num_dims
        self.hidden_size = hidden_size
        # width the kernels run the GNN at: 64 (fused / streamed), or the next multiple of 64 above (csrc/wide.h); a model of
        # another width runs on zero-padded copies of its GNN parameters (exact: padded channels stay zero), as Aether does
        self._kw = _kernel_width(hidden_size)
        self._kshapes = None
        self._padded = None
        self.field_net = _LatentFieldNetwork(num_dims, 32, 16)
        self._graphs = GraphCache()
        self.flags = 0
        self.dp_group = None               # set by aether_amd.parallel.attach_data_parallel
        self._ws = None
        self._ws_key = None
        self._plist = None
        self._struct_cache = None
        self._dummy = None
        self.to(device)
        self.params = self.__str__()

    def __str__(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        print("Network Size", params)
        return str(params)

    def _apply(self, fn, *a, **k):
        self._glayout = None
        self._plist = None                # parameter storage may move (.to / .cuda / .float)
        self._struct_cache = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plist = None
        self._struct_cache = None
        return super().load_state_dict(*a, **k)

    def _structs(self, device):
        if self._plist is None:
            self._plist = [p for _, p in self.named_parameters()]
        key = (str(device),) + tuple([p.data_ptr() for p in self._plist])
        if self._struct_cache is not None and self._struct_cache[0] == key:
            return self._struct_cache[1], self._struct_cache[2]
        ps, fps = self._build_structs(device)
        self._struct_cache = (key, ps, fps)
        return ps, fps

    def _build_structs(self, device):
        sd = dict(self.named_parameters())
        D = self.num_dims
        # the built-in field net is bypassed; its slots of AetherParams point at readable scratch of the right size
        dummy = {"field_net.net.0.weight": (32, 2 * D + 16), "field_net.net.0.bias": (32,), "field_net.net.2.weight": (32, 32),
                 "field_net.net.2.bias": (32,), "field_net.net.4.weight": (D, 32), "field_net.net.4.bias": (D,)}
        if self._dummy is None or next(iter(self._dummy.values())).device != device:
            self._dummy = {k: torch.zeros(*shape, device=device) for k, shape in dummy.items()}
        tensors = {k: v for k, v in sd.items()}
        if self.hidden_size != self._kw:
            tensors.update(self._padded_gnn(device))
        tensors.update(self._dummy)
        ps = _lib.params_struct(tensors)
        return ps, self._dyn_struct(sd)

    def _kernel_shapes(self):
        """{parameter name: shape of the tensor the kernels see} -- the GNN's at kernel width."""
        if self._kshapes is None:
            shapes = {n: tuple(p.shape) for n, p in self.named_parameters()}
            if self.hidden_size != self._kw:
                with torch.device("meta"):
                    wide = _GNN(2 * self.num_dims, self._kw, 0.0, self.num_dims, additional_features=self.num_dims)
                shapes.update({"gnn." + n: tuple(p.shape) for n, p in wide.named_parameters()})
            self._kshapes = shapes
        return self._kshapes

    def _padded_gnn(self, device):
        """Zero-padded kernel-width copies of the GNN parameters, refreshed from the parameters (every call: an optimizer
        step may lie between two calls, and inside a captured step the copies have to be part of the graph)."""
        ks = self._kernel_shapes()
        if self._padded is None or next(iter(self._padded.values())).device != device:
            self._padded = {n: torch.zeros(ks[n], dtype=torch.float32, device=device)
                            for n, _ in self.named_parameters() if n.startswith("gnn.")}
            self._struct_cache = None
        with torch.no_grad():
            for n, p in self.named_parameters():
                if n.startswith("gnn."):
                    for ss, ds in _pad_blocks(n, p.shape, self.hidden_size, self._kw):
                        self._padded[n][ds].copy_(p[ss])
        return self._padded

    def _narrow_grads(self, kgrads):
        """Kernel-side gradients cut back to the parameters' shapes."""
        if self.hidden_size == self._kw:
            return kgrads
        out = {}
        for n, p in self.named_parameters():
            g = kgrads[n]
            if n.startswith("gnn."):
                d = torch.empty_like(p)
                for ss, ds in _pad_blocks(n, p.shape, self.hidden_size, self._kw):
                    d[ss] = g[ds]
                g = d
            out[n] = g
        return out

    _DYN_NAMES = ["summary_net.summary_net.gate_nn.0", "summary_net.summary_net.gate_nn.2", "summary_net.summary_net.nn.0",
                  "summary_net.summary_net.nn.2", "wrapper.linear_1", "wrapper.linear_2", "wrapper.linear_3",
                  "wrapper.film_1.modulator.0", "wrapper.film_1.modulator.2", "wrapper.film_1.modulator.4",
                  "wrapper.film_2.modulator.0", "wrapper.film_2.modulator.2", "wrapper.film_2.modulator.4"]

    def _grad_layout(self):
        """(parameter names, offsets into one flat gradient buffer (64-float aligned), total floats)."""
        if getattr(self, "_glayout", None) is None:
            names, offsets, total = [], [], 0
            ks = self._kernel_shapes()
            for n, p in self.named_parameters():
                names.append(n)
                offsets.append(total)
                total += (int(np.prod(ks[n])) + 63) // 64 * 64
            self._glayout = (names, offsets, total)
        return self._glayout

    def _dyn_struct(self, tensors):
        """AetherDynFieldParams from {parameter name: tensor} (the parameters themselves or their gradients)."""
        ptrs = []
        for n in self._DYN_NAMES:
            ptrs += [tensors["field_net." + n + ".weight"].data_ptr(), tensors["field_net." + n + ".bias"].data_ptr()]
        ptrs.append(tensors["field_net.class_embedding.weight"].data_ptr())
        return _DynFieldParams(*ptrs)

    def _launch(self, x, vel, ea, charges, graph, n_edges, num_nodes, train):
        lib = _lib.load()
        graph, ginfo = graph
        n_nodes, D, E = x.shape[0], self.num_dims, n_edges
        if self.hidden_size != self._kw:
            self._padded_gnn(x.device)      # refresh the kernel-width copies of the GNN parameters
        kw = self._kw
        ps, fps = self._structs(x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        field = torch.empty(n_nodes, D, dtype=torch.float32, device=x.device)
        _lib.check(lib.aether_dynamic_field(C.byref(fps), D, n_nodes // int(num_nodes), int(num_nodes), x.data_ptr(),
                                            vel.data_ptr(), charges.data_ptr(), field.data_ptr(), stream),
                   "aether_dynamic_field")
        ws_bytes = lib.aether_workspace_bytes_h(n_nodes, E, D, kw, 1 if train else 0)
        flags = self.flags & ~_lib.FLAG_KEEP_INTERMEDIATES
        if kw != 64:
            flags &= ~(_lib.FLAG_FORCE_FUSED | _lib.FLAG_FORCE_STREAMED)
        ws_key = None
        if train:                           # the backward reads this forward's intermediates: one workspace per call
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            flags |= _lib.FLAG_KEEP_INTERMEDIATES | (0 if self.flags & _lib.FLAG_KEEP_INTERMEDIATES else _lib.FLAG_BACKWARD_ONLY)
            if self.dropout_prob > 0.0 and self.training:
                # nn.Dropout after the two SiLUs of the out MLP (locs.py:163,166): scale masks into the training workspace
                off = lib.aether_dropout_mask_offset_h(n_nodes, E, D, kw)
                masks = ws[off:off + 2 * n_nodes * kw * 4].view(torch.float32).view(2, n_nodes, kw)
                given = self.__dict__.get("_dropout_masks")          # tests: explicit masks [2, n_nodes, width]
                if given is not None and given.shape[-1] != kw:      # a narrow model's masks: padded channels are zero anyway
                    given = torch.nn.functional.pad(given, (0, kw - given.shape[-1]), value=1.0)
                if given is not None:
                    masks.copy_(given.to(device=x.device, dtype=torch.float32))
                else:
                    masks.bernoulli_(1.0 - self.dropout_prob).mul_(1.0 / (1.0 - self.dropout_prob))
                flags |= _lib.FLAG_DROPOUT
        else:
            if self._ws is None or self._ws.numel() < ws_bytes or self._ws.device != x.device:
                self._ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
            ws = self._ws
            fused = ginfo.n_groups > 0 and E > 0 and not (flags & _lib.FLAG_FORCE_STREAMED)
            ws_key = (ws.data_ptr(), n_nodes, E, D, graph.data_ptr()) if fused else None
            if ws_key is not None and self._ws_key == ws_key:
                flags |= _lib.FLAG_WORKSPACE_REUSED
        self._ws_key = None
        out = torch.empty_like(x)
        if kw == 64:
            st = lib.aether_forward_field(C.byref(ps), D, n_nodes, E, x.data_ptr(), vel.data_ptr(), charges.data_ptr(),
                                          field.data_ptr(), ea.data_ptr(), graph.data_ptr(), C.byref(ginfo),
                                          ws.data_ptr(), ws.numel(), out.data_ptr(), flags, stream)
        else:       # hidden_size > 64: csrc/wide.h with the external field
            st = lib.aether_forward_h(C.byref(ps), D, kw, n_nodes, E, x.data_ptr(), vel.data_ptr(), charges.data_ptr(),
                                      field.data_ptr(), ea.data_ptr(), graph.data_ptr(), C.byref(ginfo),
                                      ws.data_ptr(), ws.numel(), out.data_ptr(), flags, stream)
        _lib.check(st, "aether_forward_field")
        self._ws_key = ws_key
        self.last_field = field
        return out, field, ws

    @torch.no_grad()
    def rollout(self, x, vel, edges, charges, steps, dt=1.0, num_nodes=None):
        """``steps`` autoregressive steps on the device (``aether_rollout_dynamic_field``), the protocol of
        ``aether_amd.rollout``: x_{t+1} = self(x_t, v_t), v_{t+1} = (x_{t+1} - x_t) / dt, edge attributes rebuilt in the
        kernels, the latent field recomputed from the current state every step.  -> [steps, n_nodes, D]."""
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd.DynamicFieldAether runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if num_nodes is None:
            raise ValueError("num_nodes (objects per graph) is required, as in forward")
        lib = _lib.load()
        send, recv = edges
        n_nodes, D = x.shape
        E = send.numel()
        if D != self.num_dims or vel.shape != x.shape or charges.numel() != n_nodes or n_nodes % int(num_nodes) != 0:
            raise ValueError("x/vel must be [B * num_nodes, num_dims], charges [B * num_nodes, 1]")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        x, vel, charges = f32(x), f32(vel), f32(charges)
        graph, ginfo = self._graphs.get(send.contiguous(), recv.contiguous(), n_nodes)
        if self.dropout_prob > 0.0 and self.training:
            raise RuntimeError("DynamicFieldAether.rollout is an inference path (no dropout masks): call .eval() first")
        if self.hidden_size != self._kw:
            self._padded_gnn(x.device)
        ps, fps = self._structs(x.device)
        ws_bytes = lib.aether_workspace_bytes_h(n_nodes, E, D, self._kw, 0)
        if self._ws is None or self._ws.numel() < ws_bytes or self._ws.device != x.device:
            self._ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        self._ws_key = None
        traj = torch.empty(int(steps), n_nodes, D, dtype=torch.float32, device=x.device)
        if int(steps) <= 0:
            return traj
        field = torch.empty(n_nodes, D, dtype=torch.float32, device=x.device)
        flags = self.flags & ~_lib.FLAG_KEEP_INTERMEDIATES
        if self._kw != 64:
            flags &= ~(_lib.FLAG_FORCE_FUSED | _lib.FLAG_FORCE_STREAMED)
        st = lib.aether_rollout_dynamic_field_h(C.byref(ps), C.byref(fps), D, self._kw, n_nodes, E, int(num_nodes), x.data_ptr(),
                                                vel.data_ptr(), charges.data_ptr(), graph.data_ptr(), C.byref(ginfo),
                                                self._ws.data_ptr(), self._ws.numel(), field.data_ptr(), traj.data_ptr(),
                                                int(steps), float(dt), flags,
                                                torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_rollout_dynamic_field")
        return traj

    def forward(self, h, x, edges, vel, edge_attr_orig, charges, num_nodes):
        """``h`` is ignored, as in the reference (dynamic_field_aether.py:79-100)."""
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd.DynamicFieldAether runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        send, recv = edges
        if send.dtype != torch.int64 or recv.dtype != torch.int64:
            raise TypeError("edges must be int64 (torch.LongTensor), as in the reference")
        n_nodes, D = x.shape
        E = send.numel()
        if D != self.num_dims or vel.shape != x.shape or n_nodes % int(num_nodes) != 0:
            raise ValueError("x/vel must be [B * num_nodes, num_dims]")
        if recv.numel() != E or edge_attr_orig.shape != (E, 2) or charges.numel() != n_nodes:
            raise ValueError("edge index / edge_attr / charges shapes do not match")
        # differentiable in x / vel / edge_attr_orig, as the reference's forward (dynamic_field_aether.py:79-100)
        wants_in = torch.is_grad_enabled() and (x.requires_grad or vel.requires_grad or edge_attr_orig.requires_grad)
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        f32g = (lambda t: t.to(torch.float32).contiguous() if t.requires_grad else f32(t)) if wants_in else f32
        x, vel, ea, charges = f32g(x), f32g(vel), f32g(edge_attr_orig), f32(charges)
        graph = self._graphs.get(send.contiguous(), recv.contiguous(), n_nodes)
        if wants_in or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return _DynStep.apply(self, x, vel, ea, charges, graph, E, int(num_nodes), *self.parameters())
        with torch.no_grad():
            # (a train()-mode forward applies dropout even without autograd, as nn.Dropout does)
            drops = self.dropout_prob > 0.0 and self.training
            return self._launch(x, vel, ea, charges, graph, E, int(num_nodes), train=drops)[0]
