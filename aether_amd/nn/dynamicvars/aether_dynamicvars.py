"""MI355X prediction path of the reference's variable-N model:
``nn.dynamicvars.aether_dynamicvars.AetherDynamicVars`` (aether_dynamicvars.py:15-273) -- ``predict_field``,
``single_step_forward`` and ``predict_future``, the path experiments/ind runs for its forecasting metrics.

Sub-modules carry the reference's names (``encoder``, ``decoder``, ``field_net``, ``coordinate_embedding``,
``angular_embedding``) and are created in its order, so a seeded construction or ``load_state_dict`` of a reference
checkpoint gives the same weights.  The training loss (``calculate_loss``) and the posterior encoder are not part
of this path.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from ... import _lib
from ..seq2seq.encoder import gumbel_softmax_hard
from ..seq2seq.field import _CoordinateEmbedding
from .decoder import Decoder
from .encoder import Encoder


class _DynFieldQueryParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("B", "ang_w", "ang_b", "w0", "b0", "w2", "b2", "w4", "b4")]


class _DynStepConfig(C.Structure):                      # AetherDynStepConfig of include/aether_hip.h
    _fields_ = [(n, C.c_int) for n in ("field_hidden", "encoder_hidden", "rnn_hidden", "prior_layers", "prior_hidden",
                                       "num_edge_types", "decoder_hidden", "skip_first", "encoder_polar", "decoder_polar",
                                       "knn_k")] + [("gumbel_tau", C.c_float)]


class AetherDynamicVars(nn.Module):
    one_call_step = True        # predict_future's steps go through aether_dyn_step (False: the three staged calls + torch glue)

    def __init__(self, params, device="cuda"):
        super().__init__()
        self.encoder = Encoder(params, device=None)                       # creation order of :19-62
        self.decoder = Decoder(params, device=None)
        self.num_edge_types = params.get("num_edge_types")
        self.gumbel_temp = params.get("gumbel_temp")
        self.kl_coef = params.get("kl_coef", 1.)                      # read by the training scripts
        self.num_dims = 2
        self.field_hidden = hidden_size = params["field_hidden"]
        if hidden_size % 32 != 0:
            raise ValueError("field_hidden must be a multiple of 32")
        self.field_net = nn.Sequential(nn.Linear(2 * hidden_size, hidden_size), nn.SiLU(),
                                       nn.Linear(hidden_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, 2))
        self.coordinate_embedding = _CoordinateEmbedding(2, hidden_size // 2, params.get("rff_std", 1.0))
        self.angular_embedding = nn.Linear(2, hidden_size)
        self._ws = None
        if device is not None:
            self.to(device)

    def save(self, path):
        torch.save(self.state_dict(), path)                       # as the reference's save / load

    def load(self, path):
        self.load_state_dict(torch.load(path))

    def _field_struct(self):
        fn = self.field_net
        tensors = [self.coordinate_embedding.B, self.angular_embedding.weight, self.angular_embedding.bias, fn[0].weight,
                   fn[0].bias, fn[2].weight, fn[2].bias, fn[4].weight, fn[4].bias]
        for t in tensors:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _lib.AetherHipError("field parameters must be contiguous fp32 CUDA tensors")
        return _DynFieldQueryParams(*[t.data_ptr() for t in tensors])

    @torch.no_grad()
    def predict_field(self, x, masks=None, n_present=None):
        """:64-79.  x [..., Nmax, 4] -> (field [..., Nmax, 2], zero where masks is 0; coords of the present objects).
        ``n_present``: the number of non-zero mask entries when the caller knows it on the host (no device round trip)."""
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd AetherDynamicVars runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        if masks is None:
            masks = torch.ones_like(x[..., 0], dtype=torch.bool)
        m = masks.to(x.device).bool()
        predicted_field = torch.zeros_like(x[..., :2], dtype=torch.float32)
        idx = None
        if n_present is None:
            xs = x[m].detach().to(torch.float32).contiguous()
        else:
            idx = torch.nonzero_static(m.reshape(-1), size=int(n_present))[:, 0]
            xs = x.detach().reshape(-1, x.shape[-1]).index_select(0, idx).to(torch.float32)
        coords = torch.cat([xs[..., :2], nn.functional.normalize(xs[..., 2:], dim=-1)], -1)
        n = xs.shape[0]
        if n == 0:
            return predicted_field, coords
        h = self.field_hidden
        ps = self._field_struct()
        need = lib.aether_dyn_field_workspace_bytes(n, h)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        out = torch.empty(n, 2, dtype=torch.float32, device=x.device)
        st = lib.aether_dyn_field(C.byref(ps), h, n, xs.data_ptr(), self._ws.data_ptr(), self._ws.numel(), out.data_ptr(),
                                  torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_dyn_field")
        if idx is None:
            predicted_field[m] = out
        else:
            predicted_field.view(-1, 2).index_copy_(0, idx, out)
        return predicted_field, coords

    @torch.no_grad()
    def single_step_forward(self, inputs, node_masks, graph_info, decoder_hidden, edge_logits, hard_sample, current_field,
                            uniform=None, n_present=None):
        """:133-145.  ``uniform``: the U(0,1) draw of gumbel_softmax ([E, K]); drawn on the device when omitted."""
        if not hard_sample:
            raise _lib.AetherHipError("only hard_sample=True (evaluation / prediction) is part of this path")
        if isinstance(edge_logits, (list, tuple)):                     # B scenes (single_step_forward_batched of the encoder)
            # one sampling pass over the edges of all scenes (torch.rand fills a tensor in index order: the draws of scene b
            # differ from those of B separate calls, as any batched sampler's do; pass `uniform` to fix them)
            K = edge_logits[0].shape[-1]
            sizes = [lg.nelement() // K for lg in edge_logits]
            lg_all = torch.cat([lg.reshape(-1, K) for lg in edge_logits])
            if uniform is None:
                u_all = torch.rand(lg_all.shape, device=lg_all.device)
            else:
                u_all = torch.cat([uniform[b].reshape(-1, K).to(lg_all.device) for b in range(len(edge_logits))])
            edges = list(gumbel_softmax_hard(lg_all, u_all, self.gumbel_temp).reshape(-1, K).split(sizes)) if lg_all.numel() \
                else [lg.reshape(0, K) for lg in edge_logits]
            predictions, decoder_hidden = self.decoder.forward_batched(inputs, decoder_hidden, edges, node_masks, graph_info,
                                                                       current_field)
            return predictions, decoder_hidden, edges
        if edge_logits.nelement() != 0:
            if uniform is None:
                uniform = torch.rand(edge_logits.shape, device=edge_logits.device)
            edges = gumbel_softmax_hard(edge_logits, uniform.reshape(edge_logits.shape).to(edge_logits.device), self.gumbel_temp)
        else:
            edges = torch.empty_like(edge_logits)
        predictions, decoder_hidden = self.decoder(inputs, decoder_hidden, edges, node_masks, graph_info, current_field,
                                                   n_present=n_present)
        return predictions, decoder_hidden, edges

    def _step_config(self):
        enc, dec = self.encoder, self.decoder
        _, n_layers, prior_hidden = enc._param_struct()
        return _DynStepConfig(self.field_hidden, enc.hidden_size, enc.rnn_hidden_size, n_layers, prior_hidden,
                              self.num_edge_types, dec.msg_out_shape, 1 if dec.skip_first_edge_type else 0,
                              1 if enc.pos_representation == "polar" else 0, 1 if dec.pos_representation == "polar" else 0,
                              10, float(self.gumbel_temp))

    @torch.no_grad()
    def _step_one_call(self, state, present, node_inds_t, gsend, grecv, e2n, prior_h, prior_c, dec_state, uniform_t,
                       pass_node_inds=True, return_edge_types=False):
        """``_step_core`` as ONE library call (``aether_dyn_step``): the same stage kernels, the index work between them in
        seven small kernels of the library instead of ~60 torch launches; bit-identical to the staged path."""
        lib = _lib.load()
        dev = state.device
        if self.encoder.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        Nmax, n, E = int(state.shape[1]), int(node_inds_t.numel()), int(gsend.numel())
        if grecv.numel() != E or e2n.ndim != 2 or e2n.shape[0] != n or uniform_t.numel() != E * self.num_edge_types:
            raise ValueError("graph_info / uniform do not match the present objects")
        i64 = lambda t: t.to(device=dev, dtype=torch.int64).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        state, present, uniform_t = f32(state), f32(present).reshape(-1), f32(uniform_t)
        ni, gs, gr, e2n = i64(node_inds_t), i64(gsend), i64(grecv), i64(e2n)
        if self.__dict__.get("_kernel_copies"):        # diagnostic (tools/dyn_graph_repro.py): copies as kernels, not memcpy nodes
            new_h, new_c, new_dec = (torch.add(f32(t), 0.0) for t in (prior_h, prior_c, dec_state))
        else:
            new_h, new_c, new_dec = f32(prior_h).clone(), f32(prior_c).clone(), f32(dec_state).clone()
        cfg = self._step_config()
        need = lib.aether_dyn_step_workspace_bytes(C.byref(cfg), Nmax, n, E)
        if need == 0:
            raise _lib.AetherHipError("aether_dyn_step_workspace_bytes: " + lib.aether_last_error().decode())
        ws = self.__dict__.get("_step_ws")
        if ws is None or ws.numel() < need or ws.device != dev:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.AetherHipError("the step workspace must be reserved outside graph capture (reserve())")
            ws = self.__dict__["_step_ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        pred = torch.empty(1, Nmax, 4, dtype=torch.float32, device=dev)
        edge_types = torch.empty(E, self.num_edge_types, dtype=torch.float32, device=dev) if return_edge_types else None
        fs = self._field_struct()
        ps_e = self.encoder._param_struct()[0]
        ps_d = self.decoder._param_struct()
        # node_inds = None: the library takes the mask's non-zero rows (what the data set's node_inds are)
        st = lib.aether_dyn_step(C.byref(fs), C.byref(ps_e), C.byref(ps_d), C.byref(cfg), Nmax, n, E, state.data_ptr(),
                                 present.data_ptr(), ni.data_ptr() if pass_node_inds else None, gs.data_ptr(), gr.data_ptr(),
                                 e2n.data_ptr(), int(e2n.shape[1]), new_h.data_ptr(), new_c.data_ptr(), new_dec.data_ptr(),
                                 uniform_t.data_ptr(), pred.data_ptr(),
                                 edge_types.data_ptr() if edge_types is not None else None, ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_step")
        if return_edge_types:
            return pred, new_h, new_c, new_dec, edge_types
        return pred, new_h, new_c, new_dec

    @torch.no_grad()
    def _step_core(self, state, present, node_inds_t, gsend, grecv, e2n, prior_h, prior_c, dec_state, uniform_t):
        """One step of ``predict_future`` for one scene: (prediction [1, Nmax, 4], prior state h, c, decoder state).  The
        number of present objects is ``node_inds_t.numel()`` (host-known): no stage reads a count back from the device."""
        if self.one_call_step:
            return self._step_one_call(state, present, node_inds_t, gsend, grecv, e2n, prior_h, prior_c, dec_state, uniform_t)
        n_t = int(node_inds_t.numel())
        gi = (gsend, grecv, e2n)
        field, _ = self.predict_field(state, present, n_present=n_t)
        logits, (new_h, new_c) = self.encoder.single_step_forward(state, present, node_inds_t, gi, (prior_h, prior_c), field,
                                                                   n_present=n_t)
        last, new_dec, _ = self.single_step_forward(state, present, gi, dec_state, logits, True, field, uniform_t,
                                                    n_present=n_t)
        return last, new_h, new_c, new_dec

    def reserve(self, n_objects_max, k=10):
        """Size the workspaces of the three stages for scenes of up to ``n_objects_max`` present objects (kNN graphs with
        ``k`` neighbours).  Captured steps (``predict_future(graph=True)``) bake workspace addresses into their graphs, so
        no stage may re-allocate after the first capture: this runs before it, outside any capture."""
        lib = _lib.load()
        dev = next(self.parameters()).device
        need_f = need_e = need_d = 0
        ps_e = self.encoder._param_struct()
        for n in range(2, int(n_objects_max) + 1):                    # host arithmetic only; the sizes need not be monotone
            E = n * min(int(k), n - 1)
            need_f = max(need_f, lib.aether_dyn_field_workspace_bytes(n, self.field_hidden))
            need_e = max(need_e, lib.aether_dyn_prior_workspace_bytes(self.encoder.hidden_size, self.encoder.rnn_hidden_size,
                                                                      ps_e[2], n, E))
            need_d = max(need_d, lib.aether_dyn_decoder_workspace_bytes(self.decoder.msg_out_shape, n, E))
        if self.one_call_step:
            cfg = self._step_config()
            need_s = 0
            for n in range(2, int(n_objects_max) + 1):
                need_s = max(need_s, lib.aether_dyn_step_workspace_bytes(C.byref(cfg), int(n_objects_max), n,
                                                                         n * min(int(k), n - 1)))
            ws = self.__dict__.get("_step_ws")
            if ws is None or ws.numel() < need_s or ws.device != dev:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.AetherHipError("reserve() must run outside graph capture")
                self.__dict__["_step_ws"] = torch.empty(need_s, dtype=torch.uint8, device=dev)
        for mod, need in ((self, need_f), (self.encoder, need_e), (self.decoder, need_d)):
            if mod._ws is None or mod._ws.numel() < need or mod._ws.device != dev:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.AetherHipError("reserve() must run outside graph capture")
                mod._ws = torch.empty(need, dtype=torch.uint8, device=dev)

    def _captured_step(self, args):
        """``_step_core`` as a hipGraph replay, one graph per (present objects, edges, slots) signature.

        Round 2 tried this and met a GPU memory access fault at the largest scene of a test.  The recorded
        facts point at buffer lifetime, not at the kernels (the eager path launches the same ones): the stages re-allocated
        their workspace whenever a step needed more than the last -- inside a capture that is an allocation from the
        graph's pool, and the buffer it replaced went back to a pool that earlier graphs still write to on replay.  Here
        nothing persistent is allocated during capture: workspaces are sized for the largest scene first (``reserve``),
        weight-derived caches are built by an eager warm-up call of the same signature, and the capture is discarded if a
        workspace pointer moved nevertheless.  The warm-up also validates, once per signature, the sizes the host
        hands down: the mask really has ``n_present`` non-zero entries."""
        state, present, ni, gs, gr, e2n, ph, pc, dec, u = args
        key = (int(ni.numel()), int(gs.numel()), tuple(e2n.shape), tuple(state.shape), tuple(ph.shape), tuple(dec.shape))
        cache = self.__dict__.setdefault("_step_graphs", {})
        hit = cache.get(key)
        if hit is None:
            if int((present != 0).sum()) != key[0]:                  # one device round trip per NEW signature only
                raise ValueError("node_inds and the mask disagree about the number of present objects")
            self.reserve(state.shape[1])
            static = [a.clone() for a in args]
            self._step_core(*static)                                  # eager: image caches, pointer structs, lazy init
            torch.cuda.synchronize()
            wsp = lambda: (self._ws.data_ptr(), self.encoder._ws.data_ptr(), self.decoder._ws.data_ptr(),
                           self.__dict__["_step_ws"].data_ptr() if self.__dict__.get("_step_ws") is not None else 0)
            ptrs = wsp()
            g = torch.cuda.CUDAGraph()
            pool = self.__dict__.setdefault("_step_pool", torch.cuda.graph_pool_handle())
            with torch.cuda.graph(g, pool=pool):
                outs = self._step_core(*static)
            if ptrs != wsp():
                raise _lib.AetherHipError("a workspace was re-allocated during capture: the graph is not safe to replay")
            hit = cache[key] = (g, static, outs)
        g, static, outs = hit
        for dst, src in zip(static, args):
            dst.copy_(src)
        # one replay in flight at a time: replays of a captured step enqueued behind a running one ended in a GPU memory
        # access fault while the decoder's all-types filter kernel was a graph node (bisected to that node
        # and the runtime's graph packet capture; the library no longer launches it) -- kept as a second line of defence
        if not self.__dict__.get("_capture_one_call"):
            torch.cuda.current_stream(static[0].device).synchronize()
        g.replay()
        # graphs share one memory pool: a later replay of another signature may reuse these buffers
        return tuple(o.clone() for o in outs)

    @torch.no_grad()
    def predict_future(self, inputs, masks, node_inds, graph_info, burn_in_masks, uniform=None, graph=False):
        """:245-273.  inputs [1, T, Nmax, 4], masks / burn_in_masks [1, T, Nmax], node_inds[0][t], graph_info[0][t]: the
        present objects and their graph per time step.  ``uniform``: per-step Gumbel draws (list of [E_t, K]).
        With B > 1 scenes (inputs [B, T, Nmax, 4], node_inds[b][t], graph_info[b][t], uniform[t][b]) every time step is
        ONE batched call per stage (predict_future_batched) -- the reference raises on batch > 1 (:588-591).
        One scene, default: the whole loop is ONE library call (``aether_dyn_rollout``).  With ``one_call_step = False``
        (the staged calls + torch glue of rounds 1-2) ``graph=True`` replays a captured hipGraph per step signature
        (``_captured_step``); bit-identical to the eager loop either way."""
        if inputs.size(0) > 1:
            return self.predict_future_batched(inputs, masks, node_inds, graph_info, burn_in_masks, uniform)
        n_steps = inputs.size(1) - 1
        if self.one_call_step and n_steps > 0 and not (graph and self.__dict__.get("_capture_one_call")):
            # ONE library call queues the whole loop (52 launches per step, no host round trip): nothing left for a
            # captured graph to save -- measured 0.35 ms per step against 0.41 ms for replays of a captured
            # aether_dyn_step.  ``graph=True`` is accepted and means the same thing here.  (replays of a
            # captured step in flight behind each other faulted while the decoder's all-types filter kernel was a graph
            # node -- a runtime replay problem, bisected and avoided in the library.)
            return self._predict_future_rollout(inputs, masks, node_inds, graph_info, burn_in_masks, uniform)
        prior_state = self.encoder.get_initial_hidden(inputs)
        dec_state = self.decoder.get_initial_hidden(inputs)
        last = inputs[:, 0]
        preds = []
        dev = inputs.device
        for t in range(n_steps):
            present = masks[:, t]
            observed = burn_in_masks[:, t].unsqueeze(-1).type(inputs.dtype)
            # observed objects are fed their ground truth, the others the model's own last prediction (:264)
            state = observed * inputs[:, t] + (1 - observed) * last
            # the data set hands over the present objects of every step (node_inds, built from the masks): their number
            # is known on the host, which spares the three stages their device round trips (mask -> index list)
            ni_t = node_inds[0][t]
            n_t = int(ni_t.numel())
            if n_t >= 2:
                gs, gr, e2n = (g.to(dev) for g in graph_info[0][t])
                u_t = uniform[t] if uniform is not None else torch.rand(gs.numel(), self.num_edge_types, device=dev)
                args = (state.to(torch.float32), present.to(dev).to(torch.float32), ni_t.to(dev), gs, gr, e2n,
                        prior_state[0], prior_state[1], dec_state.to(torch.float32),
                        u_t.to(dev).reshape(gs.numel(), self.num_edge_types).to(torch.float32))
                last, new_h, new_c, dec_state = self._captured_step(args) if graph else self._step_core(*args)
                prior_state = (new_h, new_c)
            else:                                                  # nobody or one object: the stages' own early exits
                field, _ = self.predict_field(state, present, n_present=n_t)
                logits, prior_state = self.encoder.single_step_forward(state, present, ni_t, graph_info[0][t], prior_state,
                                                                       field, n_present=n_t)
                last, dec_state, _ = self.single_step_forward(state, present, graph_info[0][t], dec_state, logits, True, field,
                                                              None if uniform is None else uniform[t], n_present=n_t)
            preds.append(last)
        return torch.stack(preds, dim=1)

    @torch.no_grad()
    def _predict_future_rollout(self, inputs, masks, node_inds, graph_info, burn_in_masks, uniform=None):
        """The whole loop of ``predict_future`` for one scene as ONE library call (``aether_dyn_rollout``): per step the
        burn-in mix and ``aether_dyn_step``, queued on the current stream without a host round trip in between."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd AetherDynamicVars runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.encoder.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        dev = inputs.device
        T, Nmax = int(inputs.size(1)), int(inputs.size(2))
        n_steps, K = T - 1, self.num_edge_types
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.to(device=dev, dtype=torch.int64).contiguous()
        x, m, burn = f32(inputs)[0], f32(masks)[0], f32(burn_in_masks)[0]
        if m.shape[0] < n_steps or burn.shape[0] < n_steps:
            raise ValueError("masks / burn_in_masks must cover every step")
        keep = []                                                  # device tensors the pointer arrays refer to
        n_host = (C.c_int64 * n_steps)()
        e_host = (C.c_int64 * n_steps)()
        e2n_max = []                                               # largest edge id of every step's edge2node (one sync below)
        deg = (C.c_int * n_steps)()
        ptrs = {k: (C.c_void_p * n_steps)() for k in ("ni", "gs", "gr", "e2n", "u")}
        for t in range(n_steps):
            ni_t = node_inds[0][t]
            n_t = int(ni_t.numel())
            n_host[t] = n_t
            if n_t < 2:
                continue
            gs, gr, e2n = (i64(g) for g in graph_info[0][t])
            E = int(gs.numel())
            if gr.numel() != E or e2n.ndim != 2 or e2n.shape[0] != n_t:
                raise ValueError(f"graph_info of step {t} does not match its present objects")
            u = f32(uniform[t]).reshape(-1, K) if uniform is not None else torch.rand(E, K, device=dev)
            if u.shape[0] != E:
                raise ValueError(f"uniform of step {t} must be [E, K]")
            ni = i64(ni_t)
            keep += [gs, gr, e2n, u, ni]
            e_host[t] = E
            e2n_max.append((t, E, e2n.max() if e2n.numel() else None))
            deg[t] = int(e2n.shape[1])
            for k, v in (("ni", ni), ("gs", gs), ("gr", gr), ("e2n", e2n), ("u", u)):
                ptrs[k][t] = v.data_ptr()
        cfg = self._step_config()
        # edge2node holds edge ids of the step's graph: an id >= E would index past the step's message rows
        live = [(t, E, mx) for t, E, mx in e2n_max if mx is not None]
        if live:
            mxs = torch.stack([mx for _, _, mx in live]).cpu().tolist()
            for (t, E, _), mx in zip(live, mxs):
                if mx >= E:
                    raise ValueError(f"graph_info of step {t}: edge2node names edge {mx}, the graph has {E} edges")
        need = lib.aether_dyn_rollout_workspace_bytes(C.byref(cfg), Nmax, n_steps, n_host, e_host)
        if need == 0:
            raise _lib.AetherHipError("aether_dyn_rollout_workspace_bytes: bad sizes (2..8192 object rows, hidden sizes, or a "
                                      "step whose graph does not list n * min(knn_k, n - 1) edges, the size of the encoder's "
                                      "own kNN graph)")
        ws = self.__dict__.get("_step_ws")
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self.__dict__["_step_ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        prior_h, prior_c = (f32(s)[0].clone() for s in self.encoder.get_initial_hidden(inputs))
        dec = f32(self.decoder.get_initial_hidden(inputs))[0].clone()
        preds = torch.empty(n_steps, Nmax, 4, dtype=torch.float32, device=dev)
        fs, ps_e, ps_d = self._field_struct(), self.encoder._param_struct()[0], self.decoder._param_struct()
        st = lib.aether_dyn_rollout(C.byref(fs), C.byref(ps_e), C.byref(ps_d), C.byref(cfg), Nmax, n_steps, x.data_ptr(),
                                    m.data_ptr(), burn.data_ptr(), n_host, e_host, ptrs["ni"], ptrs["gs"], ptrs["gr"], ptrs["e2n"], deg,
                                    ptrs["u"], prior_h.data_ptr(), prior_c.data_ptr(), dec.data_ptr(), preds.data_ptr(),
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_rollout")
        del keep
        return preds.unsqueeze(0)

    @torch.no_grad()
    def _predict_future_rollout_batched(self, inputs, masks, node_inds, graph_info, burn_in_masks, uniform=None):
        """``predict_future`` of B scenes as ONE library call (``aether_dyn_rollout_batched``): per step the burn-in mix and
        ``aether_dyn_step_batched`` over the present objects of all scenes, queued without a host round trip.  The
        per-scene index arrays of a step are concatenated in scene order (scene-local numbering, as the single-scene call
        takes them); the time axis comes first in what the library sees."""
        if not inputs.is_cuda:
            raise _lib.AetherHipError("aether_amd AetherDynamicVars runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        if self.encoder.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        dev = inputs.device
        B, T, Nmax = int(inputs.size(0)), int(inputs.size(1)), int(inputs.size(2))
        n_steps, K = T - 1, self.num_edge_types
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        i64 = lambda t: t.to(device=dev, dtype=torch.int64)
        x = f32(inputs).transpose(0, 1).contiguous()                                   # [T][B][Nmax][4]
        m = f32(masks)[:, :n_steps].transpose(0, 1).contiguous()                       # [n_steps][B][Nmax]
        burn = f32(burn_in_masks)[:, :n_steps].transpose(0, 1).contiguous()
        if m.shape[0] < n_steps or burn.shape[0] < n_steps:
            raise ValueError("masks / burn_in_masks must cover every step")
        n_host = (C.c_int64 * (n_steps * B))()
        e_host = (C.c_int64 * (n_steps * B))()
        deg = (C.c_int * (n_steps * B))()
        ptrs = {k: (C.c_void_p * n_steps)() for k in ("ni", "gs", "gr", "e2n", "u")}
        keep, e2n_max = [], []
        empty = torch.zeros(0, dtype=torch.int64, device=dev)
        for t in range(n_steps):
            ni_l, gs_l, gr_l, e2n_l, u_l, bounds = [], [], [], [], [], []
            for b in range(B):
                n_b = int(node_inds[b][t].numel())
                n_host[t * B + b] = n_b
                if n_b < 2:
                    continue                                   # empty scene (0); a single object is refused by the library
                gs, gr, e2n = (i64(g) for g in graph_info[b][t])
                E = int(gs.numel())
                if gr.numel() != E or e2n.ndim != 2 or e2n.shape[0] != n_b:
                    raise ValueError(f"graph_info of scene {b}, step {t} does not match its present objects")
                e_host[t * B + b] = E
                deg[t * B + b] = int(e2n.shape[1])
                bounds.append((E, int(e2n.numel())))
                if uniform is not None:
                    u = f32(uniform[t][b]).reshape(-1, K)
                    if u.shape[0] != E:
                        raise ValueError(f"uniform of scene {b}, step {t} must be [E, K]")
                    u_l.append(u)
                ni_l.append(i64(node_inds[b][t])); gs_l.append(gs); gr_l.append(gr); e2n_l.append(e2n.reshape(-1))
            cat = lambda l: torch.cat(l).contiguous() if l else empty
            E_t = sum(int(g.numel()) for g in gs_l)
            u_t = torch.cat(u_l).contiguous() if uniform is not None and u_l else torch.rand(max(E_t, 1), K, device=dev)
            step = dict(ni=cat(ni_l), gs=cat(gs_l), gr=cat(gr_l), e2n=cat(e2n_l), u=u_t)
            keep.append(step)
            if step["e2n"].numel():        # edge2node names edges of its own scene's graph: one comparison per step
                bound = torch.repeat_interleave(torch.tensor([e for e, _ in bounds], device=dev),
                                                torch.tensor([c for _, c in bounds], device=dev), output_size=step["e2n"].numel())
                e2n_max.append((t, ((step["e2n"] >= bound) | (step["e2n"] < 0)).any()))
            for k in ptrs:
                ptrs[k][t] = step[k].data_ptr()
        if e2n_max:                                                # (one device round trip for all steps)
            flags = torch.stack([f for _, f in e2n_max]).cpu().tolist()
            for (t, _), bad in zip(e2n_max, flags):
                if bad:
                    raise ValueError(f"graph_info of step {t}: an edge2node entry names an edge its scene's graph does not have")
        cfg = self._step_config()
        need = lib.aether_dyn_rollout_batched_workspace_bytes(C.byref(cfg), B, Nmax, n_steps, n_host, e_host, deg)
        if need == 0:
            raise _lib.AetherHipError("aether_dyn_rollout_batched_workspace_bytes failed: " + lib.aether_last_error().decode())
        ws = self.__dict__.get("_step_ws")
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self.__dict__["_step_ws"] = torch.empty(need, dtype=torch.uint8, device=dev)
        prior_h, prior_c = (f32(s)[0].clone() for s in self.encoder.get_initial_hidden(inputs))
        dec = f32(self.decoder.get_initial_hidden(inputs)).clone()                       # [B][Nmax][hd]
        preds = torch.empty(n_steps, B, Nmax, 4, dtype=torch.float32, device=dev)
        fs, ps_e, ps_d = self._field_struct(), self.encoder._param_struct()[0], self.decoder._param_struct()
        st = lib.aether_dyn_rollout_batched(C.byref(fs), C.byref(ps_e), C.byref(ps_d), C.byref(cfg), B, Nmax, n_steps,
                                            x.data_ptr(), m.data_ptr(), burn.data_ptr(), n_host, e_host, deg, ptrs["ni"],
                                            ptrs["gs"], ptrs["gr"], ptrs["e2n"], ptrs["u"], prior_h.data_ptr(),
                                            prior_c.data_ptr(), dec.data_ptr(), preds.data_ptr(), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_dyn_rollout_batched")
        del keep
        return preds.transpose(0, 1).contiguous()

    @torch.no_grad()
    def predict_future_batched(self, inputs, masks, node_inds, graph_info, burn_in_masks, uniform=None):
        """``predict_future`` for B scenes at once: per time step one field query, one kNN + prior step, one sampling pass
        and one decoder step over the present objects of ALL scenes (BASELINE config 4: 64 scenes).  Default: the whole
        loop is ONE library call (``aether_dyn_rollout_batched``, up to 256 scenes); ``one_call_step = False`` keeps the
        staged calls with torch glue of round 2 (same stage kernels: bit-identical)."""
        if self.one_call_step and inputs.size(1) > 1 and inputs.size(0) <= 256:
            return self._predict_future_rollout_batched(inputs, masks, node_inds, graph_info, burn_in_masks, uniform)
        B, n_steps = inputs.size(0), inputs.size(1) - 1
        prior_state = self.encoder.get_initial_hidden(inputs)
        dec_state = self.decoder.get_initial_hidden(inputs)
        last = inputs[:, 0]
        preds = []
        for t in range(n_steps):
            present = masks[:, t]
            observed = burn_in_masks[:, t].unsqueeze(-1).type(inputs.dtype)
            state = observed * inputs[:, t] + (1 - observed) * last
            field, _ = self.predict_field(state, present)
            ni_t = [node_inds[b][t] for b in range(B)]
            gi_t = [graph_info[b][t] for b in range(B)]
            logits, prior_state = self.encoder.single_step_forward_batched(state, present, ni_t, gi_t, prior_state, field)
            last, dec_state, _ = self.single_step_forward(state, present, gi_t, dec_state, logits, True, field,
                                                          None if uniform is None else uniform[t])
            preds.append(last)
        return torch.stack(preds, dim=1)
