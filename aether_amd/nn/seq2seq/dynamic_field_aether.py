"""MI355X prediction path of the reference's seq2seq ``DynamicFieldAether`` (SURVEY.md 8f N3;
nn/seq2seq/dynamic_field_aether.py, the model scripts/gravitational_field_3d_aether.sh trains).

It is the seq2seq ``Aether`` whose field query is conditioned on the burn-in trajectories: ``graph_pooler``
(``GraphSummary``, nn/nn/graph_pool.py:31-71) summarises ``inputs[:, :-1]`` once per sequence and
``film_net`` (``FilmedNetwork``, nn/nn/filmed_network.py:7-35) replaces ``field_net``.  Same constructor
dictionary, sub-module names and ``state_dict`` keys as the reference (``load_state_dict`` of a reference
checkpoint works); ``use_charges`` -- which no runner of the reference sets -- must be False.  The encoder prior
step and the decoder step are the ones of ``aether_amd.nn.seq2seq.aether``.  Inference only; the computation
runs in libaether_hip.so (``aether_s2s_graph_summary``, ``aether_s2s_film_modulation``,
``aether_s2s_film_field``); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from ... import _lib
from ..state2state.dynamic_field_aether import _AttentionalAggregation
from .decoder import RecurrentDecoder
from .aether import _EvalLoss, _StepLoop
from .encoder import Encoder, gumbel_softmax_hard
from .field import _CoordinateEmbedding


class _SummaryParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "emb_w", "emb_b", "gru_w_ih", "gru_w_hh", "gru_b_ih", "gru_b_hh", "pe",
        "gate_w0", "gate_b0", "gate_w2", "gate_b2", "nn_w0", "nn_b0", "nn_w2", "nn_b2")]


class _FilmParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("B", "lin1_w", "lin1_b", "lin2_w", "lin2_b", "lin3_w", "lin3_b")] + \
               [(n, C.c_void_p * 4) for n in ("mod_w0", "mod_b0", "mod_w2", "mod_b2")]


def _check(tensors, what):
    for t in tensors:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.AetherHipError(f"{what} parameters must be contiguous fp32 CUDA tensors")


class _PositionalEncoding(nn.Module):
    """Holder of the ``pe`` buffer (graph_pool.py:10-20); dropout is the identity in eval mode."""

    def __init__(self, d_model, max_len=100):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class GraphSummary(nn.Module):
    """graph_pool.py:31-71: x [B, N, T, input_size] -> [B, hidden_size]."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        if hidden_size % 16 != 0:
            raise ValueError("graph_hidden must be a multiple of 16")
        if (input_size + hidden_size) % 2 != 0:
            raise ValueError("input_size + graph_hidden must be even (the reference's positional encoding)")
        self.input_size, self.hidden_size = input_size, hidden_size
        d = input_size + hidden_size
        self.summary_net = _AttentionalAggregation(                     # creation order of graph_pool.py:35-49
            nn.Sequential(nn.Linear(d, hidden_size), nn.SiLU(), nn.Linear(hidden_size, 1)),
            nn.Sequential(nn.Linear(d, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size)))
        self.particle_embedding = nn.Linear(input_size, hidden_size)
        self.rnn = nn.GRU(hidden_size, hidden_size, batch_first=True)
        self.pe = _PositionalEncoding(d)
        self._ws = None

    @torch.no_grad()
    def forward(self, x):
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd GraphSummary runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        B, N, T, in_size = x.shape
        if in_size != self.input_size:
            raise ValueError(f"last dimension of x must be {self.input_size}")
        if T > self.pe.pe.shape[1]:
            raise ValueError("more time steps than rows of the positional encoding (max_len=100)")
        H = self.hidden_size
        xf = x.detach().to(torch.float32).contiguous()
        g, v, r = self.summary_net.gate_nn, self.summary_net.nn, self.rnn
        tensors = [self.particle_embedding.weight, self.particle_embedding.bias, r.weight_ih_l0, r.weight_hh_l0,
                   r.bias_ih_l0, r.bias_hh_l0, self.pe.pe, g[0].weight, g[0].bias, g[2].weight, g[2].bias,
                   v[0].weight, v[0].bias, v[2].weight, v[2].bias]
        _check(tensors, "GraphSummary")
        ps = _SummaryParams(*[t.data_ptr() for t in tensors])
        need = lib.aether_s2s_graph_summary_workspace_bytes(B, N, T, in_size, H)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        out = torch.empty(B, H, dtype=torch.float32, device=x.device)
        st = lib.aether_s2s_graph_summary(C.byref(ps), B, N, T, in_size, H, self.pe.pe.shape[1], xf.data_ptr(),
                                          self._ws.data_ptr(), self._ws.numel(), out.data_ptr(),
                                          torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_s2s_graph_summary")
        return out


class _FiLM(nn.Module):
    def __init__(self, x_size, z_size, hidden_size):                   # nn/nn/film.py:43-55
        super().__init__()
        self.gamma = nn.Sequential(nn.Linear(z_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, x_size))
        self.beta = nn.Sequential(nn.Linear(z_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, x_size))


class _FilmedNetwork(nn.Module):
    def __init__(self, x_size, z_size, hidden_size, out_size):         # nn/nn/filmed_network.py:14-25
        super().__init__()
        self.linear_1 = nn.Linear(x_size, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)
        self.linear_3 = nn.Linear(hidden_size, out_size)
        self.film_1 = _FiLM(hidden_size, z_size, hidden_size)
        self.film_2 = _FiLM(hidden_size, z_size, hidden_size)


class DynamicFieldAether(_StepLoop, _EvalLoss, nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        if params.get("use_charges", False):
            raise ValueError("use_charges=True is not part of this path (no runner of the reference sets it)")
        self.num_vars = params["num_vars"]
        self.encoder = Encoder(params, device=None)                       # creation order of dynamic_field_aether.py:20-87
        if params.get("decoder_type", None) == "ref_mlp":
            raise ValueError("decoder_type 'ref_mlp' (MarkovDecoder) is not part of this path")
        self.decoder = RecurrentDecoder(params, device=None)
        self.num_edge_types = params.get("num_edge_types")
        self.gumbel_temp = params.get("gumbel_temp")
        self.kl_coef = params.get("kl_coef", 1.)                      # read by the training scripts
        self._init_loss_config(params)
        self.use_3d = params.get("use_3d", False)
        self.num_dims = 3 if self.use_3d else 2
        self.hidden_size = hidden_size = params["encoder_hidden"]
        self.graph_hidden, self.mlp_hidden = params["graph_hidden"], params["mlp_hidden"]
        if hidden_size % 32 != 0 or self.mlp_hidden % 16 != 0:
            raise ValueError("encoder_hidden must be a multiple of 32 and mlp_hidden a multiple of 16")
        self.coordinate_embedding = _CoordinateEmbedding(self.num_dims, hidden_size // 2, params.get("rff_std", 1.0))
        self.graph_pooler = GraphSummary(params["input_size"], self.graph_hidden)
        self.film_net = _FilmedNetwork(hidden_size, self.graph_hidden, self.mlp_hidden, self.num_dims)
        self.field = params.get("field")                                  # data-side grid helper (:88-95), unused here
        self._mod, self._ws, self._mod_buf = None, None, {}
        if device is not None:
            self.to(device)

    def calculate_loss(self, inputs, is_train=False, teacher_forcing=True, return_edges=False, return_logits=False,
                       use_prior_logits=False, charges=None, uniform=None):
        """dynamic_field_aether.py:151-205 in evaluation mode: as ``Aether.calculate_loss`` with the field conditioned on
        the summary of ``inputs[:, :-1]``.  (experiments/electrostatic/evaluate.py:42-45 inspects the signature for
        ``charges``, which must stay None: use_charges is not part of this path.)"""
        if charges is not None:
            raise _lib.AetherHipError("charges (use_charges) are not part of this path")
        return _EvalLoss.calculate_loss(self, inputs, is_train, teacher_forcing, return_edges, return_logits,
                                        use_prior_logits, uniform)

    def _sequence_field(self, inputs):
        x = inputs[:, :-1].transpose(2, 1).contiguous()                    # :160-166
        gr_summary = self.graph_pooler(x)
        predicted_field, _ = self.predict_field(x, gr_summary)
        return predicted_field, lambda state: self.predict_field(state, gr_summary)[0]

    def create_grid_points(self, box_size=5.0, grid_size=21, normalize=True):
        """dynamic_field_aether.py:89-95: the grid of the data-side ``field`` object (``params['field']``)."""
        test_positions = self.field._make_grid(box_size=box_size, grid_size=grid_size, ndim=self.num_dims)
        if normalize:
            test_positions = self.field._normalize(test_positions)
        return test_positions

    @torch.no_grad()
    def predict_field_at_grid(self, inputs, box_size=5.0, grid_size=21, charges=None, oracle=None):
        """dynamic_field_aether.py:103-115 (experiments/gravitational/evaluate.py:38-41): the field the summary of
        ``inputs[:, :-1]`` induces at the grid points, [B, grid points, D]."""
        if charges is not None:
            raise _lib.AetherHipError("charges (use_charges) are not part of this path")
        test_positions = self.create_grid_points(box_size=box_size, grid_size=grid_size, normalize=True).to(inputs.device)
        test_positions = test_positions.unsqueeze(0).repeat(inputs.size(0), 1, 1)
        x = inputs[:, :-1].transpose(2, 1).contiguous()
        gr_summary = self.graph_pooler(x)
        predicted_field, _ = self.predict_field(test_positions.contiguous(), gr_summary)
        return predicted_field

    # -- field query ---------------------------------------------------------------------
    def _film_struct(self):
        f = self.film_net
        tensors = [self.coordinate_embedding.B, f.linear_1.weight, f.linear_1.bias, f.linear_2.weight, f.linear_2.bias,
                   f.linear_3.weight, f.linear_3.bias]
        mods = [f.film_1.gamma, f.film_1.beta, f.film_2.gamma, f.film_2.beta]
        _check(tensors + [p for m in mods for p in m.parameters()], "film_net")
        ps = _FilmParams(*[t.data_ptr() for t in tensors])
        for k, m in enumerate(mods):
            ps.mod_w0[k], ps.mod_b0[k] = m[0].weight.data_ptr(), m[0].bias.data_ptr()
            ps.mod_w2[k], ps.mod_b2[k] = m[2].weight.data_ptr(), m[2].bias.data_ptr()
        version = sum(p._version for m in mods for p in m.parameters())
        return ps, version

    def _modulation(self, lib, ps, version, summary):
        """gamma / beta of both FiLM layers for this summary: computed once, reused while the summary tensor and
        the modulator weights are unchanged (the summary is fixed for a sequence, dynamic_field_aether.py:218)."""
        hit = self._mod
        if hit is not None and hit[0] is summary and hit[1] == summary._version and hit[2] == version:
            return hit[3]
        B = summary.shape[0]
        nbytes = lib.aether_s2s_film_modulation_bytes(B, self.mlp_hidden)
        mod = self._mod_buf.get((B, str(summary.device)))          # one buffer per batch size, rewritten in place:
        if mod is None:                                            # a captured step graph keeps reading it
            mod = self._mod_buf[(B, str(summary.device))] = torch.empty(nbytes // 4, dtype=torch.float32,
                                                                        device=summary.device)
        st = lib.aether_s2s_film_modulation(C.byref(ps), self.graph_hidden, self.mlp_hidden, B, summary.data_ptr(),
                                            mod.data_ptr(), nbytes, torch.cuda.current_stream(summary.device).cuda_stream)
        _lib.check(st, "aether_s2s_film_modulation")
        self._mod = (summary, summary._version, version, mod)
        return mod

    @torch.no_grad()
    def predict_field(self, x, graph_summary, charge_emb=None):
        """dynamic_field_aether.py:117-134: x [B, N, >=D] or [B, N, T, >=D], graph_summary [B, graph_hidden]
        -> (field [..., D], coords)."""
        if charge_emb is not None:
            raise _lib.AetherHipError("charge embeddings (use_charges) are not part of this path")
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd DynamicFieldAether runs on an MI355X only; got a CPU tensor "
                                      "(there is no CPU fallback)")
        lib = _lib.load()
        D, h, mh = self.num_dims, self.hidden_size, self.mlp_hidden
        if x.ndim < 2 or x.shape[-1] < D:
            raise ValueError(f"last dimension of x must hold at least {D} coordinates")
        B = x.shape[0]
        if graph_summary.shape != (B, self.graph_hidden):
            raise ValueError("graph_summary must be [batch, graph_hidden]")
        if not (graph_summary.is_cuda and graph_summary.dtype == torch.float32 and graph_summary.is_contiguous()):
            raise _lib.AetherHipError("graph_summary must be a contiguous fp32 CUDA tensor")
        coords = x[..., :D]
        pts = x.detach().to(torch.float32).reshape(-1, x.shape[-1]).contiguous()
        n = pts.shape[0]
        out = torch.empty(n, D, dtype=torch.float32, device=x.device)
        if n == 0:
            return out.reshape(*x.shape[:-1], D), coords
        ps, version = self._film_struct()
        mod = self._modulation(lib, ps, version, graph_summary)
        need = lib.aether_s2s_film_field_workspace_bytes(n, h, mh)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        st = lib.aether_s2s_film_field(C.byref(ps), D, h, mh, n, n // B, pts.data_ptr(), pts.shape[1], mod.data_ptr(), B,
                                       self._ws.data_ptr(), self._ws.numel(), out.data_ptr(),
                                       torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(st, "aether_s2s_film_field")
        return out.reshape(*x.shape[:-1], D), coords

    # -- steps -----------------------------------------------------------------------------
    @torch.no_grad()
    def single_step_forward(self, inputs, decoder_hidden, edge_logits, hard_sample, predicted_field, charge_emb=None,
                            uniform=None):
        """dynamic_field_aether.py:140-149; ``uniform`` as in ``aether_amd.nn.seq2seq.aether.Aether``."""
        if charge_emb is not None:
            raise _lib.AetherHipError("charge embeddings (use_charges) are not part of this path")
        if not hard_sample:
            raise _lib.AetherHipError("only hard_sample=True (evaluation / prediction) is part of this path")
        if uniform is None:
            uniform = torch.rand(edge_logits.shape, device=edge_logits.device)
        edges = gumbel_softmax_hard(edge_logits, uniform, self.gumbel_temp)
        predictions, decoder_hidden = self.decoder(inputs, decoder_hidden, edges, predicted_field)
        return predictions, decoder_hidden, edges

    def _graph_keepalive(self):
        return super()._graph_keepalive() + [self._ws, list(self._mod_buf.values())]

    @torch.no_grad()
    def predict_future(self, inputs, prediction_steps, return_edges=False, return_everything=False, charges=None,
                       uniform=None, graph=False):
        """dynamic_field_aether.py:207-246.  inputs [B, T, N, 2D].  The summary of ``inputs[:, :-1]`` conditions every
        field query; the burn-in half runs the (causal) prior step by step, as in ``Aether.predict_future``.
        ``uniform`` [T - 1 + steps, B, E, K] fixes the Gumbel draws.  ``graph``: replay the step from a captured
        hipGraph (not with ``return_everything``, which also collects the burn-in predictions)."""
        if charges is not None:
            raise _lib.AetherHipError("charges (use_charges) are not part of this path")
        B, T, N, _ = inputs.shape
        E = N * (N - 1)
        decoder_hidden = self.decoder.get_initial_hidden(inputs)
        R = self.encoder.rnn_hidden_size
        prior_hidden = (torch.zeros(B, E, R, device=inputs.device), torch.zeros(B, E, R, device=inputs.device))
        x = inputs[:, :-1].transpose(2, 1).contiguous()                    # :214
        gr_summary = self.graph_pooler(x)                                  # :218
        predicted_field, _ = self.predict_field(x, gr_summary)             # :219, [B, N, T - 1, D]
        if graph and not return_everything:
            mod = self._mod_buf[(B, str(inputs.device))]
            preds, edges, _ = self._graphed(lambda xx: self.predict_field(xx, gr_summary)[0], inputs[:, :T - 1].float(),
                                            inputs[:, T - 1].float(), decoder_hidden, prior_hidden,
                                            int(prediction_steps), uniform, return_edges, extra_key=(mod.data_ptr(),))
            return (preds, edges) if return_edges else preds
        # The fused step (aether_s2s_step) with the FiLM field handed in; the burn-in chains it on the observations (the
        # reference takes the burn-in's prior logits from the full-sequence encoder, :221-222: its prior path is causal, so
        # both give the same logits and state to rounding).
        all_predictions, all_edges = [], []
        E1 = self.encoder.recv_edges.shape[0]
        K = self.num_edge_types
        if uniform is None:
            uniform = torch.rand(T - 1 + int(prediction_steps), B, E1, K, device=inputs.device)
        uniform = uniform.reshape(T - 1 + int(prediction_steps), B, E1, K)
        for step in range(T - 1):
            field = predicted_field[:, :, step].contiguous()
            predictions, decoder_hidden, prior_hidden, edges = self._fused_step(
                inputs[:, step], decoder_hidden, prior_hidden, uniform[step], field)
            if return_everything:
                all_edges.append(edges)
                all_predictions.append(predictions)
        predictions = inputs[:, T - 1]
        for step in range(int(prediction_steps)):
            current_field, _ = self.predict_field(predictions, gr_summary)
            predictions, decoder_hidden, prior_hidden, edges = self._fused_step(
                predictions, decoder_hidden, prior_hidden, uniform[T - 1 + step], current_field)
            all_predictions.append(predictions)
            all_edges.append(edges)
        predictions = torch.stack(all_predictions, dim=1)
        if return_edges:
            return predictions, torch.stack(all_edges, dim=1)
        return predictions
