"""CPU restatement of the seq2seq Aether's field query.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under aether_amd/).

* ``fourier_features``  <- nn/nn/fourier_feature_mapper.py:7-21 (``FourierFeatureMapper``):
  ``x_proj = (2*pi*x) @ B``; ``cat[sin(x_proj), cos(x_proj)]``; ``B [D, h/2]`` is drawn once from
  ``numpy.random.default_rng(42).normal(0, std)`` and cast to fp32 (buffer ``coordinate_embedding.B``).
* ``predict_field``     <- nn/seq2seq/aether.py:72-78,86-90: ``coords = x[..., :D]``, then
  ``Linear(h,h) - SiLU - Linear(h,h) - SiLU - Linear(h,D)`` on the features.  Unlike the
  state2state field (A1) it sees positions only.

Parity status: PINNED by tests/golden/s2s_field_D{2,3}.npz, produced by oracle/make_golden_seq2seq.py
from the imported reference ``FourierFeatureMapper`` and a ``field_net`` built exactly as the
reference builds it (tests/test_oracle_golden.py::test_seq2seq_field).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def rff_matrix(num_dims: int, half: int, std: float = 1.0) -> torch.Tensor:
    """The buffer ``coordinate_embedding.B`` (fourier_feature_mapper.py:12-15)."""
    rng = np.random.default_rng(42)
    return torch.from_numpy(rng.normal(0, std, size=(num_dims, half))).float()


def fourier_features(x: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    x_proj = (2 * math.pi * x) @ B.to(x.dtype)                 # fourier_feature_mapper.py:19-20
    return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


def predict_field(sd, x: torch.Tensor, num_dims: int) -> torch.Tensor:
    """``sd``: 'coordinate_embedding.B', 'field_net.{0,2,4}.{weight,bias}' (aether.py:86-90)."""
    coords = x[..., :num_dims]
    h = fourier_features(coords, sd["coordinate_embedding.B"])
    h = F.silu(F.linear(h, sd["field_net.0.weight"], sd["field_net.0.bias"]))
    h = F.silu(F.linear(h, sd["field_net.2.weight"], sd["field_net.2.bias"]))
    return F.linear(h, sd["field_net.4.weight"], sd["field_net.4.bias"])


# ---------------------------------------------------------------------------------------------
# Row A9: augmented local frames of the seq2seq model
#   AugmentedLocalizer.forward            <- nn/utils/augmented_global_to_local.py:52-68
#   canonicalize_augmented_inputs         <- nn/utils/canonicalization.py:33-56
#   create_augmented_edge_attr_pos_vel    <- canonicalization.py:111-140 (2-D)
#   create_augmented_3d_edge_attr_pos_vel <- canonicalization.py:143-172 (3-D)
#   geometry helpers                      <- nn/utils/geometry.py:7-66,76-127
# Parity status: PINNED by tests/golden/s2s_localizer_D{2,3}_{polar,cart}.npz (imported reference
# AugmentedLocalizer, oracle/make_golden_seq2seq.py).
# ---------------------------------------------------------------------------------------------
_PI = math.pi


def _rot2(theta):
    c, s = torch.cos(theta), torch.sin(theta)                  # geometry.py:13-23; theta [..., 1]
    return torch.stack([torch.cat([c, -s], -1), torch.cat([s, c], -1)], -2)


def _rot3(theta, phi):
    c, s, cp, sp = torch.cos(theta), torch.sin(theta), torch.cos(phi), torch.sin(phi)     # geometry.py:24-34
    return torch.stack([torch.cat([cp * c, -s, sp * c], -1), torch.cat([cp * s, c, sp * s], -1),
                        torch.cat([-sp, torch.zeros_like(c), cp], -1)], -2)


def _spherical3(v):
    """rho, theta in [0, 2pi), phi (geometry.py:37-66 with symmetric_theta=False)."""
    rho = torch.norm(v, p=2, dim=-1, keepdim=True)
    theta = torch.atan2(v[..., [1]], v[..., [0]])
    theta = theta + (theta < 0).type_as(theta) * (2 * _PI)
    phi = torch.acos(torch.clamp(v[..., 2:] / (rho + 1e-7), min=-1.0, max=1.0))
    return rho, theta, phi


def _mv(R, v):
    return (R @ v.unsqueeze(-1)).squeeze(-1)


def canonicalize_augmented(x, use_3d):
    """canonicalization.py:33-56: ([0 | R^T v (2-D: |v| in slot 2) | R^T f], Rinv)."""
    if use_3d:
        vel, forces = x[..., 3:6], x[..., 6:9]
        _, theta, phi = _spherical3(vel)
        Rinv = _rot3(theta, phi)
        r = Rinv.transpose(-1, -2)
        return torch.cat([torch.zeros_like(x[..., :3]), _mv(r, vel), _mv(r, forces)], -1), Rinv
    vel, forces = x[..., 2:4], x[..., 4:6]
    Rinv = _rot2(torch.atan2(vel[..., [1]], vel[..., [0]]))
    canon = torch.zeros_like(x)
    canon[..., 2] = torch.norm(vel, dim=-1)
    canon[..., 4:6] = _mv(Rinv.transpose(-1, -2), forces)
    return canon, Rinv


def augmented_edge_attr(x, send, recv, use_3d):
    """Edge features of j -> i in i's frame; x [B, M, 3D], send / recv index the M axis."""
    xj, xi = x[:, send], x[:, recv]
    if not use_3d:                                              # canonicalization.py:111-140
        yaw_i = torch.atan2(xi[..., [3]], xi[..., [2]])
        r = _rot2(yaw_i).transpose(-1, -2)
        dyaw = torch.atan2(xj[..., 3], xj[..., 2]) - torch.atan2(xi[..., 3], xi[..., 2])   # angle_diff, geometry.py:116-127
        dyaw = torch.where(dyaw >= _PI, dyaw - 2 * _PI, dyaw)
        dyaw = torch.where(dyaw < -_PI, dyaw + 2 * _PI, dyaw)
        dyaw = (dyaw / _PI).unsqueeze(-1)
        dp = xj[..., :2] - xi[..., :2]
        dist = torch.norm(dp, dim=-1, keepdim=True)
        dth = torch.atan2(dp[..., 1], dp[..., 0]).unsqueeze(-1) - yaw_i
        dth = dth + (dth <= -_PI).type_as(dth) * (2 * _PI)      # wrap_angles(normalize=True), geometry.py:108-113
        dth = dth - (dth > _PI).type_as(dth) * (2 * _PI)
        dth = dth / _PI
        return torch.cat([_mv(r, dp), dyaw, dist, dth, _mv(r, xj[..., 2:4]), _mv(r, xj[..., 4:6])], -1)
    _, yaw_j, pitch_j = _spherical3(xj[..., 3:6])              # canonicalization.py:143-172
    _, yaw_i, pitch_i = _spherical3(xi[..., 3:6])
    r = _rot3(yaw_i, pitch_i).transpose(-1, -2)
    dp = xj[..., :3] - xi[..., :3]
    dist, _, _ = _spherical3(dp)
    M = r @ _rot3(yaw_j, pitch_j).transpose(-1, -2)
    euler = torch.stack([torch.atan2(M[..., 1, 0], M[..., 0, 0]), torch.asin(-M[..., 2, 0]),
                         torch.atan2(M[..., 2, 1], M[..., 2, 2])], -1)             # ZYX, not normalised (:154)
    rdp = _mv(r, dp)
    _, dth, dph = _spherical3(rdp)
    return torch.cat([rdp, euler, dist, dth, dph, _mv(r, xj[..., 3:6]), _mv(r, xj[..., 6:9])], -1)


EDGE_POS_IDX = {(False, "cart"): [0, 1, 2], (False, "polar"): [2, 3, 4],
                (True, "cart"): [0, 1, 2, 3, 4, 5], (True, "polar"): [3, 4, 5, 6, 7, 8]}    # :19-24


def augmented_localizer(x, use_3d=False, pos_representation="polar"):
    """augmented_global_to_local.py:52-68.  x [B, N, 3D] -> rel_feat [B, N, 7D+O], Rinv [B, N, D, D],
    edge_attr [B, N(N-1), 2(4D+O)+3D], edge_pos [B, N(N-1), D+O] for the fully connected graph."""
    B, N, F = x.shape
    D = 3 if use_3d else 2
    send, recv = torch.where(~torch.eye(N, dtype=bool))                               # :31-32
    send_x, recv_x = torch.where(~torch.eye(N + 1, N, dtype=bool))                    # :33-34, origin = node N
    canon, Rinv = canonicalize_augmented(x, use_3d)
    origin = torch.zeros(3 * D, dtype=x.dtype)
    origin[D] = 1.0                                                                    # pos 0, vel e1, force 0 (:41)
    ext = torch.cat([x, origin.expand(B, 1, 3 * D)], 1)
    ea = augmented_edge_attr(ext, send_x, recv_x, use_3d)
    ne = recv.shape[0]
    origin_ea, ea = ea[:, ne:], ea[:, :ne]                                             # :61-62
    edge_pos = ea[..., EDGE_POS_IDX[(use_3d, pos_representation)]]
    edge_attr = torch.cat([ea, canon[:, recv], origin_ea[:, recv]], -1)
    rel_feat = torch.cat([canon, origin_ea], -1)
    return rel_feat, Rinv, edge_attr, edge_pos


# ---------------------------------------------------------------------------------------------
# Row A10 (decoder half): one step of the recurrent decoder
#   RecurrentDecoder.forward   <- nn/seq2seq/aether.py:590-654
#   Globalizer.forward         <- nn/utils/local_to_global.py:7-13
# Parity status: PINNED by tests/golden/s2s_decoder_D{2,3}.npz (imported reference RecurrentDecoder,
# oracle/make_golden_seq2seq.py; torch_scatter stand-in and an identity `.cuda()` as documented there).
# ---------------------------------------------------------------------------------------------
def _scatter_mean_dim1(src, index, n):
    """torch_scatter.scatter(src, index, dim=1, reduce='mean'): sum / max(count, 1) (aether.py:617,635)."""
    out = torch.zeros(src.shape[0], n, src.shape[2], dtype=src.dtype)
    out.index_add_(1, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
    return out / cnt.clamp(min=1).view(1, n, 1)


def decoder_step(sd, inputs, hidden, edges, predicted_field, use_3d=False, skip_first=False, send=None,
                 recv=None):
    """``sd``: the reference RecurrentDecoder's state_dict.  inputs [B, N, 2D], hidden [B, N, h], edges
    [B, N(N-1), K] (edge-type weights), predicted_field [B, N, D] -> (outputs [B, N, 2D], hidden')."""
    B, N, _ = inputs.shape
    D = 3 if use_3d else 2
    if send is None:
        send, recv = torch.where(~torch.eye(N, dtype=bool))
    K = edges.shape[-1]
    lin = lambda name, v: F.linear(v, sd[name + ".weight"], sd.get(name + ".bias"))
    pre_msg = torch.cat([hidden[:, recv], hidden[:, send]], -1)                       # :597-601 (receivers first)
    all_msgs = torch.zeros(B, recv.shape[0], hidden.shape[-1], dtype=inputs.dtype)
    for k in range(1 if skip_first else 0, K):                                        # :606-614
        msg = torch.tanh(lin(f"msg_fc1.{k}", pre_msg))
        msg = torch.tanh(lin(f"msg_fc2.{k}", msg))
        all_msgs = all_msgs + msg * edges[:, :, k:k + 1]
    agg = _scatter_mean_dim1(all_msgs, recv, N)                                       # :617
    ext = torch.cat([inputs, predicted_field], -1)                                    # :620
    rel_feat, Rinv, edge_attr, _ = augmented_localizer(ext, use_3d, "polar") if recv.shape[0] == N * (N - 1) \
        else (None, None, None, None)
    present = torch.zeros_like(all_msgs)
    for k in range(1 if skip_first else 0, K):                                        # :626-633
        msg = torch.relu(lin(f"present_msg_fc1.{k}", edge_attr))
        msg = torch.relu(lin(f"present_msg_fc2.{k}", msg))
        present = present + msg * edges[:, :, k:k + 1]
    pagg = _scatter_mean_dim1(present, recv, N)                                       # :635
    inp_r = lin("input_r", rel_feat) + lin("present_r", pagg)                         # :639-641
    inp_i = lin("input_i", rel_feat) + lin("present_i", pagg)
    inp_n = lin("input_n", rel_feat) + lin("present_n", pagg)
    r = torch.sigmoid(inp_r + lin("hidden_r", agg))                                   # :643-646
    i = torch.sigmoid(inp_i + lin("hidden_i", agg))
    n = torch.tanh(inp_n + r * lin("hidden_h", agg))
    hidden = (1 - i) * n + i * hidden
    pred = lin("out_mlp.6", torch.relu(lin("out_mlp.3", torch.relu(lin("out_mlp.0", hidden)))))   # :649
    pred_global = torch.cat([torch.einsum("...ij,...j->...i", Rinv, c) for c in pred.split(D, dim=-1)], -1)
    return inputs + pred_global, hidden                                               # :651-654


# ---------------------------------------------------------------------------------------------
# Row A10 (prior half): one step of the encoder's prior
#   Encoder.single_step_forward   <- nn/seq2seq/aether.py:384-410
#   AnisotropicEdgeFilter.forward <- nn/nn/anisotropic_filter.py:34-40
#   RefNRIMLP (eval)              <- nn/utils/model_utils.py:15-55
#   gumbel_softmax(hard)          <- nn/utils/model_utils.py:58-118 (noise passed in)
# Parity status: PINNED by tests/golden/s2s_prior_D{2,3}.npz (imported reference Encoder).
# ---------------------------------------------------------------------------------------------
def _refnri_mlp(sd, prefix, x):
    """Linear-ELU-Linear-ELU-BatchNorm1d in eval mode (running statistics), model_utils.py:18-55."""
    x = F.elu(F.linear(x, sd[prefix + ".model.0.weight"], sd[prefix + ".model.0.bias"]))
    x = F.elu(F.linear(x, sd[prefix + ".model.3.weight"], sd[prefix + ".model.3.bias"]))
    return (x - sd[prefix + ".bn.running_mean"]) / torch.sqrt(sd[prefix + ".bn.running_var"] + 1e-5) \
        * sd[prefix + ".bn.weight"] + sd[prefix + ".bn.bias"]


def prior_step(sd, inputs, prior_state, predicted_field, use_3d=False, pos_representation="cart",
               prior_layers=3):
    """``sd``: the reference Encoder's state_dict.  inputs [B, N, 2D], prior_state (h, c) each
    [B, E, rnn], predicted_field [B, N, D] -> (logits [B, E, K], (h', c'))."""
    B, N, _ = inputs.shape
    send, recv = torch.where(~torch.eye(N, dtype=bool))          # np.where(ones - eye): the same order (:251-253)
    ext = torch.cat([inputs, predicted_field], -1)                                     # :385
    rel_feat, _, edge_attr, edge_pos = augmented_localizer(ext, use_3d, pos_representation)
    hw = F.elu(F.linear(edge_pos, sd["edge_filter.edge_filter.0.weight"], sd["edge_filter.edge_filter.0.bias"]))
    ew = F.linear(hw, sd["edge_filter.edge_filter.2.weight"], sd["edge_filter.edge_filter.2.bias"])
    nrf = edge_attr.shape[-1]
    ew = ew.reshape(ew.shape[:-1] + (nrf, ew.shape[-1] // nrf))                        # anisotropic_filter.py:36-38
    ea = (edge_attr.unsqueeze(-2) @ ew).squeeze(-2)                                    # :39
    res_x = F.linear(rel_feat, sd["res1.weight"], sd["res1.bias"])                     # :393
    incoming = torch.zeros(B, N, ea.shape[-1], dtype=ea.dtype).index_add_(1, recv, ea) # edge2node, :340-348
    x = incoming / (N - 1) + res_x
    x = _refnri_mlp(sd, "mlp3", x)
    x = torch.cat([x[:, send], x[:, recv], ea], -1)                                    # node2edge + skip, :333-338,397
    x = _refnri_mlp(sd, "mlp4", x)
    h0, c0 = prior_state
    gates = F.linear(x, sd["forward_rnn.weight_ih_l0"], sd["forward_rnn.bias_ih_l0"]) + \
        F.linear(h0, sd["forward_rnn.weight_hh_l0"], sd["forward_rnn.bias_hh_l0"])   # nn.LSTM, one step (:405)
    i, f, g, o = gates.chunk(4, -1)
    c1 = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(g)
    h1 = torch.sigmoid(o) * torch.tanh(c1)
    y = h1
    for layer in range(prior_layers):                                                  # prior_fc_out, :291-303
        name = "prior_fc_out" if prior_layers == 1 else f"prior_fc_out.{2 * layer}"   # a bare Linear when 1 layer
        y = F.linear(y, sd[name + ".weight"], sd[name + ".bias"])
        if layer + 1 < prior_layers:
            y = F.elu(y)
    return y, (h1, c1)


def gumbel_hard(logits, uniform, tau):
    """gumbel_softmax(hard=True) with the uniform noise U supplied (model_utils.py:58-118):
    g = -log(eps - log(U + eps)); y_soft = softmax((logits + g) / tau); y = onehot(argmax) - y_soft + y_soft."""
    eps = 1e-10
    g = -torch.log(eps - torch.log(uniform + eps))
    y_soft = F.softmax((logits + g) / tau, dim=-1)
    y_hard = torch.zeros_like(y_soft).scatter_(-1, y_soft.argmax(-1, keepdim=True), 1.0)
    return y_hard - y_soft + y_soft


# ---------------------------------------------------------------------------------------------
# predict_future (nn/seq2seq/aether.py:155-191) restated with the step functions above.  The burn-in half
# runs the encoder step by step: its prior path is causal (forward LSTM from the zero state, per-feature
# BatchNorm in eval mode), so Encoder.forward's prior logits / state (:350-382) equal the chained
# single_step_forward; the reverse LSTM and encoder_fc_out only feed the training posterior.
# Parity status: PINNED by tests/golden/s2s_future_D2.npz (the imported reference Aether.predict_future).
# ---------------------------------------------------------------------------------------------
def predict_future(sd, inputs, prediction_steps, uniform, tau, use_3d=False, pos_representation="cart",
                   prior_layers=3, rnn_hidden=None, return_edges=False, field_fn=None):
    """``sd``: the reference seq2seq Aether's state_dict.  inputs [B, T, N, 2D]; ``uniform``
    [T - 1 + prediction_steps, B * E, K]: the U(0,1) draws of gumbel_softmax in call order."""
    D = 3 if use_3d else 2
    B, T, N, _ = inputs.shape
    E = N * (N - 1)
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    dec = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    h = dec["hidden_r.weight"].shape[0]
    R = enc["forward_rnn.weight_hh_l0"].shape[1]
    hidden = torch.zeros(B, N, h, dtype=inputs.dtype)
    state = (torch.zeros(B, E, R, dtype=inputs.dtype), torch.zeros(B, E, R, dtype=inputs.dtype))
    draw = 0
    preds, edges_all = [], []

    if field_fn is None:
        field_fn = lambda x: predict_field(sd, x, D)

    def step(x, hidden, state, draw):
        field = field_fn(x)
        logits, state = prior_step(enc, x, state, field, use_3d, pos_representation, prior_layers)
        z = gumbel_hard(logits.reshape(-1, logits.shape[-1]), uniform[draw], tau).view(logits.shape)
        out, hidden = decoder_step(dec, x, hidden, z, field, use_3d)
        return out, hidden, state, z

    for t in range(T - 1):                                            # burn-in, :165-173
        _, hidden, state, _ = step(inputs[:, t], hidden, state, draw)
        draw += 1
    x = inputs[:, T - 1]                                              # :174
    for _ in range(prediction_steps):                                 # :175-185
        x, hidden, state, z = step(x, hidden, state, draw)
        draw += 1
        preds.append(x)
        edges_all.append(z)
    preds = torch.stack(preds, 1)
    return (preds, torch.stack(edges_all, 1)) if return_edges else preds


# ---------------------------------------------------------------------------------------------
# seq2seq dynamic-field variant (nn/seq2seq/dynamic_field_aether.py, the model
# scripts/gravitational_field_3d_aether.sh trains).  With use_charges False (no runner sets it) it is the
# seq2seq Aether whose field query is FiLM-conditioned on a summary of the burn-in trajectories:
#   GraphSummary.forward   <- nn/nn/graph_pool.py:50-71  (particle embedding, GRU over time, sinusoidal
#                             positional encoding :10-28, attention pooling over all (object, time) items)
#   FilmedNetwork / FiLM   <- nn/nn/filmed_network.py:27-35, nn/nn/film.py:53-60
#   predict_field          <- dynamic_field_aether.py:117-134
#   predict_future         <- dynamic_field_aether.py:207-246 (summary from inputs[:, :-1], fixed afterwards)
# torch_geometric's AttentionalAggregation (not installed) is restated from its published definition:
# out_b = sum_i softmax_b(gate_nn(x))_i * nn(x)_i with softmax = exp(g - max_b) / (sum_b exp(g - max_b) + 1e-16).
# Parity status: PINNED by tests/golden/s2s_dynfield_D3.npz (the imported reference GraphSummary /
# DynamicFieldAether with the same documented stand-in for AttentionalAggregation as
# oracle/make_golden_dynfield.py).
# ---------------------------------------------------------------------------------------------
def positional_encoding(d_model: int, length: int, dtype=torch.float32) -> torch.Tensor:
    """graph_pool.py:15-20: pe[t, 0::2] = sin(t * div), pe[t, 1::2] = cos(t * div)."""
    position = torch.arange(length).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def _gru_last(sd, prefix, y):
    """Final hidden state of nn.GRU(batch_first=True) from the zero state; y [S, T, H]."""
    w_ih, w_hh = sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"]
    b_ih, b_hh = sd[prefix + "bias_ih_l0"], sd[prefix + "bias_hh_l0"]
    H = w_hh.shape[1]
    h = torch.zeros(y.shape[0], H, dtype=y.dtype)
    for t in range(y.shape[1]):
        gi = F.linear(y[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
    return h


def graph_summary(sd, x):
    """``sd``: the 'graph_pooler.' keys with the prefix removed.  x [B, N, T, input_size] -> [B, H]."""
    Bt, N, T, _ = x.shape
    y = F.linear(x, sd["particle_embedding.weight"], sd["particle_embedding.bias"]).flatten(0, 1)
    emb = _gru_last(sd, "rnn.", y).reshape(Bt, N, 1, -1).expand(Bt, N, T, -1)
    a = torch.cat([x, emb], -1)                                            # graph_pool.py:66-67
    a = a + positional_encoding(a.shape[-1], T, x.dtype)                   # :68 (dropout: eval mode)
    a = a.reshape(Bt, N * T, -1)                                           # items of graph b: all (object, time)
    g = F.linear(F.silu(F.linear(a, sd["summary_net.gate_nn.0.weight"], sd["summary_net.gate_nn.0.bias"])),
                 sd["summary_net.gate_nn.2.weight"], sd["summary_net.gate_nn.2.bias"])
    v = F.linear(F.silu(F.linear(a, sd["summary_net.nn.0.weight"], sd["summary_net.nn.0.bias"])),
                 sd["summary_net.nn.2.weight"], sd["summary_net.nn.2.bias"])
    e = torch.exp(g - g.max(dim=1, keepdim=True).values)
    w = e / (e.sum(dim=1, keepdim=True) + 1e-16)
    return (w * v).sum(dim=1)


def _film(sd, prefix, y, z):
    def mlp(name):
        return F.linear(F.silu(F.linear(z, sd[f"{prefix}{name}.0.weight"], sd[f"{prefix}{name}.0.bias"])),
                        sd[f"{prefix}{name}.2.weight"], sd[f"{prefix}{name}.2.bias"])
    return (1.0 + mlp("gamma")) * y + mlp("beta")                          # film.py:58-60


def film_field(sd, x, summary, num_dims):
    """dynamic_field_aether.py:117-134 with use_charges False.  ``sd``: 'coordinate_embedding.B' and the
    'film_net.' keys; x [B, ..., >= D]; summary [B, Hg], broadcast over the middle axes."""
    coords = x[..., :num_dims]
    h = fourier_features(coords, sd["coordinate_embedding.B"])
    z = summary.reshape(summary.shape[0], *([1] * (x.ndim - 2)), summary.shape[-1])
    y = F.linear(h, sd["film_net.linear_1.weight"], sd["film_net.linear_1.bias"])
    y = F.silu(_film(sd, "film_net.film_1.", y, z))
    y = F.linear(y, sd["film_net.linear_2.weight"], sd["film_net.linear_2.bias"])
    y = F.silu(_film(sd, "film_net.film_2.", y, z))
    return F.linear(y, sd["film_net.linear_3.weight"], sd["film_net.linear_3.bias"])


def predict_future_dynamic_field(sd, inputs, prediction_steps, uniform, tau, use_3d=True, pos_representation="cart",
                                 prior_layers=3, return_edges=False):
    """dynamic_field_aether.py:207-246: the summary of inputs[:, :-1] conditions every field query."""
    D = 3 if use_3d else 2
    gp = {k[len("graph_pooler."):]: v for k, v in sd.items() if k.startswith("graph_pooler.")}
    summary = graph_summary(gp, inputs[:, :-1].transpose(2, 1).contiguous())
    return predict_future(sd, inputs, prediction_steps, uniform, tau, use_3d, pos_representation, prior_layers,
                          return_edges=return_edges, field_fn=lambda x: film_field(sd, x, summary, D))


# ---------------------------------------------------------------------------------------------
# Full-sequence encoder and the evaluation-mode loss (validation metrics of experiments/electrostatic/train.py)
#   Encoder.forward          <- nn/seq2seq/aether.py:350-382 (per-time-step features as in prior_step, a forward and
#                               a reverse LSTM over time, prior_fc_out on the forward states, encoder_fc_out on both)
#   Aether.calculate_loss    <- :103-153 with is_train=False (hard samples of the posterior, teacher forcing per
#                               val_teacher_forcing_steps), nll_* :186-216, kl_categorical_learned / _avg :218-236
# Parity status: PINNED by tests/golden/s2s_loss_D2.npz (the imported reference Aether.calculate_loss).
# ---------------------------------------------------------------------------------------------
def _encoder_features(sd, inputs, predicted_field, use_3d, pos_representation):
    """prior_step up to mlp4 for one time step: [B, E, h]."""
    B, N, _ = inputs.shape
    send, recv = torch.where(~torch.eye(N, dtype=bool))
    ext = torch.cat([inputs, predicted_field], -1)
    rel_feat, _, edge_attr, edge_pos = augmented_localizer(ext, use_3d, pos_representation)
    hw = F.elu(F.linear(edge_pos, sd["edge_filter.edge_filter.0.weight"], sd["edge_filter.edge_filter.0.bias"]))
    ew = F.linear(hw, sd["edge_filter.edge_filter.2.weight"], sd["edge_filter.edge_filter.2.bias"])
    nrf = edge_attr.shape[-1]
    ew = ew.reshape(ew.shape[:-1] + (nrf, ew.shape[-1] // nrf))
    ea = (edge_attr.unsqueeze(-2) @ ew).squeeze(-2)
    x = torch.zeros(B, N, ea.shape[-1], dtype=ea.dtype).index_add_(1, recv, ea) / (N - 1) + \
        F.linear(rel_feat, sd["res1.weight"], sd["res1.bias"])
    x = _refnri_mlp(sd, "mlp3", x)
    return _refnri_mlp(sd, "mlp4", torch.cat([x[:, send], x[:, recv], ea], -1))


def _lstm_seq(sd, prefix, xs):
    """nn.LSTM(batch_first) from the zero state over a list of [rows, h] inputs -> list of hidden states, final (h, c)."""
    R = sd[prefix + "weight_hh_l0"].shape[1]
    h = torch.zeros(xs[0].shape[:-1] + (R,), dtype=xs[0].dtype)
    c = torch.zeros_like(h)
    out = []
    for x in xs:
        g = F.linear(x, sd[prefix + "weight_ih_l0"], sd[prefix + "bias_ih_l0"]) + \
            F.linear(h, sd[prefix + "weight_hh_l0"], sd[prefix + "bias_hh_l0"])
        i, f, gg, o = g.chunk(4, -1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out.append(h)
    return out, (h, c)


def _head(sd, name, x):
    keys = sorted({k.rsplit(".", 1)[0] for k in sd if k.startswith(name)},
                  key=lambda n: int(n.split(".")[-1]) if n.split(".")[-1].isdigit() else 0)
    for j, n in enumerate(keys):
        x = F.linear(x, sd[n + ".weight"], sd[n + ".bias"])
        if j + 1 < len(keys):
            x = F.elu(x)
    return x


def encoder_forward(sd, inputs, predicted_field, use_3d=False, pos_representation="cart"):
    """``sd``: the Encoder's state_dict.  inputs [B, T, N, 2D], predicted_field [B, N, T, D] ->
    (prior_logits [B, T, E, K], posterior_logits [B, T, E, K], (h, c) of the forward LSTM [B, E, R])."""
    T = inputs.shape[1]
    xs = [_encoder_features(sd, inputs[:, t], predicted_field[:, :, t], use_3d, pos_representation) for t in range(T)]
    fwd, state = _lstm_seq(sd, "forward_rnn.", xs)
    rev, _ = _lstm_seq(sd, "reverse_rnn.", xs[::-1])
    rev = rev[::-1]
    prior = torch.stack([_head(sd, "prior_fc_out", h) for h in fwd], 1)
    post = torch.stack([_head(sd, "encoder_fc_out", torch.cat([a, b], -1)) for a, b in zip(fwd, rev)], 1)
    return prior, post, state


def nll_and_kl(cfg, preds, target, posterior_logits, prior_logits, num_vars):
    """nll (:186-216) and kl_categorical_learned (:218-226) as ``cfg`` selects."""
    kind = cfg.get("nll_loss_type", "crossent")
    if kind == "gaussian":
        neg = (preds - target) ** 2 / (2 * cfg["prior_variance"])
        const = 0.5 * math.log(2 * math.pi * cfg["prior_variance"])
        if cfg.get("normalize_nll_per_var", False):
            nll = neg.sum() / (target.size(0) * target.size(2))
        elif cfg.get("normalize_nll", False):
            nll = (neg.sum(-1) + const).view(preds.size(0), -1).mean(dim=1)
        else:
            nll = neg.view(target.size(0), -1).sum() / target.size(1)
    elif kind == "crossent":
        e = F.binary_cross_entropy_with_logits(preds, target, reduction="none").view(preds.size(0), -1)
        nll = e.mean(dim=1) if cfg.get("normalize_nll", False) else e.sum(dim=1)
    else:
        e = F.poisson_nll_loss(preds, target, reduction="none").view(preds.size(0), -1)
        nll = e.mean(dim=1) if cfg.get("normalize_nll", False) else e.sum(dim=1)
    prob = F.softmax(posterior_logits, dim=-1)
    kl_div = prob * (torch.log(prob + 1e-16) - F.log_softmax(prior_logits, dim=-1))
    if cfg.get("normalize_kl", False):
        kl = kl_div.sum(-1).view(prob.size(0), -1).mean(dim=1)
    elif cfg.get("normalize_kl_per_var", False):
        kl = kl_div.sum() / (num_vars * prob.size(0))
    else:
        kl = kl_div.view(prob.size(0), -1).sum(dim=1)
    if cfg.get("add_uniform_prior"):                                                   # :139-140, :40-58, :228-236
        K = prob.shape[-1]
        prior = np.full(K, 1.0 / K)
        if cfg.get("no_edge_prior") is not None:
            prior = np.full(K, (1 - cfg["no_edge_prior"]) / (K - 1))
            prior[0] = cfg["no_edge_prior"]
        log_prior = torch.FloatTensor(np.log(prior)).view(1, 1, K)
        avg = prob.mean(dim=2)
        kd = avg * (torch.log(avg + 1e-16) - log_prior)
        if cfg.get("normalize_kl", False):
            kl_avg = kd.sum(-1).view(prob.size(0), -1).mean(dim=1)
        elif cfg.get("normalize_kl_per_var", False):
            kl_avg = kd.sum() / (num_vars * prob.size(0))
        else:
            kl_avg = kd.view(prob.size(0), -1).sum(dim=1)
        kl = 0.5 * kl + 0.5 * kl_avg
    return nll, kl


def calculate_loss_eval(sd, cfg, inputs, uniform, use_3d=False, pos_representation="cart", teacher_forcing=True,
                        use_prior_logits=False):
    """Aether.calculate_loss(inputs, is_train=False, ...): -> (loss, nll, kl, posterior_logits, predictions).
    ``cfg``: the params dictionary entries the loss reads; ``uniform`` [T - 1, B E, K]: gumbel draws per step."""
    D = 3 if use_3d else 2
    B, T, N, _ = inputs.shape
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    dec = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    x = inputs[:, :-1].transpose(2, 1).contiguous()
    field = predict_field(sd, x, D)                                                   # [B, N, T - 1, D]
    prior, post, _ = encoder_forward(enc, inputs[:, :-1], field, use_3d, pos_representation)
    hidden = torch.zeros(B, N, dec["hidden_r.weight"].shape[0], dtype=inputs.dtype)
    tf_steps = cfg.get("val_teacher_forcing_steps", -1)
    preds = []
    for step in range(T - 1):
        if (teacher_forcing and (tf_steps == -1 or step < tf_steps)) or step == 0:
            cur, cur_f = inputs[:, step], field[:, :, step]
        else:
            cur, cur_f = predictions, predict_field(sd, predictions, D)
        logits = (prior if use_prior_logits else post)[:, step]
        z = gumbel_hard(logits.reshape(-1, logits.shape[-1]), uniform[step], cfg["gumbel_temp"]).view(logits.shape)
        predictions, hidden = decoder_step(dec, cur, hidden, z, cur_f, use_3d, cfg.get("skip_first", False))
        preds.append(predictions)
    preds = torch.stack(preds, 1)
    nll, kl = nll_and_kl(cfg, preds, inputs[:, 1:], post, prior, N)
    loss = (nll + cfg.get("kl_coef", 1.0) * kl).mean()
    return loss, nll, kl, post, preds
