"""CPU oracle for the Aether state2state step.  TEST INFRASTRUCTURE ONLY.

This is a restatement (not a copy) of the reference algorithm for the hot path
named by BASELINE.json, written as plain functions over a ``state_dict``.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; the product path (``aether_amd``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every stage of
this file against ``tests/golden/*.npz``, which ``oracle/make_golden.py``
produced by importing and running the reference itself in the build container.

Reference lines followed, stage by stage (paths relative to the reference root):

* field net ............ nn/state2state/aether.py:108-134
* node frames .......... nn/utils/geometry.py:7-73, nn/state2state/aether.py:33-50
* edge features ........ nn/state2state/aether.py:52-100, nn/utils/geometry.py:76-105
* GNN layer x4 ......... nn/state2state/locs/locs.py:197-243 (scatter-mean :236-238)
* out MLP, globalise ... nn/state2state/locs/locs.py:160-168,193; nn/utils/local_to_global.py:7-13
* glue / residual ...... nn/state2state/aether.py:169-186

The functions are dtype-generic (fp32 for parity and the CPU baseline, fp64 for
the noise-floor figure) and differentiable, so ``torch.autograd`` provides the
gradient oracle for the backward kernels.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

TWO_PI = 2.0 * math.pi
EPS = 1e-7


def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


# ---------------------------------------------------------------- field net
def field_net(sd, x, vel, charges):
    """aether.py:108-134.  ``charges`` [Nn,1] float in {-1,0,+1}."""
    idx = (charges + 1).long().squeeze(1)                       # :122-124 truncation
    emb = sd["field_net.class_embedding.weight"][idx]
    z = torch.cat([x, vel, emb], dim=-1)
    z = F.silu(_linear(sd, "field_net.net.0", z))
    z = F.silu(_linear(sd, "field_net.net.2", z))
    return _linear(sd, "field_net.net.4", z)


# ------------------------------------------------------------------ geometry
def spherical_angles(v, symmetric_theta):
    """geometry.py:37-66 -> (rho, theta[, phi]) each [...,1]."""
    D = v.shape[-1]
    rho = torch.linalg.vector_norm(v, ord=2, dim=-1, keepdim=True)
    theta = torch.atan2(v[..., 1:2], v[..., 0:1])
    if not symmetric_theta:
        theta = theta + (theta < 0).to(theta.dtype) * TWO_PI      # :51-53
    if D == 2:
        return rho, theta, None
    phi = torch.acos(torch.clamp(v[..., 2:3] / (rho + EPS), min=-1.0, max=1.0))  # :63-64
    return rho, theta, phi


def frame_from_velocity(vel):
    """geometry.py:7-34,69-73 -> R [...,D,D] (un-transposed)."""
    D = vel.shape[-1]
    _, theta, phi = spherical_angles(vel, symmetric_theta=False)
    c, s = torch.cos(theta), torch.sin(theta)
    if D == 2:
        return torch.stack([torch.cat([c, -s], -1), torch.cat([s, c], -1)], -2)
    cp, sp = torch.cos(phi), torch.sin(phi)
    return torch.stack(
        [
            torch.cat([cp * c, -s, sp * c], -1),
            torch.cat([cp * s, c, sp * s], -1),
            torch.cat([-sp, torch.zeros_like(c), cp], -1),
        ],
        -2,
    )


def apply_rot(R, x):
    """geometry.py:104-105: einsum('...ij,...j->...i')."""
    return torch.einsum("...ij,...j->...i", R, x)


def euler_from_matrix(M, D):
    """geometry.py:76-101, normalised by pi (the default the localizer uses)."""
    if D == 2:
        e = torch.atan2(M[..., 1, 0:1], M[..., 0, 0:1])
    else:
        e = torch.stack(
            [
                torch.atan2(M[..., 1, 0], M[..., 0, 0]),
                torch.asin(-M[..., 2, 0]),                        # no clamp (:93)
                torch.atan2(M[..., 2, 1], M[..., 2, 2]),
            ],
            -1,
        )
    return e / math.pi


def canonical_nodes(ext, D):
    """aether.py:33-50: ext=[p|v|f] -> rel_feat [Nn,3D], R [Nn,D,D]."""
    vel = ext[..., D:2 * D]
    f = ext[..., 2 * D:3 * D]
    R = frame_from_velocity(vel)
    Rt = R.transpose(-1, -2)
    cv = apply_rot(Rt, vel)
    cf = apply_rot(Rt, f)
    return torch.cat([torch.zeros_like(cv), cv, cf], dim=-1), R


def edge_features(ext, send, recv, D):
    """aether.py:52-92: 4D + D(D-1)/2 local-frame columns per edge j->i."""
    xj, xi = ext[send], ext[recv]
    Ri = frame_from_velocity(xi[..., D:2 * D])
    Rit = Ri.transpose(-1, -2)
    rel = xj[..., :D] - xi[..., :D]
    rrel = apply_rot(Rit, rel)
    Rj = frame_from_velocity(xj[..., D:2 * D])
    euler = euler_from_matrix(Rit @ Rj, D)
    dist = torch.linalg.vector_norm(rel, ord=2, dim=-1, keepdim=True)
    _, th, ph = spherical_angles(rrel, symmetric_theta=True)
    sph = th if D == 2 else torch.cat([th, ph], -1)
    rv = apply_rot(Rit, xj[..., D:2 * D])
    rf = apply_rot(Rit, xj[..., 2 * D:3 * D])
    return torch.cat([rrel, euler, dist, sph, rv, rf], -1)


# ----------------------------------------------------------------------- GNN
def scatter_mean(e, recv, n_rows):
    """pytorch-scatter ``reduce='mean'`` as called at locs.py:236-238:
    sum over edges with that receiver / in-degree (clamped to >=1); rows with no
    in-edge are 0.  (The reference infers n_rows = max(recv)+1; the caller
    passes the node count, which is equal whenever the reference itself runs.)
    Summed sequentially in edge order via index_add_ on one thread."""
    out = torch.zeros(n_rows, e.shape[1], dtype=e.dtype, device=e.device)
    out = out.index_add(0, recv, e)
    deg = torch.zeros(n_rows, dtype=e.dtype, device=e.device)
    deg = deg.index_add(0, recv, torch.ones_like(recv, dtype=e.dtype))
    return out / deg.clamp(min=1).unsqueeze(1)


def gnn_layer(sd, prefix, x, e_in, send, recv, first):
    """locs.py:227-243."""
    if not first:
        e_in = torch.cat([x[send], x[recv], e_in], dim=-1)        # :233
    m = F.silu(_linear(sd, prefix + ".message_fn.0", e_in))
    m = F.silu(_linear(sd, prefix + ".message_fn.2", m))
    aggr = scatter_mean(m, recv, x.shape[0])
    x = (_linear(sd, prefix + ".res", x) if first else x) + aggr   # :214-218,240
    u = F.silu(_linear(sd, prefix + ".update_fn.0", x))
    x = x + _linear(sd, prefix + ".update_fn.2", u)                # :241
    return x, m


def out_mlp(sd, x, dropout_masks=None):
    """locs.py:160-168.  ``dropout_masks``: the two scale masks [n_nodes, hidden] (0 or 1 / (1 - p)) nn.Dropout would apply
    after the SiLUs in train() mode (:163, :166); None = Dropout p=0 / eval (runner: main.py:143)."""
    x = F.silu(_linear(sd, "gnn.out_mlp.0", x))
    if dropout_masks is not None:
        x = x * dropout_masks[0]
    x = F.silu(_linear(sd, "gnn.out_mlp.3", x))
    if dropout_masks is not None:
        x = x * dropout_masks[1]
    return _linear(sd, "gnn.out_mlp.6", x)


# --------------------------------------------------------------------- whole
def aether_forward(sd, x, vel, edges, edge_attr_orig, charges, return_all=False, field=None, dropout_masks=None):
    """aether.py:169-186.  ``sd`` maps reference state_dict keys to tensors.  ``field``: a precomputed
    per-node field replaces the built-in field net (the dynamic-field variant, dynamic_field_aether.py:84-97,
    is this function with ``dynamic_field`` below)."""
    D = x.shape[-1]
    send, recv = edges
    inputs = torch.cat([x, vel], dim=-1)
    if field is None:
        field = field_net(sd, x, vel, charges)
    ext = torch.cat([inputs, field], dim=-1)
    rel_feat, R = canonical_nodes(ext, D)
    ea = edge_features(ext, send, recv, D)
    ea_local = torch.cat([ea, rel_feat[recv]], -1)                  # aether.py:99
    ea_full = torch.cat([ea_local, edge_attr_orig], -1)             # aether.py:177
    res = {"field": field, "rel_feat": rel_feat, "R": R, "edge_attr_local": ea_local}
    h, e = rel_feat, ea_full
    for k in range(1, 5):
        h, e = gnn_layer(sd, f"gnn.layer_{k}", h, e, send, recv, first=(k == 1))
        res[f"x{k}"], res[f"e{k}"] = h, e
    pred = out_mlp(sd, h, dropout_masks)
    res["pred_local"] = pred
    pred = apply_rot(R, pred)                                       # local_to_global.py:12-13
    res["pred_global"] = pred
    out = x + pred
    res["out"] = out
    return res if return_all else out


# --------------------------------------------------------------------- dynamic-field variant
def dynamic_field(sd, x, vel, charges, num_nodes):
    """LatentFieldNetwork.forward (nn/state2state/dynamic_field_aether.py:31-48): per-graph attention-pooled
    summary of [pos | vel] (GraphSummary, graph_pool.py:7-29, with PyG's AttentionalAggregation semantics:
    softmax of gate_nn over the nodes of a graph -- exp(g - max) / (sum + 1e-16) -- weighting nn(x)), then a
    FiLM-modulated MLP on [pos | vel | class embedding] (film.py:5-60).  Graphs are consecutive blocks of
    ``num_nodes`` nodes.  Parity: pinned by tests/golden/dynfield_D*.npz (imported reference with the
    documented stand-in for the uninstalled torch_geometric op, oracle/make_golden_dynfield.py)."""
    pre = "field_net."
    lin = lambda name, v: F.linear(v, sd[pre + name + ".weight"], sd[pre + name + ".bias"])
    inputs = torch.cat([x, vel], -1)
    Bn = inputs.shape[0] // num_nodes
    xs = inputs.reshape(Bn, num_nodes, -1)
    gate = lin("summary_net.summary_net.gate_nn.2", F.silu(lin("summary_net.summary_net.gate_nn.0", xs)))
    val = lin("summary_net.summary_net.nn.2", F.silu(lin("summary_net.summary_net.nn.0", xs)))
    w = (gate - gate.max(dim=1, keepdim=True).values).exp()
    w = w / (w.sum(dim=1, keepdim=True) + 1e-16)
    summary = (w * val).sum(dim=1)                                             # [B, hidden]
    z = summary.repeat_interleave(num_nodes, dim=0)                            # graph_summary[batch], :41-44
    emb = sd[pre + "class_embedding.weight"][(charges + 1).long()].squeeze(1)  # :28-34
    y = lin("wrapper.linear_1", torch.cat([inputs, emb], -1))

    def film(name, y):                                                         # film.py:41-60
        m = lin(name + ".modulator.4", F.silu(lin(name + ".modulator.2", F.silu(lin(name + ".modulator.0", z)))))
        gamma, beta = torch.chunk(m, 2, dim=-1)
        return (1.0 + gamma) * y + beta

    y = F.silu(film("wrapper.film_1", y))
    y = F.silu(film("wrapper.film_2", lin("wrapper.linear_2", y)))
    return lin("wrapper.linear_3", y)


def dynamic_field_aether_forward(sd, x, vel, edges, edge_attr_orig, charges, num_nodes, dropout_masks=None):
    """DynamicFieldAether.forward (dynamic_field_aether.py:79-100)."""
    return aether_forward(sd, x, vel, edges, edge_attr_orig, charges,
                          field=dynamic_field(sd, x, vel, charges, num_nodes), dropout_masks=dropout_masks)


def cut_margin(edge_attr_local, D):
    """Distance (radians) of every edge's local-frame features from the branch cuts of the reference's own
    feature map: the relative-orientation angles atan2(.)/pi of euler_from_matrix (geometry.py:87-100) jump
    from +1 to -1 when sender and receiver headings are anti-parallel, and the un-normalised bearing
    atan2(r_y, r_x) (aether.py:72-75, symmetric theta) jumps by 2 pi when the sender sits exactly behind the
    receiver.  Two evaluations whose states differ by less than this margin's worth can land on different
    sides; the step is discontinuous there, in the reference as much as anywhere else."""
    a = edge_attr_local
    if D == 2:
        m = torch.minimum((1.0 - a[:, 2].abs()) * math.pi, math.pi - a[:, 4].abs())
    else:
        m = torch.minimum((1.0 - a[:, 3].abs()) * math.pi, (1.0 - a[:, 5].abs()) * math.pi)
        m = torch.minimum(m, math.pi - a[:, 7].abs())
    return m


def rollout(sd, x, vel, edges, charges, steps, dt=1.0, with_margin=False):
    """SURVEY.md 8(d) metric 2 for the state2state module:
    x_{t+1} = Aether(x_t, v_t); v_{t+1} = (x_{t+1} - x_t) / dt.
    ``with_margin``: also returns ``[steps, E]`` -- every edge's distance from a branch cut (``cut_margin``)."""
    rows, cols = edges
    qprod = charges[rows] * charges[cols]
    traj, margins = [], []
    D = x.shape[-1]
    for _ in range(steps):
        dist = torch.sqrt(torch.sum((x[rows] - x[cols]) ** 2, 1)).unsqueeze(1)
        ea = torch.cat([qprod, dist], 1)
        if with_margin:
            r = aether_forward(sd, x, vel, edges, ea, charges, return_all=True)
            xn = r["out"]
            margins.append(cut_margin(r["edge_attr_local"], D))
        else:
            xn = aether_forward(sd, x, vel, edges, ea, charges)
        vel = (xn - x) / dt
        x = xn
        traj.append(x)
    if with_margin:
        return torch.stack(traj), torch.stack(margins)
    return torch.stack(traj)
