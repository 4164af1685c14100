"""Synthetic N-body batches shaped like the reference simulators' output.

Follows SURVEY.md section 8(d): positions ~ N(0, loc_std^2) with
loc_std = 1 for <=5 bodies and (N/5)^(1/3) otherwise
(experiments/lorentz/dataset/synthetic_sim.py:155,
experiments/electrostatic/dataset/electrostatic_field_sim.py:98-100),
velocities with random direction and norm 0.5 (:101-106), charges +-1 with
p = 1/2 (:63-64,81-84), ``edge_attr_orig = [q_i q_j, ||x_i - x_j||]``
(experiments/lorentz/main.py:243-246).  Everything is generated on the CPU
with a seeded torch.Generator so that a given seed yields the same batch on
every machine; tensors are moved to ``device`` afterwards.
"""
from __future__ import annotations

import torch

from .edges import get_edges, prepare_edge_attr


def make_batch(batch_size: int, n_nodes: int, num_dims: int, seed: int = 0,
               device="cpu", vel_norm: float = 0.5):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    B, N, D = int(batch_size), int(n_nodes), int(num_dims)
    loc_std = 1.0 if N <= 5 else (N / 5.0) ** (1.0 / 3.0)
    loc = torch.randn(B * N, D, generator=g) * loc_std
    vel = torch.randn(B * N, D, generator=g)
    vel = vel * vel_norm / vel.norm(dim=-1, keepdim=True)
    charges = (torch.randint(0, 2, (B * N, 1), generator=g).float() * 2.0 - 1.0)
    target = loc + vel * 1.0 + 0.05 * torch.randn(B * N, D, generator=g)
    edges = get_edges(B, N)
    rows, cols = edges
    q_prod = charges[rows] * charges[cols]
    edge_attr = prepare_edge_attr(loc, edges, q_prod)
    h = vel.norm(dim=-1, keepdim=True)
    out = dict(h=h, x=loc, vel=vel, charges=charges, edge_attr=edge_attr,
               target=target)
    out = {k: v.to(device) for k, v in out.items()}
    out["edges"] = [e.to(device) for e in edges]
    out["meta"] = dict(B=B, N=N, D=D, seed=int(seed))
    return out
