"""Host-side edge-index and edge-attribute helpers for the Aether hot path.

These mirror the callers either side of ``Aether.forward`` in the reference
runner:

* ``get_edges``            <- experiments/lorentz/dataset4newton.py:54-61,84-94
* ``prepare_edge_attr``    <- experiments/lorentz/main.py:243-246

Edge order is sender-major: for graph b, ``for i in range(N): for j != i`` gives
``send = b*N + i`` and ``recv = b*N + j``; int64, identical to
``torch.where(~torch.eye(N, dtype=bool))``.
"""
from __future__ import annotations

import torch


def fully_connected_edges(n_nodes: int):
    """Per-graph (rows, cols) python lists, dataset4newton.py:54-61."""
    rows, cols = [], []
    for i in range(n_nodes):
        for j in range(n_nodes):
            if i != j:
                rows.append(i)
                cols.append(j)
    return rows, cols


def get_edges(batch_size: int, n_nodes: int, device=None):
    """Batched fully-connected edge index, dataset4newton.py:84-94.

    Returns ``[send, recv]`` (two int64 tensors of length B*N*(N-1)).  Built
    with tensor arithmetic rather than the reference's per-graph python loop;
    the result is bit-identical (tests/test_edges.py pins it against tensors
    captured from the reference).
    """
    n = int(n_nodes)
    b = int(batch_size)
    if b < 1:
        raise ValueError("batch_size must be >= 1")
    i = torch.arange(n, dtype=torch.int64).repeat_interleave(n - 1)
    jj = torch.arange(n - 1, dtype=torch.int64).repeat(n)
    j = jj + (jj >= i).to(torch.int64)
    off = (torch.arange(b, dtype=torch.int64) * n).repeat_interleave(n * (n - 1))
    send = i.repeat(b) + off
    recv = j.repeat(b) + off
    if device is not None:
        send = send.to(device)
        recv = recv.to(device)
    return [send, recv]


def prepare_edge_attr(loc: torch.Tensor, edges, charge_edge_attr: torch.Tensor):
    """``edge_attr_orig = [q_i q_j, ||x_i - x_j||]``, main.py:243-246."""
    rows, cols = edges
    loc_dist = torch.sqrt(torch.sum((loc[rows] - loc[cols]) ** 2, 1)).unsqueeze(1)
    return torch.cat([charge_edge_attr, loc_dist], 1).detach()
