"""k-nearest-neighbour edge builder for variable-N scenes on the MI355X (first half).

``knn_edges(x, masks, k=10)`` mirrors ``Encoder.knn_edges`` of the reference's variable-N models
(nn/dynamicvars/aether_dynamicvars.py:559-586): same arguments, same ``(send_index, recv_index,
num_edges_per_batch)`` result, computed by ``aether_knn_edges`` (three launches, no [S, N, N] distance
matrix).  ``get_knn_graph_info(inputs, masks, num_vars)`` mirrors the send / recv half of
experiments/ind/single_ind_data.py:186-217 for one scene; instead of the reference's ``edge2node_inds`` (whose
``.view(-1, k)`` presumes k in-edges per object, which kNN graphs do not have) ``csr_by_receiver`` gives the
receiver-sorted order the library's kernels consume.  No CPU fallback.
"""
from __future__ import annotations


import torch

from . import _lib

MAX_K = 16


def _run(x, masks, k, n_present=None):
    if not (x.is_cuda and masks.is_cuda):
        raise _lib.AetherHipError("aether_amd.knn runs on an MI355X only; got a CPU tensor (there is no CPU fallback)")
    lib = _lib.load()
    N, D = x.shape[-2], x.shape[-1]
    if masks.shape != x.shape[:-1]:
        raise ValueError("masks must have the shape of x without its last axis")
    if D < 2:
        raise ValueError("x must hold at least the 2-D position in its last axis")
    k = min(int(k), N - 1)                                           # aether_dynamicvars.py:569
    if k > MAX_K:
        raise ValueError(f"k must be at most {MAX_K}")
    xs = x.detach().to(torch.float32).reshape(-1, N, D).contiguous()
    ms = masks.detach().to(torch.float32).reshape(-1, N).contiguous()
    S = xs.shape[0]
    dev = x.device
    scene_edges = torch.zeros(S, dtype=torch.int64, device=dev)
    scene_nodes = torch.zeros(S, dtype=torch.int64, device=dev)
    if S == 0 or k < 1:
        if S > 0:
            scene_nodes = (ms != 0).sum(-1)
        empty = torch.empty(0, dtype=torch.int64, device=dev)
        return empty, empty.clone(), scene_edges, scene_nodes
    cap = S * N * k
    send = torch.empty(cap, dtype=torch.int64, device=dev)
    recv = torch.empty(cap, dtype=torch.int64, device=dev)
    totals = torch.empty(2, dtype=torch.int64, device=dev)
    need = lib.aether_knn_workspace_bytes(S, N, k)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    st = lib.aether_knn_edges(xs.data_ptr(), D, ms.data_ptr(), S, N, k, send.data_ptr(), recv.data_ptr(),
                              scene_edges.data_ptr(), scene_nodes.data_ptr(), totals.data_ptr(), ws.data_ptr(),
                              ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "aether_knn_edges")
    if n_present is not None and S == 1:
        # one scene whose number of present objects the caller knows: every one of them has min(k, n - 1) neighbours
        n_present = int(n_present)
        E = n_present * min(k, max(n_present - 1, 0))
    else:
        E = int(totals[0].item())                                    # the reference's boolean-mask filter syncs here too
    return send[:E], recv[:E], scene_edges, scene_nodes


@torch.no_grad()
def knn_edges(x, masks, k=10, n_present=None):
    """x [..., T, N, D >= 2], masks [..., T, N] -> (send_index [E], recv_index [E], num_edges_per_batch):
    aether_dynamicvars.py:559-586.  ``num_edges_per_batch`` sums the edges over the last scene axis, as the
    reference does (``.sum([-1, -2, -3])`` of [..., T, N, k]).  ``n_present`` (one scene only): the number of non-zero
    mask entries, when the caller knows it on the host -- the edge count then needs no device round trip."""
    if x.ndim < 3:
        raise ValueError("x must be [..., T, N, D]")
    send, recv, scene_edges, _ = _run(x, masks, k, n_present)
    return send, recv, scene_edges.reshape(x.shape[:-2]).sum(-1)


@torch.no_grad()
def get_knn_graph_info(inputs, masks, num_vars=None, k=10):
    """One scene: inputs [N, D >= 2], masks [N] -> (send_edges, recv_edges) in the scene's compacted numbering
    (single_ind_data.py:186-217 with ``use_edge2node=False``; ``num_vars`` = present objects, which caps k)."""
    if num_vars is not None:
        if num_vars == 1:
            return None, None
        k = min(k, int(num_vars) - 1)
    send, recv, _, _ = _run(inputs.unsqueeze(0), masks.unsqueeze(0), k)
    return send, recv


@torch.no_grad()
def csr_by_receiver(recv, n_nodes):
    """(order, rowptr): edge ids grouped by receiver (stable) and CSR offsets -- what ``aether_s2s_decoder_step`` /
    ``aether_s2s_prior_step`` take in place of the reference's ``edge2node_inds``."""
    order = torch.argsort(recv, stable=True).contiguous()
    rowptr = torch.zeros(int(n_nodes) + 1, dtype=torch.int64, device=recv.device)
    # in-degrees without torch.bincount (which reads back the largest index: a device round trip per call)
    counts = torch.zeros(int(n_nodes), dtype=torch.int64, device=recv.device).scatter_add_(0, recv, torch.ones_like(recv))
    rowptr[1:] = torch.cumsum(counts, 0)
    return order, rowptr
