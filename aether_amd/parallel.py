"""Batch-parallel (data-parallel) use of the Aether step across the GPUs of one node.

Graphs of a batch are independent (the edge index never crosses graphs and the model has no
cross-sample operation), so each rank takes a contiguous block of graphs with rank-local node
numbering; the forward needs no communication.  Training adds ONE all-reduce of the flat fp32
gradient buffer (131,892 floats at D=2, ~0.5 MB: latency-bound) issued from inside the backward
of the step (`_AetherStep.backward`), then divides by the world size.  With the NCCL backend of
torch.distributed this is RCCL over xGMI; with gloo the same code runs on CPU tensors (tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_graphs(n_graphs: int, rank: int, world: int):
    """Contiguous block [lo, hi) of graphs for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_graphs), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    params = [p.data for p in module.parameters()]
    flat = torch.cat([p.reshape(-1) for p in params])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for p in params:
        p.copy_(flat[off:off + p.numel()].view_as(p))
        off += p.numel()


def allreduce_mean_(flat: torch.Tensor, group=None):
    """In-place mean over ranks of one flat buffer (what the step's backward does)."""
    dist.all_reduce(flat, group=group)
    flat.div_(dist.get_world_size(group))
    return flat


def attach_data_parallel(module, group=None, broadcast=True):
    """Mark `module` (aether_amd Aether or DynamicFieldAether) as data-parallel over `group` (default: WORLD)."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    module.dp_group = group if group is not None else dist.group.WORLD
    if broadcast:
        broadcast_parameters(module, 0, module.dp_group)
    return module
