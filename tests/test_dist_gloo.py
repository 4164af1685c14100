"""world_size = 2 (gloo) coverage of the data-parallel path: graph sharding, parameter broadcast and the flat-buffer
gradient all-reduce.  Without a GPU the local step is the oracle (test infrastructure: this container has none); on a box
with a GPU the same two ranks ALSO run the product's HIP step (the module under attach_data_parallel) and compare its
all-reduced gradients with the full-batch reference."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, load_state_dict
from aether_amd.parallel import allreduce_mean_, broadcast_parameters, shard_graphs


def test_shard_graphs_partition():
    for B in (1, 7, 128, 256):
        for W in (1, 2, 3, 8):
            spans = [shard_graphs(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from aether_amd.edges import get_edges, prepare_edge_attr
    from aether_amd.synthetic import make_batch
    from oracle import aether_oracle as O
    D, B, N = 2, 6, 5
    sd = load_state_dict(D)
    # rank 1 starts from different weights; the broadcast must overwrite them
    holder = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(v.clone() + (0.1 if rank else 0.0))
                                     for k, v in sd.items()})
    broadcast_parameters(holder, 0)
    for k, v in sd.items():
        assert torch.equal(holder[k.replace(".", "_")].data, v)
    full = make_batch(B, N, D, seed=3)
    lo, hi = shard_graphs(B, rank, world)
    sl = slice(lo * N, hi * N)
    edges = get_edges(hi - lo, N)                       # rank-local node numbering
    q_ = full["charges"][sl]
    ea = prepare_edge_attr(full["x"][sl], edges, q_[edges[0]] * q_[edges[1]])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.aether_forward(sdg, full["x"][sl], full["vel"][sl], edges, ea, q_)
    # the forward of a shard equals the rows of the full-batch forward (graphs are independent)
    ref_out = O.aether_forward(sd, full["x"], full["vel"], full["edges"], full["edge_attr"], full["charges"])
    assert torch.allclose(out.detach(), ref_out[sl], atol=2e-6)
    # local loss = mean over local rows; equal shard sizes => mean of means = global mean
    loss = torch.nn.functional.mse_loss(out, full["target"][sl])
    loss.backward()
    flat = torch.cat([sdg[k].grad.reshape(-1) for k in sd])
    allreduce_mean_(flat)
    # reference: gradient of the full-batch loss on one process
    sdf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    of = O.aether_forward(sdf, full["x"], full["vel"], full["edges"], full["edge_attr"], full["charges"])
    torch.nn.functional.mse_loss(of, full["target"]).backward()
    want = torch.cat([sdf[k].grad.reshape(-1) for k in sd])
    err = float((flat - want).abs().max() / want.abs().max())
    if torch.cuda.is_available():
        # the product's step on this rank's shard: all-reduce + mean happen inside the module's backward
        from aether_amd.nn.state2state.aether import Aether
        from aether_amd.parallel import attach_data_parallel
        dev = torch.device("cuda", 0)
        m = Aether(2 * D, 64, 0.0, D, device=dev)
        m.load_state_dict(sd)
        attach_data_parallel(m)
        dv = lambda t: t.to(dev)
        o = m(None, dv(full["x"][sl]), [dv(e) for e in edges], dv(full["vel"][sl]), dv(ea), dv(q_))
        torch.nn.functional.mse_loss(o, dv(full["target"][sl])).backward()
        got = torch.cat([p.grad.reshape(-1) for _, p in m.named_parameters()]).cpu()
        err = max(err, float((got - want).abs().max() / want.abs().max()) / 5.0)      # (HIP gradients: 5e-5 bar)
    q.put((rank, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gradient_allreduce_matches_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(170)
        assert p.exitcode == 0
    errs = dict(q.get(timeout=5) for _ in range(2))
    assert set(errs) == {0, 1}
    assert max(errs.values()) <= 1e-5, errs
