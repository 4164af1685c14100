"""world_size = 2 on ONE GPU (gloo): the HIP step under attach_data_parallel.  Each rank runs the module
on its block of graphs; the all-reduced flat gradient buffer must equal the full-batch gradients of a
single process, and both ranks must hold identical parameters after an optimizer step."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, load_state_dict

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, variant):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aether_amd.edges import get_edges, prepare_edge_attr
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.parallel import attach_data_parallel, shard_graphs
    from aether_amd.synthetic import make_batch
    D, B, N = 2, 8, 20
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)                        # ranks start from different weights
    if variant == "dynamic_field":
        from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
        torch.manual_seed(100 + 7 * rank)
        m = DynamicFieldAether(2 * D, 64, 0.0, D, device=dev)
    elif variant == "narrow":                            # hidden_size 32: the zero-padded 64-wide engine
        m = Aether(2 * D, 32, 0.0, D, device=dev)
    elif variant in ("wide", "wide_padded"):             # hidden_size 128 (csrc/wide.h) / 96 (zero-padded on it)
        m = Aether(2 * D, 128 if variant == "wide" else 96, 0.0, D, device=dev)
    else:
        m = Aether(2 * D, 64, 0.0, D, device=dev)
        if rank == 0:
            m.load_state_dict(load_state_dict(D))
    attach_data_parallel(m)                              # broadcast from rank 0
    full = make_batch(B, N, D, seed=9)
    lo, hi = shard_graphs(B, rank, world)
    sl = slice(lo * N, hi * N)
    edges = get_edges(hi - lo, N, device=dev)
    x, v, q_, tgt = (full[k][sl].to(dev) for k in ("x", "vel", "charges", "target"))
    ea = prepare_edge_attr(x, edges, q_[edges[0]] * q_[edges[1]])
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    extra = (N,) if variant == "dynamic_field" else ()
    start = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}     # after the broadcast
    out = m(v.norm(dim=-1, keepdim=True), x, edges, v, ea, q_, *extra)
    torch.nn.functional.mse_loss(out, tgt).backward()    # all-reduce + mean happen inside the backward
    grads = {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    params = {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()}
    q.put((rank, grads, params, out.detach().cpu().numpy().copy(), (lo, hi), start))      # numpy: pickled by value
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("variant", ["aether", "dynamic_field", "narrow", "wide", "wide_padded"])
def test_two_ranks_on_one_gpu_match_single_process_gradients(variant):
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    from conftest import scale_rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, variant)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    D, B, N = 2, 8, 20
    if variant == "dynamic_field":
        from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
        m = DynamicFieldAether(2 * D, 64, 0.0, D, device="cuda")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in res[0][5].items()})      # rank 0's broadcast weights
        assert all((res[0][5][k] == res[1][5][k]).all() for k in res[0][5])
        extra = (N,)
    elif variant in ("narrow", "wide", "wide_padded"):
        m = Aether(2 * D, {"narrow": 32, "wide": 128, "wide_padded": 96}[variant], 0.0, D, device="cuda")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in res[0][5].items()})
        assert all((res[0][5][k] == res[1][5][k]).all() for k in res[0][5])
        extra = ()
    else:
        m = Aether(2 * D, 64, 0.0, D, device="cuda")
        m.load_state_dict(load_state_dict(D))
        extra = ()
    full = make_batch(B, N, D, seed=9, device="cuda")
    out = m(full["h"], full["x"], full["edges"], full["vel"], full["edge_attr"], full["charges"], *extra)
    torch.nn.functional.mse_loss(out, full["target"]).backward()
    for rank, grads, params, out_r, (lo, hi), _ in res:
        assert torch.allclose(torch.from_numpy(out_r), out.detach().cpu()[lo * N:hi * N], atol=2e-6)   # no forward collective
        for k, p in m.named_parameters():
            if k.endswith("gate_nn.2.bias"):            # exactly zero (softmax shift invariance): rounding noise
                assert abs(float(grads[k][0])) <= 1e-9
                continue
            assert scale_rel_err(torch.from_numpy(grads[k]), p.grad.cpu()) <= 5e-5, (rank, k)
    for k in res[0][2]:
        assert (res[0][2][k] == res[1][2][k]).all(), k                                      # replicas stay in sync


def _worker_graphed(rank, world, port, q, hidden=64):
    """Three data-parallel steps through GraphedTrainStep (two graphs around an eager collective) and through the
    eager module path; both from the same start."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aether_amd.edges import get_edges, prepare_edge_attr
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.parallel import attach_data_parallel, shard_graphs
    from aether_amd.synthetic import make_batch
    from aether_amd.training import GraphedTrainStep
    D, B, N = 2, 8, 20
    dev = torch.device("cuda", 0)
    full = make_batch(B, N, D, seed=9)
    lo, hi = shard_graphs(B, rank, world)
    sl = slice(lo * N, hi * N)
    edges = get_edges(hi - lo, N, device=dev)
    x, v, q_, tgt = (full[k][sl].to(dev) for k in ("x", "vel", "charges", "target"))
    ea = prepare_edge_attr(x, edges, q_[edges[0]] * q_[edges[1]])
    h = v.norm(dim=-1, keepdim=True)
    res = {}
    for mode in ("eager", "graphed"):
        torch.manual_seed(100 + rank)
        m = Aether(2 * D, hidden, 0.0, D, device=dev)
        if rank == 0 and hidden == 64:
            m.load_state_dict(load_state_dict(D))
        attach_data_parallel(m)
        start = {k: t.detach().clone() for k, t in m.state_dict().items()}
        if mode == "graphed":
            step = GraphedTrainStep(m, [h, x, edges, v, ea, q_], tgt, warmup=1)
            # the warm-up steps moved the weights: restart both modes from rank 0's weights
            m.load_state_dict(start)
            for st in step.optimizer.state.values():
                for val in st.values():
                    if torch.is_tensor(val):
                        val.zero_()
            losses = [float(step.step().detach()) for _ in range(3)]
        else:
            opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-12)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                loss = torch.nn.functional.mse_loss(m(h, x, edges, v, ea, q_), tgt)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        res[mode] = (losses, {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()})
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("hidden", [64, 32])
def test_graphed_data_parallel_step_keeps_replicas_identical(hidden):
    """GraphedTrainStep under attach_data_parallel: forward + backward graph, eager all-reduce of the flat gradient
    buffer, optimizer graph.  Replicas bit-identical after three steps; same trajectory as the eager module path."""
    from conftest import scale_rel_err
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_graphed, args=(r, 2, port, q, hidden)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, r0), (_, r1) = res
    for k in r0["graphed"][1]:
        assert (r0["graphed"][1][k] == r1["graphed"][1][k]).all(), k          # replicas in sync, bit for bit
        a, b = torch.from_numpy(r0["graphed"][1][k]), torch.from_numpy(r0["eager"][1][k])
        assert scale_rel_err(a, b) <= 1e-4, k                                   # fused AdamW vs torch AdamW, 3 steps
    # each rank's loss is over its own graphs; the graphed and the eager path see the same numbers
    for a, b in zip(r0["graphed"][0], r0["eager"][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
