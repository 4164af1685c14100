"""hidden_size < 64 (experiments/lorentz/main.py:42-43, --nf): the narrow model runs zero-padded on the 64-wide kernels.
Forward, parameter gradients, device rollout and a captured training step against the oracle (generic in the width)."""
import pytest
import torch

from conftest import scale_rel_err
from aether_amd import _lib
from aether_amd.nn.state2state.aether import Aether
from aether_amd.rollout import rollout
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
GTOL = 5e-5


def _dev(inp):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()} | {"edges": [e.cuda() for e in inp["edges"]]}


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D,H", [(2, 32), (3, 32), (2, 48), (3, 16), (2, 20)])
def test_narrow_model_forward_and_gradients_vs_oracle(D, H, flags):
    torch.manual_seed(11)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    m.flags = flags
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    host = make_batch(6, 7, D, seed=5)
    inp = _dev(host)
    # forward (inference)
    with torch.no_grad():
        got = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]).cpu()
        want = O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(got, want) <= TOL
    # parameter gradients of an MSE loss, narrow shapes
    m.zero_grad(set_to_none=True)
    out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    torch.nn.functional.mse_loss(out, inp["target"]).backward()
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ow = O.aether_forward(psd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    torch.nn.functional.mse_loss(ow, host["target"]).backward()
    for name, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, name
        ref = psd[name].grad
        scale = max(float(ref.abs().max()), 1e-6)
        assert float((p.grad.cpu() - ref).abs().max()) <= GTOL * max(scale, 1e-3), name


def test_narrow_model_follows_its_parameters_rollout_and_captured_training():
    D, H = 2, 32
    torch.manual_seed(12)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    host = make_batch(8, 20, D, seed=6)
    inp = _dev(host)
    call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    with torch.no_grad():
        a = call().clone()
        m.gnn.layer_3.message_fn[0].weight.mul_(0.5)           # in-place update: the padded engine has to follow
        b = call()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(b.cpu(), want) <= TOL and scale_rel_err(a.cpu(), want) > 1e-4
    # device rollout
    traj = rollout(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], 5).cpu()
    with torch.no_grad():
        wt = O.rollout(sd, host["x"], host["vel"], host["edges"], host["charges"], 5)
    assert scale_rel_err(traj, wt) <= TOL
    # a few optimizer steps as a captured hipGraph (the engine is re-synchronised inside the graph); the eager call after
    # them sees the updated weights (GraphedTrainStep bumps the version counters a replay leaves untouched)
    from aether_amd.training import GraphedTrainStep
    m.train()
    step = GraphedTrainStep(m, [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]], inp["target"],
                            lr=5e-4, weight_decay=1e-12, warmup=1)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}      # weights after the warm-up step
    losses = [float(step.step().item()) for _ in range(3)]
    assert losses[0] > losses[-1] or abs(losses[0] - losses[-1]) < 1e-3 * abs(losses[0])
    sd1 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert any(not torch.equal(sd0[k], sd1[k]) for k in sd0)
    assert all(sd1[k].shape == sd0[k].shape for k in sd0)
    with torch.no_grad():
        m.eval()
        got = call().cpu()
        want = O.aether_forward(sd1, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(got, want) <= TOL


def test_narrow_model_captured_training_follows_the_eager_loop():
    """Three AdamW steps of a hidden_size = 32 model through GraphedTrainStep and through the eager module: the padded
    engine has to be re-synchronised inside the graph on every replay (fused AdamW does not bump version counters)."""
    from aether_amd.training import GraphedTrainStep
    D, H = 2, 32
    inp = _dev(make_batch(8, 20, D, seed=6))
    args = [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]]
    res = {}
    for mode in ("eager", "graphed"):
        torch.manual_seed(12)
        m = Aether(2 * D, H, 0.0, D, device="cuda")
        start = {k: v.detach().clone() for k, v in m.state_dict().items()}
        if mode == "graphed":
            step = GraphedTrainStep(m, args, inp["target"], lr=5e-4, weight_decay=1e-12, warmup=1)
            m.load_state_dict(start)
            for st in step.optimizer.state.values():
                for val in st.values():
                    if torch.is_tensor(val):
                        val.zero_()
            losses = [float(step.step().item()) for _ in range(3)]
        else:
            opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-12)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                loss = torch.nn.functional.mse_loss(m(*args), inp["target"])
                loss.backward()
                opt.step()
                losses.append(float(loss.item()))
        res[mode] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    for a, b in zip(res["graphed"][0], res["eager"][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for k in res["eager"][1]:
        assert scale_rel_err(res["graphed"][1][k], res["eager"][1][k]) <= 1e-4, k


# ---------------------------------------------------------------------------------------------------------------------
# hidden_size > 64 (round 4): widths that are multiples of 64 run on the layer-by-layer GEMM path (csrc/wide.h), the
# others zero-padded on it.  Reference: nn/state2state/aether.py:143-158 with hidden_size = --nf
# (experiments/lorentz/main.py:42-43), nn/state2state/locs/locs.py:142-243.
import contextlib
import io
import os

import numpy as np

from conftest import GOLDEN


def _wide_fixture(H, D):
    d = np.load(os.path.join(GOLDEN, f"wide_H{H}_D{D}.npz"))
    host = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in.")}
    host["edges"] = [host.pop("send"), host.pop("recv")]
    return d, host


def _grad_check(m, psd, tol=GTOL):
    for name, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, name
        ref = psd[name].grad
        scale = max(float(ref.abs().max()), 1e-6)
        assert float((p.grad.cpu() - ref).abs().max()) <= tol * max(scale, 1e-3), name


@pytest.mark.parametrize("H,D", [(96, 2), (128, 3), (256, 2)])
def test_wide_model_vs_reference_golden(H, D):
    """The reference's own output, node states, messages and gradients at widths 96 / 128 / 256 (oracle/make_golden_wide.py);
    the module is built under the reference's seed, so its initialisation is the reference's."""
    d, host = _wide_fixture(H, D)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = Aether(2 * D, H, 0.0, D, device="cuda")
    for n, p in m.state_dict().items():
        assert tuple(p.shape) == tuple(int(v) for v in d["shape." + n]), n
        assert abs(float(p.double().sum()) - float(d["sum." + n])) <= 1e-9 * max(1.0, float(d["abs." + n])), n
    inp = _dev(host)
    m.flags = _lib.FLAG_KEEP_INTERMEDIATES
    with torch.no_grad():
        got = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    assert scale_rel_err(got.cpu(), torch.from_numpy(d["ref.out"])) <= TOL
    eng = m if m.hidden_size == m._kw else m._engine
    n_nodes, E = host["x"].shape[0], host["edges"][0].numel()
    perm = eng.graph_perm(inp["edges"], n_nodes).cpu()
    for k in range(1, 5):
        xk = eng.debug_fetch(f"x{k}", n_nodes, E, eng._kw).cpu()[:, :H]
        assert scale_rel_err(xk, torch.from_numpy(d[f"ref.x{k}"])) <= TOL, k
    for k in range(1, 4):
        ek = eng.debug_fetch(f"e{k}", n_nodes, E, eng._kw).cpu()[:, :H]
        assert scale_rel_err(ek, torch.from_numpy(d[f"ref.e{k}"])[perm]) <= TOL, k
    m.flags = 0
    m.zero_grad(set_to_none=True)
    out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    loss = torch.nn.functional.mse_loss(out, inp["target"])
    assert abs(float(loss.item()) - float(d["ref.loss"][0])) <= 1e-5 * abs(float(d["ref.loss"][0]))
    loss.backward()
    for n, p in m.named_parameters():
        g = p.grad.cpu()
        if "grad." + n in d.files:
            ref = torch.from_numpy(d["grad." + n])
            assert float((g - ref).abs().max()) <= GTOL * max(float(ref.abs().max()), 1e-3), n
        else:
            head = torch.from_numpy(d["ghead." + n])
            assert float((g.reshape(-1)[:head.numel()] - head).abs().max()) <= GTOL * max(float(head.abs().max()), 1e-3), n
            assert abs(float(g.double().sum()) - float(d["gsum." + n])) <= GTOL * max(float(d["gabs." + n]), 1e-3), n


@pytest.mark.parametrize("D,H,B,N", [(2, 96, 6, 7), (3, 128, 5, 9), (2, 192, 4, 20), (3, 256, 3, 20), (2, 128, 1, 2), (2, 70, 3, 5)])
def test_wide_model_forward_and_gradients_vs_oracle(D, H, B, N):
    torch.manual_seed(11)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    host = make_batch(B, N, D, seed=5)
    inp = _dev(host)
    with torch.no_grad():
        got = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]).cpu()
        want = O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(got, want) <= TOL
    # parameter and input gradients of an MSE loss
    m.zero_grad(set_to_none=True)
    xg, vg, eg = (inp[k].clone().requires_grad_(True) for k in ("x", "vel", "edge_attr"))
    out = m(inp["h"], xg, inp["edges"], vg, eg, inp["charges"])
    torch.nn.functional.mse_loss(out, inp["target"]).backward()
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hx, hv, he = (host[k].clone().requires_grad_(True) for k in ("x", "vel", "edge_attr"))
    ow = O.aether_forward(psd, hx, hv, host["edges"], he, host["charges"])
    torch.nn.functional.mse_loss(ow, host["target"]).backward()
    _grad_check(m, psd)
    for got_g, ref_g, name in ((xg.grad, hx.grad, "x"), (vg.grad, hv.grad, "vel"), (eg.grad, he.grad, "edge_attr")):
        assert float((got_g.cpu() - ref_g).abs().max()) <= GTOL * max(float(ref_g.abs().max()), 1e-3), name


def test_wide_model_sparse_graph_isolated_nodes_and_no_edges():
    """Irregular multigraph with receivers that get nothing, and a graph without any edge (locs.py:236-238: the mean of
    nothing is zero)."""
    D, H = 2, 128
    torch.manual_seed(3)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    n = 37
    x = torch.randn(n, D, generator=g)
    v = torch.randn(n, D, generator=g)
    v = 0.5 * v / v.norm(dim=-1, keepdim=True)
    q = torch.randint(0, 3, (n, 1), generator=g).float() - 1.0
    for E in (211, 0):
        send = torch.randint(0, n, (E,), generator=g)
        recv = torch.randint(0, n - 5, (E,), generator=g)           # the last five nodes receive nothing
        if E:
            recv[-1] = n - 6
        ea = torch.randn(E, 2, generator=g)
        target = x + v
        psd = {k: t.clone().requires_grad_(True) for k, t in sd.items()}
        if E:
            # the reference infers dim_size = max(recv) + 1 (locs.py:236-238): pad the oracle's aggregate to n rows by
            # giving the last node one self-loop-free in-edge in BOTH implementations
            send = torch.cat([send, torch.tensor([0])]); recv = torch.cat([recv, torch.tensor([n - 1])])
            ea = torch.cat([ea, torch.randn(1, 2, generator=g)])
            want = O.aether_forward(psd, x, v, [send, recv], ea, q)
            torch.nn.functional.mse_loss(want, target).backward()
        m.zero_grad(set_to_none=True)
        out = m(None, x.cuda(), [send.cuda(), recv.cuda()], v.cuda(), ea.cuda(), q.cuda())
        torch.nn.functional.mse_loss(out, target.cuda()).backward()
        assert torch.isfinite(out).all()
        if E:
            assert scale_rel_err(out.detach().cpu(), want.detach()) <= TOL
            _grad_check(m, psd)
        else:
            for name, p in m.named_parameters():
                assert torch.isfinite(p.grad).all(), name
                if ".message_fn." in name:
                    assert float(p.grad.abs().max()) == 0.0, name


def test_wide_model_dropout_rollout_and_captured_training():
    D, H = 2, 128
    torch.manual_seed(12)
    m = Aether(2 * D, H, 0.25, D, device="cuda")
    host = make_batch(8, 20, D, seed=6)
    inp = _dev(host)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    n = host["x"].shape[0]
    # train()-mode dropout with explicit masks (locs.py:163,166) against the oracle with the same masks
    m.train()
    gen = torch.Generator().manual_seed(4)
    masks = (torch.rand(2, n, H, generator=gen) < 0.75).float() / 0.75
    m.__dict__["_dropout_masks"] = masks
    m.zero_grad(set_to_none=True)
    out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    torch.nn.functional.mse_loss(out, inp["target"]).backward()
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ow = O.aether_forward(psd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"],
                          dropout_masks=(masks[0], masks[1]))
    torch.nn.functional.mse_loss(ow, host["target"]).backward()
    assert scale_rel_err(out.detach().cpu(), ow.detach()) <= TOL
    _grad_check(m, psd)
    del m.__dict__["_dropout_masks"]
    with pytest.raises(RuntimeError):
        rollout(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], 2)
    # eval(): identity; device rollout against the oracle's
    m.eval()
    traj = rollout(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], 5).cpu()
    with torch.no_grad():
        wt = O.rollout(sd, host["x"], host["vel"], host["edges"], host["charges"], 5)
    assert scale_rel_err(traj, wt) <= TOL
    # captured training step follows the eager loop
    from aether_amd.training import GraphedTrainStep
    args = [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]]
    res = {}
    for mode in ("eager", "graphed"):
        torch.manual_seed(12)
        mm = Aether(2 * D, H, 0.0, D, device="cuda")
        start = {k: v.detach().clone() for k, v in mm.state_dict().items()}
        if mode == "graphed":
            step = GraphedTrainStep(mm, args, inp["target"], lr=5e-4, weight_decay=1e-12, warmup=1)
            mm.load_state_dict(start)
            for st in step.optimizer.state.values():
                for val in st.values():
                    if torch.is_tensor(val):
                        val.zero_()
            losses = [float(step.step().item()) for _ in range(3)]
        else:
            opt = torch.optim.AdamW(mm.parameters(), lr=5e-4, weight_decay=1e-12)
            losses = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                loss = torch.nn.functional.mse_loss(mm(*args), inp["target"])
                loss.backward()
                opt.step()
                losses.append(float(loss.item()))
        res[mode] = (losses, {k: v.detach().cpu().clone() for k, v in mm.state_dict().items()})
    for a, b in zip(res["graphed"][0], res["eager"][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for k in res["eager"][1]:
        assert scale_rel_err(res["graphed"][1][k], res["eager"][1][k]) <= 1e-4, k


def test_wide_model_is_deterministic_and_full_size():
    """cfg2-sized batch (B = 128, N = 20) at width 128: bit-identical re-runs of forward and backward (no float atomics),
    output against the oracle."""
    D, H = 2, 128
    torch.manual_seed(1)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    host = make_batch(128, 20, D, seed=0)
    inp = _dev(host)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    runs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        torch.nn.functional.mse_loss(out, inp["target"]).backward()
        runs.append((out.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0])
    for a, b in zip(runs[0][1], runs[1][1]):
        assert torch.equal(a, b)
    with torch.no_grad():
        want = O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(runs[0][0].cpu(), want) <= TOL
