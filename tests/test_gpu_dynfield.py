"""the dynamic-field variant (attention-pooled graph summary + FiLM field net) vs the golden
vectors captured from the imported reference DynamicFieldAether and vs the oracle on fresh batches."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, scale_rel_err
from aether_amd import _lib
from aether_amd.edges import get_edges
from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _load(D):
    d = np.load(os.path.join(GOLDEN, f"dynfield_D{D}.npz"))
    sd = {str(k): torch.from_numpy(d["sd." + str(k)]) for k in d["keys"]}
    m = DynamicFieldAether(2 * D, 64, 0.0, D, device="cuda")
    assert list(m.state_dict().keys()) == list(sd.keys())               # reference key order
    m.load_state_dict(sd)
    return d, sd, m


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_matches_reference(D, flags):
    d, sd, m = _load(D)
    m.flags = flags
    for name in ("small", "cfg"):
        B, N = int(d[f"{name}.B"]), int(d[f"{name}.N"])
        t = lambda k: torch.from_numpy(d[f"{name}.in.{k}"]).cuda()
        edges = get_edges(B, N, device="cuda")
        with torch.no_grad():
            out = m(None, t("x"), edges, t("vel"), t("edge_attr"), t("charges"), N)
        assert scale_rel_err(m.last_field.cpu(), torch.from_numpy(d[f"{name}.ref.field"])) <= TOL, name
        assert scale_rel_err(out.cpu(), torch.from_numpy(d[f"{name}.ref.out"])) <= TOL, name


@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_fresh_batches_vs_oracle(D):
    d, sd, m = _load(D)
    for (B, N, seed) in [(1, 2, 1), (128, 20, 2), (3, 300, 3)]:          # 300 nodes: streamed path, several passes per thread
        inp = make_batch(B, N, D, seed=seed)
        want = O.dynamic_field_aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"], N)
        with torch.no_grad():
            out = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                    inp["charges"].cuda(), N)
        assert scale_rel_err(out.cpu(), want) <= TOL, (B, N)
        with torch.no_grad():
            out2 = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                     inp["charges"].cuda(), N)
        assert torch.equal(out, out2)                                   # deterministic, workspace reuse


# ---------------------------------------------------------------- training (backward)
GTOL = 5e-5


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_gradients_vs_oracle_autograd(D, flags):
    """Every parameter gradient of the HIP backward (aether_backward_field + aether_dynamic_field_backward) vs
    torch autograd through the oracle; also dL/dfield itself.  40 nodes: two LDS chunks per graph in the
    field-net backward; 130 graphs: more graphs than CUs / 2."""
    d, sd, m = _load(D)
    m.flags = flags
    for (B, N, seed) in [(4, 5, 11), (130, 20, 12), (3, 40, 13)]:
        inp = make_batch(B, N, D, seed=seed)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        field = O.dynamic_field(sdg, inp["x"], inp["vel"], inp["charges"], N)
        field.retain_grad()
        want = O.aether_forward(sdg, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"], field=field)
        torch.nn.functional.mse_loss(want, inp["target"]).backward()
        m.zero_grad(set_to_none=True)
        out = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                inp["charges"].cuda(), N)
        assert out.requires_grad and scale_rel_err(out.detach().cpu(), want.detach()) <= TOL
        torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
        assert scale_rel_err(m.last_grad_field.cpu(), field.grad) <= GTOL, (B, N)
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            if k.endswith("gate_nn.2.bias"):          # softmax is shift invariant: this gradient is exactly zero
                assert float(p.grad.abs().max()) <= 1e-9 and float(sdg[k].grad.abs().max()) <= 1e-9
                continue
            assert scale_rel_err(p.grad.cpu(), sdg[k].grad) <= GTOL, (B, N, k)


@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_dropout_step_vs_oracle_with_the_same_masks(D):
    """dropout_prob = 0.2 in train() mode: forward and every parameter gradient against the oracle's autograd with the same
    two out-MLP masks; eval() is the p = 0 model."""
    from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
    d, sd, m0 = _load(D)
    m = DynamicFieldAether(2 * D, 64, 0.2, D, device="cuda")
    m.load_state_dict(sd)
    B, N = 6, 9
    inp = make_batch(B, N, D, seed=77)
    g = torch.Generator().manual_seed(78)
    masks = (torch.rand(2, B * N, 64, generator=g) >= 0.2).float() / 0.8
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = O.dynamic_field_aether_forward(sdg, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"], N,
                                          dropout_masks=masks)
    torch.nn.functional.mse_loss(want, inp["target"]).backward()
    args = (None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
            inp["charges"].cuda(), N)
    m.train()
    m._dropout_masks = masks
    out = m(*args)
    assert scale_rel_err(out.detach().cpu(), want.detach()) <= TOL
    torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
    for k, p in m.named_parameters():
        if k.endswith("gate_nn.2.bias"):
            continue
        assert scale_rel_err(p.grad.cpu(), sdg[k].grad) <= GTOL, k
    with torch.no_grad():
        assert torch.equal(m(*args), out.detach())                 # train() mode without autograd: the same masks apply
        m.eval()
        plain = O.dynamic_field_aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"], N)
        assert scale_rel_err(m(*args).cpu(), plain) <= TOL


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_input_gradients_vs_oracle_autograd(D, flags):
    """The reference's DynamicFieldAether.forward is differentiable in x / vel / edge_attr_orig (dynamic_field_aether.py:
    79-100): d/dx and d/dvel now also run through the latent field -- the FiLM net's inputs and every node's row in its
    graph's attention pooling (aether_dynamic_field_backward_inputs) -- on top of the frames / features / residual part
    (aether_backward_inputs).  Against the oracle's fp64 autograd, the fp32 oracle's own distance beside it; parameter
    gradients of the same backward unchanged."""
    d, sd, m = _load(D)
    m.flags = flags
    for (B, N, seed) in [(4, 5, 21), (6, 20, 22), (2, 40, 23)]:
        inp = make_batch(B, N, D, seed=seed)

        def oracle(dtype):
            c = lambda t: t.to(dtype) if t.is_floating_point() else t
            leaves = {k: c(inp[k]).clone().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
            out = O.dynamic_field_aether_forward({k: c(v) for k, v in sd.items()}, leaves["x"], leaves["vel"], inp["edges"],
                                                 leaves["edge_attr"], c(inp["charges"]), N)
            torch.nn.functional.mse_loss(out, c(inp["target"])).backward()
            return {k: v.grad for k, v in leaves.items()}

        want, o32 = oracle(torch.float64), oracle(torch.float32)
        m.zero_grad(set_to_none=True)
        leaves = {k: inp[k].cuda().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
        out = m(None, leaves["x"], [e.cuda() for e in inp["edges"]], leaves["vel"], leaves["edge_attr"], inp["charges"].cuda(), N)
        torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
        pg = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        for k, v in leaves.items():
            assert torch.isfinite(v.grad).all(), (B, N, k)
            err, err32 = scale_rel_err(v.grad.cpu(), want[k]), scale_rel_err(o32[k], want[k])
            assert err <= max(GTOL, 4 * err32), (B, N, k, err, err32)
        m.zero_grad(set_to_none=True)                                   # parameters only: the same parameter gradients
        out = m(None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                inp["charges"].cuda(), N)
        torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
        for k, p in m.named_parameters():
            assert torch.equal(p.grad, pg[k]), k


def test_dynamic_field_training_step_reduces_loss():
    """A few optimizer steps through the drop-in module, as the runner's loop does (main.py:200-260); gradients
    accumulate like autograd's; a second backward through the same graph is refused by autograd itself."""
    D = 2
    d, sd, m = _load(D)
    inp = make_batch(16, 20, D, seed=31)
    args = (None, inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
            inp["charges"].cuda(), 20)
    target = inp["target"].cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m(*args), target)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    m.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(m(*args), target).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    torch.nn.functional.mse_loss(m(*args), target).backward()           # accumulates
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-6, atol=1e-12), k
    frozen = [p for n, p in m.named_parameters() if n.startswith("gnn.")]
    for p in frozen:
        p.requires_grad_(False)
    m.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(m(*args), target).backward()
    assert all(p.grad is None for p in frozen)
    assert all(p.grad is not None for n, p in m.named_parameters() if n.startswith("field_net."))


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
def test_dynamic_field_device_rollout(flags):
    """aether_rollout_dynamic_field: 20 steps on the device vs the loop of module calls (same arithmetic up to fma
    contraction) and vs the oracle's rollout protocol with the dynamic field."""
    D, B, N, T = 2, 6, 20, 20
    d, sd, m = _load(D)
    m.flags = flags
    inp = make_batch(B, N, D, seed=41, device="cuda")
    traj = m.rollout(inp["x"], inp["vel"], inp["edges"], inp["charges"], T, 0.5, num_nodes=N)
    assert traj.shape == (T, B * N, D)
    x, v = inp["x"], inp["vel"]
    rows, cols = inp["edges"]
    qprod = inp["charges"][rows] * inp["charges"][cols]
    loop = []
    with torch.no_grad():
        for _ in range(T):
            ea = torch.cat([qprod, (x[rows] - x[cols]).norm(dim=1, keepdim=True)], 1)
            xn = m(None, x, inp["edges"], v, ea, inp["charges"], N)
            v, x = (xn - x) / 0.5, xn
            loop.append(x)
    loop = torch.stack(loop)
    assert scale_rel_err(traj[0].cpu(), loop[0].cpu()) <= 1e-6 and scale_rel_err(traj.cpu(), loop.cpu()) <= TOL
    # oracle, a few steps
    xo, vo = inp["x"].cpu(), inp["vel"].cpu()
    eo = [e.cpu() for e in inp["edges"]]
    qo = inp["charges"].cpu()
    for t in range(3):
        ea = torch.cat([qo[eo[0]] * qo[eo[1]], (xo[eo[0]] - xo[eo[1]]).norm(dim=1, keepdim=True)], 1)
        xn = O.dynamic_field_aether_forward(sd, xo, vo, eo, ea, qo, N)
        vo, xo = (xn - xo) / 0.5, xn
        assert scale_rel_err(traj[t].cpu(), xo) <= TOL, t
    assert m.rollout(inp["x"], inp["vel"], inp["edges"], inp["charges"], 0, num_nodes=N).shape == (0, B * N, D)


def test_dynamic_field_graphed_train_step():
    """GraphedTrainStep (forward + both HIP backward halves + fused AdamW in one hipGraph) on the dynamic-field model:
    same parameters as the eager loop."""
    from aether_amd.training import GraphedTrainStep
    D, B, N = 2, 16, 20
    d, sd, m1 = _load(D)
    _, _, m2 = _load(D)
    b = make_batch(B, N, D, seed=91, device="cuda")
    args = (None, b["x"], b["edges"], b["vel"], b["edge_attr"], b["charges"], N)
    step = GraphedTrainStep(m1, args, b["target"], lr=1e-3, warmup=1)
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-12, capturable=True, fused=True)
    for k in range(4):
        if k > 0:
            lg = float(step.step().detach())
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m2(*args), b["target"])
        loss.backward()
        opt.step()
        if k > 0:
            assert abs(lg - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    for (k, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert scale_rel_err(p.detach().cpu(), q.detach().cpu()) <= 1e-5, k


# ---------------------------------------------------------------- any hidden_size (round 4)
@pytest.mark.parametrize("D,H", [(2, 32), (3, 96), (2, 128), (3, 192)])
def test_dynamic_field_any_hidden_size_vs_oracle(D, H):
    """DynamicFieldAether(hidden_size = --nf) (experiments/lorentz/main.py:42-43,148-149; dynamic_field_aether.py:51-100):
    the GNN runs at the kernel width (64, or the next multiple of 64 on csrc/wide.h), other widths zero-padded.  Forward,
    dL/dfield, every parameter gradient, the input gradients and the device rollout against the oracle."""
    torch.manual_seed(21)
    m = DynamicFieldAether(2 * D, H, 0.0, D, device="cuda")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert sd["gnn.layer_2.message_fn.0.weight"].shape == (H, 3 * H) and sd["gnn.layer_1.update_fn.0.weight"].shape == (2 * H, H)
    B, N = 5, 8
    inp = make_batch(B, N, D, seed=31)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hx, hv, he = (inp[k].clone().requires_grad_(True) for k in ("x", "vel", "edge_attr"))
    field = O.dynamic_field(sdg, hx, hv, inp["charges"], N)
    field.retain_grad()
    want = O.aether_forward(sdg, hx, hv, inp["edges"], he, inp["charges"], field=field)
    torch.nn.functional.mse_loss(want, inp["target"]).backward()
    edges = [e.cuda() for e in inp["edges"]]
    with torch.no_grad():
        out0 = m(None, inp["x"].cuda(), edges, inp["vel"].cuda(), inp["edge_attr"].cuda(), inp["charges"].cuda(), N)
    assert scale_rel_err(out0.cpu(), want.detach()) <= TOL
    m.zero_grad(set_to_none=True)
    xg, vg, eg = (inp[k].cuda().requires_grad_(True) for k in ("x", "vel", "edge_attr"))
    out = m(None, xg, edges, vg, eg, inp["charges"].cuda(), N)
    assert scale_rel_err(out.detach().cpu(), want.detach()) <= TOL
    torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
    assert scale_rel_err(m.last_grad_field.cpu(), field.grad) <= GTOL
    for k, p in m.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and torch.isfinite(p.grad).all(), k
        if k.endswith("gate_nn.2.bias"):
            continue
        ref = sdg[k].grad
        assert float((p.grad.cpu() - ref).abs().max()) <= GTOL * max(float(ref.abs().max()), 1e-3), k
    for got_g, ref_g, name in ((xg.grad, hx.grad, "x"), (vg.grad, hv.grad, "vel"), (eg.grad, he.grad, "edge_attr")):
        assert float((got_g.cpu() - ref_g).abs().max()) <= GTOL * max(float(ref_g.abs().max()), 1e-3), name
    # device rollout: the latent field recomputed from the current state every step
    traj = m.rollout(inp["x"].cuda(), inp["vel"].cuda(), edges, inp["charges"].cuda(), 3, num_nodes=N).cpu()
    rows, cols = inp["edges"]
    qprod = inp["charges"][rows] * inp["charges"][cols]
    x, v = inp["x"], inp["vel"]
    with torch.no_grad():
        for t in range(3):
            dist = torch.sqrt(torch.sum((x[rows] - x[cols]) ** 2, 1)).unsqueeze(1)
            xn = O.dynamic_field_aether_forward(sd, x, v, inp["edges"], torch.cat([qprod, dist], 1), inp["charges"], N)
            v, x = (xn - x) / 1.0, xn
            assert scale_rel_err(traj[t], x) <= TOL, t
