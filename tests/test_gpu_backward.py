"""Parameter gradients of the HIP backward vs the reference's (golden) and the oracle's autograd."""
import pytest
import torch

from conftest import GRAD_CASES, load_case, load_state_dict, scale_rel_err
from aether_amd import _lib
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
GTOL = 5e-5     # gradients: sums over thousands of edges in a different (fixed) order than autograd


def _model(D, path):
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    m.load_state_dict(load_state_dict(D))
    m.flags = _lib.FLAG_FORCE_FUSED if path == "fused" else _lib.FLAG_FORCE_STREAMED
    return m


def _loss_backward(m, inp):
    dev = "cuda"
    m.zero_grad(set_to_none=True)
    out = m(inp["h"].to(dev), inp["x"].to(dev), [e.to(dev) for e in inp["edges"]], inp["vel"].to(dev),
            inp["edge_attr"].to(dev), inp["charges"].to(dev))
    loss = torch.nn.functional.mse_loss(out, inp["target"].to(dev))
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.detach().cpu() for k, p in m.named_parameters()}


@pytest.mark.parametrize("path", ["fused", "streamed"])
@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("case", GRAD_CASES)
def test_parameter_gradients_match_reference(D, case, path):
    inp, ref, ref64, meta = load_case(f"case_D{D}_{case}.npz")
    m = _model(D, path)
    loss, grads = _loss_backward(m, inp)
    assert abs(loss - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
    worst = 0.0
    for k, g in grads.items():
        assert torch.isfinite(g).all(), k
        err = scale_rel_err(g, ref["grad." + k])
        worst = max(worst, err)
        assert err <= GTOL, (k, err)
    if ref64:       # the reference's own fp64 gradients, where stored
        for k, g in grads.items():
            if "grad." + k in ref64:
                assert scale_rel_err(g, ref64["grad." + k]) <= GTOL, k


@pytest.mark.parametrize("D", [2, 3])
def test_gradients_vs_oracle_autograd_fresh_inputs(D):
    sd = load_state_dict(D)
    # (2, 70): average degree 69 > 64 -> the backward's row sums run as their own kernel
    for (B, N, seed, path) in [(5, 7, 51, "fused"), (3, 40, 52, "streamed"), (130, 20, 53, "fused"),
                               (2, 70, 54, "streamed")]:
        inp = make_batch(B, N, D, seed=seed)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.aether_forward(sdg, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
        torch.nn.functional.mse_loss(out, inp["target"]).backward()
        m = _model(D, path)
        if path == "streamed" and N >= 40:
            m.flags = 0                      # too big for a fused group: default dispatch must cope
        _, grads = _loss_backward(m, inp)
        for k, g in grads.items():
            assert scale_rel_err(g, sdg[k].grad) <= GTOL, (B, N, k)


@pytest.mark.parametrize("seed", [21, 22, 23])
@pytest.mark.parametrize("D", [2, 3])
def test_gradients_on_random_multigraphs(D, seed):
    """Self loops, duplicate edges, nodes without in-edges, components of 1..40 nodes: gradients of both
    dispatch paths against the oracle's autograd."""
    from test_gpu_configs import _random_multigraph_batch
    sd = load_state_dict(D)
    inp = _random_multigraph_batch(seed, D)
    g = torch.Generator().manual_seed(seed)
    inp["target"] = inp["x"] + 0.1 * torch.randn(inp["x"].shape, generator=g)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.aether_forward(sdg, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    torch.nn.functional.mse_loss(out, inp["target"]).backward()
    for path in ("default", "streamed"):
        m = _model(D, "streamed")
        if path == "default":
            m.flags = 0
        _, grads = _loss_backward(m, inp)
        for k, gk in grads.items():
            assert torch.isfinite(gk).all(), (path, k)
            assert scale_rel_err(gk, sdg[k].grad) <= GTOL, (path, k)


@pytest.mark.parametrize("D", [2, 3])
def test_per_layer_weight_gradient_launches_match_deferred(D):
    """Above `outer_defer_max_edges` the backward multiplies each layer's weight gradients before the
    next layer overwrites their operands; below it everything is deferred to one launch at the end.
    Both orders run the same kernels on the same rows: identical bits."""
    lib = _lib.load()
    inp = make_batch(9, 12, D, seed=71)
    m = _model(D, "fused")
    _, deferred = _loss_backward(m, inp)
    try:
        _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 0), "set_option")
        _, per_layer = _loss_backward(m, inp)
    finally:
        _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 1 << 20), "set_option")
    for k in deferred:
        assert torch.equal(deferred[k], per_layer[k]), k


def test_backward_is_deterministic_and_optimizer_step_runs():
    D = 2
    m = _model(D, "fused")
    inp = make_batch(16, 20, D, seed=61)
    _, g1 = _loss_backward(m, inp)
    _, g2 = _loss_backward(m, inp)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-12)   # main.py:86,164
    l0, _ = _loss_backward(m, inp)
    for _ in range(20):
        _loss_backward(m, inp)
        opt.step()
    l1, _ = _loss_backward(m, inp)
    assert l1 < l0
