"""Parameter gradients of the HIP backward vs the reference's (golden) and the oracle's autograd."""
import pytest
import torch

import os

import numpy as np

from conftest import GOLDEN, GRAD_CASES, load_case, load_state_dict, scale_rel_err
from aether_amd import _lib
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
DEFAULT_EDGE_ACC = 3          # aether_set_option("edge_acc"): the library's default
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
    try:
        _lib.check(lib.aether_set_option(b"fused_backward", 0), "set_option")      # the layer-by-layer kernels (backward.h)
        _lib.check(lib.aether_set_option(b"edge_acc", 0), "set_option")            # (edge_acc.h sums in another order: below)
        _, deferred = _loss_backward(m, inp)
        _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 0), "set_option")
        _, per_layer = _loss_backward(m, inp)
    finally:
        _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 1 << 20), "set_option")
        _lib.check(lib.aether_set_option(b"edge_acc", DEFAULT_EDGE_ACC), "set_option")
        _lib.check(lib.aether_set_option(b"fused_backward", 1), "set_option")
    for k in deferred:
        assert torch.equal(deferred[k], per_layer[k]), k


@pytest.mark.parametrize("D", [2, 3])
def test_edge_level_weight_gradients_accumulated_in_the_edge_kernel(D):
    """Large graphs (E > outer_defer_max_edges; forced here): kb_edge_acc keeps dpre2, h, G and e_prev of a tile on chip
    and accumulates dW2, dW_e (layer 1: dW1), db2, db1 in registers (edge_acc.h) instead of writing the rows for k_outer.
    All 47 gradients against the oracle's autograd and against the row-tensor path, twice (bit-stable); ragged tile
    counts (N = 37: 1,332 edges per graph, not a multiple of 16) and fewer tiles than waves."""
    lib = _lib.load()
    sd = load_state_dict(D)
    for (B, N, seed) in [(7, 37, 81), (1, 6, 82), (3, 70, 83)]:
        inp = make_batch(B, N, D, seed=seed)
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out = O.aether_forward(sdg, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
        torch.nn.functional.mse_loss(out, inp["target"]).backward()
        m = _model(D, "streamed")
        res = {}
        try:
            _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 0), "set_option")
            # 3 / 2: kb_edge_acc8 (round 4, two waves per SIMD: two workgroups of four / one of eight), 1: kb_edge_acc, 0: row tensors
            for acc in (3, 2, 1, 0, 1, 2, 3):
                _lib.check(lib.aether_set_option(b"edge_acc", acc), "set_option")
                _, g = _loss_backward(m, inp)
                if acc in res:
                    for k in g:
                        assert torch.equal(g[k], res[acc][k]), (acc, k)   # same bits on a second run
                res[acc] = g
        finally:
            _lib.check(lib.aether_set_option(b"outer_defer_max_edges", 1 << 20), "set_option")
            _lib.check(lib.aether_set_option(b"edge_acc", DEFAULT_EDGE_ACC), "set_option")
        for acc in (1, 2, 3):
            for k in res[acc]:
                assert torch.isfinite(res[acc][k]).all(), k
                assert scale_rel_err(res[acc][k], sdg[k].grad) <= GTOL, (acc, B, N, k)
                assert scale_rel_err(res[acc][k], res[0][k]) <= GTOL, (acc, B, N, k)


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


@pytest.mark.parametrize("as_view", [True, False])
def test_gradient_accumulation_like_autograd(as_view):
    """A second backward without zero_grad adds (with .grad aliasing the flat buffer the kernels used to
    overwrite it), and the module applied twice in one autograd graph gets the sum of both applications."""
    D = 2
    m = _model(D, "fused")
    m.grad_as_view = as_view
    a, b = make_batch(6, 7, D, seed=81), make_batch(6, 7, D, seed=82)

    def loss(inp):
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
        e = [t.cuda() for t in inp["edges"]]
        return torch.nn.functional.mse_loss(m(d["h"], d["x"], e, d["vel"], d["edge_attr"], d["charges"]), d["target"])

    def grads():
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    m.zero_grad(set_to_none=True)
    loss(a).backward()
    ga = grads()
    m.zero_grad(set_to_none=True)
    loss(b).backward()
    gb = grads()
    # two backwards, no zero_grad in between
    m.zero_grad(set_to_none=True)
    loss(a).backward()
    loss(b).backward()
    g2 = grads()
    # zero_grad(set_to_none=False) keeps the aliasing views alive and zeroes them
    m.zero_grad(set_to_none=False)
    loss(a).backward()
    g3 = grads()
    # both applications in ONE graph
    m.zero_grad(set_to_none=True)
    (loss(a) + loss(b)).backward()
    g4 = grads()
    for k in ga:
        want = ga[k] + gb[k]
        tol = 1e-6 * float(want.abs().max()) + 1e-12
        assert float((g2[k] - want).abs().max()) <= tol, k
        assert float((g4[k] - want).abs().max()) <= tol, k
        assert torch.equal(g3[k], ga[k]), k
    m.grad_as_view = True


def _input_gradients(m, inp, which=("x", "vel", "edge_attr")):
    dev = "cuda"
    leaves = {k: inp[k].to(dev).clone().requires_grad_(k in which) for k in ("x", "vel", "edge_attr")}
    m.zero_grad(set_to_none=True)
    out = m(inp["h"].to(dev), leaves["x"], [e.to(dev) for e in inp["edges"]], leaves["vel"], leaves["edge_attr"],
            inp["charges"].to(dev))
    torch.nn.functional.mse_loss(out, inp["target"].to(dev)).backward()
    torch.cuda.synchronize()
    return ({k: v.grad.detach().cpu() for k, v in leaves.items() if k in which},
            {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None})


def _oracle_input_gradients(sd, inp, dtype):
    c = lambda t: t.to(dtype) if t.is_floating_point() else t
    leaves = {k: c(inp[k]).clone().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
    out = O.aether_forward({k: c(v) for k, v in sd.items()}, leaves["x"], leaves["vel"], inp["edges"], leaves["edge_attr"],
                           c(inp["charges"]))
    torch.nn.functional.mse_loss(out, c(inp["target"])).backward()
    return {k: v.grad for k, v in leaves.items()}


@pytest.mark.parametrize("D", [2, 3])
def test_input_gradients_match_oracle_autograd(D):
    """The reference's forward is differentiable in x / vel / edge_attr_orig (aether.py:169-186); so is the HIP step
    (aether_backward_inputs, input_grad.h).  Against the oracle's autograd in fp64 (the fp32 oracle's own distance to fp64
    printed beside it: the angle derivatives 1/(x^2+y^2), 1/sqrt(1-c^2) make these gradients less well conditioned than the
    parameters'), on both dispatch paths, a degree > 64 graph and a narrow (hidden 32) model; parameter gradients of the
    same backward are unchanged by asking for the inputs' as well."""
    sd = load_state_dict(D)
    for (B, N, seed, path) in [(5, 7, 61, "fused"), (3, 40, 62, "streamed"), (130, 20, 63, "fused"), (2, 70, 64, "streamed"),
                               (16, 20, 65, "streamed")]:
        inp = make_batch(B, N, D, seed=seed)
        want = _oracle_input_gradients(sd, inp, torch.float64)
        o32 = _oracle_input_gradients(sd, inp, torch.float32)
        m = _model(D, path)
        if path == "streamed" and N >= 40:
            m.flags = 0
        got, pgrads = _input_gradients(m, inp)
        _, pref = _loss_backward(m, inp)
        for k in pref:
            assert torch.equal(pgrads[k], pref[k]), k
        for k, g in got.items():
            assert torch.isfinite(g).all(), (B, N, k)
            err, err32 = scale_rel_err(g, want[k]), scale_rel_err(o32[k], want[k])
            print(f"[input gradients] D={D} B={B} N={N} {path} d/d{k}: HIP {err:.2e}, oracle fp32 {err32:.2e}")
            assert err <= max(GTOL, 4 * err32), (B, N, k, err, err32)
        only_v, _ = _input_gradients(m, inp, which=("vel",))          # any subset
        assert torch.equal(only_v["vel"], got["vel"])


@pytest.mark.parametrize("D", [2, 3])
def test_input_gradients_on_random_multigraphs(D):
    """Self loops, duplicate edges, nodes without in- or out-edges, components of 1..40 nodes (the batch of
    test_gradients_on_random_multigraphs): d/dx, d/dvel, d/dedge_attr of both dispatch paths against the oracle's fp64
    autograd.  (A self loop has rel = 0: distance and bearing sit at their clamps, where both sides give zero slope.)"""
    from test_gpu_configs import _random_multigraph_batch
    sd = load_state_dict(D)
    inp = _random_multigraph_batch(31, D)
    g = torch.Generator().manual_seed(31)
    inp["target"] = inp["x"] + 0.1 * torch.randn(inp["x"].shape, generator=g)
    want = _oracle_input_gradients(sd, inp, torch.float64)
    o32 = _oracle_input_gradients(sd, inp, torch.float32)
    for path in ("default", "streamed"):
        m = _model(D, "streamed")
        if path == "default":
            m.flags = 0
        got, _ = _input_gradients(m, inp)
        for k, gk in got.items():
            assert torch.isfinite(gk).all(), (path, k)
            err, err32 = scale_rel_err(gk, want[k]), scale_rel_err(o32[k], want[k])
            assert err <= max(GTOL, 4 * err32), (path, k, err, err32)


@pytest.mark.parametrize("path", ["fused", "streamed"])
@pytest.mark.parametrize("tag", ["dropout", "inputgrad"])
@pytest.mark.parametrize("D", [2, 3])
def test_dropout_step_and_input_gradients_match_the_reference_itself(D, tag, path):
    """The same fixtures against the HIP path: a train()-mode step with the masks the REFERENCE's nn.Dropout layers drew
    (dropout_prob = 0.25), and the reference's own d/dx, d/dvel, d/dedge_attr -- not only the oracle's."""
    d = np.load(os.path.join(GOLDEN, f"case_D{D}_{tag}.npz"))
    inp, ref, _, meta = load_case(f"case_D{D}_{tag}.npz")
    p = float(d["dropout_prob"][0])
    m = Aether(2 * D, 64, p, D, device="cuda")
    m.load_state_dict(load_state_dict(D))
    m.flags = _lib.FLAG_FORCE_FUSED if path == "fused" else _lib.FLAG_FORCE_STREAMED
    m.train()
    if p > 0:
        m._dropout_masks = torch.stack([torch.from_numpy(d["mask1"]), torch.from_numpy(d["mask2"])])
    leaves = {k: inp[k].cuda().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
    out = m(inp["h"].cuda(), leaves["x"], [e.cuda() for e in inp["edges"]], leaves["vel"], leaves["edge_attr"],
            inp["charges"].cuda())
    assert scale_rel_err(out.detach().cpu(), ref["out"]) <= 1e-5
    torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
    for k, q in m.named_parameters():
        assert scale_rel_err(q.grad.cpu(), ref["grad." + k]) <= GTOL, k
    for k, v in leaves.items():
        assert scale_rel_err(v.grad.cpu(), ref["grad_in." + k]) <= 2 * GTOL, k


def test_input_gradients_of_a_narrow_model_and_with_frozen_parameters():
    D = 3
    inp = make_batch(4, 9, D, seed=66)
    torch.manual_seed(5)
    m = Aether(2 * D, 32, 0.0, D, device="cuda")
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    want = _oracle_input_gradients(sd, inp, torch.float64)
    got, _ = _input_gradients(m, inp)
    for k, g in got.items():
        assert scale_rel_err(g, want[k]) <= 4 * GTOL, k
    for p in m.parameters():                      # inputs only: no parameter asks for a gradient
        p.requires_grad_(False)
    frozen, pg = _input_gradients(m, inp)
    assert not pg
    for k in got:
        assert torch.equal(frozen[k], got[k]), k
    with torch.no_grad():                         # without autograd the same call is plain inference
        x = inp["x"].cuda().requires_grad_(True)
        out = m(inp["h"].cuda(), x, [e.cuda() for e in inp["edges"]], inp["vel"].cuda(), inp["edge_attr"].cuda(),
                inp["charges"].cuda())
    assert out.grad_fn is None


@pytest.mark.parametrize("shape", [(128, 20), (16, 20), (300, 5), (7, 9), (40, 3), (3, 12)])
@pytest.mark.parametrize("D", [2, 3])
def test_fused_backward_matches_layer_by_layer_kernels(D, shape):
    """The one-launch GNN backward (fused_bwd.h: split and unsplit workgroups, one to three tiles per wave, several
    graphs per workgroup) against the layer-by-layer kernels of backward.h on the same saved intermediates; both are
    separately held to the reference's gradients above.  Bit-stable on a second run."""
    lib = _lib.load()
    B, N = shape
    inp = make_batch(B, N, D, seed=90 + B)
    m = _model(D, "fused")
    _, g_fused = _loss_backward(m, inp)
    _, g_again = _loss_backward(m, inp)
    try:
        _lib.check(lib.aether_set_option(b"fused_backward", 0), "set_option")
        _, g_layers = _loss_backward(m, inp)
    finally:
        _lib.check(lib.aether_set_option(b"fused_backward", 1), "set_option")
    for k in g_fused:
        assert torch.equal(g_fused[k], g_again[k]), k
        scale = float(g_layers[k].abs().max())
        assert float((g_fused[k] - g_layers[k]).abs().max()) <= 2e-5 * scale + 1e-10, (k, scale)


@pytest.mark.parametrize("streamed", [False, True])
def test_backward_only_flag_leaves_the_gradients_alone(streamed):
    """Training forwards pass AETHER_FLAG_BACKWARD_ONLY (the last layer's messages are not written: only the debug fetch
    reads them).  With the module's own KEEP flag set the forward writes everything; both must give the same bits."""
    from aether_amd import _lib
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    D = 2
    b = make_batch(8, 20, D, seed=21, device="cuda")
    grads = []
    for keep_all in (False, True):
        torch.manual_seed(3)
        m = Aether(2 * D, 64, 0.0, D, device="cuda")
        if streamed:
            m.flags |= _lib.FLAG_FORCE_STREAMED
        if keep_all:
            m.flags |= _lib.FLAG_KEEP_INTERMEDIATES
        out = m(b["h"], b["x"], b["edges"], b["vel"], b["edge_attr"], b["charges"])
        torch.nn.functional.mse_loss(out, b["target"]).backward()
        grads.append((out.detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("D", [2, 3])
def test_streamed_forward_then_fused_backward_on_a_poisoned_workspace(D):
    """aether_backward chooses the fused backward from the graph alone, so a STREAMED forward must
    leave everything that kernel reads -- the split weight images included -- in the workspace.  The workspace handed
    to the forward is filled with NaN bit patterns first; the gradients must equal the layer-by-layer backward's."""
    lib = _lib.load()
    inp = make_batch(6, 20, D, seed=77)
    grads = {}
    for fused_bwd in (1, 0):
        torch.manual_seed(1)
        m = Aether(2 * D, 64, 0.0, D, device="cuda")
        m.flags = _lib.FLAG_FORCE_STREAMED
        n_nodes, n_edges = inp["x"].shape[0], inp["edges"][0].numel()
        m._train_ws = torch.full((m._workspace_bytes(n_nodes, n_edges, True),), 0xFF, dtype=torch.uint8, device="cuda")
        poisoned = m._train_ws.data_ptr()
        _lib.check(lib.aether_set_option(b"fused_backward", fused_bwd), "set_option")
        try:
            _, grads[fused_bwd] = _loss_backward(m, inp)
        finally:
            _lib.check(lib.aether_set_option(b"fused_backward", 1), "set_option")
        assert m._last_ws.data_ptr() == poisoned                 # the forward really ran on the poisoned buffer
    for k in grads[1]:
        assert torch.isfinite(grads[1][k]).all(), k
        assert scale_rel_err(grads[1][k], grads[0][k]) <= GTOL, k
