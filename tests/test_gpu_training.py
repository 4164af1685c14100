"""End-to-end training parity: the runner's loop (forward, MSE loss, backward, Adam; experiments/lorentz/
main.py:247-291) driven through the HIP module vs the same loop on the oracle with torch autograd on the
CPU, from the same initial parameters; plus the error behaviour of the drop-in surface."""
import ctypes as C

import pytest
import torch

from conftest import load_state_dict, scale_rel_err
from aether_amd import _lib
from aether_amd.edges import get_edges
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("D", [2, 3])
def test_five_adam_steps_track_the_oracle(D):
    steps, lr = 5, 5e-4                                  # the runner's learning rate (main.py:33)
    batches = [make_batch(8, 20, D, seed=200 + t) for t in range(steps)]
    # --- oracle side: functional parameters + torch.optim.Adam on the CPU
    sd = {k: v.clone().requires_grad_(True) for k, v in load_state_dict(D).items()}
    opt_o = torch.optim.Adam(list(sd.values()), lr=lr)
    losses_o = []
    for b in batches:
        opt_o.zero_grad()
        out = O.aether_forward(sd, b["x"], b["vel"], b["edges"], b["edge_attr"], b["charges"])
        loss = torch.nn.functional.mse_loss(out, b["target"])
        loss.backward()
        opt_o.step()
        losses_o.append(float(loss.detach()))
    # --- HIP module
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    m.load_state_dict(load_state_dict(D))
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    losses = []
    for b in batches:
        d = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items() if k != "edges"}
        edges = [e.cuda() for e in b["edges"]]
        opt.zero_grad()
        out = m(d["h"], d["x"], edges, d["vel"], d["edge_attr"], d["charges"])
        loss = torch.nn.functional.mse_loss(out, d["target"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    for a, b_ in zip(losses, losses_o):
        assert abs(a - b_) <= 2e-5 * abs(b_), (losses, losses_o)
    # Adam divides by sqrt(v): parameters whose gradient is ~0 amplify round-off, so compare the update
    # of every tensor at the scale of the largest update (lr per step)
    init = load_state_dict(D)
    for k, p in m.state_dict().items():
        upd, upd_o = p.cpu() - init[k], sd[k].detach() - init[k]
        assert float((upd - upd_o).abs().max()) <= 0.05 * steps * lr, k
        assert scale_rel_err(p.cpu(), sd[k].detach()) <= 1e-3, k


def test_drop_in_error_behaviour():
    D = 2
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    inp = make_batch(2, 5, D, seed=1, device="cuda")
    args = lambda **kw: [kw.get(k, inp[k]) for k in ("h", "x", "edges", "vel", "edge_attr", "charges")]
    with pytest.raises(_lib.AetherHipError, match="no CPU fallback"):
        m(*args(x=inp["x"].cpu()))
    with pytest.raises(TypeError):
        m(*args(edges=[e.int() for e in inp["edges"]]))
    with pytest.raises(ValueError):
        m(*args(vel=inp["vel"][:-1]))
    bad = [inp["edges"][0].clone(), inp["edges"][1].clone()]
    bad[1][3] = inp["x"].shape[0]                       # receiver out of range
    with pytest.raises(_lib.AetherHipError, match="outside"):
        m(*args(edges=bad))
    for ctor in (lambda: Aether(4, 0, 0.0, 2), lambda: Aether(4, 6, 0.0, 2), lambda: Aether(4, 64, 1.0, 2), lambda: Aether(4, 64, 0.0, 4)):
        with pytest.raises(ValueError):
            ctor()
    # C ABI: a workspace that is too small is refused, nothing is launched
    lib = _lib.load()
    send, recv = get_edges(2, 5, device="cuda")
    graph, info = m.prepare_graph((send, recv), 10)
    need = lib.aether_workspace_bytes(10, 40, D, 0)
    ws = torch.empty(need - 256, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(inp["x"])
    rc = lib.aether_forward(C.byref(m._param_struct()), D, 10, 40, inp["x"].data_ptr(), inp["vel"].data_ptr(),
                            inp["charges"].data_ptr(), inp["edge_attr"].data_ptr(), graph.data_ptr(),
                            C.byref(info), ws.data_ptr(), ws.numel(), out.data_ptr(), 0,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == -4 and b"workspace" in lib.aether_last_error(), (rc, lib.aether_last_error())
    rc = lib.aether_forward(C.byref(m._param_struct()), 4, 10, 40, inp["x"].data_ptr(), inp["vel"].data_ptr(),
                            inp["charges"].data_ptr(), inp["edge_attr"].data_ptr(), graph.data_ptr(),
                            C.byref(info), ws.data_ptr(), ws.numel(), out.data_ptr(), 0,
                            torch.cuda.current_stream().cuda_stream)
    assert rc == -1
    # the width-generic entry points accept multiples of 64 only (other widths are zero-padded by the module)
    assert lib.aether_workspace_bytes_h(10, 40, D, 100, 0) == 0 and lib.aether_workspace_bytes_h(10, 40, D, 128, 0) > 0
    assert lib.aether_workspace_bytes_h(10, 40, D, 64, 0) == need
    rc = lib.aether_forward_h(C.byref(m._param_struct()), D, 100, 10, 40, inp["x"].data_ptr(), inp["vel"].data_ptr(),
                              inp["charges"].data_ptr(), None, inp["edge_attr"].data_ptr(), graph.data_ptr(),
                              C.byref(info), ws.data_ptr(), ws.numel(), out.data_ptr(), 0,
                              torch.cuda.current_stream().cuda_stream)
    assert rc == -1 and b"multiple of 64" in lib.aether_last_error()


def test_graphed_train_step_matches_eager():
    """aether_amd.training.GraphedTrainStep: forward + HIP backward + fused AdamW as one hipGraph replay gives the
    same parameters as the same steps launched eagerly (same kernels), also after a new batch is copied in."""
    from aether_amd.training import GraphedTrainStep
    D, B, N = 2, 16, 20
    batches = [make_batch(B, N, D, seed=70 + k, device="cuda") for k in range(3)]
    args = lambda b: (b["h"], b["x"], b["edges"], b["vel"], b["edge_attr"], b["charges"])

    def fresh():
        m = Aether(2 * D, 64, 0.0, D, device="cuda")
        m.load_state_dict(load_state_dict(D))
        return m

    m1 = fresh()
    step = GraphedTrainStep(m1, args(batches[0]), batches[0]["target"], lr=1e-3, warmup=1)
    m2 = fresh()
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-12, capturable=True, fused=True)

    def eager(b):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m2(*args(b)), b["target"])
        loss.backward()
        opt.step()
        return float(loss.detach())

    eager(batches[0])                                        # the helper's one warm-up step
    losses = []
    for b in (batches[0], batches[1], batches[2], batches[1]):
        lg = float(step.step(args(b), b["target"]).detach())
        le = eager(b)
        losses.append(lg)
        assert abs(lg - le) <= 1e-5 * abs(le)
    for (k, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert scale_rel_err(p.detach().cpu(), q.detach().cpu()) <= 1e-5, k
    assert all(l == l and abs(l) < 1e30 for l in losses)


def test_eval_after_graph_replays_sees_the_updated_weights():
    """A hipGraph replay rewrites the parameters without bumping their version counters; the module decides by those
    counters whether the split weight images in its workspace are current.  GraphedTrainStep bumps them after a replay:
    an inference call between / after replayed training steps must use the new weights."""
    from aether_amd.training import GraphedTrainStep
    from oracle import aether_oracle as O
    D = 2
    torch.manual_seed(5)
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    host = make_batch(16, 20, D, seed=9)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in host.items()}
    inp["edges"] = [e.cuda() for e in host["edges"]]
    call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    m.eval()
    with torch.no_grad():
        call()                                              # leaves prepared images + their key behind
    m.train()
    step = GraphedTrainStep(m, [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]], inp["target"],
                            lr=1e-3, weight_decay=1e-12, warmup=1)
    for _ in range(2):
        m.eval()
        with torch.no_grad():
            call()
        m.train()
        for _ in range(3):
            step.step()
        m.eval()
        with torch.no_grad():
            got = call().cpu()
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        with torch.no_grad():
            want = O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
        assert scale_rel_err(got, want) <= 1e-5


@pytest.mark.parametrize("H", [64, 32])
def test_eval_after_a_fused_optimizer_step_sees_the_updated_weights(H):
    """torch's fused AdamW writes the parameters without bumping their version counters (the module's cue that weight
    images / the narrow model's padded engine are stale): a training forward marks them stale itself.  Eager loop,
    inference calls in between; the second training step must also run on the updated weights."""
    D = 2
    torch.manual_seed(5)
    m = Aether(2 * D, H, 0.0, D, device="cuda")
    host = make_batch(16, 20, D, seed=9)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in host.items()}
    inp["edges"] = [e.cuda() for e in host["edges"]]
    call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    oracle = lambda sd: O.aether_forward(sd, host["x"], host["vel"], host["edges"], host["edge_attr"], host["charges"])
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=1e-12, fused=True)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    opt_o = torch.optim.AdamW(list(sd.values()), lr=1e-2, weight_decay=1e-12)
    with torch.no_grad():
        before = call().cpu()
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(call(), inp["target"]).backward()
        opt.step()
        opt_o.zero_grad()
        torch.nn.functional.mse_loss(oracle(sd), host["target"]).backward()
        opt_o.step()
        with torch.no_grad():
            got = call().cpu()
            want = oracle({k: v.detach().cpu() for k, v in m.state_dict().items()})
        assert scale_rel_err(got, want) <= 1e-5
        assert scale_rel_err(before, want) > 1e-4                    # the step did move the output
    # two steps on stale weights would have ended elsewhere: the oracle's own two AdamW steps agree
    for k, v in m.state_dict().items():
        assert scale_rel_err(v.detach().cpu(), sd[k].detach()) <= 2e-3, k


def test_eval_between_backward_and_fused_adamw_step_does_not_leave_stale_weight_images():
    """train forward, backward, a validation forward (which prepares and marks weight images fresh), then
    FusedAdamW.step(), then another validation forward.  The library's optimizer writes the parameters from a kernel; it
    has to bump their version counters itself, or the second validation forward reuses the pre-step images."""
    from aether_amd.optim import FusedAdamW
    D = 2
    torch.manual_seed(5)
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    host = make_batch(16, 20, D, seed=9)
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in host.items()}
    inp["edges"] = [e.cuda() for e in host["edges"]]
    call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    opt = FusedAdamW(m.parameters(), lr=1e-2)
    versions = [p._version for p in m.parameters()]
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(call(), inp["target"]).backward()
    with torch.no_grad():
        before = call().clone()                                      # validation between backward and step
        again = call()
        assert torch.equal(before, again)
    opt.step()
    assert all(p._version > v for p, v in zip(m.parameters(), versions))
    with torch.no_grad():
        got = call().cpu()
        want = O.aether_forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, host["x"], host["vel"],
                                host["edges"], host["edge_attr"], host["charges"])
    assert scale_rel_err(got, want) <= 1e-5
    assert scale_rel_err(before.cpu(), want) > 1e-4                  # the step did move the output


@pytest.mark.parametrize("n", [1, 7, 5120, 300001])
def test_mse_loss_grad_matches_torch(n):
    """aether_mse_loss_grad (nn.MSELoss + the seed of its backward, main.py:86,289-290) vs torch autograd in fp64."""
    from aether_amd.optim import mse_loss_grad
    g = torch.Generator().manual_seed(n)
    pred = torch.randn(n, generator=g).cuda().view(-1, 1) if n > 1 else torch.randn(1, generator=g).cuda()
    tgt = torch.randn(pred.shape, generator=g).cuda()
    loss, grad = mse_loss_grad(pred, tgt)
    loss2, grad2 = mse_loss_grad(pred, tgt)                       # the scratch counter re-armed itself; fixed-order sum
    assert torch.equal(loss, loss2) and torch.equal(grad, grad2)
    p64 = pred.double().requires_grad_(True)
    want = torch.nn.functional.mse_loss(p64, tgt.double())
    want.backward()
    assert abs(float(loss) - float(want.detach())) <= 2e-6 * abs(float(want.detach()))
    assert scale_rel_err(grad.cpu(), p64.grad.float().cpu()) <= 1e-6


def test_fused_adamw_tracks_torch_adamw():
    """aether_adamw_step vs torch.optim.AdamW over 70 tensors (two launches per step) of ragged sizes, 12 steps, with a
    learning-rate change on the way; the shared step counter sits under torch's state key."""
    from aether_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 192), (64,), (3, 16), (1,), (1025,), (128, 64)] + [(5, k + 1) for k in range(64)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    opt_m, opt_r = FusedAdamW(mine, **kw), torch.optim.AdamW(ref, **kw)
    for it in range(12):
        if it == 6:
            for o in (opt_m, opt_r):
                o.param_groups[0]["lr"] = 1e-3
        for a, b in zip(mine, ref):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (it % 3 - 1))
            a.grad, b.grad = gr.clone(), gr.clone()
        opt_m.step()
        opt_r.step()
    assert opt_m.steps_taken() == 12 and float(opt_m.state[mine[3]]["step"]) == 12.0
    for a, b in zip(mine, ref):
        assert scale_rel_err(a.detach().cpu(), b.detach().cpu()) <= 2e-6, tuple(a.shape)
        assert scale_rel_err(opt_m.state[a]["exp_avg_sq"].cpu(), opt_r.state[b]["exp_avg_sq"].cpu()) <= 2e-6
    # state_dict round trip keeps counting from 12
    opt_2 = FusedAdamW(mine, **kw)
    opt_2.load_state_dict(opt_m.state_dict())
    for a, b in zip(mine, ref):
        gr = torch.randn(a.shape, generator=g).cuda()
        a.grad, b.grad = gr.clone(), gr.clone()
    opt_2.param_groups[0]["lr"] = 1e-3
    opt_2.step()
    opt_r.step()
    assert opt_2.steps_taken() == 13
    for a, b in zip(mine, ref):
        assert scale_rel_err(a.detach().cpu(), b.detach().cpu()) <= 2e-6


def test_graphed_step_with_library_loss_and_optimizer_matches_the_torch_ones():
    """GraphedTrainStep(optimizer="aether", MSE in one launch) vs GraphedTrainStep(optimizer="torch", autograd MSE): same
    trajectory over 5 replays; a learning rate set between replays reaches the captured optimizer launch (lr = 0 stops
    the weights)."""
    from aether_amd.training import GraphedTrainStep
    D = 2
    inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in make_batch(16, 20, D, seed=4).items()}
    inp["edges"] = [e.cuda() for e in inp["edges"]]
    args = [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]]
    res = {}
    for kind in ("aether", "torch"):
        torch.manual_seed(21)
        m = Aether(2 * D, 64, 0.0, D, device="cuda")
        start = {k: v.detach().clone() for k, v in m.state_dict().items()}
        step = GraphedTrainStep(m, args, inp["target"], lr=1e-3, weight_decay=1e-12, warmup=1, optimizer=kind,
                                loss_fn=None if kind == "aether" else torch.nn.functional.mse_loss)
        m.load_state_dict(start)
        for st in step.optimizer.state.values():
            for val in st.values():
                if torch.is_tensor(val):
                    val.zero_()
        losses = [float(step.step().item()) for _ in range(5)]
        res[kind] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
        if kind == "aether":
            assert step.optimizer.steps_taken() == 5
            step.optimizer.param_groups[0]["lr"] = 0.0
            step.step()
            assert step.optimizer.steps_taken() == 6
            assert all(torch.equal(v.detach().cpu(), res[kind][1][k]) for k, v in m.state_dict().items())
    for a, b in zip(res["aether"][0], res["torch"][0]):
        assert abs(a - b) <= 1e-5 * abs(b)
    for k in res["torch"][1]:
        assert scale_rel_err(res["aether"][1][k], res["torch"][1][k]) <= 1e-4, k


def test_dropout_is_identity_in_eval_and_active_in_training():
    """out_mlp's nn.Dropout (locs.py:160-168): identity in eval(), so a model built with dropout_prob > 0 gives the
    p = 0 result there; in train() mode -- with or without autograd, as nn.Dropout keys on the module's mode -- the two
    masks are drawn per call: scales 0 or 1 / (1 - p) with the right frequency, different from call to call."""
    D = 2
    inp = make_batch(4, 20, D, seed=3)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    dev["edges"] = [e.cuda() for e in inp["edges"]]
    torch.manual_seed(1)
    m0 = Aether(2 * D, 64, 0.0, D, device="cuda")
    torch.manual_seed(1)
    m1 = Aether(2 * D, 64, 0.3, D, device="cuda")
    call = lambda m: m(dev["h"], dev["x"], dev["edges"], dev["vel"], dev["edge_attr"], dev["charges"])
    m0.eval(); m1.eval()
    with torch.no_grad():
        ref = call(m0)
        assert torch.equal(ref, call(m1))
    m1.train()
    a = call(m1)
    lib = _lib.load()
    n_nodes, n_edges = dev["x"].shape[0], dev["edges"][0].numel()
    off = lib.aether_dropout_mask_offset(n_nodes, n_edges, D)
    masks = m1._last_ws[off:off + 2 * n_nodes * 64 * 4].view(torch.float32).view(2, n_nodes, 64).clone()
    vals = torch.unique(masks)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.7) < 1e-6
    assert abs(float((masks == 0).float().mean()) - 0.3) < 0.02                 # 10,240 draws
    with torch.no_grad():
        b = call(m1)
    assert not torch.equal(a.detach(), b) and not torch.equal(b, ref)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


@pytest.mark.parametrize("path", ["fused", "streamed"])
@pytest.mark.parametrize("D", [2, 3])
def test_dropout_training_step_matches_oracle_with_the_same_masks(D, path):
    """Forward, all 47 parameter gradients and the input gradients of a train()-mode step with dropout_prob = 0.25 against
    the oracle's autograd with the SAME two masks (the module takes explicit masks through a test hook; otherwise it draws
    them with torch's bernoulli_)."""
    sd = load_state_dict(D)
    for (B, N, seed) in [(6, 9, 81), (3, 40, 82)]:
        inp = make_batch(B, N, D, seed=seed)
        g = torch.Generator().manual_seed(seed)
        masks = (torch.rand(2, B * N, 64, generator=g) >= 0.25).float() / 0.75
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xg, vg = inp["x"].clone().requires_grad_(True), inp["vel"].clone().requires_grad_(True)
        want = O.aether_forward(sdg, xg, vg, inp["edges"], inp["edge_attr"], inp["charges"], dropout_masks=masks)
        torch.nn.functional.mse_loss(want, inp["target"]).backward()
        m = Aether(2 * D, 64, 0.25, D, device="cuda")
        m.load_state_dict(sd)
        m.flags = _lib.FLAG_FORCE_FUSED if path == "fused" and N < 40 else (_lib.FLAG_FORCE_STREAMED if path == "streamed" else 0)
        m.train()
        m._dropout_masks = masks
        x, v = inp["x"].cuda().requires_grad_(True), inp["vel"].cuda().requires_grad_(True)
        out = m(inp["h"].cuda(), x, [e.cuda() for e in inp["edges"]], v, inp["edge_attr"].cuda(), inp["charges"].cuda())
        assert scale_rel_err(out.detach().cpu(), want.detach()) <= 1e-5, (B, N)
        torch.nn.functional.mse_loss(out, inp["target"].cuda()).backward()
        for k, p in m.named_parameters():
            assert scale_rel_err(p.grad.cpu(), sdg[k].grad) <= 5e-5, (B, N, k)
        assert scale_rel_err(x.grad.cpu(), xg.grad) <= 5e-5 and scale_rel_err(v.grad.cpu(), vg.grad) <= 1e-4
        # the same model in eval(): no masks, the p = 0 result
        m.eval()
        with torch.no_grad():
            ev = m(inp["h"].cuda(), inp["x"].cuda(), [e.cuda() for e in inp["edges"]], inp["vel"].cuda(),
                   inp["edge_attr"].cuda(), inp["charges"].cuda())
        plain = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
        assert scale_rel_err(ev.cpu(), plain) <= 1e-5


def test_dropout_with_a_narrow_model():
    """hidden_size 32 runs zero-padded on the 64-wide engine: with dropout_prob > 0 in train() mode the engine draws the
    masks (the padded channels stay zero whatever their mask says); gradients are finite, eval() equals the p = 0 model."""
    D = 2
    inp = make_batch(4, 9, D, seed=5)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in inp.items()}
    dev["edges"] = [e.cuda() for e in inp["edges"]]
    torch.manual_seed(2)
    m0 = Aether(2 * D, 32, 0.0, D, device="cuda")
    torch.manual_seed(2)
    m1 = Aether(2 * D, 32, 0.4, D, device="cuda")
    call = lambda m: m(dev["h"], dev["x"], dev["edges"], dev["vel"], dev["edge_attr"], dev["charges"])
    m0.eval(); m1.eval()
    with torch.no_grad():
        ref = call(m0)
        assert torch.equal(ref, call(m1))
    m1.train()
    out = call(m1)
    assert not torch.equal(out.detach(), ref)
    torch.nn.functional.mse_loss(out, dev["target"]).backward()
    for k, p in m1.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


def test_dropout_inside_a_captured_training_step():
    """GraphedTrainStep with dropout_prob > 0: the mask draw is part of the captured graph (torch's graph-safe Philox
    offsets), so every replay trains with fresh masks; the loss stays finite and goes down."""
    from aether_amd.training import GraphedTrainStep
    D = 2
    inp = make_batch(8, 20, D, seed=91)
    dev = "cuda"
    torch.manual_seed(3)
    m = Aether(2 * D, 64, 0.1, D, device=dev).train()
    args = [inp["h"].to(dev), inp["x"].to(dev), [e.to(dev) for e in inp["edges"]], inp["vel"].to(dev), inp["edge_attr"].to(dev),
            inp["charges"].to(dev)]
    gs = GraphedTrainStep(m, args, inp["target"].to(dev), lr=1e-3, weight_decay=1e-12, warmup=1)
    lib = _lib.load()
    off = lib.aether_dropout_mask_offset(inp["x"].shape[0], inp["edges"][0].numel(), D)
    losses, seen = [], []
    for _ in range(30):
        losses.append(float(gs.step(args, inp["target"].to(dev))))
        seen.append(m._last_ws[off:off + 4096].clone())
    gs.check()
    assert all(l == l and l < 1e6 for l in losses)
    assert sum(losses[-5:]) < sum(losses[:5])
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])


