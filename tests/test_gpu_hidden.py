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
