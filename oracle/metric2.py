"""Metric 2 -- 20-step rollout MSE against simulated ground truth -- as a parity statement.  TEST INFRASTRUCTURE: imported
only by tests/ and bench.py's cpu_baseline leg, never by the product (aether_amd/).

Protocol: experiments/electrostatic/evaluate.py:33-70 (burn-in 29 frames, predict 20, per-step MSE over (sample,
particle, feature) on un-normalised values; the runner prints steps 1 / 10 / 20, experiments/electrostatic/main.py:162-168)
applied to the state2state module with the rollout SURVEY.md 8(d) defines (x_{t+1} = Aether(x_t, v_t),
v_{t+1} = x_{t+1} - x_t).  Trajectories come from the build's electrostatic simulator (N charged balls + 20 static
field charges, 5,000 leap-frog steps sampled every 100 -> 49 frames).

``report`` runs THREE rollouts from the same frame with the same weights -- the HIP device rollout, the oracle in fp32
and the oracle in fp64 -- and states both fp32 paths against fp64: MSE at steps 1 / 10 / 20, the largest relative MSE
difference over the 20 steps, the scale-relative trajectory error per step, the first step / graph at which a path
leaves the 1e-5 band, and the graphs whose fp64 trajectory passes a branch cut of the reference's feature map
(oracle cut_margin).  The fp32 oracle is the reference's own arithmetic: what it cannot hold against fp64 is the
conditioning of the rollout, not a property of the kernels.

``train_on_frames`` produces the well-conditioned case: the same model after a few hundred captured training steps
(aether_amd.training.GraphedTrainStep, the runner's MSE + AdamW) on one-frame targets of the simulated trajectories'
burn-in part.  With seed-1 (untrained) weights the rollout is chaotic -- MSE 0.05 -> 5 over 20 steps, any two fp32
evaluations separate by 1e-3..1e-2 -- while a model that follows the trajectories keeps every path within 1e-5.
"""
from __future__ import annotations

import contextlib
import io

import torch


def simulate(dev, B, N, D, burn_in=29, pred=20, seed_offset=0):
    """-> dict(loc [B, 49, N, D] fp32 on dev, q [B*N, 1], edges (host int64), x0, v0 [B*N, D] on dev, truth [pred, B*N, D] host)."""
    from aether_amd.edges import get_edges
    from aether_amd.sim import ElectrostaticFieldSim
    with contextlib.redirect_stdout(io.StringIO()):
        sim = ElectrostaticFieldSim(n_balls=N, loc_std=(N / 5.0) ** (1.0 / 3.0), dim=D, static_balls=20, device=dev)
        loc, vel, _, charges = sim.sample_trajectories(B, T=5000, sample_freq=100, as_tensor=True)
    loc = loc[:, :, :N].float()
    q = charges[:, :N].float().reshape(B * N, 1)
    return {"loc": loc, "q": q, "edges": get_edges(B, N), "B": B, "N": N, "D": D, "burn_in": burn_in, "pred": pred,
            "x0": loc[:, burn_in - 1].reshape(B * N, D).contiguous(),
            "v0": (loc[:, burn_in - 1] - loc[:, burn_in - 2]).reshape(B * N, D).contiguous(),
            "truth": loc[:, burn_in:burn_in + pred].permute(1, 0, 2, 3).reshape(pred, B * N, D).cpu()}


def train_on_frames(model, data, steps=400, lr=1e-3):
    """`steps` captured training steps on (frame t -> frame t + 1) pairs of the burn-in part of `data` (frames the
    evaluation never predicts).  Returns the final loss."""
    from aether_amd.training import GraphedTrainStep
    loc, q, B, N, D = data["loc"], data["q"], data["B"], data["N"], data["D"]
    dev = loc.device
    edges = [e.to(dev) for e in data["edges"]]
    rows, cols = edges
    qprod = q[rows] * q[cols]

    def batch(t):
        x = loc[:, t].reshape(B * N, D).contiguous()
        v = (loc[:, t] - loc[:, t - 1]).reshape(B * N, D).contiguous()
        ea = torch.cat([qprod, (x[rows] - x[cols]).norm(dim=1, keepdim=True)], 1)
        return [v.norm(dim=-1, keepdim=True), x, edges, v, ea, q], loc[:, t + 1].reshape(B * N, D).contiguous()

    model.train()
    args, tgt = batch(1)
    gs = GraphedTrainStep(model, args, tgt, lr=lr, weight_decay=1e-12, warmup=1)
    last = data["burn_in"] - 2
    loss = None
    for s in range(steps):
        a, t = batch(1 + s % last)
        loss = gs.step(a, t)
    gs.check()
    model.eval()
    return float(loss)


def report(sd, model, data, tol=1e-5):
    """Side-by-side statement for the weights `sd` (loaded into `model`).  Everything relative to the fp64 oracle."""
    from oracle import aether_oracle as O
    from aether_amd.rollout import rollout
    dev = data["x0"].device
    N, pred, truth = data["N"], data["pred"], data["truth"].double()
    x0, v0, q, edges = data["x0"], data["v0"], data["q"], data["edges"]
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    with torch.no_grad():
        hip = rollout(model, x0, v0, [e.to(dev) for e in edges], q, pred).cpu().double()
        sd32 = {k: v.detach().cpu().float() for k, v in sd.items()}
        o32 = O.rollout(sd32, x0.cpu(), v0.cpu(), edges, q.cpu(), pred).double()
        o64, margins = O.rollout({k: v.double() for k, v in sd32.items()}, x0.cpu().double(), v0.cpu().double(), edges,
                                 q.cpu().double(), pred, with_margin=True)
    recv = edges[1]
    exposed_at = {}                                              # graph -> first step its fp64 trajectory passes a cut
    for t in range(pred):
        for e in torch.nonzero(margins[t] < 2e-5).flatten().tolist():
            exposed_at.setdefault(int(recv[e]) // N, t + 1)
    clean = torch.ones(o64.shape[1], dtype=torch.bool)
    for g in exposed_at:
        clean[g * N:(g + 1) * N] = False
    scale = float(o64.abs().max())
    mse = lambda a: ((a - truth) ** 2).mean(dim=(1, 2))
    m64, mh, m32 = mse(o64), mse(hip), mse(o32)
    pick = [0, 9, pred - 1]

    def side(a, m):
        err = (a - o64).abs()                                    # [pred, nodes, D]
        per_step = (err.flatten(1).max(1).values / scale).tolist()
        first = next((t for t, v in enumerate(per_step) if v > tol), None)
        worst_graph = int(err[first].flatten(1).max(1).values.argmax()) // N if first is not None else None
        return {"mse_steps_1_10_20": [float(m[k]) for k in pick],
                "max_rel_mse_difference": float(((m - m64).abs() / m64).max()),
                "rel_mse_difference_steps_1_10_20": [float((m[k] - m64[k]).abs() / m64[k]) for k in pick],
                "trajectory_max_rel_err": max(per_step),
                "trajectory_max_rel_err_outside_cut_exposed": float(err[:, clean].max()) / scale if bool(clean.any()) else 0.0,
                "per_step_max_rel_err": [float(f"{v:.3e}") for v in per_step],
                "first_step_above_tolerance": None if first is None else first + 1,
                "graph_of_first_excess": worst_graph,
                "first_excess_graph_is_cut_exposed": None if worst_graph is None else worst_graph in exposed_at}

    h, o = side(hip, mh), side(o32, m32)
    return {"protocol": f"burn-in {data['burn_in']} frames, predict {pred}; electrostatic simulator, {data['B']} x {N} balls "
                        "+ 20 static charges; HIP device rollout and the fp32 oracle, both against the fp64 oracle",
            "mse_oracle_fp64_steps_1_10_20": [float(m64[k]) for k in pick],
            "hip": h, "oracle_fp32": o,
            "hip_over_oracle_fp32_envelope": {
                "max_rel_mse_difference": h["max_rel_mse_difference"] / max(o["max_rel_mse_difference"], 1e-30),
                "trajectory_max_rel_err": h["trajectory_max_rel_err"] / max(o["trajectory_max_rel_err"], 1e-30)},
            "cut_exposed_graphs": {str(g): t for g, t in sorted(exposed_at.items())}, "n_graphs": data["B"],
            "tolerance": tol,
            "hip_mse_within_tolerance": bool(h["max_rel_mse_difference"] <= tol),
            "oracle_fp32_mse_within_tolerance": bool(o["max_rel_mse_difference"] <= tol),
            "hip_mse_and_trajectory_within_tolerance": bool(h["max_rel_mse_difference"] <= tol and h["trajectory_max_rel_err_outside_cut_exposed"] <= tol),
            "oracle_fp32_mse_and_trajectory_within_tolerance": bool(o["max_rel_mse_difference"] <= tol
                                                 and o["trajectory_max_rel_err_outside_cut_exposed"] <= tol)}
