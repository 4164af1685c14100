"""Autoregressive rollout of the state2state step (metric 2).

The reference's state2state module predicts positions only and has no rollout of its own; the
protocol used for the "20-step rollout MSE" figure is the one SURVEY.md defines and the oracle
restates (oracle/aether_oracle.py::rollout): x_{t+1} = Aether(x_t, v_t), v_{t+1} = (x_{t+1} - x_t) / dt,
with ``edge_attr = [q_i q_j, |x_i - x_j|]`` rebuilt from the current positions every step
(experiments/lorentz/main.py:243-246).  Everything stays on the device; the edge index (and therefore
the receiver-sorted graph view) is reused across steps.  ``rollout`` runs the loop inside the library
(``aether_rollout``); ``rollout_stepwise`` is the loop of module calls it replaces.
"""
from __future__ import annotations

import torch


def rollout(model, x, vel, edges, charges, steps: int, dt: float = 1.0):
    """Predicted positions ``[steps, n_nodes, D]``: the device rollout (``aether_rollout``, one kernel
    launch per step, edge attributes derived in the kernels)."""
    return model.rollout(x, vel, edges, charges, steps, dt)


@torch.no_grad()
def rollout_stepwise(model, x, vel, edges, charges, steps: int, dt: float = 1.0):
    """The same protocol as a loop of module calls with the runner's tensor ops in between (how a
    reference user would write it; kept as the cross-check of ``rollout``)."""
    rows, cols = edges
    qprod = charges[rows] * charges[cols]
    traj = []
    for _ in range(int(steps)):
        dist = torch.sqrt(torch.sum((x[rows] - x[cols]) ** 2, 1)).unsqueeze(1)
        ea = torch.cat([qprod, dist], 1)
        h = vel.norm(dim=-1, keepdim=True)            # `nodes` of the runner; ignored by the model
        xn = model(h, x, edges, vel, ea, charges)
        vel = (xn - x) / dt
        x = xn
        traj.append(x)
    return torch.stack(traj)


def rollout_mse(pred, truth):
    """Per-step MSE over (sample, particle, feature), experiments/electrostatic/evaluate.py:61-70."""
    return ((pred - truth) ** 2).mean(dim=(1, 2))
