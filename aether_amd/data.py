"""The Lorentz runner's data set, generated and kept on the device.

``SimulatedNBodyDataset`` has the surface of ``NBodyDataset`` (experiments/lorentz/dataset4newton.py:7-94) -- ``__len__``,
``__getitem__`` = (loc[frame_0], vel[frame_0], edge_attr, charges, loc[frame_T]), ``get_n_nodes``, ``get_edges`` -- but
instead of loading the ``loc_/vel_/edges_/charges_*.npy`` files that experiments/lorentz/dataset/generate_dataset.py
writes, it runs the same simulator (``aether_amd.sim``: 'charged' / 'static' / 'dynamic') for the requested seeds in one
launch and leaves everything in HBM.  ``batches(batch_size)`` yields whole batches in the flattened layout the runner
builds per batch (main.py:205-246: ``[B n, 3]`` positions / velocities, ``edge_attr = [q_i q_j, |x_i - x_j|]``, the
edge index of ``get_edges``), so the training loop does no host-side work per batch.
"""
from __future__ import annotations

import torch

from . import sim as _sim
from .edges import get_edges, prepare_edge_attr

_SIMS = {"charged": _sim.ChargedParticlesSim, "static": _sim.GravitySim, "dynamic": _sim.DynamicSim,
         "fixcharge": _sim.FixCharge}


class SimulatedNBodyDataset:
    def __init__(self, seeds, simulation="charged", n_balls=5, length=5000, sample_freq=100, device="cuda",
                 frame_0=30, frame_T=40, initial_vel_norm=0.5):
        if simulation not in _SIMS:
            raise ValueError(f"simulation must be one of {sorted(_SIMS)}")
        self.sim = _SIMS[simulation](noise_var=0.0, n_balls=n_balls, vel_norm=initial_vel_norm, device=device)
        loc, vel, edges, charges = self.sim.sample_trajectories(list(seeds), T=length, sample_freq=sample_freq, as_tensor=True)
        if loc.shape[1] <= frame_T:
            raise ValueError("the trajectories are shorter than frame_T")
        # dataset4newton.py:46-48: [S, T, 3, n] -> [S, T, n, 3], fp32
        self.loc = loc.transpose(2, 3).to(torch.float32).contiguous()
        self.vel = vel.transpose(2, 3).to(torch.float32).contiguous()
        self.charges = charges.to(torch.float32)                              # [S, n, 1]
        n = n_balls
        rows, cols = torch.where(~torch.eye(n, dtype=torch.bool, device=self.loc.device))    # i != j, row-major (:56-61)
        self.edges = [rows, cols]
        self.edge_attr = edges.to(torch.float32)[:, rows, cols].unsqueeze(2)  # q_i q_j per edge, [S, n (n - 1), 1]
        self.frame_0, self.frame_T = frame_0, frame_T
        self.n_nodes = n

    def __len__(self):
        return self.loc.shape[0]

    def get_n_nodes(self):
        return self.n_nodes

    def __getitem__(self, i):
        return self.loc[i, self.frame_0], self.vel[i, self.frame_0], self.edge_attr[i], self.charges[i], self.loc[i, self.frame_T]

    def get_edges(self, batch_size, n_nodes):
        return get_edges(batch_size, n_nodes, device=self.loc.device)

    def batches(self, batch_size, drop_last=True):
        """Batches as the runner flattens them (main.py:205-246): dict with x, vel [B n, 3], charges [B n, 1], edges,
        edge_attr [B n (n - 1), 2] = [q_i q_j, distance], h = |v| [B n, 1] and the target positions."""
        n, S = self.n_nodes, len(self)
        for lo in range(0, S - (batch_size - 1 if drop_last else 0), batch_size):
            hi = min(S, lo + batch_size)
            B = hi - lo
            x = self.loc[lo:hi, self.frame_0].reshape(B * n, 3)
            v = self.vel[lo:hi, self.frame_0].reshape(B * n, 3)
            q = self.charges[lo:hi].reshape(B * n, 1)
            edges = self.get_edges(B, n)
            ea = prepare_edge_attr(x, edges, self.edge_attr[lo:hi].reshape(-1, 1))
            yield {"h": v.norm(dim=-1, keepdim=True), "x": x, "vel": v, "charges": q, "edges": edges, "edge_attr": ea,
                   "target": self.loc[lo:hi, self.frame_T].reshape(B * n, 3)}
