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


class SimulatedFieldDataset:
    """The seq2seq runners' data set (experiments/electrostatic/static_electrostatic_field_data.py:8-117: ``feats``
    [S, T, N, 2D] = positions | velocities of the moving particles, the four normalisation modes, ``unnormalize`` /
    ``torch_unnormalize``, items {'inputs', 'edges', 'charges'}) over trajectories simulated on the device with
    ``aether_amd.sim.ElectrostaticFieldSim`` (a static field: the same field sources for every simulation, as
    ``generate_dataset.py --static_field``) instead of the ``*_feats`` files.  ``stats_from``: another
    ``SimulatedFieldDataset`` (the training split) whose statistics normalise this one, as the reference normalises
    validation / test data with the training set's statistics (:40-62)."""

    def __init__(self, num_sims, params=None, n_balls=5, static_balls=20, ndim=2, length=5000, sample_freq=100, box_size=5.0,
                 particle_seed=0, field_seed=1, device="cuda", stats_from=None):
        params = params or {}
        self.ndim = ndim
        self.same_norm = params.get("same_data_norm", False)
        self.symmetric_norm = params.get("symmetric_data_norm", False)
        self.no_norm = params.get("no_data_norm", False)
        self.vel_norm_norm = params.get("vel_norm_norm", False)
        sim = _sim.ElectrostaticFieldSim(noise_var=0.0, n_balls=n_balls, static_balls=static_balls, box_size=box_size, dim=ndim,
                                         device=device)
        sim._particle_seed, sim._field_seed = particle_seed, field_seed
        sim.reset_particle_rng()
        loc, vel, edges, charges = sim.sample_trajectories(num_sims, T=length, sample_freq=sample_freq,
                                                           reset_field_rng=static_balls > 0, as_tensor=True)
        n = n_balls
        self.feats = torch.cat([loc[:, :, :n], vel[:, :, :n]], -1).to(torch.float32)       # [S, T, N, 2D]
        self.edges = edges[:, :n, :n].to(torch.float32)
        self.charges = charges[:, :n, 0].to(torch.float32)
        self.static_field = loc[0, 0, n:].to(torch.float32)
        self.static_charges = charges[0, n:].to(torch.float32)
        if not self.no_norm:
            self._normalize_data(self.feats if stats_from is None else stats_from._raw)
        self._raw = self.feats if self.no_norm else self._raw

    def _normalize_data(self, train):
        D = self.ndim
        self._raw = self.feats.clone()
        if self.same_norm:
            self.feat_max, self.feat_min = train.max(), train.min()
            self.feats = (self.feats - self.feat_min) * 2 / (self.feat_max - self.feat_min) - 1
        elif self.vel_norm_norm:
            self.vel_norm_max = train[..., D:].norm(dim=-1).max()
            self.feats = self.feats / self.vel_norm_max
        else:
            if self.symmetric_norm:
                self.loc_max, self.vel_max = train[..., :D].abs().max(), train[..., D:].abs().max()
                self.loc_min, self.vel_min = -self.loc_max, -self.vel_max
            else:
                self.loc_max, self.loc_min = train[..., :D].max(), train[..., :D].min()
                self.vel_max, self.vel_min = train[..., D:].max(), train[..., D:].min()
            self.feats = torch.cat([(self.feats[..., :D] - self.loc_min) * 2 / (self.loc_max - self.loc_min) - 1,
                                    (self.feats[..., D:] - self.vel_min) * 2 / (self.vel_max - self.vel_min) - 1], -1)

    def torch_unnormalize(self, data):
        D = self.ndim
        if self.no_norm:
            return data
        if self.same_norm:
            return (data + 1) * (self.feat_max - self.feat_min) / 2. + self.feat_min
        if self.vel_norm_norm:
            return data * self.vel_norm_max
        return torch.cat([(data[..., :D] + 1) * (self.loc_max - self.loc_min) / 2. + self.loc_min,
                          (data[..., D:] + 1) * (self.vel_max - self.vel_min) / 2. + self.vel_min], -1)

    def unnormalize(self, data):
        return self.torch_unnormalize(torch.as_tensor(data, device=self.feats.device)).cpu().numpy()

    def __getitem__(self, idx):
        return {"inputs": self.feats[idx], "edges": self.edges[idx], "charges": self.charges[idx]}

    def __len__(self):
        return len(self.feats)
