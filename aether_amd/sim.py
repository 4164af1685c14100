"""The reference's dataset simulators with the integration on the MI355X.

``ElectrostaticFieldSim`` and ``GravitationalFieldSim`` keep the constructors, attributes, random-number protocol
and ``sample_trajectory`` results of experiments/electrostatic/dataset/electrostatic_field_sim.py and
experiments/gravitational/dataset/gravitational_field_sim.py: every random draw (charges, initial state, field
sources, observation noise) is made on the host with the same numpy generators in the same order, so a dataset
comes out draw for draw; the T-step integration -- the part that takes the reference hours for 70 000
simulations -- runs in ``aether_sim_electrostatic`` / ``aether_sim_gravitational`` (fp64).
``sample_trajectories(num_sims, ...)`` batches the loop of experiments/electrostatic/dataset/generate_dataset.py
:15-60 into one launch.  ``as_tensor=True`` leaves the frames on the device for the model.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _device(device):
    if not torch.cuda.is_available():
        raise _lib.AetherHipError("aether_amd.sim integrates on an MI355X only (there is no CPU fallback)")
    return torch.device(device)


class ElectrostaticFieldSim(object):
    def __init__(self, n_balls=5, box_size=5., loc_std=1., vel_norm=0.5, interaction_strength=1., noise_var=0., dim=2,
                 static_balls=0, static_charge_strength=1.0, device="cuda"):
        self.n_balls, self.box_size, self.loc_std, self.vel_norm = n_balls, box_size, loc_std, vel_norm
        self.interaction_strength, self.noise_var, self.dim, self.static_balls = interaction_strength, noise_var, dim, static_balls
        self._charge_types = np.array([-1., 0., 1.])
        self._static_charge_strength = static_charge_strength
        self._delta_T = 0.001
        self._max_F = 0.1 / self._delta_T
        self._particle_seed = 0
        self.reset_particle_rng()
        self._field_seed = 1
        self.reset_field_rng()
        self.sampler = self.sample_location_inside_box
        self.device = device
        self.last_maxed_out = None
        if n_balls + static_balls > 64 or dim not in (2, 3):
            raise ValueError("at most 64 balls per simulation, dim 2 or 3")

    def reset_particle_rng(self):
        self.particle_rng = np.random.default_rng(self._particle_seed)

    def reset_field_rng(self):
        self.field_rng = np.random.default_rng(self._field_seed)

    def sample_location_inside_box(self):
        return self.field_rng.uniform(-self.box_size, self.box_size, (self.static_balls, self.dim))

    # -- host side: the random draws of sample_trajectory, in its order --------------------------------------
    def _draw_initial(self, charge_prob, field_charge_prob):
        n = self.n_balls
        if self.static_balls > 0:                                          # electrostatic_field_sim.py:79-92
            field_charge_prob = charge_prob if field_charge_prob is None else field_charge_prob
            charges = np.concatenate([
                self.particle_rng.choice(self._charge_types, size=(n, 1), p=charge_prob),
                self.field_rng.choice(self._charge_types, size=(self.static_balls, 1), p=field_charge_prob)
                * self._static_charge_strength])
        else:
            charges = self.particle_rng.choice(self._charge_types, size=(n, 1), p=charge_prob)
        loc0 = np.concatenate([self.particle_rng.normal(size=(n, self.dim)) * self.loc_std, self.sampler()], 0)   # :99-101
        vel0 = self.particle_rng.normal(size=(n, self.dim))                                                         # :102-104
        vel0 = vel0 * self.vel_norm / np.sqrt((vel0 ** 2).sum(axis=1, keepdims=True))
        vel0 = np.concatenate([vel0, np.zeros((self.static_balls, self.dim))], 0)
        # the reference asserts that no pair starts without interaction (:121)
        total = n + self.static_balls
        diff = loc0[:, None, :] - loc0[None, :, :]
        with np.errstate(divide="ignore"):
            fs = self.interaction_strength * (charges @ charges.T) / np.power((diff ** 2).sum(-1), 1.5)
        assert np.abs(fs[~np.eye(total, dtype=bool)]).min() > 1e-10
        return charges, loc0, vel0

    def _draw_noise(self, T_save):
        n = self.n_balls                                                    # :164-166
        return (self.particle_rng.normal(size=(T_save, n, self.dim)) * self.noise_var,
                self.particle_rng.normal(size=(T_save, n, self.dim)) * self.noise_var)

    # -- device side ----------------------------------------------------------------------------------------------
    def _integrate(self, loc0, vel0, charges, T, sample_freq):
        """loc0, vel0 [S, M, D], charges [S, M] (numpy fp64) -> loc, vel [S, T_save, M, D] (CUDA fp64), capped [S]."""
        lib = _lib.load()
        dev = _device(self.device)
        S, M, D = loc0.shape
        T_save = T // sample_freq - 1
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        l0, v0, q = up(loc0), up(vel0), up(charges)
        loc = torch.empty(S, max(T_save, 0), M, D, dtype=torch.float64, device=dev)
        vel = torch.empty_like(loc)
        maxed = torch.empty(S, dtype=torch.int64, device=dev)
        st = lib.aether_sim_electrostatic(l0.data_ptr(), v0.data_ptr(), q.data_ptr(), S, self.n_balls, M, D, T, sample_freq,
                                          float(self.interaction_strength), float(self._delta_T), float(self._max_F),
                                          loc.data_ptr(), vel.data_ptr(), maxed.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_sim_electrostatic")
        return loc, vel, maxed

    def sample_trajectories(self, num_sims, T=10000, sample_freq=10, charge_prob=[0.5, 0.0, 0.5], field_charge_prob=None,
                            field_seeds=None, reset_field_rng=False, as_tensor=False):
        """``num_sims`` consecutive ``sample_trajectory`` calls in one launch.  ``field_seeds`` (an iterator) and
        ``reset_field_rng`` reproduce generate_dataset.py:31-34 (a new field seed per simulation / the same
        field for all).  Returns loc, vel [S, T_save, M, D], edges [S, M, M], charges [S, M, 1]."""
        assert T % sample_freq == 0
        T_save = int(T / sample_freq - 1)
        n = self.n_balls
        draws = []
        for _ in range(num_sims):
            if field_seeds is not None:
                self._field_seed = next(field_seeds)
            if reset_field_rng or field_seeds is not None:
                self.reset_field_rng()
            draws.append(self._draw_initial(charge_prob, field_charge_prob) + self._draw_noise(T_save))
        charges = np.stack([d[0] for d in draws])
        loc, vel, maxed = self._integrate(np.stack([d[1] for d in draws]), np.stack([d[2] for d in draws]),
                                          charges[..., 0], T, sample_freq)
        self.last_maxed_out = maxed
        if self.noise_var != 0:
            loc[:, :, :n] += torch.from_numpy(np.stack([d[3] for d in draws])).to(loc.device)
            vel[:, :, :n] += torch.from_numpy(np.stack([d[4] for d in draws])).to(loc.device)
        edges = charges @ charges.transpose(0, 2, 1)
        if as_tensor:
            return loc, vel, torch.from_numpy(edges).to(loc.device), torch.from_numpy(charges).to(loc.device)
        return loc.cpu().numpy(), vel.cpu().numpy(), edges, charges

    def sample_trajectory(self, T=10000, sample_freq=10, charge_prob=[0.5, 0.0, 0.5], field_charge_prob=None):
        """electrostatic_field_sim.py:63-170: (loc, vel [T_save, M, D], edges [M, M], charges [M, 1])."""
        loc, vel, edges, charges = self.sample_trajectories(1, T, sample_freq, charge_prob, field_charge_prob)
        print(int(self.last_maxed_out[0]))                                  # :168
        return loc[0], vel[0], edges[0], charges[0]


class GravitationalFieldSim(object):
    def __init__(self, n_balls=100, box_size=1.0, loc_std=1, vel_norm=0.5, interaction_strength=1, noise_var=0, dt=0.001,
                 softening=0.1, dim=3, static_balls=0, static_mass=1.0, device="cuda", **kwargs):
        self.n_balls, self.loc_std, self.vel_norm, self.interaction_strength = n_balls, loc_std, vel_norm, interaction_strength
        self.noise_var, self.dt, self.softening = noise_var, dt, softening
        self.position_variance = 1.0
        self.dim, self.static_balls, self.static_mass = dim, static_balls, static_mass
        self._field_seed = 1
        self.reset_field_rng()
        self.box_size = box_size
        self.device = device
        if n_balls + static_balls > 64 or dim not in (2, 3):
            raise ValueError("at most 64 balls per simulation, dim 2 or 3")

    def reset_field_rng(self):
        self.field_rng = np.random.default_rng(self._field_seed)

    def sample_location_inside_box(self):
        return self.field_rng.uniform(-self.box_size, self.box_size, (self.static_balls, self.dim))

    def _draw_initial(self):
        N, total = self.n_balls, self.n_balls + self.static_balls           # gravitational_field_sim.py:87-96
        mass = np.concatenate([np.ones((N, 1)), self.static_mass * np.ones((self.static_balls, 1))], 0)
        pos = self.position_variance * np.random.randn(total, self.dim)
        vel = np.concatenate([np.random.randn(N, self.dim), np.zeros((self.static_balls, self.dim))], 0)
        vel -= np.mean(mass * vel, 0) / np.mean(mass)
        return mass, pos, vel

    def _draw_noise(self, T_save):
        N = self.n_balls                                                    # :128-130
        return tuple(np.random.randn(T_save, N, self.dim) * self.noise_var for _ in range(3))

    def _integrate(self, pos0, vel0, mass, T, sample_freq):
        lib = _lib.load()
        dev = _device(self.device)
        S, M, D = pos0.shape
        T_save = T // sample_freq
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        p0, v0, m = up(pos0), up(vel0), up(mass)
        pos = torch.empty(S, T_save, M, D, dtype=torch.float64, device=dev)
        vel, force = torch.empty_like(pos), torch.empty_like(pos)
        st = lib.aether_sim_gravitational(p0.data_ptr(), v0.data_ptr(), m.data_ptr(), S, self.n_balls, M, D, T, sample_freq,
                                          float(self.interaction_strength), float(self.dt), float(self.softening),
                                          pos.data_ptr(), vel.data_ptr(), force.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_sim_gravitational")
        return pos, vel, force

    def sample_trajectories(self, num_sims, T=10000, sample_freq=10, as_tensor=False):
        """``num_sims`` consecutive ``sample_trajectory`` calls (global numpy generator, as the reference) in one
        launch: pos, vel, force [S, T_save, M, D], mass [S, M, 1]."""
        assert T % sample_freq == 0
        T_save = int(T / sample_freq)
        N = self.n_balls
        draws = [self._draw_initial() + self._draw_noise(T_save) for _ in range(num_sims)]
        mass = np.stack([d[0] for d in draws])
        pos, vel, force = self._integrate(np.stack([d[1] for d in draws]), np.stack([d[2] for d in draws]), mass[..., 0],
                                          T, sample_freq)
        if self.noise_var != 0:
            for out, k in ((pos, 3), (vel, 4), (force, 5)):
                out[:, :, :N] += torch.from_numpy(np.stack([d[k] for d in draws])).to(out.device)
        if as_tensor:
            return pos, vel, force, torch.from_numpy(mass).to(pos.device)
        return pos.cpu().numpy(), vel.cpu().numpy(), force.cpu().numpy(), mass

    def _energy_total(self, pos, vel, mass):
        """Kinetic + softened potential energy of one frame (the quantity the integrator conserves; the reference's
        ``_energy`` (:45-73) uses the unsoftened potential)."""
        ke = 0.5 * np.sum(mass * vel ** 2)
        d = pos[None, :, :] - pos[:, None, :]
        r = np.sqrt((d ** 2).sum(-1) + self.softening ** 2)
        pe = -self.interaction_strength * np.sum(np.triu((mass * mass.T) / r, 1))
        return ke + pe

    def sample_trajectory(self, T=10000, sample_freq=10):
        """gravitational_field_sim.py:75-131: (pos, vel, force [T_save, M, D], mass [M, 1])."""
        pos, vel, force, mass = self.sample_trajectories(1, T, sample_freq)
        return pos[0], vel[0], force[0], mass[0]


class _LorentzFamilySim(object):
    """Common part of experiments/lorentz/dataset/synthetic_sim.py's ChargedParticlesSim / GravitySim / DynamicSim
    (the data sets of the state2state runner): constructor, ``_clamp`` and the random draws of ``sample_trajectory``
    on the host (global numpy generator, seeded per trajectory as the reference does), integration in
    ``aether_sim_charged``."""

    _ext_mode, _ext, _ext_strength = 0, (0.0, 0.0, 0.0), 0.0

    def __init__(self, n_balls=5, box_size=5., loc_std=1., vel_norm=0.5, interaction_strength=1., noise_var=0.,
                 device="cuda"):
        self.n_balls = n_balls
        self.box_size = box_size
        self.loc_std = loc_std * (float(n_balls) / 5.) ** (1 / 3)
        print(self.loc_std)                                                 # synthetic_sim.py:156
        self.vel_norm = vel_norm
        self.interaction_strength = interaction_strength
        self.noise_var = noise_var
        self._charge_types = np.array([-1., 0., 1.])
        self._delta_T = 0.001
        self._max_F = 0.1 / self._delta_T
        self.dim = 3
        self.device = device
        if n_balls > 64:
            raise ValueError("at most 64 balls per simulation")

    def _clamp(self, loc, vel):
        """Reflect positions outside the box back inside, velocities pointing inwards (synthetic_sim.py:196-219;
        in place, as the reference: the clamped arrays are the initial state of the integration)."""
        lim = self.box_size
        assert np.all(np.abs(loc) < 3 * lim)
        hi, lo = loc > lim, loc < -lim
        loc[hi] = 2 * lim - loc[hi]
        vel[hi] = -np.abs(vel[hi])
        loc[lo] = -2 * lim - loc[lo]
        vel[lo] = np.abs(vel[lo])
        return loc, vel

    def _draw_initial(self, seed, charge_prob):
        n = self.n_balls
        np.random.seed(seed)                                               # :229
        charges = np.random.choice(self._charge_types, size=(n, 1), p=charge_prob)
        loc0 = np.random.randn(self.dim, n) * self.loc_std
        vel0 = np.random.randn(self.dim, n)
        vel0 = vel0 * self.vel_norm / np.sqrt((vel0 ** 2).sum(axis=0)).reshape(1, -1)
        loc0, vel0 = self._clamp(loc0, vel0)
        if n > 1:                                                          # the reference's start-up assertion (:258 / :571-572)
            A = loc0.transpose()
            an = (A ** 2).sum(axis=1)
            l2 = an.reshape(n, 1) + an.reshape(1, n) - 2 * A.dot(A.transpose()) + 1e-6
            with np.errstate(divide="ignore"):
                fs = self.interaction_strength * charges.dot(charges.transpose()) / np.power(l2, 1.5)
            assert np.abs(fs[~np.eye(n, dtype=bool)]).min() > 1e-10
        return charges, loc0, vel0

    def sample_trajectories(self, seeds, T=10000, sample_freq=10, charge_prob=[1. / 2, 0, 1. / 2], as_tensor=False):
        """One ``sample_trajectory(seed, ...)`` per entry of ``seeds`` in one launch: loc, vel [S, T_save, 3, n],
        edges [S, n, n], charges [S, n, 1].  (The reference reseeds the global generator per trajectory, so the
        draws of a batch are those of the individual calls; with noise_var > 0 the noise of trajectory k is drawn
        right after its initial state, as in the reference.)"""
        assert T % sample_freq == 0
        lib = _lib.load()
        dev = _device(self.device)
        T_save = int(T / sample_freq - 1)
        n = self.n_balls
        draws = []
        for seed in seeds:
            c, l0, v0 = self._draw_initial(seed, charge_prob)
            noise = None
            if self.noise_var > 0:                                         # :293-295
                noise = (np.random.randn(T_save, self.dim, n) * self.noise_var, np.random.randn(T_save, self.dim, n) * self.noise_var)
            draws.append((c, l0, v0, noise))
        S = len(draws)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        charges = np.stack([d[0] for d in draws])
        l0, v0, q = up(np.stack([d[1] for d in draws])), up(np.stack([d[2] for d in draws])), up(charges[..., 0])
        loc = torch.empty(S, max(T_save, 0), 3, n, dtype=torch.float64, device=dev)
        vel = torch.empty_like(loc)
        ext = (C.c_double * 3)(*self._ext)
        st = lib.aether_sim_charged(l0.data_ptr(), v0.data_ptr(), q.data_ptr(), None, S, n, T, sample_freq,
                                    float(self.interaction_strength), float(self._delta_T), float(self._max_F),
                                    int(self._ext_mode), ext, float(self._ext_strength), loc.data_ptr(), vel.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_sim_charged")
        if self.noise_var > 0:
            loc += up(np.stack([d[3][0] for d in draws]))
            vel += up(np.stack([d[3][1] for d in draws]))
        edges = charges @ charges.transpose(0, 2, 1)
        if as_tensor:
            return loc, vel, torch.from_numpy(edges).to(dev), torch.from_numpy(charges).to(dev)
        return loc.cpu().numpy(), vel.cpu().numpy(), edges, charges

    def sample_trajectory(self, seed, T=10000, sample_freq=10, charge_prob=[1. / 2, 0, 1. / 2]):
        """synthetic_sim.py:221-300 / :375-460 / :536-622: (loc, vel [T_save, 3, n], edges [n, n], charges [n, 1])."""
        loc, vel, edges, charges = self.sample_trajectories([seed], T, sample_freq, charge_prob)
        return loc[0], vel[0], edges[0], charges[0]


class ChargedParticlesSim(_LorentzFamilySim):
    """'charged' (synthetic_sim.py:149-300)."""


class GravitySim(_LorentzFamilySim):
    """'static': a constant force 0.098 on z (synthetic_sim.py:303-460)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.gravity_constant = 0.098
        self._ext_mode, self._ext = 1, (0.0, 0.0, self.gravity_constant)


class DynamicSim(_LorentzFamilySim):
    """'dynamic': the Lorentz force q (v x B), B = 0.5 (1, 1, 1) (synthetic_sim.py:463-622)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lorentz_field = np.ones([1, 3]) * 0.5
        self._ext_mode, self._ext = 2, tuple(self.lorentz_field[0])


class FixCharge(_LorentzFamilySim):
    """'fixcharge': the Coulomb force of a fixed charge at (10, 10, 10) with strength 0.1 (synthetic_sim.py:626-790)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.fix_pos = np.array([10, 10, 10])
        self.interaction_stren_fix = 0.1
        self._ext_mode, self._ext, self._ext_strength = 3, tuple(float(v) for v in self.fix_pos), self.interaction_stren_fix


class SpringSim(object):
    """'springs' (synthetic_sim.py:6-146): random spring constants {0, 0.5, 1} between pairs, no charges; draws from the
    global numpy generator without reseeding, noise drawn even when noise_var is 0 -- as the reference."""

    def __init__(self, n_balls=5, box_size=5., loc_std=.5, vel_norm=.5, interaction_strength=.1, noise_var=0., device="cuda"):
        self.n_balls, self.box_size, self.loc_std, self.vel_norm = n_balls, box_size, loc_std, vel_norm
        self.interaction_strength, self.noise_var = interaction_strength, noise_var
        self._spring_types = np.array([0., 0.5, 1.])
        self._delta_T = 0.001
        self._max_F = 0.1 / self._delta_T
        self.dim = 3
        self.device = device
        if n_balls > 64:
            raise ValueError("at most 64 balls per simulation")

    _clamp = _LorentzFamilySim._clamp

    def _draw_initial(self, spring_prob):
        n = self.n_balls
        edges = np.random.choice(self._spring_types, size=(n, n), p=spring_prob)          # :83-87
        edges = np.tril(edges) + np.tril(edges, -1).T
        np.fill_diagonal(edges, 0)
        loc0 = np.random.randn(self.dim, n) * self.loc_std
        vel0 = np.random.randn(self.dim, n)
        vel0 = vel0 * self.vel_norm / np.sqrt((vel0 ** 2).sum(axis=0)).reshape(1, -1)
        loc0, vel0 = self._clamp(loc0, vel0)
        return edges, loc0, vel0

    def sample_trajectories(self, num_sims, T=10000, sample_freq=10, spring_prob=[1. / 2, 0, 1. / 2], as_tensor=False):
        """``num_sims`` consecutive ``sample_trajectory`` calls in one launch: loc, vel [S, T_save, 3, n], edges [S, n, n]."""
        assert T % sample_freq == 0
        lib = _lib.load()
        dev = _device(self.device)
        T_save = int(T / sample_freq - 1)
        n = self.n_balls
        draws = []
        for _ in range(num_sims):
            e, l0, v0 = self._draw_initial(spring_prob)
            draws.append((e, l0, v0, np.random.randn(T_save, self.dim, n) * self.noise_var,
                          np.random.randn(T_save, self.dim, n) * self.noise_var))           # :143-144
        S = len(draws)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        edges = np.stack([d[0] for d in draws])
        l0, v0, pair = up(np.stack([d[1] for d in draws])), up(np.stack([d[2] for d in draws])), up(edges)
        loc = torch.empty(S, max(T_save, 0), 3, n, dtype=torch.float64, device=dev)
        vel = torch.empty_like(loc)
        st = lib.aether_sim_charged(l0.data_ptr(), v0.data_ptr(), None, pair.data_ptr(), S, n, T, sample_freq,
                                    float(self.interaction_strength), float(self._delta_T), float(self._max_F), 0, None, 0.0,
                                    loc.data_ptr(), vel.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_sim_charged")
        if self.noise_var != 0:
            loc += up(np.stack([d[3] for d in draws]))
            vel += up(np.stack([d[4] for d in draws]))
        if as_tensor:
            return loc, vel, torch.from_numpy(edges).to(dev)
        return loc.cpu().numpy(), vel.cpu().numpy(), edges

    def sample_trajectory(self, T=10000, sample_freq=10, spring_prob=[1. / 2, 0, 1. / 2]):
        loc, vel, edges = self.sample_trajectories(1, T, sample_freq, spring_prob)
        return loc[0], vel[0], edges[0]
