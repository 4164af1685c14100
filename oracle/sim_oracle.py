"""CPU restatement of the reference's dataset simulators.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under aether_amd/).

* ``electrostatic_trajectory`` <- experiments/electrostatic/dataset/electrostatic_field_sim.py:108-163
  (``ElectrostaticFieldSim.sample_trajectory`` from the first force evaluation to the end of the leap-frog
  loop): Coulomb forces between all balls, force norm capped at ``max_F``, moving balls first.
* ``gravitational_trajectory`` <- experiments/gravitational/dataset/gravitational_field_sim.py:34-43,99-125
  (``compute_acceleration`` and the kick-drift-kick loop of ``sample_trajectory``).

* ``charged_trajectory``       <- experiments/lorentz/dataset/synthetic_sim.py:167-178,221-300 (``ChargedParticlesSim``),
  :375-460 (``GravitySim``: + 0.098 on z), :536-622 (``DynamicSim``: + q (v x B)); frames [T_save, 3, n].

All take the initial state explicitly (the random draws before and after the integration are the caller's,
see aether_amd/sim.py) and work in fp64 numpy with the reference's operation order, without scipy's cdist /
einsum.  Parity status: PINNED by tests/golden/sim_{electrostatic,gravitational}.npz (the imported reference
classes, oracle/make_golden_sim.py); the lorentz-family simulators by tests/golden/sim_charged.npz.
"""
from __future__ import annotations

import numpy as np


def _coulomb(loc, edges, strength, max_F):
    diff = loc[:, None, :] - loc[None, :, :]
    l2 = np.zeros(diff.shape[:2])
    for d in range(loc.shape[1]):                                  # cdist(A, B, 'sqeuclidean')
        l2 = l2 + diff[..., d] * diff[..., d]
    with np.errstate(divide="ignore", invalid="ignore"):
        fs = strength * edges / np.power(l2, 1.5)                  # :144-145
    np.fill_diagonal(fs, 0)
    F = np.zeros_like(loc)
    for j in range(loc.shape[0]):                                  # F.sum(axis=1), in order
        F = F + fs[:, j, None] * diff[:, j, :]
    norm = np.sqrt((F * F).sum(-1, keepdims=True))
    capped = (norm > max_F).squeeze(-1)
    F[capped] = max_F * F[capped] / norm[capped]
    return F, int(capped.sum())


def electrostatic_trajectory(loc0, vel0, charges, n_balls, T, sample_freq, strength=1.0, dt=0.001, max_F=100.0):
    """loc0, vel0 [M, D], charges [M] -> (loc, vel [T/sample_freq - 1, M, D], capped count)."""
    n = n_balls
    M, D = loc0.shape
    T_save = T // sample_freq - 1
    q = np.asarray(charges, dtype=np.float64).reshape(M, 1)
    edges = q @ q.T
    loc, vel = np.zeros((T_save, M, D)), np.zeros((T_save, M, D))
    x, v = np.array(loc0, dtype=np.float64), np.array(vel0, dtype=np.float64)
    if T_save > 0:
        loc[0], vel[0] = x, v
        loc[:, n:] = loc[[0], n:]
    F, count = _coulomb(x, edges, strength, max_F)
    v[:n] += dt * F[:n]
    counter = 0
    for i in range(1, T):
        x[:n] += dt * v[:n]
        if i % sample_freq == 0:
            loc[counter, :n], vel[counter, :n] = x[:n], v[:n]
            counter += 1
        F, c = _coulomb(x, edges, strength, max_F)
        count += c
        v[:n] += dt * F[:n]
    return loc, vel, count


def _gravity(pos, mass, G, softening):
    diff = pos[None, :, :] - pos[:, None, :]
    r2 = (diff ** 2).sum(-1) + softening ** 2
    inv_r3 = np.where(r2 > 0, r2, 1.0) ** (-1.5) * (r2 > 0) + r2 * (r2 <= 0)
    a = np.zeros_like(pos)
    for j in range(pos.shape[0]):
        a = a + (G * (diff[:, j, :] * inv_r3[:, j, None])) * mass[j, 0]
    return a


def gravitational_trajectory(pos0, vel0, mass, n_balls, T, sample_freq, G=1.0, dt=0.001, softening=0.1):
    """pos0, vel0 [M, D] (velocities in the centre-of-mass frame), mass [M, 1] -> pos, vel, force [T/sample_freq, M, D]."""
    N = n_balls
    M, D = pos0.shape
    T_save = T // sample_freq
    pos_save, vel_save, force_save = (np.zeros((T_save, M, D)) for _ in range(3))
    pos, vel = np.array(pos0, dtype=np.float64), np.array(vel0, dtype=np.float64)
    mass = np.asarray(mass, dtype=np.float64).reshape(M, 1)
    acc = _gravity(pos, mass, G, softening)
    for i in range(T):
        if i % sample_freq == 0:
            k = i // sample_freq
            pos_save[k] = pos
            if i > 0:
                vel_save[k], force_save[k] = vel, acc * mass
        vel[:N] += acc[:N] * dt / 2.0
        pos[:N] += vel[:N] * dt
        acc = _gravity(pos, mass, G, softening)
        vel[:N] += acc[:N] * dt / 2.0
    return pos_save, vel_save, force_save


def charged_trajectory(loc0, vel0, charges, T, sample_freq, strength=1.0, dt=0.001, max_F=100.0, ext_mode=0,
                       ext=(0.0, 0.0, 0.0), ext_strength=0.0, pair=None):
    """loc0, vel0 [3, n] (already clamped to the box, :240), charges [n, 1] -> loc, vel [T/sample_freq - 1, 3, n].
    ext_mode 3: FixCharge (:742-746); pair [n, n]: SpringSim's pair forces (:98-110) instead of Coulomb's."""
    n = loc0.shape[1]
    T_save = T // sample_freq - 1
    q = np.zeros((n, 1)) if charges is None else np.asarray(charges, dtype=np.float64).reshape(n, 1)
    edges = q.dot(q.transpose())
    ext = np.asarray(ext, dtype=np.float64).reshape(1, 3)
    loc, vel = np.zeros((T_save, 3, n)), np.zeros((T_save, 3, n))
    x, v = np.array(loc0, dtype=np.float64), np.array(vel0, dtype=np.float64)
    if T_save > 0:
        loc[0], vel[0] = x, v

    def force(x, v):
        A = x.transpose()
        if pair is not None:
            fs = -strength * np.asarray(pair, dtype=np.float64)
        else:
            an = (A ** 2).sum(axis=1)
            l2 = an.reshape(n, 1) + an.reshape(1, n) - 2 * A.dot(A.transpose()) + 1e-6   # _l2
            with np.errstate(divide="ignore"):
                fs = strength * edges / np.power(l2, 1.5)
        fs = np.array(fs)
        np.fill_diagonal(fs, 0)
        F = np.zeros((3, n))
        for j in range(n):                                                               # sum over the last axis
            F = F + fs[:, j].reshape(1, n) * (x - x[:, [j]])
        if ext_mode == 1:
            F = F + ext.transpose()
        elif ext_mode == 2:
            F = F + (np.cross(v.transpose(), ext) * q).transpose()
        elif ext_mode == 3:
            l2f = np.power(np.sum((A - ext) ** 2, axis=-1), 3 / 2)
            F = F + ((ext_strength * q / l2f[:, None]) * (A - ext)).transpose()
        return np.clip(F, -max_F, max_F)

    v += dt * force(x, v)
    counter = 0
    for i in range(1, T):
        x += dt * v
        if i % sample_freq == 0:
            loc[counter], vel[counter] = x, v
            counter += 1
        v += dt * force(x, v)
    return loc, vel
