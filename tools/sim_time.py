#!/usr/bin/env python3
"""Time the device-side dataset simulators at the reference's dataset sizes
(generate_dataset.py: 5 balls + 10 field sources, T = 5000, sample_freq = 100).  (The CPU figure quoted next to these
in DESIGN.md comes from tests/test_sim.py::test_oracle_speed -- tools do not touch oracle/.)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.sim import ElectrostaticFieldSim, GravitationalFieldSim

T, SF = 5000, 100
for dim in (2, 3):
    sim = ElectrostaticFieldSim(n_balls=5, static_balls=10, dim=dim)
    sim.sample_trajectories(8, T=200, sample_freq=100)                  # warm-up
    for S in (1024, 16384):
        draws = [sim._draw_initial([0.5, 0.0, 0.5], None) for _ in range(S)]
        l0, v0, q = (np.stack([d[k] for d in draws]) for k in (1, 2, 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loc, vel, maxed = sim._integrate(l0, v0, q[..., 0], T, SF)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pairs = S * T * 15.0 * 15.0
        print("electrostatic %d-D  %6d simulations x %d steps: %.3f s  (%.0f sims/s, %.1f G pair-forces/s, capped %d)" %
              (dim, S, T, dt, S / dt, pairs / dt / 1e9, int(maxed.sum())))
np.random.seed(0)
g = GravitationalFieldSim(n_balls=5, static_balls=10, dim=3)
g.sample_trajectories(8, T=200, sample_freq=100)
for S in (1024, 16384):
    draws = [g._draw_initial() for _ in range(S)]
    m, p0, v0 = (np.stack([d[k] for d in draws]) for k in (0, 1, 2))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g._integrate(p0, v0, m[..., 0], T, SF)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("gravitational 3-D  %6d simulations x %d steps: %.3f s  (%.0f sims/s, %.1f G pair-forces/s)" %
          (S, T, dt, S / dt, S * T * 225.0 / dt / 1e9))
