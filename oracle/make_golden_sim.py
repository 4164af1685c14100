#!/usr/bin/env python3
"""Golden fixtures for the dataset simulators from the imported reference.

TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference is mounted.  Imported, unmodified:
``experiments.electrostatic.dataset.electrostatic_field_sim.ElectrostaticFieldSim`` and
``experiments.gravitational.dataset.gravitational_field_sim.GravitationalFieldSim`` (numpy + scipy only),
driven the way experiments/electrostatic/dataset/generate_dataset.py:15-60 drives them (a new field seed per
simulation).  Stored: the configuration, the seeds and the outputs; the tests replay the random draws through
the drop-in classes' host side and integrate with the oracle (CPU) or the HIP kernels (GPU).
Usage:  python oracle/make_golden_sim.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AETHER_REFERENCE", "/root/reference")

ELECTRO_CASES = {   # name: (ctor kwargs, T, sample_freq, number of simulations, first field seed or None)
    "field3d": (dict(noise_var=0.0, n_balls=5, static_balls=10, box_size=5.0, dim=3, static_charge_strength=1.0), 500, 10, 3, 7),
    "field2d": (dict(noise_var=0.0, n_balls=5, static_balls=10, box_size=5.0, dim=2, static_charge_strength=2.0), 400, 20, 2, 11),
    "free2d_noise": (dict(noise_var=0.01, n_balls=7, static_balls=0, dim=2), 200, 20, 2, None),
    "dense3d": (dict(noise_var=0.0, n_balls=12, static_balls=20, box_size=1.0, loc_std=0.5, dim=3), 300, 10, 1, 3),
}
LORENTZ_CASES = {   # name: (class name, ctor kwargs, T, sample_freq, seeds)
    "charged5": ("ChargedParticlesSim", dict(noise_var=0.0, n_balls=5, vel_norm=0.5), 500, 10, [3, 4]),
    "static5": ("GravitySim", dict(noise_var=0.0, n_balls=5, vel_norm=0.5), 400, 20, [5]),
    "dynamic5": ("DynamicSim", dict(noise_var=0.0, n_balls=5, vel_norm=0.5), 500, 10, [6, 7]),
    "dynamic20_noise": ("DynamicSim", dict(noise_var=0.01, n_balls=20, vel_norm=0.5), 300, 10, [8]),
    "fixcharge5": ("FixCharge", dict(noise_var=0.0, n_balls=5, vel_norm=0.5), 400, 10, [9, 10]),
}
SPRING_CASES = {"springs5": (dict(noise_var=0.0, n_balls=5), 400, 10, 2, 12), "springs10_noise": (dict(noise_var=0.02, n_balls=10), 300, 20, 1, 13)}
GRAV_CASES = {      # name: (ctor kwargs, T, sample_freq, number of simulations, numpy seed)
    "grav3d": (dict(n_balls=5, static_balls=3, dim=3, static_mass=2.0, noise_var=0.0), 300, 10, 2, 5),
    "grav2d_noise": (dict(n_balls=6, static_balls=0, dim=2, noise_var=0.01, softening=0.05), 200, 20, 2, 6),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, REF)
    from experiments.electrostatic.dataset.electrostatic_field_sim import ElectrostaticFieldSim
    from experiments.gravitational.dataset.gravitational_field_sim import GravitationalFieldSim
    out = {}
    for name, (kw, T, sf, S, seed0) in ELECTRO_CASES.items():
        sim = ElectrostaticFieldSim(**kw)
        res, counts = [], []
        for i in range(S):
            if seed0 is not None:                                  # generate_dataset.py:31-34
                sim._field_seed = seed0 + i
                sim.reset_field_rng()
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), np.errstate(invalid="ignore"):
                res.append(sim.sample_trajectory(T=T, sample_freq=sf))
            counts.append(int(buf.getvalue().split()[-1]))
        for k, key in enumerate(("loc", "vel", "edges", "charges")):
            out[f"{name}.{key}"] = np.stack([r[k] for r in res])
        out[f"{name}.maxed"] = np.asarray(counts, dtype=np.int64)
        print(name, out[f"{name}.loc"].shape, "capped", counts)
    np.savez(os.path.join(args.out, "sim_electrostatic.npz"), **out)
    out = {}
    for name, (kw, T, sf, S, seed) in GRAV_CASES.items():
        np.random.seed(seed)
        sim = GravitationalFieldSim(**kw)
        res = [sim.sample_trajectory(T=T, sample_freq=sf) for _ in range(S)]
        for k, key in enumerate(("pos", "vel", "force", "mass")):
            out[f"{name}.{key}"] = np.stack([r[k] for r in res])
        print(name, out[f"{name}.pos"].shape)
    np.savez(os.path.join(args.out, "sim_gravitational.npz"), **out)
    import experiments.lorentz.dataset.synthetic_sim as LS
    out = {}
    for name, (cls, kw, T, sf, seeds) in LORENTZ_CASES.items():
        with contextlib.redirect_stdout(io.StringIO()):
            sim = getattr(LS, cls)(**kw)
        res = [sim.sample_trajectory(seed, T=T, sample_freq=sf) for seed in seeds]
        for k, key in enumerate(("loc", "vel", "edges", "charges")):
            out[f"{name}.{key}"] = np.stack([r[k] for r in res])
        print(name, out[f"{name}.loc"].shape, "max |F-limited| frames ok")
    for name, (kw, T, sf, S, seed) in SPRING_CASES.items():          # SpringSim: global generator, no reseeding inside
        np.random.seed(seed)
        sim = LS.SpringSim(**kw)
        res = [sim.sample_trajectory(T=T, sample_freq=sf) for _ in range(S)]
        for k, key in enumerate(("loc", "vel", "edges")):
            out[f"{name}.{key}"] = np.stack([r[k] for r in res])
        print(name, out[f"{name}.loc"].shape)
    np.savez(os.path.join(args.out, "sim_charged.npz"), **out)


if __name__ == "__main__":
    main()
