"""Dataset simulators: host-side random-draw protocol + oracle vs the imported reference's outputs
(CPU), and the HIP integration vs the same outputs (GPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
import contextlib
import io

import aether_amd.sim as AS
from aether_amd.sim import ElectrostaticFieldSim, GravitationalFieldSim
from oracle import sim_oracle as SO
from oracle.make_golden_sim import ELECTRO_CASES, GRAV_CASES, LORENTZ_CASES, SPRING_CASES


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _electro_replay(name):
    """The drop-in's host side replays the reference's draws; yields per simulation (charges, loc0, vel0, noise)."""
    kw, T, sf, S, seed0 = ELECTRO_CASES[name]
    sim = ElectrostaticFieldSim(**kw)
    T_save = T // sf - 1
    for i in range(S):
        if seed0 is not None:
            sim._field_seed = seed0 + i
            sim.reset_field_rng()
        yield sim, sim._draw_initial([0.5, 0.0, 0.5], None) + sim._draw_noise(T_save)


@pytest.mark.parametrize("name", list(ELECTRO_CASES))
def test_electrostatic_oracle_and_host_protocol(name):
    d = np.load(os.path.join(GOLDEN, "sim_electrostatic.npz"))
    kw, T, sf, S, _ = ELECTRO_CASES[name]
    for i, (sim, (charges, loc0, vel0, nl, nv)) in enumerate(_electro_replay(name)):
        assert np.array_equal(charges, d[name + ".charges"][i])                      # same draws as the reference
        n = sim.n_balls
        assert np.array_equal(loc0[n:], d[name + ".loc"][i, 0, n:])                  # field sources
        loc, vel, count = SO.electrostatic_trajectory(loc0, vel0, charges[:, 0], n, T, sf, sim.interaction_strength,
                                                      sim._delta_T, sim._max_F)
        loc[:, :n] += nl
        vel[:, :n] += nv
        assert _rel(loc, d[name + ".loc"][i]) <= 1e-11 and _rel(vel, d[name + ".vel"][i]) <= 1e-11
        assert count == int(d[name + ".maxed"][i])
        assert np.array_equal(charges @ charges.T, d[name + ".edges"][i])


@pytest.mark.parametrize("name", list(GRAV_CASES))
def test_gravitational_oracle_and_host_protocol(name):
    d = np.load(os.path.join(GOLDEN, "sim_gravitational.npz"))
    kw, T, sf, S, seed = GRAV_CASES[name]
    np.random.seed(seed)
    sim = GravitationalFieldSim(**kw)
    for i in range(S):
        mass, pos0, vel0 = sim._draw_initial()
        noise = sim._draw_noise(T // sf)
        assert np.array_equal(mass, d[name + ".mass"][i])
        pos, vel, force = SO.gravitational_trajectory(pos0, vel0, mass, sim.n_balls, T, sf, sim.interaction_strength,
                                                      sim.dt, sim.softening)
        for out, nz, key in ((pos, noise[0], "pos"), (vel, noise[1], "vel"), (force, noise[2], "force")):
            out[:, :sim.n_balls] += nz
            assert _rel(out, d[f"{name}.{key}"][i]) <= 1e-11, key


def _lorentz_sim(name):
    cls, kw, T, sf, seeds = LORENTZ_CASES[name]
    with contextlib.redirect_stdout(io.StringIO()):             # the constructor prints loc_std, as the reference's
        sim = getattr(AS, cls)(**kw)
    return sim, T, sf, seeds


@pytest.mark.parametrize("name", list(LORENTZ_CASES))
def test_lorentz_family_oracle_and_host_protocol(name):
    """experiments/lorentz/dataset/synthetic_sim.py: charged / static (gravity) / dynamic (Lorentz force) data sets."""
    d = np.load(os.path.join(GOLDEN, "sim_charged.npz"))
    sim, T, sf, seeds = _lorentz_sim(name)
    for k, seed in enumerate(seeds):
        charges, loc0, vel0 = sim._draw_initial(seed, [1. / 2, 0, 1. / 2])
        assert np.array_equal(charges, d[name + ".charges"][k])
        loc, vel = SO.charged_trajectory(loc0, vel0, charges, T, sf, sim.interaction_strength, sim._delta_T, sim._max_F,
                                         sim._ext_mode, sim._ext, sim._ext_strength)
        if sim.noise_var > 0:
            loc += np.random.randn(*loc.shape) * sim.noise_var
            vel += np.random.randn(*vel.shape) * sim.noise_var
        assert _rel(loc, d[name + ".loc"][k]) <= 1e-10 and _rel(vel, d[name + ".vel"][k]) <= 1e-10


@pytest.mark.parametrize("name", list(SPRING_CASES))
def test_springs_oracle_and_host_protocol(name):
    d = np.load(os.path.join(GOLDEN, "sim_charged.npz"))
    kw, T, sf, S, seed = SPRING_CASES[name]
    np.random.seed(seed)
    sim = AS.SpringSim(**kw)
    for k in range(S):
        edges, loc0, vel0 = sim._draw_initial([1. / 2, 0, 1. / 2])
        assert np.array_equal(edges, d[name + ".edges"][k])
        loc, vel = SO.charged_trajectory(loc0, vel0, None, T, sf, sim.interaction_strength, sim._delta_T, sim._max_F, pair=edges)
        loc += np.random.randn(*loc.shape) * sim.noise_var
        vel += np.random.randn(*vel.shape) * sim.noise_var
        assert _rel(loc, d[name + ".loc"][k]) <= 1e-10 and _rel(vel, d[name + ".vel"][k]) <= 1e-10


# ----------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SPRING_CASES))
def test_springs_sim_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, "sim_charged.npz"))
    kw, T, sf, S, seed = SPRING_CASES[name]
    np.random.seed(seed)
    sim = AS.SpringSim(**kw)
    loc, vel, edges = sim.sample_trajectories(S, T, sf)
    assert np.array_equal(edges, d[name + ".edges"])
    assert _rel(loc, d[name + ".loc"]) <= 1e-9 and _rel(vel, d[name + ".vel"]) <= 1e-9
    np.random.seed(seed)
    l1, v1, e1 = AS.SpringSim(**kw).sample_trajectory(T=T, sample_freq=sf)          # the reference's call
    assert np.array_equal(l1, loc[0]) and np.array_equal(e1, edges[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(LORENTZ_CASES))
def test_lorentz_family_sim_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, "sim_charged.npz"))
    sim, T, sf, seeds = _lorentz_sim(name)
    loc, vel, edges, charges = sim.sample_trajectories(seeds, T, sf)
    assert loc.shape == d[name + ".loc"].shape and np.array_equal(charges, d[name + ".charges"])
    assert np.array_equal(edges, d[name + ".edges"])
    assert _rel(loc, d[name + ".loc"]) <= 1e-9 and _rel(vel, d[name + ".vel"]) <= 1e-9
    l1, v1, e1, c1 = sim.sample_trajectory(seeds[0], T=T, sample_freq=sf)       # the reference's call
    if sim.noise_var == 0:
        assert np.array_equal(l1, loc[0]) and np.array_equal(v1, vel[0])
    assert l1.shape == (T // sf - 1, 3, sim.n_balls)

@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ELECTRO_CASES))
def test_electrostatic_sim_matches_reference(name, capsys):
    d = np.load(os.path.join(GOLDEN, "sim_electrostatic.npz"))
    kw, T, sf, S, seed0 = ELECTRO_CASES[name]
    # one batched launch, with the generator script's field-seed protocol
    sim = ElectrostaticFieldSim(**kw)
    seeds = None if seed0 is None else iter(range(seed0, seed0 + S))
    loc, vel, edges, charges = sim.sample_trajectories(S, T, sf, field_seeds=seeds)
    assert loc.dtype == np.float64 and loc.shape == d[name + ".loc"].shape
    assert np.array_equal(charges, d[name + ".charges"]) and np.array_equal(edges, d[name + ".edges"])
    assert _rel(loc, d[name + ".loc"]) <= 1e-9 and _rel(vel, d[name + ".vel"]) <= 1e-9
    assert sim.last_maxed_out.tolist() == d[name + ".maxed"].tolist()
    # the reference's own call sequence, one simulation at a time
    sim = ElectrostaticFieldSim(**kw)
    for i in range(S):
        if seed0 is not None:
            sim._field_seed = seed0 + i
            sim.reset_field_rng()
        l1, v1, e1, c1 = sim.sample_trajectory(T=T, sample_freq=sf)
        assert capsys.readouterr().out.split()[-1] == str(int(d[name + ".maxed"][i]))      # prints the capped count
        assert np.array_equal(l1, loc[i]) and np.array_equal(v1, vel[i]) and np.array_equal(c1, charges[i])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GRAV_CASES))
def test_gravitational_sim_matches_reference(name):
    d = np.load(os.path.join(GOLDEN, "sim_gravitational.npz"))
    kw, T, sf, S, seed = GRAV_CASES[name]
    np.random.seed(seed)
    sim = GravitationalFieldSim(**kw)
    for i in range(S):                                             # sequential calls share the global generator
        pos, vel, force, mass = sim.sample_trajectory(T=T, sample_freq=sf)
        assert np.array_equal(mass, d[name + ".mass"][i])
        for got, key in ((pos, "pos"), (vel, "vel"), (force, "force")):
            assert _rel(got, d[f"{name}.{key}"][i]) <= 1e-9, key


@pytest.mark.gpu
def test_sim_batches_vs_oracle_and_errors():
    """Many simulations per launch (several share a wavefront; the last wavefront is ragged), tensors left on the
    device, 64 balls, and the argument checks."""
    from aether_amd import _lib
    sim = ElectrostaticFieldSim(n_balls=4, static_balls=5, dim=2, box_size=3.0)
    sim._particle_seed = 9
    sim.reset_particle_rng()
    loc, vel, edges, charges = sim.sample_trajectories(37, T=120, sample_freq=10, as_tensor=True)
    assert loc.is_cuda and loc.shape == (37, 11, 9, 2)
    ref = ElectrostaticFieldSim(n_balls=4, static_balls=5, dim=2, box_size=3.0)
    ref._particle_seed = 9
    ref.reset_particle_rng()
    for i in range(37):
        c, l0, v0 = ref._draw_initial([0.5, 0.0, 0.5], None)
        ref._draw_noise(11)
        want_l, want_v, cnt = SO.electrostatic_trajectory(l0, v0, c[:, 0], 4, 120, 10)
        assert _rel(loc[i].cpu().numpy(), want_l) <= 1e-9 and _rel(vel[i].cpu().numpy(), want_v) <= 1e-9
        assert int(sim.last_maxed_out[i]) == cnt
    np.random.seed(3)
    g = GravitationalFieldSim(n_balls=40, static_balls=24, dim=3, static_mass=0.5)
    pos, vel, force, mass = g.sample_trajectories(3, T=40, sample_freq=10)
    np.random.seed(3)
    g2 = GravitationalFieldSim(n_balls=40, static_balls=24, dim=3, static_mass=0.5)
    for i in range(3):
        m, p0, v0 = g2._draw_initial()
        g2._draw_noise(4)
        wp, wv, wf = SO.gravitational_trajectory(p0, v0, m, 40, 40, 10)
        assert _rel(pos[i], wp) <= 1e-9 and _rel(vel[i], wv) <= 1e-9 and _rel(force[i], wf) <= 1e-9
    with pytest.raises(ValueError):
        ElectrostaticFieldSim(n_balls=60, static_balls=10)
    with pytest.raises(AssertionError):
        sim.sample_trajectories(1, T=105, sample_freq=10)
    lib = _lib.load()
    assert lib.aether_sim_electrostatic(None, None, None, 1, 1, 1, 2, 10, 10, 1.0, 0.001, 100.0, None, None, None, None) != 0


@pytest.mark.gpu
def test_sim_conservation_properties_at_dataset_size():
    """Properties that hold at the data sets' full length (T = 5000, sample_freq = 100, 49 frames) without an oracle:
    the gravitational simulator without static masses keeps the total momentum at zero and its energy within the
    integrator's drift; the electrostatic simulator conserves energy while no force is capped; static field sources
    never move; saved frames are finite."""
    np.random.seed(11)
    g = GravitationalFieldSim(n_balls=8, static_balls=0, dim=3, softening=0.1)
    pos, vel, force, mass = g.sample_trajectories(16, T=5000, sample_freq=100)
    assert pos.shape == (16, 50, 8, 3) and np.isfinite(pos).all() and np.isfinite(vel).all()
    mom = (mass[:, None] * vel).sum(axis=2)                                  # [S, T_save, 3]
    assert np.abs(mom[:, 1:]).max() <= 1e-10
    e = np.array([[g._energy_total(pos[s, t], vel[s, t], mass[s]) for t in range(1, 50)] for s in range(16)])
    assert np.abs(e - e[:, :1]).max() <= 2e-3 * np.abs(e[:, :1]).max()
    sim = ElectrostaticFieldSim(n_balls=3, static_balls=4, dim=2, box_size=5.0, loc_std=2.0)
    loc, vel, edges, charges = sim.sample_trajectories(32, T=5000, sample_freq=100)
    assert np.isfinite(loc).all() and np.isfinite(vel).all()
    assert np.array_equal(loc[:, :, 3:], np.repeat(loc[:, :1, 3:], loc.shape[1], axis=1))      # field sources fixed
    assert np.abs(vel[:, :, 3:]).max() == 0.0


def test_oracle_speed(capsys):
    """Informative: the speed of the numpy restatement of the electrostatic simulator on one host core (the CPU figure
    DESIGN.md quotes next to the device simulators); checks only that 200 steps of 15 balls finish."""
    import time
    sim = ElectrostaticFieldSim(n_balls=5, static_balls=10, dim=3)
    charges, loc0, vel0 = sim._draw_initial([0.5, 0.0, 0.5], None)
    t0 = time.perf_counter()
    loc, vel, _ = SO.electrostatic_trajectory(loc0, vel0, charges[:, 0], 5, 200, 100)
    dt = time.perf_counter() - t0
    with capsys.disabled():
        print("\n[oracle] electrostatic, 15 balls: %.1f us per step on one core -> %.2f s per 5000-step simulation"
              % (dt / 200 * 1e6, dt / 200 * 5000))
    assert loc.shape == (1, 15, 3) and np.isfinite(loc).all()


@pytest.mark.gpu
def test_simulated_dataset_feeds_the_training_loop():
    """aether_amd.data.SimulatedNBodyDataset: the runner's data set surface (dataset4newton.py:7-94) over trajectories
    simulated on the device; its items equal what the reference's preprocessing makes of the simulator's arrays, and a
    few optimizer steps on its batches reduce the loss (simulator -> data set -> HIP training step, nothing on the host)."""
    import torch
    from aether_amd.data import SimulatedNBodyDataset
    from aether_amd.nn.state2state.aether import Aether
    with contextlib.redirect_stdout(io.StringIO()):
        ds = SimulatedNBodyDataset(range(100, 164), simulation="dynamic", n_balls=5, length=5000, sample_freq=100)
        ref = AS.DynamicSim(noise_var=0.0, n_balls=5, vel_norm=0.5)
    assert len(ds) == 64 and ds.get_n_nodes() == 5
    loc, vel, edges, charges = ref.sample_trajectory(103, T=5000, sample_freq=100)       # the arrays the files would hold
    l0, v0, ea, q, lT = ds[3]
    assert l0.is_cuda and l0.shape == (5, 3) and ea.shape == (20, 1) and q.shape == (5, 1)
    assert np.allclose(l0.cpu().numpy(), loc[30].T.astype(np.float32)) and np.allclose(lT.cpu().numpy(), loc[40].T.astype(np.float32))
    assert np.allclose(v0.cpu().numpy(), vel[30].T.astype(np.float32))
    want_ea = np.array([edges[i, j] for i in range(5) for j in range(5) if i != j], dtype=np.float32)   # :56-61
    assert np.array_equal(ea.cpu().numpy()[:, 0], want_ea) and np.array_equal(q.cpu().numpy(), charges.astype(np.float32))
    e = ds.get_edges(2, 5)
    assert e[0].tolist()[:4] == [0, 0, 0, 0] and e[1].tolist()[:4] == [1, 2, 3, 4] and e[0][20] == 5
    torch.manual_seed(0)
    model = Aether(6, 64, 0.0, 3, device="cuda")
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    losses = []
    for epoch in range(6):
        for b in ds.batches(32):
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(model(b["h"], b["x"], b["edges"], b["vel"], b["edge_attr"], b["charges"]),
                                                b["target"])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-2:]) < np.mean(losses[:2])


@pytest.mark.gpu
@pytest.mark.parametrize("norm", ["default", "same_data_norm", "vel_norm_norm", "no_data_norm", "symmetric_data_norm"])
def test_simulated_field_dataset_and_forward_prediction_metric(norm):
    """aether_amd.data.SimulatedFieldDataset (the seq2seq runners' data set over device-simulated trajectories, all four
    normalisation modes + the symmetric variant) and aether_amd.evaluate.eval_forward_prediction (evaluate.py:14-79):
    normalise / un-normalise round trip, training-set statistics reused for another split, and the 20-step MSE of a
    model equals the same metric computed from the oracle's predictions."""
    import torch
    from aether_amd.data import SimulatedFieldDataset
    from aether_amd.evaluate import eval_forward_prediction
    from aether_amd.nn.seq2seq.aether import Aether
    from oracle import seq2seq_oracle as S
    params = {} if norm == "default" else {norm: True}
    train = SimulatedFieldDataset(24, params, n_balls=5, static_balls=8, ndim=2, length=5000, sample_freq=100)
    test = SimulatedFieldDataset(6, params, n_balls=5, static_balls=8, ndim=2, length=5000, sample_freq=100,
                                 particle_seed=3, stats_from=train)
    assert train.feats.shape == (24, 49, 5, 4) and train.feats.is_cuda and torch.isfinite(train.feats).all()
    assert torch.equal(train.static_field, test.static_field)                       # the same field for every split
    back = train.torch_unnormalize(train.feats)
    assert torch.allclose(back, train._raw, rtol=1e-5, atol=1e-5)
    if norm in ("default", "same_data_norm", "symmetric_data_norm"):
        assert float(train.feats.max()) <= 1.0 + 1e-6 and float(train.feats.min()) >= -1.0 - 1e-6
    item = test[2]
    assert item["inputs"].shape == (49, 5, 4) and item["edges"].shape == (5, 5) and item["charges"].shape == (5,)
    H = 128
    mp = {"num_vars": 5, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H, "encoder_rnn_hidden": 64,
          "encoder_rnn_type": "lstm", "input_size": 4, "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 64,
          "prior_num_layers": 3, "prior_hidden_size": 64, "use_3d": False, "pos_representation": "polar", "gpu": True,
          "decoder_hidden": H, "skip_first": False, "decoder_dropout": 0.0, "gumbel_temp": 0.5}
    torch.manual_seed(1)
    model = Aether(mp, device="cuda").eval()
    g = torch.Generator().manual_seed(2)
    U = torch.rand(28 + 20, 6 * 20, 2, generator=g)
    mse, pos_mse, vel_mse = eval_forward_prediction(model, test, 29, 20, batch_size=6,
                                                    uniform=U.cuda().view(-1, 6, 20, 2))
    assert mse.shape == (20,) and torch.isfinite(mse).all()
    assert torch.allclose(mse, 0.5 * (pos_mse + vel_mse), rtol=1e-5)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want = S.predict_future(sd, test.feats[:, :29].cpu(), 20, U, 0.5, False, "polar", 3)
    p, gt = test.torch_unnormalize(want.cuda()), test.torch_unnormalize(test.feats[:, 29:49])
    want_mse = ((p - gt) ** 2).flatten(2).mean(-1).mean(0)
    assert ((mse - want_mse).abs() / want_mse).max() <= 1e-4
