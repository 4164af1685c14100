"""Pin the CPU oracle against vectors produced by the imported reference."""
import math

import numpy as np
import pytest
import torch

import os

from conftest import CASES, GOLDEN, GRAD_CASES, load_case, load_state_dict, scale_rel_err
from oracle import aether_oracle as O

STAGES = ["field", "rel_feat", "R", "edge_attr_local", "x1", "e1", "x2", "e2", "x3", "e3",
          "x4", "e4", "pred_local", "pred_global", "out"]


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("case", CASES)
def test_every_stage_matches_reference(D, case):
    torch.set_num_threads(1)
    inp, ref, ref64, meta = load_case(f"case_D{D}_{case}.npz")
    sd = load_state_dict(D)
    got = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"],
                           inp["charges"], return_all=True)
    for k in STAGES:
        assert got[k].shape == ref[k].shape, k
        err = scale_rel_err(got[k], ref[k])
        assert err <= 1e-6, (k, err)


@pytest.mark.parametrize("D", [2, 3])
def test_fp64_oracle_matches_fp64_reference(D):
    inp, ref, ref64, meta = load_case(f"case_D{D}_B2N20.npz")
    sd = {k: v.double() for k, v in load_state_dict(D).items()}
    got = O.aether_forward(sd, inp["x"].double(), inp["vel"].double(), inp["edges"],
                           inp["edge_attr"].double(), inp["charges"].double(), return_all=True)
    for k in ("field", "e3", "x4", "out"):
        assert scale_rel_err(got[k], ref64[k]) <= 1e-12, k
    # fp32 reference sits at ~1e-7 of its own fp64 evaluation (noise floor)
    assert scale_rel_err(ref["out"], ref64["out"]) <= 1e-6


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("case", GRAD_CASES)
def test_parameter_gradients_match_reference(D, case):
    torch.set_num_threads(1)
    inp, ref, ref64, meta = load_case(f"case_D{D}_{case}.npz")
    sd = {k: v.clone().requires_grad_(True) for k, v in load_state_dict(D).items()}
    out = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    loss = torch.nn.functional.mse_loss(out, inp["target"])
    loss.backward()
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 1e-6 * abs(float(ref["loss"]))
    for k, p in sd.items():
        err = scale_rel_err(p.grad, ref["grad." + k])
        assert err <= 2e-5, (k, err)



@pytest.mark.parametrize("tag", ["dropout", "inputgrad"])
@pytest.mark.parametrize("D", [2, 3])
def test_dropout_masks_and_input_gradients_match_reference(D, tag):
    """Fixtures of oracle/make_golden_dropout.py: the reference itself in train() mode with dropout_prob = 0.25 (the masks its
    two nn.Dropout layers drew, captured by hooks) and with its inputs as autograd leaves.  The oracle with those masks
    reproduces the reference's output, its 47 parameter gradients and d/dx, d/dvel, d/dedge_attr."""
    d = np.load(os.path.join(GOLDEN, f"case_D{D}_{tag}.npz"))
    inp, ref, _, meta = load_case(f"case_D{D}_{tag}.npz")
    masks = None
    if float(d["dropout_prob"][0]) > 0:
        masks = [torch.from_numpy(d["mask1"]), torch.from_numpy(d["mask2"])]
        keep = 1.0 - float(d["dropout_prob"][0])
        for m in masks:
            assert set(torch.unique(m).tolist()) <= {0.0, float(np.float32(1.0) / np.float32(keep))}
    sd = {k: v.clone().requires_grad_(True) for k, v in load_state_dict(D).items()}
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
    out = O.aether_forward(sd, leaves["x"], leaves["vel"], inp["edges"], leaves["edge_attr"], inp["charges"],
                           dropout_masks=masks)
    assert scale_rel_err(out.detach(), ref["out"]) <= 1e-5
    torch.nn.functional.mse_loss(out, inp["target"]).backward()
    for k, v in sd.items():
        assert scale_rel_err(v.grad, ref["grad." + k]) <= 2e-5, k
    for k, v in leaves.items():
        assert scale_rel_err(v.grad, ref["grad_in." + k]) <= 2e-5, k

@pytest.mark.parametrize("D", [2, 3])
def test_degenerate_inputs_finite_and_wrap_aware(D):
    """Zero velocity, coincident particles, anti-parallel headings, v || +-z.

    At these inputs atan2/acos sit on branch cuts, so a 1-ulp difference may
    flip an angle by 2*pi; compare angle columns modulo the wrap."""
    inp, ref, ref64, meta = load_case(f"case_D{D}_edge_B2N5.npz")
    sd = load_state_dict(D)
    got = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"],
                           inp["charges"], return_all=True)
    for k in STAGES:
        assert torch.isfinite(got[k]).all(), k
    assert scale_rel_err(got["field"], ref["field"]) <= 1e-6
    a, b = got["edge_attr_local"].double(), ref["edge_attr_local"].double()
    diff = (a - b).abs()
    n_orient = D * (D - 1) // 2
    for c in range(D, D + n_orient):                 # euler/pi columns wrap at +-1
        diff[:, c] = torch.minimum(diff[:, c], (2.0 - diff[:, c]).abs())
    c_theta = D + n_orient + 1                       # symmetric theta wraps at +-pi
    diff[:, c_theta] = torch.minimum(diff[:, c_theta], (2 * math.pi - diff[:, c_theta]).abs())
    assert float(diff.max()) <= 2e-3


@pytest.mark.parametrize("D", [2, 3])
def test_full_size_config_output(D):
    """cfg2 / cfg3 (B=128, N=20): output of the reference at full size."""
    from aether_amd.edges import get_edges, prepare_edge_attr
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, f"full_D{D}_B128N20.npz"))
    x, vel, q = (torch.from_numpy(d[k]) for k in ("in.x", "in.vel", "in.charges"))
    edges = get_edges(128, 20)
    ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
    sd = load_state_dict(D)
    out = O.aether_forward(sd, x, vel, edges, ea, q)
    assert scale_rel_err(out, torch.from_numpy(d["ref.out"])) <= 1e-6
    assert scale_rel_err(out, torch.from_numpy(d["ref64.out"])) <= 1e-6


@pytest.mark.parametrize("D", [2, 3])
def test_rollout_20_steps(D):
    import os
    from conftest import GOLDEN
    from aether_amd.edges import get_edges
    d = np.load(os.path.join(GOLDEN, f"rollout_D{D}_B4N5.npz"))
    x, vel, q = (torch.from_numpy(d[k]) for k in ("in.x", "in.vel", "in.charges"))
    edges = get_edges(4, 5)
    traj = O.rollout(load_state_dict(D), x, vel, edges, q, 20)
    ref = torch.from_numpy(d["ref.traj"])
    assert traj.shape == ref.shape
    assert scale_rel_err(traj, ref) <= 1e-5


def test_scatter_mean_against_dense_one_hot():
    """Independent check of the third-party scatter semantics."""
    g = torch.Generator().manual_seed(0)
    e = torch.randn(50, 8, generator=g, dtype=torch.float64)
    recv = torch.randint(0, 9, (50,), generator=g)
    recv[recv == 3] = 2                                  # row 3 has no in-edges
    n = 10
    onehot = torch.nn.functional.one_hot(recv, n).double()
    deg = onehot.sum(0).clamp(min=1)
    dense = (onehot.t() @ e) / deg[:, None]
    got = O.scatter_mean(e, recv, n)
    assert torch.allclose(got, dense, atol=1e-14)
    assert float(got[3].abs().max()) == 0.0 and float(got[9].abs().max()) == 0.0


def test_euler_zyx_known_answer():
    """geometry.py:79-86 claims PyTorch3D ZYX equivalence: R = Rz(a) Ry(b) Rx(c)."""
    a, b, c = 0.3, -0.7, 1.1
    Rz = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    Ry = torch.tensor([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(c), -math.sin(c)], [0, math.sin(c), math.cos(c)]])
    e = O.euler_from_matrix((Rz @ Ry @ Rx).double()[None], 3)[0] * math.pi
    assert torch.allclose(e, torch.tensor([a, b, c], dtype=torch.float64), atol=1e-6)


@pytest.mark.parametrize("D", [2, 3])
def test_frame_maps_axis_to_heading(D):
    """R e1 = v/|v| in 2-D; R e3 ~ v/|v| in 3-D (geometry.py:16-33)."""
    g = torch.Generator().manual_seed(1)
    v = torch.randn(64, D, generator=g, dtype=torch.float64)
    R = O.frame_from_velocity(v)
    axis = R[..., :, 0] if D == 2 else R[..., :, 2]
    # 3-D: acos(z/(rho+1e-7)) carries the reference's EPS, so ~1e-7/rho_min slack
    assert torch.allclose(axis, v / v.norm(dim=-1, keepdim=True), atol=1e-5)


@pytest.mark.parametrize("D", [2, 3])
def test_seq2seq_field(D):
    """Row A8: Fourier features + field MLP of the seq2seq model vs the imported reference."""
    from conftest import load_s2s_field
    from oracle import seq2seq_oracle as S
    d, sd = load_s2s_field(D)
    x = torch.from_numpy(d["in.x"])
    assert torch.equal(S.rff_matrix(D, int(d["hidden"]) // 2), sd["coordinate_embedding.B"])
    rff = S.fourier_features(x[..., :D], sd["coordinate_embedding.B"])
    assert scale_rel_err(rff, torch.from_numpy(d["ref.rff"])) <= 1e-6
    field = S.predict_field(sd, x, D)
    assert scale_rel_err(field, torch.from_numpy(d["ref.field"])) <= 1e-6
    sd64 = {k: v.double() for k, v in sd.items()}
    assert scale_rel_err(S.predict_field(sd64, x.double(), D), torch.from_numpy(d["ref64.field"])) <= 1e-12


@pytest.mark.parametrize("rep", ["polar", "cart"])
@pytest.mark.parametrize("D", [2, 3])
def test_seq2seq_augmented_localizer(D, rep):
    """Row A9: augmented local frames (virtual origin node) vs the imported reference AugmentedLocalizer,
    including velocities on the angle branch cuts and a near-zero velocity."""
    from oracle import seq2seq_oracle as S
    d = np.load(os.path.join(GOLDEN, f"s2s_localizer_D{D}_{rep}.npz"))
    x = torch.from_numpy(d["in.x"])
    rel_feat, Rinv, edge_attr, edge_pos = S.augmented_localizer(x, D == 3, rep)
    for got, key in ((rel_feat, "ref.rel_feat"), (Rinv, "ref.Rinv"), (edge_attr, "ref.edge_attr"),
                     (edge_pos, "ref.edge_pos")):
        want = torch.from_numpy(d[key])
        assert got.shape == want.shape, key
        assert scale_rel_err(got, want) <= 1e-6, key
    r64 = S.augmented_localizer(x.double(), D == 3, rep)
    assert scale_rel_err(r64[0], torch.from_numpy(d["ref64.rel_feat"])) <= 1e-12
    assert scale_rel_err(r64[2], torch.from_numpy(d["ref64.edge_attr"])) <= 1e-12


@pytest.mark.parametrize("D", [2, 3])
def test_seq2seq_decoder_step(D):
    """Row A10 (decoder half): one RecurrentDecoder step vs the imported reference, hard (one-hot) and soft
    edge-type weights."""
    from conftest import load_s2s_decoder
    from oracle import seq2seq_oracle as S
    d, sd, _ = load_s2s_decoder(D)
    t = lambda k: torch.from_numpy(d[k])
    for name in ("hard", "soft"):
        out, hid = S.decoder_step(sd, t("in.inputs"), t("in.hidden"), t("in.edges_" + name), t("in.field"), D == 3)
        assert scale_rel_err(out, t(f"ref.{name}.outputs")) <= 2e-6, name
        assert scale_rel_err(hid, t(f"ref.{name}.hidden")) <= 2e-6, name


@pytest.mark.parametrize("D", [2, 3])
def test_seq2seq_prior_step(D):
    """Row A10 (prior half): Encoder.single_step_forward + hard Gumbel sample vs the imported reference."""
    from conftest import load_s2s_prior
    from oracle import seq2seq_oracle as S
    d, sd, params = load_s2s_prior(D)
    t = lambda k: torch.from_numpy(d[k])
    logits, (h1, c1) = S.prior_step(sd, t("in.inputs"), (t("in.h0"), t("in.c0")), t("in.field"), D == 3,
                                    params["pos_representation"], params["prior_num_layers"])
    assert scale_rel_err(logits, t("ref.logits")) <= 5e-6
    assert scale_rel_err(h1, t("ref.h1")) <= 5e-6 and scale_rel_err(c1, t("ref.c1")) <= 5e-6
    edges = S.gumbel_hard(t("ref.logits").reshape(-1, 2), t("in.uniform"), float(d["tau"])).view_as(t("ref.edges"))
    assert torch.equal(edges.argmax(-1), t("ref.edges").argmax(-1))
    assert scale_rel_err(edges, t("ref.edges")) <= 1e-6


def test_seq2seq_predict_future():
    """End to end: burn-in + prediction loop vs the imported reference's own Aether.predict_future."""
    from conftest import load_s2s_future
    from oracle import seq2seq_oracle as S
    d, model, params = load_s2s_future()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    t = lambda k: torch.from_numpy(d[k])
    preds, edges = S.predict_future(sd, t("in.inputs"), int(d["steps"]), t("in.uniform"), 0.5, False, "polar", 3,
                                    return_edges=True)
    assert torch.equal(edges.argmax(-1), t("ref.edges").argmax(-1))
    assert scale_rel_err(preds, t("ref.predictions")) <= 2e-6


def test_seq2seq_dynamic_field_variant():
    """SURVEY 8f N3, seq2seq half: GraphSummary, the FiLM field query and predict_future of the imported
    reference nn.seq2seq.dynamic_field_aether.DynamicFieldAether (3-D)."""
    from conftest import load_s2s_dynfield
    from oracle import seq2seq_oracle as S
    d, model, params = load_s2s_dynfield()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    t = lambda k: torch.from_numpy(d[k])
    x = t("in.inputs")[:, :-1].transpose(2, 1).contiguous()
    gp = {k[len("graph_pooler."):]: v for k, v in sd.items() if k.startswith("graph_pooler.")}
    summary = S.graph_summary(gp, x)
    assert scale_rel_err(summary, t("ref.summary")) <= 2e-6
    assert scale_rel_err(S.graph_summary({k: v.double() for k, v in gp.items()}, x.double()), t("ref64.summary")) <= 1e-12
    field = S.film_field(sd, x, t("ref.summary"), 3)
    assert field.shape == x.shape[:-1] + (3,) and scale_rel_err(field, t("ref.field")) <= 2e-6
    preds, edges = S.predict_future_dynamic_field(sd, t("in.inputs"), int(d["steps"]), t("in.uniform"), 0.5, True,
                                                  "cart", 3, return_edges=True)
    assert torch.equal(edges.argmax(-1), t("ref.edges").argmax(-1))
    assert scale_rel_err(preds, t("ref.predictions")) <= 2e-6


@pytest.mark.parametrize("D", [2, 3])
def test_dynamic_field_variant(D):
    """DynamicFieldAether (attention-pooled graph summary + FiLM field net) vs the imported
    reference (torch_geometric's AttentionalAggregation replaced by a stand-in with its published semantics)."""
    from aether_amd.edges import get_edges
    d = np.load(os.path.join(GOLDEN, f"dynfield_D{D}.npz"))
    sd = {str(k): torch.from_numpy(d["sd." + str(k)]) for k in d["keys"]}
    for name in ("small", "cfg"):
        B, N = int(d[f"{name}.B"]), int(d[f"{name}.N"])
        t = lambda k: torch.from_numpy(d[f"{name}.in.{k}"])
        field = O.dynamic_field(sd, t("x"), t("vel"), t("charges"), N)
        assert scale_rel_err(field, torch.from_numpy(d[f"{name}.ref.field"])) <= 1e-6
        sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
        f64 = O.dynamic_field(sd64, t("x").double(), t("vel").double(), t("charges").double(), N)
        assert scale_rel_err(f64, torch.from_numpy(d[f"{name}.ref64.field"])) <= 1e-12
        out = O.dynamic_field_aether_forward(sd, t("x"), t("vel"), get_edges(B, N), t("edge_attr"), t("charges"), N)
        assert scale_rel_err(out, torch.from_numpy(d[f"{name}.ref.out"])) <= 1e-6


def test_knn_edge_builder():
    """the kNN edge builder vs the imported reference Encoder.knn_edges / get_knn_graph_info
    (bit-exact indices; fixtures hold scenes whose neighbour distances are separated, see make_golden_knn.py)."""
    from oracle import knn_oracle as K
    d = np.load(os.path.join(GOLDEN, "knn_edges.npz"))
    for name in ("small", "scenes", "few", "wide", "flat"):
        s, r, n = K.knn_edges(d[name + ".x"], d[name + ".masks"], int(d[name + ".k"]))
        assert np.array_equal(s, d[name + ".send"]) and np.array_equal(r, d[name + ".recv"]), name
        assert np.array_equal(n, d[name + ".num"]), name
    s, r = K.knn_graph_info(d["info.x"], d["info.masks"])
    assert np.array_equal(s, d["info.send"]) and np.array_equal(r, d["info.recv"])


@pytest.mark.parametrize("name", ["full8", "tail6", "gaps", "knn20", "empty"])
def test_dynamicvars_decoder_step(name):
    """SURVEY 8f N2, second half: the variable-N decoder step vs the imported reference Decoder (graphs from the
    reference's get_knn_graph_info): all present, trailing / interior objects missing, a 19-object kNN scene whose
    edge2node_inds rows are not per-receiver groups, an empty scene."""
    from conftest import load_dyn_decoder
    from oracle import dynamicvars_oracle as DO
    c, dec, params = load_dyn_decoder(name)
    sd = {k: v.detach() for k, v in dec.state_dict().items()}
    gi = (c["send"], c["recv"], c["e2n"]) if "send" in c else None
    pred, hid = DO.decoder_step(sd, c["inputs"], c["hidden"], c["edges"], c["masks"], gi, c["field"],
                                params["skip_first"], params["pos_representation"])
    assert scale_rel_err(pred, c["ref.pred"]) <= 2e-6 and scale_rel_err(hid, c["ref.hidden"]) <= 2e-6


def test_dynamicvars_model_prediction_path():
    """field query, encoder prior step and the whole predict_future of the imported reference
    AetherDynamicVars (objects appearing / disappearing, own predictions fed back after the burn-in)."""
    from conftest import load_dyn_model
    from oracle import dynamicvars_oracle as DO
    d, c, model, params = load_dyn_model()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    t = lambda k: torch.from_numpy(d[k])
    field0 = DO.predict_field(sd, c["inputs"][:, 0], c["masks"][:, 0])
    assert scale_rel_err(field0, t("ref.field0")) <= 2e-6
    logits0, (h1, c1) = DO.encoder_single_step(enc, c["inputs"][:, 0], c["masks"][:, 0], c["node_inds"][0],
                                               c["graph_info"][0], (t("state0.h"), t("state0.c")), t("ref.field0"))
    assert scale_rel_err(logits0, t("ref.logits0")) <= 5e-6
    assert scale_rel_err(h1, t("ref.state1.h")) <= 5e-6 and scale_rel_err(c1, t("ref.state1.c")) <= 5e-6
    preds = DO.predict_future(sd, c["inputs"], c["masks"], c["node_inds"], c["graph_info"], c["burn"], c["uniform"], 0.5,
                              True, "cart")
    assert scale_rel_err(preds, t("ref.predictions")) <= 2e-6


@pytest.mark.parametrize("name", ["gaussian_norm", "crossent_tf2_uniform"])
def test_seq2seq_encoder_forward_and_eval_loss(name):
    """The full-sequence encoder (forward + reverse LSTM, both heads) and Aether.calculate_loss(is_train=False) of the
    imported reference: Gaussian NLL with per-batch normalisation, and cross-entropy NLL with two teacher-forced
    steps, a non-uniform edge prior and per-variable KL normalisation."""
    from conftest import load_s2s_loss
    from oracle import seq2seq_oracle as S
    c, model, params = load_s2s_loss(name)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    prior, post, (h, cc) = S.encoder_forward(enc, c["inputs"][:, :-1], c["field"], False, "polar")
    assert scale_rel_err(prior, c["prior"]) <= 5e-6 and scale_rel_err(post, c["posterior"]) <= 5e-6
    assert scale_rel_err(h.reshape(c["state.h"].shape), c["state.h"]) <= 5e-6
    loss, nll, kl, post2, preds = S.calculate_loss_eval(sd, params, c["inputs"], c["uniform"], False, "polar")
    assert scale_rel_err(preds, c["predictions"]) <= 5e-6
    assert abs(float(loss) - float(c["loss"])) <= 1e-5 * abs(float(c["loss"]))
    assert scale_rel_err(nll.reshape(c["nll"].shape), c["nll"]) <= 1e-5 and scale_rel_err(kl.reshape(c["kl"].shape), c["kl"]) <= 1e-5


def test_knn_oracle_against_an_independent_formulation():
    """The loop-based kNN oracle vs a vectorised numpy formulation (argsort of the masked distance matrix) on random
    scenes with presence masks: two independent statements of the same definition must agree index for index."""
    from oracle import knn_oracle as K
    rng = np.random.default_rng(7)
    for trial in range(20):
        S, N, k = int(rng.integers(1, 5)), int(rng.integers(2, 30)), int(rng.integers(1, 12))
        x = rng.normal(size=(S, N, 3)).astype(np.float32) * 10
        m = (rng.random((S, N)) < 0.7).astype(np.float32)
        send, recv, num = K.knn_edges(x, m, k)
        ws, wr, base = [], [], 0
        for s in range(S):
            idx = np.nonzero(m[s])[0]
            p = x[s, idx, :2].astype(np.float32)
            d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1, dtype=np.float32))
            np.fill_diagonal(d, np.inf)
            kk = min(k, N - 1, max(len(idx) - 1, 0))
            order = np.argsort(d, axis=1, kind="stable")[:, :kk]
            for a in range(len(idx)):
                for b in order[a]:
                    ws.append(base + a)
                    wr.append(base + int(b))
            base += len(idx)
        assert np.array_equal(send, np.asarray(ws, dtype=np.int64)) and np.array_equal(recv, np.asarray(wr, dtype=np.int64))
        assert int(np.sum(num)) == len(ws)


def test_simulator_oracles_conserve_what_physics_conserves():
    """Oracle sanity beyond the fixtures: without capped forces the electrostatic leap-frog conserves energy to the
    integrator's order; the gravitational kick-drift-kick keeps the total momentum at zero and conserves the softened
    energy."""
    from oracle import sim_oracle as SO
    rng = np.random.default_rng(3)
    loc0 = rng.normal(size=(4, 2)) * 2.0
    vel0 = rng.normal(size=(4, 2)) * 0.3
    q = np.array([1.0, -1.0, 1.0, -1.0])
    loc, vel, capped = SO.electrostatic_trajectory(loc0, vel0, q, 4, 2000, 100)
    assert capped == 0

    def energy(x, v):
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
        np.fill_diagonal(d, np.inf)
        return 0.5 * (v ** 2).sum() + 0.5 * (np.outer(q, q) / d).sum()
    e = np.array([energy(loc[t], vel[t]) for t in range(len(loc))])
    assert np.abs(e - e[0]).max() <= 2e-3 * max(1.0, np.abs(e[0]))
    pos0 = rng.normal(size=(6, 3))
    v0 = rng.normal(size=(6, 3))
    mass = np.ones((6, 1))
    v0 -= v0.mean(0)
    pos, vel, force = SO.gravitational_trajectory(pos0, v0, mass, 6, 2000, 100)
    assert np.abs((mass * vel[1:]).sum(1)).max() <= 1e-12
