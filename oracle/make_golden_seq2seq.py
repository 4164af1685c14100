#!/usr/bin/env python3
"""Golden fixtures for the seq2seq field query from the imported reference.

TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference is mounted.  Imported,
unmodified: ``nn.nn.fourier_feature_mapper.FourierFeatureMapper`` (its ``B`` buffer and forward).
``field_net`` is the plain ``torch.nn.Sequential`` the reference builds inline in
``nn/seq2seq/aether.py:72-78`` (constructing the whole seq2seq ``Aether`` needs a params dictionary of
~40 flags and torch_scatter; the five-layer Sequential itself is generic torch) -- created here under
``torch.manual_seed(SEED)`` in the same order, so the tests can recreate the identical parameters
from the seed; a checksum of every tensor is stored to detect drift.  Positions are a batch of
trajectories ``[B, N, T, 2D]`` as ``predict_field`` receives them in ``predict_future`` (:161-162).

Usage:  python oracle/make_golden_seq2seq.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AETHER_REFERENCE", "/root/reference")
SEED, HIDDEN = 1234, 512


def build_reference_field(num_dims: int):
    sys.path.insert(0, REF)
    from nn.nn.fourier_feature_mapper import FourierFeatureMapper          # the reference class
    torch.manual_seed(SEED)
    field_net = nn.Sequential(nn.Linear(HIDDEN, HIDDEN), nn.SiLU(), nn.Linear(HIDDEN, HIDDEN), nn.SiLU(),
                              nn.Linear(HIDDEN, num_dims))                   # aether.py:72-78
    emb = FourierFeatureMapper(num_dims, HIDDEN // 2, std=1.0)              # aether.py:83-84
    return field_net, emb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for D in (2, 3):
        field_net, emb = build_reference_field(D)
        g = torch.Generator().manual_seed(77 + D)
        x = torch.randn(3, 5, 7, 2 * D, generator=g) * 1.5                   # [B, N, T, pos | vel]
        with torch.no_grad():
            coords = x[..., :D]                                               # aether.py:87
            rff = emb(coords)                                                 # :88
            field = field_net(rff)                                            # :89
            field64 = field_net.double()(emb.double()(coords.double()))
        sd = {"coordinate_embedding.B": emb.B}
        sd.update({"field_net." + k: v for k, v in field_net.float().state_dict().items()})
        out = {"in.x": x.numpy(), "ref.rff": rff.numpy(), "ref.field": field.numpy(),
               "ref64.field": field64.numpy(), "B": emb.B.numpy(), "seed": np.int64(SEED),
               "hidden": np.int64(HIDDEN)}
        for k, v in sd.items():
            out["sum." + k] = np.float64(v.double().sum().item())
            out["abs." + k] = np.float64(v.double().abs().sum().item())
        np.savez(os.path.join(args.out, f"s2s_field_D{D}.npz"), **out)
        print("wrote s2s_field_D%d.npz" % D, {k: tuple(v.shape) for k, v in sd.items()})
    return args.out


def localizer_fixtures(out_dir):
    """Row A9: the imported reference AugmentedLocalizer on random states and on degenerate ones."""
    sys.path.insert(0, REF)
    from nn.utils.augmented_global_to_local import AugmentedLocalizer
    for use_3d in (False, True):
        D = 3 if use_3d else 2
        for rep in ("polar", "cart"):
            g = torch.Generator().manual_seed(300 + D)
            B, N = 3, 6
            x = torch.randn(B, N, 3 * D, generator=g)
            x[..., :D] *= 2.0
            # degenerate rows: axis-aligned / opposite velocities (angle wraps), a tiny velocity
            x[0, 0, D:2 * D] = torch.tensor([1.0, 0.0, 0.0][:D])
            x[0, 1, D:2 * D] = torch.tensor([-1.0, 0.0, 0.0][:D])
            x[0, 2, D:2 * D] = torch.tensor([-1.0, -1e-8, 0.0][:D])
            x[1, 0, D:2 * D] = 1e-6 * x[1, 0, D:2 * D]
            loc = AugmentedLocalizer(N, use_3d=use_3d, pos_representation=rep)
            with torch.no_grad():
                rel_feat, Rinv, edge_attr, edge_pos = loc(x)
                r64 = [t.numpy() for t in AugmentedLocalizer(N, use_3d=use_3d, pos_representation=rep)(x.double())]
            np.savez(os.path.join(out_dir, f"s2s_localizer_D{D}_{rep}.npz"), **{
                "in.x": x.numpy(), "ref.rel_feat": rel_feat.numpy(), "ref.Rinv": Rinv.numpy(),
                "ref.edge_attr": edge_attr.numpy(), "ref.edge_pos": edge_pos.numpy(),
                "ref64.rel_feat": r64[0], "ref64.edge_attr": r64[2],
                "send": loc.send_edges.numpy(), "recv": loc.recv_edges.numpy()})
            print("wrote s2s_localizer_D%d_%s.npz" % (D, rep), tuple(rel_feat.shape), tuple(edge_attr.shape),
                  tuple(edge_pos.shape))


def decoder_fixtures(out_dir):
    """Row A10 (decoder half): the imported reference RecurrentDecoder, one step.  The class imports
    torch_scatter (not installed: the same stand-in as oracle/make_golden.py is placed in sys.modules)
    and calls ``.cuda()`` on its receiver index (aether.py:617,635): on this GPU-less box the method is
    replaced by the identity for the duration of the call.  Parameters come from the class's own
    constructor under ``torch.manual_seed(DEC_SEED)``; the tests recreate them with the drop-in module,
    whose constructor creates the same tensors in the same order (checksums stored).  (No fp64 run: the
    reference allocates fp32 accumulators inside forward.)"""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import contextlib, io
    import make_golden as MG
    MG._install_scatter_standin()
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.seq2seq.aether import RecurrentDecoder
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for use_3d in (False, True):
            D = 3 if use_3d else 2
            B, N, H = 2, 5, 512
            params = {"num_vars": N, "input_size": 2 * D, "gpu": False, "decoder_hidden": H, "num_edge_types": 2,
                      "skip_first": False, "decoder_dropout": 0.0, "use_3d": use_3d}
            torch.manual_seed(DEC_SEED)
            with contextlib.redirect_stdout(io.StringIO()):
                dec = RecurrentDecoder(params).eval()
            g = torch.Generator().manual_seed(500 + D)
            inputs = torch.randn(B, N, 2 * D, generator=g)
            hidden = torch.randn(B, N, H, generator=g) * 0.5
            field = torch.randn(B, N, D, generator=g) * 0.3
            E = N * (N - 1)
            hard = torch.nn.functional.one_hot(torch.randint(0, 2, (B, E), generator=g), 2).float()
            soft = torch.softmax(torch.randn(B, E, 2, generator=g), -1)
            out = {"in.inputs": inputs.numpy(), "in.hidden": hidden.numpy(), "in.field": field.numpy(),
                   "in.edges_hard": hard.numpy(), "in.edges_soft": soft.numpy(), "seed": np.int64(DEC_SEED),
                   "hidden_size": np.int64(H), "num_vars": np.int64(N)}
            with torch.no_grad():
                for name, z in (("hard", hard), ("soft", soft)):
                    o, h2 = dec(inputs, hidden, z, field)
                    out[f"ref.{name}.outputs"], out[f"ref.{name}.hidden"] = o.numpy(), h2.numpy()
            for k, v in dec.state_dict().items():
                out["sum." + k] = np.float64(v.double().sum().item())
                out["abs." + k] = np.float64(v.double().abs().sum().item())
            out["keys"] = np.array(list(dec.state_dict().keys()))
            np.savez(os.path.join(out_dir, f"s2s_decoder_D{D}.npz"), **out)
            print("wrote s2s_decoder_D%d.npz" % D, len(dec.state_dict()), "tensors")
    finally:
        torch.Tensor.cuda = orig_cuda


def prior_fixtures(out_dir):
    """Row A10 (prior half): the imported reference Encoder.single_step_forward (eval mode; BatchNorm
    running statistics set to non-trivial values) + gumbel_softmax(hard) with the uniform noise captured."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import contextlib, io
    import make_golden as MG
    MG._install_scatter_standin()
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.seq2seq.aether import Encoder
        from nn.utils import model_utils
    for use_3d in (False, True):
        D = 3 if use_3d else 2
        B, N, H, R = 2, 5, 512, 128
        params = enc_params(N, D, H, R)
        torch.manual_seed(ENC_SEED)
        with contextlib.redirect_stdout(io.StringIO()):
            enc = Encoder(params).eval()
        g = torch.Generator().manual_seed(700 + D)
        with torch.no_grad():
            for bn in (enc.mlp3.bn, enc.mlp4.bn):                  # as after training: non-trivial statistics
                bn.running_mean.copy_(torch.randn(H, generator=g) * 0.2)
                bn.running_var.copy_(torch.rand(H, generator=g) + 0.5)
                bn.weight.copy_(1.0 + 0.1 * torch.randn(H, generator=g))
                bn.bias.copy_(0.1 * torch.randn(H, generator=g))
        E = N * (N - 1)
        inputs = torch.randn(B, N, 2 * D, generator=g)
        field = torch.randn(B, N, D, generator=g) * 0.3
        h0, c0 = torch.randn(B, E, R, generator=g) * 0.3, torch.randn(B, E, R, generator=g) * 0.3
        with torch.no_grad():
            logits, (h1, c1) = enc.single_step_forward(inputs, (h0, c0), field)
            # gumbel_softmax draws torch.rand on the CPU (model_utils.py:66): capture the draw
            torch.manual_seed(99)
            U = torch.rand(B * E, 2)
            torch.manual_seed(99)
            edges = model_utils.gumbel_softmax(logits.reshape(-1, 2), tau=0.5, hard=True).view(B, E, 2)
        out = {"in.inputs": inputs.numpy(), "in.field": field.numpy(), "in.h0": h0.numpy(), "in.c0": c0.numpy(),
               "in.uniform": U.numpy(), "ref.logits": logits.numpy(), "ref.h1": h1.numpy(), "ref.c1": c1.numpy(),
               "ref.edges": edges.numpy(), "tau": np.float64(0.5), "seed": np.int64(ENC_SEED),
               "hidden_size": np.int64(H), "rnn_hidden": np.int64(R), "num_vars": np.int64(N)}
        for k in ("mlp3.bn", "mlp4.bn"):
            for t in ("running_mean", "running_var", "weight", "bias"):
                out[f"bn.{k}.{t}"] = enc.state_dict()[f"{k}.{t}"].numpy()
        for k, v in enc.state_dict().items():
            if v.dtype.is_floating_point and ".bn." not in k:
                out["sum." + k] = np.float64(v.double().sum().item())
                out["abs." + k] = np.float64(v.double().abs().sum().item())
        out["keys"] = np.array(list(enc.state_dict().keys()))
        np.savez(os.path.join(out_dir, f"s2s_prior_D{D}.npz"), **out)
        print("wrote s2s_prior_D%d.npz" % D, len(enc.state_dict()), "tensors")


def future_fixture(out_dir):
    """End to end: the imported reference seq2seq ``Aether.predict_future`` (burn-in + prediction loop).
    gumbel_softmax draws torch.rand on the host: the global generator is seeded before the call and the
    same draws are regenerated afterwards for the fixture.  The seed is accepted only if every sampled
    edge type wins its Gumbel race by a clear margin, so that fp32 reordering cannot flip a sample."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import contextlib, io
    import make_golden as MG
    import seq2seq_oracle as S
    MG._install_scatter_standin()
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.seq2seq.aether import Aether
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        D, B, N, T, steps, H, R = 2, 2, 5, 4, 3, 128, 64
        params = dict(enc_params(N, D, H, R))
        params.update({"gpu": False, "decoder_hidden": H, "skip_first": False, "decoder_dropout": 0.0,
                       "gumbel_temp": 0.5, "encoder_mlp_hidden": 64, "prior_hidden_size": 64, "rff_std": 1.0})
        torch.manual_seed(FUT_SEED)
        with contextlib.redirect_stdout(io.StringIO()):
            model = Aether(params).eval()
        g = torch.Generator().manual_seed(900)
        inputs = torch.randn(B, T, N, 2 * D, generator=g)
        E = N * (N - 1)
        for noise_seed in range(50):
            torch.manual_seed(1000 + noise_seed)
            with torch.no_grad():
                preds, edges = model.predict_future(inputs, steps, return_edges=True)
            torch.manual_seed(1000 + noise_seed)
            U = torch.stack([torch.rand(B * E, 2) for _ in range(T - 1 + steps)])
            sd = {k: v.detach() for k, v in model.state_dict().items()}
            o_preds, o_edges = S.predict_future(sd, inputs, steps, U, 0.5, False, "polar", 3, return_edges=True)
            if torch.equal(o_edges.argmax(-1), edges.argmax(-1)):
                break
        else:
            raise RuntimeError("no noise seed with unambiguous samples")
        out = {"in.inputs": inputs.numpy(), "in.uniform": U.numpy(), "ref.predictions": preds.numpy(),
               "ref.edges": edges.numpy(), "seed": np.int64(FUT_SEED), "steps": np.int64(steps),
               "hidden_size": np.int64(H), "rnn_hidden": np.int64(R), "num_vars": np.int64(N)}
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                out["sum." + k] = np.float64(v.double().sum().item())
                out["abs." + k] = np.float64(v.double().abs().sum().item())
        out["keys"] = np.array(list(model.state_dict().keys()))
        np.savez(os.path.join(out_dir, "s2s_future_D2.npz"), **out)
        print("wrote s2s_future_D2.npz", tuple(preds.shape), "noise seed", 1000 + noise_seed,
              "oracle err", float((o_preds - preds).abs().max() / preds.abs().max()))
    finally:
        torch.Tensor.cuda = orig_cuda


def dynfield_future_fixture(out_dir):
    """SURVEY 8f N3, seq2seq half: the imported reference ``nn.seq2seq.dynamic_field_aether.DynamicFieldAether``
    (3-D, the gravitational model): ``graph_pooler`` (GraphSummary), ``predict_field`` (FilmedNetwork) and
    ``predict_future``.  Stand-ins as in oracle/make_golden_dynfield.py (torch_scatter,
    torch_geometric AttentionalAggregation) and the identity ``.cuda()``; ``params['field']`` (a data-side
    object used only to draw visualisation grids, dynamic_field_aether.py:88-95) is None."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import contextlib, io
    import make_golden as MG
    import make_golden_dynfield as MGD
    import seq2seq_oracle as S
    MG._install_scatter_standin()
    MGD.install_pyg_standin()
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.seq2seq.dynamic_field_aether import DynamicFieldAether
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        D, B, N, T, steps, H, R, GH, MH = 3, 2, 5, 5, 3, 128, 64, 64, 96
        params = dict(enc_params(N, D, H, R))
        params.update({"gpu": False, "decoder_hidden": H, "skip_first": False, "decoder_dropout": 0.0,
                       "gumbel_temp": 0.5, "encoder_mlp_hidden": 64, "prior_hidden_size": 64, "rff_std": 1.0,
                       "pos_representation": "cart", "graph_hidden": GH, "mlp_hidden": MH, "field": None})
        torch.manual_seed(DYN_SEED)
        with contextlib.redirect_stdout(io.StringIO()):
            model = DynamicFieldAether(params).eval()
        g = torch.Generator().manual_seed(901)
        inputs = torch.randn(B, T, N, 2 * D, generator=g)
        E = N * (N - 1)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        with torch.no_grad():
            x = inputs[:, :-1].transpose(2, 1).contiguous()                  # dynamic_field_aether.py:214
            summary = model.graph_pooler(x)                                   # :218
            field0, _ = model.predict_field(x, summary, None)                 # :219
            m64 = model.double()
            summary64 = m64.graph_pooler(x.double())
            field64, _ = m64.predict_field(x.double(), summary64, None)
            model.float()
        for noise_seed in range(50):
            torch.manual_seed(2000 + noise_seed)
            with torch.no_grad():
                preds, edges = model.predict_future(inputs, steps, return_edges=True)
            torch.manual_seed(2000 + noise_seed)
            U = torch.stack([torch.rand(B * E, 2) for _ in range(T - 1 + steps)])
            o_preds, o_edges = S.predict_future_dynamic_field(sd, inputs, steps, U, 0.5, True, "cart", 3,
                                                              return_edges=True)
            if torch.equal(o_edges.argmax(-1), edges.argmax(-1)):
                break
        else:
            raise RuntimeError("no noise seed with unambiguous samples")
        out = {"in.inputs": inputs.numpy(), "in.uniform": U.numpy(), "ref.predictions": preds.numpy(),
               "ref.edges": edges.numpy(), "ref.summary": summary.numpy(), "ref.field": field0.numpy(),
               "ref64.summary": summary64.numpy(), "ref64.field": field64.numpy(),
               "seed": np.int64(DYN_SEED), "steps": np.int64(steps), "hidden_size": np.int64(H),
               "rnn_hidden": np.int64(R), "num_vars": np.int64(N), "graph_hidden": np.int64(GH),
               "mlp_hidden": np.int64(MH)}
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                out["sum." + k] = np.float64(v.double().sum().item())
                out["abs." + k] = np.float64(v.double().abs().sum().item())
        out["keys"] = np.array(list(model.state_dict().keys()))
        np.savez(os.path.join(out_dir, "s2s_dynfield_D3.npz"), **out)
        gp = {k[len("graph_pooler."):]: v for k, v in sd.items() if k.startswith("graph_pooler.")}
        o_sum = S.graph_summary(gp, x)
        print("wrote s2s_dynfield_D3.npz", tuple(preds.shape), "noise seed", 2000 + noise_seed,
              "oracle err: summary", float((o_sum - summary).abs().max() / summary.abs().max()),
              "field", float((S.film_field(sd, x, summary, D) - field0).abs().max() / field0.abs().max()),
              "future", float((o_preds - preds).abs().max() / preds.abs().max()))
    finally:
        torch.Tensor.cuda = orig_cuda


DYN_SEED = 9753

LOSS_SEED = 1111
LOSS_CONFIGS = {   # name: extra params of the loss (aether.py:27-58)
    "gaussian_norm": {"nll_loss_type": "gaussian", "prior_variance": 5e-5, "normalize_nll": True, "normalize_kl": True,
                      "kl_coef": 1.0, "val_teacher_forcing_steps": -1},
    "crossent_tf2_uniform": {"nll_loss_type": "crossent", "kl_coef": 0.5, "val_teacher_forcing_steps": 2,
                             "add_uniform_prior": True, "no_edge_prior": 0.7, "normalize_kl_per_var": True},
}


def loss_params(name):
    D, N, H, R = 2, 5, 128, 64
    params = dict(enc_params(N, D, H, R))
    params.update({"gpu": False, "decoder_hidden": H, "skip_first": False, "decoder_dropout": 0.0, "gumbel_temp": 0.5,
                   "encoder_mlp_hidden": 64, "prior_hidden_size": 64, "rff_std": 1.0})
    params.update(LOSS_CONFIGS[name])
    return params


def loss_fixture(out_dir):
    """The imported reference ``Encoder.forward`` (full sequence: forward + reverse LSTM, both heads) and
    ``Aether.calculate_loss(is_train=False)`` for two loss configurations; Gumbel draws regenerated as in
    ``future_fixture``."""
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import contextlib, io
    import make_golden as MG
    import seq2seq_oracle as S
    MG._install_scatter_standin()
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.seq2seq.aether import Aether
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        B, T, N, D = 3, 6, 5, 2
        E = N * (N - 1)
        g = torch.Generator().manual_seed(902)
        inputs = torch.randn(B, T, N, 2 * D, generator=g)
        out["in.inputs"] = inputs.numpy()
        for name in LOSS_CONFIGS:
            params = loss_params(name)
            torch.manual_seed(LOSS_SEED)
            with contextlib.redirect_stdout(io.StringIO()):
                model = Aether(params).eval()
            for noise_seed in range(50):
                torch.manual_seed(3000 + noise_seed)
                with torch.no_grad():
                    loss, nll, kl, post, preds = model.calculate_loss(inputs, is_train=False, return_logits=True)
                torch.manual_seed(3000 + noise_seed)
                U = torch.stack([torch.rand(B * E, 2) for _ in range(T - 1)])
                sd = {k: v.detach() for k, v in model.state_dict().items()}
                o = S.calculate_loss_eval(sd, params, inputs, U, False, "polar")
                if float((o[4] - preds).abs().max()) <= 1e-5:            # no sample flipped between the two evaluations
                    break
            else:
                raise RuntimeError("no noise seed with unambiguous samples")
            with torch.no_grad():
                x = inputs[:, :-1].transpose(2, 1).contiguous()
                field, _ = model.predict_field(x)
                prior, post2, state = model.encoder(inputs[:, :-1], field)
            assert torch.equal(post2, post)
            for k, v in (("uniform", U), ("loss", loss), ("nll", nll), ("kl", kl), ("posterior", post), ("predictions", preds),
                         ("prior", prior), ("field", field), ("state.h", state[0]), ("state.c", state[1])):
                out[f"{name}.{k}"] = v.detach().numpy()
            if name == list(LOSS_CONFIGS)[0]:
                for k, v in model.state_dict().items():
                    if v.dtype.is_floating_point:
                        out["sum." + k] = np.float64(v.double().sum().item())
                        out["abs." + k] = np.float64(v.double().abs().sum().item())
                out["keys"] = np.array(list(model.state_dict().keys()))
            print(name, "loss", float(loss), "oracle", float(o[0]), "noise seed", 3000 + noise_seed)
        out["seed"] = np.int64(LOSS_SEED)
        np.savez(os.path.join(out_dir, "s2s_loss_D2.npz"), **out)
        print("wrote s2s_loss_D2.npz")
    finally:
        torch.Tensor.cuda = orig_cuda


FUT_SEED = 1357


def enc_params(N, D, H, R):
    return {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H,
            "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "input_size": 2 * D,
            "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256, "prior_num_layers": 3,
            "prior_hidden_size": 256, "use_3d": D == 3, "pos_representation": "polar"}


DEC_SEED = 4321
ENC_SEED = 2468

if __name__ == "__main__":
    out_dir = main()                     # every fixture goes where --out says (a dry run must not touch tests/golden)
    localizer_fixtures(out_dir)
    decoder_fixtures(out_dir)
    prior_fixtures(out_dir)
    future_fixture(out_dir)
    dynfield_future_fixture(out_dir)
    loss_fixture(out_dir)
