import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The tests bind the in-tree libaether_hip.so.  If a fresh checkout has not been built yet,
    # build it once with hipcc (cross-compiles without a GPU); a missing compiler is an error
    # of the environment, not something to fall back from.
    from aether_amd import build as _b
    if not os.path.exists(_b.LIB):
        _b.build_library(force=True, verbose=False)


def load_case(name):
    """Load one golden fixture -> (inputs dict of torch tensors, ref dict, meta)."""
    d = np.load(os.path.join(GOLDEN, name))
    inp = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("in.")}
    if "send" in inp:
        inp["edges"] = [inp.pop("send"), inp.pop("recv")]
    ref = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("ref.")}
    ref64 = {k[6:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("ref64.")}
    B, N, D, seed = (int(v) for v in d["meta"])
    return inp, ref, ref64, dict(B=B, N=N, D=D, seed=seed)


def load_state_dict(D):
    d = np.load(os.path.join(GOLDEN, f"state_dict_D{D}.npz"))
    return {k: torch.from_numpy(d[k]) for k in d.files}


def scale_rel_err(a, b):
    """SURVEY.md 8(d) tolerance definition: max|a-b| / max|b|."""
    a, b = a.double(), b.double()
    denom = b.abs().max().clamp(min=1e-30)
    return float((a - b).abs().max() / denom)


CASES = ["B1N5", "B3N5", "B2N20", "B2N2", "sparse"]
GRAD_CASES = ["B3N5", "B2N20", "sparse"]


def load_s2s_field(D):
    """Golden fixture of the seq2seq field query (oracle/make_golden_seq2seq.py) + its parameters,
    recreated from the stored seed exactly as the generator made them (checksums verified)."""
    import numpy as _np
    import torch as _torch
    from torch import nn as _nn
    d = _np.load(os.path.join(GOLDEN, f"s2s_field_D{D}.npz"))
    hidden = int(d["hidden"])
    _torch.manual_seed(int(d["seed"]))
    net = _nn.Sequential(_nn.Linear(hidden, hidden), _nn.SiLU(), _nn.Linear(hidden, hidden), _nn.SiLU(),
                         _nn.Linear(hidden, D))
    sd = {"coordinate_embedding.B": _torch.from_numpy(d["B"])}
    sd.update({"field_net." + k: v.detach() for k, v in net.state_dict().items()})
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    return d, sd


def load_s2s_decoder(D):
    """Golden fixture of the seq2seq decoder step + the reference's parameters, recreated from the stored
    seed through the drop-in module's constructor (same tensors, same order; checksums verified)."""
    import numpy as _np
    import torch as _torch
    from aether_amd.nn.seq2seq.decoder import RecurrentDecoder
    d = _np.load(os.path.join(GOLDEN, f"s2s_decoder_D{D}.npz"))
    params = {"num_vars": int(d["num_vars"]), "input_size": 2 * D, "gpu": False,
              "decoder_hidden": int(d["hidden_size"]), "num_edge_types": 2, "skip_first": False,
              "decoder_dropout": 0.0, "use_3d": D == 3}
    _torch.manual_seed(int(d["seed"]))
    dec = RecurrentDecoder(params, device=None)
    sd = {k: v.detach() for k, v in dec.state_dict().items()}
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k, v in sd.items():
        assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    return d, sd, params


def load_s2s_prior(D):
    """Golden fixture of the seq2seq prior step + the reference Encoder's parameters, recreated from the
    stored seed through the drop-in module's constructor (checksums verified) with the fixture's
    BatchNorm statistics."""
    import numpy as _np
    import torch as _torch
    from aether_amd.nn.seq2seq.encoder import Encoder
    d = _np.load(os.path.join(GOLDEN, f"s2s_prior_D{D}.npz"))
    params = {"num_vars": int(d["num_vars"]), "num_edge_types": 2, "encoder_dropout": 0.0,
              "encoder_hidden": int(d["hidden_size"]), "encoder_rnn_hidden": int(d["rnn_hidden"]),
              "encoder_rnn_type": "lstm", "input_size": 2 * D, "encoder_mlp_num_layers": 3,
              "encoder_mlp_hidden": 256, "prior_num_layers": 3, "prior_hidden_size": 256, "use_3d": D == 3,
              "pos_representation": "polar"}
    _torch.manual_seed(int(d["seed"]))
    enc = Encoder(params, device=None)
    sd = enc.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k in ("mlp3.bn", "mlp4.bn"):
        for t in ("running_mean", "running_var", "weight", "bias"):
            sd[f"{k}.{t}"].copy_(_torch.from_numpy(d[f"bn.{k}.{t}"]))
    for k, v in sd.items():
        if "sum." + k in d:
            assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    return d, {k: v.detach() for k, v in sd.items()}, params


def load_s2s_future():
    """Golden fixture of the reference's own seq2seq ``predict_future`` + the model's parameters recreated
    from the stored seed through the drop-in ``Aether`` constructor (checksums verified)."""
    import numpy as _np
    import torch as _torch
    from aether_amd.nn.seq2seq.aether import Aether
    d = _np.load(os.path.join(GOLDEN, "s2s_future_D2.npz"))
    N, H, R = int(d["num_vars"]), int(d["hidden_size"]), int(d["rnn_hidden"])
    params = {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H,
              "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "input_size": 4, "encoder_mlp_num_layers": 3,
              "encoder_mlp_hidden": 64, "prior_num_layers": 3, "prior_hidden_size": 64, "use_3d": False,
              "pos_representation": "polar", "gpu": False, "decoder_hidden": H, "skip_first": False,
              "decoder_dropout": 0.0, "gumbel_temp": 0.5, "rff_std": 1.0}
    _torch.manual_seed(int(d["seed"]))
    model = Aether(params, device=None).eval()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k, v in sd.items():
        if "sum." + k in d:
            assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    return d, model, params


def load_s2s_dynfield():
    """Golden fixture of the reference's seq2seq ``DynamicFieldAether`` (graph summary, FiLM field query,
    ``predict_future``) + the model's parameters recreated from the stored seed through the drop-in constructor
    (key order and checksums verified)."""
    import numpy as _np
    import torch as _torch
    from aether_amd.nn.seq2seq.dynamic_field_aether import DynamicFieldAether
    d = _np.load(os.path.join(GOLDEN, "s2s_dynfield_D3.npz"))
    N, H, R = int(d["num_vars"]), int(d["hidden_size"]), int(d["rnn_hidden"])
    params = {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H,
              "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "input_size": 6, "encoder_mlp_num_layers": 3,
              "encoder_mlp_hidden": 64, "prior_num_layers": 3, "prior_hidden_size": 64, "use_3d": True,
              "pos_representation": "cart", "gpu": False, "decoder_hidden": H, "skip_first": False,
              "decoder_dropout": 0.0, "gumbel_temp": 0.5, "rff_std": 1.0, "graph_hidden": int(d["graph_hidden"]),
              "mlp_hidden": int(d["mlp_hidden"]), "field": None}
    _torch.manual_seed(int(d["seed"]))
    model = DynamicFieldAether(params, device=None).eval()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k, v in sd.items():
        if "sum." + k in d:
            assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    return d, model, params


def load_dyn_decoder(name):
    """Golden case of the reference's variable-N ``Decoder`` step + its parameters recreated from the stored seed
    through the drop-in constructor (key order and checksums verified)."""
    import numpy as _np
    import torch as _torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import CASES
    from aether_amd.nn.dynamicvars.decoder import Decoder
    d = _np.load(os.path.join(GOLDEN, "dyn_decoder.npz"))
    Nmax, absent, K, skip, posrep, H = CASES[name]
    params = {"input_size": 4, "gpu": False, "decoder_hidden": H, "num_edge_types": K, "skip_first": skip,
              "decoder_dropout": 0.0, "pos_representation": posrep}
    _torch.manual_seed(int(d[name + ".seed"]))
    dec = Decoder(params, device=None).eval()
    sd = dec.state_dict()
    assert list(sd.keys()) == [str(k) for k in d[name + ".keys"]]
    for k, v in sd.items():
        # orthogonal_ goes through LAPACK's QR, whose rounding differs between host CPUs (1e-7 relative on the GPU box):
        # those tensors are checked to 1e-6, which the 1e-5 parity tolerance absorbs; everything else to 1e-9
        tol = 1e-6 if "edge_filter" in k and k.endswith("weight") else 1e-9
        assert abs(float(v.double().sum()) - float(d[f"{name}.sum.{k}"])) <= tol * max(1.0, float(d[f"{name}.abs.{k}"])), k
    case = {k[len(name) + 1:]: _torch.from_numpy(d[k]) for k in d.files
            if k.startswith(name + ".") and not any(t in k for t in (".sum.", ".abs.", ".keys", ".seed"))}
    return case, dec, params


def load_dyn_model():
    """Golden fixture of the reference's ``AetherDynamicVars`` + the model recreated from the stored seed through the
    drop-in constructor (key order and checksums verified; BatchNorm statistics perturbed as in the fixture script)."""
    import numpy as _np
    import torch as _torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    d = _np.load(os.path.join(GOLDEN, "dyn_model.npz"))
    _torch.manual_seed(int(d["seed"]))
    model = AetherDynamicVars(dict(MODEL_PARAMS), device=None).eval()
    perturb_bn_(model)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k, v in sd.items():
        if "sum." + k in d:
            tol = 1e-6 if "edge_filter" in k and k.endswith("weight") else 1e-9       # see load_dyn_decoder
            assert abs(float(v.double().sum()) - float(d["sum." + k])) <= tol * max(1.0, float(d["abs." + k])), k
    T = int(d["T"])
    t = lambda k: _torch.from_numpy(d[k])
    case = {"inputs": t("inputs"), "masks": t("masks"), "burn": t("burn"),
            "graph_info": [tuple(t(f"{n}.{i}") for n in ("send", "recv", "e2n")) for i in range(T)],
            "uniform": [t(f"uniform.{i}") for i in range(T - 1)]}
    case["node_inds"] = [case["masks"][0, i].nonzero()[:, -1] for i in range(T)]
    return d, case, model, dict(MODEL_PARAMS)


def load_s2s_loss(name):
    """Golden fixture of the reference's ``Encoder.forward`` / ``Aether.calculate_loss(is_train=False)`` for one loss
    configuration + the model recreated from the stored seed (checksums verified for the first configuration; the
    configurations share the seed and the architecture)."""
    import numpy as _np
    import torch as _torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_seq2seq import loss_params
    from aether_amd.nn.seq2seq.aether import Aether
    d = _np.load(os.path.join(GOLDEN, "s2s_loss_D2.npz"))
    params = loss_params(name)
    _torch.manual_seed(int(d["seed"]))
    model = Aether(params, device=None).eval()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in d["keys"]]
    for k, v in sd.items():
        if "sum." + k in d:
            assert abs(float(v.double().sum()) - float(d["sum." + k])) <= 1e-9 * max(1.0, float(d["abs." + k])), k
    case = {k[len(name) + 1:]: _torch.from_numpy(_np.asarray(d[k])) for k in d.files if k.startswith(name + ".")}
    case["inputs"] = _torch.from_numpy(d["in.inputs"])
    return case, model, params
