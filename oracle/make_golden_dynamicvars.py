#!/usr/bin/env python3
"""Golden fixtures for the variable-N decoder step from the imported reference.

TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference is mounted.  Imported, unmodified:
``nn.dynamicvars.aether_dynamicvars.Decoder`` and ``experiments.ind.single_ind_data.get_knn_graph_info`` (the graph
the inD data set attaches to every time step, single_ind_data.py:87).  Stand-ins as in make_golden_seq2seq.py:
torch_scatter (imported by the module) and an identity ``.cuda()`` (:794).  The parameters come from the class's own constructor under ``torch.manual_seed``; the tests recreate them
with the drop-in module, whose constructor creates the same tensors in the same order (checksums stored).
Usage:  python oracle/make_golden_dynamicvars.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AETHER_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "oracle"))

# name: (Nmax, absent objects, edge types, skip_first, pos_representation, hidden)
CASES = {"full8": (8, [], 2, False, "cart", 128), "tail6": (9, [6, 7, 8], 4, True, "cart", 128),
         "gaps": (12, [0, 5, 6], 3, True, "polar", 128), "knn20": (20, [19], 2, False, "cart", 128),
         "empty": (4, [0, 1, 2, 3], 2, False, "cart", 128)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import make_golden as MG
    import dynamicvars_oracle as DO
    MG._install_scatter_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.dynamicvars.aether_dynamicvars import Decoder
        from experiments.ind.single_ind_data import get_knn_graph_info
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = {}
    try:
        for idx, (name, (Nmax, absent, K, skip, posrep, H)) in enumerate(CASES.items()):
            params = {"input_size": 4, "gpu": False, "decoder_hidden": H, "num_edge_types": K, "skip_first": skip,
                      "decoder_dropout": 0.0, "pos_representation": posrep}
            torch.manual_seed(300 + idx)
            with contextlib.redirect_stdout(io.StringIO()):
                dec = Decoder(params).eval()
            g = torch.Generator().manual_seed(400 + idx)
            inputs = torch.randn(1, Nmax, 4, generator=g)
            hidden = torch.randn(1, Nmax, H, generator=g) * 0.3
            field = torch.randn(1, Nmax, 2, generator=g) * 0.3
            masks = torch.ones(Nmax)
            masks[absent] = 0.0
            nv = int(masks.sum())
            if nv > 1:
                send, recv, e2n = get_knn_graph_info(inputs[0], masks, nv)
                edges = torch.softmax(torch.randn(1, send.numel(), K, generator=g), -1)
                graph_info = (send, recv, e2n)
            else:
                edges, graph_info = torch.zeros(1, 0, K), None
            with torch.no_grad():
                pred, hid = dec(inputs, hidden, edges, masks.unsqueeze(0), graph_info, field)
            sd = {k: v.detach() for k, v in dec.state_dict().items()}
            if nv > 1:
                o_pred, o_hid = DO.decoder_step(sd, inputs, hidden, edges, masks, graph_info, field, skip, posrep)
                err = max(float((o_pred - pred).abs().max()), float((o_hid - hid).abs().max()))
            else:
                err = 0.0
            for k, v in (("inputs", inputs), ("hidden", hidden), ("field", field), ("masks", masks), ("edges", edges),
                         ("ref.pred", pred), ("ref.hidden", hid)):
                out[f"{name}.{k}"] = v.numpy()
            if nv > 1:
                out[f"{name}.send"], out[f"{name}.recv"], out[f"{name}.e2n"] = send.numpy(), recv.numpy(), e2n.numpy()
            for k, v in sd.items():                                  # the tests recreate the parameters from the seed
                out[f"{name}.sum.{k}"] = np.float64(v.double().sum().item())
                out[f"{name}.abs.{k}"] = np.float64(v.double().abs().sum().item())
            out[f"{name}.keys"] = np.array(list(sd.keys()))
            out[f"{name}.seed"] = np.int64(300 + idx)
            print(name, "present", nv, "edges", 0 if nv <= 1 else send.numel(), "oracle abs err", err)
    finally:
        torch.Tensor.cuda = orig_cuda
    np.savez_compressed(os.path.join(args.out, "dyn_decoder.npz"), **out)
    print("wrote dyn_decoder.npz")
    return args.out


MODEL_PARAMS = {"input_size": 4, "gpu": False, "decoder_hidden": 128, "num_edge_types": 3, "skip_first": True,
                "decoder_dropout": 0.0, "pos_representation": "cart", "no_encoder_bn": False, "encoder_dropout": 0.0,
                "encoder_hidden": 128, "encoder_rnn_hidden": 64, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3,
                "encoder_mlp_hidden": 64, "prior_num_layers": 3, "prior_hidden_size": 64,
                "encoder_normalize_mode": "normalize_all", "train_data_len": 50, "field_hidden": 64, "gumbel_temp": 0.5,
                "rff_std": 1.0}
MODEL_SEED = 505


def perturb_bn_(model):
    """Non-trivial BatchNorm running statistics (a fresh module has mean 0 / var 1), from a private generator."""
    g = torch.Generator().manual_seed(606)
    for n, b in model.named_buffers():
        if n.endswith("running_mean"):
            b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        elif n.endswith("running_var"):
            b.copy_(torch.rand(b.shape, generator=g) + 0.5)


def model_fixture(out_dir):
    """The imported reference ``AetherDynamicVars``: predict_field, one Encoder.single_step_forward and the whole
    predict_future (6 time steps, 9 object slots, objects appearing / disappearing, burn-in masks switching to the
    model's own predictions after step 3), graphs from the reference's get_knn_graph_info."""
    import make_golden as MG
    MG._install_scatter_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
        from experiments.ind.single_ind_data import get_knn_graph_info
    import dynamicvars_oracle as DO
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(MODEL_SEED)
        with contextlib.redirect_stdout(io.StringIO()):
            m = AetherDynamicVars(dict(MODEL_PARAMS)).eval()
        perturb_bn_(m)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        g = torch.Generator().manual_seed(1)
        T, N = 6, 9
        inputs = torch.randn(1, T, N, 4, generator=g)
        masks = (torch.rand(1, T, N, generator=g) < 0.8).float()
        masks[:, :, :2] = 1
        burn = torch.ones(1, T, N)
        burn[:, 3:] = 0
        node_inds = [[masks[0, t].nonzero()[:, -1] for t in range(T)]]
        graph_info = [[get_knn_graph_info(inputs[0, t], masks[0, t], int(masks[0, t].sum())) for t in range(T)]]
        with torch.no_grad():
            field0, _ = m.predict_field(inputs[:, 0], masks[:, 0])
            state0 = m.encoder.get_initial_hidden(inputs)
            state0 = (torch.randn(state0[0].shape, generator=g) * 0.2, torch.randn(state0[1].shape, generator=g) * 0.2)
            logits0, state1 = m.encoder.single_step_forward(inputs[:, 0], masks[:, 0], node_inds[0][0], graph_info[0][0],
                                                            state0, field0)
        torch.manual_seed(77)
        with torch.no_grad():
            ref = m.predict_future(inputs, masks, node_inds, graph_info, burn)
        torch.manual_seed(77)
        U = [torch.rand(graph_info[0][t][0].numel(), MODEL_PARAMS["num_edge_types"]) for t in range(T - 1)]
        o = DO.predict_future(sd, inputs, masks, node_inds[0], graph_info[0], burn, U, 0.5, True, "cart")
        out = {"inputs": inputs.numpy(), "masks": masks.numpy(), "burn": burn.numpy(), "ref.predictions": ref.numpy(),
               "ref.field0": field0.numpy(), "state0.h": state0[0].numpy(), "state0.c": state0[1].numpy(),
               "ref.logits0": logits0.numpy(), "ref.state1.h": state1[0].numpy(), "ref.state1.c": state1[1].numpy(),
               "seed": np.int64(MODEL_SEED), "T": np.int64(T)}
        for t in range(T):
            out[f"send.{t}"], out[f"recv.{t}"], out[f"e2n.{t}"] = (x.numpy() for x in graph_info[0][t])
            if t < T - 1:
                out[f"uniform.{t}"] = U[t].numpy()
        for k, v in sd.items():
            if v.dtype.is_floating_point:
                out["sum." + k] = np.float64(v.double().sum().item())
                out["abs." + k] = np.float64(v.double().abs().sum().item())
        out["keys"] = np.array(list(sd.keys()))
        np.savez_compressed(os.path.join(out_dir, "dyn_model.npz"), **out)
        print("wrote dyn_model.npz", tuple(ref.shape), "oracle abs err", float((o - ref).abs().max()))
    finally:
        torch.Tensor.cuda = orig_cuda


if __name__ == "__main__":
    model_fixture(main())                # both fixtures go where --out says
