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


if __name__ == "__main__":
    main()
