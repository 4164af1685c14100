#!/usr/bin/env python3
"""Golden fixtures for the dynamic-field state2state variant from the imported reference.

TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference is mounted.  Imported,
unmodified: ``nn.state2state.dynamic_field_aether.DynamicFieldAether`` (LatentFieldNetwork, GraphSummary,
FilmedNetwork / FiLM and the shared localizer / GNN / globalizer).  Two third-party packages it imports are
not installed here and are unpinned upstream (README.md:30-32, ``conda install pyg pytorch-scatter``):

* ``torch_scatter.scatter``   -- the stand-in of oracle/make_golden.py;
* ``torch_geometric.nn.aggr.AttentionalAggregation`` -- a stand-in with PyG's published semantics
  (torch_geometric/nn/aggr/attentional.py + torch_geometric/utils/softmax.py):
  ``gate = gate_nn(x); x = nn(x); gate = softmax(gate, index)  [out = exp(src - max_group);
  out / (sum_group(out) + 1e-16)]; return sum_group(gate * x)``, sub-modules named ``gate_nn`` / ``nn``.

Both stand-ins are my own code and live only in these scripts.
Usage:  python oracle/make_golden_dynfield.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AETHER_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def install_pyg_standin():
    class AttentionalAggregation(nn.Module):
        def __init__(self, gate_nn, nn=None):
            super().__init__()
            self.gate_nn = gate_nn
            self.nn = nn

        def forward(self, x, index=None, ptr=None, dim_size=None, dim=-2):
            gate = self.gate_nn(x)
            x = self.nn(x) if self.nn is not None else x
            n = int(index.max()) + 1 if dim_size is None else dim_size
            gmax = torch.full((n, gate.shape[-1]), -float("inf"), dtype=gate.dtype).scatter_reduce(
                0, index.unsqueeze(-1).expand_as(gate), gate, reduce="amax", include_self=True)
            out = (gate - gmax[index]).exp()
            denom = torch.zeros(n, gate.shape[-1], dtype=gate.dtype).index_add_(0, index, out) + 1e-16
            gate = out / denom[index]
            return torch.zeros(n, x.shape[-1], dtype=x.dtype).index_add_(0, index, gate * x)

    tg, tg_nn, tg_aggr = (types.ModuleType(n) for n in ("torch_geometric", "torch_geometric.nn", "torch_geometric.nn.aggr"))
    tg_aggr.AttentionalAggregation = AttentionalAggregation
    tg.nn, tg_nn.aggr = tg_nn, tg_aggr
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tg_nn, "torch_geometric.nn.aggr": tg_aggr})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    import make_golden as MG
    MG._install_scatter_standin()
    install_pyg_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.state2state.dynamic_field_aether import DynamicFieldAether
    from aether_amd.synthetic import make_batch
    for D in (2, 3):
        torch.manual_seed(50 + D)
        with contextlib.redirect_stdout(io.StringIO()):
            model = DynamicFieldAether(2 * D, 64, 0.0, D, device="cpu").eval()
        out = {}
        for name, (B, N) in (("small", (3, 5)), ("cfg", (16, 20))):
            inp = make_batch(B, N, D, seed=60 + D)
            with torch.no_grad():
                inputs = torch.cat([inp["x"], inp["vel"]], -1)
                field = model.field_net(inputs, inp["charges"], N)
                pred = model(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"], N)
                m64 = model.double()
                field64 = m64.field_net(inputs.double(), inp["charges"].double(), N)
                model.float()
            for k in ("x", "vel", "charges", "edge_attr"):
                out[f"{name}.in.{k}"] = inp[k].numpy()
            out[f"{name}.B"], out[f"{name}.N"] = np.int64(B), np.int64(N)
            out[f"{name}.ref.field"], out[f"{name}.ref.out"] = field.numpy(), pred.numpy()
            out[f"{name}.ref64.field"] = field64.numpy()
        for k, v in model.state_dict().items():
            out["sd." + k] = v.numpy()
        out["keys"] = np.array(list(model.state_dict().keys()))
        np.savez(os.path.join(args.out, f"dynfield_D{D}.npz"), **out)
        print("wrote dynfield_D%d.npz" % D, len(model.state_dict()), "tensors")


if __name__ == "__main__":
    main()
