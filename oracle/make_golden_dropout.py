#!/usr/bin/env python3
"""Golden fixtures for two things the round-3 build added, from the *imported* reference on the CPU:

* a train()-mode step with ``dropout_prob = 0.25``: the two masks its ``nn.Dropout`` layers drew (captured by forward hooks
  on ``gnn.out_mlp[2]`` / ``[5]``, nn/state2state/locs/locs.py:160-168) together with the output and all gradients, so that
  the oracle's and the HIP path's mask placement and scaling are pinned to the reference and not to a restatement;
* gradients with respect to the INPUTS (x, vel, edge_attr_orig) of the reference's differentiable forward
  (nn/state2state/aether.py:169-186), with and without dropout.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference is mounted); never on the GPU box.  Inputs
and expected outputs only; no reference source is copied.  Same import recipe as oracle/make_golden.py (its torch_scatter
stand-in).

Usage:  python oracle/make_golden_dropout.py [--out tests/golden]
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
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import make_golden as MG
    from aether_amd.synthetic import make_batch
    Aether, _ = MG._import_reference()
    torch.set_num_threads(1)
    for D in (2, 3):
        sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(REPO, "tests", "golden", f"state_dict_D{D}.npz")).items()}
        for tag, p, B, N, seed in (("dropout", 0.25, 4, 9, 41), ("inputgrad", 0.0, 3, 12, 42)):
            with contextlib.redirect_stdout(io.StringIO()):
                model = Aether(2 * D, 64, p, D, device="cpu")
            model.load_state_dict(sd)
            model.train()
            inp = make_batch(B, N, D, seed=seed)
            cap = {}

            def hook(name):
                def fn(_m, i, o):
                    cap[name] = (o != 0).float() / (1.0 - p) if p > 0 else torch.ones_like(o)
                return fn

            hooks = [model.gnn.out_mlp[2].register_forward_hook(hook("mask1")),
                     model.gnn.out_mlp[5].register_forward_hook(hook("mask2"))]
            leaves = {k: inp[k].clone().requires_grad_(True) for k in ("x", "vel", "edge_attr")}
            torch.manual_seed(100 + seed)
            out = model(inp["h"], leaves["x"], inp["edges"], leaves["vel"], leaves["edge_attr"], inp["charges"])
            loss = torch.nn.functional.mse_loss(out, inp["target"])
            loss.backward()
            for h in hooks:
                h.remove()
            blob = {"in." + k: inp[k] for k in ("h", "x", "vel", "charges", "edge_attr", "target")}
            blob["in.send"], blob["in.recv"] = inp["edges"]
            blob["meta"] = np.array([B, N, D, seed])
            blob["dropout_prob"] = np.array([p])
            blob["mask1"], blob["mask2"] = cap["mask1"], cap["mask2"]
            blob["ref.out"], blob["ref.loss"] = out.detach(), loss.detach().reshape(1)
            for k, v in leaves.items():
                blob["ref.grad_in." + k] = v.grad
            for n, q in model.named_parameters():
                blob["ref.grad." + n] = q.grad
            path = os.path.join(args.out, f"case_D{D}_{tag}.npz")
            np.savez_compressed(path, **MG._to_np(blob))
            frac = float((cap["mask1"] == 0).float().mean())
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB); zeros in mask1: {frac:.3f}")
    return args.out


if __name__ == "__main__":
    main()
