#!/usr/bin/env python3
"""Golden fixtures for hidden_size > 64, from the *imported* reference (TEST INFRASTRUCTURE ONLY).

The reference builds ``Aether(..., hidden_size=args.nf, ...)`` with whatever ``--nf`` says
(experiments/lorentz/main.py:42-43,143; nn/state2state/aether.py:143-158).  This script runs the imported reference
class at widths 96 / 128 / 256 on small seeded batches and stores, per width:

  * the shapes of its ``state_dict`` and per-tensor checksums (sum, sum of magnitudes) of its seed-1 initialisation --
    the drop-in's constructor has to reproduce both (bit-identical init under the same torch seed), so the weights
    themselves are not stored;
  * inputs, the output, the node states ``x1..x4`` and the messages ``e1..e3``;
  * the gradients of an MSE loss: in full for the width-96 model, as checksums + the leading 64 entries of every tensor
    for the wider ones (a 256-wide model has 1.3 M parameters).

Same stand-ins as oracle/make_golden.py (``torch_scatter.scatter`` with the documented semantics).  Nothing of the
reference's source is copied.  Usage:  python oracle/make_golden_wide.py [--out tests/golden]
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
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import make_golden as MG  # noqa: E402

CASES = [(96, 2, 3, 6, 21), (128, 3, 2, 7, 22), (256, 2, 2, 5, 23)]      # (hidden, D, batch, nodes per graph, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    from aether_amd.synthetic import make_batch

    Aether, _ = MG._import_reference()
    torch.set_num_threads(1)
    for (H, D, B, N, seed) in CASES:
        torch.manual_seed(1)
        with contextlib.redirect_stdout(io.StringIO()):
            model = Aether(2 * D, H, 0.0, D, device="cpu")
        inp = make_batch(B, N, D, seed=seed)
        res = MG._run_case(model, inp, want_grads=True)
        blob = {}
        for k in ("h", "x", "vel", "charges", "edge_attr", "target"):
            blob["in." + k] = inp[k]
        blob["in.send"], blob["in.recv"] = inp["edges"]
        blob["meta"] = np.array([B, N, D, seed, H])
        for k in ("out", "x1", "x2", "x3", "x4", "e1", "e2", "e3", "field", "loss"):
            blob["ref." + k] = res[k]
        names = []
        for n, p in model.state_dict().items():
            names.append(n)
            blob["shape." + n] = np.array(p.shape, dtype=np.int64)
            blob["sum." + n] = np.array(float(p.double().sum()))
            blob["abs." + n] = np.array(float(p.double().abs().sum()))
            g = res["grad." + n]
            if H <= 96:
                blob["grad." + n] = g
            else:
                blob["gsum." + n] = np.array(float(g.double().sum()))
                blob["gabs." + n] = np.array(float(g.double().abs().sum()))
                blob["ghead." + n] = g.reshape(-1)[:64].clone()
        blob["names"] = np.array(names)
        path = os.path.join(args.out, f"wide_H{H}_D{D}.npz")
        np.savez_compressed(path, **MG._to_np(blob))
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
