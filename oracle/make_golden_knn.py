#!/usr/bin/env python3
"""Golden fixtures for the kNN edge builder from the imported reference.

TEST INFRASTRUCTURE ONLY; runs in the build container where /root/reference is mounted.  Imported,
unmodified: ``nn.dynamicvars.aether_dynamicvars.Encoder.knn_edges`` (called unbound: the method does not touch
``self``) and ``experiments.ind.single_ind_data.get_knn_graph_info``.  torch_scatter (imported by the module,
not used by these functions) is the stand-in of oracle/make_golden.py.

A seed is accepted only if, in every scene, the distances from an object to its k + 1 nearest neighbours are
separated by more than 1e-4 relative (evaluated in fp64): then neither the rounding of a distance evaluation
(torch.cdist uses a matmul expansion above 25 objects) nor topk's unspecified order of equal values can change
an index.
Usage:  python oracle/make_golden_knn.py [--out tests/golden]
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


def separated(x, masks, k, rel=1e-4):
    xs = x.reshape(-1, x.shape[-2], x.shape[-1]).double().numpy()
    ms = masks.reshape(-1, masks.shape[-1]).numpy()
    for s in range(xs.shape[0]):
        idx = np.nonzero(ms[s])[0]
        p = xs[s, idx, :2]
        d = np.sqrt(((p[:, None] - p[None]) ** 2).sum(-1))
        np.fill_diagonal(d, np.inf)
        d.sort(axis=1)
        top = d[:, :k + 1]
        top = top[:, np.isfinite(top).all(0)] if top.size else top
        if top.shape[1] >= 2 and ((top[:, 1:] - top[:, :-1]) < rel * top[:, 1:]).any():
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    import make_golden as MG
    MG._install_scatter_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        from nn.dynamicvars.aether_dynamicvars import Encoder
        from experiments.ind.single_ind_data import get_knn_graph_info
    out = {}
    # name: (leading shape, N, D, k, presence probability)
    cases = {"small": ((2, 3), 6, 4, 3, 0.7), "scenes": ((4, 12), 20, 4, 10, 0.6), "few": ((1, 5), 4, 4, 10, 0.5),
             "wide": ((1, 3), 60, 4, 10, 0.8), "flat": ((7,), 15, 2, 10, 0.9)}
    for name, (lead, N, D, k, p) in cases.items():
        for seed in range(100):
            g = torch.Generator().manual_seed(1000 * len(name) + seed)
            x = torch.randn(*lead, N, D, generator=g) * 10.0
            m = (torch.rand(*lead, N, generator=g) < p).float()
            if name == "few":
                m[0, 0] = 0.0                                    # an empty scene
                m[0, 1] = torch.tensor([0., 1., 0., 0.])         # a scene with a single object
            if separated(x, m, min(k, N - 1)):
                break
        else:
            raise RuntimeError("no separated seed for " + name)
        with torch.no_grad():
            send, recv, num = Encoder.knn_edges(None, x, m, k=k)
        out[f"{name}.x"], out[f"{name}.masks"], out[f"{name}.k"] = x.numpy(), m.numpy(), np.int64(k)
        out[f"{name}.send"], out[f"{name}.recv"], out[f"{name}.num"] = send.numpy(), recv.numpy(), np.asarray(num.numpy())
        print(name, tuple(x.shape), "edges", send.numel(), "seed", seed)
    # one scene through the data-side builder (k = 10 fixed inside; num_vars = number of present objects)
    g = torch.Generator().manual_seed(77)
    for seed in range(100):
        x = torch.randn(30, 4, generator=g) * 10.0
        m = (torch.rand(30, generator=g) < 0.7).float()
        if separated(x[None], m[None], 10):
            break
    s, r = get_knn_graph_info(x, m, int(m.sum()), use_edge2node=False)
    out["info.x"], out["info.masks"], out["info.send"], out["info.recv"] = x.numpy(), m.numpy(), s.numpy(), r.numpy()
    np.savez(os.path.join(args.out, "knn_edges.npz"), **out)
    print("wrote knn_edges.npz; info edges", s.numel())


if __name__ == "__main__":
    main()
