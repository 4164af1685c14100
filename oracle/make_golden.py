#!/usr/bin/env python3
"""Generate golden fixtures by running the *imported* reference on the CPU.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (where /root/reference
is mounted); never on the GPU box.  It writes small ``.npz`` files of inputs
and expected outputs under ``tests/golden/``; no reference source is copied.

What is imported: ``nn.state2state.aether.Aether`` (reference
nn/state2state/aether.py:142-186) unmodified.  ``torch_scatter`` is a
third-party dependency that is not installed here (README.md:30-32 installs it
unpinned from conda), so this script places a stand-in module exposing
``scatter(src, index, dim, reduce)`` with the documented pytorch-scatter
semantics (sum / count clamped to >=1, ``dim_size = max(index)+1``) in
``sys.modules`` before the import.  The stand-in is my own code and lives only
in this script.

Usage:  python oracle/make_golden.py [--out tests/golden]
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

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("AETHER_REFERENCE", "/root/reference")


def _install_scatter_standin():
    mod = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
        assert out is None
        if dim < 0:
            dim += src.dim()
        if dim_size is None:
            dim_size = int(index.max()) + 1 if index.numel() else 0
        shape = list(src.shape)
        shape[dim] = dim_size
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        res.index_add_(dim, index, src)
        if reduce in ("sum", "add"):
            return res
        if reduce == "mean":
            cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
            cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
            cnt = cnt.clamp(min=1)
            view = [1] * src.dim()
            view[dim] = dim_size
            return res / cnt.view(view)
        raise NotImplementedError(reduce)

    mod.scatter = scatter
    sys.modules["torch_scatter"] = mod


def _import_reference():
    _install_scatter_standin()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from nn.state2state.aether import Aether  # noqa: WPS433 (reference import)
    from experiments.lorentz.dataset4newton import NBodyDataset
    return Aether, NBodyDataset


def _build_model(Aether, D, seed=1, dtype=torch.float32):
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = Aether(2 * D, 64, 0.0, D, device="cpu")
    return m.to(dtype)


def _run_case(model, inp, want_grads=True):
    """Run the reference forward (and backward) capturing every intermediate."""
    cap = {}
    hooks = []

    def hook(name):
        def fn(_m, _i, o):
            cap[name] = o
        return fn

    hooks.append(model.field_net.register_forward_hook(hook("field")))
    hooks.append(model.localizer.register_forward_hook(hook("localizer")))
    for k in range(1, 5):
        hooks.append(getattr(model.gnn, f"layer_{k}").register_forward_hook(hook(f"layer_{k}")))
    hooks.append(model.gnn.register_forward_hook(hook("pred_local")))
    hooks.append(model.globalizer.register_forward_hook(hook("pred_global")))
    model.zero_grad(set_to_none=True)
    out = model(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    res = {
        "field": cap["field"],
        "rel_feat": cap["localizer"][0],
        "R": cap["localizer"][1],
        "edge_attr_local": cap["localizer"][2],
        "pred_local": cap["pred_local"],
        "pred_global": cap["pred_global"],
        "out": out,
    }
    for k in range(1, 5):
        res[f"x{k}"] = cap[f"layer_{k}"][0]
        res[f"e{k}"] = cap[f"layer_{k}"][1]
    grads = {}
    if want_grads:
        loss = torch.nn.functional.mse_loss(out, inp["target"])
        loss.backward()
        res["loss"] = loss.detach().reshape(1)
        for n, p in model.named_parameters():
            grads["grad." + n] = p.grad.detach().clone()
    for h in hooks:
        h.remove()
    res = {k: v.detach().clone() for k, v in res.items()}
    res.update(grads)
    return res


def _to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def _edge_cases(inp, D, N):
    """Overwrite a few nodes with the degenerate inputs SURVEY.md 8c lists."""
    x, v = inp["x"].clone(), inp["vel"].clone()
    v[0] = 0.0                                   # zero velocity -> theta=0, phi=pi/2
    x[2] = x[1]                                  # coincident particles
    v[3] = -v[4]                                 # anti-parallel headings
    if D == 2:
        v[1] = torch.tensor([-0.5, 0.0])         # theta = pi exactly
        v[2] = torch.tensor([-0.5, -1e-8])       # theta just below -pi -> wraps to ~pi
    else:
        v[1] = torch.tensor([0.0, 0.0, 0.5])     # v || +z
        v[2] = torch.tensor([0.0, 0.0, -0.5])    # v || -z
    inp = dict(inp)
    inp["x"], inp["vel"] = x, v
    return inp


def _refresh_edge_attr(inp):
    from aether_amd.edges import prepare_edge_attr
    rows, cols = inp["edges"]
    q = inp["charges"][rows] * inp["charges"][cols]
    inp["edge_attr"] = prepare_edge_attr(inp["x"], inp["edges"], q)
    inp["h"] = inp["vel"].norm(dim=-1, keepdim=True)
    return inp


def _sparse_case(D, seed):
    """Irregular graph: random directed edges, one isolated receiver, unsorted."""
    g = torch.Generator().manual_seed(seed)
    n = 11
    x = torch.randn(n, D, generator=g)
    v = torch.randn(n, D, generator=g)
    v = 0.5 * v / v.norm(dim=-1, keepdim=True)
    q = torch.randint(0, 3, (n, 1), generator=g).float() - 1.0   # includes neutral 0
    E = 37
    send = torch.randint(0, n, (E,), generator=g)
    recv = torch.randint(0, n, (E,), generator=g)
    keep = (send != recv) & (recv != 4)          # node 4 receives nothing
    send, recv = send[keep], recv[keep]
    # reference infers dim_size = max(recv)+1, so the last node must receive
    send = torch.cat([send, torch.tensor([0])])
    recv = torch.cat([recv, torch.tensor([n - 1])])
    edges = [send.long(), recv.long()]
    inp = dict(x=x, vel=v, charges=q, edges=edges,
               target=x + v + 0.05 * torch.randn(n, D, generator=g))
    inp = _refresh_edge_attr(inp)
    inp["meta"] = dict(B=1, N=n, D=D, seed=seed)
    return inp


def _save_case(path, inp, res, extra=None):
    blob = {}
    for k in ("h", "x", "vel", "charges", "edge_attr", "target"):
        blob["in." + k] = inp[k]
    blob["in.send"] = inp["edges"][0]
    blob["in.recv"] = inp["edges"][1]
    blob["meta"] = np.array([inp["meta"]["B"], inp["meta"]["N"], inp["meta"]["D"], inp["meta"]["seed"]])
    blob.update({"ref." + k: v for k, v in res.items()})
    if extra:
        blob.update(extra)
    np.savez_compressed(path, **_to_np(blob))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    sys.path.insert(0, REPO)
    from aether_amd.synthetic import make_batch

    Aether, NBodyDataset = _import_reference()
    torch.set_num_threads(1)          # deterministic summation order on the CPU

    # ---- edge index fixtures (bit-exact target), dataset4newton.py:84-94 ----
    ds = NBodyDataset.__new__(NBodyDataset)
    edge_blob = {}
    for (B, N) in [(1, 5), (3, 5), (128, 20), (2, 2), (1, 3)]:
        rows, cols = [], []
        for i in range(N):
            for j in range(N):
                if i != j:
                    rows.append(i)
                    cols.append(j)
        ds.edges = [rows, cols]
        e = ds.get_edges(B, N)
        edge_blob[f"send_B{B}_N{N}"] = e[0].numpy()
        edge_blob[f"recv_B{B}_N{N}"] = e[1].numpy()
    np.savez_compressed(os.path.join(args.out, "edges.npz"), **edge_blob)
    print("wrote edges.npz")

    for D in (2, 3):
        model = _build_model(Aether, D, seed=1)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        np.savez_compressed(os.path.join(args.out, f"state_dict_D{D}.npz"), **_to_np(sd))
        model64 = _build_model(Aether, D, seed=1, dtype=torch.float64)

        def run(tag, inp, grads=True, grads64=False):
            res = _run_case(model, inp, want_grads=grads)
            inp64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v)
                     for k, v in inp.items()}
            res64 = _run_case(model64, inp64, want_grads=grads)
            extra = {"ref64.out": res64["out"], "ref64.field": res64["field"],
                     "ref64.x4": res64["x4"], "ref64.e3": res64["e3"]}
            if grads and grads64:
                extra.update({"ref64." + k: v.float() for k, v in res64.items() if k.startswith("grad.")})
            _save_case(os.path.join(args.out, f"case_D{D}_{tag}.npz"), inp, res, extra)

        run("B1N5", make_batch(1, 5, D, seed=0), grads=False)
        run("B3N5", make_batch(3, 5, D, seed=1), grads64=True)
        run("B2N20", make_batch(2, 20, D, seed=2))
        run("B2N2", make_batch(2, 2, D, seed=3), grads=False)
        run("edge_B2N5", _refresh_edge_attr(_edge_cases(make_batch(2, 5, D, seed=4), D, 5)))
        run("sparse", _sparse_case(D, seed=5))

        # full-size headline config: store inputs + output only (cfg2 / cfg3)
        inp = make_batch(128, 20, D, seed=0)
        torch.set_num_threads(8)
        with torch.no_grad():
            out = model(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
            out64 = model64(inp["h"].double(), inp["x"].double(), inp["edges"], inp["vel"].double(),
                            inp["edge_attr"].double(), inp["charges"].double())
        torch.set_num_threads(1)
        np.savez_compressed(
            os.path.join(args.out, f"full_D{D}_B128N20.npz"),
            **_to_np({"in.x": inp["x"], "in.vel": inp["vel"], "in.charges": inp["charges"],
                      "in.target": inp["target"], "ref.out": out, "ref64.out": out64,
                      "meta": np.array([128, 20, D, 0])}))
        print(f"wrote full_D{D}_B128N20.npz")

        # 20-step rollout of the state2state module (metric 2):
        # x_{t+1} = Aether(x_t, v_t), v_{t+1} = (x_{t+1} - x_t) / dt, dt = 1.
        inp = make_batch(4, 5, D, seed=7)
        xs, x, v = [], inp["x"].clone(), inp["vel"].clone()
        rows, cols = inp["edges"]
        qprod = inp["charges"][rows] * inp["charges"][cols]
        with torch.no_grad():
            for _ in range(20):
                dist = torch.sqrt(torch.sum((x[rows] - x[cols]) ** 2, 1)).unsqueeze(1)
                ea = torch.cat([qprod, dist], 1)
                xn = model(v.norm(dim=-1, keepdim=True), x, inp["edges"], v, ea, inp["charges"])
                v = (xn - x) / 1.0
                x = xn
                xs.append(x.clone())
        np.savez_compressed(
            os.path.join(args.out, f"rollout_D{D}_B4N5.npz"),
            **_to_np({"in.x": inp["x"], "in.vel": inp["vel"], "in.charges": inp["charges"],
                      "ref.traj": torch.stack(xs), "meta": np.array([4, 5, D, 7])}))
        print(f"wrote rollout_D{D}_B4N5.npz")


if __name__ == "__main__":
    main()
