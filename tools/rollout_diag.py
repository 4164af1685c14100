#!/usr/bin/env python3
"""Where does the 20-step rollout of the headline batch leave the oracle?  

For the bench batch (make_batch(128, 20, 2, seed=0)) and either the seed-1 weights or the weights after
`--train` AdamW steps (what bench.py holds when it runs its parity block), prints per step the max
|traj - oracle_fp64| of: the oracle in fp32, the device rollout (fused, split), the fused kernel without
the split, the streamed kernels, and the loop of module calls.  Then, for the worst (step, node) of the
device rollout: a single step from the fp64 oracle's state at that step through every path (is the step
itself off?), and the oracle's own sensitivity to a 1e-7 perturbation of that state (is the node
ill-conditioned?).  Diagnostic only; imports oracle/ like the tests do.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=0, help="AdamW steps before the rollout (bench.py runs ~60)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=20)
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from aether_amd import _lib
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.rollout import rollout, rollout_stepwise
    from aether_amd.synthetic import make_batch
    from oracle import aether_oracle as O

    dev = torch.device("cuda", 0)
    B, N, D, T = args.batch, args.nodes, args.dims, args.steps
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Aether(2 * D, 64, 0.0, D, device=dev)
    host = make_batch(B, N, D, seed=args.seed)
    inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in host.items() if k != "edges"}
    edges_d = [e.to(dev) for e in host["edges"]]
    if args.train:
        opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-12)
        for _ in range(args.train):
            opt.zero_grad(set_to_none=True)
            o = model(inp["h"], inp["x"], edges_d, inp["vel"], inp["edge_attr"], inp["charges"])
            torch.nn.functional.mse_loss(o, inp["target"]).backward()
            opt.step()
        torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd.items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        t64 = O.rollout(sd64, host["x"].double(), host["vel"].double(), host["edges"], host["charges"].double(), T)
        t32 = O.rollout(sd, host["x"], host["vel"], host["edges"], host["charges"], T)
    paths = {"oracle_fp32": t32.double()}
    with torch.no_grad():
        paths["hip_rollout_fused_split"] = rollout(model, inp["x"], inp["vel"], edges_d, inp["charges"], T).cpu().double()
        paths["hip_rollout_again"] = rollout(model, inp["x"], inp["vel"], edges_d, inp["charges"], T).cpu().double()
        paths["hip_loop_of_calls"] = rollout_stepwise(model, inp["x"], inp["vel"], edges_d, inp["charges"], T).cpu().double()
        model.flags = _lib.FLAG_FORCE_STREAMED
        paths["hip_rollout_streamed"] = rollout(model, inp["x"], inp["vel"], edges_d, inp["charges"], T).cpu().double()
        model.flags = 0
        # the same fused kernel, one workgroup per graph (takes effect at the next graph build: fresh index tensors)
        _lib.check(_lib.load().aether_set_option(b"fused_split", 0), "set_option")
        e2 = [e.clone() for e in edges_d]
        model._graphs = type(model._graphs)()
        model._ws_key = None
        paths["hip_rollout_fused_unsplit"] = rollout(model, inp["x"], inp["vel"], e2, inp["charges"], T).cpu().double()
        _lib.check(_lib.load().aether_set_option(b"fused_split", 1), "set_option")
        model._graphs = type(model._graphs)()
        model._ws_key = None
    scale = float(t64.abs().max())
    rep = {"scale_max_abs_oracle": scale, "train_steps": args.train, "paths": {}}
    print(f"scale max|oracle_fp64| = {scale:.4f}")
    for name, tr in paths.items():
        err = (tr - t64).abs()                      # [T, Nn, D]
        per_step = err.amax(dim=(1, 2))
        rms = float(err.pow(2).mean().sqrt())
        node_err = err.amax(dim=2)                  # [T, Nn]
        outl = int((node_err[-1] > 10 * float(node_err[-1].pow(2).mean().sqrt())).sum())
        w = int(node_err.argmax())
        ts, nd = divmod(w, node_err.shape[1])
        rep["paths"][name] = {"max_rel": float(err.max()) / scale, "rms": rms, "per_step_max_abs": per_step.tolist(),
                              "worst_step": ts, "worst_node": nd, "outliers_last_step_gt_10rms": outl}
        print(f"{name:28s} max_rel={float(err.max()) / scale:.3e} rms={rms:.3e} worst(step {ts}, node {nd}, graph {nd // N}) "
              f"outliers@T={outl}")
        print("   per-step max abs:", " ".join(f"{v:.1e}" for v in per_step.tolist()))
    same = bool(torch.equal(paths["hip_rollout_fused_split"], paths["hip_rollout_again"]))
    print("device rollout bit-identical on a second run:", same)
    rep["rerun_bit_identical"] = same

    # ---- the worst node of the device rollout: first step at which its error leaves the pack
    err = (paths["hip_rollout_fused_split"] - t64).abs().amax(dim=2)
    nd = rep["paths"]["hip_rollout_fused_split"]["worst_node"]
    g0 = nd // N
    print(f"worst node {nd} (graph {g0}): error by step:", " ".join(f"{float(err[t, nd]):.1e}" for t in range(T)))
    o32 = (paths["oracle_fp32"] - t64).abs().amax(dim=2)
    print(f"   oracle fp32, same node      :", " ".join(f"{float(o32[t, nd]):.1e}" for t in range(T)))
    # graph-level: max over the nodes of that graph
    gs = slice(g0 * N, (g0 + 1) * N)
    print(f"   device, whole graph {g0}     :", " ".join(f"{float(err[t, gs].max()):.1e}" for t in range(T)))
    print(f"   oracle fp32, whole graph    :", " ".join(f"{float(o32[t, gs].max()):.1e}" for t in range(T)))

    # ---- single steps from the oracle's own fp64 states: is any single step off?
    print("single step from the fp64 oracle state of step t (max abs error of the step's output vs fp64 step):")
    rows, cols = host["edges"]
    qprod = host["charges"][rows] * host["charges"][cols]
    worst_single = []
    for t in range(T):
        xs = host["x"].double() if t == 0 else t64[t - 1]
        vs = host["vel"].double() if t == 0 else (t64[t - 1] - (host["x"].double() if t == 1 else t64[t - 2]))
        x32, v32 = xs.float(), vs.float()
        dist = torch.sqrt(torch.sum((x32[rows] - x32[cols]) ** 2, 1)).unsqueeze(1)
        ea = torch.cat([qprod, dist], 1)
        with torch.no_grad():
            d64 = torch.sqrt(torch.sum((x32.double()[rows] - x32.double()[cols]) ** 2, 1)).unsqueeze(1)
            w64 = O.aether_forward(sd64, x32.double(), v32.double(), host["edges"], torch.cat([qprod.double(), d64], 1),
                                   host["charges"].double())
            w32 = O.aether_forward(sd, x32, v32, host["edges"], ea, host["charges"])
            got = model(inp["h"], x32.to(dev), edges_d, v32.to(dev), ea.to(dev), inp["charges"]).cpu()
            tr1 = rollout(model, x32.to(dev), v32.to(dev), edges_d, inp["charges"], 1).cpu()[0]
        e_h = float((got.double() - w64).abs().max())
        e_r = float((tr1.double() - w64).abs().max())
        e_o = float((w32.double() - w64).abs().max())
        worst_single.append((e_h, e_r, e_o))
        print(f"   t={t:2d} hip_forward {e_h:.2e}  hip_rollout1 {e_r:.2e}  oracle_fp32 {e_o:.2e}")
    rep["single_step_from_oracle_state"] = worst_single

    # ---- conditioning: the oracle's response (fp64) to a 1e-7 relative perturbation of the initial state
    with torch.no_grad():
        gen = torch.Generator().manual_seed(5)
        dx = torch.randn(host["x"].shape, generator=gen, dtype=torch.float64) * 1e-7
        dv = torch.randn(host["x"].shape, generator=gen, dtype=torch.float64) * 1e-7
        tp = O.rollout(sd64, host["x"].double() + dx, host["vel"].double() + dv, host["edges"], host["charges"].double(), T)
    amp = (tp - t64).abs().amax(dim=(1, 2))
    print("fp64 oracle, initial state perturbed by N(0, 1e-7): per-step max abs deviation:")
    print("   ", " ".join(f"{v:.1e}" for v in amp.tolist()))
    rep["fp64_perturbation_1e-7_per_step"] = amp.tolist()
    nd_amp = (tp - t64).abs().amax(dim=2)[-1]
    print(f"   at T: worst node {int(nd_amp.argmax())}, its deviation {float(nd_amp.max()):.2e}; device-worst node {nd}: {float(nd_amp[nd]):.2e}")
    # ---- a jump in ANY path (the fp32 oracle included): redo that step from that path's own fp32 state in
    # fp32 and fp64 and look for a local-frame feature that sits on a branch cut (geometry.py:37-66,76-101)
    def find_jump(tr):
        ne = (tr - t64).abs().amax(dim=2)           # [T, Nn]
        for t in range(1, T):
            prev = float(ne[t - 1].max())
            if float(ne[t].max()) > 5.0 * prev and float(ne[t].max()) > 1e-5:
                return t, int(ne[t].argmax())
        return None
    rep["jumps"] = {}
    for name, tr in paths.items():
        j = find_jump(tr)
        if j is None:
            continue
        t, nd = j
        g0 = nd // N
        print(f"JUMP in {name}: step {t}, node {nd} (graph {g0})")
        tr32 = tr.float()
        xs = tr32[t - 1]
        vs = tr32[t - 1] - (host["x"] if t == 1 else tr32[t - 2])
        dist = torch.sqrt(torch.sum((xs[rows] - xs[cols]) ** 2, 1)).unsqueeze(1)
        ea = torch.cat([qprod, dist], 1)
        with torch.no_grad():
            a32 = O.aether_forward(sd, xs, vs, host["edges"], ea, host["charges"], return_all=True)
            a64 = O.aether_forward(sd64, xs.double(), vs.double(), host["edges"], ea.double(), host["charges"].double(),
                                   return_all=True)
            got = model(inp["h"], xs.to(dev), edges_d, vs.to(dev), ea.to(dev), inp["charges"]).cpu()
        fd = (a32["edge_attr_local"].double() - a64["edge_attr_local"]).abs()      # [E, F]
        ecol = fd.amax(dim=0)
        e_w = int(fd.amax(dim=1).argmax())
        c_w = int(fd[e_w].argmax())
        print(f"   from that path's state: oracle fp32 vs fp64 feature diff per column: " + " ".join(f"{v:.1e}" for v in ecol.tolist()))
        print(f"   worst edge {e_w} (send {int(rows[e_w])} -> recv {int(cols[e_w])}), column {c_w}: fp32 {float(a32['edge_attr_local'][e_w, c_w]):+.7f}"
              f"  fp64 {float(a64['edge_attr_local'][e_w, c_w]):+.7f}")
        o32 = float((a32["out"].double() - a64["out"]).abs().max())
        oh = float((got.double() - a64["out"]).abs().max())
        print(f"   step output from that state: oracle fp32 vs fp64 {o32:.2e}; HIP vs fp64 {oh:.2e}; HIP vs oracle fp32 "
              f"{float((got - a32['out']).abs().max()):.2e}")
        rep["jumps"][name] = {"step": t, "node": nd, "edge": e_w, "send": int(rows[e_w]), "recv": int(cols[e_w]), "column": c_w,
                              "fp32": float(a32["edge_attr_local"][e_w, c_w]), "fp64": float(a64["edge_attr_local"][e_w, c_w]),
                              "out_oracle32_vs_64": o32, "out_hip_vs_64": oh}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
