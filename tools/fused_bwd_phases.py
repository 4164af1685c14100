#!/usr/bin/env python3
"""Diagnostic: where does k_fused_bwd spend its time?  Stamped library variant (build_diagnostic), headline
workload, median over workgroups of the time between phase stamps.  Run on the GPU box."""
import argparse, contextlib, io, os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from aether_amd import build as B, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=2); ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--nodes", type=int, default=20)
a = ap.parse_args()
_lib.LIB_PATH = B.build_diagnostic()
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    m = Aether(2 * a.dims, 64, 0.0, a.dims, device="cuda")
inp = make_batch(a.batch, a.nodes, a.dims, seed=0, device="cuda")
Nn, E = inp["x"].shape[0], inp["edges"][0].numel()
for _ in range(5):
    m.zero_grad(set_to_none=True)
    o = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    torch.nn.functional.mse_loss(o, inp["target"]).backward()
torch.cuda.synchronize()
G = m.prepare_graph(inp["edges"], Nn)[1].n_groups
dst = torch.zeros(4096, 512, device="cuda")
lib = _lib.load()
_lib.check(lib.aether_debug_fetch(b"stamps", a.dims, Nn, E, m._last_ws.data_ptr(), dst.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "fetch stamps")
torch.cuda.synchronize()
st = dst.cpu().numpy()[:min(G, 4096)]
med = np.median(st, axis=0)
print(f"groups={G} (median over workgroups, microseconds since kernel entry)")
print(f"  prologue done                 t={med[0]:8.2f}")
prev = med[0]
for li, l in enumerate((4, 3, 2, 1)):
    b = 8 * li
    for k, name in ((40 + 4 * li - b, "  node: fragment loads issued"), (41 + 4 * li - b, "  node: W_e DMA issued"), (42 + 4 * li - b, "  node: P rows in LDS"), (6, "  node: loads issued + weights staged"), (7, "  node: stage A (pre_u, du) done"), (1, "node update backward + staging"), (2, "edge tiles (all waves)"), (3, "incidence sums + hand-off"),
                    (4, "weight-gradient reduce + partial"), (5, "dx GEMM")):
        t = med[b + k]
        print(f"  L{l} {name:34s} t={t:8.2f}  d={t-prev:7.2f}")
        prev = t
    for w in range(4):
        row = [med[64 + w * 100 + 20 * li + 4 * r] for r in (2, 1, 0)] + [med[64 + w * 100 + 20 * li + 16]]
        print(f"       wave {w}: tile starts (r=2,1,0) and end: " + " ".join(f"{x:8.2f}" for x in row))
names = ["tile start", "GEMM1 (W_e e_prev)", "sigmoid, stage h", "GEMM2 (W2 h)", "dpre2, stage", "GEMM3 (W2^T)", "dW2 product",
         "G, stage", "dW_e product", "incidence sums"]
for w in range(4):
    base = 64 + w * 100
    v = [med[base + 20 * 1 + 4 * 2]] + [med[base + 90 + k] for k in range(9)]
    print(f"  layer 3, wave {w}, first tile (us): " + " ".join(f"{nm}={b - a:.2f}" for nm, a, b in zip(names[1:], v[:-1], v[1:])))
print(f"  kernel span: max {st[:, 8 * 3 + 5].max():.2f} us, median {med[8 * 3 + 5]:.2f}")
