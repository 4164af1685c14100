#!/usr/bin/env python3
"""Diagnostic: where does k_fused spend its time?  Builds the stamped library variant, runs the
headline workload and prints the median (over workgroups) time between phase stamps.
Run on the GPU box:  python tools/fused_phases.py [--dims 2] [--batch 128] [--nodes 20]"""
import argparse, contextlib, io, os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from aether_amd import build as B, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=2); ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--nodes", type=int, default=20); ap.add_argument("--keep", action="store_true", help="time the save-for-backward variant")
a = ap.parse_args()
_lib.LIB_PATH = B.build_diagnostic()
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    m = Aether(2 * a.dims, 64, 0.0, a.dims, device="cuda")
inp = make_batch(a.batch, a.nodes, a.dims, seed=0, device="cuda")
if a.keep:
    m.flags = _lib.FLAG_KEEP_INTERMEDIATES
Nn, E = inp["x"].shape[0], inp["edges"][0].numel()
with torch.no_grad():
    for _ in range(20):
        m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
torch.cuda.synchronize()
G = m.prepare_graph(inp["edges"], Nn)[1].n_groups
dst = torch.zeros(4096, 512, device="cuda")
lib = _lib.load()
_lib.check(lib.aether_debug_fetch(b"stamps", a.dims, Nn, E, m._last_ws.data_ptr(), dst.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream), "fetch stamps")
torch.cuda.synchronize()
st = dst.cpu().numpy()[:min(G, 4096)]
med = np.median(st, axis=0)
names = {0: "entry", 1: "w1 staged (issued)", 2: "field+frames+x0", 3: "edge features", 40: "out mlp + store"}
for l in range(4):
    b = 4 + 8 * l
    names[b + 2] = f"L{l+1} edge tiles (wave 0 done)"
    names[b + 3] = f"L{l+1} barrier (all tiles done)"
    names[b + 7] = f"L{l+1} n = x + mean"
    names[b + 4] = f"L{l+1} u = silu(W3 n)"
    names[b + 5] = f"L{l+1} x = n + W4 u"
    names[b + 6] = f"L{l+1} P_s, P_r"
prev = 0.0
print(f"groups={G}  (median over workgroups, microseconds)")
order = [0, 1, 2, 3] + [4 + 8 * l + o for l in range(4) for o in (2, 3, 7, 4, 5, 6)] + [40]
for k in order:
    if med[k] == 0 and k != 0:
        continue
    print(f"  {names[k]:32s} t={med[k]:8.2f}  d={med[k]-prev:7.2f}")
    prev = med[k]
clk = np.median(st[:, 41] / st[:, 40])
print(f"  shader clock during the kernel: {clk:.0f} MHz (s_memtime cycles / wall us)")
print("  layer-2 per-wave tile timeline, cycles since kernel entry (median over workgroups):")
print("   wave round   start   gemm1   silu1   gemm2  silu2+stage  reduce   | total")
for w in range(8):
    for r in range(3):
        v = med[64 + w * 24 + r * 8: 64 + w * 24 + r * 8 + 6]
        if v[5] == 0: continue
        d = np.diff(v)
        print(f"   {w:4d} {r:5d} {v[0]:8.0f} " + " ".join(f"{x:7.0f}" for x in d) + f"    | {v[5]-v[0]:6.0f}")
print(f"  kernel span (max over WGs of last stamp): {st[:, 40].max():.2f} us; min start->end {st[:,40].min():.2f}")
