#!/usr/bin/env python3
"""Soak the split-mode fused kernel: many back-to-back launches on a reused workspace (no memset node),
results must stay bit-identical; also alternating shapes and a concurrent copy stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
D = 2
m = Aether(4, 64, 0.0, D, device="cuda")
a = make_batch(128, 20, D, seed=0, device="cuda")
b = make_batch(40, 20, D, seed=1, device="cuda")
def run(inp):
    with torch.no_grad():
        return m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
ref_a, ref_b = run(a).clone(), run(b).clone()
side = torch.cuda.Stream()
junk = torch.empty(64 << 20, device="cuda")
t0 = time.perf_counter()
bad = 0
for it in range(30000):
    out = run(a)
    if it % 500 == 0:
        bad += int(not torch.equal(out, ref_a))
        with torch.cuda.stream(side):
            junk.zero_()                          # unrelated traffic on another stream
    if it % 1500 == 0:
        bad += int(not torch.equal(run(b), ref_b))
torch.cuda.synchronize()
print("30000 launches in %.2f s, mismatches: %d" % (time.perf_counter() - t0, bad))
sys.exit(1 if bad else 0)
