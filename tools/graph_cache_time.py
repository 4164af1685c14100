#!/usr/bin/env python3
"""What a runner that rebuilds the edge index every batch (experiments/lorentz/main.py:211-212) pays per
call: same tensors, memoized get_edges, fresh tensors with equal content, and a true graph rebuild."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.edges import get_edges
from aether_amd.synthetic import make_batch
D, B, N = 2, 128, 20
m = Aether(4, 64, 0.0, D, device="cuda")
inp = make_batch(B, N, D, seed=0, device="cuda")
def call(edges):
    with torch.no_grad():
        return m(inp["h"], inp["x"], edges, inp["vel"], inp["edge_attr"], inp["charges"])
def timed(name, make, reps=30):
    for _ in range(3):
        call(make())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call(make())
    torch.cuda.synchronize()
    print("%-44s %.3f ms per call" % (name, 1e3 * (time.perf_counter() - t0) / reps))
timed("same tensors", lambda: inp["edges"])
timed("get_edges (memoized)", lambda: get_edges(B, N, device="cuda"))
timed("get_edges(cache=False): equal content", lambda: get_edges(B, N, device="cuda", cache=False))
perm = [torch.randperm(inp["edges"][0].numel(), device="cuda") for _ in range(40)]
it = iter(perm)
def shuffled():
    p = next(it)
    return [inp["edges"][0][p], inp["edges"][1][p]]      # new content: full rebuild (edge_attr order no longer matches: timing only)
timed("new edge order every call: graph rebuild", shuffled)
