#!/usr/bin/env python3
"""Time the 20-step device rollout at the headline shape (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.rollout import rollout, rollout_stepwise
from aether_amd.synthetic import make_batch
D, B, N, T = 2, 128, 20, 20
torch.manual_seed(0)
m = Aether(2 * D, 64, 0.0, D, device="cuda")
inp = make_batch(B, N, D, seed=0, device="cuda")
args = (m, inp["x"], inp["vel"], inp["edges"], inp["charges"], T)
for _ in range(3):
    rollout(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 20
for _ in range(R):
    rollout(*args)
torch.cuda.synchronize()
print("device rollout, eager launches: %.3f ms per %d-step rollout" % (1e3 * (time.perf_counter() - t0) / R, T))
for _ in range(3):
    rollout_stepwise(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    rollout_stepwise(*args)
torch.cuda.synchronize()
print("loop of module calls:           %.3f ms per %d-step rollout" % (1e3 * (time.perf_counter() - t0) / R, T))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    rollout(*args)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    traj = rollout(*args)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    g.replay()
torch.cuda.synchronize()
print("device rollout, one hipGraph:   %.3f ms per %d-step rollout" % (1e3 * (time.perf_counter() - t0) / R, T))
