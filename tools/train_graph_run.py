#!/usr/bin/env python3
"""Replay the captured training step (GraphedTrainStep) at cfg2; under rocprofv3 --kernel-trace the replays' kernels show
which launches a step consists of (tools/train_graph_prof.sh).  Prints the wall time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
from aether_amd.training import GraphedTrainStep
D, B, N = 2, 128, 20
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(1)
m = Aether(2 * D, 64, 0.0, D, device="cuda")
inp = make_batch(B, N, D, seed=0, device="cuda")
step = GraphedTrainStep(m, [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]], inp["target"])
for _ in range(20):
    step.step()
torch.cuda.synchronize()
print("MARK replays start", flush=True)
t0 = time.perf_counter()
for _ in range(steps):
    step.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("captured training step: %.4f ms per step, loss %.6f" % (dt * 1e3, float(step.loss)))
