#!/usr/bin/env python3
"""Host-side cost of one eager training step (forward, MSE, backward, Adam) as the reference's runner
runs it (experiments/lorentz/main.py:247-291): wall time per step and the top cProfile entries."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
D, B, N = 2, 128, 20
m = Aether(2 * D, 64, 0.0, D, device="cuda")
inp = make_batch(B, N, D, seed=0, device="cuda")
opt = torch.optim.Adam(m.parameters(), lr=5e-4)
def step():
    opt.zero_grad()
    out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    loss = torch.nn.functional.mse_loss(out, inp["target"])
    loss.backward()
    opt.step()
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step()
torch.cuda.synchronize()
print("eager training step: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 100))
# the same loop with the library's loss and optimizer (aether_amd.optim: one launch each)
from aether_amd.optim import FusedAdamW, mse_loss_grad
opt2 = FusedAdamW(m.parameters(), lr=5e-4, weight_decay=1e-12)
def step2():
    opt2.zero_grad()
    out = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    loss, grad = mse_loss_grad(out, inp["target"])
    out.backward(grad)
    opt2.step()
for _ in range(10):
    step2()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    step2()
torch.cuda.synchronize()
print("eager training step, aether_amd.optim loss + AdamW: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 100))
for fn in (step, step2):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
