#!/usr/bin/env python3
"""Soak the fused forward (split-GEMM) + fused backward at the headline shape: every step's output and all 47
gradients must be bit-identical to the first step's (fixed weights, no optimizer).  A rare data hazard (e.g. a
matrix-core operand read before its producer landed) shows up here as a mismatch count > 0."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch
N_IT = int(sys.argv[1]) if len(sys.argv) > 1 else 600
D = 2
torch.manual_seed(1)
m = Aether(4, 64, 0.0, D, device="cuda")
a = make_batch(128, 20, D, seed=0, device="cuda")
def step():
    m.zero_grad(set_to_none=True)
    o = m(a["h"], a["x"], a["edges"], a["vel"], a["edge_attr"], a["charges"])
    torch.nn.functional.mse_loss(o, a["target"]).backward()
    return o.detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
o0, g0 = step()
bad_o = bad_g = 0
t0 = time.perf_counter()
for it in range(N_IT):
    o, g = step()
    bad_o += int(not torch.equal(o, o0))
    bad_g += int(not torch.equal(g, g0))
torch.cuda.synchronize()
print(f"{N_IT} training steps (forward KEEP + fused backward) in {time.perf_counter() - t0:.1f} s: output mismatches {bad_o}, "
      f"gradient mismatches {bad_g}")
# inference kernel, many launches
with torch.no_grad():
    r0 = m(a["h"], a["x"], a["edges"], a["vel"], a["edge_attr"], a["charges"]).clone()
    bad = 0
    for it in range(20000):
        out = m(a["h"], a["x"], a["edges"], a["vel"], a["edge_attr"], a["charges"])
        if it % 20 == 0:
            bad += int(not torch.equal(out, r0))
torch.cuda.synchronize()
print(f"20000 inference launches, 1000 compared: mismatches {bad}")
# the captured training step (library loss + AdamW, aether_amd.training.GraphedTrainStep): two runs from the same start
from aether_amd.training import GraphedTrainStep
finals = []
for run in range(2):
    torch.manual_seed(1)
    m2 = Aether(4, 64, 0.0, D, device="cuda")
    start = {k: v.detach().clone() for k, v in m2.state_dict().items()}
    gs = GraphedTrainStep(m2, [a["h"], a["x"], a["edges"], a["vel"], a["edge_attr"], a["charges"]], a["target"], warmup=1)
    m2.load_state_dict(start)
    gs.optimizer.reset_state()
    for _ in range(400):
        gs.step()
    torch.cuda.synchronize()
    finals.append((float(gs.loss), torch.cat([p.detach().reshape(-1) for p in m2.parameters()]).clone(), gs.optimizer.steps_taken()))
same = torch.equal(finals[0][1], finals[1][1]) and finals[0][0] == finals[1][0]
print(f"2 x 400 captured training steps from the same start: final loss {finals[0][0]:.6g} / {finals[1][0]:.6g}, "
      f"optimizer steps {finals[0][2]} / {finals[1][2]}, weights bit-identical: {same}")
sys.exit(1 if (bad or bad_o or bad_g or not same) else 0)
