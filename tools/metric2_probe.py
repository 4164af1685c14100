#!/usr/bin/env python3
"""Metric 2 (oracle/metric2.py) at the headline shape for seed-1 weights and after training: how many training steps
make the 20-step rollout well conditioned?  Diagnostic (imports the oracle: not product code)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from oracle import metric2 as M2

dev = torch.device("cuda")
B, N, D = 128, 20, 2
data = M2.simulate(dev, B, N, D)
cfgs = [(0, 0.0)] + [(int(a.split(":")[0]), float(a.split(":")[1])) for a in sys.argv[1:]]
for steps, lr in cfgs:
    torch.manual_seed(1)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = Aether(2 * D, 64, 0.0, D, device=dev)
    loss = M2.train_on_frames(m, data, steps, lr) if steps else None
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    t0 = time.time()
    r = M2.report(sd, m, data)
    keep = {"steps": steps, "lr": lr, "loss": loss, "mse64": r["mse_oracle_fp64_steps_1_10_20"],
            "hip": {k: r["hip"][k] for k in ("max_rel_mse_difference", "trajectory_max_rel_err", "trajectory_max_rel_err_outside_cut_exposed", "first_step_above_tolerance")},
            "o32": {k: r["oracle_fp32"][k] for k in ("max_rel_mse_difference", "trajectory_max_rel_err", "trajectory_max_rel_err_outside_cut_exposed", "first_step_above_tolerance")},
            "exposed": r["cut_exposed_graphs"], "t": round(time.time() - t0, 1)}
    print(json.dumps(keep), flush=True)
