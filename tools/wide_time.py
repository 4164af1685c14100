#!/usr/bin/env python3
"""Time the hidden_size > 64 path (csrc/wide.h) at the headline shape: forward (eager / hipGraph) and a training step.
Usage: wide_time.py [H ...]   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import contextlib, io, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.synthetic import make_batch

def timed(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

for H in [int(a) for a in sys.argv[1:]] or [128, 256]:
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = Aether(4, H, 0.0, 2, device="cuda").eval()
    inp = make_batch(128, 20, 2, seed=0, device="cuda")
    E = inp["edges"][0].numel()
    call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    with torch.no_grad():
        ms_e = timed(call, 30)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call()
        ms_g = timed(g.replay, 30)
    flops = E * (2 * (32 * H + H * H) + 6 * (3 * H * H + H * H)) + 2560 * (4 * 2 * (2 * H * H * 2) + 3 * 2 * 2 * H * H + 2 * 2 * H * H)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4)
    def tstep():
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(call(), inp["target"]).backward()
        opt.step()
    ms_t = timed(tstep, 10)
    print(f"H={H}: forward eager {ms_e:.3f} ms, hipgraph {ms_g:.3f} ms ({flops / (ms_g * 1e-3) / 1e12:.1f} algorithmic TFLOP/s, "
          f"{4 * E / (ms_g * 1e-3) / 1e9:.2f} G edge-messages/s); training step (eager, torch AdamW) {ms_t:.3f} ms", flush=True)
