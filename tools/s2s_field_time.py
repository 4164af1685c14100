#!/usr/bin/env python3
"""Time the seq2seq field query (row A8) on the GPU box: one step (B*N points) and a whole trajectory
batch (B*N*T points, as predict_future queries it)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.field import FieldQuery
D, H = 2, 512
m = FieldQuery(D, H, device="cuda")
for name, shape in (("one step  B=128 N=20", (128, 20, 2 * D)), ("49 steps  B=128 N=20 T=49", (128, 20, 49, 2 * D))):
    x = torch.randn(*shape, device="cuda")
    n = x.numel() // x.shape[-1]
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = n * (2.0 * H * H * 2 + 2.0 * H * D + 2.0 * D * H / 2)
    print("%-28s %8d points  %.3f ms  %.1f TFLOP/s (fp32 MFMA peak 157.3)" % (name, n, dt * 1e3, flop / dt / 1e12))
