#!/usr/bin/env python3
"""Time the seq2seq augmented localizer (row A9) on the GPU box; bytes moved vs the HBM roof."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.localizer import AugmentedLocalizer
for D in (2, 3):
    for (B, N) in ((128, 20), (6272, 20)):          # one step; 49 steps' worth of graphs in one call
        loc = AugmentedLocalizer(N, use_3d=D == 3)
        x = torch.randn(B, N, 3 * D, device="cuda")
        for _ in range(5):
            loc(x)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            loc(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        E, nf = B * N * (N - 1), 4 * D + D * (D - 1) // 2
        nbytes = 4 * (E * (2 * nf + 3 * D + D + D * (D - 1) // 2) + B * N * (3 * D + 3 * D + nf + D * D))
        print("D=%d B=%5d N=%d: %8d edges  %.3f ms  %.2f G edges/s  %.2f TB/s of output+input (HBM ~6.3 achievable)"
              % (D, B, N, E, dt * 1e3, E / dt / 1e9, nbytes / dt / 1e12))
