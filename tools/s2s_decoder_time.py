#!/usr/bin/env python3
"""Time one step of the seq2seq decoder path on the GPU box: field query + decoder step (B=128, N=20, h=512)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.decoder import RecurrentDecoder
from aether_amd.nn.seq2seq.field import FieldQuery
D, N, B, H = 2, 20, 128, 512
params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": H, "num_edge_types": 2,
          "skip_first": False, "decoder_dropout": 0.0, "use_3d": False}
dec = RecurrentDecoder(params, device="cuda")
fq = FieldQuery(D, H, device="cuda")
x = torch.randn(B, N, 2 * D, device="cuda")
hid = torch.zeros(B, N, H, device="cuda")
E = N * (N - 1)
z = torch.nn.functional.one_hot(torch.randint(0, 2, (B, E), device="cuda"), 2).float()
def step():
    f, _ = fq(x)
    return dec(x, hid, z, f)
for _ in range(3):
    step()
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
edges, nodes = B * E, B * N
flop = edges * 2 * (2.0 * H * H * 2 + 2.0 * 24 * H) + nodes * (2 * 2 * 2.0 * H * H + 7 * 2.0 * H * H + 3 * 2.0 * 16 * H + 2 * 2.0 * H * H + 2.0 * H * H * 2)
print("field + decoder step: %.3f ms  (%.1f M edge-steps/s, ~%.0f GFLOP executed -> %.1f TFLOP/s)" % (dt * 1e3, edges / dt / 1e6, flop / 1e9, flop / dt / 1e12))
