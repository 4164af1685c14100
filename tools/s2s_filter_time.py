#!/usr/bin/env python3
"""Time the prior step of the seq2seq model (dominated by the filter GEMM) for a list of k-split counts."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib
from aether_amd.nn.seq2seq.encoder import Encoder
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=2)
ap.add_argument("--nodes", type=int, default=20)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--splits", type=str, default="0")
a = ap.parse_args()
D, N, B, H, R = a.dims, a.nodes, a.batch, 512, 128
lib = _lib.load()
eparams = {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H, "encoder_rnn_hidden": R,
           "encoder_rnn_type": "lstm", "input_size": 2 * D, "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
           "prior_num_layers": 3, "prior_hidden_size": 256, "use_3d": D == 3,
           "pos_representation": "polar" if D == 2 else "cart"}
enc = Encoder(eparams, device="cuda").eval()
E = N * (N - 1)
x = torch.randn(B, N, 2 * D, device="cuda")
f = torch.randn(B, N, D, device="cuda")
ps = (torch.zeros(B, E, R, device="cuda"), torch.zeros(B, E, R, device="cuda"))
def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for sp in [int(v) for v in a.splits.split(",")]:
    lib.aether_set_option(b"filter_splits", sp)
    enc._cache.pop("ws", None)
    t = timed(lambda: enc.single_step_forward(x, ps, f))
    print("splits %d: prior step %.3f ms" % (sp, t), flush=True)
