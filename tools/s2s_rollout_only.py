#!/usr/bin/env python3
"""Device-side seq2seq rollouts only (for rocprofv3 --kernel-trace --stats): D=3, N=5, B=128, h=512, hd=256 by default."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.aether import Aether
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=3)
ap.add_argument("--nodes", type=int, default=5)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--decoder-hidden", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--opt", action="append", default=[], help="library option name=value (aether_set_option)")
a = ap.parse_args()
from aether_amd import _lib
for kv in a.opt:
    k, v = kv.split("=")
    _lib.check(_lib.load().aether_set_option(k.encode(), int(v)), "set_option " + kv)
D, N, B, H, R = a.dims, a.nodes, a.batch, 512, 128
params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": a.decoder_hidden, "num_edge_types": 2,
          "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3, "encoder_dropout": 0.0, "encoder_hidden": H,
          "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
          "prior_num_layers": 3, "prior_hidden_size": 256, "pos_representation": "polar" if D == 2 else "cart",
          "gumbel_temp": 0.5, "rff_std": 1.0}
torch.manual_seed(0)
m = Aether(params, device="cuda").eval()
E, T = N * (N - 1), a.steps
x = torch.randn(B, N, 2 * D, device="cuda")
dh = torch.zeros(B, N, a.decoder_hidden, device="cuda")
ps = (torch.zeros(B, E, R, device="cuda"), torch.zeros(B, E, R, device="cuda"))
U = torch.rand(T, B, E, 2, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    m.predict_from_state(x, dh, ps, T, uniform=U)
torch.cuda.synchronize()
print("rollout %.3f ms per step steps=%d" % ((time.perf_counter() - t0) / (a.reps * T) * 1e3, a.reps * T))
