#!/usr/bin/env python3
"""Time one autoregressive step of the seq2seq model on the GPU box (B=128, N=20, h=512):
field query -> prior step -> hard Gumbel sample -> decoder step (predict_future, aether.py:176-185)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.decoder import RecurrentDecoder
from aether_amd.nn.seq2seq.encoder import Encoder, gumbel_softmax_hard
from aether_amd.nn.seq2seq.field import FieldQuery
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, default=2)
ap.add_argument("--nodes", type=int, default=20)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--decoder-hidden", type=int, default=512)
ap.add_argument("--filter-wg-target", type=int, default=0)
ap.add_argument("--linear-small-wgs", type=int, default=-1)
ap.add_argument("--linear-kwaves", type=int, default=0)
a = ap.parse_args()
D, N, B, H, R = a.dims, a.nodes, a.batch, 512, 128
if a.linear_small_wgs >= 0:
    from aether_amd import _lib
    _lib.load().aether_set_option(b"linear_small_wgs", a.linear_small_wgs)
if a.linear_kwaves:
    from aether_amd import _lib
    _lib.load().aether_set_option(b"linear_kwaves", a.linear_kwaves)
if a.filter_wg_target:
    from aether_amd import _lib
    _lib.load().aether_set_option(b"filter_wg_target", a.filter_wg_target)
dparams = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": a.decoder_hidden, "num_edge_types": 2,
           "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3}
eparams = {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H, "encoder_rnn_hidden": R,
           "encoder_rnn_type": "lstm", "input_size": 2 * D, "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
           "prior_num_layers": 3, "prior_hidden_size": 256, "use_3d": D == 3,
           "pos_representation": "polar" if D == 2 else "cart"}
dec = RecurrentDecoder(dparams, device="cuda")
enc = Encoder(eparams, device="cuda").eval()
fq = FieldQuery(D, H, device="cuda")
E = N * (N - 1)
x = torch.randn(B, N, 2 * D, device="cuda")
hid = torch.zeros(B, N, a.decoder_hidden, device="cuda")
ps = (torch.zeros(B, E, R, device="cuda"), torch.zeros(B, E, R, device="cuda"))
U = torch.rand(B, E, 2, device="cuda")
def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
f, _ = fq(x)
logits, _ = enc.single_step_forward(x, ps, f)
z = gumbel_softmax_hard(logits, U, 0.5)
t_field = timed(lambda: fq(x))
t_prior = timed(lambda: enc.single_step_forward(x, ps, f))
t_dec = timed(lambda: dec(x, hid, z, f))
def step():
    f, _ = fq(x)
    lg, s2 = enc.single_step_forward(x, ps, f)
    zz = gumbel_softmax_hard(lg, U, 0.5)
    return dec(x, hid, zz, f)
t_all = timed(step)
edges = B * E
print("field %.3f ms | prior step %.3f ms (filter GEMM %.0f GFLOP) | decoder step %.3f ms | whole step %.3f ms = %.2f M edge-steps/s"
      % (t_field, t_prior, edges * (24 if D == 2 else 39) * H * H * 2 / 1e9, t_dec, t_all, edges / t_all / 1e3))
