#!/usr/bin/env python3
"""Bit-stability soak of the fused seq2seq step (filter GEMM, split GEMMs, job-table launches): the same inputs, many runs,
every output compared bitwise with the first run."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.aether import Aether
ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=200)
a = ap.parse_args()
bad = 0
for (D, N, B, hd) in ((3, 5, 128, 256), (2, 20, 128, 512), (2, 20, 44, 256), (3, 12, 20, 128)):
    params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": hd, "num_edge_types": 2,
              "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3, "encoder_dropout": 0.0, "encoder_hidden": 512,
              "encoder_rnn_hidden": 128, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
              "prior_num_layers": 3, "prior_hidden_size": 256, "pos_representation": "polar" if D == 2 else "cart",
              "gumbel_temp": 0.5, "rff_std": 1.0}
    torch.manual_seed(0)
    m = Aether(params, device="cuda").eval()
    E = N * (N - 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, 2 * D, generator=g).cuda()
    dh = (torch.randn(B, N, hd, generator=g) * 0.3).cuda()
    ps = ((torch.randn(B, E, 128, generator=g) * 0.3).cuda(), (torch.randn(B, E, 128, generator=g) * 0.3).cuda())
    u = torch.rand(B, E, 2, generator=g).cuda()
    first = m._fused_step(x, dh, ps, u)
    mism = 0
    for _ in range(a.runs):
        out = m._fused_step(x, dh, ps, u)
        same = (torch.equal(out[0], first[0]) and torch.equal(out[1], first[1]) and torch.equal(out[2][0], first[2][0]) and
                torch.equal(out[2][1], first[2][1]) and torch.equal(out[3], first[3]))
        mism += 0 if same else 1
    print("D=%d N=%d B=%d hd=%d: %d of %d runs differ from the first" % (D, N, B, hd, mism, a.runs), flush=True)
    bad += mism
sys.exit(1 if bad else 0)
