#!/usr/bin/env python3
"""Which output of the fused seq2seq step differs between runs, and under which library option (diagnostic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd import _lib
from aether_amd.nn.seq2seq.aether import Aether
D, N, B, hd = 2, 20, 128, 512
params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": hd, "num_edge_types": 2,
          "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3, "encoder_dropout": 0.0, "encoder_hidden": 512,
          "encoder_rnn_hidden": 128, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
          "prior_num_layers": 3, "prior_hidden_size": 256, "pos_representation": "polar" if D == 2 else "cart",
          "gumbel_temp": 0.5, "rff_std": 1.0}
lib = _lib.load()
for opt in (None, ("gemm_split", 0), ("filter_splits", 1), ("filter_splits", 4)):
    if opt:
        _lib.check(lib.aether_set_option(opt[0].encode(), opt[1]), "opt")
    torch.manual_seed(0)
    m = Aether(params, device="cuda").eval()
    E = N * (N - 1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, 2 * D, generator=g).cuda()
    dh = (torch.randn(B, N, hd, generator=g) * 0.3).cuda()
    ps = ((torch.randn(B, E, 128, generator=g) * 0.3).cuda(), (torch.randn(B, E, 128, generator=g) * 0.3).cuda())
    u = torch.rand(B, E, 2, generator=g).cuda()
    first = m._fused_step(x, dh, ps, u)
    names = ["x_out", "dh_out", "h1", "c1", "edges"]
    cnt = [0] * 5
    worst = [0.0] * 5
    for _ in range(60):
        out = m._fused_step(x, dh, ps, u)
        outs = [out[0], out[1], out[2][0], out[2][1], out[3]]
        firsts = [first[0], first[1], first[2][0], first[2][1], first[3]]
        for k in range(5):
            if not torch.equal(outs[k], firsts[k]):
                cnt[k] += 1
                worst[k] = max(worst[k], float((outs[k] - firsts[k]).abs().max()))
    print(opt, {n: (c, w) for n, c, w in zip(names, cnt, worst)}, flush=True)
    if opt:
        _lib.check(lib.aether_set_option(opt[0].encode(), 1 if opt[0] == "gemm_split" else 0), "opt")
