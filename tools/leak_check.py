#!/usr/bin/env python3
"""Run the main entry points in loops and report device-memory growth (torch allocator) between the 10th and the last
iteration: workspaces and graph views are cached per shape, so the numbers must be flat."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.state2state.aether import Aether
from aether_amd.nn.state2state.dynamic_field_aether import DynamicFieldAether
from aether_amd.nn.seq2seq.aether import Aether as S2S
from aether_amd.knn import knn_edges
from aether_amd.synthetic import make_batch


def loop(name, fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    print("%-42s %6d iterations: allocated %8.1f -> %8.1f MiB (%+.3f)" % (name, n, m0 / 2**20, m1 / 2**20, (m1 - m0) / 2**20))


D, B, N = 2, 128, 20
inp = make_batch(B, N, D, seed=0, device="cuda")
m = Aether(2 * D, 64, 0.0, D, device="cuda")
args = (inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
with torch.no_grad():
    loop("state2state forward", lambda: m(*args))
opt = torch.optim.Adam(m.parameters(), lr=1e-4)


def train():
    opt.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(m(*args), inp["target"]).backward()
    opt.step()


loop("state2state training step", train)
loop("state2state 20-step rollout", lambda: m.rollout(inp["x"], inp["vel"], inp["edges"], inp["charges"], 20), 100)
dm = DynamicFieldAether(2 * D, 64, 0.0, D, device="cuda")
dopt = torch.optim.Adam(dm.parameters(), lr=1e-4)


def dtrain():
    dopt.zero_grad(set_to_none=True)
    torch.nn.functional.mse_loss(dm(*args, N), inp["target"]).backward()
    dopt.step()


loop("dynamic-field training step", dtrain)
sp = {"num_vars": 5, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": 128, "encoder_rnn_hidden": 64,
      "encoder_rnn_type": "lstm", "input_size": 4, "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 64, "prior_num_layers": 3,
      "prior_hidden_size": 64, "use_3d": False, "pos_representation": "polar", "gpu": True, "decoder_hidden": 128,
      "skip_first": False, "decoder_dropout": 0.0, "gumbel_temp": 0.5}
s2s = S2S(sp, device="cuda").eval()
seq = torch.randn(16, 12, 5, 4, device="cuda")
loop("seq2seq predict_future (11 + 5 steps)", lambda: s2s.predict_future(seq, 5), 40)
loop("seq2seq calculate_loss (eval)", lambda: s2s.calculate_loss(seq, is_train=False), 40)
x = torch.randn(64, 49, 40, 4, device="cuda")
msk = (torch.rand(64, 49, 40, device="cuda") < 0.6).float()
loop("kNN edges (3136 scenes)", lambda: knn_edges(x, msk), 100)
