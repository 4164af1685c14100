#!/usr/bin/env python3
"""Time the seq2seq dynamic-field variant's pieces at the gravitational runner's sizes
(scripts/gravitational_field_3d_aether.sh: 3-D, 5 objects, 49 burn-in steps, encoder_hidden = graph_hidden =
mlp_hidden = 512): the once-per-sequence graph summary + FiLM modulation, and the per-step FiLM field query."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.nn.seq2seq.dynamic_field_aether import DynamicFieldAether

D, B, N, T, H, GH, MH = 3, 128, 5, 49, 512, 512, 512
params = {"num_vars": N, "num_edge_types": 2, "encoder_dropout": 0.0, "encoder_hidden": H, "encoder_rnn_hidden": 128,
          "encoder_rnn_type": "lstm", "input_size": 2 * D, "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
          "prior_num_layers": 3, "prior_hidden_size": 256, "use_3d": True, "pos_representation": "cart", "gpu": True,
          "decoder_hidden": 256, "skip_first": False, "decoder_dropout": 0.0, "gumbel_temp": 0.5, "graph_hidden": GH,
          "mlp_hidden": MH, "field": None}
torch.manual_seed(0)
model = DynamicFieldAether(params, device="cuda").eval()
x = torch.randn(B, N, T, 2 * D, device="cuda")
x1 = torch.randn(B, N, 2 * D, device="cuda")


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


summary = model.graph_pooler(x)
R = B * N * T
d = 2 * D + GH
flop_sum = R * 2.0 * (2 * D * GH + GH * 3 * GH) + T * B * N * 2.0 * GH * 3 * GH + R * 2.0 * (2 * d * GH + GH + GH * GH)
dt = timed(lambda: model.graph_pooler(x), 10)
print("graph summary   B=%d N=%d T=%d H=%d : %.3f ms  (%.1f TFLOP/s; %d GRU steps of %d rows)" %
      (B, N, T, GH, dt * 1e3, flop_sum / dt / 1e12, T, B * N))


def fresh_mod():
    model._mod = None
    model.predict_field(x1, summary)


for name, inp in (("one step (B*N points)", x1), ("burn-in (B*N*T points)", x)):
    n = inp.numel() // inp.shape[-1]
    dt = timed(lambda: model.predict_field(inp, summary), 30)
    flop = n * 2.0 * (H * MH + MH * MH + MH * D)
    print("FiLM field query %-24s %7d points : %.3f ms  (%.1f TFLOP/s)" % (name, n, dt * 1e3, flop / dt / 1e12))
dt = timed(fresh_mod, 30)
print("FiLM modulation + one-step query (cache miss)          : %.3f ms" % (dt * 1e3))
U = torch.rand(T + 20, B, N * (N - 1), 2, device="cuda")
inputs = torch.randn(B, T + 1, N, 2 * D, device="cuda")
for name, graph in (("step by step", False), ("captured step graph", True)):
    dt = timed(lambda: model.predict_future(inputs, 20, uniform=U, graph=graph), 3)
    print("predict_future  %d burn-in + 20 prediction steps, %-20s: %.1f ms  (%.2f ms per step)" %
          (T, name, dt * 1e3, dt * 1e3 / (T + 20)))
