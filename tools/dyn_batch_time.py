#!/usr/bin/env python3
"""BASELINE config 4's shape: 64 inD-sized scenes (up to 40 objects, kNN k = 10) per prediction step.  One library call for
the loop (aether_dyn_rollout_batched) against the staged batched path of round 2 (three library calls + torch glue per step)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.knn import get_knn_graph_info
from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
mp = {"input_size": 4, "gpu": True, "decoder_hidden": 256, "num_edge_types": 4, "skip_first": True, "decoder_dropout": 0.0,
      "pos_representation": "cart", "no_encoder_bn": False, "encoder_dropout": 0.0, "encoder_hidden": 256,
      "encoder_rnn_hidden": 64, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 128,
      "prior_num_layers": 3, "prior_hidden_size": 128, "encoder_normalize_mode": "normalize_all", "train_data_len": 50,
      "field_hidden": 256, "gumbel_temp": 0.5}
model = AetherDynamicVars(mp, device="cuda").eval()
B, T, N = 64, 21, 40
g = torch.Generator().manual_seed(7)
inputs = torch.randn(B, T, N, 4, generator=g).cuda()
masks = torch.zeros(B, T, N)
for b in range(B):
    c = int(torch.randint(2, N + 1, (1,), generator=g))
    masks[b, :, torch.randperm(N, generator=g)[:c]] = 1
masks = masks.cuda()
burn = torch.ones(B, T, N).cuda(); burn[:, 10:] = 0
node_inds, graph_info = [], []
for b in range(B):
    ni_b, gi_b = [], []
    for t in range(T):
        nv = int(masks[b, t].sum())
        send, recv = get_knn_graph_info(inputs[b, t], masks[b, t], nv)
        gi_b.append((send, recv, torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))))
        ni_b.append(masks[b, t].nonzero()[:, -1])
    node_inds.append(ni_b); graph_info.append(gi_b)
n_obj = int(masks[:, 0].sum())
for mode in ("one call (aether_dyn_rollout_batched)", "staged (round 2: three calls + torch glue per step)"):
    model.one_call_step = mode.startswith("one")
    model.predict_future(inputs[:, :3], masks[:, :3], [n[:3] for n in node_inds], [gi[:3] for gi in graph_info], burn[:, :3])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    model.predict_future(inputs, masks, node_inds, graph_info, burn)
    ev[1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("64 scenes (%d present objects, %d edges per step), %d steps, %s: %.1f ms (%.3f ms per step, %.0f scene-steps/s)"
          % (n_obj, sum(int(gi[0][0].numel()) for gi in graph_info), T - 1, mode, dt * 1e3, dt * 1e3 / (T - 1),
             B * (T - 1) / dt), flush=True)
