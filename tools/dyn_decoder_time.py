#!/usr/bin/env python3
"""Time the variable-N decoder step at inD-like sizes (scripts/ind_aether.sh: decoder_hidden 256,
4 edge types, the first skipped; kNN graph, k = 10) for scenes of 20 / 40 / 400 present objects."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.knn import get_knn_graph_info
from aether_amd.nn.dynamicvars.decoder import Decoder
params = {"input_size": 4, "gpu": True, "decoder_hidden": 256, "num_edge_types": 4, "skip_first": True,
          "decoder_dropout": 0.0, "pos_representation": "cart"}
dec = Decoder(params, device="cuda").eval()
for N in (20, 40, 400):
    g = torch.Generator().manual_seed(N)
    inputs = torch.randn(1, N, 4, generator=g).cuda()
    hidden = (torch.randn(1, N, 256, generator=g) * 0.3).cuda()
    field = (torch.randn(1, N, 2, generator=g) * 0.3).cuda()
    masks = torch.ones(N).cuda()
    send, recv = get_knn_graph_info(inputs[0], masks, N)
    e2n = torch.argsort(recv, stable=True).view(-1, 10)
    edges = torch.nn.functional.one_hot(torch.randint(0, 4, (send.numel(),), generator=g), 4).float().unsqueeze(0).cuda()
    fn = lambda: dec(inputs, hidden, edges, masks, (send, recv, e2n), field)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(50):
        get_knn_graph_info(inputs[0], masks, N)
    torch.cuda.synchronize()
    dk = (time.perf_counter() - t0) / 50
    print("N=%3d (%5d edges): decoder step %.3f ms, kNN graph %.3f ms" % (N, send.numel(), dt * 1e3, dk * 1e3))
