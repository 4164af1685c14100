#!/usr/bin/env python3
"""Time the kNN edge builder at cfg4-like sizes: 64 scenes x 49 time steps of up to 40 agents,
k = 10, next to the reference's formulation (cdist + topk + boolean filters) run with torch ops on the same GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aether_amd.knn import knn_edges


def torch_formulation(x, masks, k=10):
    """The steps of aether_dynamicvars.py:559-586 with torch ops (timing comparison only)."""
    _x, _m = x.flatten(0, -3), masks.flatten(0, -2)
    n = _x.shape[-2]
    per = _m.sum(-1).long()
    cum = torch.cat([torch.zeros_like(per[[0]]), per.cumsum(0)])[:-1]
    k = min(k, n - 1)
    dm = torch.cdist(_x[..., :2], _x[..., :2])
    dm.masked_fill_((_m.unsqueeze(-1) * _m.unsqueeze(-2)) == 0, float("inf"))
    ar = torch.arange(n, device=x.device)
    dm[:, ar, ar] = float("inf")
    md, ri = dm.topk(dim=-1, k=k, largest=False)
    si = ar.unsqueeze(1).unsqueeze(0).repeat(_x.shape[0], 1, k)
    sp = _m.cumsum(-1).long() - 1
    rows = torch.arange(_x.shape[0], device=x.device).unsqueeze(1)
    keep = ~torch.isinf(md.flatten())
    ri = (sp[rows, ri.flatten(1)].reshape(*ri.shape) + cum[:, None, None]).flatten()[keep]
    si = (sp[rows, si.flatten(1)].reshape(*si.shape) + cum[:, None, None]).flatten()[keep]
    return si, ri


for B, T, N in ((64, 49, 40), (1, 49, 40), (64, 49, 200)):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(B, T, N, 4, generator=g) * 20).cuda()
    m = (torch.rand(B, T, N, generator=g) < 0.6).float().cuda()
    for name, fn in (("aether_knn_edges", lambda: knn_edges(x, m)), ("torch cdist+topk", lambda: torch_formulation(x, m))):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("B=%d T=%d N=%d  %-18s %.3f ms  (%d edges, %.1f M edges/s)" % (B, T, N, name, dt * 1e3, out[0].numel(),
                                                                          out[0].numel() / dt / 1e6))
    a, b = knn_edges(x, m), torch_formulation(x, m)
    same = a[0].numel() == b[0].numel() and bool(torch.equal(a[0], b[0]))
    diff = int((a[1] != b[1]).sum()) if same else -1
    print("   neighbour entries that differ from the torch formulation: %d of %d (above 25 objects torch.cdist uses a "
          "matmul expansion whose rounding reorders near-equal distances)" % (diff, a[1].numel()))
