"""kNN edge builder: the HIP path vs the reference's own outputs (bit-exact indices) and vs the
oracle on fresh scenes."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from aether_amd import _lib
from aether_amd.knn import csr_by_receiver, get_knn_graph_info, knn_edges
from oracle import knn_oracle as K

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["small", "scenes", "few", "wide", "flat"])
def test_knn_edges_match_reference(name):
    d = np.load(os.path.join(GOLDEN, "knn_edges.npz"))
    x, m = torch.from_numpy(d[name + ".x"]).cuda(), torch.from_numpy(d[name + ".masks"]).cuda()
    send, recv, num = knn_edges(x, m, k=int(d[name + ".k"]))
    assert send.dtype == torch.int64 and recv.dtype == torch.int64
    assert np.array_equal(send.cpu().numpy(), d[name + ".send"])
    assert np.array_equal(recv.cpu().numpy(), d[name + ".recv"])
    assert np.array_equal(num.cpu().numpy(), d[name + ".num"])


def test_knn_graph_info_matches_reference():
    d = np.load(os.path.join(GOLDEN, "knn_edges.npz"))
    x, m = torch.from_numpy(d["info.x"]).cuda(), torch.from_numpy(d["info.masks"]).cuda()
    send, recv = get_knn_graph_info(x, m, int(d["info.masks"].sum()))
    assert np.array_equal(send.cpu().numpy(), d["info.send"]) and np.array_equal(recv.cpu().numpy(), d["info.recv"])
    order, rowptr = csr_by_receiver(recv, int(d["info.masks"].sum()))
    r = recv[order].cpu().numpy()
    assert (np.diff(r) >= 0).all() and int(rowptr[-1]) == recv.numel()


@pytest.mark.parametrize("S,N,k,p", [(64, 40, 10, 0.6), (3, 300, 10, 0.9), (500, 7, 10, 0.5), (2, 17, 16, 1.0),
                                     (5, 12, 1, 0.8)])
def test_knn_edges_vs_oracle(S, N, k, p):
    """cfg4-like batches (64 scenes of up to 40 agents), a large scene (several passes per thread), many tiny
    scenes, the maximum k, k = 1; lattice positions with many exactly equal distances (index order decides)."""
    g = torch.Generator().manual_seed(S * 1000 + N)
    x = torch.randn(S, N, 4, generator=g) * 20.0
    m = (torch.rand(S, N, generator=g) < p).float()
    for xx in (x, torch.round(x / 8.0)):                      # random, then a coarse integer lattice (ties)
        send, recv, num = knn_edges(xx.cuda(), m.cuda(), k=k)
        ws, wr, wn = K.knn_edges(xx.numpy(), m.numpy(), k)
        assert np.array_equal(send.cpu().numpy(), ws) and np.array_equal(recv.cpu().numpy(), wr)
        assert int(num) == int(wn)
    # structure: sources ascending, k (or fewer) neighbours each, no self edges, scenes do not mix
    s, r = send.cpu().numpy(), recv.cpu().numpy()
    assert (np.diff(s) >= 0).all() and (s != r).all()
    assert np.bincount(s).max() <= min(k, N - 1)


def test_knn_edges_empty_and_errors():
    x = torch.randn(2, 3, 5, 4, device="cuda")
    send, recv, num = knn_edges(x, torch.zeros(2, 3, 5, device="cuda"))
    assert send.numel() == 0 and recv.numel() == 0 and num.tolist() == [0, 0]
    one = torch.zeros(2, 3, 5, device="cuda"); one[..., 2] = 1.0      # single objects: no edges
    assert knn_edges(x, one)[0].numel() == 0
    with pytest.raises(_lib.AetherHipError):
        knn_edges(x.cpu(), one.cpu())
    with pytest.raises(ValueError):
        knn_edges(x, one[..., :4])
    with pytest.raises(ValueError):
        knn_edges(torch.randn(2, 30, 4, device="cuda"), torch.ones(2, 30, device="cuda"), k=17)
    assert get_knn_graph_info(x[0, 0], one[0, 0], 1) == (None, None)


def test_knn_full_size_properties():
    """cfg4's full size (64 scenes x 49 time steps x 40 agent slots, k = 10), checked through properties that do
    not need the oracle: every present object lists min(k, present - 1) distinct present neighbours of its own
    scene, never itself, in non-decreasing distance; the lists are contiguous and ordered by object; a second
    call returns the same edges; masking an object removes exactly its edges."""
    g = torch.Generator().manual_seed(99)
    B, T, N, k = 64, 49, 40, 10
    x = (torch.randn(B, T, N, 4, generator=g) * 20).cuda()
    m = (torch.rand(B, T, N, generator=g) < 0.6).float().cuda()
    send, recv, num = knn_edges(x, m, k=k)
    present = m.reshape(-1, N).sum(-1).long()                                  # per scene
    want_edges = (present * torch.clamp(present - 1, max=k)).sum()
    assert send.numel() == int(want_edges) and int(num.sum()) == send.numel() and num.shape == (B,)
    # compacted numbering -> (scene, object)
    flat_mask = m.reshape(-1).bool()
    obj_of = torch.arange(B * T * N, device="cuda")[flat_mask]                  # compacted id -> global slot
    gs, gr = obj_of[send], obj_of[recv]
    assert torch.equal(gs // N, gr // N) and (gs != gr).all()                  # same scene, no self edges
    assert (send[1:] >= send[:-1]).all()                                       # grouped by querying object, in order
    pos = x.reshape(-1, 4)[:, :2]
    d = (pos[gs] - pos[gr]).norm(dim=-1)
    same = send[1:] == send[:-1]
    assert (d[1:][same] >= d[:-1][same]).all()                                 # nearest first
    deg = torch.bincount(send, minlength=int(flat_mask.sum()))
    scene_of = (obj_of // N)
    assert torch.equal(deg, torch.clamp(present[scene_of] - 1, max=k))
    pair = send * (B * T * N) + recv
    assert pair.unique().numel() == pair.numel()                               # distinct neighbours
    s2, r2, _ = knn_edges(x, m, k=k)
    assert torch.equal(send, s2) and torch.equal(recv, r2)                     # deterministic
    m2 = m.clone()
    victim = int(obj_of[0])
    m2.view(-1)[victim] = 0
    s3, r3, _ = knn_edges(x, m2, k=k)
    assert not ((obj_of[1:][s3] == victim).any() or (obj_of[1:][r3] == victim).any())


def test_knn_tiny_scenes():
    """One object slot (k = min(10, 0) = 0: no edges), two slots, and k larger than the number of present objects."""
    x1 = torch.randn(3, 4, 1, 2, device="cuda")
    s, r, n = knn_edges(x1, torch.ones(3, 4, 1, device="cuda"))
    assert s.numel() == 0 and r.numel() == 0 and n.tolist() == [0, 0, 0]
    x2 = torch.randn(2, 3, 2, 2, device="cuda")
    m2 = torch.ones(2, 3, 2, device="cuda")
    m2[1, 2, 0] = 0
    s, r, n = knn_edges(x2, m2)
    ws, wr, wn = K.knn_edges(x2.cpu().numpy(), m2.cpu().numpy(), 10)
    assert np.array_equal(s.cpu().numpy(), ws) and np.array_equal(r.cpu().numpy(), wr) and n.tolist() == wn.tolist()
    x3 = torch.randn(1, 1, 12, 3, device="cuda")
    m3 = torch.zeros(1, 1, 12, device="cuda")
    m3[..., [1, 4, 9]] = 1                                             # 3 present of 12 slots, k = 10
    s, r, n = knn_edges(x3, m3)
    ws, wr, wn = K.knn_edges(x3.cpu().numpy(), m3.cpu().numpy(), 10)
    assert s.numel() == 6 and np.array_equal(s.cpu().numpy(), ws) and np.array_equal(r.cpu().numpy(), wr)
