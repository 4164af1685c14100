"""CPU restatement of the reference's k-nearest-neighbour edge builder for variable-N scenes
(first half).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under aether_amd/).

* ``knn_edges``          <- nn/dynamicvars/aether_dynamicvars.py:559-586 (``Encoder.knn_edges``; the same method is
  repeated in locs/glocs/dnri/aether_origin ``*_dynamicvars.py``): per scene (time step) the k = min(10, N - 1)
  nearest present objects of every present object by 2-D distance (``x[..., :2]``), absent objects
  (mask 0) and the object itself excluded; edges listed scene by scene, object by object, nearest first, in
  the compacted numbering (present objects of all scenes counted consecutively).  The reference names the
  querying object ``send`` and its neighbour ``recv``.
* ``knn_graph_info``     <- experiments/ind/single_ind_data.py:186-217 (``get_knn_graph_info`` with
  ``use_edge2node=False``): the same for one scene, numbering local to the scene.  (Its ``edge2node_inds``
  ``.view(-1, k)`` assumes every object is a neighbour of exactly k others, which kNN graphs do not satisfy --
  SURVEY.md 8f N2 -- so it is not restated; the library returns a CSR by receiver instead.)

Written as plain loops over numpy arrays -- no cdist / topk -- so that it is an independent statement:
squared distances in fp32 (dx*dx + dy*dy, then sqrt, as a direct evaluation gives them), a stable sort by
(distance, index).  torch.cdist switches to a matmul expansion above 25 objects and torch.topk does not
specify the order of exactly equal distances: parity on indices is defined for scenes whose neighbour
distances are separated by more than the rounding of either evaluation (the fixtures check that they are).

Parity status: PINNED by tests/golden/knn_edges.npz (the imported reference ``Encoder.knn_edges`` and
``get_knn_graph_info``, oracle/make_golden_knn.py).
"""
from __future__ import annotations

import numpy as np


def _scene(xy, mask, k):
    """Neighbour lists of one scene: [(i, [j nearest first])] over present objects, local indices."""
    n = xy.shape[0]
    k = min(k, n - 1)
    out = []
    for i in range(n):
        if mask[i] == 0:
            continue
        cand = []
        for j in range(n):
            if j == i or mask[j] == 0:
                continue
            dx, dy = np.float32(xy[i, 0] - xy[j, 0]), np.float32(xy[i, 1] - xy[j, 1])
            cand.append((float(np.sqrt(np.float32(dx * dx + dy * dy))), j))
        cand.sort()
        out.append((i, [j for _, j in cand[:k]]))
    return out


def knn_edges(x, masks, k=10):
    """x [..., N, D >= 2], masks [..., N] (0/1) -> (send int64[E], recv int64[E], num_edges per leading-most
    batch entry, as the reference sums it: over the last three axes of [..., T, N, k])."""
    x = np.asarray(x, dtype=np.float32)
    masks = np.asarray(masks, dtype=np.float32)
    lead = x.shape[:-2]
    xs = x.reshape(-1, x.shape[-2], x.shape[-1])
    ms = masks.reshape(-1, masks.shape[-1])
    send, recv, per_scene = [], [], []
    base = 0
    for s in range(xs.shape[0]):
        sparse = np.cumsum(ms[s]).astype(np.int64) - 1
        cnt = 0
        for i, nbrs in _scene(xs[s, :, :2], ms[s], k):
            for j in nbrs:
                send.append(base + sparse[i])
                recv.append(base + sparse[j])
                cnt += 1
        per_scene.append(cnt)
        base += int(ms[s].sum())
    per_scene = np.asarray(per_scene, dtype=np.int64).reshape(lead)
    num = per_scene.sum(axis=-1) if per_scene.ndim >= 1 else per_scene       # aether_dynamicvars.py:585
    return np.asarray(send, dtype=np.int64), np.asarray(recv, dtype=np.int64), num


def knn_graph_info(inputs, masks, k=10):
    """One scene: inputs [N, D >= 2], masks [N] -> (send, recv) in the scene's compacted numbering."""
    s, r, _ = knn_edges(np.asarray(inputs)[None], np.asarray(masks)[None], k)
    return s, r
