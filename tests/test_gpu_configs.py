"""Parity on the other BASELINE.json configs: 20-step rollout, variable-N sparse kNN scenes (cfg4),
one large dense graph (cfg5 shape at reduced size), and size-independent properties."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_state_dict, scale_rel_err
from aether_amd import _lib
from aether_amd.edges import get_edges, prepare_edge_attr
from aether_amd.nn.state2state.aether import Aether
from aether_amd.rollout import rollout, rollout_mse, rollout_stepwise
from aether_amd.synthetic import make_batch
from oracle import aether_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _model(D, flags=0):
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    m.load_state_dict(load_state_dict(D))
    m.flags = flags
    return m


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_rollout_20_steps_matches_reference(D, flags):
    """Trajectory produced by driving the imported reference module through the same loop."""
    d = np.load(os.path.join(GOLDEN, f"rollout_D{D}_B4N5.npz"))
    x, vel, q = (torch.from_numpy(d[k]).cuda() for k in ("in.x", "in.vel", "in.charges"))
    edges = get_edges(4, 5, device="cuda")
    traj = rollout(_model(D, flags), x, vel, edges, q, 20).cpu()
    ref = torch.from_numpy(d["ref.traj"])
    assert scale_rel_err(traj, ref) <= TOL
    # metric 2: per-step MSE against an arbitrary "truth" agrees to 1e-5 relative at every step
    g = torch.Generator().manual_seed(0)
    truth = ref + 0.1 * torch.randn(ref.shape, generator=g)
    a, b = rollout_mse(traj, truth), rollout_mse(ref, truth)
    assert float(((a - b).abs() / b).max()) <= 1e-5


def _cut_exposed_graphs(margins, N, width):
    """Graphs in which some edge comes within `width` radians of a branch cut of the reference's feature map
    (oracle cut_margin) at some step: from that step on, two evaluations that agree to ~width may legitimately
    sit on different sides of the discontinuity."""
    recv = get_edges(margins.shape[1] // (N * (N - 1)), N)[1]
    first = {}
    for t in range(margins.shape[0]):
        for e in torch.nonzero(margins[t] < width).flatten().tolist():
            first.setdefault(int(recv[e]) // N, t)
    return first


@pytest.mark.parametrize("trained", [False, True])
@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
def test_headline_rollout_20_steps_vs_fp64_oracle(flags, trained):
    """BASELINE.json's headline shape (2-D, N=20, B=128), 20 steps, against the oracle in fp64 -- with the
    seed-1 weights and with the weights bench.py holds when it checks parity (60 AdamW steps on the batch).
    Bound: 1e-5 scale-relative on every node, except in graphs where the fp64 trajectory itself passes within
    2e-5 rad of a branch cut of the feature map (anti-parallel headings: relative orientation +1 <-> -1,
    geometry.py:87-100; sender exactly behind the receiver: bearing +pi <-> -pi, aether.py:72-75): there the
    step is discontinuous and which side an fp32 evaluation takes is decided by its last bit.  DESIGN.md 5.1:
    that is what the 7.7e-5 of BENCH_r01 was -- the ORACLE's fp32 trajectory crossing such a cut at step 15
    (graph 50, edge 1004 -> 1006), not the HIP path, which stays within 2.1e-6 of fp64.  The fp32 oracle is
    held to the same rule here, so the exemption is shown to be a property of the map, not of the kernels."""
    B, N, D, T = 128, 20, 2, 20
    torch.manual_seed(1)
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    m.flags = flags
    host = make_batch(B, N, D, seed=0)
    inp = {k: v.cuda() for k, v in host.items() if torch.is_tensor(v)}
    edges_d = [e.cuda() for e in host["edges"]]
    if trained:
        opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=1e-12)
        for _ in range(60):
            opt.zero_grad(set_to_none=True)
            o = m(inp["h"], inp["x"], edges_d, inp["vel"], inp["edge_attr"], inp["charges"])
            torch.nn.functional.mse_loss(o, inp["target"]).backward()
            opt.step()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        t64, margins = O.rollout(sd64, host["x"].double(), host["vel"].double(), host["edges"],
                                 host["charges"].double(), T, with_margin=True)
        t32 = O.rollout(sd, host["x"], host["vel"], host["edges"], host["charges"], T)
        got = rollout(m, inp["x"], inp["vel"], edges_d, inp["charges"], T).cpu()
    # width: the fp32 paths drift up to ~3e-5 (positions) / ~5e-6 rad (headings) from fp64 over 20 steps
    exposed = _cut_exposed_graphs(margins, N, 2e-5)
    assert len(exposed) <= B // 4, exposed              # ~1e6 edge-steps x 2 cuts x 2e-5/pi: a dozen or two of the 128 graphs
    scale = float(t64.abs().max())
    for name, tr in (("hip", got), ("oracle_fp32", t32)):
        err = (tr.double() - t64).abs().amax(dim=2)      # [T, Nn]
        bad = torch.nonzero(err > TOL * scale)
        for t, nd in bad.tolist():
            g = nd // N
            assert g in exposed and t >= exposed[g], (name, t, nd, float(err[t, nd]) / scale, exposed)
        clean = torch.ones(B * N, dtype=torch.bool)
        for g in exposed:
            clean[g * N:(g + 1) * N] = False
        assert float(err[:, clean].max()) <= TOL * scale, (name, float(err[:, clean].max()) / scale)
    if not exposed:
        assert scale_rel_err(got, t64.float()) <= TOL


@pytest.mark.parametrize("flags", [0, _lib.FLAG_FORCE_STREAMED])
@pytest.mark.parametrize("D", [2, 3])
def test_device_rollout_equals_loop_of_module_calls(D, flags):
    """aether_rollout (edge attributes and velocities derived in the kernels) vs the loop a user of
    the reference would write around forward(); also at a non-unit dt and at the headline shape."""
    m = _model(D, flags)
    for (B, N, T, dt, seed) in [(4, 5, 20, 1.0, 3), (128, 20, 6, 0.25, 4), (3, 9, 5, 2.0, 5)]:
        inp = make_batch(B, N, D, seed=seed, device="cuda")
        a = rollout(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], T, dt)
        b = rollout_stepwise(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], T, dt)
        assert a.shape == (T, B * N, D)
        assert scale_rel_err(a[0].cpu(), b[0].cpu()) <= 1e-6          # one step: same arithmetic up to fma contraction
        assert scale_rel_err(a.cpu(), b.cpu()) <= TOL
    assert rollout(m, inp["x"], inp["vel"], inp["edges"], inp["charges"], 0).shape == (0, 27, D)


def _knn_scene_batch(n_scenes, D, k, seed):
    """cfg4-like: scenes with N ~ U{2..40} agents, k nearest neighbours as in-edges
    (experiments/ind/single_ind_data.py:186-217), concatenated with node offsets."""
    g = torch.Generator().manual_seed(seed)
    xs, vs, qs, send, recv, off = [], [], [], [], [], 0
    for _ in range(n_scenes):
        n = int(torch.randint(2, 41, (1,), generator=g))
        x = torch.randn(n, D, generator=g) * 3.0
        v = torch.randn(n, D, generator=g)
        v = 0.5 * v / v.norm(dim=-1, keepdim=True)
        q = torch.randint(0, 2, (n, 1), generator=g).float() * 2 - 1
        dist = torch.cdist(x, x) + torch.eye(n) * 1e9
        kk = min(k, n - 1)
        nbr = dist.topk(kk, largest=False).indices            # [n, kk] senders of each receiver
        r = torch.arange(n).repeat_interleave(kk)
        s = nbr.reshape(-1)
        send.append(s + off); recv.append(r + off)
        xs.append(x); vs.append(v); qs.append(q); off += n
    x, v, q = torch.cat(xs), torch.cat(vs), torch.cat(qs)
    edges = [torch.cat(send).long(), torch.cat(recv).long()]
    perm = torch.randperm(edges[0].numel(), generator=g)        # unsorted edge list
    edges = [edges[0][perm], edges[1][perm]]
    ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
    return dict(x=x, vel=v, charges=q, edges=edges, edge_attr=ea, h=v.norm(dim=-1, keepdim=True))


@pytest.mark.parametrize("D", [2, 3])
def test_variable_n_knn_scenes(D):
    sd = load_state_dict(D)
    inp = _knn_scene_batch(64, D, 10, seed=5)
    want = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    m = _model(D)
    dev = "cuda"
    edges = [e.to(dev) for e in inp["edges"]]
    with torch.no_grad():
        out = m(inp["h"].to(dev), inp["x"].to(dev), edges, inp["vel"].to(dev), inp["edge_attr"].to(dev),
                inp["charges"].to(dev))
    assert scale_rel_err(out.cpu(), want) <= TOL
    info = m.prepare_graph(edges, inp["x"].shape[0])[1]
    # scenes with > 32 agents do not fit a fused group -> the whole batch takes the streamed path
    assert info.n_groups == 0 or info.max_group_nodes <= 32


def test_cfg4_device_built_knn_graph():
    """cfg4 end to end on the device: 64 padded scenes with presence masks -> aether_knn_edges (compacted
    numbering) -> Aether.forward on the compacted agents, vs the oracle on the oracle's own kNN graph."""
    from aether_amd.knn import knn_edges
    from oracle import knn_oracle as K
    D, S, N = 2, 64, 40
    sd = load_state_dict(D)
    g = torch.Generator().manual_seed(21)
    x = torch.randn(S, N, D, generator=g) * 3.0
    v = torch.randn(S, N, D, generator=g)
    v = 0.5 * v / v.norm(dim=-1, keepdim=True)
    q = torch.randint(0, 2, (S, N, 1), generator=g).float() * 2 - 1
    masks = (torch.rand(S, N, generator=g) < 0.6).float()
    keep = masks.bool().reshape(-1)
    xc, vc, qc = x.reshape(-1, D)[keep], v.reshape(-1, D)[keep], q.reshape(-1, 1)[keep]      # compacted agents
    ws, wr, _ = K.knn_edges(x.numpy(), masks.numpy(), 10)
    # the reference lists (query, neighbour); messages flow neighbour -> query
    o_edges = [torch.from_numpy(wr), torch.from_numpy(ws)]
    want = O.aether_forward(sd, xc, vc, o_edges, prepare_edge_attr(xc, o_edges, qc[o_edges[0]] * qc[o_edges[1]]), qc)
    dev = "cuda"
    send, recv, num = knn_edges(x.to(dev), masks.to(dev), k=10)
    assert torch.equal(send.cpu(), torch.from_numpy(ws)) and torch.equal(recv.cpu(), torch.from_numpy(wr))
    edges = [recv, send]
    xd, vd, qd = xc.to(dev), vc.to(dev), qc.to(dev)
    ea = prepare_edge_attr(xd, edges, qd[edges[0]] * qd[edges[1]])
    m = _model(D)
    with torch.no_grad():
        out = m(vd.norm(dim=-1, keepdim=True), xd, edges, vd, ea, qd)
    assert int(num) == send.numel() and scale_rel_err(out.cpu(), want) <= TOL


def test_large_dense_graph_cfg5_shape():
    """One fully connected graph, N=512 (261,632 edges): cfg5's shape at a size the oracle finishes."""
    D = 2
    sd = load_state_dict(D)
    inp = make_batch(1, 512, D, seed=9)
    want = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    m = _model(D)
    dev = "cuda"
    with torch.no_grad():
        out = m(inp["h"].to(dev), inp["x"].to(dev), [e.to(dev) for e in inp["edges"]], inp["vel"].to(dev),
                inp["edge_attr"].to(dev), inp["charges"].to(dev))
    assert scale_rel_err(out.cpu(), want) <= TOL


@pytest.mark.parametrize("D", [2, 3])
def test_equivariance_with_field_off(D):
    """f(Qx + t, Qv) = Q f(x, v) + t when the field net's last layer is zero;
    in 3-D only for rotations about z (the frames use yaw/pitch of the velocity only)."""
    m = _model(D)
    with torch.no_grad():
        m.field_net.net[4].weight.zero_()
        m.field_net.net[4].bias.zero_()
    inp = make_batch(6, 9, D, seed=17)
    th = 0.7
    c, s = math.cos(th), math.sin(th)
    Q = torch.tensor([[c, -s], [s, c]]) if D == 2 else torch.tensor([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]])
    t = torch.tensor([0.3, -1.1, 0.4][:D])
    dev = "cuda"
    edges = [e.to(dev) for e in inp["edges"]]

    def run(x, v):
        ea = prepare_edge_attr(x, inp["edges"], inp["charges"][inp["edges"][0]] * inp["charges"][inp["edges"][1]])
        with torch.no_grad():
            return m(None, x.to(dev), edges, v.to(dev), ea.to(dev), inp["charges"].to(dev)).cpu()

    a = run(inp["x"] @ Q.T + t, inp["vel"] @ Q.T)
    b = run(inp["x"], inp["vel"]) @ Q.T + t
    assert scale_rel_err(a, b) <= 2e-5


def test_full_size_multi_rank_shards_are_independent():
    """Rows a rank computes for its block of graphs equal the rows of the full batch (no exchange)."""
    from aether_amd.parallel import shard_graphs
    D, B, N = 2, 128, 20
    m = _model(D)
    inp = make_batch(B, N, D, seed=23, device="cuda")
    with torch.no_grad():
        full = m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        for rank in (0, 5):
            lo, hi = shard_graphs(B, rank, 8)
            sl = slice(lo * N, hi * N)
            e = get_edges(hi - lo, N, device="cuda")
            q = inp["charges"][sl]
            ea = prepare_edge_attr(inp["x"][sl], e, q[e[0]] * q[e[1]])
            part = m(None, inp["x"][sl], e, inp["vel"][sl], ea, q)
            assert scale_rel_err(part, full[sl]) <= 2e-6


def test_workspace_reuse_keeps_split_mode_handoff_armed():
    """Inference reuses one workspace; from the second call on the module passes
    AETHER_FLAG_WORKSPACE_REUSED and the library no longer zeroes the hand-off words of the split
    (two workgroups per graph) kernel.  Outputs must stay bit-identical call after call, also when
    another shape or the streamed path used the buffer in between."""
    D = 2
    m = Aether(2 * D, 64, 0.0, D, device="cuda")
    m.load_state_dict(load_state_dict(D))
    a = make_batch(16, 20, D, seed=5, device="cuda")       # 16 graphs on 256 CUs: split mode
    b = make_batch(7, 9, D, seed=6, device="cuda")

    def run(inp, flags=0):
        m.flags = flags
        with torch.no_grad():
            return m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]).clone()

    ref_a, ref_b = run(a), run(b)
    want = O.aether_forward(load_state_dict(D), a["x"].cpu(), a["vel"].cpu(), [e.cpu() for e in a["edges"]],
                            a["edge_attr"].cpu(), a["charges"].cpu())
    assert scale_rel_err(ref_a.cpu(), want) <= 1e-5
    for _ in range(3):
        assert torch.equal(run(a), ref_a)                  # reused, flags left armed by the kernel
    assert m._ws_key is not None
    assert torch.equal(run(b), ref_b)                      # other layout in the same buffer
    assert torch.equal(run(a), ref_a)
    run(a, _lib.FLAG_FORCE_STREAMED)                       # streamed kernels in between
    assert torch.equal(run(a), ref_a)
    assert torch.equal(run(a), ref_a)


def _random_multigraph_batch(seed, D):
    """Arbitrary edge lists as the reference's scatter accepts them: components of 1..40 nodes,
    random edges with self loops and duplicates, some nodes without in-edges (the last node of the
    batch always receives one, so that scatter's output has n_nodes rows), unsorted."""
    g = torch.Generator().manual_seed(seed)
    n_comp = int(torch.randint(1, 12, (1,), generator=g))
    send, recv, off = [], [], 0
    for _ in range(n_comp):
        n = int(torch.randint(1, 41, (1,), generator=g))
        m = int(torch.randint(0, 6 * n + 1, (1,), generator=g))
        if m:
            send.append(torch.randint(0, n, (m,), generator=g) + off)
            recv.append(torch.randint(0, n, (m,), generator=g) + off)
        off += n
    send.append(torch.tensor([max(off - 2, 0)])); recv.append(torch.tensor([off - 1]))
    edges = [torch.cat(send).long(), torch.cat(recv).long()]
    perm = torch.randperm(edges[0].numel(), generator=g)
    edges = [edges[0][perm], edges[1][perm]]
    x = torch.randn(off, D, generator=g) * 2.0
    v = torch.randn(off, D, generator=g)
    q = torch.randint(-1, 2, (off, 1), generator=g).float()
    ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
    return dict(x=x, vel=v, charges=q, edges=edges, edge_attr=ea, h=v.norm(dim=-1, keepdim=True))


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("D", [2, 3])
def test_random_multigraphs_match_oracle(D, seed):
    sd = load_state_dict(D)
    inp = _random_multigraph_batch(seed, D)
    want = O.aether_forward(sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    assert torch.isfinite(want).all()
    dev = "cuda"
    edges = [e.to(dev) for e in inp["edges"]]
    outs = {}
    for name, flags in (("default", 0), ("streamed", _lib.FLAG_FORCE_STREAMED)):
        m = _model(D, flags)
        with torch.no_grad():
            outs[name] = m(inp["h"].to(dev), inp["x"].to(dev), edges, inp["vel"].to(dev),
                           inp["edge_attr"].to(dev), inp["charges"].to(dev)).cpu()
        assert scale_rel_err(outs[name], want) <= TOL, (name, seed)


def test_cfg5_full_shard_properties():
    """BASELINE config 5 at its full per-GPU size (32 fully connected graphs of 1,024 bodies: 32,768 nodes, 33.5 M
    edges, the streamed path), through properties that need no oracle: graphs are independent (graph 0 of the batch
    = the same graph run alone), a second run is bit-identical, relabelling the bodies of a graph permutes its
    outputs, everything is finite.  (Parity against the oracle at this shape: the N=512 test above.)"""
    D, B, N = 2, 32, 1024
    m = _model(D)
    g = torch.Generator(device="cuda").manual_seed(5)
    scale = (N / 5.0) ** (1.0 / 3.0)
    x = torch.randn(B * N, D, device="cuda", generator=g) * scale
    v = torch.randn(B * N, D, device="cuda", generator=g)
    v = 0.5 * v / v.norm(dim=-1, keepdim=True)
    q = (torch.randint(0, 2, (B * N, 1), device="cuda", generator=g).float() * 2 - 1)
    edges = get_edges(B, N, device="cuda")
    assert edges[0].numel() == B * N * (N - 1) == 33_521_664
    ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
    h = v.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        out = m(h, x, edges, v, ea, q)
        assert out.shape == (B * N, D) and torch.isfinite(out).all()
        out2 = m(h, x, edges, v, ea, q)
        assert torch.equal(out, out2)                                         # deterministic
        del out2
        # graph 0 alone
        e1 = get_edges(1, N, device="cuda")
        ea1 = prepare_edge_attr(x[:N], e1, q[:N][e1[0]] * q[:N][e1[1]])
        alone = m(h[:N], x[:N], e1, v[:N], ea1, q[:N])
        assert scale_rel_err(out[:N].cpu(), alone.cpu()) <= 1e-6               # no cross-graph term
        # relabel the bodies of that graph
        perm = torch.randperm(N, device="cuda", generator=g)
        xp, vp, qp = x[:N][perm], v[:N][perm], q[:N][perm]
        eap = prepare_edge_attr(xp, e1, qp[e1[0]] * qp[e1[1]])
        outp = m(vp.norm(dim=-1, keepdim=True), xp, e1, vp, eap, qp)
        assert scale_rel_err(outp.cpu(), alone[perm].cpu()) <= TOL             # sums over 1,023 in-edges reorder


def test_metric2_mse_vs_simulated_truth_is_a_parity_statement():
    """BASELINE.json's metric 2 -- 20-step rollout MSE against simulated ground truth, protocol of
    experiments/electrostatic/evaluate.py:33-70 -- at the headline shape (2-D, N = 20, B = 128), HIP and the fp32 oracle
    side by side against the fp64 oracle (oracle/metric2.py).

    * seed-1 (untrained) weights: the rollout is chaotic (MSE 0.05 -> 5); the reference's own fp32 arithmetic separates
      from fp64 by 5e-5 on the MSE and 5e-3 on the trajectory, so 1e-5 cannot be asserted of anything there -- HIP has
      to stay inside that envelope (factor 3, the two fp32 paths being two different roundings of the same chaos);
    * the same model after 200 captured training steps on one-frame targets of the burn-in frames: the MSE keeps 1e-5
      (north_star: "20-step rollout MSE within 1e-5 of reference") -- asserted for HIP, and shown for the fp32 oracle."""
    import contextlib
    import io
    from oracle import metric2 as M2
    dev = torch.device("cuda")
    B, N, D = 128, 20, 2
    data = M2.simulate(dev, B, N, D)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = Aether(2 * D, 64, 0.0, D, device=dev)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    r0 = M2.report(sd, m, data)
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    M2.train_on_frames(m, data, steps=200, lr=1e-3)
    sd_t = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    r1 = M2.report(sd_t, m, data)
    for name, r in (("seed-1", r0), ("trained", r1)):
        h, o = r["hip"], r["oracle_fp32"]
        print(f"\n[metric 2, {name}] MSE fp64 {r['mse_oracle_fp64_steps_1_10_20']}; rel. MSE difference HIP "
              f"{h['max_rel_mse_difference']:.2e} / oracle fp32 {o['max_rel_mse_difference']:.2e}; trajectory HIP "
              f"{h['trajectory_max_rel_err']:.2e} / oracle fp32 {o['trajectory_max_rel_err']:.2e}; first step above 1e-5: HIP "
              f"{h['first_step_above_tolerance']} / oracle fp32 {o['first_step_above_tolerance']}")
        assert all(np.isfinite(v) for v in h["mse_steps_1_10_20"])
        # inside the envelope of the reference's own fp32 arithmetic (or inside the tolerance outright)
        assert h["max_rel_mse_difference"] <= max(1e-5, 3.0 * o["max_rel_mse_difference"]), name
        assert h["trajectory_max_rel_err"] <= max(1e-5, 3.0 * o["trajectory_max_rel_err"]), name
        # one step is always inside the tolerance: the divergence is the rollout's, not the step's
        assert h["per_step_max_rel_err"][0] <= 1e-5 and o["per_step_max_rel_err"][0] <= 1e-5, name
    # the untrained rollout really is ill-conditioned for the reference's arithmetic too (else the bound above is vacuous)
    assert r0["oracle_fp32"]["max_rel_mse_difference"] > 1e-5
    # the well-conditioned case: metric 2 within 1e-5 of the fp64 reference
    assert r1["mse_oracle_fp64_steps_1_10_20"][0] < 0.2 * r0["mse_oracle_fp64_steps_1_10_20"][0]      # training helped
    assert r1["hip"]["max_rel_mse_difference"] <= 1e-5
    assert r1["oracle_fp32"]["max_rel_mse_difference"] <= 1e-5
