"""Variable-N decoder step (second half): the HIP path vs the reference's own outputs and vs the
oracle on fresh scenes with graphs built on the device."""
import pytest
import torch

from conftest import load_dyn_decoder, scale_rel_err
from aether_amd import _lib
from aether_amd.knn import get_knn_graph_info
from aether_amd.nn.dynamicvars.decoder import Decoder
from oracle import dynamicvars_oracle as DO

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize("name", ["full8", "tail6", "gaps", "knn20", "empty"])
def test_decoder_step_matches_reference(name):
    c, dec, params = load_dyn_decoder(name)
    dec = dec.cuda()
    gi = tuple(c[k].cuda() for k in ("send", "recv", "e2n")) if "send" in c else None
    pred, hid = dec(c["inputs"].cuda(), c["hidden"].cuda(), c["edges"].cuda(), c["masks"].unsqueeze(0).cuda(), gi,
                    c["field"].cuda())
    assert pred.shape == c["ref.pred"].shape and hid.shape == c["ref.hidden"].shape
    assert scale_rel_err(pred.cpu(), c["ref.pred"]) <= TOL and scale_rel_err(hid.cpu(), c["ref.hidden"]) <= TOL
    absent = c["masks"] == 0
    assert (pred.cpu()[0, absent] == 0).all() and torch.equal(hid.cpu()[0, absent], c["hidden"][0, absent])


@pytest.mark.parametrize("Nmax,p,K,skip,posrep,H", [(40, 0.8, 4, True, "cart", 256), (64, 0.5, 2, False, "polar", 128),
                                                     (3, 1.0, 1, False, "cart", 128)])
def test_decoder_step_vs_oracle_device_graph(Nmax, p, K, skip, posrep, H):
    """inD-like sizes (4 edge types, the first skipped, hidden 256), graphs from aether_knn_edges; one-hot edge
    types as the sampler produces them."""
    params = {"input_size": 4, "gpu": True, "decoder_hidden": H, "num_edge_types": K, "skip_first": skip,
              "decoder_dropout": 0.0, "pos_representation": posrep}
    torch.manual_seed(17)
    dec = Decoder(params, device="cuda").eval()
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(Nmax)
    inputs = torch.randn(1, Nmax, 4, generator=g)
    hidden = torch.randn(1, Nmax, H, generator=g) * 0.3
    field = torch.randn(1, Nmax, 2, generator=g) * 0.3
    masks = (torch.rand(Nmax, generator=g) < p).float()
    masks[:2] = 1.0
    nv = int(masks.sum())
    send, recv = get_knn_graph_info(inputs[0].cuda(), masks.cuda(), nv)
    k = min(10, nv - 1)
    # the reference's edge2node_inds: edge ids sorted by `recv`, k per row (single_ind_data.py:213-215)
    e2n = torch.argsort(recv, stable=True).view(-1, k)
    types = torch.randint(0, K, (send.numel(),), generator=g)
    edges = torch.nn.functional.one_hot(types, K).float().unsqueeze(0)
    want_p, want_h = DO.decoder_step(sd, inputs, hidden, edges, masks, (send.cpu(), recv.cpu(), e2n.cpu()), field, skip, posrep)
    got_p, got_h = dec(inputs.cuda(), hidden.cuda(), edges.cuda(), masks.cuda(), (send, recv, e2n), field.cuda())
    assert scale_rel_err(got_p.cpu(), want_p) <= TOL and scale_rel_err(got_h.cpu(), want_h) <= TOL


def test_decoder_step_64_scenes_batched_equals_single_scene_oracle():
    """BASELINE config 4 (inD-like scenes, batch = 64): ONE batched call for 64 scenes of 2..40 present objects against 64
    single-scene oracle steps (the reference itself refuses batch > 1, aether_dynamicvars.py:588-591).  Includes an empty
    scene, scenes with interior objects missing (the reference's compacted-index quirk, :823) and kNN graphs with k < 10."""
    import time
    B, Nmax, K, H = 64, 40, 4, 256
    params = {"input_size": 4, "gpu": True, "decoder_hidden": H, "num_edge_types": K, "skip_first": True,
              "decoder_dropout": 0.0, "pos_representation": "cart"}
    torch.manual_seed(23)
    dec = Decoder(params, device="cuda").eval()
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator().manual_seed(64)
    inputs = torch.randn(B, Nmax, 4, generator=g)
    hidden = torch.randn(B, Nmax, H, generator=g) * 0.3
    field = torch.randn(B, Nmax, 2, generator=g) * 0.3
    masks = torch.zeros(B, Nmax)
    for b in range(B):
        nv = int(torch.randint(2, Nmax + 1, (1,), generator=g))
        masks[b, torch.randperm(Nmax, generator=g)[:nv]] = 1.0           # interior objects missing in most scenes
    masks[5] = 0.0                                                        # an empty scene
    masks[9] = 0.0
    masks[9, :3] = 1.0                                                    # three objects: k = 2
    edges_l, gi_l, want_p, want_h = [], [], [], []
    for b in range(B):
        nv = int(masks[b].sum())
        if nv == 0:
            edges_l.append(torch.zeros(0, K).cuda()); gi_l.append(None)
            want_p.append(torch.zeros(1, Nmax, 4)); want_h.append(hidden[b:b + 1])
            continue
        send, recv = get_knn_graph_info(inputs[b].cuda(), masks[b].cuda(), nv)
        k = min(10, nv - 1)
        e2n = torch.argsort(recv, stable=True).view(-1, k)
        types = torch.randint(0, K, (send.numel(),), generator=g)
        e_b = torch.nn.functional.one_hot(types, K).float()
        p_b, h_b = DO.decoder_step(sd, inputs[b:b + 1], hidden[b:b + 1], e_b.unsqueeze(0), masks[b],
                                   (send.cpu(), recv.cpu(), e2n.cpu()), field[b:b + 1], True, "cart")
        edges_l.append(e_b.cuda()); gi_l.append((send, recv, e2n))
        want_p.append(p_b); want_h.append(h_b)
    want_p, want_h = torch.cat(want_p), torch.cat(want_h)
    args = (inputs.cuda(), hidden.cuda(), edges_l, masks.cuda(), gi_l, field.cuda())
    got_p, got_h = dec(*args)                                             # B > 1 with per-scene lists -> forward_batched
    assert got_p.shape == (B, Nmax, 4) and got_h.shape == (B, Nmax, H)
    assert scale_rel_err(got_p.cpu(), want_p) <= TOL and scale_rel_err(got_h.cpu(), want_h) <= TOL
    assert (got_p.cpu()[masks == 0] == 0).all() and torch.equal(got_h.cpu()[masks == 0], hidden[masks == 0])
    # one batched call vs 64 single-scene calls of the same module (timing line of DESIGN 4.11 / profiles/)
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    def singles():
        for b in range(B):
            if gi_l[b] is not None:
                dec(args[0][b:b + 1], args[1][b:b + 1], edges_l[b].unsqueeze(0), args[3][b:b + 1], gi_l[b], args[5][b:b + 1])
    t_b, t_s = timed(lambda: dec(*args)), timed(singles, 2)
    print(f"\n[cfg4] decoder step, 64 scenes ({int(masks.sum())} objects): batched {t_b:.2f} ms, 64 single-scene calls {t_s:.2f} ms")


def test_decoder_errors():
    params = {"input_size": 4, "gpu": True, "decoder_hidden": 128, "num_edge_types": 2, "skip_first": False,
              "decoder_dropout": 0.0, "pos_representation": "cart"}
    dec = Decoder(params, device="cuda")
    x, h, f = torch.randn(1, 4, 4), torch.zeros(1, 4, 128), torch.zeros(1, 4, 2)
    with pytest.raises(_lib.AetherHipError):
        dec(x, h, None, torch.ones(4), None, f)                                       # CPU tensors: no fallback
    one = torch.tensor([0., 1., 0., 0.]).cuda()
    with pytest.raises(_lib.AetherHipError):
        dec(x.cuda(), h.cuda(), None, one, None, f.cuda())                            # single object: as the reference
    with pytest.raises(ValueError):
        dec(torch.randn(2, 4, 4).cuda(), h.cuda(), None, one, None, f.cuda())
    with pytest.raises(ValueError):
        Decoder(dict(params, decoder_hidden=96), device="cuda")


# ---------------------------------------------------------------- encoder prior step, field query, predict_future
def test_model_prediction_path_matches_reference():
    from conftest import load_dyn_model
    d, c, model, params = load_dyn_model()
    model = model.cuda()
    t = lambda k: torch.from_numpy(d[k])
    x0, m0 = c["inputs"][:, 0].cuda(), c["masks"][:, 0].cuda()
    field0, coords = model.predict_field(x0, m0)
    assert scale_rel_err(field0.cpu(), t("ref.field0")) <= TOL
    assert coords.shape == (int(c["masks"][0, 0].sum()), 4)
    gi0 = tuple(g.cuda() for g in c["graph_info"][0])
    logits0, (h1, c1) = model.encoder.single_step_forward(x0, m0, c["node_inds"][0].cuda(), gi0,
                                                          (t("state0.h").cuda(), t("state0.c").cuda()), t("ref.field0").cuda())
    assert scale_rel_err(logits0.cpu(), t("ref.logits0")) <= TOL
    assert scale_rel_err(h1.cpu(), t("ref.state1.h")) <= TOL and scale_rel_err(c1.cpu(), t("ref.state1.c")) <= TOL
    node_inds = [[n.cuda() for n in c["node_inds"]]]
    graph_info = [[tuple(g.cuda() for g in gi) for gi in c["graph_info"]]]
    preds = model.predict_future(c["inputs"].cuda(), c["masks"].cuda(), node_inds, graph_info, c["burn"].cuda(),
                                 uniform=[u.cuda() for u in c["uniform"]])
    assert preds.shape == t("ref.predictions").shape
    assert scale_rel_err(preds.cpu(), t("ref.predictions")) <= TOL
    free = model.predict_future(c["inputs"].cuda(), c["masks"].cuda(), node_inds, graph_info, c["burn"].cuda())
    assert torch.isfinite(free).all()


def test_model_vs_oracle_ind_sizes():
    """inD-like sizes: hidden 256, 4 edge types (first skipped), 30 object slots, kNN graphs with k = 10 built on the
    device; BatchNorm with perturbed statistics; polar edge positions."""
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    params = dict(MODEL_PARAMS, decoder_hidden=256, encoder_hidden=256, num_edge_types=4, pos_representation="polar",
                  field_hidden=128, encoder_rnn_hidden=64)
    torch.manual_seed(23)
    model = AetherDynamicVars(params, device=None).eval()
    perturb_bn_(model)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    g = torch.Generator().manual_seed(24)
    T, N = 4, 30
    inputs = torch.randn(1, T, N, 4, generator=g)
    masks = (torch.rand(1, T, N, generator=g) < 0.7).float()
    masks[:, :, :3] = 1
    burn = torch.ones(1, T, N)
    burn[:, 2:] = 0
    node_inds, graph_info, U = [], [], []
    for step in range(T):
        nv = int(masks[0, step].sum())
        send, recv = get_knn_graph_info(inputs[0, step].cuda(), masks[0, step].cuda(), nv)
        e2n = torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))
        graph_info.append((send.cpu(), recv.cpu(), e2n.cpu()))
        node_inds.append(masks[0, step].nonzero()[:, -1])
        U.append(torch.rand(send.numel(), 4, generator=g))
    want = DO.predict_future(sd, inputs, masks, node_inds, graph_info, burn, U[:T - 1], 0.5, True, "polar")
    got = model.predict_future(inputs.cuda(), masks.cuda(), [[n.cuda() for n in node_inds]],
                               [[tuple(x.cuda() for x in gi) for gi in graph_info]], burn.cuda(),
                               uniform=[u.cuda() for u in U[:T - 1]])
    assert scale_rel_err(got.cpu(), want) <= 2 * TOL


def test_predict_future_batched_scenes_vs_oracle_per_scene():
    """B = 6 scenes through ``predict_future`` at once (one field query / kNN + prior step / sampling / decoder step per
    time step for all scenes) against the ORACLE run scene by scene -- the reference raises on batch > 1
    (aether_dynamicvars.py:588-591).  inD-like sizes; one scene goes empty at a step, one has two objects."""
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    params = dict(MODEL_PARAMS, decoder_hidden=256, encoder_hidden=256, num_edge_types=4, pos_representation="polar",
                  field_hidden=128, encoder_rnn_hidden=64)
    torch.manual_seed(29)
    model = AetherDynamicVars(params, device=None).eval()
    perturb_bn_(model)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    g = torch.Generator().manual_seed(30)
    B, T, N = 6, 4, 24
    inputs = torch.randn(B, T, N, 4, generator=g)
    masks = (torch.rand(B, T, N, generator=g) < 0.7).float()
    masks[:, :, :3] = 1
    masks[2, 1] = 0                                           # scene 2 is empty at step 1
    masks[4] = 0
    masks[4, :, :2] = 1                                       # scene 4: two objects throughout
    burn = torch.ones(B, T, N)
    burn[:, 2:] = 0
    node_inds, graph_info, U = [], [], [[None] * B for _ in range(T)]
    for b in range(B):
        ni_b, gi_b = [], []
        for step in range(T):
            nv = int(masks[b, step].sum())
            if nv >= 2:
                send, recv = get_knn_graph_info(inputs[b, step].cuda(), masks[b, step].cuda(), nv)
                e2n = torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))
                gi_b.append((send.cpu(), recv.cpu(), e2n.cpu()))
            else:
                z = torch.zeros(0, dtype=torch.int64)
                gi_b.append((z, z, torch.zeros(0, 1, dtype=torch.int64)))
            ni_b.append(masks[b, step].nonzero()[:, -1])
            U[step][b] = torch.rand(gi_b[-1][0].numel(), 4, generator=g)
        node_inds.append(ni_b); graph_info.append(gi_b)
    want = torch.cat([DO.predict_future(sd, inputs[b:b + 1], masks[b:b + 1], node_inds[b], graph_info[b], burn[b:b + 1],
                                        [U[t][b] for t in range(T - 1)], 0.5, True, "polar") for b in range(B)])
    cu = lambda t: t.cuda()
    got = model.predict_future(inputs.cuda(), masks.cuda(), [[cu(n) for n in ni_b] for ni_b in node_inds],
                               [[tuple(cu(x) for x in gi) for gi in gi_b] for gi_b in graph_info], burn.cuda(),
                               uniform=[[cu(u) for u in U[t]] for t in range(T - 1)])
    assert got.shape == want.shape
    assert scale_rel_err(got.cpu(), want) <= 2 * TOL
    # round 4: `got` came from ONE library call for the whole loop (aether_dyn_rollout_batched).  The staged path of round 2
    # (three library calls + torch glue per step) runs the same stage kernels on the same concatenated rows: identical bits;
    # scene by scene through the single-scene call the stages see other sizes (other k-splits of the filter GEMM): tolerance
    gi_c = [[tuple(cu(x) for x in gi) for gi in gi_b] for gi_b in graph_info]
    ni_c = [[cu(n) for n in ni_b] for ni_b in node_inds]
    u_c = [[cu(u) for u in U[t]] for t in range(T - 1)]
    model.one_call_step = False
    try:
        staged = model.predict_future(inputs.cuda(), masks.cuda(), ni_c, gi_c, burn.cuda(), uniform=u_c)
    finally:
        model.one_call_step = True
    assert torch.equal(got, staged)
    again = model.predict_future(inputs.cuda(), masks.cuda(), ni_c, gi_c, burn.cuda(), uniform=u_c)
    assert torch.equal(got, again)
    for b in (0, 4):
        one = model.predict_future(inputs[b:b + 1].cuda(), masks[b:b + 1].cuda(), [ni_c[b]], [gi_c[b]], burn[b:b + 1].cuda(),
                                   uniform=[u_c[t][b] for t in range(T - 1)])
        assert scale_rel_err(got[b:b + 1].cpu(), one.cpu()) <= 2 * TOL, b
    # a scene with a single present object is refused before anything is queued, as the single-scene call refuses it
    bad_ni = [list(n) for n in ni_c]
    bad_ni[1][0] = ni_c[1][0][:1]
    with pytest.raises(_lib.AetherHipError, match="one present object"):
        model.predict_future(inputs.cuda(), masks.cuda(), bad_ni, gi_c, burn.cuda(), uniform=u_c)


def test_predict_future_64_scenes_one_call():
    """BASELINE config 4's batch: 64 inD-sized scenes (up to 40 objects, kNN k = 10) through ONE library call for the loop;
    equal to the staged batched path bit for bit, every scene within tolerance of its own single-scene call."""
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    params = dict(MODEL_PARAMS, decoder_hidden=256, encoder_hidden=256, num_edge_types=4, pos_representation="cart",
                  field_hidden=128, encoder_rnn_hidden=64)
    torch.manual_seed(41)
    model = AetherDynamicVars(params, device=None).eval()
    perturb_bn_(model)
    model = model.cuda()
    g = torch.Generator().manual_seed(42)
    B, T, N = 64, 4, 40
    inputs = torch.randn(B, T, N, 4, generator=g).cuda()
    masks = torch.zeros(B, T, N)
    for b in range(B):
        c = int(torch.randint(2, N + 1, (1,), generator=g))
        masks[b, :, torch.randperm(N, generator=g)[:c]] = 1
    masks = masks.cuda()
    burn = torch.ones(B, T, N).cuda()
    burn[:, 2:] = 0
    node_inds, graph_info, U = [], [], [[None] * B for _ in range(T - 1)]
    for b in range(B):
        ni_b, gi_b = [], []
        for step in range(T):
            nv = int(masks[b, step].sum())
            send, recv = get_knn_graph_info(inputs[b, step], masks[b, step], nv)
            gi_b.append((send, recv, torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))))
            ni_b.append(masks[b, step].nonzero()[:, -1])
            if step < T - 1:
                U[step][b] = torch.rand(send.numel(), 4, generator=g).cuda()
        node_inds.append(ni_b); graph_info.append(gi_b)
    got = model.predict_future(inputs, masks, node_inds, graph_info, burn, uniform=U)
    assert got.shape == (B, T - 1, N, 4) and torch.isfinite(got).all()
    model.one_call_step = False
    try:
        staged = model.predict_future(inputs, masks, node_inds, graph_info, burn, uniform=U)
    finally:
        model.one_call_step = True
    assert torch.equal(got, staged)
    for b in (0, 17, 63):
        one = model.predict_future(inputs[b:b + 1], masks[b:b + 1], [node_inds[b]], [graph_info[b]], burn[b:b + 1],
                                   uniform=[U[t][b] for t in range(T - 1)])
        assert scale_rel_err(got[b:b + 1].cpu(), one.cpu()) <= 2 * TOL, b


def test_decoder_and_encoder_two_objects():
    """The smallest scene the reference accepts: two present objects (one edge each way)."""
    from aether_amd.nn.dynamicvars.encoder import Encoder
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS
    params = dict(MODEL_PARAMS)
    torch.manual_seed(3)
    dec = Decoder(params, device="cuda").eval()
    enc = Encoder(params, device="cuda").eval()
    sdd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    sde = {k: v.detach().cpu() for k, v in enc.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    N = 5
    inputs, hidden, field = torch.randn(1, N, 4, generator=g), torch.randn(1, N, 128, generator=g) * 0.2, torch.randn(1, N, 2, generator=g) * 0.2
    masks = torch.tensor([0., 1., 0., 1., 0.])
    send, recv = get_knn_graph_info(inputs[0].cuda(), masks.cuda(), 2)
    e2n = torch.argsort(recv, stable=True).view(-1, 1)
    edges = torch.softmax(torch.randn(1, 2, 3, generator=g), -1)
    gi = (send.cpu(), recv.cpu(), e2n.cpu())
    wp, wh = DO.decoder_step(sdd, inputs, hidden, edges, masks, gi, field, True, "cart")
    gp, gh = dec(inputs.cuda(), hidden.cuda(), edges.cuda(), masks.cuda(), (send, recv, e2n), field.cuda())
    assert scale_rel_err(gp.cpu(), wp) <= TOL and scale_rel_err(gh.cpu(), wh) <= TOL
    node_inds = masks.nonzero()[:, -1]
    st = (torch.randn(1, N * (N - 1), 64, generator=g) * 0.2, torch.randn(1, N * (N - 1), 64, generator=g) * 0.2)
    wl, (wh0, wc0) = DO.encoder_single_step(sde, inputs, masks, node_inds, gi, st, field, "cart")
    gl, (gh0, gc0) = enc.single_step_forward(inputs.cuda(), masks.cuda(), node_inds.cuda(), (send, recv, e2n),
                                             (st[0].cuda(), st[1].cuda()), field.cuda())
    assert scale_rel_err(gl.cpu(), wl) <= TOL and scale_rel_err(gh0.cpu(), wh0) <= TOL and scale_rel_err(gc0.cpu(), wc0) <= TOL



def test_rollout_with_an_empty_step_equals_the_staged_loop():
    """A time step without any present object inside the sequence: aether_dyn_rollout writes zeros for it and leaves the
    prior / decoder state alone (aether_dynamicvars.py:841-843), exactly as the staged loop does; a step with ONE present
    object is refused by both (the reference fails there as well)."""
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    params = dict(MODEL_PARAMS, decoder_hidden=256, encoder_hidden=256, num_edge_types=4, pos_representation="cart",
                  field_hidden=128, encoder_rnn_hidden=64)
    torch.manual_seed(41)
    model = AetherDynamicVars(params, device=None).eval()
    perturb_bn_(model)
    model = model.cuda()
    g = torch.Generator().manual_seed(42)
    T, N = 7, 16
    inputs = torch.randn(1, T, N, 4, generator=g)
    counts = [9, 9, 0, 5, 0, 16, 16]
    masks = torch.zeros(1, T, N)
    for t, c in enumerate(counts):
        masks[0, t, torch.randperm(N, generator=g)[:c]] = 1
    burn = torch.ones(1, T, N)
    burn[:, 3:] = 0
    node_inds, graph_info, U = [], [], []
    for step in range(T):
        nv = counts[step]
        node_inds.append(masks[0, step].nonzero()[:, -1].cuda())
        if nv >= 2:
            send, recv = get_knn_graph_info(inputs[0, step].cuda(), masks[0, step].cuda(), nv)
            graph_info.append((send, recv, torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))))
            U.append(torch.rand(send.numel(), 4, generator=g).cuda())
        else:
            e = torch.empty(0, dtype=torch.int64, device="cuda")
            graph_info.append((e, e, e.view(0, 1)))
            U.append(torch.empty(0, 4, device="cuda"))
    args = (inputs.cuda(), masks.cuda(), [node_inds], [graph_info], burn.cuda())
    got = model.predict_future(*args, uniform=U[:T - 1])
    assert float(got[0, 2].abs().max()) == 0.0 and float(got[0, 4].abs().max()) == 0.0
    model.one_call_step = False
    try:
        want = model.predict_future(*args, uniform=U[:T - 1])
    finally:
        model.one_call_step = True
    assert torch.equal(got, want)
    one = masks.clone()
    one[0, 1] = 0
    one[0, 1, 3] = 1                                          # exactly one present object at step 1
    ni1 = list(node_inds)
    ni1[1] = one[0, 1].nonzero()[:, -1].cuda()
    with pytest.raises(_lib.AetherHipError):
        model.predict_future(inputs.cuda(), one.cuda(), [ni1], [graph_info], burn.cuda(), uniform=U[:T - 1])


def test_predict_future_captured_steps_equal_the_eager_loop():
    """``predict_future(graph=True)``: every step replays a captured hipGraph of its signature.  14 steps over scenes
    whose number of present objects changes (24 -> 12 -> 3 -> 40, the last being the size at which round 2's attempt
    ended in a memory access fault: workspaces re-allocated during capture) -- bit-identical to the eager
    loop, twice (the second pass replays every graph from the cache), and a mask that disagrees with node_inds raises."""
    from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
    import sys, os
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from make_golden_dynamicvars import MODEL_PARAMS, perturb_bn_
    params = dict(MODEL_PARAMS, decoder_hidden=256, encoder_hidden=256, num_edge_types=4, pos_representation="polar",
                  field_hidden=128, encoder_rnn_hidden=64)
    torch.manual_seed(31)
    model = AetherDynamicVars(params, device=None).eval()
    perturb_bn_(model)
    model = model.cuda()
    g = torch.Generator().manual_seed(32)
    T, N = 15, 40
    inputs = torch.randn(1, T, N, 4, generator=g)
    counts = [24, 24, 12, 12, 3, 3, 40, 40, 24, 12, 40, 3, 24, 40, 40]
    masks = torch.zeros(1, T, N)
    for t, c in enumerate(counts):
        masks[0, t, torch.randperm(N, generator=g)[:c]] = 1
    burn = torch.ones(1, T, N)
    burn[:, 5:] = 0
    node_inds, graph_info, U = [], [], []
    for step in range(T):
        nv = int(masks[0, step].sum())
        send, recv = get_knn_graph_info(inputs[0, step].cuda(), masks[0, step].cuda(), nv)
        e2n = torch.argsort(recv, stable=True).view(-1, min(10, nv - 1))
        graph_info.append((send, recv, e2n))
        node_inds.append(masks[0, step].nonzero()[:, -1].cuda())
        U.append(torch.rand(send.numel(), 4, generator=g).cuda())
    args = (inputs.cuda(), masks.cuda(), [node_inds], [graph_info], burn.cuda())
    rollout = model.predict_future(*args, uniform=U[:T - 1])       # default: ONE library call for the loop (aether_dyn_rollout)
    # the staged path (three library calls + torch glue per step) runs the same stage kernels on the same rows: identical
    # bits, eagerly and as captured steps
    model.one_call_step = False
    try:
        eager = model.predict_future(*args, uniform=U[:T - 1])
        assert torch.equal(eager, rollout)
        for rep in range(2):
            got = model.predict_future(*args, uniform=U[:T - 1], graph=True)
            assert torch.equal(got, eager), rep
        assert len(model._step_graphs) == 4                   # one graph per signature: 24, 12, 3 and 40 present objects
        # ... and so does aether_dyn_step called once per step from the host loop
        model.one_call_step = True
        ph, pc = model.encoder.get_initial_hidden(inputs.cuda())
        dec = model.decoder.get_initial_hidden(inputs.cuda())
        last = inputs[:, 0].cuda()
        for t in range(T - 1):
            obs = burn[:, t].cuda().unsqueeze(-1)
            state = obs * inputs[:, t].cuda() + (1 - obs) * last
            gs, gr, e2n = graph_info[t]
            if t == 3:      # node_inds left to the library (the mask's rows) and the sampled edge types handed back
                alt = model._step_one_call(state, masks[:, t].cuda(), node_inds[t], gs, gr, e2n, ph, pc, dec, U[t],
                                           pass_node_inds=False, return_edge_types=True)
                onehot = alt[4]
                assert onehot.shape == (gs.numel(), 4) and torch.equal(onehot.sum(1), torch.ones(gs.numel(), device="cuda"))
            last, ph, pc, dec = model._step_one_call(state, masks[:, t].cuda(), node_inds[t], gs, gr, e2n, ph, pc, dec, U[t])
            assert torch.equal(last, rollout[:, t]), t
            if t == 3:
                for a, b in zip(alt[:4], (last, ph, pc, dec)):
                    assert torch.equal(a, b)
    finally:
        model.one_call_step = True
    bad = masks.clone()
    bad[0, 0, :] = 1                                            # the mask now says 40 objects, node_inds still 24
    model._step_graphs.clear()
    model.one_call_step = False
    try:
        with pytest.raises(ValueError):
            model.predict_future(inputs.cuda(), bad.cuda(), [node_inds], [graph_info], burn.cuda(), uniform=U[:T - 1], graph=True)
    finally:
        model.one_call_step = True
    # eagerly the library notices by itself: NaN outputs for the step and an error from the next call / the explicit check
    out = model.predict_future(inputs[:, :2].cuda(), bad[:, :2].cuda(), [node_inds], [graph_info], burn[:, :2].cuda(),
                               uniform=U[:1])
    torch.cuda.synchronize()
    assert torch.isnan(out).all()
    assert _lib.load().aether_check_async_error() != 0
    assert _lib.load().aether_check_async_error() == 0
    # the one-call loop validates every step's graph BEFORE anything is queued -- a graph that lists another number
    # of edges than the encoder's kNN graph (n * min(k, n - 1); here a fully connected one on 24 objects), or an edge2node
    # that names an edge the graph does not have, raises instead of reading the first n * 10 edges / past the message rows
    nv = 24
    fc_s, fc_r = torch.where(~torch.eye(nv, dtype=torch.bool))
    wrong = list(graph_info)
    wrong[1] = (node_inds[1][fc_s.cuda()], node_inds[1][fc_r.cuda()], torch.argsort(fc_r, stable=True).view(nv, nv - 1).cuda())
    with pytest.raises(_lib.AetherHipError, match="kNN graph"):
        model.predict_future(inputs[:, :3].cuda(), masks[:, :3].cuda(), [node_inds], [wrong], burn[:, :3].cuda())
    wrong = list(graph_info)
    e2n_bad = graph_info[1][2].clone()
    e2n_bad[3, 2] = graph_info[1][0].numel() + 5
    wrong[1] = (graph_info[1][0], graph_info[1][1], e2n_bad)
    with pytest.raises(ValueError, match="edge2node"):
        model.predict_future(inputs[:, :3].cuda(), masks[:, :3].cuda(), [node_inds], [wrong], burn[:, :3].cuda())
    good = model.predict_future(inputs[:, :3].cuda(), masks[:, :3].cuda(), [node_inds], [graph_info], burn[:, :3].cuda(),
                                uniform=U[:2])
    assert torch.equal(good, rollout[:, :2])                    # the refused calls queued nothing and touched no state
