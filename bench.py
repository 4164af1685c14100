#!/usr/bin/env python3
"""Benchmark of the Aether state2state hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (under torch.distributed.run
for N > 1, one rank per GPU) prints ONE JSON line on rank 0.

* workload  : BASELINE.json configs[1] -- electrostatic 2-D, N=20 fully connected,
              batch=128 graphs per GPU (2,560 nodes, 48,640 edges), synthetic inputs
              (aether_amd/synthetic.py), seed-1 random-init weights.
* step      : one forward pass of the hot path (Aether.forward) over one resident batch.
* metric    : edge-messages/s = 4 * E * n_gpus / t_step   (one edge-message =
              one GNNLayer.message_fn evaluation on one edge; a forward does 4E of them).
* scaling   : weak -- every rank processes its own 128 graphs; the forward has no exchange
              step, so there is no data-path collective (graphs are independent).
* roofline  : dominant kernel k_edge_layer (layers 2-4): ALGORITHMIC flops per launch
              (reference formulation: E * 2*(192*64 + 64*64) = E * 32,768) / its average launch
              duration, measured with HIP events on the launch stream in a separate
              instrumented pass of the same K steps; peak = 157.3 TFLOP/s dense fp32 MFMA.
* cpu_baseline : the oracle (oracle/aether_oracle.py, PyTorch CPU restatement pinned to the
              reference) on the same batch on this host's cores, bounded to ~15 s.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOAD = dict(name="electrostatic-2d-N20-B128", B=128, N=20, D=2)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md: ~2.5 PF; never the 2:1-sparsity figure)
PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, dense
FLOP_PER_EDGE_LAYER_N = 2 * (192 * 64 + 64 * 64)      # locs.py:206-212 on [x_s|x_r|e] (192 -> 64 -> 64)
FLOP_PER_EDGE_STEP = {2: 108672, 3: 109824}
FLOP_PER_NODE_STEP = {2: 151936, 3: 152640}


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def _rollout_parity(sd, inp, model, dev, T=20):
    """20-step rollout (SURVEY.md 8d metric 2 protocol) of the HIP path and of the fp32 oracle, both against
    the oracle in fp64, on `inp` with the weights `sd` (loaded into `model`).  A graph is *cut-exposed* when the
    fp64 trajectory passes within 2e-5 rad of a branch cut of the reference's own feature map (oracle
    cut_margin: anti-parallel headings / sender exactly behind the receiver): the step is discontinuous there
    and an fp32 evaluation's side is decided by its last bit, so errors are quoted with and without them."""
    from oracle import aether_oracle as O
    from aether_amd.rollout import rollout
    N = inp["meta"]["N"]
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        t64, margins = O.rollout(sd64, inp["x"].double(), inp["vel"].double(), inp["edges"], inp["charges"].double(), T,
                                 with_margin=True)
        t32 = O.rollout(sd, inp["x"], inp["vel"], inp["edges"], inp["charges"], T)
        got = rollout(model, inp["x"].to(dev), inp["vel"].to(dev), [e.to(dev) for e in inp["edges"]],
                      inp["charges"].to(dev), T).cpu()
    recv = inp["edges"][1]
    exposed = sorted({int(recv[e]) // N for t in range(T) for e in torch.nonzero(margins[t] < 2e-5).flatten().tolist()})
    clean = torch.ones(t64.shape[1], dtype=torch.bool)
    for g in exposed:
        clean[g * N:(g + 1) * N] = False
    scale = float(t64.abs().max())
    rel = lambda a, m=None: float(((a.double() - t64).abs() if m is None else (a.double() - t64).abs()[:, m]).max()) / scale
    mse = lambda a: float(((a.double() - t64) ** 2).mean())
    return {"steps": T, "hip_vs_fp64_max_rel_err": rel(got), "oracle_fp32_vs_fp64_max_rel_err": rel(t32),
            "hip_vs_oracle_fp32_max_rel_err": float((got - t32).abs().max()) / scale,
            "cut_exposed_graphs": exposed, "n_graphs": int(t64.shape[1] // N),
            "hip_vs_fp64_max_rel_err_outside_exposed": rel(got, clean),
            "oracle_fp32_vs_fp64_max_rel_err_outside_exposed": rel(t32, clean),
            "hip_vs_fp64_mse": mse(got), "oracle_fp32_vs_fp64_mse": mse(t32), "tolerance": 1e-5}


def _mse_vs_simulated_truth(sd, model, dev, B, N, D, burn_in=29, pred=20):
    """Metric 2 as experiments/electrostatic/evaluate.py:33-70 runs it, for the state2state module, as a
    parity statement (oracle/metric2.py): the HIP device rollout AND the fp32 oracle against the fp64 oracle on
    trajectories of the build's electrostatic simulator -- with the stated seed-1 weights, whose rollout is chaotic
    (MSE 0.05 -> 5: no fp32 evaluation, the reference's included, holds 1e-5 there), and with the same model after 200
    captured training steps on one-frame targets of the burn-in frames, where the MSE difference does hold 1e-5."""
    from oracle import metric2 as M2
    data = M2.simulate(dev, B, N, D, burn_in, pred)
    out = {"seed1_weights": M2.report(sd, model, data)}
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    loss = M2.train_on_frames(model, data, steps=200, lr=1e-3)
    sd_t = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    out["after_200_training_steps"] = dict(M2.report(sd_t, model, data), final_training_loss=loss)
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    return out


def _parity(sd, inp, model, dev, sd_trained=None):
    """The oracle as the checker of the HIP path on the benchmark batch: one step, the 20-step rollout against
    fp64 (HIP and the fp32 oracle side by side), and metric 2 against simulated ground truth -- all with the
    stated seed-1 weights; the rollout again with the weights the training section left (`sd_trained`)."""
    from oracle import aether_oracle as O
    args = (sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    with torch.no_grad():
        want = O.aether_forward(*args)
        w64 = O.aether_forward({k: v.double() for k, v in sd.items()}, *[a.double() if torch.is_tensor(a) and a.is_floating_point() else a
                                                                       for a in args[1:]])
        got = model(inp["h"].to(dev), inp["x"].to(dev), [e.to(dev) for e in inp["edges"]], inp["vel"].to(dev),
                    inp["edge_attr"].to(dev), inp["charges"].to(dev)).cpu()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.abs().max())
    B, N, D = inp["meta"]["B"], inp["meta"]["N"], inp["meta"]["D"]
    out = {"weights": "torch.manual_seed(1) default init (the runner's seed, main.py:28)",
           "step_max_rel_err": rel(got, want), "step_hip_vs_fp64": rel(got, w64), "step_oracle_fp32_vs_fp64": rel(want, w64),
           "step_tolerance": 1e-5,
           "rollout20": _rollout_parity(sd, inp, model, dev),
           "rollout20_mse_vs_simulated_truth": _mse_vs_simulated_truth(sd, model, dev, B, N, D),
           "checker": "oracle/aether_oracle.py (pinned to the reference's golden vectors)"}
    if sd_trained is not None:
        out["rollout20_after_training"] = _rollout_parity(sd_trained, inp, model, dev)
    return out


def _cpu_baseline(sd, inp, model=None, dev=None, budget_s=15.0, sd_trained=None):
    """Oracle forward on the host cores; bounded sample of the same workload.  With `model`, the
    oracle also serves as the checker of the HIP path on this very batch (one step and the 20-step
    rollout of SURVEY.md 8d, scale-relative max error: the 1e-5 bar of the north star)."""
    from oracle import aether_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)       # the GPU box's CPU share per GPU; more threads only oversubscribe
    torch.set_num_threads(cores)
    args = (sd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
    with torch.no_grad():
        for _ in range(2):
            O.aether_forward(*args)
        t0 = time.perf_counter()
        n = 0
        while True:
            O.aether_forward(*args)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 400:
                break
    E = inp["edges"][0].numel()
    # ---- the same forward on ONE thread (SURVEY.md 8d asks for both), bounded
    torch.set_num_threads(1)
    with torch.no_grad():
        O.aether_forward(*args)
        t1 = time.perf_counter()
        n1 = 0
        while True:
            O.aether_forward(*args)
            n1 += 1
            el1 = time.perf_counter() - t1
            if el1 > budget_s / 3 or n1 >= 50:
                break
    torch.set_num_threads(cores)
    # ---- forward + backward + AdamW on the host cores (what the runner's training loop does per batch,
    # experiments/lorentz/main.py:289-292), torch autograd through the oracle, bounded
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.AdamW(list(psd.values()), lr=5e-4, weight_decay=1e-12)
    tgt = inp["target"]

    def cpu_train_step():
        opt.zero_grad(set_to_none=True)
        o = O.aether_forward(psd, inp["x"], inp["vel"], inp["edges"], inp["edge_attr"], inp["charges"])
        torch.nn.functional.mse_loss(o, tgt).backward()
        opt.step()
    cpu_train_step()
    t2 = time.perf_counter()
    n2 = 0
    while True:
        cpu_train_step()
        n2 += 1
        el2 = time.perf_counter() - t2
        if el2 > budget_s / 2 or n2 >= 50:
            break
    parity = None
    if model is not None:
        parity = _parity(sd, inp, model, dev, sd_trained)
    B, N, D = inp["meta"]["B"], inp["meta"]["N"], inp["meta"]["D"]
    return {
        "value": 4.0 * E * n / el, "unit": "edge-messages/s", "cores": cores, "kind": "port",
        "ms_per_step": 1e3 * el / n,
        "sample": f"{n} forward steps of the same B={B} N={N} D={D} batch in {el:.1f} s, "
                  f"torch CPU fp32, {cores} threads, seed-1 weights",
        "one_thread": {"value": 4.0 * E * n1 / el1, "ms_per_step": 1e3 * el1 / n1, "cores": 1,
                       "sample": f"{n1} forward steps in {el1:.1f} s"},
        "train_step": {"value": 4.0 * E * n2 / el2, "ms_per_step": 1e3 * el2 / n2, "cores": cores,
                       "sample": f"{n2} steps of forward + autograd backward + AdamW in {el2:.1f} s"},
        "parity": parity,
    }


def _other_configs(dev, budget_s=10.0):
    """what the driver's one run says about the configurations that are not the headline -- cfg3 forward,
    hidden_size 128 / 256 at the headline shape (csrc/wide.h), one GPU's share of cfg5 forward (3 steps), the variable-N
    prediction step (cfg4's model), the seq2seq model's prediction step (round 4).  Rank 0, after the headline's timed region; bounded: a leg starts only while the block
    is inside its budget, every leg is guarded (an error is recorded, the headline line is printed regardless)."""
    import contextlib
    import io
    from aether_amd import _lib
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch
    t_block = time.perf_counter()
    out = []
    quiet = lambda: contextlib.redirect_stdout(io.StringIO())

    def timed(fn, n, warm=3, blocks=1):
        """ms per call; with blocks > 1 the fastest of that many blocks of n calls (the short eager legs: one host hiccup
        inside a 15 ms window once put 3.2 ms on a 0.73 ms step)"""
        for _ in range(warm):
            fn()
        best = None
        for _ in range(blocks):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            best = ms if best is None or ms < best else best
        return best

    def leg(name, fn):
        if time.perf_counter() - t_block > budget_s:
            out.append({"workload": name, "skipped": "block budget spent"})
            return
        t0 = time.perf_counter()
        try:
            with torch.no_grad():
                rec = fn()
            rec = dict({"workload": name}, **rec)
            _lib.check(_lib.load().aether_check_async_error(), name)
        except Exception as ex:                                  # recorded; never at the cost of the headline line
            print(f"other_configs leg {name} failed:", repr(ex), file=sys.stderr)
            rec = {"workload": name, "error": repr(ex)}
        rec["leg_seconds"] = round(time.perf_counter() - t0, 2)
        out.append(rec)
        torch.cuda.empty_cache()

    def state2state(D, B, N, H, steps, graph):
        torch.manual_seed(1)
        with quiet():
            m = Aether(2 * D, H, 0.0, D, device=dev).eval()
        inp = make_batch(B, N, D, seed=0, device=dev)
        E = inp["edges"][0].numel()
        call = lambda: m(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
        call()
        torch.cuda.synchronize()
        launch = "eager"
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                call()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    call()
            ms = timed(g.replay, max(1, steps // 10), blocks=3) / 10
            launch = "hipgraph (10 steps per replay)"
        else:
            ms = timed(call, steps, blocks=3)
        flops = E * (FLOP_PER_EDGE_STEP[D] if H == 64 else 2 * (32 * H + H * H) + 6 * (3 * H * H + H * H))
        return {"ms_per_step": ms, "value": 4.0 * E / (ms * 1e-3), "unit": "edge-messages/s", "hidden": H, "edges": E,
                "launch": launch, "timing": "fastest of 3 blocks", "edge_mlp_algorithmic_tflops": flops / (ms * 1e-3) / 1e12}

    leg("cfg3 gravitational-3d-N20-B128 forward", lambda: state2state(3, 128, 20, 64, 200, True))
    leg("electrostatic-2d-N20-B128 forward, hidden_size 128 (csrc/wide.h)", lambda: state2state(2, 128, 20, 128, 30, False))
    leg("electrostatic-2d-N20-B128 forward, hidden_size 256 (csrc/wide.h)", lambda: state2state(2, 128, 20, 256, 20, False))

    def dyn_step():
        from aether_amd.knn import get_knn_graph_info
        from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
        mp = {"input_size": 4, "gpu": True, "decoder_hidden": 256, "num_edge_types": 4, "skip_first": True, "decoder_dropout": 0.0,
              "pos_representation": "cart", "no_encoder_bn": False, "encoder_dropout": 0.0, "encoder_hidden": 256,
              "encoder_rnn_hidden": 64, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 128,
              "prior_num_layers": 3, "prior_hidden_size": 128, "encoder_normalize_mode": "normalize_all", "train_data_len": 50,
              "field_hidden": 256, "gumbel_temp": 0.5}               # scripts/ind_aether.sh sizes
        torch.manual_seed(1)
        with quiet():
            model = AetherDynamicVars(mp, device=dev).eval()
        N, T = 20, 26
        g = torch.Generator().manual_seed(N)
        inputs = torch.randn(1, T, N, 4, generator=g).to(dev)
        masks = torch.ones(1, T, N, device=dev)
        burn = torch.ones(1, T, N, device=dev)
        burn[:, 10:] = 0
        node_inds, graph_info = [[]], [[]]
        for t in range(T):
            send, recv = get_knn_graph_info(inputs[0, t], masks[0, t], N)
            graph_info[0].append((send, recv, torch.argsort(recv, stable=True).view(-1, 10)))
            node_inds[0].append(torch.arange(N, device=dev))
        fn = lambda: model.predict_future(inputs, masks, node_inds, graph_info, burn)
        ms = timed(fn, 3, warm=1) / (T - 1)
        return {"ms_per_step": ms, "value": 1e3 / ms, "unit": "prediction steps/s (one scene)", "objects": N, "edges": 10 * N,
                "launch": "aether_dyn_rollout (one library call for the loop)"}

    leg("inD-sized variable-N prediction step, 20 objects, kNN k=10 (cfg4's model, one scene)", dyn_step)

    def dyn_step_64():
        # BASELINE config 4's batch: 64 scenes of 2..40 objects per prediction step, ONE library call for the loop
        from aether_amd.knn import get_knn_graph_info
        from aether_amd.nn.dynamicvars.aether_dynamicvars import AetherDynamicVars
        mp = {"input_size": 4, "gpu": True, "decoder_hidden": 256, "num_edge_types": 4, "skip_first": True, "decoder_dropout": 0.0,
              "pos_representation": "cart", "no_encoder_bn": False, "encoder_dropout": 0.0, "encoder_hidden": 256,
              "encoder_rnn_hidden": 64, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 128,
              "prior_num_layers": 3, "prior_hidden_size": 128, "encoder_normalize_mode": "normalize_all", "train_data_len": 50,
              "field_hidden": 256, "gumbel_temp": 0.5}
        torch.manual_seed(1)
        with quiet():
            model = AetherDynamicVars(mp, device=dev).eval()
        B, T, N = 64, 9, 40
        g = torch.Generator().manual_seed(7)
        inputs = torch.randn(B, T, N, 4, generator=g).to(dev)
        masks = torch.zeros(B, T, N)
        for b in range(B):
            c = int(torch.randint(2, N + 1, (1,), generator=g))
            masks[b, :, torch.randperm(N, generator=g)[:c]] = 1
        masks = masks.to(dev)
        burn = torch.ones(B, T, N, device=dev)
        burn[:, 4:] = 0
        # the masks do not change over time here: one kNN graph per scene serves every step's graph_info (the model's own
        # kNN graph is rebuilt from the predicted state inside every step)
        node_inds, graph_info = [], []
        for b in range(B):
            nv = int(masks[b, 0].sum())
            send, recv = get_knn_graph_info(inputs[b, 0], masks[b, 0], nv)
            gi = (send, recv, torch.argsort(recv, stable=True).view(-1, min(10, nv - 1)))
            node_inds.append([masks[b, 0].nonzero()[:, -1]] * T)
            graph_info.append([gi] * T)
        fn = lambda: model.predict_future(inputs, masks, node_inds, graph_info, burn)
        ms = timed(fn, 2, warm=1) / (T - 1)
        return {"ms_per_step": ms, "value": 1e3 * B / ms, "unit": "scene-steps/s", "scenes": B,
                "objects": int(masks[:, 0].sum()), "edges": sum(int(gi[0][0].numel()) for gi in graph_info),
                "launch": "aether_dyn_rollout_batched (one library call for the loop, host-side list handling included)"}

    leg("BASELINE cfg4: 64 inD-sized scenes per variable-N prediction step (2..40 objects, kNN k=10)", dyn_step_64)

    def s2s_step(D, N, B, hd, T, reps):
        # the seq2seq model's autoregressive prediction step: device-side rollout, ONE library call for T steps
        from aether_amd.nn.seq2seq.aether import Aether as S2SAether
        H, R = 512, 128
        params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": hd, "num_edge_types": 2,
                  "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3, "encoder_dropout": 0.0, "encoder_hidden": H,
                  "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3, "encoder_mlp_hidden": 256,
                  "prior_num_layers": 3, "prior_hidden_size": 256, "pos_representation": "polar" if D == 2 else "cart",
                  "gumbel_temp": 0.5, "rff_std": 1.0}
        torch.manual_seed(0)
        with quiet():
            m = S2SAether(params, device=dev).eval()
        E = N * (N - 1)
        x = torch.randn(B, N, 2 * D, device=dev)
        dh = torch.zeros(B, N, hd, device=dev)
        ps = (torch.zeros(B, E, R, device=dev), torch.zeros(B, E, R, device=dev))
        U = torch.rand(T, B, E, 2, device=dev)
        ms = timed(lambda: m.predict_from_state(x, dh, ps, T, uniform=U), reps, warm=2) / T
        return {"ms_per_step": ms, "value": B * E / (ms * 1e-3), "unit": "edge-steps/s", "edges": B * E, "encoder_hidden": H,
                "decoder_hidden": hd, "launch": "aether_s2s_rollout (one library call for the loop)"}

    leg("seq2seq prediction step, gravitational-3d N=5 B=128 (the reference's own size)", lambda: s2s_step(3, 5, 128, 256, 20, 3))
    leg("seq2seq prediction step, 2-d N=20 B=128", lambda: s2s_step(2, 20, 128, 512, 10, 2))

    def cfg5_shard():
        # one GPU's share of BASELINE config 5 (32 graphs of 1,024 bodies, 33.5 M edges), inputs drawn on the device
        from aether_amd.edges import get_edges, prepare_edge_attr
        B, N, D = 32, 1024, 2
        torch.manual_seed(1)
        with quiet():
            m = Aether(2 * D, 64, 0.0, D, device=dev).eval()
        gen = torch.Generator(device=dev).manual_seed(5)
        x = torch.randn(B * N, D, generator=gen, device=dev) * (N / 5.0) ** (1.0 / 3.0)
        v = torch.randn(B * N, D, generator=gen, device=dev)
        v = v * 0.5 / v.norm(dim=-1, keepdim=True)
        q = torch.randint(0, 2, (B * N, 1), generator=gen, device=dev).float() * 2.0 - 1.0
        edges = get_edges(B, N, device=dev)
        ea = prepare_edge_attr(x, edges, q[edges[0]] * q[edges[1]])
        E = edges[0].numel()
        call = lambda: m(None, x, edges, v, ea, q)
        ms = timed(call, 3, warm=1)
        return {"ms_per_step": ms, "value": 4.0 * E / (ms * 1e-3), "unit": "edge-messages/s", "edges": E, "launch": "eager",
                "hbm_frac_of_8TBps_at_1560B_per_edge_step": 1560.0 * E / (ms * 1e-3) / 8e12}

    leg("cfg5 shard synthetic-2d-N1024-B32 forward (33.5 M edges, streamed path)", cfg5_shard)
    return {"budget_s": budget_s, "block_seconds": round(time.perf_counter() - t_block, 2), "legs": out}


def _seq2seq_traffic(workload):
    """HBM bytes per launch of the filter GEMM from the committed PMC passes (profiles/traffic.json), or None."""
    try:
        import json
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get("k_s2s_filter_hbm_bytes_per_launch")
    except Exception:
        return None


def _seq2seq_line(args):
    """One autoregressive step of the seq2seq model (field query -> prior step -> hard Gumbel sample ->
    decoder step; nn/seq2seq/aether.py:175-185) on one GPU, B x N from --batch / --nodes, h = 512.  A
    separate line: the headline metric of BASELINE.json is the state2state step (default mode)."""
    import contextlib, io
    from aether_amd.nn.seq2seq.aether import Aether as S2S
    B, N, D, H, R = args.batch, args.nodes, args.dims, 512, 128
    params = {"num_vars": N, "input_size": 2 * D, "gpu": True, "decoder_hidden": H, "num_edge_types": 2,
              "skip_first": False, "decoder_dropout": 0.0, "use_3d": D == 3, "encoder_dropout": 0.0,
              "encoder_hidden": H, "encoder_rnn_hidden": R, "encoder_rnn_type": "lstm", "encoder_mlp_num_layers": 3,
              "encoder_mlp_hidden": 256, "prior_num_layers": 3, "prior_hidden_size": 256,
              "pos_representation": "polar", "gumbel_temp": 0.5}
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        m = S2S(params, device="cuda").eval()
    E = N * (N - 1)
    x = torch.randn(B, N, 2 * D, device="cuda")
    hid = torch.zeros(B, N, H, device="cuda")
    ps = (torch.zeros(B, E, R, device="cuda"), torch.zeros(B, E, R, device="cuda"))
    U = torch.rand(B, E, 2, device="cuda")

    def step_modules():                                   # the four per-module entry points, as the reference's loop calls them
        f, _ = m.predict_field(x)
        lg, s2 = m.encoder.single_step_forward(x, ps, f)
        return m.single_step_forward(x, hid, lg, True, f, U)

    def step():                                           # the product path of predict_future: one C call (aether_s2s_step)
        return m._fused_step(x, hid, ps, U)

    for _ in range(3):
        step_modules()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step_modules()
    torch.cuda.synchronize()
    modules_ms = (time.perf_counter() - t0) / 10 * 1e3
    for _ in range(max(1, min(args.warmup, 5))):
        step()
    torch.cuda.synchronize()
    steps = max(1, min(args.steps, 50))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # dominant kernel: the anisotropic-filter GEMM of the prior step, timed with events on the launch stream
    f, _ = m.predict_field(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 10
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        m.encoder.single_step_forward(x, ps, f)
    ev[1].record()
    torch.cuda.synchronize()
    prior_ms = ev[0].elapsed_time(ev[1]) / reps
    nrf = 4 * D + D * (D - 1) // 2
    r_feat = 2 * nrf + 3 * D
    filt_flop = float(B * E) * r_feat * H * H * 2
    line = {"metric": "seq2seq autoregressive edge-steps/sec (field + prior + sample + decoder step, h=512)",
            "value": B * E / dt, "unit": "edge-steps/s", "n_gpus": 1, "steps": steps, "warmup": min(args.warmup, 5),
            "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"seq2seq-{D}d-N{N}-B{B}-h{H}", "num_dims": D, "nodes_per_graph": N,
                       "graphs_per_gpu": B, "edges_per_gpu": B * E, "hidden": H,
                       "launch": "eager, one C call per step (aether_s2s_step: 31 launches on a plan of prepared weights)"},
            "four_entry_points_ms_per_step": modules_ms,
            "roofline": {"bound": "mfma", "kernel": "prior step (k_s2s_filter dominates)", "unit": "TFLOP/s",
                         "achieved": filt_flop / (prior_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                         "frac": filt_flop / (prior_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": _seq2seq_traffic(f"seq2seq-{D}d-N{N}-B{B}-h{H}"),
                         "avg_launch_us": prior_ms * 1e3,
                         "algorithmic_flop_per_launch": filt_flop,
                         "note": "algorithmic = the filter contraction alone (R h^2 MACs per edge); the time is the "
                                 "whole prior step, so the fraction is a lower bound for the filter GEMM.  The GEMM runs as six "
                                 "bf16 MFMA terms on 3-way split operands (fp32-equivalent): a fraction above 1 of the fp32-MFMA "
                                 "peak quoted here is possible; of the bf16 pipe's 2,500 TFLOP/s it is 6 x achieved / 2500"},
            "cpu_baseline": None}
    if not args.no_cpu_baseline:
        # the oracle's step on the host cores (bounded sample), which also checks the HIP step on this batch
        from oracle import seq2seq_oracle as SO
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        enc_sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        dec_sd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        xc, hc, Uc = x.cpu(), hid.cpu(), U.cpu()
        psc = (ps[0].cpu(), ps[1].cpu())

        def cpu_step():
            f = SO.predict_field(sd, xc, D)
            lg, st2 = SO.prior_step(enc_sd, xc, psc, f, D == 3, "polar", 3)
            z = SO.gumbel_hard(lg.reshape(-1, 2), Uc.reshape(-1, 2), 0.5).view(lg.shape)
            out, h2 = SO.decoder_step(dec_sd, xc, hc, z, f, D == 3)
            return f, lg, z, out, h2

        with torch.no_grad():
            want = cpu_step()
            t0 = time.perf_counter()
            n = 0
            while True:
                cpu_step()
                n += 1
                el = time.perf_counter() - t0
                if el > 12.0 or n >= 20:
                    break
            f_g, _ = m.predict_field(x)
            lg_g, _ = m.encoder.single_step_forward(x, ps, f_g)
            out_g, h_g, z_g = m.single_step_forward(x, hid, want[1].cuda(), True, want[0].cuda(), U)
        rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
        line["cpu_baseline"] = {
            "value": B * E * n / el, "unit": "edge-steps/s", "cores": cores, "kind": "port",
            "ms_per_step": 1e3 * el / n,
            "sample": f"{n} steps of the same B={B} N={N} batch in {el:.1f} s, torch CPU fp32, {cores} threads",
            "parity": {"field_max_rel_err": rel(f_g, want[0]), "prior_logits_max_rel_err": rel(lg_g, want[1]),
                       "sampled_types_equal": bool(torch.equal(z_g.cpu().argmax(-1), want[2].argmax(-1))),
                       "decoder_outputs_max_rel_err": rel(out_g, want[3]),
                       "decoder_hidden_max_rel_err": rel(h_g, want[4]), "tolerance": 1e-5,
                       "checker": "oracle/seq2seq_oracle.py (pinned to the reference's golden vectors)"}}
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of hipGraph replay")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed device warm-up (forward steps) before the W warm-up steps, in milliseconds; 0 = none")
    ap.add_argument("--no-rollout", action="store_true", help="skip the 20-step device rollout timing")
    ap.add_argument("--graph-steps", type=int, default=10,
                    help="consecutive forward steps captured per hipGraph (1: one replay per step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: all ranks use cuda:0 (with --backend gloo) to rehearse the N > 1 path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step (fwd+bwd+opt) figure")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the bounded block of non-headline configurations (cfg3, wide hidden, cfg5 shard, variable-N step)")
    ap.add_argument("--streamed", action="store_true", help="force the layer-by-layer kernels")
    ap.add_argument("--opt", action="append", default=[], help="name=value passed to aether_set_option")
    ap.add_argument("--dims", type=int, default=WORKLOAD["D"], help="2 (headline) or 3 (cfg3)")
    ap.add_argument("--batch", type=int, default=WORKLOAD["B"])
    ap.add_argument("--nodes", type=int, default=WORKLOAD["N"])
    ap.add_argument("--seq2seq", action="store_true",
                    help="time the seq2seq model's autoregressive step (SURVEY 8a rows A8-A10) instead: its own JSON line")
    ap.add_argument("--config", choices=["cfg1", "cfg2", "cfg3", "cfg5shard", "cfg5"], default=None,
                    help="BASELINE.json configuration: cfg1 2-D N=5 B=1, cfg2 (default) 2-D N=20 B=128, cfg3 3-D N=20 B=128, "
                         "cfg5shard 2-D N=1024 B=32 (one GPU's share of config 5, streamed path), cfg5 the whole of config 5 "
                         "(256 graphs of 1,024 bodies) split over the --gpus ranks with aether_amd.parallel.shard_graphs: "
                         "STRONG scaling, every rank walks its graphs in chunks of 32")
    ap.add_argument("--graph-collective", action="store_true",
                    help="data-parallel training step: capture the gradient all-reduce inside the hipGraph (untested on "
                         "multi-GPU hardware; falls back to two graphs around an eager collective)")
    args = ap.parse_args()
    args.big = False
    args.chunks = 1
    if args.config is not None:
        args.dims, args.nodes, args.batch = {"cfg1": (2, 5, 1), "cfg2": (2, 20, 128), "cfg3": (3, 20, 128),
                                             "cfg5shard": (2, 1024, 32), "cfg5": (2, 1024, 32)}[args.config]
        if args.config == "cfg5":
            # strong scaling: 256 graphs in total, rank r takes shard_graphs(256, r, world) and walks them 32 at a time
            # (one 33.5 M-edge chunk is what the workspace is sized for; graphs are independent, so chunks are too)
            from aether_amd.parallel import shard_graphs
            r_, w_, _ = _dist_env()
            lo, hi = shard_graphs(256, r_, w_)
            if (hi - lo) % 32:
                raise SystemExit("--config cfg5 needs 256 / world to be a multiple of 32 (1, 2, 4 or 8 ranks)")
            args.chunks = (hi - lo) // 32
        if args.config in ("cfg5shard", "cfg5"):      # 33.5 M edges: a step is 25 ms, the CPU legs would take minutes
            args.no_cpu_baseline, args.no_rollout = True, True
            args.steps, args.warmup = min(args.steps, 20), min(args.warmup, 3)
            args.big = True                           # training figure from 3 eager steps (a backward is ~0.1 s here)
    if args.seq2seq:
        return _seq2seq_line(args)

    rank, world, local = _dist_env()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist
    from aether_amd import _lib
    from aether_amd.nn.state2state.aether import Aether
    from aether_amd.synthetic import make_batch

    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    ranks_seen = None
    if world > 1:
        # which device does every rank sit on?  (uuid / PCI bus id: a SCALE line should show N distinct GPUs.)  Plain
        # tensor all_gather of a fixed-size byte string -- nothing but the collective the backend certainly has.
        pr = torch.cuda.get_device_properties(dev)
        ident = "|".join(str(x) for x in (getattr(pr, "uuid", ""), getattr(pr, "pci_domain_id", ""), getattr(pr, "pci_bus_id", ""),
                                          getattr(pr, "pci_device_id", ""), pr.name, os.uname().nodename, local))[:120]
        mine = torch.zeros(128, dtype=torch.uint8, device=dev)
        raw = torch.tensor(list(ident.encode()), dtype=torch.uint8, device=dev)
        mine[:raw.numel()] = raw
        allb = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        idents = [bytes(b.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for b in allb]
        ranks_seen = {"devices": idents, "distinct_devices": len({i.rsplit("|", 1)[0] for i in idents}), "world_size": world}

    B, N, D = args.batch, args.nodes, args.dims
    torch.manual_seed(1)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model = Aether(2 * D, 64, 0.0, D, device=dev)
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}     # the stated seed-1 weights
    if args.streamed:
        model.flags = _lib.FLAG_FORCE_STREAMED
    if args.share_gpu and world > 1:
        # rehearsal only: ranks sharing one GPU cannot count on both workgroups of a split pair being resident together
        _lib.check(_lib.load().aether_set_option(b"fused_split", 0), "set_option fused_split")
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.check(_lib.load().aether_set_option(k.encode(), int(v)), "set_option " + kv)
    host = make_batch(B, N, D, seed=rank)              # every rank its own graphs (weak scaling)
    inp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in host.items()}
    inp["edges"] = [e.to(dev) for e in host["edges"]]
    E = inp["edges"][0].numel()
    Nn = B * N
    call1 = lambda: model(inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"])
    if args.chunks > 1:
        def call():                                    # this rank's share of config 5: `chunks` independent 32-graph chunks
            for _ in range(args.chunks):
                o = call1()
            return o
    else:
        call = call1

    model.eval()                                       # inference figures (the reference's test loop does the same)
    with torch.no_grad():
        out = call()                                   # builds the graph view + workspace
        torch.cuda.synchronize()
        use_graph = not args.no_graph
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    call()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = call()
            step = g.replay
            # `graph_steps` consecutive steps per hipGraph (as a rollout would be captured): the
            # launch gap between two graph replays is paid once per group instead of once per step
            S = max(1, min(args.graph_steps, args.steps))
            if S > 1:
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs):
                    for _ in range(S):
                        out = call()
                group = gs.replay
            if world > 1:
                # Guard for ranks that share one GPU (rehearsals with --share-gpu): hipGraph replays from two
                # processes on the same device serialise pathologically (71 ms per step measured, against 0.1 ms for
                # eager launches).  Every rank times both ways for a few steps; if graph replay is not faster on some
                # rank, all ranks launch eagerly.  One process per GPU keeps the graph.
                def _timed(fn, n):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / n
                (group if S > 1 else step)()
                t_graph = _timed(group, 2) / S if S > 1 else _timed(step, 10)
                t_eager = _timed(call, 10)
                flag = torch.tensor([1.0 if t_graph > 1.5 * t_eager else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if float(flag.item()) > 0:
                    use_graph, step, S = False, call, 1
        else:
            step = call
            S = 1

        def run_steps(k):
            if S > 1:
                for _ in range(k // S):
                    group()
                k = k % S
            for _ in range(k):
                step()

        # device warm-up before the W warm-up steps of the contract: an idle MI355X takes tens of milliseconds of load to
        # reach its sustained clocks (the first 10 ms region after 20 warm-up steps ran 5-6 % slower than the regions
        # after it, `ms_per_step_repeats`); untimed, reported as `pre_warm_ms`
        if args.prewarm_ms > 0:
            t_pw = time.perf_counter()
            while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # graph replays never pass through a C entry point: ask for asynchronous kernel errors of the timed region here
        _lib.check(_lib.load().aether_check_async_error(), "asynchronous kernel error in the timed region")
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # the same K steps a few more times: `value` stays the contract's single timed region, the spread goes next to it
        # (with the driver's --steps 20 that region is ~1 ms; VERDICT r1 asked for a robust companion)
        rep_ms = []
        if not args.big:
            for _ in range(5):
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                run_steps(args.steps)
                torch.cuda.synchronize()
                rep_ms.append(1e3 * (time.perf_counter() - t0) / args.steps)

        # ---- instrumented pass: per-kernel HIP events on the launch stream (eager) --------
        roof = None
        kernels = {}
        if rank == 0:
            lib = _lib.load()
            nk = lib.aether_profile_kernels()
            lib.aether_profile_enable(1)
            ksteps = min(args.steps, 8192 // 9)
            for _ in range(ksteps):
                call()
            torch.cuda.synchronize()
            ms = (C.c_double * nk)()
            cnt = (C.c_int64 * nk)()
            _lib.check(lib.aether_profile_read(ms, cnt, nk), "aether_profile_read")
            lib.aether_profile_enable(0)
            for k in range(nk):
                if cnt[k]:
                    kernels[lib.aether_profile_kernel_name(k).decode()] = {
                        "launches_per_step": cnt[k] / ksteps, "avg_us": 1e3 * ms[k] / cnt[k]}
            step_flops_alg = float(E) * FLOP_PER_EDGE_STEP[D] + float(Nn) * FLOP_PER_NODE_STEP[D]
            graph_us = None
            if "k_fused" in kernels:       # one launch does the whole step
                dom_name, dom, flops = "k_fused", kernels["k_fused"], step_flops_alg
                executed = None
                if use_graph:
                    # the same launch mode as `ms_per_step`: HIP events on the stream around replays of the S-step graph
                    # (one k_fused launch per step, back to back) -- per launch it includes the gap to the next launch
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    reps = max(1, min(args.steps, 200) // S)
                    (group if S > 1 else step)()
                    torch.cuda.synchronize()
                    ev[0].record()
                    for _ in range(reps):
                        (group if S > 1 else step)()
                    ev[1].record()
                    torch.cuda.synchronize()
                    graph_us = 1e3 * ev[0].elapsed_time(ev[1]) / (reps * S)
            else:
                # the node-term split leaves this kernel E x (W_e e: 2*64*64 + W2 h: 2*64*64) FLOP; the whole step's
                # algorithmic rate (reference formulation) is `step_algorithmic_tflops` of the line
                dom_name, dom = "k_edge_layer", kernels.get("k_edge_layer")
                flops = float(E) * 2 * (64 * 64 * 2)
                executed = float(E) * 2 * (64 * 64 * 2 + 16 * 64)        # + the per-tile receiver sums on the matrix core
            if dom:
                launch_us = graph_us if graph_us is not None else dom["avg_us"]
                achieved = flops / (launch_us * 1e-6) / 1e12
                traffic = None
                tpath = os.path.join(REPO, "profiles", "traffic.json")
                if os.path.exists(tpath):
                    try:
                        tj = json.load(open(tpath))
                        if tj.get("workload") == WORKLOAD["name"] and (B, N, D) == (128, 20, 2) and dom_name == "k_fused":
                            traffic = tj.get(dom_name + "_hbm_bytes_per_launch")
                            if executed is None:       # PMC SQ_INSTS_MFMA by instruction kind, same separate passes
                                executed = tj.get(dom_name + "_executed_flop_per_launch")
                        elif (B, N, D) == (32, 1024, 2) and dom_name == "k_edge_layer":
                            traffic = tj.get("cfg5shard", {}).get("k_edge_layer_hbm_bytes_per_launch")
                    except Exception:
                        traffic = None
                roof = {"bound": "mfma", "kernel": dom_name, "achieved": achieved,
                        "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                        "avg_launch_us": launch_us,
                        "launch_mode": ("hipgraph replay (same mode as ms_per_step; includes the gap between launches)"
                                        if graph_us is not None else "eager, HIP events around each launch"),
                        "eager_avg_launch_us": dom["avg_us"], "algorithmic_flop_per_launch": flops,
                        "executed_flop_per_launch": executed,
                        "source_of_traffic_and_executed": "profiles/traffic.json (rocprofv3 --pmc passes of this command)"
                                                          if traffic is not None else None}
                # Which pipe binds?  `frac` above prices the reference formulation's flops against the fp32 MFMA peak (the
                # precision delivered); the kernel executes its contractions as bf16 terms, so three more fractions say what
                # the hardware is doing (PMC passes of this command, profiles/traffic.json): the bf16 matrix pipe, the fp32
                # MFMAs (which issue on the vector ALU) and the vector ALU's instruction issue.  None can exceed 1.
                pm = tj.get(dom_name + "_pipes") if traffic is not None and isinstance(tj, dict) else None
                if pm is None and os.path.exists(tpath) and dom_name == "k_fused":      # other measured shapes: keyed by workload
                    try:
                        pm = json.load(open(tpath)).get(f"D{D}-N{N}-B{B}", {}).get("k_fused_pipes")
                    except Exception:
                        pm = None
                if pm:
                    t_s = launch_us * 1e-6
                    clk, simds = 2.4e9, 1024                      # peak shader clock the guide's peaks assume; 256 CUs x 4 SIMDs
                    pipes = {
                        "fp32_equivalent": roof["frac"],
                        "bf16_matrix_pipe": pm["bf16_mfma_insts"] * 16384.0 / t_s / (PEAK_BF16_MFMA_TFLOPS * 1e12),      # (key kept: 16-bit MFMAs, fp16 since round 4)
                        "fp32_mfma_on_valu": pm["fp32_mfma_insts"] * 2048.0 / t_s / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                        "valu_issue": (pm["valu_insts"] * 4.0 + pm["fp32_mfma_insts"] * 32.0) / (simds * t_s * clk),
                    }
                    roof["pipes"] = {k: float(f"{v:.4f}") for k, v in pipes.items()}
                    roof["pipes_note"] = ("bf16_matrix_pipe: executed 16-bit MFMAs (v_mfma_f32_16x16x32_f16 since round 4) x 16,384 FLOP over 2.5 PFLOP/s dense; "
                                          "fp32_mfma_on_valu: executed v_mfma_f32_16x16x4_f32 x 2,048 FLOP over 157.3 TFLOP/s; valu_issue: "
                                          "(VALU instructions x 4 cycles + fp32 MFMAs x 32 cycles) over 1,024 SIMDs x launch time x 2.4 GHz "
                                          "-- a lower bound (transcendentals take 8); measured matrix-pipe busy "
                                          f"{pm.get('matrix_pipe_busy')}")
                    exec_only = {k: v for k, v in pipes.items() if k != "fp32_equivalent"}
                    roof["binding_pipe"] = max(exec_only, key=exec_only.get)
                    assert all(v <= 1.0 for v in exec_only.values()), exec_only
                if roof["frac"] > 1.0:
                    # possible off the headline size: `achieved` counts the reference formulation's flops (192 -> 64 first
                    # layer per edge), the kernel executes ~0.6x of them (node-term split) and runs the 64 x 64
                    # contractions as bf16 terms on the matrix pipe, whose peak is above the fp32 MFMA peak quoted here
                    roof["frac_note"] = ("above 1: algorithmic flops of the reference formulation over the fp32 MFMA peak; "
                                         "the kernel executes fewer flops and uses the bf16 pipe (see dtype_note)")
                if dom_name == "k_edge_layer":
                    # HBM side of the streamed edge kernel (SURVEY 8d asks for both fractions): algorithmic bytes per
                    # edge and layer = read e_{l-1} 256 + write e_l 256 + indices 8 + partial rows 16
                    bytes_alg = float(E) * 536.0
                    roof["hbm"] = {"algorithmic_bytes_per_launch": bytes_alg, "achieved_GBps": bytes_alg / (launch_us * 1e-6) / 1e9,
                                   "peak_GBps": 8000.0, "hbm_frac": bytes_alg / (launch_us * 1e-6) / 8e12,
                                   "measured_bytes_per_launch": traffic}

    # ---- 20-step device rollout (aether_rollout): metric 2's protocol, one launch per step ------------
    roll = None
    if rank == 0 and not args.no_rollout:
        try:                                            # (guarded like the training leg: never at the cost of the headline line)
            from aether_amd.rollout import rollout, rollout_stepwise
            T, R = 20, 10
            rargs = (model, inp["x"], inp["vel"], inp["edges"], inp["charges"], T)
            def timed(fn):
                for _ in range(2):
                    fn(*rargs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(R):
                    fn(*rargs)
                torch.cuda.synchronize()
                return 1e3 * (time.perf_counter() - t0) / R
            ms_dev, ms_loop = timed(rollout), timed(rollout_stepwise)
            roll = {"steps": T, "ms_per_rollout": ms_dev, "ms_per_step": ms_dev / T,
                    "value": 4.0 * E * T / (ms_dev * 1e-3), "unit": "edge-messages/s",
                    "loop_of_module_calls_ms": ms_loop,
                    "includes": "edge attributes and velocities derived in the kernels, eager launches"}
        except Exception as ex:
            print("rollout leg failed:", repr(ex), file=sys.stderr)
            roll = {"error": repr(ex)}

    # ---- training step: forward + HIP backward + gradient all-reduce (N > 1) + AdamW ----------------
    train = None
    train_failed_here = 0
    if not args.no_train and args.chunks == 1:        # (config 5 on fewer than 8 ranks: forward figure only)
        # The whole leg runs inside a function: whatever goes wrong in it (a collective that fails on every rank, a capture
        # that is refused) must not cost the run its headline line -- the forward figure above is already measured.
        def train_leg():
            train = None
            model.train()
            if world > 1:
                from aether_amd.parallel import attach_data_parallel
                attach_data_parallel(model)
            tgt = inp["target"]
            if args.big:
                # config-5 shard: the library's own loss and optimizer (one launch each), launched eagerly -- at ~85 ms per
                # step launch gaps are nothing, and a captured step would hold a second 120 GB workspace in the graph's pool
                from aether_amd.optim import FusedAdamW, mse_loss_grad
                opt = FusedAdamW(model.parameters(), lr=5e-4, weight_decay=1e-12)      # main.py:86,164

                def tstep():
                    opt.zero_grad(set_to_none=True)
                    o = call()
                    _loss, grad = mse_loss_grad(o, tgt)
                    o.backward(grad)
                    opt.step()
            else:
                opt = torch.optim.AdamW(model.parameters(), lr=5e-4, weight_decay=1e-12)     # main.py:86,164

                def tstep():
                    opt.zero_grad(set_to_none=True)
                    o = call()
                    loss = torch.nn.functional.mse_loss(o, tgt)
                    loss.backward()
                    opt.step()
            tsteps = 3 if args.big else max(10, args.steps // 4)
            for _ in range(1 if args.big else 5):
                tstep()
            torch.cuda.synchronize()
            train_launch = "eager"
            gstep = None
            if use_graph and not args.big:
                # whole training step as hipGraph replays (aether_amd.training.GraphedTrainStep): one graph at N = 1; with
                # N > 1 forward + backward replay as one graph, the flat gradient buffer is all-reduced eagerly (RCCL),
                # AdamW (one launch, aether_amd.optim.FusedAdamW) replays as a second graph
                try:
                    from aether_amd.training import GraphedTrainStep
                    gstep = GraphedTrainStep(model, [inp["h"], inp["x"], inp["edges"], inp["vel"], inp["edge_attr"], inp["charges"]],
                                             tgt, lr=5e-4, weight_decay=1e-12, graph_collective=args.graph_collective)
                    eager_tstep = tstep
                    tstep = gstep.step
                    for _ in range(3):
                        tstep()
                    torch.cuda.synchronize()
                    train_launch = ("hipgraph" if world == 1 else
                                    "one hipgraph incl. the all-reduce" if gstep.collective_in_graph else
                                    "hipgraph (forward + backward) + eager all-reduce + hipgraph (AdamW, mean folded into its gradient read)")
                except Exception as ex:          # keep the eager figure if capture is not possible
                    print("train-step graph capture failed:", repr(ex), file=sys.stderr)
                    gstep = None
                    torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(tsteps):
                tstep()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tdt = time.perf_counter() - t0
            _lib.check(_lib.load().aether_check_async_error(), "asynchronous kernel error in the timed training steps")
            if world > 1:
                t = torch.tensor([tdt], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tdt = float(t.item())
            coll = None
            if world > 1:
                # the collective alone: the flat gradient buffer of the model, timed with events on the compute stream
                flat = model._grad_buffers()[0]
                grp = gstep.dp_group if gstep is not None else model.dp_group
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                for _ in range(3):
                    dist.all_reduce(flat, group=grp)
                torch.cuda.synchronize()
                ev[0].record()
                for _ in range(20):
                    dist.all_reduce(flat, group=grp)
                ev[1].record()
                torch.cuda.synchronize()
                coll = {"backend": dist.get_backend(grp), "world_size": dist.get_world_size(grp),
                        "allreduce_bytes": flat.numel() * 4, "allreduce_us": 1e3 * ev[0].elapsed_time(ev[1]) / 20,
                        "in_graph": bool(gstep is not None and gstep.collective_in_graph), "ranks_seen": ranks_seen}
            train = {"ms_per_step": 1e3 * tdt / tsteps, "steps": tsteps, "collective": coll,
                     "value": 4.0 * E * world / (tdt / tsteps), "unit": "edge-messages/s",
                     "includes": ("forward + MSE loss (aether_mse_loss_grad) + HIP backward + " if (gstep is not None or args.big)
                                  else "forward + torch MSE loss + HIP backward + ")
                                 + ((("RCCL" if coll["backend"] == "nccl" else coll["backend"]) + " grad all-reduce + ") if world > 1 else "")
                                 + ("AdamW (aether_adamw_step), " if (gstep is not None or args.big) else "torch AdamW, ")
                                 + train_launch + " launches"}
            if rank == 0:       # per-kernel breakdown of one training step (rank-local: no collective in here)
                saved_group, model.dp_group = model.dp_group, None       # (GraphedTrainStep already detached it)
                lib = _lib.load()
                nk = lib.aether_profile_kernels()
                lib.aether_profile_enable(1)
                ks = 2 if args.big else 10
                if gstep is not None:
                    tstep = eager_tstep
                for _ in range(ks):
                    tstep()
                torch.cuda.synchronize()
                ms = (C.c_double * nk)()
                cnt = (C.c_int64 * nk)()
                _lib.check(lib.aether_profile_read(ms, cnt, nk), "aether_profile_read")
                lib.aether_profile_enable(0)
                train["kernels_us_per_step"] = {lib.aether_profile_kernel_name(k).decode(): 1e3 * ms[k] / ks
                                                for k in range(nk) if cnt[k]}
                model.dp_group = saved_group
            if world > 1:
                dist.barrier()
            return train
        try:
            train = train_leg()
        except Exception as ex:
            print("training leg failed:", repr(ex), file=sys.stderr)
            train = {"error": repr(ex)}
            train_failed_here = 1

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        value = 4.0 * E * args.chunks * world / (dt / args.steps)
        step_flops = (E * FLOP_PER_EDGE_STEP[D] + Nn * FLOP_PER_NODE_STEP[D]) * args.chunks
        other = None
        if (B, N, D) == (32, 1024, 2):
            # HBM and pipe fractions of the OTHER heavy kernels of the config-5 shard, from the committed PMC
            # passes (profiles/traffic.json -> cfg5shard.r03_final) over the launch times measured in THIS run
            try:
                pm5all = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))["cfg5shard"]
                pm5 = pm5all.get("r04_final") or pm5all["r03_final"]       # (counts of the library as committed)

                def fracs(c, t_us, launches=1):
                    t = t_us * 1e-6
                    by = (2.0 * c["fetch_kb"] + c["write_kb"]) * 1024.0 * launches
                    out = {"avg_launch_us": t_us / launches, "hbm_bytes_per_launch": by / launches,
                           "hbm_frac": by / t / 8e12,
                           "bf16_matrix_pipe": c["mfma_bf16"] * launches * 16384.0 / t / (PEAK_BF16_MFMA_TFLOPS * 1e12),
                           "fp32_mfma_on_valu": c["mfma_f32"] * launches * 2048.0 / t / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                           "valu_issue": (c["valu"] * 4.0 + c["mfma_f32"] * 32.0) * launches / (1024.0 * t * 2.4e9)}
                    assert all(v <= 1.0 for k, v in out.items() if k not in ("avg_launch_us", "hbm_bytes_per_launch")), out
                    return out
                other = {"source": pm5["source"]}
                if kernels and "k_edge_layer1" in kernels:
                    other["k_edge_layer1"] = fracs(pm5["k_edge_layer1"], kernels["k_edge_layer1"]["avg_us"])
                if kernels and "k_edge_layer" in kernels:
                    other["k_edge_layer"] = fracs(pm5["k_edge_layer"], kernels["k_edge_layer"]["avg_us"])
                kb = (train or {}).get("kernels_us_per_step", {}).get("kb_edge")
                if kb and "kb_edge_acc_false" in pm5:      # four launches per step: layers 4, 3, 2 (<false>) and layer 1 (<true>), timed together
                    both = {k: 3.0 * pm5["kb_edge_acc_false"][k] + pm5["kb_edge_acc_true"][k] for k in pm5["kb_edge_acc_false"]}
                    other[pm5.get("kb_edge_acc_kernel", "kb_edge_acc") + " (4 launches)"] = fracs({k: v / 4.0 for k, v in both.items()}, kb, launches=4)
            except Exception as ex:
                other = {"error": repr(ex)}
        line = {
            "metric": ("edge-messages/sec (forward, electrostatic N=20 batch=128 per GPU)" if (B, N, D) == (128, 20, 2)
                       else f"edge-messages/sec (forward, {D}-D N={N} batch={B} per GPU)"),
            "value": value, "unit": "edge-messages/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "pre_warm_ms": args.prewarm_ms, "ms_per_step": ms_step,
            "ms_per_step_repeats": ({"n": len(rep_ms), "median": sorted(rep_ms)[len(rep_ms) // 2], "min": min(rep_ms),
                                     "max": max(rep_ms), "note": "rank 0, the same K steps timed again after the contract's region"}
                                    if rep_ms else None),
            "higher_is_better": True,
            "scaling": "strong" if args.config == "cfg5" else "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": ("fp32 in, fp32 out, fp32 accumulate; the 64 x 64 contractions of the fused kernel run as three fp16 "
                           "matrix-core terms on operands split into two fp16 pieces (22 significand bits, exact power-of-two "
                           "rescale outside 2^-6..2^15 per wave and GEMM; parity bar 1e-5, measured 4e-8 against the fp64 oracle)"),
            "data": "synthetic",
            "config": {"workload": WORKLOAD["name"] if (B, N, D) == (128, 20, 2) else f"D{D}-N{N}-B{B}",
                       "num_dims": D, "nodes_per_graph": N, "graphs_per_gpu": B * args.chunks, "edges_per_gpu": E * args.chunks,
                       "chunks_per_step": args.chunks,
                       "hidden": 64, "launch": (f"hipgraph ({S} steps per replay)" if use_graph else "eager"),
                       "parallelism": f"graphs sharded over {world} rank(s), no forward collective"},
            "edges_per_s": E * args.chunks * world / (dt / args.steps),
            "step_algorithmic_tflops": step_flops * world / (dt / args.steps) / 1e12,
            "roofline": roof, "roofline_other": other, "kernels": kernels, "rollout": roll, "train": train, "ranks_seen": ranks_seen,
        }
        if world == 1 and (B, N, D) == (128, 20, 2) and not args.no_other_configs and not args.streamed:
            del out                                    # the headline's buffers are no longer needed
            line["other_configs"] = _other_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            sd_tr = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if train is not None else None
            line["cpu_baseline"] = _cpu_baseline(sd0, host, model, dev, sd_trained=sd_tr)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    # a leg that failed is in the line as {"error": ...}; the run still says so with its exit code -- after the
    # headline line is out.  With N > 1 every rank learns of a failure anywhere before leaving (nobody waits in a
    # collective for a rank that raised).
    failed = 0
    if rank == 0:
        legs = [roll, train] + (line.get("other_configs") or {}).get("legs", [])
        failed = int(any(isinstance(x, dict) and "error" in x for x in legs))
    if world > 1:
        f = torch.tensor([float(failed or train_failed_here)], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        failed = int(f.item() > 0)
        dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
