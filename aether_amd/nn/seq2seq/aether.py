"""MI355X autoregressive prediction path of the seq2seq Aether.

Mirrors the parts of ``nn.seq2seq.aether.Aether`` (aether.py:14-191) that run once the burn-in is over:
``predict_field`` (:86-90), ``single_step_forward`` (:92-101) and the prediction loop of
``predict_future`` (:175-185).  Sub-modules carry the reference's names -- ``encoder``, ``decoder``,
``field_net``, ``coordinate_embedding`` -- so ``load_state_dict(reference_model.state_dict())`` works.
``predict_future`` runs its burn-in half with the same step (the prior path of the encoder is causal).
``calculate_loss(is_train=False)`` (:103-153) evaluates the validation loss with the full-sequence encoder
(``Encoder.forward``); training (``is_train=True``) is not part of this library.
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from ... import _lib
from .decoder import RecurrentDecoder
from .encoder import Encoder, gumbel_softmax_hard
from .field import FieldQuery


def _tensors_key(module):
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


class _StepRunner:
    """One autoregressive step -- field query -> prior step -> hard Gumbel sample -> decoder step (``aether_s2s_step``,
    one C call) -- captured once in a hipGraph (``torch.cuda.CUDAGraph``: every launch of the step is stream-ordered and free of host
    synchronisation) and replayed per time step on static state buffers.  The reference's configurations have
    5 objects per graph: the step is ~80 short kernels; a replay issues them with one launch.  Results are
    bit-identical to the eager step (same kernels).  Opt-in (``graph=True``): on the MI355X box the step is bound
    by the fixed latencies of those kernels, not by the host (1.52 ms eager vs 1.59 ms replayed at B=128, N=5,
    tools/s2s_dynfield_time.py), so it only pays where the host is slower than the GPU.
    """

    def __init__(self, model, field_fn, B, N, device):
        D, E, K = model.num_dims, N * (N - 1), model.num_edge_types
        R, h = model.encoder.rnn_hidden_size, model.decoder.msg_out_shape
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        self.x, self.dh, self.ph, self.pc = z(B, N, 2 * D), z(B, N, h), z(B, E, R), z(B, E, R)
        self.u = torch.full((B, E, K), 0.5, dtype=torch.float32, device=device)
        self.model, self.field_fn = model, field_fn
        self.edges = None
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                              # warm-up: lazy initialisation, workspaces, caches
            self._step()
            self._step()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._step()
        self.field_fn = None                                       # only needed for the capture
        self.keep = model._graph_keepalive()                       # buffers whose addresses the graph holds

    def _step(self):
        m = self.model
        field = None if self.field_fn is None else self.field_fn(self.x)         # None: the built-in field query
        pred, dh, (h1, c1), edges = m._fused_step(self.x, self.dh, (self.ph, self.pc), self.u, field)
        self.x.copy_(pred); self.dh.copy_(dh); self.ph.copy_(h1); self.pc.copy_(c1)
        self.edges = edges

    def load(self, x=None, decoder_hidden=None, prior_hidden=None):
        if x is not None:
            self.x.copy_(x)
        if decoder_hidden is not None:
            self.dh.copy_(decoder_hidden)
        if prior_hidden is not None:
            self.ph.copy_(prior_hidden[0]); self.pc.copy_(prior_hidden[1])

    def step(self, uniform):
        self.u.copy_(uniform)
        self.graph.replay()


class _RolloutRunner:
    """The whole loop of ``predict_future`` -- burn-in and prediction steps, ``aether_s2s_rollout`` -- captured once in a
    hipGraph on static buffers: one graph launch per rollout (T0 + steps steps of 31 kernels each).  Results are
    bit-identical to the eager call (same kernels)."""

    def __init__(self, model, B, N, T0, steps, device):
        D, E, K = model.num_dims, N * (N - 1), model.num_edge_types
        R, h = model.encoder.rnn_hidden_size, model.decoder.msg_out_shape
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        self.burn = z(B, T0, N, 2 * D) if T0 > 0 else None
        self.x, self.dh, self.ph, self.pc = z(B, N, 2 * D), z(B, N, h), z(B, E, R), z(B, E, R)
        self.u = torch.full((T0 + steps, B, E, K), 0.5, dtype=torch.float32, device=device)
        self.model, self.steps = model, steps
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                              # warm-up: plan, workspaces, lazy initialisation
            self._run()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        self.keep = model._graph_keepalive()

    def _run(self):
        return self.model._fused_rollout(self.burn, self.x, self.dh, (self.ph, self.pc), self.steps, self.u, True)

    def __call__(self, burn_in, x_last, decoder_hidden, prior_hidden, uniform):
        if self.burn is not None:
            self.burn.copy_(burn_in)
        self.x.copy_(x_last); self.dh.copy_(decoder_hidden)
        self.ph.copy_(prior_hidden[0]); self.pc.copy_(prior_hidden[1])
        self.u.copy_(uniform.reshape(self.u.shape))
        self.graph.replay()
        preds, edges, (dh, (h, c)) = self.out
        return preds.clone(), edges.clone(), (dh.clone(), (h.clone(), c.clone()))


class _StepLoop:
    """Mixin of the two seq2seq models: the fused autoregressive step (``aether_s2s_step`` / ``aether_s2s_rollout`` on a
    cached plan of prepared weights) and the burn-in / prediction loops on a cached ``_StepRunner``."""

    def _step_sizes(self):
        enc, dec = self.encoder, self.decoder
        return (self.num_dims, enc.hidden_size, dec.msg_out_shape, enc.rnn_hidden_size, self.num_edge_types)

    def _plan(self, device):
        """Prepared weights of the fused step (``aether_s2s_plan_build``), rebuilt -- into the same buffer, which captured
        graphs point at -- whenever an encoder / decoder tensor moved or was written to."""
        enc, dec = self.encoder, self.decoder
        fq = getattr(self, "_fq", None)
        tensors = list(enc.parameters()) + list(enc.buffers()) + list(dec.parameters())
        if fq is not None:
            tensors += list(fq[0].field_net.parameters())
        key = (str(device),) + tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self.__dict__.get("_plan_cache")
        if hit is None or hit[0] != key:
            lib = _lib.load()
            D, he, hd, R, K = self._step_sizes()
            pe, n_layers, prior_hidden = enc._param_struct(with_image=False)
            nbytes = lib.aether_s2s_plan_bytes(D, he, hd, R, n_layers, prior_hidden, K)
            if nbytes == 0:
                raise _lib.AetherHipError("fused seq2seq step: encoder_hidden must be a multiple of 128, decoder_hidden of 32")
            buf = hit[1] if hit is not None and hit[1].numel() == nbytes and hit[1].device == torch.device(device) else \
                torch.empty(nbytes, dtype=torch.uint8, device=device)
            pd = dec._param_struct()
            pf = self._field_struct()
            _lib.check(lib.aether_s2s_plan_build(None if pf is None else C.byref(pf), C.byref(pe), C.byref(pd), D, he, hd, R, n_layers,
                                                 prior_hidden, K, buf.data_ptr(), nbytes,
                                                 torch.cuda.current_stream(device).cuda_stream), "aether_s2s_plan_build")
            hit = self.__dict__["_plan_cache"] = (key, buf)
        return hit[1]

    def _field_struct(self):
        """Parameter struct of the built-in field query, or None (the dynamic-field model hands its field in)."""
        fq = getattr(self, "_fq", None)
        if fq is None:
            return None
        fn, ce = fq[0].field_net, fq[0].coordinate_embedding
        from .field import _S2SFieldParams
        return _S2SFieldParams(*[t.data_ptr() for t in (ce.B, fn[0].weight, fn[0].bias, fn[2].weight, fn[2].bias,
                                                        fn[4].weight, fn[4].bias)])

    def _step_common(self, B, N, device):
        """(plan, workspace, graph arrays, parameter structs, scalar arguments) of the fused step for B graphs of N objects."""
        if self.encoder.training:
            raise _lib.AetherHipError("the prior step uses BatchNorm running statistics: call .eval() first")
        lib = _lib.load()
        enc, dec = self.encoder, self.decoder
        D, he, hd, R, K = self._step_sizes()
        E1 = enc.recv_edges.shape[0]
        plan = self._plan(device)
        pe, n_layers, prior_hidden = enc._param_struct(with_image=False)
        pd = dec._param_struct()
        need = lib.aether_s2s_step_workspace_bytes(D, he, hd, R, prior_hidden, K, B * N, B * E1)
        ws = self.__dict__.get("_step_ws")
        if ws is None or ws.numel() < need or ws.device != torch.device(device):
            ws = self.__dict__["_step_ws"] = torch.empty(need, dtype=torch.uint8, device=device)
        pf = self._field_struct()
        scal = (D, he, hd, R, n_layers, prior_hidden, K, 1 if dec.skip_first_edge_type else 0,
                1 if enc.pos_representation == "polar" else 0, N, float(self.gumbel_temp), B * N, B * E1)
        return lib, plan, ws, enc._graph(B, N, device), (pf, pe, pd), scal

    @torch.no_grad()
    def _fused_step(self, x, decoder_hidden, prior_hidden, uniform, field=None):
        """One autoregressive step: x [B, N, 2D], decoder_hidden [B, N, hd], prior_hidden (h, c) [B, E, rnn], uniform
        [B, E, K]; ``field`` [B, N, D] replaces the built-in field query -> (predictions, decoder_hidden, (h, c), edges)."""
        if not x.is_cuda:
            raise _lib.AetherHipError("aether_amd seq2seq models run on an MI355X only; got a CPU tensor (there is no CPU fallback)")
        B, N, _ = x.shape
        dev = x.device
        lib, plan, ws, (send, recv, order, rowptr), (pf, pe, pd), scal = self._step_common(B, N, dev)
        D, he, hd, R, K = self._step_sizes()
        E1 = self.encoder.recv_edges.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        xf, dhf, h0, c0, uf = f32(x), f32(decoder_hidden), f32(prior_hidden[0]), f32(prior_hidden[1]), f32(uniform)
        if xf.shape != (B, N, 2 * D) or dhf.shape != (B, N, hd) or h0.shape != (B, E1, R) or c0.shape != h0.shape or \
                uf.numel() != B * E1 * K:
            raise ValueError("fused step: input shapes do not match the model")
        ff = None if field is None else f32(field)
        if ff is None and pf is None:
            raise _lib.AetherHipError("this model has no built-in field query: pass the field")
        out = torch.empty_like(xf)
        dh_out = torch.empty_like(dhf)
        h1, c1 = torch.empty_like(h0), torch.empty_like(c0)
        edges = torch.empty(B, E1, K, dtype=torch.float32, device=dev)
        st = lib.aether_s2s_step(None if pf is None else C.byref(pf), C.byref(pe), C.byref(pd), plan.data_ptr(), *scal,
                                 send.data_ptr(), recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(), xf.data_ptr(),
                                 None if ff is None else ff.data_ptr(), dhf.data_ptr(), h0.data_ptr(), c0.data_ptr(),
                                 uf.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(), dh_out.data_ptr(), h1.data_ptr(),
                                 c1.data_ptr(), edges.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_s2s_step")
        return out, dh_out, (h1, c1), edges

    @torch.no_grad()
    def _fused_rollout(self, burn_in, x_last, decoder_hidden, prior_hidden, steps, uniform, return_edges):
        """``aether_s2s_rollout``: burn_in [B, T0, N, 2D] (or None) teacher-forced, then ``steps`` autoregressive steps from
        x_last [B, N, 2D]; uniform [T0 + steps, B, E, K] -> (predictions [B, steps, N, 2D], edges or None, final state)."""
        B, N, _ = x_last.shape
        dev = x_last.device
        lib, plan, ws, (send, recv, order, rowptr), (pf, pe, pd), scal = self._step_common(B, N, dev)
        if pf is None:
            raise _lib.AetherHipError("this model has no built-in field query: step it with _fused_step")
        D, he, hd, R, K = self._step_sizes()
        E1 = self.encoder.recv_edges.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        T0 = 0 if burn_in is None else burn_in.shape[1]
        bi = None if T0 == 0 else f32(burn_in.transpose(0, 1))                       # [T0, B, N, 2D]
        xl = f32(x_last)
        dh = f32(decoder_hidden).clone()
        h, c = f32(prior_hidden[0]).clone(), f32(prior_hidden[1]).clone()
        if uniform is None:
            uniform = torch.rand(T0 + steps, B, E1, K, device=dev)
        uf = f32(uniform.reshape(T0 + steps, B, E1, K))
        preds = torch.empty(steps, B, N, 2 * D, dtype=torch.float32, device=dev)
        edges = torch.empty(steps, B, E1, K, dtype=torch.float32, device=dev) if return_edges else None
        st = lib.aether_s2s_rollout(C.byref(pf), C.byref(pe), C.byref(pd), plan.data_ptr(), *scal, send.data_ptr(),
                                    recv.data_ptr(), order.data_ptr(), rowptr.data_ptr(), T0,
                                    None if bi is None else bi.data_ptr(), int(steps), xl.data_ptr(), dh.data_ptr(),
                                    h.data_ptr(), c.data_ptr(), uf.data_ptr(), ws.data_ptr(), ws.numel(), preds.data_ptr(),
                                    None if edges is None else edges.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(st, "aether_s2s_rollout")
        return (preds.transpose(0, 1).contiguous(), None if edges is None else edges.transpose(0, 1).contiguous(),
                (dh, (h, c)))

    def _graph_keepalive(self):
        """Workspaces and index tensors the captured launches point at (a module may later replace its cached
        workspace by a larger one; the graph must keep the one it was captured with alive)."""
        keep = [list(m._cache.values()) for m in (self.encoder, self.decoder)]
        keep += [self.__dict__.get("_plan_cache"), self.__dict__.get("_step_ws")]
        fq = getattr(self, "_fq", None)
        if fq is not None:
            keep.append(fq[0]._ws)
        return keep

    def _runner(self, field_fn, B, N, device, extra_key=()):
        key = (B, N, str(device), _tensors_key(self)) + tuple(extra_key)
        hit = self.__dict__.setdefault("_runners", {})
        if key not in hit:
            hit.clear()                                            # parameters moved or shapes changed: drop old graphs
            hit[key] = _StepRunner(self, field_fn, B, N, device)
        return hit[key]

    def _graphed_rollout(self, burn_in, x_last, decoder_hidden, prior_hidden, steps, uniform, return_edges):
        """One hipGraph launch for the whole loop (``_RolloutRunner``); arguments as ``_fused_rollout``."""
        B, N = x_last.shape[0], x_last.shape[1]
        dev = x_last.device
        T0 = 0 if burn_in is None else burn_in.shape[1]
        E, K = N * (N - 1), self.num_edge_types
        if uniform is None:
            uniform = torch.rand(T0 + steps, B, E, K, device=dev)
        key = ("rollout", B, N, T0, int(steps), str(dev), _tensors_key(self))
        hit = self.__dict__.setdefault("_runners", {})
        if key not in hit:
            hit.clear()                                            # parameters moved or shapes changed: drop old graphs
            hit[key] = _RolloutRunner(self, B, N, T0, int(steps), dev)
        preds, edges, state = hit[key](burn_in, x_last, decoder_hidden, prior_hidden, uniform)
        return preds, (edges if return_edges else None), state

    def _graphed(self, field_fn, burn_in, x_last, decoder_hidden, prior_hidden, steps, uniform, return_edges,
                 extra_key=()):
        """burn_in: [B, T0, N, 2D] observations stepped through with teacher forcing (may be None), then ``steps``
        autoregressive steps from x_last.  uniform [T0 + steps, B, E, K] or None (drawn on the device)."""
        B, N = x_last.shape[0], x_last.shape[1]
        dev = x_last.device
        T0 = 0 if burn_in is None else burn_in.shape[1]
        E, K = N * (N - 1), self.num_edge_types
        if uniform is None:
            uniform = torch.rand(T0 + steps, B, E, K, device=dev)
        uniform = uniform.reshape(T0 + steps, B, E, K)
        run = self._runner(field_fn, B, N, dev, extra_key)
        run.load(decoder_hidden=decoder_hidden, prior_hidden=prior_hidden)
        for t in range(T0):
            run.load(x=burn_in[:, t])
            run.step(uniform[t])
        run.load(x=x_last)
        preds = torch.empty(B, steps, N, x_last.shape[-1], dtype=torch.float32, device=dev)
        edges = torch.empty(B, steps, E, K, dtype=torch.float32, device=dev) if return_edges else None
        for t in range(steps):
            run.step(uniform[T0 + t])
            preds[:, t].copy_(run.x)
            if return_edges:
                edges[:, t].copy_(run.edges)
        state = (run.dh.clone(), (run.ph.clone(), run.pc.clone()))
        return preds, edges, state


class _EvalLoss:
    """Mixin of the two seq2seq models: the loss configuration of the params dictionary (aether.py:27-58) and
    ``calculate_loss`` in evaluation mode (:103-153) with its NLL / KL terms (:186-236)."""

    def _init_loss_config(self, params):
        self.val_teacher_forcing_steps = params.get("val_teacher_forcing_steps", -1)
        self.normalize_kl = params.get("normalize_kl", False)
        self.normalize_kl_per_var = params.get("normalize_kl_per_var", False)
        self.normalize_nll = params.get("normalize_nll", False)
        self.normalize_nll_per_var = params.get("normalize_nll_per_var", False)
        self.nll_loss_type = params.get("nll_loss_type", "crossent")
        self.prior_variance = params.get("prior_variance")
        self.add_uniform_prior = params.get("add_uniform_prior")
        if self.add_uniform_prior:
            K = params["num_edge_types"]
            prior = torch.full((K,), 1.0 / K)
            if params.get("no_edge_prior") is not None:
                prior = torch.full((K,), (1 - params["no_edge_prior"]) / (K - 1))
                prior[0] = params["no_edge_prior"]
            self.log_prior = torch.log(prior).view(1, 1, K)

    def _sequence_field(self, inputs):
        """(field of inputs[:, :-1] as [B, N, T - 1, D], function giving the field of a later state [B, N, 2D])."""
        x = inputs[:, :-1].transpose(2, 1).contiguous()
        predicted_field, _ = self.predict_field(x)
        return predicted_field, lambda state: self.predict_field(state)[0]

    def calculate_loss(self, inputs, is_train=False, teacher_forcing=True, return_edges=False, return_logits=False,
                       use_prior_logits=False, uniform=None):
        """aether.py:103-153 in evaluation mode (the validation metrics of experiments/electrostatic/train.py): the
        full-sequence encoder's prior / posterior logits, the decoder stepped through the sequence with hard samples of
        the posterior (teacher forcing per ``val_teacher_forcing_steps``), negative log-likelihood and KL term as the
        params dictionary selects.  ``is_train=True`` (gradients, soft samples) is not part of this library.
        ``uniform`` [T - 1, B, E, K]: the Gumbel draws (drawn on the device when omitted)."""
        if is_train:
            raise _lib.AetherHipError("calculate_loss(is_train=True) (training of the seq2seq model) is not part of this "
                                      "library; evaluation (is_train=False), predict_future and predict_field are")
        with torch.no_grad():           # (not a decorator: evaluate.py:42-45 inspects this method's argument names)
            return self._calculate_loss_eval(inputs, teacher_forcing, return_edges, return_logits, use_prior_logits, uniform)

    def _calculate_loss_eval(self, inputs, teacher_forcing, return_edges, return_logits, use_prior_logits, uniform):
        B, T, N, _ = inputs.shape
        decoder_hidden = self.decoder.get_initial_hidden(inputs)
        predicted_field, field_fn = self._sequence_field(inputs)                     # [B, N, T - 1, D]
        prior_logits, posterior_logits, _ = self.encoder(inputs[:, :-1], predicted_field)
        tf_steps = self.val_teacher_forcing_steps
        all_predictions, edges, predictions = [], None, None
        for step in range(T - 1):
            if (teacher_forcing and (tf_steps == -1 or step < tf_steps)) or step == 0:
                current_inputs, current_field = inputs[:, step], predicted_field[:, :, step].contiguous()
            else:
                current_inputs = predictions
                current_field = field_fn(predictions)
            logits = (prior_logits if use_prior_logits else posterior_logits)[:, step].contiguous()
            predictions, decoder_hidden, edges = self.single_step_forward(
                current_inputs, decoder_hidden, logits, True, current_field,
                uniform=None if uniform is None else uniform[step])
            all_predictions.append(predictions)
        all_predictions = torch.stack(all_predictions, dim=1)
        target = inputs[:, 1:].to(torch.float32)
        loss_nll = self.nll(all_predictions, target)
        prob = torch.softmax(posterior_logits, dim=-1)
        loss_kl = self.kl_categorical_learned(prob, prior_logits)
        if self.add_uniform_prior:
            loss_kl = 0.5 * loss_kl + 0.5 * self.kl_categorical_avg(prob)
        loss = (loss_nll + self.kl_coef * loss_kl).mean()
        if return_edges:
            return loss, loss_nll, loss_kl, edges
        if return_logits:
            return loss, loss_nll, loss_kl, posterior_logits, all_predictions
        return loss, loss_nll, loss_kl

    # losses, aether.py:186-236 (scalar reductions: plain torch on the device)
    def nll(self, preds, target):
        if self.nll_loss_type == "crossent":
            e = nn.functional.binary_cross_entropy_with_logits(preds, target, reduction="none").view(preds.size(0), -1)
        elif self.nll_loss_type == "poisson":
            e = nn.functional.poisson_nll_loss(preds, target, reduction="none").view(preds.size(0), -1)
        elif self.nll_loss_type == "gaussian":
            neg_log_p = (preds - target) ** 2 / (2 * self.prior_variance)
            const = 0.5 * math.log(2 * math.pi * self.prior_variance)
            if self.normalize_nll_per_var:
                return neg_log_p.sum() / (target.size(0) * target.size(2))
            if self.normalize_nll:
                return (neg_log_p.sum(-1) + const).view(preds.size(0), -1).mean(dim=1)
            return neg_log_p.view(target.size(0), -1).sum() / target.size(1)
        else:
            raise ValueError("nll_loss_type must be 'crossent', 'gaussian' or 'poisson'")
        return e.mean(dim=1) if self.normalize_nll else e.sum(dim=1)

    def _kl_reduce(self, kl_div, batch):
        if self.normalize_kl:
            return kl_div.sum(-1).view(batch, -1).mean(dim=1)
        if self.normalize_kl_per_var:
            return kl_div.sum() / (self.num_vars * batch)
        return kl_div.view(batch, -1).sum(dim=1)

    def kl_categorical_learned(self, preds, prior_logits):
        kl_div = preds * (torch.log(preds + 1e-16) - torch.log_softmax(prior_logits, dim=-1))
        return self._kl_reduce(kl_div, preds.size(0))

    def kl_categorical_avg(self, preds, eps=1e-16):
        avg_preds = preds.mean(dim=2)
        kl_div = avg_preds * (torch.log(avg_preds + eps) - self.log_prior.to(preds.device))
        return self._kl_reduce(kl_div, preds.size(0))



class Aether(_StepLoop, _EvalLoss, nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        self.num_vars = params["num_vars"]
        self.encoder = Encoder(params, device=None)                       # creation order of aether.py:19-24,71-84
        if params.get("decoder_type", None) == "ref_mlp":
            raise ValueError("decoder_type 'ref_mlp' (MarkovDecoder) is not part of this path")
        self.decoder = RecurrentDecoder(params, device=None)
        self.num_edge_types = params.get("num_edge_types")
        self.gumbel_temp = params.get("gumbel_temp")
        self.kl_coef = params.get("kl_coef", 1.)                      # read by the training scripts
        self._init_loss_config(params)
        self.use_3d = params.get("use_3d", False)
        self.num_dims = 3 if self.use_3d else 2
        fq = FieldQuery(self.num_dims, params["encoder_hidden"], params.get("rff_std", 1.0), device=None)
        self.field_net, self.coordinate_embedding = fq.field_net, fq.coordinate_embedding
        self._fq = [fq]                                                    # not a registered sub-module: no duplicate keys
        if device is not None:
            self.to(device)

    def save(self, path):
        torch.save(self.state_dict(), path)                       # as the reference's save / load

    def load(self, path):
        self.load_state_dict(torch.load(path))

    def predict_field(self, x):
        return self._fq[0](x)

    @torch.no_grad()
    def single_step_forward(self, inputs, decoder_hidden, edge_logits, hard_sample, predicted_field, uniform=None):
        """aether.py:92-101.  ``uniform``: the U(0,1) draw of ``gumbel_softmax`` ([B, E, K]); the reference draws it
        with ``torch.rand`` on the host, here it defaults to a draw on the device."""
        if not hard_sample:
            raise _lib.AetherHipError("only hard_sample=True (evaluation / prediction) is part of this path")
        if uniform is None:
            uniform = torch.rand(edge_logits.shape, device=edge_logits.device)
        edges = gumbel_softmax_hard(edge_logits, uniform, self.gumbel_temp)
        predictions, decoder_hidden = self.decoder(inputs, decoder_hidden, edges, predicted_field)
        return predictions, decoder_hidden, edges

    @torch.no_grad()
    def predict_future(self, inputs, prediction_steps, return_edges=False, uniform=None, graph=False):
        """aether.py:155-191.  inputs [B, T, N, 2D] (burn-in observations); ``uniform`` [T - 1 + steps, B, E, K].
        ``graph``: replay the whole loop from ONE captured hipGraph (``_RolloutRunner``: a single graph launch per rollout)
        instead of launching it kernel by kernel; both ways run the same fused step and give identical results."""
        B, T, N, _ = inputs.shape
        E = N * (N - 1)
        decoder_hidden = self.decoder.get_initial_hidden(inputs)
        R = self.encoder.rnn_hidden_size
        prior_hidden = (torch.zeros(B, E, R, device=inputs.device), torch.zeros(B, E, R, device=inputs.device))
        if graph:
            preds, edges, _ = self._graphed_rollout(inputs[:, :T - 1].float() if T > 1 else None, inputs[:, T - 1].float(),
                                                    decoder_hidden, prior_hidden, int(prediction_steps), uniform, return_edges)
            return (preds, edges) if return_edges else preds
        # burn-in and prediction loop on the device (aether_s2s_rollout).  The reference takes the burn-in's prior logits
        # from the full-sequence encoder (aether.py:161-173); its prior path is causal (forward LSTM from the zero state,
        # BatchNorm in eval mode), so chaining the single step gives the same logits and state to rounding.
        preds, edges, _ = self._fused_rollout(inputs[:, :T - 1].float() if T > 1 else None, inputs[:, T - 1].float(),
                                              decoder_hidden, prior_hidden, int(prediction_steps), uniform, return_edges)
        return (preds, edges) if return_edges else preds

    @torch.no_grad()
    def predict_from_state(self, predictions, decoder_hidden, prior_hidden, prediction_steps, uniform=None,
                           return_edges=False, graph=False):
        """The prediction loop of ``predict_future`` (aether.py:175-185), starting from the state the burn-in
        leaves behind: last observed state ``predictions`` [B, N, 2D], ``decoder_hidden`` [B, N, h],
        ``prior_hidden`` = (h, c) each [B, E, rnn].  ``uniform`` [steps, B, E, K] fixes the Gumbel draws."""
        if graph:
            preds, edges, _ = self._graphed_rollout(None, predictions.float(), decoder_hidden, prior_hidden,
                                                    int(prediction_steps), uniform, return_edges)
            return (preds, edges) if return_edges else preds
        # the whole loop on the device (aether_s2s_rollout): one C call, the fused step per time step
        preds, edges, _ = self._fused_rollout(None, predictions, decoder_hidden, prior_hidden, int(prediction_steps),
                                              uniform, return_edges)
        return (preds, edges) if return_edges else preds

    @torch.no_grad()
    def predict_from_state_stepwise(self, predictions, decoder_hidden, prior_hidden, prediction_steps, uniform=None,
                                    return_edges=False):
        """The same loop through the four per-module entry points (``predict_field`` -> ``Encoder.single_step_forward`` ->
        ``single_step_forward``), as the reference writes it (aether.py:175-185); kept as the cross-check of the fused path."""
        all_predictions, all_edges = [], []
        for step in range(int(prediction_steps)):
            current_field, _ = self.predict_field(predictions)
            current_edge_logits, prior_hidden = self.encoder.single_step_forward(predictions, prior_hidden, current_field)
            predictions, decoder_hidden, edges = self.single_step_forward(
                predictions, decoder_hidden, current_edge_logits, True, current_field,
                None if uniform is None else uniform[step])
            all_predictions.append(predictions)
            all_edges.append(edges)
        predictions = torch.stack(all_predictions, dim=1)
        if return_edges:
            return predictions, torch.stack(all_edges, dim=1)
        return predictions
