"""Loss and optimizer of the runner's training step as one launch each (experiments/lorentz/main.py:86,164,289-292:
``nn.MSELoss`` and ``optim.AdamW``).  In a captured training step at N=20, batch=128 torch spends 7 launches (≈ 45 us of
300) on them; ``mse_loss_grad`` and ``FusedAdamW`` spend two.  HIP only: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class _MseScratch:
    """Per-device scratch of aether_mse_loss_grad (partial sums + counter, zero before the first launch)."""
    _bufs = {}

    @classmethod
    def get(cls, device):
        key = (device.type, device.index)
        buf = cls._bufs.get(key)
        if buf is None:
            buf = cls._bufs[key] = torch.zeros(_lib.load().aether_mse_scratch_bytes(), dtype=torch.uint8, device=device)
        return buf


def mse_loss_grad(pred, target):
    """``loss = mean((pred - target)**2)`` and ``dloss/dpred`` in one launch: ``(loss [scalar tensor], grad [like pred])``.
    ``pred.backward(grad)`` then equals ``F.mse_loss(pred, target).backward()``."""
    if not pred.is_cuda:
        raise _lib.AetherHipError("aether_amd.optim.mse_loss_grad runs on an MI355X only; got a CPU tensor")
    if pred.shape != target.shape:
        raise ValueError("pred and target must have the same shape")
    p = pred.detach().to(torch.float32).contiguous()
    t = target.detach().to(device=p.device, dtype=torch.float32).contiguous()
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    grad = torch.empty_like(p)
    scratch = _MseScratch.get(p.device)
    lib = _lib.load()
    st = lib.aether_mse_loss_grad(p.data_ptr(), t.data_ptr(), p.numel(), loss.data_ptr(), grad.data_ptr(),
                                  scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream(p.device).cuda_stream)
    _lib.check(st, "aether_mse_loss_grad")
    return loss, grad


class FusedAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` (no amsgrad, no maximize) with every parameter tensor updated by ONE launch
    (``aether_adamw_step``).  The step counter and the learning rate live in device memory, so a hipGraph that captured
    ``step()`` keeps counting and follows ``param_groups[i]["lr"]`` (call ``sync_lr()`` -- ``step()`` does -- after a
    scheduler changed it; ``GraphedTrainStep.step`` does so before every replay).
    An eager ``step()`` bumps the parameters' version counters (the kernel writes them behind autograd's back and the
    modules key prepared weight images on the versions); a captured one cannot -- ``GraphedTrainStep`` does it per replay.
    ``grad_scale`` (attribute, default 1.0) multiplies every gradient as the kernel reads it: a data-parallel step sets
    1 / world_size and all-reduces with SUM, so the mean costs no launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 < betas[0] < 1 and 0 < betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdamW: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self.grad_scale = 1.0

    def _group_state(self, gi, group):
        hit = self._tables.get(gi)
        # same gradient tensor objects (kept alive here, so identity is exact) and parameter addresses as last time: the
        # table stands (a module with one flat gradient buffer hands out the same views every step)
        if hit is not None and all(p.grad is g and p.data_ptr() == d
                                   for p, g, d in zip(group["params"], hit["grads_all"], hit["ptrs_all"])):
            return hit
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return None
        dev = params[0].device
        for p in params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                    and p.grad.dtype == torch.float32 and p.device == dev):
                raise _lib.AetherHipError("FusedAdamW: contiguous fp32 parameters and gradients on one MI355X only")
            s = self.state[p]
            if "exp_avg" not in s:
                s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        # one step counter per group, kept under torch.optim.AdamW's state key ("step", a device float) of every parameter:
        # the same tensor object, so state_dict()s are interchangeable with torch's capturable AdamW
        shared = self.state[params[0]].get("step")
        if shared is None or not torch.is_tensor(shared) or shared.device != dev:
            shared = torch.full((), float(shared) if shared is not None else 0.0, dtype=torch.float32, device=dev)
        shared = shared.to(torch.float32).reshape(()) if shared.dtype != torch.float32 or shared.ndim else shared
        for p in params:
            self.state[p]["step"] = shared
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr()) for p in params) + (shared.data_ptr(),)
        if hit is not None and hit["key"] == key:
            hit["grads_all"], hit["ptrs_all"] = [p.grad for p in group["params"]], [p.data_ptr() for p in group["params"]]
        if hit is None or hit["key"] != key:
            arr = (_lib.AetherAdamWTensor * len(params))()
            for a, p in zip(arr, params):
                s = self.state[p]
                a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.numel = (p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(),
                                                                     s["exp_avg_sq"].data_ptr(), p.numel())
            old = hit or {}
            hit = self._tables[gi] = dict(
                key=key, arr=arr, n=len(params), step=shared, params=params,
                grads_all=[p.grad for p in group["params"]], ptrs_all=[p.data_ptr() for p in group["params"]],
                lr=old.get("lr", torch.full((), float(group["lr"]), dtype=torch.float32, device=dev)),
                lr_host=old.get("lr_host", float(group["lr"])),
                counter=old.get("counter", torch.zeros((), dtype=torch.int32, device=dev)))
        return hit

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}

    def sync_lr(self):
        """Copy a changed ``param_groups[i]["lr"]`` into the device scalar the kernel reads (not while capturing)."""
        for gi, group in enumerate(self.param_groups):
            hit = self._tables.get(gi)
            if hit is not None and hit["lr_host"] != float(group["lr"]):
                hit["lr"].fill_(float(group["lr"]))
                hit["lr_host"] = float(group["lr"])

    def steps_taken(self, group=0):
        hit = self._tables.get(group)
        return 0 if hit is None else int(hit["step"].item())

    def reset_state(self):
        """Moments and step counters back to zero (in place: captured graphs keep pointing at the same buffers)."""
        for s in self.state.values():
            for v in s.values():
                if torch.is_tensor(v):
                    v.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            hit = self._group_state(gi, group)          # (raises for CPU tensors before anything touches the device)
            if hit is None:
                continue
            if hit["lr_host"] != float(group["lr"]) and not torch.cuda.is_current_stream_capturing():
                hit["lr"].fill_(float(group["lr"]))
                hit["lr_host"] = float(group["lr"])
            b1, b2 = group["betas"]
            dev = hit["step"].device
            st = lib.aether_adamw_step(C.cast(hit["arr"], C.c_void_p), hit["n"], hit["step"].data_ptr(), hit["lr"].data_ptr(),
                                       hit["counter"].data_ptr(), float(b1), float(b2), float(group["eps"]),
                                       float(group["weight_decay"]), float(self.grad_scale),
                                       torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(st, "aether_adamw_step")
            # The kernel writes the parameters behind autograd's back.  Everything keyed on (data_ptr, _version) -- the
            # modules' prepared weight images, plans, engine copies -- must see a new version (an eval forward
            # between backward and step() otherwise left stale images marked fresh).  Under capture the bump would be a
            # host-side no-op per replay: GraphedTrainStep bumps after every replay itself.
            if not torch.cuda.is_current_stream_capturing():
                torch.autograd.graph.increment_version(hit["params"])
        return loss
