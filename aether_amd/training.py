"""Whole training step of the state2state model as one hipGraph replay.

The runner's loop (experiments/lorentz/main.py:200-260) issues, per batch, the forward, ``loss.backward()`` and the
optimizer step as ~40 separate launches; on the MI355X the step is then bound by launch gaps (0.6 ms eager against
0.4 ms of kernels at N=20, batch=128).  ``GraphedTrainStep`` captures forward + HIP backward + a capturable fused
AdamW once (``torch.cuda.CUDAGraph``) on static input buffers and replays it: same arithmetic, one launch per step.
Shapes and the edge index are fixed at construction (the runner's batches have a fixed shape, main.py:211-212).

Data-parallel (``aether_amd.parallel.attach_data_parallel`` was called on the model): the collective stays outside the
graphs -- forward + backward replay as one graph, the flat gradient buffer is all-reduced eagerly (RCCL) on the same
stream, the optimizer replays as a second graph.  Every rank still pays two graph launches + one collective instead of
~45 eager launches.
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model, example_args, example_target, lr=5e-4, weight_decay=1e-12, loss_fn=None, warmup=3,
                 optimizer="aether"):
        """``example_args``: the positional arguments of ``model.forward`` for one batch (tensors are cloned into static
        buffers; the edge index list and non-tensors are kept as they are), ``example_target``: the batch's target.
        ``loss_fn`` None: the runner's ``nn.MSELoss`` (main.py:86), loss and the seed of the backward in one launch
        (``aether_amd.optim.mse_loss_grad``); any callable ``loss_fn(out, target)`` goes through autograd instead.
        ``optimizer``: "aether" (``aether_amd.optim.FusedAdamW``, one launch) or "torch" (capturable fused AdamW)."""
        dev = example_target.device
        if dev.type != "cuda":
            raise ValueError("GraphedTrainStep needs CUDA/HIP tensors")
        self.model = model
        self.loss_fn = loss_fn
        self.args = [a.clone() if isinstance(a, torch.Tensor) else a for a in example_args]
        self.target = example_target.clone()
        if optimizer == "aether":
            from .optim import FusedAdamW
            self.optimizer = FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
        elif optimizer == "torch":
            self.optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=True, fused=True)
        else:
            raise ValueError('optimizer: "aether" or "torch"')
        self._params = [p for p in model.parameters()]
        # data-parallel: the step owns the collective from here on (the module's backward no longer issues it)
        self.dp_group = getattr(model, "dp_group", None)
        if self.dp_group is not None:
            if not hasattr(model, "_grad_buffers"):
                raise ValueError("data-parallel GraphedTrainStep needs a model with one flat gradient buffer (Aether)")
            model.dp_group = None
        self.allreduce_events = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                         # warm-up off the capture: lazy initialisation, workspaces
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        if self.dp_group is None:
            self.opt_graph = None
            with torch.cuda.graph(self.graph):
                self.loss = self._forward_backward()
                self.optimizer.step()
        else:
            with torch.cuda.graph(self.graph):
                self.loss = self._forward_backward()
            # the .grad tensors are views of one flat buffer (grad_as_view); a narrow model (hidden_size < 64) hands out
            # ordinary gradients cut from its padded engine's: those are flattened around the collective
            self.flat = model._grad_buffers()[0] if self._grads_are_views() else None
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph):
                self.optimizer.step()

    def _forward_backward(self):
        out = self.model(*self.args)
        if self.loss_fn is None:
            from .optim import mse_loss_grad
            loss, grad = mse_loss_grad(out, self.target)
            out.backward(grad)
            return loss
        loss = self.loss_fn(out, self.target)
        loss.backward()
        return loss

    def _grads_are_views(self):
        return getattr(self.model, "hidden_size", 64) == 64

    def _allreduce(self):
        import torch.distributed as dist
        world = dist.get_world_size(self.dp_group)
        if not self._grads_are_views():
            grads = [p.grad for p in self._params if p.grad is not None]
            flat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(flat, group=self.dp_group)
            flat.div_(world)
            torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])
            return
        flat = getattr(self, "flat", None)
        if flat is None:
            flat = self.model._grad_buffers()[0]
        dist.all_reduce(flat, group=self.dp_group)
        flat.div_(world)

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self._forward_backward()
        if self.dp_group is not None:
            self._allreduce()
        self.optimizer.step()
        return loss

    def step(self, args=None, target=None):
        """Copy a new batch into the static buffers (tensors only, same shapes) and replay; returns the loss tensor of
        this step (a static buffer: read it before the next call)."""
        if args is not None:
            for dst, src in zip(self.args, args):
                if isinstance(dst, torch.Tensor):
                    dst.copy_(src)
        if target is not None:
            self.target.copy_(target)
        if hasattr(self.optimizer, "sync_lr"):
            self.optimizer.sync_lr()                  # a scheduler's new learning rate -> the device scalar the graph reads
        self.graph.replay()
        if self.dp_group is not None:
            if self.allreduce_events is not None:
                self.allreduce_events[0].record()
            self._allreduce()
            if self.allreduce_events is not None:
                self.allreduce_events[1].record()
            self.opt_graph.replay()
        # A replay rewrites the parameters without touching their version counters, which the modules use to decide whether
        # data derived from the weights (split weight images, the padded engine of a narrow model) is still current.
        torch.autograd.graph.increment_version(self._params)
        return self.loss

    def time_allreduce(self, on=True):
        """Bracket the collective of the following steps with a pair of events (``allreduce_events``)."""
        self.allreduce_events = ([torch.cuda.Event(enable_timing=True) for _ in range(2)] if on else None)
