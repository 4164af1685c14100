"""Whole training step of the state2state model as one hipGraph replay.

The runner's loop (experiments/lorentz/main.py:200-260) issues, per batch, the forward, ``loss.backward()`` and the
optimizer step as ~40 separate launches; on the MI355X the step is then bound by launch gaps (0.6 ms eager against
0.4 ms of kernels at N=20, batch=128).  ``GraphedTrainStep`` captures forward + HIP backward + a capturable fused
AdamW once (``torch.cuda.CUDAGraph``) on static input buffers and replays it: same arithmetic, one launch per step.
Shapes and the edge index are fixed at construction (the runner's batches have a fixed shape, main.py:211-212).

Data-parallel (``aether_amd.parallel.attach_data_parallel`` was called on the model): forward + backward replay as one
graph, the flat gradient buffer is SUM all-reduced eagerly (RCCL) on the same stream, the optimizer replays as a second
graph and takes the mean as it reads the gradients (``FusedAdamW.grad_scale`` = 1 / world): two graph launches + one
collective, no other launch.  ``graph_collective=True`` asks for the whole step -- collective included -- as ONE graph
(RCCL collectives can be captured); if that capture fails the two-graph form is used.  It is off by default: no
multi-GPU hardware was available to test it.
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model, example_args, example_target, lr=5e-4, weight_decay=1e-12, loss_fn=None, warmup=3,
                 optimizer="aether", graph_collective=False):
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
        # data-parallel: the step owns the collective from here on (the module's backward no longer issues it);
        # close() hands it back
        self.dp_group = getattr(model, "dp_group", None)
        self.world = 1
        if self.dp_group is not None:
            import torch.distributed as dist
            if not hasattr(model, "_grad_buffers"):
                raise ValueError("data-parallel GraphedTrainStep needs a model with one flat gradient buffer (Aether)")
            model.dp_group = None
            self.world = dist.get_world_size(self.dp_group)
            # the mean over ranks: folded into the optimizer's gradient read where the optimizer can do it
            self._scale_in_optimizer = hasattr(self.optimizer, "grad_scale")
            if self._scale_in_optimizer:
                self.optimizer.grad_scale = 1.0 / self.world
        self.allreduce_events = None
        self.collective_in_graph = False
        self.flat = None
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
            self.opt_graph = None
            if graph_collective:
                try:
                    with torch.cuda.graph(self.graph):
                        self.loss = self._forward_backward()
                        self._allreduce()
                        self.optimizer.step()
                    self.collective_in_graph = True
                except Exception as ex:                          # backend cannot be captured: two graphs around an eager collective
                    import sys
                    print("GraphedTrainStep: collective not capturable, using two graphs:", repr(ex), file=sys.stderr)
                    torch.cuda.synchronize()
                    self.graph = torch.cuda.CUDAGraph()
                    self.optimizer.zero_grad(set_to_none=True)
            if not self.collective_in_graph:
                with torch.cuda.graph(self.graph):
                    self.loss = self._forward_backward()
                self.opt_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.opt_graph):
                    self.optimizer.step()
            self.flat = self._flat_gradient_buffer()

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

    def _flat_gradient_buffer(self):
        """The model's flat gradient buffer if -- judged from the gradients themselves -- every ``.grad`` the optimizer
        reads is a view into it (``grad_as_view``); None otherwise (narrow models, ``grad_as_view = False``: ordinary
        gradient tensors, flattened around the collective)."""
        bufs = self.model._grad_buffers() if hasattr(self.model, "_grad_buffers") else None
        if not bufs or bufs[0] is None:
            return None
        flat = bufs[0]
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        grads = [p.grad for p in self._params if p.grad is not None]
        if not grads or not all(lo <= g.data_ptr() and g.data_ptr() + g.numel() * g.element_size() <= hi for g in grads):
            return None
        return flat

    def _allreduce(self):
        import torch.distributed as dist
        flat = self.flat if self.flat is not None else self._flat_gradient_buffer()
        scale_here = not getattr(self, "_scale_in_optimizer", False)
        if flat is None:
            grads = [p.grad for p in self._params if p.grad is not None]
            cat = torch.cat([g.reshape(-1) for g in grads])
            dist.all_reduce(cat, group=self.dp_group)
            if scale_here:
                cat.div_(self.world)
            torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(cat.split([g.numel() for g in grads]), grads)])
            return
        dist.all_reduce(flat, group=self.dp_group)
        if scale_here:
            flat.div_(self.world)

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
        if self.dp_group is not None and not self.collective_in_graph:
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

    def check(self):
        """Synchronise and raise if a kernel of the replays reported an asynchronous error (the bounded wait of the fused
        kernels' cross-workgroup hand-off sets a host-mapped word: a graph replay never passes through a C entry point
        that would notice it)."""
        from . import _lib
        torch.cuda.synchronize()
        _lib.check(_lib.load().aether_check_async_error(), "asynchronous kernel error during graph replays")

    def close(self):
        """Give the collective back to the module's own backward (eager data-parallel use after this step object)."""
        if self.dp_group is not None:
            self.model.dp_group = self.dp_group
            if getattr(self, "_scale_in_optimizer", False):
                self.optimizer.grad_scale = 1.0
