"""Whole training step of the state2state model as one hipGraph replay.

The runner's loop (experiments/lorentz/main.py:200-260) issues, per batch, the forward, ``loss.backward()`` and the
optimizer step as ~40 separate launches; on the MI355X the step is then bound by launch gaps (0.6 ms eager against
0.4 ms of kernels at N=20, batch=128).  ``GraphedTrainStep`` captures forward + HIP backward + a capturable fused
AdamW once (``torch.cuda.CUDAGraph``) on static input buffers and replays it: same arithmetic, one launch per step.
Shapes and the edge index are fixed at construction (the runner's batches have a fixed shape, main.py:211-212).
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model, example_args, example_target, lr=5e-4, weight_decay=1e-12, loss_fn=None, warmup=3):
        """``example_args``: the positional arguments of ``model.forward`` for one batch (tensors are cloned into static
        buffers; the edge index list and non-tensors are kept as they are), ``example_target``: the batch's target."""
        dev = example_target.device
        if dev.type != "cuda":
            raise ValueError("GraphedTrainStep needs CUDA/HIP tensors")
        self.model = model
        self.loss_fn = loss_fn or torch.nn.functional.mse_loss
        self.args = [a.clone() if isinstance(a, torch.Tensor) else a for a in example_args]
        self.target = example_target.clone()
        self.optimizer = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=True, fused=True)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                         # warm-up off the capture: lazy initialisation, workspaces
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = self._forward_backward()
            self.optimizer.step()

    def _forward_backward(self):
        out = self.model(*self.args)
        loss = self.loss_fn(out, self.target)
        loss.backward()
        return loss

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self._forward_backward()
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
        self.graph.replay()
        return self.loss
