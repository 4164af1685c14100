"""The forward-prediction metric of the seq2seq runners on the device (metric 2).

``eval_forward_prediction`` follows experiments/electrostatic/evaluate.py:14-79: the model sees the first
``burn_in_steps`` frames of every trajectory, predicts ``forward_pred_steps`` more with ``predict_future``, and the
per-step mean squared error is taken over (particle, feature) on UN-normalised values -- total, positions only and
velocities only -- averaged over the data set.  Inputs stay in HBM (no DataLoader, no host copies).
"""
from __future__ import annotations

import torch


@torch.no_grad()
def eval_forward_prediction(model, dataset, burn_in_steps, forward_pred_steps, batch_size=1000, num_dims=None,
                            return_total_errors=False, **predict_kwargs):
    """-> (mse, pos_mse, vel_mse), each [forward_pred_steps] (or the per-sample errors with ``return_total_errors``)."""
    model.eval()
    D = num_dims if num_dims is not None else getattr(dataset, "ndim", 2)
    feats = dataset.feats
    tot, pos, vel = [], [], []
    for lo in range(0, len(dataset), batch_size):
        inputs = feats[lo:lo + batch_size]
        preds = model.predict_future(inputs[:, :burn_in_steps], forward_pred_steps, **predict_kwargs)
        gt = inputs[:, burn_in_steps:burn_in_steps + forward_pred_steps]
        p, g = dataset.torch_unnormalize(preds), dataset.torch_unnormalize(gt)
        se = (p - g) ** 2
        tot.append(se.flatten(2).mean(-1))                                   # [B, steps]
        pos.append(se[..., :D].flatten(2).mean(-1))
        vel.append(se[..., D:].flatten(2).mean(-1))
    tot, pos, vel = torch.cat(tot), torch.cat(pos), torch.cat(vel)
    if return_total_errors:
        return tot, pos, vel
    return tot.mean(0), pos.mean(0), vel.mean(0)
