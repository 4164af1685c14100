"""CPU restatement of the seq2seq Aether's field query.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under aether_amd/).

* ``fourier_features``  <- nn/nn/fourier_feature_mapper.py:7-21 (``FourierFeatureMapper``):
  ``x_proj = (2*pi*x) @ B``; ``cat[sin(x_proj), cos(x_proj)]``; ``B [D, h/2]`` is drawn once from
  ``numpy.random.default_rng(42).normal(0, std)`` and cast to fp32 (buffer ``coordinate_embedding.B``).
* ``predict_field``     <- nn/seq2seq/aether.py:72-78,86-90: ``coords = x[..., :D]``, then
  ``Linear(h,h) - SiLU - Linear(h,h) - SiLU - Linear(h,D)`` on the features.  Unlike the
  state2state field (A1) it sees positions only.

Parity status: PINNED by tests/golden/s2s_field_D{2,3}.npz, produced by oracle/make_golden_seq2seq.py
from the imported reference ``FourierFeatureMapper`` and a ``field_net`` built exactly as the
reference builds it (tests/test_oracle_golden.py::test_seq2seq_field).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def rff_matrix(num_dims: int, half: int, std: float = 1.0) -> torch.Tensor:
    """The buffer ``coordinate_embedding.B`` (fourier_feature_mapper.py:12-15)."""
    rng = np.random.default_rng(42)
    return torch.from_numpy(rng.normal(0, std, size=(num_dims, half))).float()


def fourier_features(x: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    x_proj = (2 * math.pi * x) @ B.to(x.dtype)                 # fourier_feature_mapper.py:19-20
    return torch.cat([torch.sin(x_proj), torch.cos(x_proj)], dim=-1)


def predict_field(sd, x: torch.Tensor, num_dims: int) -> torch.Tensor:
    """``sd``: 'coordinate_embedding.B', 'field_net.{0,2,4}.{weight,bias}' (aether.py:86-90)."""
    coords = x[..., :num_dims]
    h = fourier_features(coords, sd["coordinate_embedding.B"])
    h = F.silu(F.linear(h, sd["field_net.0.weight"], sd["field_net.0.bias"]))
    h = F.silu(F.linear(h, sd["field_net.2.weight"], sd["field_net.2.bias"]))
    return F.linear(h, sd["field_net.4.weight"], sd["field_net.4.bias"])
