"""MI355X autoregressive prediction path of the seq2seq Aether.

Mirrors the parts of ``nn.seq2seq.aether.Aether`` (aether.py:14-191) that run once the burn-in is over:
``predict_field`` (:86-90), ``single_step_forward`` (:92-101) and the prediction loop of
``predict_future`` (:175-185).  Sub-modules carry the reference's names -- ``encoder``, ``decoder``,
``field_net``, ``coordinate_embedding`` -- so ``load_state_dict(reference_model.state_dict())`` works.
``predict_future`` runs its burn-in half with the same step (the prior path of the encoder is causal);
the posterior encoder (reverse LSTM, ``encoder_fc_out``) and the training loss are not part of this path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import _lib
from .decoder import RecurrentDecoder
from .encoder import Encoder, gumbel_softmax_hard
from .field import FieldQuery


class Aether(nn.Module):
    def __init__(self, params, device="cuda"):
        super().__init__()
        self.num_vars = params["num_vars"]
        self.encoder = Encoder(params, device=None)                       # creation order of aether.py:19-24,71-84
        if params.get("decoder_type", None) == "ref_mlp":
            raise ValueError("decoder_type 'ref_mlp' (MarkovDecoder) is not part of this path")
        self.decoder = RecurrentDecoder(params, device=None)
        self.num_edge_types = params.get("num_edge_types")
        self.gumbel_temp = params.get("gumbel_temp")
        self.use_3d = params.get("use_3d", False)
        self.num_dims = 3 if self.use_3d else 2
        fq = FieldQuery(self.num_dims, params["encoder_hidden"], params.get("rff_std", 1.0), device=None)
        self.field_net, self.coordinate_embedding = fq.field_net, fq.coordinate_embedding
        self._fq = [fq]                                                    # not a registered sub-module: no duplicate keys
        if device is not None:
            self.to(device)

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
    def predict_future(self, inputs, prediction_steps, return_edges=False, uniform=None):
        """aether.py:155-191.  inputs [B, T, N, 2D] (burn-in observations).  The burn-in half runs the prior
        step by step: the encoder's prior path is causal (forward LSTM from the zero state, BatchNorm in eval
        mode), so the chained ``single_step_forward`` equals ``Encoder.forward``'s prior logits and state
        (pinned against the reference's own ``predict_future``).  ``uniform`` [T - 1 + steps, B, E, K]."""
        B, T, N, _ = inputs.shape
        E = N * (N - 1)
        decoder_hidden = self.decoder.get_initial_hidden(inputs)
        R = self.encoder.rnn_hidden_size
        prior_hidden = (torch.zeros(B, E, R, device=inputs.device), torch.zeros(B, E, R, device=inputs.device))
        for step in range(T - 1):
            current_inputs = inputs[:, step]
            field, _ = self.predict_field(current_inputs)
            logits, prior_hidden = self.encoder.single_step_forward(current_inputs, prior_hidden, field)
            _, decoder_hidden, _ = self.single_step_forward(current_inputs, decoder_hidden, logits, True, field,
                                                            None if uniform is None else uniform[step])
        return self.predict_from_state(inputs[:, T - 1], decoder_hidden, prior_hidden, prediction_steps,
                                       None if uniform is None else uniform[T - 1:], return_edges)

    @torch.no_grad()
    def predict_from_state(self, predictions, decoder_hidden, prior_hidden, prediction_steps, uniform=None,
                           return_edges=False):
        """The prediction loop of ``predict_future`` (aether.py:175-185), starting from the state the burn-in
        leaves behind: last observed state ``predictions`` [B, N, 2D], ``decoder_hidden`` [B, N, h],
        ``prior_hidden`` = (h, c) each [B, E, rnn].  ``uniform`` [steps, B, E, K] fixes the Gumbel draws."""
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
