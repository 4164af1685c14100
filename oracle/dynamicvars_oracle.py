"""CPU restatement of the variable-N decoder step (second half).

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by nothing under aether_amd/).

``decoder_step`` <- nn/dynamicvars/aether_dynamicvars.py:775-870 (``Decoder.forward``; 2-D, one scene, the
objects with mask 1 take part): canonical state of the present objects (canonicalize_augmented_inputs), messages
from their hidden states per edge type (divided by the number of used types, :801-814), messages from the present
state through one AnisotropicEdgeFilter per edge type followed by ReLU (:827-835), both summed over the rows of
``edge2node_inds`` and divided by ``num_vars - 1`` (:816-819,837-840), GRU-style gate, output MLP, rotation back,
residual; absent objects keep their hidden state and get a zero prediction.

Two properties of the reference are restated as they are:
* the edge features are computed from the UN-compacted state array indexed with the compacted edge indices
  (``create_augmented_edge_attr_pos_vel(extended_inputs, send_edges, recv_edges)``, :823), while everything else
  uses the compacted arrays -- the two coincide when no object before the last present one is missing;
* a scene with exactly one present object fails in the reference (``present_agg_msgs`` is never assigned, :843-851):
  ``decoder_step`` raises for it too.

Parity status: PINNED by tests/golden/dyn_decoder.npz (the imported reference ``Decoder`` with graphs from the
reference's ``get_knn_graph_info``, oracle/make_golden_dynamicvars.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

try:                                    # imported as oracle.dynamicvars_oracle (tests) or from oracle/ (fixture script)
    from .seq2seq_oracle import EDGE_POS_IDX, augmented_edge_attr, canonicalize_augmented
except ImportError:
    from seq2seq_oracle import EDGE_POS_IDX, augmented_edge_attr, canonicalize_augmented


def decoder_step(sd, inputs, hidden, edges, node_masks, graph_info, predicted_field, skip_first=False,
                 pos_representation="cart"):
    """inputs [1, Nmax, 4], hidden [1, Nmax, h], edges [1, E, K], node_masks [1, Nmax] or [Nmax],
    graph_info = (send, recv, edge2node_inds) in compacted numbering, predicted_field [1, Nmax, 2]."""
    lin = lambda name, v: F.linear(v, sd[name + ".weight"], sd.get(name + ".bias"))
    K = edges.shape[-1] if edges is not None else len([k for k in sd if k.startswith("msg_fc2.") and k.endswith("weight")])
    ext = torch.cat([inputs, predicted_field], -1)                                     # :781
    node_inds = node_masks.reshape(-1).nonzero()[:, -1]                                # :784
    cur_h, cur_in, cur_ext = hidden[:, node_inds], inputs[:, node_inds], ext[:, node_inds]
    nv = cur_h.shape[1]
    if nv == 0:                                                                        # :841-843
        return torch.zeros_like(inputs), hidden
    if nv == 1:
        raise RuntimeError("a scene with one present object fails in the reference (present_agg_msgs unassigned)")
    rel_feat, Rinv = canonicalize_augmented(cur_ext, False)                            # :790
    send, recv, e2n = graph_info
    k0 = 1 if skip_first else 0
    norm = float(K - k0)
    pre_msg = torch.cat([cur_h[:, recv], cur_h[:, send]], -1)                          # :797-800
    all_msgs = torch.zeros(1, recv.shape[0], cur_h.shape[-1], dtype=inputs.dtype)
    for i in range(k0, K):                                                             # :809-814
        msg = torch.tanh(lin(f"msg_fc2.{i}", torch.tanh(lin(f"msg_fc1.{i}", pre_msg))))
        all_msgs = all_msgs + msg * edges[:, :, i:i + 1] / norm
    incoming = all_msgs[:, e2n[:, 0]].clone()                                          # :816-819
    for i in range(1, e2n.shape[1]):
        incoming = incoming + all_msgs[:, e2n[:, i]]
    agg = incoming / (nv - 1)
    ea = augmented_edge_attr(ext, send, recv, False)                                   # :823 (un-compacted array)
    edge_pos = ea[..., EDGE_POS_IDX[(False, pos_representation)]]
    ea = torch.cat([ea, rel_feat[:, recv]], -1)                                        # :825
    present = torch.zeros_like(all_msgs)
    for i in range(k0, K):                                                             # :829-835
        w = lin(f"edge_filter.{i}.edge_filter.2", torch.relu(lin(f"edge_filter.{i}.edge_filter.0", edge_pos)))
        w = w.reshape(w.shape[:-1] + (ea.shape[-1], -1))                               # anisotropic_filter.py:36-38
        msg = torch.relu((ea.unsqueeze(-2) @ w).squeeze(-2))
        present = present + msg * edges[:, :, i:i + 1]
    pinc = present[:, e2n[:, 0]].clone()                                               # :837-840
    for i in range(1, e2n.shape[1]):
        pinc = pinc + present[:, e2n[:, i]]
    pagg = pinc / (nv - 1)
    r = torch.sigmoid(lin("input_r", rel_feat) + lin("hidden_r", agg) + lin("present_r", pagg))      # :846-851
    i_ = torch.sigmoid(lin("input_i", rel_feat) + lin("hidden_i", agg) + lin("present_i", pagg))
    n = torch.tanh(lin("input_n", rel_feat) + r * lin("hidden_h", agg) + lin("present_n", pagg))
    cur_h = (1 - i_) * n + i_ * cur_h
    pred = lin("out_fc3", torch.relu(lin("out_fc2", torch.relu(lin("out_fc1", cur_h)))))             # :855-857
    pred = torch.cat([torch.einsum("...ij,...j->...i", Rinv, c) for c in pred.split(2, dim=-1)], -1)  # Globalizer
    pred = cur_in + pred
    new_hidden = hidden.clone()
    new_hidden[:, node_inds] = cur_h
    pred_all = torch.zeros_like(inputs)
    pred_all[0, node_inds] = pred[0]
    return pred_all, new_hidden
