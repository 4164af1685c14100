"""ctypes binding of libaether_hip.so (declared in include/aether_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails,
the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libaether_hip.so")

# Order and names follow struct AetherParams in include/aether_hip.h; the second
# element is the reference state_dict key.
PARAM_FIELDS = [
    ("field_w0", "field_net.net.0.weight"), ("field_b0", "field_net.net.0.bias"),
    ("field_w2", "field_net.net.2.weight"), ("field_b2", "field_net.net.2.bias"),
    ("field_w4", "field_net.net.4.weight"), ("field_b4", "field_net.net.4.bias"),
    ("field_emb", "field_net.class_embedding.weight"),
    ("l1_msg_w0", "gnn.layer_1.message_fn.0.weight"), ("l1_msg_b0", "gnn.layer_1.message_fn.0.bias"),
    ("l1_msg_w2", "gnn.layer_1.message_fn.2.weight"), ("l1_msg_b2", "gnn.layer_1.message_fn.2.bias"),
    ("l1_res_w", "gnn.layer_1.res.weight"), ("l1_res_b", "gnn.layer_1.res.bias"),
    ("l1_upd_w0", "gnn.layer_1.update_fn.0.weight"), ("l1_upd_b0", "gnn.layer_1.update_fn.0.bias"),
    ("l1_upd_w2", "gnn.layer_1.update_fn.2.weight"), ("l1_upd_b2", "gnn.layer_1.update_fn.2.bias"),
    ("ln_msg_w0", "gnn.layer_{}.message_fn.0.weight"), ("ln_msg_b0", "gnn.layer_{}.message_fn.0.bias"),
    ("ln_msg_w2", "gnn.layer_{}.message_fn.2.weight"), ("ln_msg_b2", "gnn.layer_{}.message_fn.2.bias"),
    ("ln_upd_w0", "gnn.layer_{}.update_fn.0.weight"), ("ln_upd_b0", "gnn.layer_{}.update_fn.0.bias"),
    ("ln_upd_w2", "gnn.layer_{}.update_fn.2.weight"), ("ln_upd_b2", "gnn.layer_{}.update_fn.2.bias"),
    ("out_w0", "gnn.out_mlp.0.weight"), ("out_b0", "gnn.out_mlp.0.bias"),
    ("out_w3", "gnn.out_mlp.3.weight"), ("out_b3", "gnn.out_mlp.3.bias"),
    ("out_w6", "gnn.out_mlp.6.weight"), ("out_b6", "gnn.out_mlp.6.bias"),
]


class AetherParams(C.Structure):
    _fields_ = [(n, C.c_void_p * 3 if "{}" in k else C.c_void_p) for n, k in PARAM_FIELDS]


def params_struct(tensors: dict) -> AetherParams:
    """Fill an AetherParams from {state_dict key: tensor-like with .data_ptr()}."""
    p = AetherParams()
    for name, key in PARAM_FIELDS:
        if "{}" in key:
            arr = (C.c_void_p * 3)(*[tensors[key.format(l)].data_ptr() for l in (2, 3, 4)])
            setattr(p, name, arr)
        else:
            setattr(p, name, tensors[key].data_ptr())
    return p


class AetherAdamWTensor(C.Structure):
    """include/aether_hip.h: one parameter tensor of aether_adamw_step."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64)]


class AetherGraphInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_edges", C.c_int64), ("n_groups", C.c_int32),
                ("max_group_nodes", C.c_int32), ("max_group_edges", C.c_int32), ("reserved", C.c_int32)]


FLAG_KEEP_INTERMEDIATES = 1
FLAG_FORCE_STREAMED = 2
FLAG_FORCE_FUSED = 4
FLAG_WORKSPACE_REUSED = 8
FLAG_WEIGHTS_PREPARED = 16
FLAG_BACKWARD_ONLY = 32
FLAG_DROPOUT = 64

# name -> (restype, argtypes); every symbol include/aether_hip.h declares
SIGNATURES = {
    "aether_version": (C.c_char_p, []),
    "aether_last_error": (C.c_char_p, []),
    "aether_graph_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "aether_graph_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_size_t, C.POINTER(AetherGraphInfo), C.c_void_p]),
    "aether_graph_perm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "aether_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "aether_forward": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int64, C.c_int64,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t, C.c_void_p,
                                 C.c_int, C.c_void_p]),
    "aether_backward": (C.c_int, [C.POINTER(AetherParams), C.POINTER(AetherParams), C.c_int, C.c_int64,
                                  C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t, C.c_void_p,
                                  C.c_void_p]),
    "aether_rollout": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t,
                                 C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "aether_s2s_field_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "aether_s2s_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p]),
    "aether_s2s_localize": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_s2s_decoder_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "aether_s2s_decoder_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "aether_s2s_plan_bytes": (C.c_size_t, [C.c_int] * 7),
    "aether_s2s_plan_build": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 7 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "aether_s2s_step_workspace_bytes": (C.c_size_t, [C.c_int] * 6 + [C.c_int64, C.c_int64]),
    "aether_s2s_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_float, C.c_int64, C.c_int64] + [C.c_void_p] * 10 +
                        [C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "aether_s2s_rollout": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 10 + [C.c_float, C.c_int64, C.c_int64] + [C.c_void_p] * 4 +
                           [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_void_p, C.c_size_t] + [C.c_void_p] * 3),
    "aether_s2s_filter_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "aether_s2s_filter_prepare": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aether_s2s_prior_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "aether_s2s_prior_step": (C.c_int, [C.c_void_p] + [C.c_int] * 8 + [C.c_int64, C.c_int64] + [C.c_void_p] * 9 +
                              [C.c_size_t] + [C.c_void_p] * 4),
    "aether_s2s_graph_summary_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "aether_s2s_graph_summary": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "aether_s2s_film_modulation_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "aether_s2s_film_modulation": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_void_p]),
    "aether_s2s_film_field_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "aether_s2s_film_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]),
    "aether_knn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "aether_knn_edges": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 6 +
                         [C.c_size_t, C.c_void_p]),
    "aether_sim_electrostatic": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 5 +
                                 [C.c_double] * 3 + [C.c_void_p] * 4),
    "aether_sim_gravitational": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 5 +
                                 [C.c_double] * 3 + [C.c_void_p] * 4),
    "aether_dropout_mask_offset": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "aether_backward_inputs": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_backward_field": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 6 +
                              [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_dynamic_field_backward_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "aether_dynamic_field_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 5 +
                                      [C.c_size_t, C.c_void_p]),
    "aether_dynamic_field_backward_inputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int] + [C.c_void_p] * 5 +
                                             [C.c_size_t, C.c_void_p, C.c_void_p]),
    "aether_dyn_decoder_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64, C.c_int64]),
    "aether_dyn_decoder_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64] +
                                [C.c_void_p] * 9 + [C.c_float, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_dyn_decoder_step_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64] +
                                        [C.c_void_p] * 11 + [C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                             C.c_void_p, C.c_void_p]),
    "aether_dyn_prior_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "aether_dyn_prior_step": (C.c_int, [C.c_void_p] + [C.c_int] * 6 + [C.c_int64, C.c_int64] + [C.c_void_p] * 9 +
                              [C.c_size_t] + [C.c_void_p] * 4),
    "aether_dyn_field_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "aether_dyn_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "aether_dyn_step_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int64, C.c_int64]),
    "aether_dyn_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 6 + [C.c_int] +
                        [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]),
    "aether_dyn_rollout_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "aether_dyn_rollout": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 15 + [C.c_void_p, C.c_size_t,
                                                                                               C.c_void_p]),
    "aether_dyn_step_batched_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_dyn_step_batched": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int] + [C.c_void_p] * 3 + [C.c_void_p] * 12 +
                                [C.c_void_p, C.c_size_t, C.c_void_p]),
    "aether_dyn_rollout_batched_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                               C.c_void_p]),
    "aether_dyn_rollout_batched": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 +
                                   [C.c_void_p] * 3 + [C.c_void_p] * 5 + [C.c_void_p] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "aether_sim_charged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_double, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "aether_rollout_dynamic_field": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int] +
                                     [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                                         C.c_void_p]),
    "aether_s2s_encoder_features": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64] +
                                    [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p, C.c_void_p]),
    "aether_s2s_lstm_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int64] + [C.c_void_p] * 7),
    "aether_s2s_mlp_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aether_s2s_gumbel_hard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int64, C.c_void_p,
                                         C.c_void_p]),
    "aether_dynamic_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "aether_forward_field": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo),
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "aether_debug_fetch": (C.c_int64, [C.c_char_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "aether_workspace_bytes_h": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "aether_dropout_mask_offset_h": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "aether_forward_h": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo),
                                   C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p]),
    "aether_backward_h": (C.c_int, [C.POINTER(AetherParams), C.POINTER(AetherParams), C.c_int, C.c_int, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "aether_backward_inputs_h": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo), C.c_void_p,
                                           C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "aether_rollout_h": (C.c_int, [C.POINTER(AetherParams), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(AetherGraphInfo), C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "aether_rollout_dynamic_field_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int] +
                                       [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                                           C.c_void_p]),
    "aether_debug_fetch_h": (C.c_int64, [C.c_char_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "aether_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "aether_mse_scratch_bytes": (C.c_size_t, []),
    "aether_mse_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "aether_adamw_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "aether_graph_matches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "aether_check_async_error": (C.c_int, []),
    "aether_profile_enable": (C.c_int, [C.c_int]),
    "aether_profile_kernels": (C.c_int, []),
    "aether_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "aether_profile_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
}

_lib = None


class AetherHipError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AetherHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python aether_amd/build.py`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status < 0:
        msg = load().aether_last_error().decode()
        raise AetherHipError(f"{what} failed ({status}): {msg}")
    return status
