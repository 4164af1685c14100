/*
 * aether_hip.h -- C ABI of libaether_hip.so: the MI355X (gfx950) implementation of
 * the Aether state2state step.
 *
 * The reference has no FFI / plugin registry: the seam of this path is the Python
 * nn.Module surface.  The entry points below are what the
 * build's own nn.Module (aether_amd/nn/state2state/aether.py) binds through ctypes;
 * each one names the reference interface it replaces.  Plain pointers and sizes only:
 * every pointer is a DEVICE pointer (HIP) unless it says "host"; `stream` is a
 * hipStream_t passed as void*.  No call allocates, frees or synchronises except
 * aether_graph_build (which synchronises `stream` to report bad indices and to size the
 * fused kernel's groups on the host).
 * All functions return 0 on success, a negative AETHER_E* code on error;
 * aether_last_error() returns a host string for the calling thread's last error.
 */
#ifndef AETHER_HIP_H
#define AETHER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AETHER_OK 0
#define AETHER_EINVAL (-1)   /* bad argument (dims, null pointer, size)       */
#define AETHER_EINDEX (-2)   /* edge index outside [0, n_nodes)               */
#define AETHER_EHIP (-3)     /* a HIP runtime call or kernel launch failed    */
#define AETHER_ESPACE (-4)   /* workspace too small                           */

#define AETHER_HIDDEN 64     /* hidden width the kernels are built for (experiments/lorentz/main.py:42-43) */

/*
 * Parameter pointers, one per tensor of the reference state_dict
 * (nn/state2state/aether.py:143-160, nn/state2state/locs/locs.py:142-225; key list
 * in SURVEY.md 8b).  nn.Linear layout: weight[out][in] row-major, bias[out].
 * The same struct, pointing at gradient buffers, receives the parameter gradients.
 */
typedef struct AetherParams {
    /* field_net.* -- aether.py:108-121 */
    float* field_w0; float* field_b0;       /* [32][2D+16], [32] */
    float* field_w2; float* field_b2;       /* [32][32],    [32] */
    float* field_w4; float* field_b4;       /* [D][32],     [D]  */
    float* field_emb;                       /* [3][16]           */
    /* gnn.layer_1.* -- locs.py:170-178 (only_edge_attr=True) */
    float* l1_msg_w0; float* l1_msg_b0;     /* [64][7D+O+2], [64] */
    float* l1_msg_w2; float* l1_msg_b2;     /* [64][64],     [64] */
    float* l1_res_w;  float* l1_res_b;      /* [64][3D],     [64] */
    float* l1_upd_w0; float* l1_upd_b0;     /* [128][64],    [128] */
    float* l1_upd_w2; float* l1_upd_b2;     /* [64][128],    [64] */
    /* gnn.layer_{2,3,4}.* -- locs.py:179-181 */
    float* ln_msg_w0[3]; float* ln_msg_b0[3];   /* [64][192], [64] */
    float* ln_msg_w2[3]; float* ln_msg_b2[3];   /* [64][64],  [64] */
    float* ln_upd_w0[3]; float* ln_upd_b0[3];   /* [128][64], [128] */
    float* ln_upd_w2[3]; float* ln_upd_b2[3];   /* [64][128], [64] */
    /* gnn.out_mlp.{0,3,6} -- locs.py:160-168 */
    float* out_w0; float* out_b0;           /* [64][64], [64] */
    float* out_w3; float* out_b3;           /* [64][64], [64] */
    float* out_w6; float* out_b6;           /* [D][64],  [D]  */
} AetherParams;

/* Host-side summary of a built graph; filled by aether_graph_build, passed back to aether_forward. */
typedef struct AetherGraphInfo {
    int64_t n_nodes, n_edges;
    int32_t n_groups;          /* > 0: workgroups of the fused kernel (0: streamed path only)      */
    int32_t max_group_nodes;   /* most nodes / in-edges one workgroup owns (<= 32 / <= 384)         */
    int32_t max_group_edges;
    int32_t reserved;          /* bit 0: groups are split over two cooperating workgroups           */
} AetherGraphInfo;

/* aether_forward flags */
#define AETHER_FLAG_KEEP_INTERMEDIATES 1  /* also write nodeinfo, x0..x4, e1..e4 to the workspace */
#define AETHER_FLAG_FORCE_STREAMED 2      /* use the layer-by-layer kernels even for small graphs */
#define AETHER_FLAG_FORCE_FUSED 4         /* fail instead of falling back to the streamed kernels */
/* Accepted and ignored since 0.4: the fused kernel's inter-workgroup hand-off words now live in the `graph` buffer
 * (zeroed by aether_graph_build, re-armed by every launch that used them), so no call zeroes anything and a
 * workspace may be re-purposed freely.  Consequence: two launches that use the SAME graph buffer must not run
 * concurrently (different streams); build a second graph view for that. */
#define AETHER_FLAG_WORKSPACE_REUSED 8
/* The fused kernel multiplies its 64 x 64 layers (edge MLPs, node update, out MLP) as three fp16 matrix-core terms on
 * operands split into two fp16 pieces each, with exact power-of-two rescaling where a wave's activations leave fp16's
 * comfortable range (22 significand bits; parity 1e-5, measured 4e-8: csrc/common.h).  Weights must be below 65,504 in
 * magnitude (beyond: non-finite outputs).  The weights' pieces are prepared by a small kernel in front of every call
 * (workspace region).  Set this flag when `workspace` still holds the pieces written by an earlier call with the SAME
 * parameter values (inference loops, rollouts): that kernel is then skipped.  Never set it after the weights changed. */
#define AETHER_FLAG_WEIGHTS_PREPARED 16
/* With AETHER_FLAG_KEEP_INTERMEDIATES: keep only what aether_backward reads -- the last layer's messages e4 (12 MB at
 * N=20, batch=128; 8.6 GB for a 33.5 M-edge shard) are then not written (aether_debug_fetch("e4") is undefined). */
#define AETHER_FLAG_BACKWARD_ONLY 32
/* Training with dropout_prob > 0 (round 3).  The reference's only Dropout layers follow the two SiLUs of the out MLP
 * (nn/state2state/locs/locs.py:160-168).  The caller draws the masks: float[2][n_nodes][64] of SCALES (0 or 1 / (1 - p)),
 * written at byte offset aether_dropout_mask_offset(n_nodes, n_edges, num_dims) of the training workspace before
 * aether_forward(... AETHER_FLAG_KEEP_INTERMEDIATES | AETHER_FLAG_DROPOUT); aether_backward on the same workspace applies
 * the same masks (a word the forward leaves there says whether it used any).  The module fills them with torch's bernoulli_:
 * the same distribution as nn.Dropout, not its random stream. */
#define AETHER_FLAG_DROPOUT 64
size_t aether_dropout_mask_offset(int64_t n_nodes, int64_t n_edges, int num_dims);

/* Library / build identification (host string, static storage). */
const char* aether_version(void);
const char* aether_last_error(void);

/*
 * Graph preparation: receiver-sorted (CSR) view of an edge index.
 * Replaces what torch_scatter.scatter does implicitly on every call
 * (nn/state2state/locs/locs.py:236-238) with a one-time stable sort by receiver, so
 * that the per-layer mean is a deterministic segmented sum in edge order.
 *   send, recv : int64[E]   edges[0], edges[1] of Aether.forward (aether.py:169)
 *   graph      : device buffer of aether_graph_bytes(E, n_nodes) bytes, filled here
 * It also finds the contiguous node ranges that no edge leaves (whole graphs of a batch) and
 * packs them into groups for the single-launch fused kernel (small graphs only; `info`).
 * Synchronises `stream` (it reports bad indices and sizes the groups on the host); uses
 * temporary host memory.  Returns AETHER_EINDEX if any index is out of range.
 */
size_t aether_graph_bytes(int64_t n_edges, int64_t n_nodes);
/* 1 if (send, recv) is element for element the edge index `graph` was built from, 0 if not, < 0 on error.  One small
 * kernel + a stream synchronisation (a 4-byte flag in host-mapped memory): for callers that rebuild identical index
 * tensors every batch (experiments/lorentz/main.py:211-212) and want to reuse the view without sorting again. */
int aether_graph_matches(const int64_t* send, const int64_t* recv, int64_t n_edges, int64_t n_nodes, const void* graph,
                         void* stream);
int aether_graph_build(const int64_t* send, const int64_t* recv, int64_t n_edges,
                       int64_t n_nodes, void* graph, size_t graph_bytes, AetherGraphInfo* info,
                       void* stream);
/* Debug / test access: copies the sorted-position -> original-edge-id map (int32[E]). */
int aether_graph_perm(const void* graph, int64_t n_edges, int64_t n_nodes, int32_t* perm_out,
                      void* stream);

/* Bytes of scratch the forward (and, with keep_for_backward, the backward) needs. */
size_t aether_workspace_bytes(int64_t n_nodes, int64_t n_edges, int num_dims,
                              int keep_for_backward);

/*
 * Forward step: replaces Aether.forward (nn/state2state/aether.py:169-186):
 * field query -> local frames -> edge features -> 4 x (edge MLP, mean by receiver,
 * node update) -> out MLP -> rotate back -> x + pred.
 *   x, vel          : float[n_nodes][D]      positions, velocities
 *   charges         : float[n_nodes]         in {-1, 0, +1}  (aether.py:122-124)
 *   edge_attr_orig  : float[E][2]            [q_i q_j, |x_i - x_j|], ORIGINAL edge order
 *   graph           : from aether_graph_build for this (send, recv)
 *   out             : float[n_nodes][D]
 * `h` of the reference signature is unused by the reference and has no counterpart.
 * Small graphs (info->n_groups > 0) run as ONE kernel launch (csrc/fused.h); anything else runs
 * layer by layer (csrc/streamed.h).  Stream-ordered, re-entrant per (workspace, out) pair.
 */
int aether_forward(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                   const float* x, const float* vel, const float* charges,
                   const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                   void* workspace, size_t workspace_bytes, float* out, int flags, void* stream);

/*
 * Autoregressive rollout, all on the device: `steps` forward steps back to back,
 *   x_{t+1} = Aether(x_t, v_t),  v_{t+1} = (x_{t+1} - x_t) / dt,
 * with edge_attr_orig = [q_i q_j, |x_i - x_j|] derived inside the kernels from the current positions
 * (the runner's per-batch prep, experiments/lorentz/main.py:243-246) -- the protocol of the
 * "20-step rollout MSE" metric (oracle/aether_oracle.py::rollout restates it).  Replaces a
 * Python loop of model calls, gathers and concatenations (about 2/3 of the time of such a loop at
 * N=20, batch=128) by one kernel launch per step.
 *   x0, vel0   : float[n_nodes][D], state at t = 0 (not modified)
 *   trajectory : float[steps][n_nodes][D], positions x_1 .. x_steps
 *   workspace  : aether_workspace_bytes(n_nodes, n_edges, D, 0) bytes
 *   flags      : AETHER_FLAG_FORCE_* / AETHER_FLAG_WORKSPACE_REUSED as for aether_forward
 * Stream-ordered; capture it in a hipGraph to replay a whole rollout with one launch.
 */
int aether_rollout(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                   const float* x0, const float* vel0, const float* charges, const void* graph,
                   const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                   float* trajectory, int steps, float dt, int flags, void* stream);

/*
 * Backward step: gradients of a scalar loss w.r.t. every parameter, given grad_out = dL/d(out).
 * Replaces torch.autograd through Aether.forward (the runner's loss.backward(),
 * experiments/lorentz/main.py:289-291).  Inputs (x, vel, charges, edge attributes) are data and
 * receive no gradient, as in the runner (main.py:243-247 detaches them).
 *   The forward on the same `workspace` must have run with AETHER_FLAG_KEEP_INTERMEDIATES and a
 *   workspace of aether_workspace_bytes(..., keep_for_backward = 1) bytes.
 *   grads : AetherParams whose pointers address the gradient buffers (same shapes as params);
 *           every element is overwritten (not accumulated).
 * Deterministic (ordered partial sums, no atomics).  Stream-ordered.
 */
int aether_backward(const AetherParams* params, const AetherParams* grads, int num_dims, int64_t n_nodes,
                    int64_t n_edges, const float* x, const float* vel, const float* charges,
                    const void* graph, const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                    const float* grad_out, void* stream);

/*
 * Test hook: copy one named intermediate of the last aether_forward on `workspace`
 * (the fused path writes them only under AETHER_FLAG_KEEP_INTERMEDIATES)
 * into `dst` (device).  Names: "field"[n][D] "canon"[n][2D] (= rel_feat[:, D:]) "R"[n][D*D] "x0".."x4"[n][64]
 * "e1".."e4"[E][64] (receiver-sorted order; map back with aether_graph_perm).
 * Returns the number of floats written, or a negative error.
 */
int64_t aether_debug_fetch(const char* name, int num_dims, int64_t n_nodes, int64_t n_edges,
                           const void* workspace, float* dst, void* stream);

/*
 * Dynamic-field variant of the state2state model: DynamicFieldAether.forward
 * (nn/state2state/dynamic_field_aether.py:79-100) = aether_dynamic_field (LatentFieldNetwork, :31-48:
 * attention-pooled graph summary + FiLM field net, hidden 32) followed by aether_forward_field, i.e.
 * aether_forward with the per-node field supplied instead of the built-in field net (params->field_* are
 * not read for the result but must point to readable memory of the documented sizes).
 *   graphs are consecutive blocks of nodes_per_graph nodes (x.reshape(-1, num_nodes, .), :38);
 *   field : float[n_nodes][D]
 */
typedef struct AetherDynFieldParams {
    const float* gate_w0; const float* gate_b0; const float* gate_w2; const float* gate_b2;   /* [32][2D],[32],[1][32],[1] */
    const float* nn_w0; const float* nn_b0; const float* nn_w2; const float* nn_b2;           /* [32][2D],[32],[32][32],[32] */
    const float* lin1_w; const float* lin1_b; const float* lin2_w; const float* lin2_b;       /* [32][2D+16],[32],[32][32],[32] */
    const float* lin3_w; const float* lin3_b;                                                 /* [D][32],[D] */
    const float* film1_w0; const float* film1_b0; const float* film1_w2; const float* film1_b2;
    const float* film1_w4; const float* film1_b4;                                             /* [32][32] x2, [64][32] */
    const float* film2_w0; const float* film2_b0; const float* film2_w2; const float* film2_b2;
    const float* film2_w4; const float* film2_b4;
    const float* emb;                                                                         /* [3][16] */
} AetherDynFieldParams;
int aether_dynamic_field(const AetherDynFieldParams* params, int num_dims, int64_t n_graphs, int nodes_per_graph,
                         const float* x, const float* vel, const float* charges, float* field, void* stream);
int aether_forward_field(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                         const float* x, const float* vel, const float* charges, const float* field,
                         const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                         void* workspace, size_t workspace_bytes, float* out, int flags, void* stream);
/*
 * aether_rollout for the dynamic-field model: every step runs aether_dynamic_field on the current state (into
 * field_scratch, float[n_nodes][D]) and then the step with that field; otherwise as aether_rollout.
 */
int aether_rollout_dynamic_field(const AetherParams* params, const AetherDynFieldParams* dyn_params, int num_dims,
                                 int64_t n_nodes, int64_t n_edges, int nodes_per_graph, const float* x0,
                                 const float* vel0, const float* charges, const void* graph,
                                 const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                                 float* field_scratch, float* trajectory, int steps, float dt, int flags, void* stream);
/*
 * Gradients with respect to the INPUTS of the step (round 3) -- the reference's forward is differentiable in x / vel
 * (nn/state2state/aether.py:169-186).  Call after aether_backward, on the same workspace (it reads what that call left:
 * dL/d(layer-1 edge features), dL/dn_1, dL/df) with `out` = the forward's output and the same grad_out:
 *   grad_x [n_nodes][D], grad_vel [n_nodes][D] (overwritten); grad_edge_attr [n_edges][2] in the caller's edge order, or
 *   NULL.  field_input_grad = NULL: the built-in field net (its input gradient is recomputed from dL/df).  For a step that
 *   ran through aether_forward_field: after aether_backward_field, float[n_nodes][2D] = dL/d[x | vel] THROUGH the external
 *   field, e.g. from aether_dynamic_field_backward_inputs below.  Positions and velocities enter through
 *   x + R(v) y, the field net's inputs, rel_feat = [0 | R^T v | R^T f] and the local-frame edge features (r, Euler angles
 *   of R_i^T R_j, distance, bearing, R_i^T v_j, R_i^T f_j); charges are an index (no gradient).
 */
int aether_backward_inputs(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges, const float* x,
                           const float* vel, const float* charges, const void* graph, const AetherGraphInfo* info,
                           void* workspace, size_t workspace_bytes, const float* out, const float* grad_out, float* grad_x,
                           float* grad_vel, float* grad_edge_attr, const float* field_input_grad, void* stream);

/*
 * Training of the dynamic-field variant.  aether_backward_field = aether_backward for a step that ran through
 * aether_forward_field with AETHER_FLAG_KEEP_INTERMEDIATES: gradients of the GNN / res / out-MLP tensors into
 * `grads` (grads->field_* are not written) and dL/dfield into grad_field [n_nodes][D].
 * aether_dynamic_field_backward then differentiates LatentFieldNetwork (dynamic_field_aether.py:31-48: FiLM field
 * net, FiLM modulators, attention pooling): `grads` holds one output pointer per tensor of `params`;
 * workspace: aether_dynamic_field_backward_workspace_bytes(num_dims, n_graphs) bytes (one partial gradient row per
 * graph, added over the graphs in order: no atomics).
 */
int aether_backward_field(const AetherParams* params, const AetherParams* grads, int num_dims, int64_t n_nodes,
                          int64_t n_edges, const float* x, const float* vel, const float* charges,
                          const void* graph, const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                          const float* grad_out, float* grad_field, void* stream);
size_t aether_dynamic_field_backward_workspace_bytes(int num_dims, int64_t n_graphs);
int aether_dynamic_field_backward(const AetherDynFieldParams* params, const AetherDynFieldParams* grads, int num_dims,
                                  int64_t n_graphs, int nodes_per_graph, const float* x, const float* vel,
                                  const float* charges, const float* grad_field, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* The same, and grad_field_inputs [n_nodes][2D] (or NULL) = dL/d[x | vel] through the field: the FiLM net's first Linear and
 * the graph summary's nn / gate_nn (every node's row enters its graph's attention pooling).  Feed it to
 * aether_backward_inputs(field_input_grad). */
int aether_dynamic_field_backward_inputs(const AetherDynFieldParams* params, const AetherDynFieldParams* grads, int num_dims,
                                         int64_t n_graphs, int nodes_per_graph, const float* x, const float* vel,
                                         const float* charges, const float* grad_field, void* workspace,
                                         size_t workspace_bytes, float* grad_field_inputs, void* stream);

/*
 * Any hidden_size (round 4).  The reference builds Aether / DynamicFieldAether with whatever `--nf` says
 * (experiments/lorentz/main.py:42-43,143; nn/state2state/aether.py:143-158, nn/state2state/locs/locs.py:142-181): every
 * [64]-sized dimension of AetherParams above becomes [hidden], [128] becomes [2 hidden], [192] becomes [3 hidden].
 * The calls below are the calls above with the width as an argument:
 *   hidden == 64        : exactly the functions above (fused / streamed 64-wide kernels);
 *   hidden == 64 m > 64 : layer-by-layer on one generic split-operand (2 x fp16) MFMA GEMM kernel with fused epilogues (csrc/wide.h);
 *   other widths        : not accepted here -- zero-pad the parameters to the next multiple of 64 (padded channels stay
 *                         exactly zero through SiLU, the mean and the residuals; the Python module does this).
 * `field` (forward) / `grad_field` (backward) may be NULL (the built-in field net) or the external field of
 * aether_forward_field / aether_backward_field.  Workspaces are sized by aether_workspace_bytes_h; the dropout masks are
 * float[2][n_nodes][hidden] at aether_dropout_mask_offset_h.  aether_debug_fetch_h reads a KEEP_INTERMEDIATES workspace.
 */
size_t aether_workspace_bytes_h(int64_t n_nodes, int64_t n_edges, int num_dims, int hidden, int keep_for_backward);
size_t aether_dropout_mask_offset_h(int64_t n_nodes, int64_t n_edges, int num_dims, int hidden);
int aether_forward_h(const AetherParams* params, int num_dims, int hidden, int64_t n_nodes, int64_t n_edges,
                     const float* x, const float* vel, const float* charges, const float* field,
                     const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                     void* workspace, size_t workspace_bytes, float* out, int flags, void* stream);
int aether_backward_h(const AetherParams* params, const AetherParams* grads, int num_dims, int hidden, int64_t n_nodes,
                      int64_t n_edges, const float* x, const float* vel, const float* charges, const void* graph,
                      const AetherGraphInfo* info, void* workspace, size_t workspace_bytes, const float* grad_out,
                      float* grad_field, void* stream);
int aether_backward_inputs_h(const AetherParams* params, int num_dims, int hidden, int64_t n_nodes, int64_t n_edges,
                             const float* x, const float* vel, const float* charges, const void* graph,
                             const AetherGraphInfo* info, void* workspace, size_t workspace_bytes, const float* out,
                             const float* grad_out, float* grad_x, float* grad_vel, float* grad_edge_attr,
                             const float* field_input_grad, void* stream);
int aether_rollout_h(const AetherParams* params, int num_dims, int hidden, int64_t n_nodes, int64_t n_edges, const float* x0,
                     const float* vel0, const float* charges, const void* graph, const AetherGraphInfo* info,
                     void* workspace, size_t workspace_bytes, float* trajectory, int steps, float dt, int flags,
                     void* stream);
int aether_rollout_dynamic_field_h(const AetherParams* params, const AetherDynFieldParams* dyn_params, int num_dims,
                                   int hidden, int64_t n_nodes, int64_t n_edges, int nodes_per_graph, const float* x0,
                                   const float* vel0, const float* charges, const void* graph, const AetherGraphInfo* info,
                                   void* workspace, size_t workspace_bytes, float* field_scratch, float* trajectory,
                                   int steps, float dt, int flags, void* stream);
int64_t aether_debug_fetch_h(const char* name, int num_dims, int hidden, int64_t n_nodes, int64_t n_edges,
                             const void* workspace, float* dst, void* stream);

/*
 * seq2seq Aether, field query: replaces Aether.predict_field
 * (nn/seq2seq/aether.py:86-90) = FourierFeatureMapper (nn/nn/fourier_feature_mapper.py:7-21) followed by
 * field_net (aether.py:72-78).  Unlike the state2state field it sees positions only.
 *   B      : float[D][hidden/2]    buffer coordinate_embedding.B
 *   w0..b4 : field_net.{0,2,4}.{weight,bias}: [h][h],[h],[h][h],[h],[D][h],[D]
 *   x      : float[n_points][x_stride], the first D columns are the coordinates (x[..., :D], :87)
 *   field  : float[n_points][D]
 *   workspace : aether_s2s_field_workspace_bytes(n_points, hidden) bytes
 * Stream-ordered; four launches (features, three Linear layers on the matrix core).
 */
typedef struct AetherS2SFieldParams {
    const float* B;
    const float* w0; const float* b0;
    const float* w2; const float* b2;
    const float* w4; const float* b4;
} AetherS2SFieldParams;
size_t aether_s2s_field_workspace_bytes(int64_t n_points, int hidden);
int aether_s2s_field(const AetherS2SFieldParams* params, int num_dims, int hidden, int64_t n_points,
                     const float* x, int x_stride, void* workspace, size_t workspace_bytes, float* field,
                     void* stream);

/*
 * seq2seq Aether, augmented local frames: replaces AugmentedLocalizer.forward
 * (nn/utils/augmented_global_to_local.py:52-68; canonicalize_augmented_inputs and
 * create_augmented[_3d]_edge_attr_pos_vel, nn/utils/canonicalization.py:33-56,111-172) on a flattened
 * batch: nodes of all graphs in one array, edges as global node indices (no out-of-range check).
 *   x         : float[n_nodes][3D]           pos | vel | force
 *   send/recv : int64[n_edges]               edge j -> i, features in i's frame
 *   polar     : 1 = pos_representation 'polar', 0 = 'cart' (columns picked for edge_pos, :19-24)
 *   rel_feat  : float[n_nodes][7D+O]         canonical state | features of the edge from the virtual origin node
 *   Rinv      : float[n_nodes][D][D]         un-transposed frame, as the reference returns it
 *   edge_attr : float[n_edges][2(4D+O)+3D]   edge features | rel_feat[recv]      (24 | 39 columns)
 *   edge_pos  : float[n_edges][D+O]
 * O = D(D-1)/2.  Stream-ordered; two launches.
 */
int aether_s2s_localize(int num_dims, int64_t n_nodes, int64_t n_edges, const float* x,
                        const int64_t* send, const int64_t* recv, int polar, float* rel_feat,
                        float* Rinv, float* edge_attr, float* edge_pos, void* stream);

/*
 * seq2seq Aether, one step of the recurrent decoder (decoder half): replaces
 * RecurrentDecoder.forward (nn/seq2seq/aether.py:590-654) on a flattened batch -- messages from the
 * hidden states and from the present local-frame features per edge type, mean by receiver, GRU-style gate,
 * output MLP, rotate back (Globalizer, nn/utils/local_to_global.py:7-13), residual.
 *   params     : pointers to the reference module's tensors (nn.Linear layout weight[out][in]);
 *                index k of the arrays = edge type k (at most 4)
 *   inputs     : float[n_nodes][2D]  pos | vel          hidden_in : float[n_nodes][h]
 *   edge_w     : float[n_edges][K]   edge-type weights (one-hot or soft; `edges` of the reference)
 *   field      : float[n_nodes][D]   predicted field (aether_s2s_field)
 *   send, recv : int64[n_edges]      global node indices, messages j -> i
 *   order, rowptr : int64[n_edges], int64[n_nodes + 1]  edges grouped by receiver (stable), CSR offsets
 *   outputs    : float[n_nodes][2D]  hidden_out : float[n_nodes][h]  (may not alias hidden_in)
 *   workspace  : aether_s2s_decoder_workspace_bytes(D, h, n_nodes, n_edges) bytes
 * Dropout is 0 (the reference forces it to 0 outside training, aether.py:594).  Stream-ordered.
 */
typedef struct AetherS2SDecoderParams {
    const float* msg_fc1_w[4]; const float* msg_fc1_b[4];    /* [h][2h] (receiver | sender halves), [h] */
    const float* msg_fc2_w[4]; const float* msg_fc2_b[4];    /* [h][h], [h] */
    const float* hidden_r_w; const float* hidden_i_w; const float* hidden_h_w;      /* [h][h], no bias */
    const float* present_r_w; const float* present_r_b;
    const float* present_i_w; const float* present_i_b;
    const float* present_n_w; const float* present_n_b;      /* [h][h], [h] */
    const float* out0_w; const float* out0_b; const float* out3_w; const float* out3_b;   /* [h][h], [h] */
    const float* out6_w; const float* out6_b;                /* [2D][h], [2D] */
    const float* pmsg_fc1_w[4]; const float* pmsg_fc1_b[4];  /* [h][2(4D+O)+3D], [h] */
    const float* pmsg_fc2_w[4]; const float* pmsg_fc2_b[4];  /* [h][h], [h] */
    const float* input_r_w; const float* input_r_b;
    const float* input_i_w; const float* input_i_b;
    const float* input_n_w; const float* input_n_b;          /* [h][7D+O], [h] */
} AetherS2SDecoderParams;
size_t aether_s2s_decoder_workspace_bytes(int num_dims, int hidden, int64_t n_nodes, int64_t n_edges);
int aether_s2s_decoder_step(const AetherS2SDecoderParams* params, int num_dims, int hidden, int num_edge_types,
                            int skip_first, int64_t n_nodes, int64_t n_edges, const float* inputs,
                            const float* hidden_in, const float* edge_w, const float* field,
                            const int64_t* send, const int64_t* recv, const int64_t* order,
                            const int64_t* rowptr, void* workspace, size_t workspace_bytes, float* outputs,
                            float* hidden_out, void* stream);

/*
 * seq2seq Aether, one step of the encoder's prior (prior half): replaces
 * Encoder.single_step_forward (nn/seq2seq/aether.py:384-410) on a flattened batch -- local frames,
 * anisotropic edge filter (nn/nn/anisotropic_filter.py:34-40; the [E, R h] filter bank is never
 * materialised), edge2node / res1, mlp3, node2edge + skip, mlp4 (RefNRIMLP in eval mode: BatchNorm with
 * running statistics, nn/utils/model_utils.py:15-55), one LSTM step per edge, prior_fc_out.
 *   params  : the reference Encoder's tensors (nn.Linear / nn.LSTM layouts; gate order i, f, g, o)
 *   inputs  : float[n_nodes][2D]   field : float[n_nodes][D]   h0, c0 : float[n_edges][rnn_hidden]
 *   send, recv, order, rowptr : as for aether_s2s_decoder_step;  num_vars = objects per graph (edge2node
 *                               divides by num_vars - 1, aether.py:348)
 *   logits  : float[n_edges][K]    h1, c1 : float[n_edges][rnn_hidden]
 *   workspace : aether_s2s_prior_workspace_bytes(...) bytes
 * aether_s2s_gumbel_hard: gumbel_softmax(hard=True) of nn/utils/model_utils.py:58-118 with the uniform
 * draw U[n_edges][K] supplied by the caller (the reference draws it with torch.rand on the host).
 */
typedef struct AetherS2SPriorParams {
    const float* mlp3_w0; const float* mlp3_b0; const float* mlp3_w3; const float* mlp3_b3;       /* [h][h] */
    const float* mlp3_bn_w; const float* mlp3_bn_b; const float* mlp3_bn_mean; const float* mlp3_bn_var;
    const float* mlp4_w0; const float* mlp4_b0; const float* mlp4_w3; const float* mlp4_b3;       /* [h][3h], [h][h] */
    const float* mlp4_bn_w; const float* mlp4_bn_b; const float* mlp4_bn_mean; const float* mlp4_bn_var;
    const float* lstm_w_ih; const float* lstm_w_hh; const float* lstm_b_ih; const float* lstm_b_hh; /* [4R][h], [4R][R] */
    const float* prior_w[4]; const float* prior_b[4];                                            /* prior_fc_out */
    const float* res1_w; const float* res1_b;                                                    /* [h][7D+O] */
    const float* filt_w0; const float* filt_b0;                                                  /* [h][D+O] */
    const float* filt_w2; const float* filt_b2;                                                  /* [R h][h], R = 2(4D+O)+3D */
    const void* filt_image;   /* aether_s2s_filter_prepare(filt_w2, R, h, ..) of the CURRENT filt_w2 values, or NULL: the
                                 step then builds the image in its workspace on every call (reads all of filt_w2) */
} AetherS2SPriorParams;
/*
 * The filter GEMM (anisotropic_filter.py:34-40) runs on the matrix cores as three fp16 terms on operands split into two
 * fp16 pieces each, scaled by exact powers of two (22 significand bits: fp32-equivalent at the 1e-5 bar); its
 * weight operand is a re-ordered two-piece image of filt_w2 (4 bytes per weight + a 256-byte trailer holding max |filt_w2|,
 * from which the scale is derived).  Prepare it once per weight version (three small launches on `stream`):
 *   image : device buffer of aether_s2s_filter_image_bytes(n_features, hidden) bytes, 16-byte aligned
 */
size_t aether_s2s_filter_image_bytes(int n_features, int hidden);
int aether_s2s_filter_prepare(const float* filt_w2, int n_features, int hidden, void* image, size_t image_bytes,
                              void* stream);
size_t aether_s2s_prior_workspace_bytes(int num_dims, int hidden, int rnn_hidden, int prior_hidden,
                                        int64_t n_nodes, int64_t n_edges);
int aether_s2s_prior_step(const AetherS2SPriorParams* params, int num_dims, int hidden, int rnn_hidden,
                          int prior_layers, int prior_hidden, int num_edge_types, int polar, int num_vars,
                          int64_t n_nodes, int64_t n_edges, const float* inputs, const float* field,
                          const float* h0, const float* c0, const int64_t* send, const int64_t* recv,
                          const int64_t* order, const int64_t* rowptr, void* workspace, size_t workspace_bytes,
                          float* logits, float* h1, float* c1, void* stream);
int aether_s2s_gumbel_hard(const float* logits, const float* uniform, float tau, int num_edge_types,
                           int64_t n_edges, float* edges, void* stream);
/*
 * The pieces of aether_s2s_prior_step on their own, for the full-sequence encoder (Encoder.forward,
 * nn/seq2seq/aether.py:350-382: the same per-time-step features, then a forward and a reverse LSTM over time and the two
 * heads prior_fc_out / encoder_fc_out):
 *   aether_s2s_encoder_features : everything up to mlp4 (:354-369) -> features float[n_edges][hidden]; workspace as for
 *                                 aether_s2s_prior_step (aether_s2s_prior_workspace_bytes with any rnn / prior sizes)
 *   aether_s2s_lstm_step        : one nn.LSTM cell step on n_rows rows (gate order i, f, g, o); gates: scratch
 *                                 float[n_rows][4 rnn_hidden]
 *   aether_s2s_mlp_head         : Linear (- ELU - Linear)* as prior_fc_out / encoder_fc_out build it (:286-300); w, b: HOST
 *                                 arrays of `layers` device pointers; scratch float[2][n_rows][hidden_size] when layers > 1
 */
int aether_s2s_encoder_features(const AetherS2SPriorParams* params, int num_dims, int hidden, int polar, int num_vars,
                                int64_t n_nodes, int64_t n_edges, const float* inputs, const float* field,
                                const int64_t* send, const int64_t* recv, const int64_t* order, const int64_t* rowptr,
                                void* workspace, size_t workspace_bytes, float* features, void* stream);
int aether_s2s_lstm_step(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int in_size,
                         int rnn_hidden, int64_t n_rows, const float* x, const float* h0, const float* c0, float* gates,
                         float* h1, float* c1, void* stream);
int aether_s2s_mlp_head(const float* const* w, const float* const* b, int layers, int in_size, int hidden_size,
                        int out_size, int64_t n_rows, const float* x, float* scratch, float* out, void* stream);

/*
 * The whole autoregressive step in one call: field query -> prior step -> hard Gumbel sample -> decoder
 * step, nn/seq2seq/aether.py:176-185 (= predict_field :86-90, Encoder.single_step_forward :384-410, gumbel_softmax
 * :92-98, RecurrentDecoder.forward :590-654), ~31 launches instead of the ~75 of the four entry points above: dense
 * layers that are independent of each other share a launch, the gate pre-activations are K-concatenated products, the
 * local frames are built once for prior and decoder, and everything derived from the weights alone comes from a PLAN:
 *   aether_s2s_plan_build : filter image, padded input layers, BatchNorm affines, concatenated gate weights / summed gate
 *                           biases, summed LSTM biases, two-piece fp16 images of the dense layers (used from 2 K rows on; `field` may be
 *                           NULL when every step will be given ext_field, otherwise pass the field net the steps will use) -> plan (device, 256-byte aligned, aether_s2s_plan_bytes); rebuild after the weights change
 *   aether_s2s_step       : inputs [n_nodes][2D], decoder_hidden_in [n_nodes][hd], h0 / c0 [n_edges][rnn], uniform
 *                           [n_edges][K] (the U(0,1) draw of gumbel_softmax) -> outputs, decoder_hidden_out, h1, c1 and,
 *                           when edges_out is not NULL, the sampled edge types [n_edges][K].  ext_field != NULL replaces the
 *                           built-in field query by a given field [n_nodes][D] (the dynamic-field model); field params may
 *                           then be NULL.  Graph arrays as for aether_s2s_prior_step.
 *   aether_s2s_rollout    : the loop of predict_future (:166-185) on the device: burn_in_steps teacher-forced steps on
 *                           burn_in [burn_in_steps][n_nodes][2D] (predictions discarded), then `steps` autoregressive steps
 *                           from `inputs`; predictions [steps][n_nodes][2D], edges_out [steps][n_edges][K] or NULL; uniform
 *                           [burn_in_steps + steps][n_edges][K]; decoder_state [n_nodes][hd], h, c are read at entry and hold the final
 *                           state at exit.
 * Results equal the four separate calls up to the rounding of re-associated sums (gate biases are added once, as a sum).
 */
size_t aether_s2s_plan_bytes(int num_dims, int encoder_hidden, int decoder_hidden, int rnn_hidden, int prior_layers,
                             int prior_hidden, int num_edge_types);
int aether_s2s_plan_build(const AetherS2SFieldParams* field, const AetherS2SPriorParams* prior,
                          const AetherS2SDecoderParams* decoder, int num_dims, int encoder_hidden, int decoder_hidden,
                          int rnn_hidden, int prior_layers, int prior_hidden, int num_edge_types, void* plan,
                          size_t plan_bytes, void* stream);
size_t aether_s2s_step_workspace_bytes(int num_dims, int encoder_hidden, int decoder_hidden, int rnn_hidden, int prior_hidden,
                                       int num_edge_types, int64_t n_nodes, int64_t n_edges);
int aether_s2s_step(const AetherS2SFieldParams* field, const AetherS2SPriorParams* prior, const AetherS2SDecoderParams* decoder,
                    const void* plan, int num_dims, int encoder_hidden, int decoder_hidden, int rnn_hidden, int prior_layers,
                    int prior_hidden, int num_edge_types, int skip_first, int polar, int num_vars, float tau, int64_t n_nodes,
                    int64_t n_edges, const int64_t* send, const int64_t* recv, const int64_t* order, const int64_t* rowptr,
                    const float* inputs, const float* ext_field, const float* decoder_hidden_in, const float* h0,
                    const float* c0, const float* uniform, void* workspace, size_t workspace_bytes, float* outputs,
                    float* decoder_hidden_out, float* h1, float* c1, float* edges_out, void* stream);
int aether_s2s_rollout(const AetherS2SFieldParams* field, const AetherS2SPriorParams* prior, const AetherS2SDecoderParams* decoder,
                       const void* plan, int num_dims, int encoder_hidden, int decoder_hidden, int rnn_hidden, int prior_layers,
                       int prior_hidden, int num_edge_types, int skip_first, int polar, int num_vars, float tau, int64_t n_nodes,
                       int64_t n_edges, const int64_t* send, const int64_t* recv, const int64_t* order, const int64_t* rowptr,
                       int burn_in_steps, const float* burn_in, int steps, const float* inputs, float* decoder_state, float* h,
                       float* c, const float* uniform, void* workspace, size_t workspace_bytes, float* predictions,
                       float* edges_out, void* stream);

/*
 * seq2seq dynamic-field variant: nn/seq2seq/dynamic_field_aether.py, the model
 * scripts/gravitational_field_3d_aether.sh trains (use_charges is never set by a runner: False).
 *
 * aether_s2s_graph_summary replaces GraphSummary.forward (nn/nn/graph_pool.py:50-71), run once per
 * sequence on the burn-in trajectories (dynamic_field_aether.py:214-218): particle embedding, a GRU over
 * time per object, [x | last hidden] + sinusoidal positional encoding (:10-28, eval mode: no dropout), and
 * torch_geometric's AttentionalAggregation over all (object, time) items of a graph
 * (softmax = exp(g - max) / (sum + 1e-16)).
 *   x       : float[batch][num_objects][timesteps][input_size]
 *   params  : nn.Linear / nn.GRU layouts (gates r, z, n); pe = buffer `pe.pe` [pe_len][input_size + hidden]
 *   summary : float[batch][hidden]
 * aether_s2s_film_modulation: the four FiLM modulators of FilmedNetwork (nn/nn/film.py:43-60) applied to the
 * summary: mod = gamma_1 | beta_1 | gamma_2 | beta_2, each [batch][mlp_hidden] (+ one scratch plane);
 * the summary is fixed for a sequence, so this also runs once.
 * aether_s2s_film_field replaces predict_field (dynamic_field_aether.py:117-134): Fourier features, then
 * linear_1 - FiLM - SiLU - linear_2 - FiLM - SiLU - linear_3 (nn/nn/filmed_network.py:27-35); point n belongs
 * to graph n / rows_per_graph.
 */
typedef struct AetherS2SGraphSummaryParams {
    const float* emb_w; const float* emb_b;                     /* particle_embedding [H][in], [H] */
    const float* gru_w_ih; const float* gru_w_hh;               /* rnn.weight_{ih,hh}_l0 [3H][H] */
    const float* gru_b_ih; const float* gru_b_hh;               /* [3H] */
    const float* pe;                                            /* [pe_len][in + H] */
    const float* gate_w0; const float* gate_b0; const float* gate_w2; const float* gate_b2;  /* [H][in+H], [H], [1][H], [1] */
    const float* nn_w0; const float* nn_b0; const float* nn_w2; const float* nn_b2;          /* [H][in+H], [H], [H][H], [H] */
} AetherS2SGraphSummaryParams;
size_t aether_s2s_graph_summary_workspace_bytes(int64_t batch, int num_objects, int timesteps, int input_size,
                                                int hidden);
int aether_s2s_graph_summary(const AetherS2SGraphSummaryParams* params, int64_t batch, int num_objects,
                             int timesteps, int input_size, int hidden, int pe_len, const float* x,
                             void* workspace, size_t workspace_bytes, float* summary, void* stream);

typedef struct AetherS2SFilmParams {
    const float* B;                                             /* coordinate_embedding.B [D][h/2] */
    const float* lin1_w; const float* lin1_b;                   /* [mh][h] */
    const float* lin2_w; const float* lin2_b;                   /* [mh][mh] */
    const float* lin3_w; const float* lin3_b;                   /* [D][mh] */
    /* film_{1,2}.{gamma,beta}.{0,2}: [mh][gh], [mh], [mh][mh], [mh]; index = 2 * (film - 1) + (beta ? 1 : 0) */
    const float* mod_w0[4]; const float* mod_b0[4]; const float* mod_w2[4]; const float* mod_b2[4];
} AetherS2SFilmParams;
size_t aether_s2s_film_modulation_bytes(int64_t batch, int mlp_hidden);
int aether_s2s_film_modulation(const AetherS2SFilmParams* params, int graph_hidden, int mlp_hidden, int64_t batch,
                               const float* summary, float* mod, size_t mod_bytes, void* stream);
size_t aether_s2s_film_field_workspace_bytes(int64_t n_points, int hidden, int mlp_hidden);
int aether_s2s_film_field(const AetherS2SFilmParams* params, int num_dims, int hidden, int mlp_hidden,
                          int64_t n_points, int64_t rows_per_graph, const float* x, int x_stride,
                          const float* mod, int64_t batch, void* workspace, size_t workspace_bytes,
                          float* field, void* stream);

/*
 * k-nearest-neighbour edge builder for variable-N scenes: replaces Encoder.knn_edges
 * (nn/dynamicvars/aether_dynamicvars.py:559-586; the same method in the other *_dynamicvars.py) and the send /
 * recv half of get_knn_graph_info (experiments/ind/single_ind_data.py:186-217).
 *   x        : float[n_scenes][n_objects][x_stride]   the first two columns are the 2-D position
 *   masks    : float[n_scenes][n_objects]             != 0: the object is present in the scene
 *   k        : neighbours per object (the reference: min(10, n_objects - 1)); 1 <= k <= 16
 *   send, recv : int64[capacity], capacity >= n_scenes * n_objects * k; the first totals[0] entries are
 *              written: scene by scene, object by object, nearest neighbour first, in the compacted
 *              numbering (present objects of all scenes counted consecutively).  As in the reference,
 *              `send` is the querying object and `recv` its neighbour.
 *   scene_edges, scene_nodes : int64[n_scenes]  edges / present objects per scene
 *   totals   : int64[2] = {edges, present objects}
 * Equal distances keep index order (torch.topk leaves that order unspecified).  Stream-ordered, no host
 * synchronisation; read totals[0] after the stream to size the result.
 */
size_t aether_knn_workspace_bytes(int64_t n_scenes, int n_objects, int k);
int aether_knn_edges(const float* x, int x_stride, const float* masks, int64_t n_scenes, int n_objects, int k,
                     int64_t* send, int64_t* recv, int64_t* scene_edges, int64_t* scene_nodes, int64_t* totals,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * On-device data side: the simulators that generate the reference's datasets, batched over
 * simulations (fp64, as numpy).  The random draws (charges, initial state, field sources, observation noise)
 * stay with the caller's numpy generators, so a dataset is reproduced draw for draw.
 *
 * aether_sim_electrostatic replaces the integration of ElectrostaticFieldSim.sample_trajectory
 * (experiments/electrostatic/dataset/electrostatic_field_sim.py:108-163): Coulomb forces between all balls
 * (n_balls moving, then total_balls - n_balls static field sources), force capped at max_F, leap-frog.
 *   loc0, vel0 : double[n_sims][total_balls][dim]   initial state (:96-104; static velocities 0)
 *   charges    : double[n_sims][total_balls]        (:79-92)
 *   loc, vel   : double[n_sims][T/sample_freq - 1][total_balls][dim]   saved frames (:139-142; static rows constant)
 *   maxed_out  : int64[n_sims]   number of capped forces (the count the reference prints, :168)
 * aether_sim_gravitational replaces GravitationalFieldSim.sample_trajectory's loop
 * (experiments/gravitational/dataset/gravitational_field_sim.py:99-125): softened gravity, kick-drift-kick.
 *   pos0, vel0 : double[n_sims][total_balls][dim] (velocities in the centre-of-mass frame, :96); mass [n_sims][total_balls]
 *   pos, vel, force : double[n_sims][T/sample_freq][total_balls][dim]
 * total_balls <= 64, dim 2 or 3.  Stream-ordered.
 */
int aether_sim_electrostatic(const double* loc0, const double* vel0, const double* charges, int64_t n_sims,
                             int n_balls, int total_balls, int dim, int T, int sample_freq,
                             double interaction_strength, double delta_T, double max_F, double* loc, double* vel,
                             int64_t* maxed_out, void* stream);
/*
 * aether_sim_charged: the n-body simulators of the state2state runner's data sets
 * (experiments/lorentz/dataset/synthetic_sim.py: ChargedParticlesSim :221-300, GravitySim :375-460, DynamicSim
 * :536-622): 3-D Coulomb forces between n_balls moving charges, squared distances |a|^2 + |b|^2 - 2 a.b + 1e-6, plus
 * ext_mode 0: nothing ('charged'), 1: the constant force ext ('static': (0, 0, 0.098)), 2: the Lorentz force
 * q (v x ext) ('dynamic': ext = (0.5, 0.5, 0.5)), 3: a fixed charge at ext, ext_strength q (x - ext) / |x - ext|^3
 * (FixCharge :626-790: ext = (10, 10, 10), ext_strength 0.1).  pair != NULL (double[n_sims][n_balls][n_balls]): the pair
 * force is -interaction_strength * pair_ij (x_i - x_j) instead of Coulomb's and charges may be NULL (SpringSim :6-146).
 * Every force component is clipped to +-max_F; leap-frog.
 *   loc0, vel0 : double[n_sims][3][n_balls] (the reference's layout);  charges : double[n_sims][n_balls]
 *   ext        : HOST pointer to 3 doubles (read during the call; may be NULL for ext_mode 0)
 *   loc, vel   : double[n_sims][T/sample_freq - 1][3][n_balls]
 */
int aether_sim_charged(const double* loc0, const double* vel0, const double* charges, const double* pair, int64_t n_sims,
                       int n_balls, int T, int sample_freq, double interaction_strength, double delta_T, double max_F,
                       int ext_mode, const double* ext, double ext_strength, double* loc, double* vel, void* stream);
int aether_sim_gravitational(const double* pos0, const double* vel0, const double* mass, int64_t n_sims, int n_balls,
                             int total_balls, int dim, int T, int sample_freq, double interaction_strength,
                             double dt, double softening, double* pos, double* vel, double* force, void* stream);

/*
 * Variable-N models, one step of the decoder (second half): replaces Decoder.forward of
 * nn/dynamicvars/aether_dynamicvars.py:775-870 (2-D, one scene) on the objects present in the scene.
 * Differences from aether_s2s_decoder_step: rel_feat = the canonical state only (canonicalize_augmented_inputs,
 * 6 columns), messages from the present state through one AnisotropicEdgeFilter per edge type
 * (nn/nn/anisotropic_filter.py:34-40 with a ReLU hidden layer, in_size 15 = 9 edge features + 6 receiver columns)
 * followed by ReLU, hidden messages divided by the number of used edge types (:801-814), both aggregations are
 * sums over the caller's lists divided by agg_div (the reference: edge2node_inds rows, / (num_vars - 1), :816-819).
 *   inputs [n][4], field [n][2], hidden_in [n][h]: the present objects, compacted; edge_w [e][K]
 *   edge_state : float[m][6] rows [pos | vel | field] indexed by send / recv for the edge features; NULL = the
 *                compacted [inputs | field].  (The reference indexes the un-compacted array with compacted
 *                indices, :823 -- pass that array to reproduce it when objects are missing.)
 *   send, recv : int64[e] compacted object indices; agg_order int64[*], agg_rowptr int64[n + 1]: the edges summed
 *                into each object (the reference's edge2node_inds, flattened, with rowptr[i] = i * k)
 *   outputs [n][4], hidden_out [n][h];  h % 128 == 0
 */
typedef struct AetherDynDecoderParams {
    const float* msg_fc1_w[4]; const float* msg_fc1_b[4];    /* [h][2h] (receiver | sender halves), [h] */
    const float* msg_fc2_w[4]; const float* msg_fc2_b[4];    /* [h][h], [h] */
    const float* hidden_r_w; const float* hidden_i_w; const float* hidden_h_w;      /* [h][h], no bias */
    const float* input_r_w; const float* input_r_b;
    const float* input_i_w; const float* input_i_b;
    const float* input_n_w; const float* input_n_b;          /* [h][6], [h] */
    const float* present_r_w; const float* present_r_b;
    const float* present_i_w; const float* present_i_b;
    const float* present_n_w; const float* present_n_b;      /* [h][h], [h] */
    const float* out1_w; const float* out1_b; const float* out2_w; const float* out2_b;   /* out_fc1, out_fc2: [h][h] */
    const float* out3_w; const float* out3_b;                /* out_fc3: [4][h] */
    const float* filt_w0[4]; const float* filt_b0[4];        /* edge_filter.k.edge_filter.0: [h][3], [h] */
    const float* filt_w2[4]; const float* filt_b2[4];        /* edge_filter.k.edge_filter.2: [15 h][h], [15 h] */
    const void* filt_image[4];   /* aether_s2s_filter_prepare(filt_w2[k], 15, h, ..) of the current values, or NULL (built per call) */
} AetherDynDecoderParams;
size_t aether_dyn_decoder_workspace_bytes(int hidden, int64_t n_nodes, int64_t n_edges);
int aether_dyn_decoder_step(const AetherDynDecoderParams* params, int hidden, int num_edge_types, int skip_first,
                            int polar, int64_t n_nodes, int64_t n_edges, const float* inputs, const float* hidden_in,
                            const float* edge_w, const float* field, const float* edge_state, const int64_t* send,
                            const int64_t* recv, const int64_t* agg_order, const int64_t* agg_rowptr, float agg_div,
                            void* workspace, size_t workspace_bytes, float* outputs, float* hidden_out, void* stream);
/*
 * The same step for SEVERAL scenes at once -- what the reference cannot do (its variable-N models raise on batch > 1,
 * aether_dynamicvars.py:588-591): the present objects of all scenes are concatenated (n_nodes rows), `send` / `recv` /
 * `agg_order` are in that concatenated numbering (no edge crosses scenes), and the two per-scene quantities of the
 * single-scene call become arrays:
 *   agg_div_node : float[n_nodes], the divisor of each object's sums (the reference: its scene's num_vars - 1);
 *                  NULL = the scalar agg_div for everyone
 *   state_send, state_recv : int64[e] rows of `edge_state` the edge features read (NULL = send / recv).  The reference
 *                  indexes its scene's UN-compacted state with compacted indices (:823); with several scenes that is
 *                  scene_base_of_uncompacted_rows + compacted index within the scene, which differs from `send`.
 * aether_dyn_decoder_step is this function with both NULL.
 */
int aether_dyn_decoder_step_batched(const AetherDynDecoderParams* params, int hidden, int num_edge_types, int skip_first,
                                    int polar, int64_t n_nodes, int64_t n_edges, const float* inputs,
                                    const float* hidden_in, const float* edge_w, const float* field,
                                    const float* edge_state, const int64_t* state_send, const int64_t* state_recv,
                                    const int64_t* send, const int64_t* recv, const int64_t* agg_order,
                                    const int64_t* agg_rowptr, float agg_div, const float* agg_div_node, void* workspace,
                                    size_t workspace_bytes, float* outputs, float* hidden_out, void* stream);

/*
 * Variable-N models, one step of the encoder's prior and the field query: with
 * aether_knn_edges, aether_dyn_decoder_step and aether_s2s_gumbel_hard the whole prediction step of
 * AetherDynamicVars.predict_future (nn/dynamicvars/aether_dynamicvars.py:245-273).
 *
 * aether_dyn_prior_step replaces Encoder.compute_feat_transform for one time step (:505-557) + the LSTM / prior
 * half of Encoder.single_step_forward (:688-696) on the present objects of a scene: canonical states, edge features
 * [9 | receiver's canonical state 6], anisotropic filter (ReLU hidden layer), SUM over the in-edges of an object +
 * mlp1(canonical state), mlp3, [x_send | x_recv | edge] -> mlp4 (RefNRIMLP in eval mode; bn pointers may all be
 * NULL: no_encoder_bn), one LSTM step per edge, prior_fc_out.
 *   inputs [n][4], field [n][2] : the present objects, compacted;  send, recv int64[e] (compacted; features of
 *   send -> recv in recv's frame, as aether_knn_edges lists them);  order, rowptr : edges grouped by recv (CSR)
 *   h0, c0 [e][R] : the LSTM state rows of these edges (the reference keeps one slot per fully connected pair and
 *   gathers / scatters them, :684-694: that indexing stays with the caller);  logits [e][K], h1, c1 [e][R]
 * aether_dyn_field replaces AetherDynamicVars.predict_field (:64-79) on the present objects: Fourier features of the
 * position (hidden/2 frequencies), angular_embedding Linear(2, hidden) of the unit velocity, field_net
 * Linear(2 hidden, hidden)-SiLU-Linear-SiLU-Linear(hidden, 2).   x [n][4] -> field [n][2]
 */
typedef struct AetherDynPriorParams {
    const float* mlp1_w0; const float* mlp1_b0; const float* mlp1_w3; const float* mlp1_b3;       /* [h][6], [h][h] */
    const float* mlp1_bn_w; const float* mlp1_bn_b; const float* mlp1_bn_mean; const float* mlp1_bn_var;
    const float* mlp3_w0; const float* mlp3_b0; const float* mlp3_w3; const float* mlp3_b3;       /* [h][h] */
    const float* mlp3_bn_w; const float* mlp3_bn_b; const float* mlp3_bn_mean; const float* mlp3_bn_var;
    const float* mlp4_w0; const float* mlp4_b0; const float* mlp4_w3; const float* mlp4_b3;       /* [h][3h], [h][h] */
    const float* mlp4_bn_w; const float* mlp4_bn_b; const float* mlp4_bn_mean; const float* mlp4_bn_var;
    const float* lstm_w_ih; const float* lstm_w_hh; const float* lstm_b_ih; const float* lstm_b_hh; /* [4R][h], [4R][R] */
    const float* prior_w[4]; const float* prior_b[4];                                            /* prior_fc_out */
    const float* filt_w0; const float* filt_b0;                                                  /* [h][3] */
    const float* filt_w2; const float* filt_b2;                                                  /* [15 h][h] */
    const void* filt_image;   /* aether_s2s_filter_prepare(filt_w2, 15, h, ..) of the current values, or NULL (built per call) */
} AetherDynPriorParams;
size_t aether_dyn_prior_workspace_bytes(int hidden, int rnn_hidden, int prior_hidden, int64_t n_nodes, int64_t n_edges);
int aether_dyn_prior_step(const AetherDynPriorParams* params, int hidden, int rnn_hidden, int prior_layers,
                          int prior_hidden, int num_edge_types, int polar, int64_t n_nodes, int64_t n_edges,
                          const float* inputs, const float* field, const float* h0, const float* c0,
                          const int64_t* send, const int64_t* recv, const int64_t* order, const int64_t* rowptr,
                          void* workspace, size_t workspace_bytes, float* logits, float* h1, float* c1, void* stream);
typedef struct AetherDynFieldQueryParams {
    const float* B;                                            /* coordinate_embedding.B [2][hidden/2] */
    const float* ang_w; const float* ang_b;                    /* angular_embedding [hidden][2], [hidden] */
    const float* w0; const float* b0; const float* w2; const float* b2; const float* w4; const float* b4;
                                                               /* field_net.{0,2,4}: [h][2h], [h][h], [2][h] */
} AetherDynFieldQueryParams;
size_t aether_dyn_field_workspace_bytes(int64_t n_points, int hidden);
int aether_dyn_field(const AetherDynFieldQueryParams* params, int hidden, int64_t n_points, const float* x,
                     void* workspace, size_t workspace_bytes, float* field, void* stream);

/*
 * One prediction step of the variable-N model as ONE call (round 3): the body of AetherDynamicVars.predict_future's loop
 * (nn/dynamicvars/aether_dynamicvars.py:245-273) for one scene -- field at the present objects (:64-79), the encoder's
 * own kNN graph + prior step with the per-pair LSTM slots gathered and scattered (:672-699), the hard Gumbel sample
 * (:133-141) and the decoder step (:775-870) -- with the index work between the stages (mask -> rows, receiver CSR of the
 * kNN graph, slot arithmetic, scatter of the results) in seven small kernels instead of ~60 framework launches.
 *   state [n_objects_max][4] : the scene's rows (observed objects: ground truth, the others: the last prediction, :264)
 *   mask  [n_objects_max]    : non-zero = present; n_present = their number, known to the caller (len(node_inds)).
 *                              A mask that disagrees makes the step's outputs NaN and the next call return AETHER_EHIP.
 *   node_inds int64[n_present] or NULL (= the mask's non-zero rows, ascending): rows of the present objects
 *   graph_send, graph_recv int64[n_edges], edge2node int64[n_present][in_degree] : the data set's graph_info of the step, in
 *                              the numbering of the present objects (single_ind_data.py:186-217);
 *                              n_edges = n_present * min(knn_k, n_present - 1), as the encoder's own graph has
 *   prior_h, prior_c [n_objects_max (n_objects_max - 1)][rnn_hidden] : one LSTM slot per ordered pair, UPDATED IN PLACE
 *   decoder_hidden [n_objects_max][decoder_hidden] : UPDATED IN PLACE at the present rows
 *   uniform [n_edges][K] : the U(0,1) draw of gumbel_softmax;  edge_types [n_edges][K] (one-hot, out) or NULL
 *   prediction [n_objects_max][4] : the decoder's output at the present rows, zero elsewhere (:866-868)
 * Every stage is the entry point documented above (same kernels, same order): results equal the staged calls bit for bit.
 */
typedef struct AetherDynStepConfig {
    int field_hidden, encoder_hidden, rnn_hidden, prior_layers, prior_hidden, num_edge_types, decoder_hidden;
    int skip_first, encoder_polar, decoder_polar, knn_k;
    float gumbel_tau;
} AetherDynStepConfig;
size_t aether_dyn_step_workspace_bytes(const AetherDynStepConfig* config, int n_objects_max, int64_t n_present,
                                       int64_t n_edges);
int aether_dyn_step(const AetherDynFieldQueryParams* field_params, const AetherDynPriorParams* prior_params,
                    const AetherDynDecoderParams* decoder_params, const AetherDynStepConfig* config, int n_objects_max,
                    int64_t n_present, int64_t n_edges, const float* state, const float* mask, const int64_t* node_inds,
                    const int64_t* graph_send, const int64_t* graph_recv, const int64_t* edge2node, int in_degree,
                    float* prior_h, float* prior_c, float* decoder_hidden, const float* uniform, float* prediction,
                    float* edge_types, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The prediction loop around aether_dyn_step (AetherDynamicVars.predict_future, :245-273) for one scene: for step t the
 * state is  observed * inputs[t] + (1 - observed) * (t == 0 ? inputs[0] : predictions[t - 1])  with observed =
 * burn_in_masks[t] (:264), then one aether_dyn_step.  Device arrays: inputs [n_steps + 1][n_objects_max][4] (row
 * n_steps is not read), masks, burn_in_masks [n_steps][n_objects_max] (at least), predictions [n_steps][n_objects_max][4];
 * prior_h / prior_c / decoder_hidden as in aether_dyn_step (initial state in, final state out).  HOST arrays of length
 * n_steps: n_present, n_edges (entries of the step's graph_send / graph_recv / uniform rows; as in aether_dyn_step it has
 * to equal n_present * min(knn_k, n_present - 1), the size of the encoder's own kNN graph), in_degree, and the per-step
 * device pointers node_inds (or NULL, or NULL entries), graph_send, graph_recv, edge2node, uniform.  Steps without present
 * objects yield zeros and leave the states alone (:841-843); a step with exactly one is refused (the reference fails there
 * too).  EVERY step is validated on the host before the first launch: a refused call has queued nothing and touched no
 * state.  No host synchronisation: the whole loop is queued on `stream` (52 launches per step).  A mask that disagrees
 * with n_present poisons that step's outputs AND the LSTM state rows it would have written with NaN (async error word).
 */
size_t aether_dyn_rollout_workspace_bytes(const AetherDynStepConfig* config, int n_objects_max, int n_steps,
                                          const int64_t* n_present, const int64_t* n_edges);
int aether_dyn_rollout(const AetherDynFieldQueryParams* field_params, const AetherDynPriorParams* prior_params,
                       const AetherDynDecoderParams* decoder_params, const AetherDynStepConfig* config, int n_objects_max,
                       int n_steps, const float* inputs, const float* masks, const float* burn_in_masks,
                       const int64_t* n_present, const int64_t* n_edges, const int64_t* const* node_inds,
                       const int64_t* const* graph_send, const int64_t* const* graph_recv, const int64_t* const* edge2node,
                       const int* in_degree, const float* const* uniform, float* prior_h, float* prior_c,
                       float* decoder_hidden, float* predictions, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same step and loop for B scenes per call (round 4; BASELINE config 4 is batch = 64; the reference's models raise on
 * batch > 1, aether_dynamicvars.py:588-591).  The present objects of all scenes are numbered consecutively, scene-major;
 * every stage runs once over all of them (same stage kernels as the single-scene call).  HOST arrays of length n_scenes:
 * n_present (0, or 2 .. n_objects_max per scene), n_edges (= n_present * min(knn_k, n_present - 1) per scene), in_degree
 * (1 .. 255).  Device arrays: state [B][n_max][4], mask [B][n_max], decoder_hidden [B][n_max][hd], prediction [B][n_max][4],
 * prior_h / prior_c [B * n_max (n_max - 1)][R] (scene b's slots from b * n_max (n_max - 1) on); node_inds (or NULL),
 * graph_send, graph_recv, edge2node, uniform: the scenes' arrays CONCATENATED in scene order, indices scene-local exactly
 * as the single-scene call takes them.  n_scenes <= 256.  A scene whose mask disagrees with its n_present poisons the
 * step's outputs with NaN (async error word).  aether_dyn_rollout_batched: every per-step tensor TIME-MAJOR -- inputs
 * [n_steps + 1][B][n_max][4], masks / burn_in_masks [n_steps][B][n_max], predictions [n_steps][B][n_max][4], host size arrays
 * [n_steps][B], per-step device pointers as in aether_dyn_rollout; everything is validated before the first launch.
 */
size_t aether_dyn_step_batched_workspace_bytes(const AetherDynStepConfig* config, int n_scenes, int n_objects_max,
                                               const int64_t* n_present, const int64_t* n_edges, const int* in_degree);
int aether_dyn_step_batched(const AetherDynFieldQueryParams* field_params, const AetherDynPriorParams* prior_params,
                            const AetherDynDecoderParams* decoder_params, const AetherDynStepConfig* config, int n_scenes,
                            int n_objects_max, const int64_t* n_present, const int64_t* n_edges, const int* in_degree,
                            const float* state, const float* mask, const int64_t* node_inds, const int64_t* graph_send,
                            const int64_t* graph_recv, const int64_t* edge2node, float* prior_h, float* prior_c,
                            float* decoder_hidden, const float* uniform, float* prediction, float* edge_types,
                            void* workspace, size_t workspace_bytes, void* stream);
size_t aether_dyn_rollout_batched_workspace_bytes(const AetherDynStepConfig* config, int n_scenes, int n_objects_max, int n_steps,
                                                  const int64_t* n_present, const int64_t* n_edges, const int* in_degree);
int aether_dyn_rollout_batched(const AetherDynFieldQueryParams* field_params, const AetherDynPriorParams* prior_params,
                               const AetherDynDecoderParams* decoder_params, const AetherDynStepConfig* config, int n_scenes,
                               int n_objects_max, int n_steps, const float* inputs, const float* masks,
                               const float* burn_in_masks, const int64_t* n_present, const int64_t* n_edges,
                               const int* in_degree, const int64_t* const* node_inds, const int64_t* const* graph_send,
                               const int64_t* const* graph_recv, const int64_t* const* edge2node, const float* const* uniform,
                               float* prior_h, float* prior_c, float* decoder_hidden, float* predictions, void* workspace,
                               size_t workspace_bytes, void* stream);

/*
 * The rest of the runner's training step (experiments/lorentz/main.py:86,164,289-292): nn.MSELoss with the seed of its
 * backward, and optim.AdamW over every parameter tensor -- one launch each (torch: 4 + 3).
 *
 * aether_mse_loss_grad: *loss = mean((pred - target)^2), dpred[i] = 2 (pred[i] - target[i]) / n.  scratch: at least
 *   aether_mse_scratch_bytes() bytes of device memory, ZERO before the first call (the kernel re-arms it), not shared by
 *   launches that may run concurrently.  Partial sums are added in a fixed order (bit-stable).
 * aether_adamw_step: torch.optim.AdamW(betas, eps, weight_decay; amsgrad = maximize = False) for n_tensors fp32
 *   tensors (64 per launch).  `step` (device float: steps taken so far, incremented by the launch), `lr` (device float) and `counter`
 *   (device int, zero before the first call) live in device memory, so a captured launch follows the step counter and a
 *   learning-rate schedule.  fp32 arithmetic; the bias corrections 1 - beta^t as -expm1(t log beta), betas in (0, 1).
 *   `grad_scale` multiplies every gradient as it is read (1.0: torch's AdamW exactly): a data-parallel step passes
 *   1 / world_size after the SUM all-reduce of the flat gradient buffer, so the mean needs no launch of its own.
 */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} AetherAdamWTensor;
size_t aether_mse_scratch_bytes(void);
int aether_mse_loss_grad(const float* pred, const float* target, int64_t n, float* loss, float* dpred, void* scratch,
                         size_t scratch_bytes, void* stream);
int aether_adamw_step(const AetherAdamWTensor* tensors, int n_tensors, float* step, const float* lr, int* counter,
                      double beta1, double beta2, double eps, double weight_decay, double grad_scale, void* stream);

/*
 * Tuning knobs (process-wide; not thread-safe): "fused_split" 0|1 (two workgroups per group when
 * there are fewer groups than half the CUs; read by aether_graph_build), "fused_pair_stride" 1|8 (index distance of the
 * two workgroups of a split group: 8, the default, puts both on one XCD; read by aether_graph_build),
 * "outer_defer_max_edges" n (aether_backward keeps every layer's weight-gradient operands and
 * multiplies them in one launch when n_edges <= n, default 2^20; changes aether_workspace_bytes),
 * "filter_wg_target" n (the anisotropic-filter GEMM of the seq2seq / variable-N steps splits its k-groups, up to
 * 16 ways, until it launches at least n workgroups; default 768, measured best at 2,560 edges; changes the prior /
 * decoder workspace sizes), "linear_small_wgs" n (dense layers of the seq2seq / variable-N steps whose 64 x 32-per-wave
 * tiling would launch fewer than n workgroups use 16 x 32 blocks per wave instead: four times the waves for the
 * 5-object graphs; default 128), "linear_kwaves" 1|4 (when even those blocks are few, the four waves of a workgroup
 * share one block and split its k-groups, partial sums added in wave order; default 4, 1 turns it off),
 * "gemm_split" 0|1|2|3 (dense layers of the fused seq2seq step from 128 workgroups on: 1, the default, multiplies
 * fp16 x 2 pieces on the 16-bit matrix pipe and picks the kernel structure per launch -- both operands through an
 * LDS-DMA ring up to 256 workgroups, X rows in registers above; 2 / 3 force the second / the first structure for
 * every launch; 0 sends these layers through the fp32-MFMA job kernel).
 */
int aether_set_option(const char* name, int value);

/*
 * A kernel cannot return a status.  The one bounded wait in the library -- a split-mode workgroup of the fused
 * kernel polling for its partner's rows -- sets a word in host-mapped memory when it gives up (partner not
 * resident within ~seconds); the results of that launch are then invalid.  Every launching entry point checks and
 * clears the word first (so the NEXT call returns AETHER_EHIP), and this function does so on request, e.g. after
 * synchronising the stream.  Returns AETHER_OK or AETHER_EHIP (message in aether_last_error).
 */
int aether_check_async_error(void);

/*
 * Per-kernel timing for bench.py's roofline line: when enabled, every launch made by
 * aether_forward is bracketed by a pair of HIP events on the launch stream.
 * aether_profile_read synchronises those events, adds up elapsed milliseconds and launch
 * counts per kernel id (0 .. aether_profile_kernels()-1) and clears the record.
 * Not thread-safe; measurement only (the event records perturb back-to-back launches).
 */
int aether_profile_enable(int on);
int aether_profile_kernels(void);
const char* aether_profile_kernel_name(int id);
int aether_profile_read(double* total_ms, int64_t* launches, int n);

#ifdef __cplusplus
}
#endif
#endif /* AETHER_HIP_H */
