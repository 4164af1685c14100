// seq2seq dynamic-field variant: nn/seq2seq/dynamic_field_aether.py, the model of
// scripts/gravitational_field_3d_aether.sh.  It is the seq2seq Aether whose field query is conditioned on
// a summary of the burn-in trajectories:
//   GraphSummary.forward      nn/nn/graph_pool.py:50-71   (once per sequence)
//   FilmedNetwork / FiLM      nn/nn/filmed_network.py:27-35, nn/nn/film.py:53-60   (every step)
// All Linear layers run through k_s2s_linear (the FiLM modulation is its pre-activation epilogue, one
// (gamma, beta) row per graph); this file holds the element-wise pieces in between.
#pragma once
#include "common.h"

namespace {

// nn.GRU cell, gates (r, z, n): gi = W_ih y_t + b_ih for every time step at once ([S][T][3H], this step's
// row of sequence s at gi + s * gi_stride), gh = W_hh h + b_hh ([S][3H]);
// r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h.
__global__ void __launch_bounds__(256)
k_s2s_gru_gate(const float* __restrict__ gi, int64_t gi_stride, const float* __restrict__ gh, float* __restrict__ h,
               int H, int64_t S) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * H) return;
    const int64_t s = idx / H;
    const int c = (int)(idx - s * H);
    const float* a = gi + s * gi_stride;
    const float* b = gh + s * 3 * H;
    const float r = sigmoid1(a[c] + b[c]);
    const float z = sigmoid1(a[H + c] + b[H + c]);
    const float n = tanh1(a[2 * H + c] + r * b[2 * H + c]);
    h[idx] = (1.0f - z) * n + z * h[idx];
}

// a[(s, t)][:] = [x[(s, t)] | h_last[s]] + pe[t], zero padded to Kp columns (graph_pool.py:62-68)
__global__ void __launch_bounds__(256)
k_s2s_augment(const float* __restrict__ x, const float* __restrict__ hlast, const float* __restrict__ pe,
              float* __restrict__ a, int in, int H, int Kp, int T, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * Kp) return;
    const int64_t r = idx / Kp;
    const int c = (int)(idx - r * Kp);
    const int64_t s = r / T;
    const int t = (int)(r - s * T);
    const int d = in + H;
    float v = 0.0f;
    if (c < in) v = x[r * in + c] + pe[(size_t)t * d + c];
    else if (c < d) v = hlast[s * H + (c - in)] + pe[(size_t)t * d + c];
    a[idx] = v;
}

// torch_geometric AttentionalAggregation over the items of one graph (all (object, time) pairs):
// w_i = exp(g_i - max) / (sum_j exp(g_j - max) + 1e-16), out[c] = sum_i w_i V[i][c] in item order.
// grid = (graphs, ceil(H / 256)); every workgroup reduces the gate values itself (items are few).
__global__ void __launch_bounds__(256)
k_s2s_attn_pool(const float* __restrict__ gate, const float* __restrict__ V, float* __restrict__ out, int items,
                int H) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const float* g = gate + (size_t)blockIdx.x * items;
    float mx = -INFINITY;
    for (int i = tid; i < items; i += 256) mx = fmaxf(mx, g[i]);
    red[tid] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = fmaxf(red[tid], red[tid + w]);
        __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float sum = 0.0f;
    for (int i = tid; i < items; i += 256) sum += expf(g[i] - mx);
    red[tid] = sum;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const float inv = 1.0f / (red[0] + 1e-16f);
    const int c = (int)blockIdx.y * 256 + tid;
    if (c >= H) return;
    const float* v = V + (size_t)blockIdx.x * items * H + c;
    float acc = 0.0f;
    for (int i = 0; i < items; ++i) acc = fmaf(expf(g[i] - mx) * inv, v[(size_t)i * H], acc);
    out[(size_t)blockIdx.x * H + c] = acc;
}

}  // namespace
