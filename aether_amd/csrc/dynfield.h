// Dynamic-field variant of the state2state model: LatentFieldNetwork.forward,
// nn/state2state/dynamic_field_aether.py:31-48 --
//   graph summary  (GraphSummary, nn/state2state/graph_pool.py:7-29; torch_geometric AttentionalAggregation:
//                   softmax over the nodes of a graph of gate_nn([p | v]) weighting nn([p | v]))
//   FiLM field net (FilmedNetwork / FiLM, nn/state2state/film.py:5-60) on [p | v | class embedding].
// Everything is 32 wide and per graph / per node: one workgroup per graph, thread-level FMAs, the
// parameters (10 K floats) come through L1.  The result feeds aether_forward_field, which runs the
// unchanged frames / GNN / globaliser kernels.
#pragma once
#include "common.h"

struct AetherDynFieldParams;      // include/aether_hip.h

namespace {

constexpr int DYNFIELD_MAX_NODES = 2048;
constexpr int DFH = 32;           // field_hidden_size = summary_dim (dynamic_field_aether.py:68-72)

// One workgroup per graph.  Every layer is evaluated by (node, unit) threads over chunks of 64 nodes
// staged in LDS (a thread per node running whole 32-wide layers from private arrays went through scratch
// memory: 250 us at D = 3, N = 20).  Softmax in two passes over the nodes (gates, then the weighted sum
// with the hidden layer of `nn` recomputed per chunk); every reduction has a fixed order.
template <int D>
__global__ void __launch_bounds__(256)
k_dynfield(AetherDynFieldParams P, const float* __restrict__ x, const float* __restrict__ vel,
           const float* __restrict__ charges, float* __restrict__ field, int N) {
    constexpr int XI = 2 * D, FI = 2 * D + 16, CH = 64, LH = DFH + 1;
    __shared__ float gate[DYNFIELD_MAX_NODES];
    __shared__ float red[256];
    __shared__ float part[8][DFH];
    __shared__ float summary[DFH], hbuf[2][DFH], mod[2][2 * DFH];
    __shared__ float zs[CH][FI];            // chunk inputs: [p | v | class embedding]
    __shared__ float ya[CH][LH], yb[CH][LH];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * N;
    auto silu_f = [](float s) { return s / (1.0f + expf(-s)); };
    auto stage_x = [&](int c0, int cnt) {                   // zs[n][0 .. 2D) = [p | v]
        for (int idx = tid; idx < cnt * XI; idx += 256) {
            const int n = idx / XI, k = idx - n * XI;
            zs[n][k] = k < D ? x[(base + c0 + n) * D + k] : vel[(base + c0 + n) * D + (k - D)];
        }
    };
    // hidden layer of a two-layer net on [p | v]: ya[n][o] = SiLU(b0[o] + w0[o] . zs[n][0 .. 2D))
    auto hidden = [&](const float* w0, const float* b0, int cnt) {
        for (int idx = tid; idx < cnt * DFH; idx += 256) {
            const int n = idx >> 5, o = idx & 31;
            float s = b0[o];
#pragma unroll
            for (int k = 0; k < XI; ++k) s = fmaf(w0[o * XI + k], zs[n][k], s);
            ya[n][o] = silu_f(s);
        }
    };
    // ---- pass 1: gates (gate_nn: Linear(2D, 32) - SiLU - Linear(32, 1)) and their maximum
    float lmax = -INFINITY;
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        __syncthreads();
        hidden(P.gate_w0, P.gate_b0, cnt);
        __syncthreads();
        if (tid < cnt) {
            float g = P.gate_b2[0];
            for (int o = 0; o < DFH; ++o) g = fmaf(P.gate_w2[o], ya[tid][o], g);
            gate[c0 + tid] = g;
            lmax = fmaxf(lmax, g);
        }
        __syncthreads();
    }
    red[tid] = lmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    const float gmax = red[0];
    __syncthreads();
    float lsum = 0.0f;
    for (int n = tid; n < N; n += 256) { const float e = expf(gate[n] - gmax); gate[n] = e; lsum += e; }
    red[tid] = lsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float denom = red[0] + 1e-16f;                    // torch_geometric.utils.softmax
    __syncthreads();
    // ---- pass 2: summary[c] = sum_n softmax_n * nn([p | v])_c; thread (c = tid & 31, slice = tid >> 5)
    {
        const int c = tid & 31, slice = tid >> 5;
        float acc = 0.0f;
        for (int c0 = 0; c0 < N; c0 += CH) {
            const int cnt = N - c0 < CH ? N - c0 : CH;
            stage_x(c0, cnt);
            __syncthreads();
            hidden(P.nn_w0, P.nn_b0, cnt);
            __syncthreads();
            for (int n = slice; n < cnt; n += 8) {
                float v = P.nn_b2[c];
                for (int o = 0; o < DFH; ++o) v = fmaf(P.nn_w2[c * DFH + o], ya[n][o], v);
                acc += (gate[c0 + n] / denom) * v;
            }
            __syncthreads();
        }
        part[slice][c] = acc;
    }
    __syncthreads();
    if (tid < DFH) {
        float s = 0.0f;
        for (int k = 0; k < 8; ++k) s += part[k][tid];
        summary[tid] = s;
    }
    __syncthreads();
    // ---- FiLM modulators (film.py:41-60): Linear(32,32) - SiLU - Linear(32,32) - SiLU - Linear(32,64), twice
    for (int f = 0; f < 2; ++f) {
        const float* w0 = f ? P.film2_w0 : P.film1_w0; const float* b0 = f ? P.film2_b0 : P.film1_b0;
        const float* w2 = f ? P.film2_w2 : P.film1_w2; const float* b2 = f ? P.film2_b2 : P.film1_b2;
        const float* w4 = f ? P.film2_w4 : P.film1_w4; const float* b4 = f ? P.film2_b4 : P.film1_b4;
        if (tid < DFH) {
            float s = b0[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w0[tid * DFH + k], summary[k], s);
            hbuf[0][tid] = silu_f(s);
        }
        __syncthreads();
        if (tid < DFH) {
            float s = b2[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w2[tid * DFH + k], hbuf[0][k], s);
            hbuf[1][tid] = silu_f(s);
        }
        __syncthreads();
        if (tid < 2 * DFH) {
            float s = b4[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w4[tid * DFH + k], hbuf[1][k], s);
            mod[f][tid] = s;                                // gamma = [0, 32), beta = [32, 64)
        }
        __syncthreads();
    }
    // ---- pass 3: FilmedNetwork per node (film.py:26-35)
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        for (int idx = tid; idx < cnt * 16; idx += 256) {
            const int n = idx >> 4, k = idx & 15;
            long ci = (long)(charges[base + c0 + n] + 1.0f);  // charge_to_index: (q + 1).long()
            ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
            zs[n][XI + k] = P.emb[ci * 16 + k];
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) {
            const int n = idx >> 5, o = idx & 31;
            float s = P.lin1_b[o];
#pragma unroll
            for (int k = 0; k < FI; ++k) s = fmaf(P.lin1_w[o * FI + k], zs[n][k], s);
            ya[n][o] = silu_f((1.0f + mod[0][o]) * s + mod[0][DFH + o]);
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) {
            const int n = idx >> 5, o = idx & 31;
            float s = P.lin2_b[o];
#pragma unroll
            for (int k = 0; k < DFH; ++k) s = fmaf(P.lin2_w[o * DFH + k], ya[n][k], s);
            yb[n][o] = silu_f((1.0f + mod[1][o]) * s + mod[1][DFH + o]);
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * D; idx += 256) {
            const int n = idx / D, d = idx - n * D;
            float s = P.lin3_b[d];
#pragma unroll
            for (int k = 0; k < DFH; ++k) s = fmaf(P.lin3_w[d * DFH + k], yb[n][k], s);
            field[(base + c0 + n) * D + d] = s;
        }
        __syncthreads();
    }
}

}  // namespace
