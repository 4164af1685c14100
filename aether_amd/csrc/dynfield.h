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

// ------------------------------------------------------------------------------------------------------
// Backward of LatentFieldNetwork (training of the dynamic-field variant): given dL/dfield per node, the
// gradients of all 27 parameter tensors.  One workgroup per graph recomputes the forward quantities it needs
// (chunks of 32 nodes in LDS) and adds its contributions to its own row of `partial` [n_graphs][total] --
// every entry is owned by one thread and summed over the nodes in order -- and k_dynfield_reduce adds the rows
// over the graphs in order: no atomics, bit-reproducible.
// Flat layout of a row = the field order of AetherDynFieldParams.
template <int D> struct DynOff {
    static constexpr int XI = 2 * D, FI = 2 * D + 16, H = DFH;
    static constexpr int gate_w0 = 0, gate_b0 = gate_w0 + H * XI, gate_w2 = gate_b0 + H, gate_b2 = gate_w2 + H;
    static constexpr int nn_w0 = gate_b2 + 1, nn_b0 = nn_w0 + H * XI, nn_w2 = nn_b0 + H, nn_b2 = nn_w2 + H * H;
    static constexpr int lin1_w = nn_b2 + H, lin1_b = lin1_w + H * FI, lin2_w = lin1_b + H, lin2_b = lin2_w + H * H;
    static constexpr int lin3_w = lin2_b + H, lin3_b = lin3_w + D * H;
    static constexpr int film1 = lin3_b + D;              // w0 [H][H], b0 [H], w2 [H][H], b2 [H], w4 [2H][H], b4 [2H]
    static constexpr int F_w0 = 0, F_b0 = F_w0 + H * H, F_w2 = F_b0 + H, F_b2 = F_w2 + H * H, F_w4 = F_b2 + H,
                         F_b4 = F_w4 + 2 * H * H, FILM = F_b4 + 2 * H;
    static constexpr int film2 = film1 + FILM, emb = film2 + FILM, total = emb + 3 * 16;
};

__device__ __forceinline__ float dsilu_f(float s) {       // d/ds [s sigmoid(s)]
    const float sg = 1.0f / (1.0f + expf(-s));
    return sg * (1.0f + s * (1.0f - sg));
}

template <int D>
__global__ void __launch_bounds__(256)
kb_dynfield(AetherDynFieldParams P, const float* __restrict__ x, const float* __restrict__ vel,
            const float* __restrict__ charges, const float* __restrict__ gfield, float* __restrict__ partial, int N,
            float* __restrict__ gz = nullptr /* [nodes][2D]: dL/d[p | v] through the field, or null (round 3) */) {
    using OF = DynOff<D>;
    constexpr int XI = 2 * D, FI = 2 * D + 16, CH = 32, LH = DFH + 1;
    __shared__ float gate[DYNFIELD_MAX_NODES];              // softmax weights w_n
    __shared__ float dwn[DYNFIELD_MAX_NODES];               // dL/dw_n, then dL/dg_n
    __shared__ float red[256];
    __shared__ float part[8][DFH];
    __shared__ float summary[DFH], dsum[DFH], qv[DFH], ubar[DFH];
    __shared__ float fpre[2][2][DFH], fh[2][2][DFH], mod[2][2 * DFH], dmod[2][2 * DFH], dp1[DFH], dp0[DFH];
    __shared__ float zs[CH][FI], gfs[CH][4];
    __shared__ float pa[CH][LH], ya[CH][LH], pb[CH][LH], yb[CH][LH], da[CH][LH], db[CH][LH];
    __shared__ int cls[CH];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * N;
    float* g = partial + (size_t)blockIdx.x * OF::total;
    auto silu_f = [](float s) { return s / (1.0f + expf(-s)); };
    for (int e = tid; e < OF::total; e += 256) g[e] = 0.0f;
    if (tid < 2 * DFH) { dmod[0][tid] = 0.0f; dmod[1][tid] = 0.0f; }
    if (tid < DFH) { dsum[tid] = 0.0f; ubar[tid] = 0.0f; }
    auto stage_x = [&](int c0, int cnt) {
        for (int idx = tid; idx < cnt * XI; idx += 256) {
            const int n = idx / XI, k = idx - n * XI;
            zs[n][k] = k < D ? x[(base + c0 + n) * D + k] : vel[(base + c0 + n) * D + (k - D)];
        }
    };
    // pre[n][o] = b0[o] + w0[o] . zs[n][0 .. 2D), act[n][o] = SiLU(pre)
    auto hidden = [&](const float* w0, const float* b0, int cnt, float (*pre)[LH], float (*act)[LH]) {
        for (int idx = tid; idx < cnt * DFH; idx += 256) {
            const int n = idx >> 5, o = idx & 31;
            float s = b0[o];
#pragma unroll
            for (int k = 0; k < XI; ++k) s = fmaf(w0[o * XI + k], zs[n][k], s);
            pre[n][o] = s;
            act[n][o] = silu_f(s);
        }
    };
    // dst[o][k] += sum_n A[n][o] B[n][k]  /  dst[o] += sum_n A[n][o]   (entry owned by one thread, nodes in order)
    auto outer_acc = [&](int off, int O, int K, const float* A, int lda, const float* B, int ldb, int cnt) {
        for (int e = tid; e < O * K; e += 256) {
            const int o = e / K, k = e - o * K;
            float s = 0.0f;
            for (int n = 0; n < cnt; ++n) s = fmaf(A[n * lda + o], B[n * ldb + k], s);
            g[off + e] += s;
        }
    };
    auto col_acc = [&](int off, int O, const float* A, int lda, int cnt) {
        for (int o = tid; o < O; o += 256) {
            float s = 0.0f;
            for (int n = 0; n < cnt; ++n) s += A[n * lda + o];
            g[off + o] += s;
        }
    };
    __syncthreads();
    // ================= A. forward quantities: softmax weights, summary, FiLM modulators (as k_dynfield)
    float lmax = -INFINITY;
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        __syncthreads();
        hidden(P.gate_w0, P.gate_b0, cnt, pa, ya);
        __syncthreads();
        if (tid < cnt) {
            float gg = P.gate_b2[0];
            for (int o = 0; o < DFH; ++o) gg = fmaf(P.gate_w2[o], ya[tid][o], gg);
            gate[c0 + tid] = gg;
            lmax = fmaxf(lmax, gg);
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
    const float denom = red[0] + 1e-16f;
    __syncthreads();
    for (int n = tid; n < N; n += 256) gate[n] = gate[n] / denom;          // w_n
    __syncthreads();
    {
        const int c = tid & 31, slice = tid >> 5;
        float acc = 0.0f;
        for (int c0 = 0; c0 < N; c0 += CH) {
            const int cnt = N - c0 < CH ? N - c0 : CH;
            stage_x(c0, cnt);
            __syncthreads();
            hidden(P.nn_w0, P.nn_b0, cnt, pa, ya);
            __syncthreads();
            for (int n = slice; n < cnt; n += 8) {
                float v = P.nn_b2[c];
                for (int o = 0; o < DFH; ++o) v = fmaf(P.nn_w2[c * DFH + o], ya[n][o], v);
                acc += gate[c0 + n] * v;
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
    for (int f = 0; f < 2; ++f) {
        const float* w0 = f ? P.film2_w0 : P.film1_w0; const float* b0 = f ? P.film2_b0 : P.film1_b0;
        const float* w2 = f ? P.film2_w2 : P.film1_w2; const float* b2 = f ? P.film2_b2 : P.film1_b2;
        const float* w4 = f ? P.film2_w4 : P.film1_w4; const float* b4 = f ? P.film2_b4 : P.film1_b4;
        if (tid < DFH) {
            float s = b0[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w0[tid * DFH + k], summary[k], s);
            fpre[f][0][tid] = s; fh[f][0][tid] = silu_f(s);
        }
        __syncthreads();
        if (tid < DFH) {
            float s = b2[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w2[tid * DFH + k], fh[f][0][k], s);
            fpre[f][1][tid] = s; fh[f][1][tid] = silu_f(s);
        }
        __syncthreads();
        if (tid < 2 * DFH) {
            float s = b4[tid];
            for (int k = 0; k < DFH; ++k) s = fmaf(w4[tid * DFH + k], fh[f][1][k], s);
            mod[f][tid] = s;
        }
        __syncthreads();
    }
    // ================= B. FilmedNetwork per node, backward
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        for (int idx = tid; idx < cnt * 16; idx += 256) {
            const int n = idx >> 4, k = idx & 15;
            long ci = (long)(charges[base + c0 + n] + 1.0f);
            ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
            zs[n][XI + k] = P.emb[ci * 16 + k];
            if (k == 0) cls[n] = (int)ci;
        }
        for (int idx = tid; idx < cnt * 4; idx += 256) {
            const int n = idx >> 2, d = idx & 3;
            gfs[n][d] = d < D ? gfield[(base + c0 + n) * D + d] : 0.0f;
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) {            // s1, y1
            const int n = idx >> 5, o = idx & 31;
            float s = P.lin1_b[o];
#pragma unroll
            for (int k = 0; k < FI; ++k) s = fmaf(P.lin1_w[o * FI + k], zs[n][k], s);
            pa[n][o] = s;
            ya[n][o] = silu_f((1.0f + mod[0][o]) * s + mod[0][DFH + o]);
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) {            // s2, y2, d(pre2)
            const int n = idx >> 5, o = idx & 31;
            float s = P.lin2_b[o];
#pragma unroll
            for (int k = 0; k < DFH; ++k) s = fmaf(P.lin2_w[o * DFH + k], ya[n][k], s);
            pb[n][o] = s;
            const float pre = (1.0f + mod[1][o]) * s + mod[1][DFH + o];
            yb[n][o] = silu_f(pre);
            float dy = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) dy = fmaf(P.lin3_w[d * DFH + o], gfs[n][d], dy);
            db[n][o] = dy * dsilu_f(pre);
        }
        __syncthreads();
        outer_acc(OF::lin3_w, D, DFH, &gfs[0][0], 4, &yb[0][0], LH, cnt);
        col_acc(OF::lin3_b, D, &gfs[0][0], 4, cnt);
        if (tid < DFH) {                                              // d gamma_2, d beta_2
            float sg = 0.0f, sb = 0.0f;
            for (int n = 0; n < cnt; ++n) { sg = fmaf(db[n][tid], pb[n][tid], sg); sb += db[n][tid]; }
            dmod[1][tid] += sg; dmod[1][DFH + tid] += sb;
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) { const int n = idx >> 5, o = idx & 31; db[n][o] *= 1.0f + mod[1][o]; }   // d s2
        __syncthreads();
        outer_acc(OF::lin2_w, DFH, DFH, &db[0][0], LH, &ya[0][0], LH, cnt);
        col_acc(OF::lin2_b, DFH, &db[0][0], LH, cnt);
        for (int idx = tid; idx < cnt * DFH; idx += 256) {            // d(pre1)
            const int n = idx >> 5, k = idx & 31;
            float dy = 0.0f;
#pragma unroll
            for (int o = 0; o < DFH; ++o) dy = fmaf(P.lin2_w[o * DFH + k], db[n][o], dy);
            da[n][k] = dy * dsilu_f((1.0f + mod[0][k]) * pa[n][k] + mod[0][DFH + k]);
        }
        __syncthreads();
        if (tid < DFH) {
            float sg = 0.0f, sb = 0.0f;
            for (int n = 0; n < cnt; ++n) { sg = fmaf(da[n][tid], pa[n][tid], sg); sb += da[n][tid]; }
            dmod[0][tid] += sg; dmod[0][DFH + tid] += sb;
        }
        __syncthreads();
        for (int idx = tid; idx < cnt * DFH; idx += 256) { const int n = idx >> 5, o = idx & 31; da[n][o] *= 1.0f + mod[0][o]; }   // d s1
        __syncthreads();
        outer_acc(OF::lin1_w, DFH, FI, &da[0][0], LH, &zs[0][0], FI, cnt);
        col_acc(OF::lin1_b, DFH, &da[0][0], LH, cnt);
        if (gz != nullptr) {                                          // dL/d[p | v] through the FiLM net's first Linear
            for (int idx = tid; idx < cnt * XI; idx += 256) {
                const int n = idx / XI, k = idx - n * XI;
                float sgz = 0.0f;
                for (int o = 0; o < DFH; ++o) sgz = fmaf(P.lin1_w[o * FI + k], da[n][o], sgz);
                gz[(base + c0 + n) * XI + k] = sgz;
            }
        }
        if (tid < 48) {                                               // class embedding rows
            const int c = tid >> 4, j = tid & 15;
            float s = 0.0f;
            for (int n = 0; n < cnt; ++n) {
                if (cls[n] != c) continue;
                for (int o = 0; o < DFH; ++o) s = fmaf(P.lin1_w[o * FI + XI + j], da[n][o], s);
            }
            g[OF::emb + tid] += s;
        }
        __syncthreads();
    }
    // ================= C. FiLM modulators, backward (one row per graph)
    for (int f = 0; f < 2; ++f) {
        const float* w0 = f ? P.film2_w0 : P.film1_w0;
        const float* w2 = f ? P.film2_w2 : P.film1_w2;
        const float* w4 = f ? P.film2_w4 : P.film1_w4;
        const int fo = f ? OF::film2 : OF::film1;
        for (int e = tid; e < 2 * DFH * DFH; e += 256) g[fo + OF::F_w4 + e] = dmod[f][e >> 5] * fh[f][1][e & 31];
        if (tid < 2 * DFH) g[fo + OF::F_b4 + tid] = dmod[f][tid];
        if (tid < DFH) {
            float s = 0.0f;
            for (int o = 0; o < 2 * DFH; ++o) s = fmaf(w4[o * DFH + tid], dmod[f][o], s);
            dp1[tid] = s * dsilu_f(fpre[f][1][tid]);
        }
        __syncthreads();
        for (int e = tid; e < DFH * DFH; e += 256) g[fo + OF::F_w2 + e] = dp1[e >> 5] * fh[f][0][e & 31];
        if (tid < DFH) {
            g[fo + OF::F_b2 + tid] = dp1[tid];
            float s = 0.0f;
            for (int o = 0; o < DFH; ++o) s = fmaf(w2[o * DFH + tid], dp1[o], s);
            dp0[tid] = s * dsilu_f(fpre[f][0][tid]);
        }
        __syncthreads();
        for (int e = tid; e < DFH * DFH; e += 256) g[fo + OF::F_w0 + e] = dp0[e >> 5] * summary[e & 31];
        if (tid < DFH) {
            g[fo + OF::F_b0 + tid] = dp0[tid];
            float s = 0.0f;
            for (int o = 0; o < DFH; ++o) s = fmaf(w0[o * DFH + tid], dp0[o], s);
            dsum[tid] += s;
        }
        __syncthreads();
    }
    // ================= D. attention pooling, backward
    if (tid < DFH) {                                                  // q[o] = sum_c nn_w2[c][o] dsum[c]
        float s = 0.0f;
        for (int c = 0; c < DFH; ++c) s = fmaf(P.nn_w2[c * DFH + tid], dsum[c], s);
        qv[tid] = s;
    }
    float ldot = 0.0f;
    for (int c0 = 0; c0 < N; c0 += CH) {                              // dL/dw_n = dsum . v_n
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        __syncthreads();
        hidden(P.nn_w0, P.nn_b0, cnt, pa, ya);
        __syncthreads();
        if (tid < cnt) {
            float dw = 0.0f;
            for (int c = 0; c < DFH; ++c) dw = fmaf(dsum[c], P.nn_b2[c], dw);
            for (int o = 0; o < DFH; ++o) dw = fmaf(qv[o], ya[tid][o], dw);
            dwn[c0 + tid] = dw;
            ldot = fmaf(gate[c0 + tid], dw, ldot);
        }
        __syncthreads();
    }
    red[tid] = ldot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float dbar = red[0];
    __syncthreads();
    float lw = 0.0f;
    for (int n = tid; n < N; n += 256) { dwn[n] = gate[n] * (dwn[n] - dbar); lw += gate[n]; }     // dL/dg_n
    red[tid] = lw;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const float wsum = red[0];
    __syncthreads();
    for (int c0 = 0; c0 < N; c0 += CH) {
        const int cnt = N - c0 < CH ? N - c0 : CH;
        stage_x(c0, cnt);
        __syncthreads();
        hidden(P.nn_w0, P.nn_b0, cnt, pa, ya);                        // u_n
        hidden(P.gate_w0, P.gate_b0, cnt, pb, yb);                    // a_n
        __syncthreads();
        if (tid < DFH) {
            float su = 0.0f, sg = 0.0f;
            for (int n = 0; n < cnt; ++n) { su = fmaf(gate[c0 + n], ya[n][tid], su); sg = fmaf(dwn[c0 + n], yb[n][tid], sg); }
            ubar[tid] += su;
            g[OF::gate_w2 + tid] += sg;
        }
        if (tid == 32) {
            float s = 0.0f;
            for (int n = 0; n < cnt; ++n) s += dwn[c0 + n];
            g[OF::gate_b2] += s;
        }
        for (int idx = tid; idx < cnt * DFH; idx += 256) {
            const int n = idx >> 5, o = idx & 31;
            da[n][o] = gate[c0 + n] * qv[o] * dsilu_f(pa[n][o]);
            db[n][o] = dwn[c0 + n] * P.gate_w2[o] * dsilu_f(pb[n][o]);
        }
        __syncthreads();
        outer_acc(OF::nn_w0, DFH, XI, &da[0][0], LH, &zs[0][0], FI, cnt);
        col_acc(OF::nn_b0, DFH, &da[0][0], LH, cnt);
        outer_acc(OF::gate_w0, DFH, XI, &db[0][0], LH, &zs[0][0], FI, cnt);
        col_acc(OF::gate_b0, DFH, &db[0][0], LH, cnt);
        if (gz != nullptr) {            // ... and through the graph summary: the node's row enters nn and gate_nn
            for (int idx = tid; idx < cnt * XI; idx += 256) {          // (same thread -> element map as section B: no race)
                const int n = idx / XI, k = idx - n * XI;
                float sgz = 0.0f;
                for (int o = 0; o < DFH; ++o) sgz = fmaf(P.nn_w0[o * XI + k], da[n][o], fmaf(P.gate_w0[o * XI + k], db[n][o], sgz));
                gz[(base + c0 + n) * XI + k] += sgz;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < DFH * DFH; e += 256) g[OF::nn_w2 + e] = dsum[e >> 5] * ubar[e & 31];
    if (tid < DFH) g[OF::nn_b2 + tid] = dsum[tid] * wsum;
}

// grads.<tensor>[i] = sum over graphs of partial[graph][offset + i].  A thread summing all graphs one after the other
// is a chain of n_graphs dependent loads (34 us at 128 graphs): eight threads share an element instead, thread group j
// adds the graphs j, j + 8, ... in order (independent loads, unrolled), and the eight partial sums are combined through
// LDS in a fixed order -- the same tree for every launch, so the result is reproducible.
template <int D>
__global__ void __launch_bounds__(256)
k_dynfield_reduce(const float* __restrict__ partial, int64_t n_graphs, AetherDynFieldParams G) {
    using OF = DynOff<D>;
    __shared__ float red[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), j = threadIdx.x >> 5;      // 32 consecutive elements per row read
    float s = 0.0f;
    if (e < OF::total) {
        int64_t k = j;
        for (; k + 24 < n_graphs; k += 32) {
            const float a0 = partial[(size_t)k * OF::total + e], a1 = partial[(size_t)(k + 8) * OF::total + e];
            const float a2 = partial[(size_t)(k + 16) * OF::total + e], a3 = partial[(size_t)(k + 24) * OF::total + e];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; k < n_graphs; k += 8) s += partial[(size_t)k * OF::total + e];
    }
    red[j][threadIdx.x & 31] = s;
    __syncthreads();
    if (e >= OF::total || j != 0) return;
    s = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) +
        ((red[4][threadIdx.x] + red[5][threadIdx.x]) + (red[6][threadIdx.x] + red[7][threadIdx.x]));   // a fixed tree
    const float* ptrs[27] = {G.gate_w0, G.gate_b0, G.gate_w2, G.gate_b2, G.nn_w0, G.nn_b0, G.nn_w2, G.nn_b2, G.lin1_w, G.lin1_b,
                             G.lin2_w, G.lin2_b, G.lin3_w, G.lin3_b, G.film1_w0, G.film1_b0, G.film1_w2, G.film1_b2, G.film1_w4,
                             G.film1_b4, G.film2_w0, G.film2_b0, G.film2_w2, G.film2_b2, G.film2_w4, G.film2_b4, G.emb};
    const int offs[28] = {OF::gate_w0, OF::gate_b0, OF::gate_w2, OF::gate_b2, OF::nn_w0, OF::nn_b0, OF::nn_w2, OF::nn_b2,
                          OF::lin1_w, OF::lin1_b, OF::lin2_w, OF::lin2_b, OF::lin3_w, OF::lin3_b,
                          OF::film1 + OF::F_w0, OF::film1 + OF::F_b0, OF::film1 + OF::F_w2, OF::film1 + OF::F_b2,
                          OF::film1 + OF::F_w4, OF::film1 + OF::F_b4, OF::film2 + OF::F_w0, OF::film2 + OF::F_b0,
                          OF::film2 + OF::F_w2, OF::film2 + OF::F_b2, OF::film2 + OF::F_w4, OF::film2 + OF::F_b4, OF::emb,
                          OF::total};
    int t = 0;
    while (e >= offs[t + 1]) ++t;
    const_cast<float*>(ptrs[t])[e - offs[t]] = s;
}

}  // namespace
