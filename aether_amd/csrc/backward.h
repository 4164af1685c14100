// backward.h -- gradients of the Aether step w.r.t. its parameters (positions, velocities, charges
// and edge attributes are data: the reference detaches them, experiments/lorentz/main.py:243-247).
//
// Structure (first version; correctness before speed):
//   * data-gradient kernels mirror the forward kernels tile for tile -- the same accumulator-layout
//     chains, with the weights staged TRANSPOSED in LDS so that W^T g is again
//     `gemm_tile(A = W^T, B = g)`; pre-activations are recomputed from the saved x / n / e tensors;
//   * every weight / bias gradient is a sum over rows (nodes or edges) of an outer product
//     A[row] (x) B[row]: ONE generic MFMA kernel (k_outer) computes all of them from the [rows, .]
//     tensors the data-gradient kernels leave in the workspace, in fixed order (per-chunk partials +
//     ordered reduce): deterministic, no atomics;
//   * sums over a node's out-edges (gradients flowing back along the sender index) use a second,
//     sender-sorted edge list built with the graph.
// Reference: the oracle's autograd (oracle/aether_oracle.py) is the gradient oracle; see
// tests/test_gpu_backward.py.
#pragma once
#include "common.h"

namespace {

// d silu(z) / dz = s * (1 + z * (1 - s)), s = sigmoid(z)
__device__ __forceinline__ float dsilu(float z) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f));
    return s * (1.0f + z * (1.0f - s));
}
__device__ __forceinline__ f32x4 dsilu4(f32x4 v) {
    f32x4 o;
    o[0] = dsilu(v[0]); o[1] = dsilu(v[1]); o[2] = dsilu(v[2]); o[3] = dsilu(v[3]);
    return o;
}

// LDS copy of W^T: dst[c][r] = W[r][col0 + c] for r < rows, c < cols; dst rows padded to ldt, zero
// filled up to rows_pad x cols_pad.
__device__ __forceinline__ void stage_weight_T(float* lds, const float* __restrict__ w, int rows,
                                               int cols, int src_ld, int col0, int ldt, int cols_pad) {
    for (int idx = threadIdx.x; idx < cols_pad * ldt; idx += blockDim.x) {
        const int c = idx / ldt, r = idx - c * ldt;
        lds[idx] = (c < cols && r < rows) ? w[(size_t)r * src_ld + col0 + c] : 0.0f;
    }
}

// load / store a 16-item x 64 tile in accumulator layout from a row-major [rows][ld] tensor
__device__ __forceinline__ void load_tile64(f32x4 (&t)[4], const float* __restrict__ p, int64_t row,
                                            int ld, int q) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) t[mb] = ld4(p + row * ld + 16 * mb + 4 * q);
}
__device__ __forceinline__ void store_tile64(float* __restrict__ p, int64_t row, int ld, int q,
                                             const f32x4 (&t)[4]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) st4(p + row * ld + 16 * mb + 4 * q, t[mb]);
}

// ------------------------------------------------------------------ generic outer-product kernel
// C[m][n] (ldc) = sum_row A[row][m] * B[row][n], m < M, n < N; optional bias[m] = sum_row A[row][m].
struct OuterTask {
    const float* A; const float* B; float* C; float* bias;
    int lda, ldb, ldc, M, N;        // M, N = valid output extents (blocks of 16 cover them)
    int chunks;                     // row chunks of this task (<= batch.chunks)
    int64_t rows;
};
constexpr int OUTER_MAX_TASKS = 8;
struct OuterBatch { OuterTask t[OUTER_MAX_TASKS]; int n_tasks; int chunks; };

// grid = (chunks, n_tasks).  A workgroup owns a chunk of row tiles and computes EVERY 16x16 block of
// its task's output for it: each row tile of A and B is read from global memory once (coalesced
// 16-byte loads) into LDS, then read back transposed as MFMA operands.  Wave w owns the block rows
// mb = w, w + 4.  Column sums of A (bias gradients) ride along in the staging threads.
// partial[(task, chunk)][block][256 + 16]
constexpr int LDO = 144;      // LDS row stride of a staged tile: 144 % 32 == 16 -> conflict-free b32 reads
// BIG = false serves tasks of at most 64 x 64 (16 accumulator registers instead of 64: more resident
// workgroups hide the row-load latency of the [E, 64] tasks); tasks of the other class exit at once.
template <bool BIG>
__global__ void __launch_bounds__(256)
k_outer(OuterBatch batch, float* __restrict__ partial) {
    const OuterTask T = batch.t[blockIdx.y];
    if ((((T.M + 15) >> 4) > 4 || ((T.N + 15) >> 4) > 4) != BIG) return;
    const int MBn = (T.M + 15) >> 4, NBn = (T.N + 15) >> 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    if ((int)blockIdx.x >= T.chunks) return;
    const int64_t tiles = (T.rows + 15) >> 4;
    const int64_t per = (tiles + T.chunks - 1) / T.chunks;
    const int64_t t0 = per * blockIdx.x, t1 = t0 + per < tiles ? t0 + per : tiles;
    __shared__ __attribute__((aligned(16))) float sa[16 * LDO], sb[16 * LDO];
    __shared__ float scol[16][132];
    constexpr int AM = BIG ? 2 : 1, NBM = BIG ? 8 : 4;
    f32x4 acc[AM][NBM];           // block rows {wave, wave + 4} x up to 8 block columns
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < NBM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging map: thread -> (row, 4 columns); the same thread always serves the same columns, so
    // its running sums are this chunk's column sums of A restricted to its row index
    const int a4 = MBn * 4, b4 = NBn * 4;                 // float4 per row
    f32x4 csum[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // staging coordinates of this thread (tile independent): hoisted out of the loop -- the runtime
    // integer divisions were most of the per-tile instruction count
    int ra[2], ca[2], rb[2], cb[2];
    bool oka[2], okb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = threadIdx.x + 256 * u;
        oka[u] = idx < 16 * a4; okb[u] = idx < 16 * b4;
        ra[u] = idx / a4; ca[u] = (idx - ra[u] * a4) * 4;
        rb[u] = idx / b4; cb[u] = (idx - rb[u] * b4) * 4;
    }
    // register prefetch, two tiles deep: rows of tiles t+1 and t+2 are in flight while tile t is multiplied
    f32x4 va[2][2], vb[2][2];              // [stage][u]
    auto fetch = [&](int64_t t, f32x4 (&xa)[2], f32x4 (&xb)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xa[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            xb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < t1) {
                if (oka[u] && 16 * t + ra[u] < T.rows) xa[u] = ld4(T.A + (16 * t + ra[u]) * T.lda + ca[u]);
                if (okb[u] && 16 * t + rb[u] < T.rows) xb[u] = ld4(T.B + (16 * t + rb[u]) * T.ldb + cb[u]);
            }
        }
    };
    auto consume = [&](const f32x4 (&xa)[2], const f32x4 (&xb)[2]) {
        __syncthreads();                                   // previous tile consumed
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (oka[u]) { st4(sa + ra[u] * LDO + ca[u], xa[u]); csum[u] += xa[u]; }
            if (okb[u]) st4(sb + rb[u] * LDO + cb[u], xb[u]);
        }
    };
    auto multiply = [&]() {
        __syncthreads();
#pragma unroll
        for (int a = 0; a < AM; ++a) {
            const int mb = wave + 4 * a;
            if (mb < MBn) {
                float av[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) av[s4] = sa[(4 * s4 + q) * LDO + 16 * mb + i];
#pragma unroll
                for (int nb = 0; nb < NBM; ++nb) {
                    if (nb < NBn) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4)
                            acc[a][nb] = mfma16(av[s4], sb[(4 * s4 + q) * LDO + 16 * nb + i], acc[a][nb]);
                    }
                }
            }
        }
    };
    fetch(t0, va[0], vb[0]);
    fetch(t0 + 1, va[1], vb[1]);
    for (int64_t t = t0; t < t1; t += 2) {
        consume(va[0], vb[0]);
        fetch(t + 2, va[0], vb[0]);
        multiply();
        if (t + 1 < t1) {
            consume(va[1], vb[1]);
            fetch(t + 3, va[1], vb[1]);
            multiply();
        }
    }
    float* dst0 = partial + ((size_t)blockIdx.y * batch.chunks + blockIdx.x) * 32 * 272;
#pragma unroll
    for (int a = 0; a < AM; ++a) {
        const int mb = wave + 4 * a;
#pragma unroll
        for (int nb = 0; nb < NBM; ++nb) {
            if (mb < MBn && nb < NBn) {
                float* dst = dst0 + (size_t)(mb * NBn + nb) * 272;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(4 * q + r) * 16 + i] = acc[a][nb][r];
            }
        }
    }
    if (T.bias != nullptr) {                               // column sums: reduce the 16 row indices in order
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = threadIdx.x + 256 * u;
            if (idx < 16 * a4) {
                const int r = idx / a4, c = (idx - r * a4) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) scol[r][c + j] = csum[u][j];
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 16 * MBn) {
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += scol[r][threadIdx.x];
            // bias of block row mb lives in the partial of block (mb, nb = 0), slots 256..271
            dst0[(size_t)((threadIdx.x >> 4) * NBn) * 272 + 256 + (threadIdx.x & 15)] = sm;
        }
    }
}

// grid = (blocks, n_tasks), 1024 threads: element (r, c) of one 16x16 block x 4 chunk groups.  Each
// group adds its contiguous quarter of the chunks in order; the four group sums are combined in
// order through LDS: a fixed summation tree, hence deterministic.
__global__ void __launch_bounds__(1024)
k_outer_reduce(OuterBatch batch, const float* __restrict__ partial) {
    const OuterTask T = batch.t[blockIdx.y];
    const int MBn = (T.M + 15) >> 4, NBn = (T.N + 15) >> 4;
    if ((int)blockIdx.x >= MBn * NBn) return;
    const int mb = blockIdx.x / NBn, nb = blockIdx.x - mb * NBn;
    const int grp = threadIdx.x >> 8, e = threadIdx.x & 255;
    const int r = e >> 4, c = e & 15;
    const int per = (T.chunks + 3) / 4;
    const int c0 = grp * per, c1 = c0 + per < T.chunks ? c0 + per : T.chunks;
    float s = 0.0f, sb = 0.0f;
    const float* src0 = partial + ((size_t)blockIdx.y * batch.chunks * 32 + blockIdx.x) * 272;
    int ch = c0;
    for (; ch + 8 <= c1; ch += 8) {                     // 8 chunk loads in flight, summed in chunk order
        float v[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* src = src0 + (size_t)(ch + u) * 32 * 272;
            v[u] = src[r * 16 + c];
            vb[u] = c == 0 ? src[256 + r] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s += v[u]; sb += vb[u]; }
    }
    for (; ch < c1; ++ch) {
        const float* src = src0 + (size_t)ch * 32 * 272;
        s += src[r * 16 + c];
        if (c == 0) sb += src[256 + r];
    }
    __shared__ float red[4][256], redb[4][16];
    red[grp][e] = s;
    if (c == 0) redb[grp][r] = sb;
    __syncthreads();
    if (grp == 0) {
        const float tot = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        const int m = 16 * mb + r, n = 16 * nb + c;
        if (m < T.M && n < T.N) T.C[(size_t)m * T.ldc + n] = tot;
        if (T.bias != nullptr && nb == 0 && c == 0 && m < T.M)
            T.bias[m] = ((redb[0][r] + redb[1][r]) + redb[2][r]) + redb[3][r];
    }
}

// ------------------------------------------------------------------ transposed weight copies
// One launch at the start of the backward writes W^T for every Linear whose input gradient is
// needed, so that W^T g is `gemm_tile(A = W^T, ...)` with the same 16-byte fragment reads as the
// forward.  dst[c][r] = src[r][col0 + c] for r < rows, c < cols; dst is [cols_pad][ldd], zero padded.
struct TransposeTask { const float* src; float* dst; int rows, cols, src_ld, col0, ldd, cols_pad; };
constexpr int TRANSPOSE_MAX_TASKS = 20;
struct TransposeBatch { TransposeTask t[TRANSPOSE_MAX_TASKS]; int n_tasks; };

__global__ void __launch_bounds__(256)
k_transpose(TransposeBatch batch) {
    const TransposeTask T = batch.t[blockIdx.y];
    const int total = T.cols_pad * T.ldd;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int c = idx / T.ldd, r = idx - c * T.ldd;
        T.dst[idx] = (c < T.cols && r < T.rows) ? T.src[(size_t)r * T.src_ld + T.col0 + c] : 0.0f;
    }
}

// pointers into the transposed-weight region of the workspace
struct BwdWT {
    const float* out_w0t; const float* out_w3t; const float* out_w6t;      // [64][64], [64][64], [64][16]
    const float* upd_w2t[4]; const float* upd_w0t[4];                       // W4^T [128][64], W3^T [64][128]
    const float* msg_w2t[4];                                                // W2^T [64][64]
    const float* msg_w0t[4];         // layer 1: W1^T [32][64]; layers 2-4: [192][64] = W_s^T | W_r^T | W_e^T
};

// ------------------------------------------------------------------ out MLP backward
// forward (locs.py:160-168, local_to_global.py:12-13, aether.py:185):
//   o1 = silu(Wo0 x4 + b), o2 = silu(Wo3 o1 + b), y = Wo6 o2 + b, out = p + R y
// in: g = dL/dout.  out: dx4 and the row tensors of the weight gradients.
// One wave per 16-node tile; weights are read from L2 in fragment shape.
template <int D>
__global__ void __launch_bounds__(64)
kb_out(AetherParams P, BwdWT WT, const float* __restrict__ x4, const float* __restrict__ nodeinfo,
       const float* __restrict__ g_out, float* __restrict__ DX, float* __restrict__ O1,
       float* __restrict__ O2, float* __restrict__ DPO1, float* __restrict__ DPO2,
       float* __restrict__ DY, int64_t n_nodes) {
    using NI = NodeInfo<D>;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    f32x4 xt[4], p1[4], p2[4], o1[4], o2[4];
    load_tile64(xt, x4, nc, H, q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(P.out_b0 + 16 * mb + 4 * q);
    gemm_tile<4, 4>(P.out_w0, H, xt, p1, i, q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) { o1[mb] = silu4(p1[mb]); p2[mb] = ld4(P.out_b3 + 16 * mb + 4 * q); }
    gemm_tile<4, 4>(P.out_w3, H, o1, p2, i, q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) o2[mb] = silu4(p2[mb]);
    // dy = R^T g (rows 0..D-1 of a 16-row block: lanes q == 0, registers 0..D-1)
    f32x4 dy = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q == 0 && ok) {
        const float* ni = nodeinfo + node * NI::STRIDE;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < D; ++b) s += ni[NI::R + b * D + a] * g_out[node * D + b];
            dy[a] = s;
        }
    }
    // do2 = Wo6^T dy  (K = 16, only k < D non-zero)
    f32x4 d2[4], d1[4], dx[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        d2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 wv = ld4(WT.out_w6t + (16 * mb + i) * 16 + 4 * q);
#pragma unroll
        for (int b = 0; b < 4; ++b) d2[mb] = mfma16(wv[b], dy[b], d2[mb]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) { d2[mb] = d2[mb] * dsilu4(p2[mb]); d1[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    gemm_tile<4, 4>(WT.out_w3t, H, d2, d1, i, q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) { d1[mb] = d1[mb] * dsilu4(p1[mb]); dx[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    gemm_tile<4, 4>(WT.out_w0t, H, d1, dx, i, q);
    if (ok) {
        store_tile64(DX, node, H, q, dx);
        store_tile64(O1, node, H, q, o1);
        store_tile64(O2, node, H, q, o2);
        store_tile64(DPO1, node, H, q, d1);
        store_tile64(DPO2, node, H, q, d2);
        st4(DY + node * 16 + 4 * q, dy);
    }
}

// ------------------------------------------------------------------ node update backward
// forward (locs.py:240-241): u = silu(W3 n + b3), x = n + W4 u + b4.   in: dx.  out: dn, u, dpre_u.
__global__ void __launch_bounds__(64)
kb_node(const float* __restrict__ w3g, const float* __restrict__ b3g, const float* __restrict__ w4t,
        const float* __restrict__ w3t, const float* __restrict__ nbuf, const float* __restrict__ DX,
        float* __restrict__ DN, float* __restrict__ U, float* __restrict__ DPU, int64_t n_nodes) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    f32x4 nt[4], dx[4], pu[8], du[8], dn[4];
    load_tile64(nt, nbuf, nc, H, q);
    load_tile64(dx, DX, nc, H, q);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) { pu[mb] = ld4(b3g + 16 * mb + 4 * q); du[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    gemm_tile<8, 4>(w3g, H, nt, pu, i, q);          // pre_u = W3 n + b3
    gemm_tile<8, 4>(w4t, H, dx, du, i, q);          // du = W4^T dx
    f32x4 u[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) { u[mb] = silu4(pu[mb]); du[mb] = du[mb] * dsilu4(pu[mb]); }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) dn[mb] = dx[mb];
    gemm_tile<4, 8>(w3t, 2 * H, du, dn, i, q);      // dn = dx + W3^T dpre_u
    if (ok) {
        store_tile64(DN, node, H, q, dn);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            st4(U + node * 2 * H + 16 * mb + 4 * q, u[mb]);
            st4(DPU + node * 2 * H + 16 * mb + 4 * q, du[mb]);
        }
    }
}

// ------------------------------------------------------------------ edge MLP backward
// forward (locs.py:227-238): pre1 = [layer 1: W1 a + b1 | else: P_s[s] + P_r[r] + W_e e_prev],
//   h = silu(pre1), pre2 = W2 h + b2, e = silu(pre2), aggr_i = mean_{k: recv = i} e_k.
// de_k = DN[recv_k] / deg[recv_k] (+ DE[k], the gradient through the next layer's W_e)
// out: G = dL/dpre1, H1 = h, DP2 = dL/dpre2, and  FIRST: DA = W1^T G   else: DE[k] <- W_e^T G.
// Weights (forward and pre-transposed) are staged once per workgroup in LDS.
template <bool FIRST>
__global__ void __launch_bounds__(256, 2)
kb_edge(const float* __restrict__ w_in /*FIRST: W1 [64][F1] else msg_w0 [64][192]*/, int f1,
        const float* __restrict__ b_in, const float* __restrict__ w2g, const float* __restrict__ b2g,
        const float* __restrict__ w_in_t /*FIRST: W1^T [32][64] else W_e^T [64][64]*/,
        const float* __restrict__ w2t, const float* __restrict__ Ps, const float* __restrict__ Pr,
        const float* __restrict__ e_prev, const float* __restrict__ feat,
        const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
        const int32_t* __restrict__ rowptr, const float* __restrict__ DN, float* __restrict__ DE,
        int have_de, float* __restrict__ G, float* __restrict__ H1, float* __restrict__ DP2,
        float* __restrict__ DA, int64_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wi = smem;                  // W_in   [64][LDW]   (FIRST: K = 32 used)
    float* w2 = wi + H * LDW;          // W2
    float* w2ts = w2 + H * LDW;        // W2^T
    float* wit = w2ts + H * LDW;       // W_in^T [64 | 32][LDW]
    if (FIRST) {
        stage_weight(wi, w_in, H, f1, f1, LDW);
        for (int idx = threadIdx.x; idx < FPAD * (H / 4); idx += 256) {
            const int r = idx >> 4, c = (idx & 15) * 4;
            st4(wit + r * LDW + c, ld4(w_in_t + (size_t)r * H + c));
        }
    } else {
        stage_weight64<256>(wi, w_in + 2 * H, 3 * H);
        stage_weight64<256>(wit, w_in_t, H);
    }
    stage_weight64<256>(w2, w2g, H);
    stage_weight64<256>(w2ts, w2t, H);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t tiles = (n_edges + 15) >> 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < tiles; t += (int64_t)gridDim.x * 4) {
        int z = 0;
        asm volatile("" : "+v"(z));      // opaque offset: weight fragments are re-read from LDS per tile
                                         // instead of being hoisted into ~256 registers (1 wave per SIMD)
        const int64_t k = 16 * t + i;
        const bool ok = k < n_edges;
        const int64_t kc = ok ? k : n_edges - 1;
        const int64_t s = send_s[kc], r = recv_s[kc];
        f32x4 p1[4], p2[4], h[4];
        if (FIRST) {
            f32x4 bop[2];
            bop[0] = ld4(feat + kc * FPAD + 4 * q);
            bop[1] = ld4(feat + kc * FPAD + 16 + 4 * q);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(b_in + 16 * mb + 4 * q);
            gemm_tile<4, 2>(wi + z, LDW, bop, p1, i, q);
        } else {
            f32x4 bop[4];
            load_tile64(bop, e_prev, kc, H, q);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                p1[mb] = ld4(Ps + s * H + 16 * mb + 4 * q) + ld4(Pr + r * H + 16 * mb + 4 * q);
            gemm_tile<4, 4>(wi + z, LDW, bop, p1, i, q);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { h[mb] = silu4(p1[mb]); p2[mb] = ld4(b2g + 16 * mb + 4 * q); }
        gemm_tile<4, 4>(w2 + z, LDW, h, p2, i, q);
        // de = dn[recv] / deg (+ gradient through the next layer's edge input)
        const int deg = rowptr[r + 1] - rowptr[r];
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        f32x4 d2[4], dh[4], g[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 de = ld4(DN + r * H + 16 * mb + 4 * q) * inv;
            if (have_de) de += ld4(DE + kc * H + 16 * mb + 4 * q);
            d2[mb] = de * dsilu4(p2[mb]);
            dh[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        gemm_tile<4, 4>(w2ts + z, LDW, d2, dh, i, q);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) g[mb] = dh[mb] * dsilu4(p1[mb]);
        if (FIRST) {
            f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            gemm_tile<2, 4>(wit + z, LDW, g, da, i, q);
            if (ok) { st4(DA + k * FPAD + 4 * q, da[0]); st4(DA + k * FPAD + 16 + 4 * q, da[1]); }
        } else {
            f32x4 dep[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) dep[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_tile<4, 4>(wit + z, LDW, g, dep, i, q);
            if (ok) store_tile64(DE, k, H, q, dep);
        }
        if (ok) {
            store_tile64(G, k, H, q, g);
            store_tile64(H1, k, H, q, h);
            store_tile64(DP2, k, H, q, d2);
        }
    }
}

// ------------------------------------------------------------------ sums of G onto nodes
// dP_r[i] = sum_{k: recv = i} G_k (contiguous run), dP_s[j] = sum_{k: send = j} G_k (sender list,
// rows gathered through sperm).  One wave per node, lane = column, fixed order, 8 loads in flight.
__global__ void __launch_bounds__(256)
kb_sum_g(const float* __restrict__ G, const int32_t* __restrict__ rowptr,
         const int32_t* __restrict__ srowptr, const int32_t* __restrict__ sperm,
         float* __restrict__ DPS, float* __restrict__ DPR, int64_t n_nodes) {
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    {
        const int beg = rowptr[node], end = rowptr[node + 1];
        const float* p = G + (int64_t)beg * H + lane;
        float s = 0.f;
        int k = beg;
        for (; k + 8 <= end; k += 8, p += 8 * H) {
            const float a0 = p[0], a1 = p[H], a2 = p[2 * H], a3 = p[3 * H], a4 = p[4 * H], a5 = p[5 * H],
                        a6 = p[6 * H], a7 = p[7 * H];
            s += a0; s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;
        }
        for (; k < end; ++k, p += H) s += p[0];
        DPR[node * H + lane] = s;
    }
    {
        const int beg = srowptr[node], end = srowptr[node + 1];
        float s = 0.f;
        int k = beg;
        for (; k + 4 <= end; k += 4) {
            const int r0 = sperm[k], r1 = sperm[k + 1], r2 = sperm[k + 2], r3 = sperm[k + 3];
            const float a0 = G[(int64_t)r0 * H + lane], a1 = G[(int64_t)r1 * H + lane],
                        a2 = G[(int64_t)r2 * H + lane], a3 = G[(int64_t)r3 * H + lane];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; k < end; ++k) s += G[(int64_t)sperm[k] * H + lane];
        DPS[node * H + lane] = s;
    }
}

// dx_{l-1} = dn_l + W_s^T dP_s + W_r^T dP_r   (locs.py:233 split, transposed); wave per node tile
__global__ void __launch_bounds__(64)
kb_gather(const float* __restrict__ w1t /*[192][64]: W_s^T | W_r^T | W_e^T*/,
          const float* __restrict__ DPS, const float* __restrict__ DPR, const float* __restrict__ DN,
          float* __restrict__ DX, int64_t n_nodes) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    f32x4 dps[4], dpr[4], dx[4];
    load_tile64(dps, DPS, nc, H, q);
    load_tile64(dpr, DPR, nc, H, q);
    load_tile64(dx, DN, nc, H, q);
    gemm_tile<4, 4>(w1t, H, dps, dx, i, q);
    gemm_tile<4, 4>(w1t + H * H, H, dpr, dx, i, q);
    if (ok) store_tile64(DX, node, H, q, dx);
}

// ------------------------------------------------------------------ field net backward
// 32 threads per node (8 nodes per workgroup).  Gradient of the loss w.r.t. the field f of node j
// arrives through
//   * the R_i^T f_j columns of its out-edges' features (DA columns RF, rotated back with R_recv),
//   * the rel_feat[recv] columns R_j^T f_j of its in-edges' features (DA columns CF),
//   * layer_1.res(rel_feat) (aether.py:39-48, locs.py:214-218): W_res^T dn_1, columns 2D..3D,
// then through the 3-layer field MLP (aether.py:113-119).  Leaves the row tensors for k_outer.
__device__ __forceinline__ float sum32(float v) {       // all-reduce over the 32 threads of a node
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}

template <int D>
__global__ void __launch_bounds__(256)
kb_field(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel,
         const float* __restrict__ charges, const float* __restrict__ nodeinfo,
         const float* __restrict__ DA, const float* __restrict__ DN1,
         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ recv_s,
         const int32_t* __restrict__ srowptr, const int32_t* __restrict__ sperm,
         float* __restrict__ RELF, float* __restrict__ Z, float* __restrict__ H1f,
         float* __restrict__ H2f, float* __restrict__ DPH1, float* __restrict__ DPH2,
         float* __restrict__ DF, float* __restrict__ DZE, float* __restrict__ ONEHOT, int64_t n_nodes) {
    using NI = NodeInfo<D>;
    constexpr int FIN = 2 * D + 16;
    constexpr int O = D * (D - 1) / 2;
    constexpr int C_RF = 3 * D + O;           // feature columns of R_i^T f_j
    constexpr int C_CF = 6 * D + O;           // feature columns of rel_feat[recv]'s R_i^T f_i part
    __shared__ float sz[8][32], sh1[8][32], sd2[8][32], sd1[8][32];
    const int g = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int64_t n = (int64_t)blockIdx.x * 8 + g;
    const bool ok = n < n_nodes;
    const int64_t nc = ok ? n : n_nodes - 1;
    const float* ni = nodeinfo + nc * NI::STRIDE;
    // ---- dL/dcf, cf = R^T f
    float dcf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dcf[d] = 0.f;
    for (int k = rowptr[nc] + t; k < rowptr[nc + 1]; k += 32) {
#pragma unroll
        for (int d = 0; d < D; ++d) dcf[d] += DA[(int64_t)k * FPAD + C_CF + d];
    }
#pragma unroll
    for (int o = t; o < H; o += 32) {
        const float gg = DN1[nc * H + o];
#pragma unroll
        for (int d = 0; d < D; ++d) dcf[d] += P.l1_res_w[o * 3 * D + 2 * D + d] * gg;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dcf[d] = sum32(dcf[d]);
    // ---- dL/df = R dcf + sum over out-edges of R_recv d(rf)
    float df[D];
#pragma unroll
    for (int a = 0; a < D; ++a) df[a] = 0.f;
    for (int kk = srowptr[nc] + t; kk < srowptr[nc + 1]; kk += 32) {
        const int64_t k = sperm[kk];
        const float* nr = nodeinfo + (int64_t)recv_s[k] * NI::STRIDE;
#pragma unroll
        for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int b = 0; b < D; ++b) df[a] += nr[NI::R + a * D + b] * DA[k * FPAD + C_RF + b];
        }
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s = sum32(df[a]);
#pragma unroll
        for (int b = 0; b < D; ++b) s += ni[NI::R + a * D + b] * dcf[b];
        df[a] = s;
    }
    // ---- recompute the field MLP: thread t owns hidden unit t
    long ci = (long)(charges[nc] + 1.0f);
    ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
    float zt = 0.f;
    if (t < D) zt = x[nc * D + t];
    else if (t < 2 * D) zt = vel[nc * D + t - D];
    else if (t < FIN) zt = P.field_emb[ci * 16 + t - 2 * D];
    sz[g][t] = zt;
    __syncthreads();
    float p1 = P.field_b0[t];
#pragma unroll
    for (int k = 0; k < FIN; ++k) p1 += P.field_w0[t * FIN + k] * sz[g][k];
    const float h1 = silu(p1);
    sh1[g][t] = h1;
    __syncthreads();
    float p2 = P.field_b2[t];
#pragma unroll
    for (int k = 0; k < 32; ++k) p2 += P.field_w2[t * 32 + k] * sh1[g][k];
    const float h2 = silu(p2);
    float d2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) d2 += P.field_w4[d * 32 + t] * df[d];
    d2 *= dsilu(p2);
    sd2[g][t] = d2;
    __syncthreads();
    float d1 = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o) d1 += P.field_w2[o * 32 + t] * sd2[g][o];
    d1 *= dsilu(p1);
    sd1[g][t] = d1;
    __syncthreads();
    if (!ok) return;
    Z[n * 32 + t] = zt;
    H1f[n * 32 + t] = h1;
    H2f[n * 32 + t] = h2;
    DPH1[n * 32 + t] = d1;
    DPH2[n * 32 + t] = d2;
    if (t < 16) {
        float rv = 0.f;                                       // rel_feat row for dW_res
        if (t >= D && t < 2 * D) rv = ni[NI::CV + t - D];
        if (t >= 2 * D && t < 3 * D) rv = ni[NI::CF + t - 2 * D];
        RELF[n * 16 + t] = rv;
        float dfv = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) if (t == d) dfv = df[d];
        DF[n * 16 + t] = dfv;
        ONEHOT[n * 16 + t] = t == (int)ci ? 1.f : 0.f;
        float s = 0.f;                                        // d z[2D + t] (embedding columns)
#pragma unroll 8
        for (int o = 0; o < 32; ++o) s += P.field_w0[o * FIN + 2 * D + t] * sd1[g][o];
        DZE[n * 16 + t] = s;
    }
}

}  // namespace
