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
    int chunks;                     // row chunks (= workgroups = partials) of this task
    int part0;                      // index of its first partial
    int64_t rows;
};
constexpr int OUTER_MAX_TASKS = 48;  // 48 x 72 bytes of kernel arguments
struct OuterBatch { OuterTask t[OUTER_MAX_TASKS]; int n_tasks; };

// Weight gradients C[M, N] = A[rows, M]^T B[rows, N] (+ column sums of A as the bias gradient).
// grid = (chunks, n_tasks), 4 waves per workgroup.  Every wave owns whole 16-row tiles (t = t0 + wave,
// + 4, ..): it parks the 16 rows of A and B in its private LDS rows (coalesced 16-byte loads, the next
// tile prefetched into registers) and reads them back as MFMA operands.  Output block (mb, nb) holds
// the elements m = MB*i' + mb, n = NB*i + nb, so ONE 16-byte LDS read per k-step yields the A (resp. B)
// operands of four blocks: 8 reads feed the 64 MFMAs of a <4,4> tile.  No workgroup barrier in the
// loop; at the end the four waves add their accumulators in wave order through LDS (deterministic)
// and write one partial per workgroup: partial[task.part0 + chunk] = [16MB][16NB] row-major + bias at 8192.
// Tasks of another block class than <MB, NB> exit at once.
constexpr int OUTER_PART = 32 * 272;        // floats per partial (>= 128*64 + 128)
template <int MB, int NB>
__global__ void __launch_bounds__(256)
k_outer(OuterBatch batch, float* __restrict__ partial) {
    const OuterTask T = batch.t[blockIdx.y];
    const int MBn = (T.M + 15) >> 4, NBn = (T.N + 15) >> 4;
    if ((MBn > 4 ? 8 : 4) != MB || (NBn > 4 ? 8 : 4) != NB) return;
    if ((int)blockIdx.x >= T.chunks) return;
    constexpr int SA = 16 * MB + 16, SB = 16 * NB + 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    float* sa = smem + wave * (16 * (SA + SB));
    float* sb = sa + 16 * SA;
    const int tiles = (int)((T.rows + 15) >> 4);
    const int per = (tiles + T.chunks - 1) / T.chunks;
    const int t0 = per * blockIdx.x, t1 = t0 + per < tiles ? t0 + per : tiles;
    f32x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // staging coordinates (tile independent): float4 number f = lane + 64 j -> (row, 4 columns)
    f32x4 va[MB], vb[NB], csum[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int f = lane + 64 * j, r = f / (4 * MB), c = (f % (4 * MB)) * 4;
            const int64_t row = 16 * (int64_t)t + r;
            va[j] = (t < t1 && row < T.rows && c < 16 * MBn) ? ld4(T.A + row * T.lda + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int f = lane + 64 * j, r = f / (4 * NB), c = (f % (4 * NB)) * 4;
            const int64_t row = 16 * (int64_t)t + r;
            vb[j] = (t < t1 && row < T.rows && c < 16 * NBn) ? ld4(T.B + row * T.ldb + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(t0 + wave);
    for (int t = t0 + wave; t < t1; t += 4) {
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int f = lane + 64 * j, r = f / (4 * MB), c = (f % (4 * MB)) * 4;
            st4(sa + r * SA + c, va[j]);
            csum[j] += va[j];
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int f = lane + 64 * j, r = f / (4 * NB), c = (f % (4 * NB)) * 4;
            st4(sb + r * SB + c, vb[j]);
        }
        fetch(t + 4);                                      // next tile of this wave: in flight under the MFMAs
        __builtin_amdgcn_wave_barrier();
        f32x4 av[4][MB / 4], bv[4][NB / 4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
            for (int h = 0; h < MB / 4; ++h) av[s4][h] = ld4(sa + (4 * s4 + q) * SA + MB * i + 4 * h);
#pragma unroll
            for (int h = 0; h < NB / 4; ++h) bv[s4][h] = ld4(sb + (4 * s4 + q) * SB + NB * i + 4 * h);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = mfma16(av[s4][mb >> 2][mb & 3], bv[s4][nb >> 2][nb & 3], acc[mb][nb]);
        __builtin_amdgcn_wave_barrier();
    }
    // ---- the four waves add up in wave order: red[m][n], m = MB*(4q + r) + mb, n = NB*i + nb
    constexpr int LDP = 16 * NB;
    float* red = smem;
    float* redb = smem + 16 * MB * LDP;
    if (T.bias != nullptr) {          // column sums within the wave: lanes with equal columns differ in row
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const int f = lane + 64 * j, r = f / (4 * MB), c = (f % (4 * MB)) * 4;
            st4(sa + r * SA + c, csum[j]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    float bsum[(16 * MB + 63) / 64];
#pragma unroll
    for (int u = 0; u < (16 * MB + 63) / 64; ++u) {
        const int c = lane + 64 * u;
        float sm = 0.f;
        if (T.bias != nullptr && c < 16 * MB)
#pragma unroll
            for (int r = 0; r < 16; ++r) sm += sa[r * SA + c];
        bsum[u] = sm;
    }
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* dst = red + (MB * (4 * q + r) + mb) * LDP + NB * i;
#pragma unroll
                    for (int h = 0; h < NB / 4; ++h) {
                        f32x4 v = f32x4{acc[mb][4 * h][r], acc[mb][4 * h + 1][r], acc[mb][4 * h + 2][r], acc[mb][4 * h + 3][r]};
                        if (w > 0) v += ld4(dst + 4 * h);
                        st4(dst + 4 * h, v);
                    }
                }
#pragma unroll
            for (int u = 0; u < (16 * MB + 63) / 64; ++u) {
                const int c = lane + 64 * u;
                if (c < 16 * MB) redb[c] = w > 0 ? redb[c] + bsum[u] : bsum[u];
            }
        }
    }
    __syncthreads();
    float* dst = partial + ((size_t)T.part0 + blockIdx.x) * OUTER_PART;
    for (int f = threadIdx.x; f < 16 * MB * LDP / 4; f += 256) st4(dst + 4 * f, ld4(red + 4 * f));
    if (T.bias != nullptr && (int)threadIdx.x < 16 * MB) dst[8192 + threadIdx.x] = redb[threadIdx.x];
}

// grid = (1 + 16MB*16NB/256, n_tasks), 1024 threads: 256 elements x 4 chunk groups.  Each group adds
// its contiguous quarter of the chunks in order; the four group sums are combined in order through
// LDS: a fixed summation tree, hence deterministic.  The last block of a task reduces the bias row.
__device__ __forceinline__ void outer_reduce_body(const OuterBatch& batch, const float* __restrict__ partial, int bx,
                                                  int by) {
    const OuterTask T = batch.t[by];
    const int MBn = (T.M + 15) >> 4, NBn = (T.N + 15) >> 4;
    const int ldp = NBn > 4 ? 128 : 64, mpad = MBn > 4 ? 128 : 64;
    const int nblk = mpad * ldp / 256;
    if (bx > nblk) return;
    const bool is_bias = bx == nblk;
    if (is_bias && T.bias == nullptr) return;
    const int grp = threadIdx.x >> 8, e = threadIdx.x & 255;
    const int off = is_bias ? 8192 + e : bx * 256 + e;
    const int m = is_bias ? e : off / ldp, n = is_bias ? 0 : off % ldp;
    const bool valid = is_bias ? e < T.M : (m < T.M && n < T.N);
    const int per = (T.chunks + 3) / 4;
    const int c0 = grp * per, c1 = c0 + per < T.chunks ? c0 + per : T.chunks;
    const float* src0 = partial + (size_t)T.part0 * OUTER_PART + off;
    float s = 0.0f;
    if (valid) {
        int ch = c0;
        for (; ch + 8 <= c1; ch += 8) {                     // 8 chunk loads in flight, summed in chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src0[(size_t)(ch + u) * OUTER_PART];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; ch < c1; ++ch) s += src0[(size_t)ch * OUTER_PART];
    }
    __shared__ float red[4][256];
    red[grp][e] = s;
    __syncthreads();
    if (grp == 0 && valid) {
        const float tot = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
        if (is_bias) T.bias[e] = tot;
        else T.C[(size_t)m * T.ldc + n] = tot;
    }
}
__global__ void __launch_bounds__(1024)
k_outer_reduce(OuterBatch batch, const float* __restrict__ partial) {
    outer_reduce_body(batch, partial, (int)blockIdx.x, (int)blockIdx.y);
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
// in: g = dL/dout.  out: dx4 and the row tensors of the weight gradients.  Weights are read from L2 in fragment shape.
// Four waves per 16-node tile, the output rows of every product split over them (as kb_node below): one wave per tile
// chained 4 x 64 dependent MFMAs behind 64 weight-fragment loads (11.5 us per launch at 2,560 nodes; now 6.5).  A wave owns
// 16 of the 64 rows of each product, requests all its fragments up front and hands its rows on through LDS.
template <int D>
__global__ void __launch_bounds__(256)
kb_out(AetherParams P, BwdWT WT, const float* __restrict__ x4, const float* __restrict__ nodeinfo,
        const float* __restrict__ g_out, float* __restrict__ DX, float* __restrict__ O1,
        float* __restrict__ O2, float* __restrict__ DPO1, float* __restrict__ DPO2,
        float* __restrict__ DY, int64_t n_nodes, const float* __restrict__ drop1 = nullptr,
        const float* __restrict__ drop2 = nullptr, const int* __restrict__ dropword = nullptr) {
    using NI = NodeInfo<D>;
    __shared__ __attribute__((aligned(16))) float sbuf[3][16 * LDW];      // o1 | d2 | d1 rows of the tile's 16 nodes
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    const int row = 16 * wave + i;                                        // the weight row this lane's fragments come from
    // every fragment of the wave's rows, requested before the first MFMA
    f32x4 w0f[4], w3f[4], w3tf[4], w0tf[4], xt[4];
    load_tile64(xt, x4, nc, H, q);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        w0f[a] = ld4(P.out_w0 + (size_t)row * H + 16 * a + 4 * q);
        w3f[a] = ld4(P.out_w3 + (size_t)row * H + 16 * a + 4 * q);
        w3tf[a] = ld4(WT.out_w3t + (size_t)row * H + 16 * a + 4 * q);
        w0tf[a] = ld4(WT.out_w0t + (size_t)row * H + 16 * a + 4 * q);
    }
    const f32x4 w6f = ld4(WT.out_w6t + row * 16 + 4 * q);
    f32x4 p1 = ld4(P.out_b0 + 16 * wave + 4 * q), p2 = ld4(P.out_b3 + 16 * wave + 4 * q);
    // dropout masks of the forward (scale 0 or 1 / (1 - p) per node and channel), when it applied any
    f32x4 m1 = f32x4{1.f, 1.f, 1.f, 1.f}, m2 = m1;
    if (dropword != nullptr && *dropword != 0) {
        m1 = ld4(drop1 + nc * H + 16 * wave + 4 * q);
        m2 = ld4(drop2 + nc * H + 16 * wave + 4 * q);
    }
    // dy = R^T g (rows 0..D-1 of a 16-row block: lanes q == 0, registers 0..D-1); every wave computes it
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
    auto full = [&](const float* buf, f32x4 (&t)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a) t[a] = ld4(buf + i * LDW + 16 * a + 4 * q);
    };
    auto mine = [&](float* buf, const f32x4 v) { st4(buf + i * LDW + 16 * wave + 4 * q, v); };
    auto out = [&](float* dst, const f32x4 v) { if (ok) st4(dst + node * H + 16 * wave + 4 * q, v); };
    // forward recompute: o1 = silu(W0 x + b0), o2 = silu(W3 o1 + b3)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) p1 = mfma16(w0f[a][b], xt[a][b], p1);
    const f32x4 o1 = silu4(p1) * m1;
    mine(sbuf[0], o1);
    out(O1, o1);
    __syncthreads();
    f32x4 t[4];
    full(sbuf[0], t);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) p2 = mfma16(w3f[a][b], t[a][b], p2);
    out(O2, silu4(p2) * m2);
    // d2 = (Wo6^T dy) * silu'(p2)   (K = 16, only k < D non-zero)
    f32x4 d2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) d2 = mfma16(w6f[b], dy[b], d2);
    d2 = d2 * dsilu4(p2) * m2;
    mine(sbuf[1], d2);
    out(DPO2, d2);
    __syncthreads();
    full(sbuf[1], t);
    f32x4 d1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) d1 = mfma16(w3tf[a][b], t[a][b], d1);
    d1 = d1 * dsilu4(p1) * m1;
    mine(sbuf[2], d1);
    out(DPO1, d1);
    __syncthreads();
    full(sbuf[2], t);
    f32x4 dx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) dx = mfma16(w0tf[a][b], t[a][b], dx);
    out(DX, dx);
    if (ok && wave == 0) st4(DY + node * 16 + 4 * q, dy);
}

// Node update backward of one layer (locs.py:240-241): x = n + W4 silu(W3 n + b3) + b4.
// One workgroup of 4 waves per 16-node tile, output rows split over the waves (a single wave would
// chain 384 dependent-latency MFMAs behind 96 weight-fragment loads: 15 us per launch at 2,560 nodes):
//   stage A: wave w owns rows 32w .. 32w+31 of pre_u = W3 n + b3 and of du = W4^T dx
//            -> U = silu(pre_u), DPU = du * silu'(pre_u) (to global for the weight gradients, to LDS for B)
//   stage B: wave w owns rows 16w .. 16w+15 of dn = dx + W3^T dpre_u
// Every weight fragment is requested before the first MFMA.
__global__ void __launch_bounds__(256)
kb_node(const float* __restrict__ w3g, const float* __restrict__ b3g, const float* __restrict__ w4t,
        const float* __restrict__ w3t, const float* __restrict__ nbuf, const float* __restrict__ DX,
        float* __restrict__ DN, float* __restrict__ U, float* __restrict__ DPU, int64_t n_nodes) {
    __shared__ __attribute__((aligned(16))) float dpu_s[16 * LDU];     // [node][128 + 8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    f32x4 nt[4], dx[4], w3f[2][4], w4f[2][4], w3tf[8], pu[2], du[2];
    load_tile64(nt, nbuf, nc, H, q);
    load_tile64(dx, DX, nc, H, q);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mb = 2 * wave + m;
        pu[m] = ld4(b3g + 16 * mb + 4 * q);
        du[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            w3f[m][a] = ld4(w3g + (size_t)(16 * mb + i) * H + 16 * a + 4 * q);
            w4f[m][a] = ld4(w4t + (size_t)(16 * mb + i) * H + 16 * a + 4 * q);
        }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) w3tf[a] = ld4(w3t + (size_t)(16 * wave + i) * (2 * H) + 16 * a + 4 * q);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                pu[m] = mfma16(w3f[m][a][b], nt[a][b], pu[m]);
                du[m] = mfma16(w4f[m][a][b], dx[a][b], du[m]);
            }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int mb = 2 * wave + m;
        const f32x4 sg = sigmoid4(pu[m]);
        const f32x4 u = pu[m] * sg;
        const f32x4 dpu = du[m] * dsilu_from_sigmoid(pu[m], sg);
        st4(dpu_s + i * LDU + 16 * mb + 4 * q, dpu);
        if (ok) {
            st4(U + node * 2 * H + 16 * mb + 4 * q, u);
            st4(DPU + node * 2 * H + 16 * mb + 4 * q, dpu);
        }
    }
    __syncthreads();
    f32x4 dn = dx[wave];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const f32x4 bv = ld4(dpu_s + i * LDU + 16 * a + 4 * q);
#pragma unroll
        for (int b = 0; b < 4; ++b) dn = mfma16(w3tf[a][b], bv[b], dn);
    }
    if (ok) st4(DN + node * H + 16 * wave + 4 * q, dn);
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
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    if (t >= tiles) return;
    // One memory round trip per tile: the indices (and the receiver's in-degree) of the NEXT tile are
    // fetched while this one computes, and every row this tile needs is requested at the top of the
    // iteration; each sigmoid is evaluated once and serves both SiLU and its derivative.
    auto clampk = [&](int64_t tt) { const int64_t k = 16 * tt + i; return k < n_edges ? k : n_edges - 1; };
    int64_t kc = clampk(t);
    int32_t s = send_s[kc], r = recv_s[kc];
    int deg = rowptr[r + 1] - rowptr[r];
    for (; t < tiles; t += stride) {
        int z = 0;
        asm volatile("" : "+v"(z));      // opaque offset: weight fragments are re-read from LDS per tile
                                         // instead of being hoisted into ~256 registers (1 wave per SIMD)
        const int64_t k = 16 * t + i;
        const bool ok = k < n_edges;
        // ---- this tile's rows
        f32x4 bop[4], p1[4], dnv[4], dev[4];
        if (FIRST) {
            bop[0] = ld4(feat + kc * FPAD + 4 * q);
            bop[1] = ld4(feat + kc * FPAD + 16 + 4 * q);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(b_in + 16 * mb + 4 * q);
        } else {
            load_tile64(bop, e_prev, kc, H, q);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                p1[mb] = ld4(Ps + (int64_t)s * H + 16 * mb + 4 * q) + ld4(Pr + (int64_t)r * H + 16 * mb + 4 * q);
        }
        load_tile64(dnv, DN, r, H, q);
        if (have_de) load_tile64(dev, DE, kc, H, q);
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        // ---- next tile's indices
        const int64_t kn = clampk(t + stride < tiles ? t + stride : t);
        const int32_t s_n = send_s[kn], r_n = recv_s[kn];
        const int deg_n = rowptr[r_n + 1] - rowptr[r_n];
        // ---- forward recompute: pre1, h = silu(pre1), pre2
        f32x4 p2[4], h[4], sg1[4];
        if (FIRST) {
            f32x4 b2[2] = {bop[0], bop[1]};
            gemm_tile<4, 2>(wi + z, LDW, b2, p1, i, q);
        } else {
            gemm_tile<4, 4>(wi + z, LDW, bop, p1, i, q);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            sg1[mb] = sigmoid4(p1[mb]);
            h[mb] = p1[mb] * sg1[mb];
            p2[mb] = ld4(b2g + 16 * mb + 4 * q);
        }
        gemm_tile<4, 4>(w2 + z, LDW, h, p2, i, q);
        // ---- de = dn[recv] / deg (+ gradient through the next layer's edge input); back through both Linears
        f32x4 d2[4], dh[4], g[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 de = dnv[mb] * inv;
            if (have_de) de += dev[mb];
            d2[mb] = de * dsilu_from_sigmoid(p2[mb], sigmoid4(p2[mb]));
            dh[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        gemm_tile<4, 4>(w2ts + z, LDW, d2, dh, i, q);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) g[mb] = dh[mb] * dsilu_from_sigmoid(p1[mb], sg1[mb]);
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
        kc = kn; s = s_n; r = r_n; deg = deg_n;
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
kb_gather_rows(const float* __restrict__ w1t /*[192][64]: W_s^T | W_r^T | W_e^T*/,
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

// dL/dx_{l-1} of layers 2-4 in one launch per 16-node tile (16 waves); used up to an average degree of
// 64 (above it the row sums dominate and the two kernels above, with 4-wave workgroups, stream better):
//   phase 1, wave w = node 16*tile + w: DPR = sum of G over the node's in-edges (contiguous rows of the
//            receiver-sorted G), DPS = sum over its out-edges (rows listed by sperm / srowptr); lane =
//            column, 8 / 4 row loads in flight, fixed order -> deterministic.  Both go to global memory
//            (operands of the weight gradients) and to LDS.
//   phase 2, waves 0-3 (rows 16w .. 16w+15): dx = dn + W_s^T DPS + W_r^T DPR (locs.py:233 split), the
//            weight fragments were requested before phase 1.
__global__ void __launch_bounds__(1024)
kb_gather(const float* __restrict__ G, const int32_t* __restrict__ rowptr,
          const int32_t* __restrict__ srowptr, const int32_t* __restrict__ sperm,
          const float* __restrict__ w1t /*[192][64]: W_s^T | W_r^T | W_e^T*/, const float* __restrict__ DN,
          float* __restrict__ DPS, float* __restrict__ DPR, float* __restrict__ DX, int64_t n_nodes) {
    __shared__ __attribute__((aligned(16))) float sums[2][16 * LDW];      // [DPS | DPR][node][64 + 8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    f32x4 wsf[4], wrf[4], dx;
    const int64_t tnode = (int64_t)blockIdx.x * 16 + i;
    const int64_t tc = tnode < n_nodes ? tnode : n_nodes - 1;
    if (wave < 4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            wsf[a] = ld4(w1t + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
            wrf[a] = ld4(w1t + (size_t)H * H + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
        }
        dx = ld4(DN + tc * H + 16 * wave + 4 * q);
    }
    const int64_t node = (int64_t)blockIdx.x * 16 + wave;
    float sr = 0.f, ss = 0.f;
    if (node < n_nodes) {
        {
            const int beg = rowptr[node], end = rowptr[node + 1];
            const float* p = G + (int64_t)beg * H + lane;
            int k = beg;
            for (; k + 8 <= end; k += 8, p += 8 * H) {
                const float a0 = p[0], a1 = p[H], a2 = p[2 * H], a3 = p[3 * H], a4 = p[4 * H], a5 = p[5 * H],
                            a6 = p[6 * H], a7 = p[7 * H];
                sr += a0; sr += a1; sr += a2; sr += a3; sr += a4; sr += a5; sr += a6; sr += a7;
            }
            for (; k < end; ++k, p += H) sr += p[0];
            DPR[node * H + lane] = sr;
        }
        {
            const int beg = srowptr[node], end = srowptr[node + 1];
            int k = beg;
            for (; k + 4 <= end; k += 4) {
                const int r0 = sperm[k], r1 = sperm[k + 1], r2 = sperm[k + 2], r3 = sperm[k + 3];
                const float a0 = G[(int64_t)r0 * H + lane], a1 = G[(int64_t)r1 * H + lane],
                            a2 = G[(int64_t)r2 * H + lane], a3 = G[(int64_t)r3 * H + lane];
                ss += a0; ss += a1; ss += a2; ss += a3;
            }
            for (; k < end; ++k) ss += G[(int64_t)sperm[k] * H + lane];
            DPS[node * H + lane] = ss;
        }
    }
    sums[0][wave * LDW + lane] = ss;
    sums[1][wave * LDW + lane] = sr;
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 bs = ld4(sums[0] + i * LDW + 16 * a + 4 * q);
            const f32x4 br = ld4(sums[1] + i * LDW + 16 * a + 4 * q);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                dx = mfma16(wsf[a][b], bs[b], dx);
                dx = mfma16(wrf[a][b], br[b], dx);
            }
        }
        if (tnode < n_nodes) st4(DX + tnode * H + 16 * wave + 4 * q, dx);
    }
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

// EXT: the field was supplied from outside (aether_forward_field): only dL/df is produced (grad_field [n][D]).
template <int D, bool EXT>
__global__ void __launch_bounds__(256)
kb_field(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel,
         const float* __restrict__ charges, const float* __restrict__ nodeinfo,
         const float* __restrict__ DA, const float* __restrict__ DN1,
         const int32_t* __restrict__ rowptr, const int32_t* __restrict__ recv_s,
         const int32_t* __restrict__ srowptr, const int32_t* __restrict__ sperm,
         float* __restrict__ RELF, float* __restrict__ Z, float* __restrict__ H1f,
         float* __restrict__ H2f, float* __restrict__ DPH1, float* __restrict__ DPH2,
         float* __restrict__ DF, float* __restrict__ DZE, float* __restrict__ ONEHOT, float* __restrict__ grad_field,
         int64_t n_nodes, int hid = H /* width of DN1 / rows of W_res (wide.h) */) {
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
    for (int o = t; o < hid; o += 32) {
        const float gg = DN1[nc * hid + o];
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
    if (EXT) {
        if (ok && t < 16) {
            float rv = 0.f;                                       // rel_feat row for dW_res
            if (t >= D && t < 2 * D) rv = ni[NI::CV + t - D];
            if (t >= 2 * D && t < 3 * D) rv = ni[NI::CF + t - 2 * D];
            RELF[n * 16 + t] = rv;
#pragma unroll
            for (int d = 0; d < D; ++d) if (t == d) grad_field[n * D + d] = df[d];
        }
        return;
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
