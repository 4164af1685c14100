// streamed.h -- layer-by-layer kernels: the general path (any graph size); every [E,64] message
// tensor streams through HBM/L2 once per layer in receiver-sorted order.
#pragma once
#include "common.h"

namespace {

// ------------------------------------------------------------------ K1: node preparation
// Field query (aether.py:108-134), frame + canonical velocity / force (geometry.py:7-73,
// aether.py:33-50) and x0 = layer_1.res(rel_feat) (locs.py:214-218) on the matrix core: a wave owns 16
// nodes as the columns of its MFMA tiles; the three Linear layers of the field net chain in accumulator
// layout, lanes q == 0 end up with the node's force and build its NodeInfo record, x0 is one more
// tile product.  Every global load (inputs, the 2,080 field parameters, res weights) is issued up front.
template <int D>
__global__ void __launch_bounds__(64)
k_node_prep(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel,
            const float* __restrict__ charges, float* __restrict__ nodeinfo,
            float* __restrict__ x0, const float* __restrict__ ext_field, int64_t n_nodes) {
    using NI = NodeInfo<D>;
    constexpr int FIN = 2 * D + 16;
    __shared__ float rel[16 * 8];                              // [node][cv | cf] for the x0 operand
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool live = node < n_nodes;
    const int64_t g = live ? node : n_nodes - 1;
    float pz[D], vz[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { pz[d] = x[g * D + d]; vz[d] = vel[g * D + d]; }
    const float ch = charges[g];
    // layer-1 operands: B = z[k][node], A = W0[16mb + i][k], k = 4s + q (FIN = 2D + 16 <= 24)
    float zc[6][3], a1[2][6];
#pragma unroll
    for (int s4 = 0; s4 < 6; ++s4) {
        const int k = 4 * s4 + q;
#pragma unroll
        for (int c = 0; c < 3; ++c) zc[s4][c] = (k >= 2 * D && k < FIN) ? P.field_emb[c * 16 + (k - 2 * D)] : 0.0f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) a1[mb][s4] = k < FIN ? P.field_w0[(16 * mb + i) * FIN + k] : 0.0f;
    }
    f32x4 w2f[2][2], w4f[2], acc1[2], acc2[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        acc1[mb] = ld4(P.field_b0 + 16 * mb + 4 * q);
        acc2[mb] = ld4(P.field_b2 + 16 * mb + 4 * q);
#pragma unroll
        for (int a = 0; a < 2; ++a) w2f[mb][a] = ld4(P.field_w2 + (16 * mb + i) * 32 + 16 * a + 4 * q);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) w4f[a] = i < D ? ld4(P.field_w4 + i * 32 + 16 * a + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc3[r] = (4 * q + r < D) ? P.field_b4[4 * q + r] : 0.0f;
    // x0 operands: A = W_res[16mb + i][D + k], k = 4s + q < 2D (the first D inputs of rel_feat are zero)
    float ar[4][2];
    f32x4 acc0[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        acc0[mb] = ld4(P.l1_res_b + 16 * mb + 4 * q);
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4)
            ar[mb][s4] = 4 * s4 + q < 2 * D ? P.l1_res_w[(16 * mb + i) * 3 * D + D + 4 * s4 + q] : 0.0f;
    }
    long ci = (long)(ch + 1.0f);                               // charge_to_index: (q + 1).long(), aether.py:122-124
    ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
#pragma unroll
    for (int s4 = 0; s4 < 6; ++s4) {
        const int k = 4 * s4 + q;
        float zk = ci == 0 ? zc[s4][0] : (ci == 1 ? zc[s4][1] : zc[s4][2]);
#pragma unroll
        for (int d = 0; d < D; ++d) { zk = k == d ? pz[d] : zk; zk = k == D + d ? vz[d] : zk; }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) acc1[mb] = mfma16(a1[mb][s4], zk, acc1[mb]);
    }
    f32x4 hh[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) hh[mb] = silu4(acc1[mb]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc2[mb] = mfma16(w2f[mb][a][b], hh[a][b], acc2[mb]);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) hh[mb] = silu4(acc2[mb]);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc3 = mfma16(w4f[a][b], hh[a][b], acc3);
    // lanes q == 0: acc3[d] = force component d of node i -> frame + NodeInfo record
    if (q == 0) {
        float f[D], R[D][D], cv[D], cf[D];
#pragma unroll
        for (int d = 0; d < D; ++d) f[d] = ext_field ? ext_field[g * D + d] : acc3[d];
        node_frame<D>(vz, f, R, cv, cf);
#pragma unroll
        for (int d = 0; d < D; ++d) { rel[i * 8 + d] = cv[d]; rel[i * 8 + D + d] = cf[d]; }
        if (live) {
            float* ni = nodeinfo + node * NI::STRIDE;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                ni[NI::P + d] = pz[d]; ni[NI::V + d] = vz[d]; ni[NI::F + d] = f[d];
                ni[NI::CV + d] = cv[d]; ni[NI::CF + d] = cf[d];
#pragma unroll
                for (int e = 0; e < D; ++e) ni[NI::R + d * D + e] = R[d][e];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();                           // one wave: its LDS operations execute in order
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
        const int k = 4 * s4 + q;
        const float rk = k < 2 * D ? rel[i * 8 + k] : 0.0f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc0[mb] = mfma16(ar[mb][s4], rk, acc0[mb]);
    }
    if (live && x0 != nullptr) {            // null: the caller computes x0 itself (wide.h, hidden > 64)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) st4(x0 + node * H + 16 * mb + 4 * q, acc0[mb]);
    }
}

// ------------------------------------------------------------------ per-tile receiver sums
// Same matrix-core segment sum as the fused kernel (fused.h), with the partial rows going to
// global memory: row (receiver + tile) of `part` gets the sum of the tile's messages for that
// receiver.  `gs` is this lane's word of the graph's tile table (k_graph_gtiles): bits 0-3 the 0/1
// column of the segment matrix, then per result register (first row : 4 bits, valid : 1 bit).
// `wst` = 16 wave-private LDS rows of LDST floats; `rcv` = receiver of this lane's edge.
__device__ __forceinline__ void tile_receiver_sums(const f32x4 (&e)[4], float* wst, unsigned gs, int rcv,
                                                   int64_t tile, float* __restrict__ part, int i, int q,
                                                   int lane) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) st4(wst + i * LDST + 16 * mb + 4 * q, e[mb]);
    __builtin_amdgcn_wave_barrier();
    f32x4 red[4];
#pragma unroll
    for (int nbk = 0; nbk < 4; ++nbk) red[nbk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float sel = (gs >> s4) & 1u ? 1.0f : 0.0f;
        const float* erow = wst + (4 * s4 + q) * LDST + i;
#pragma unroll
        for (int nbk = 0; nbk < 4; ++nbk) red[nbk] = mfma16(sel, erow[16 * nbk], red[nbk]);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const unsigned f = gs >> (4 + 5 * r4);
        const int node = __shfl(rcv, (lane & 48) + (int)(f & 15u));     // receiver of the segment's first row
        if (f & 16u) {
            float* dst = part + ((int64_t)node + tile) * H + i;
            dst[0] = red[0][r4]; dst[16] = red[1][r4]; dst[32] = red[2][r4]; dst[48] = red[3][r4];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------ K1: layer-1 edge kernel
// Phase A (one thread per edge): local-frame edge features, aether.py:52-100 +
// geometry.py:76-101, followed by [rel_feat[recv] | edge_attr_orig] (aether.py:99,177).
// Phase B (one wave per 16-edge tile): e1 = SiLU(W2 SiLU(W1 a + b1) + b2), locs.py:206-212.
template <int D>
__global__ void __launch_bounds__(256, 2)
k_edge_layer1(AetherParams P, const float* __restrict__ nodeinfo,
              const float* __restrict__ edge_attr_orig, const int32_t* __restrict__ perm,
              const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
              const uint32_t* __restrict__ gsel, float* __restrict__ part, float* __restrict__ e_out,
              float* __restrict__ feat_dbg, const float* __restrict__ qattr, int64_t n_edges) {
    using NI = NodeInfo<D>;
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w1 = smem;                       // split image of W1, K padded to 32 (half an image; common.h gemm_split)
    float* w2 = w1 + SPLIT_WIMG / 2;        // split image of W2
    float* bias = w2 + SPLIT_WIMG;          // [128]: b1 | b2
    float* scratch = bias + 2 * H;          // [4 waves][64][LDF]: features of the wave's 64 edges, then
                                            // (once they sit in registers) its 16 tile-staging rows
    for (int idx = threadIdx.x; idx < H * FPAD / 4; idx += 256) {
        const int rr = idx >> 3, c0 = (idx & 7) * 4;
        f32x4 v;
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = c0 + b < F1 ? P.l1_msg_w0[rr * F1 + c0 + b] : 0.0f;
        stage_split4<4, 1>(w1, rr, c0, v);
    }
    for (int idx = threadIdx.x; idx < H * H / 4; idx += 256) {
        const int rr = idx >> 4, cc = (idx & 15) * 4;
        stage_split4<4, 2>(w2, rr, cc, ld4(P.l1_msg_w2 + (size_t)rr * H + cc));
    }
    if (threadIdx.x < H) {
        bias[threadIdx.x] = P.l1_msg_b0[threadIdx.x];
        bias[H + threadIdx.x] = P.l1_msg_b2[threadIdx.x];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    float* wfeat = scratch + wave * (64 * LDF);
    // each wave works on its own 64-edge batches: every lane builds one edge's features, then the
    // wave runs the batch's four 16-edge tiles; no workgroup barrier in the loop
    const int64_t n_batches = (n_edges + 63) / 64;
    for (int64_t batch = (int64_t)blockIdx.x * 4 + wave; batch < n_batches; batch += (int64_t)gridDim.x * 4) {
        {
            const int64_t k = batch * 64 + lane;
            float o[FPAD];
            if (k < n_edges) {
                const float* nj = nodeinfo + (int64_t)send_s[k] * NI::STRIDE;
                const float* nr = nodeinfo + (int64_t)recv_s[k] * NI::STRIDE;
                float njl[NI::STRIDE], nrl[NI::STRIDE];
#pragma unroll
                for (int t = 0; t < NI::STRIDE; t += 4) {
                    f32x4 a = ld4(nj + t), b = ld4(nr + t);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { njl[t + u] = a[u]; nrl[t + u] = b[u]; }
                }
                float eal[2];
                if (qattr) {                   // main.py:243-246: [q_i q_j, sqrt(sum((x_i - x_j)^2))]
                    float d2 = 0.0f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float df = njl[NI::P + d] - nrl[NI::P + d];
                        d2 += df * df;
                    }
                    eal[0] = qattr[send_s[k]] * qattr[recv_s[k]];
                    eal[1] = sqrtf(d2);
                } else {
                    const float* ea = edge_attr_orig + 2 * (int64_t)perm[k];
                    eal[0] = ea[0]; eal[1] = ea[1];
                }
                edge_features<D>(njl, nrl, eal, o);
                if (feat_dbg) {
#pragma unroll
                    for (int t = 0; t < FPAD; ++t) feat_dbg[k * FPAD + t] = o[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < FPAD; ++t) o[t] = 0.0f;
            }
            float* fr = wfeat + lane * LDF;
#pragma unroll
            for (int t = 0; t < FPAD; t += 4) st4(fr + t, f32x4{o[t], o[t + 1], o[t + 2], o[t + 3]});
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 bop[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bop[t][0] = ld4(wfeat + (16 * t + i) * LDF + 4 * q);
            bop[t][1] = ld4(wfeat + (16 * t + i) * LDF + 16 + 4 * q);
        }
        __builtin_amdgcn_wave_barrier();    // features are in registers: the rows become tile staging
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t k = batch * 64 + 16 * t + i;
            if (batch * 64 + 16 * t < n_edges) {                       // wave-uniform
                const int64_t tile = batch * 4 + t;
                f32x4 acc[4], acc2[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    acc[mb] = ld4(bias + 16 * mb + 4 * q);
                    acc2[mb] = ld4(bias + H + 16 * mb + 4 * q);
                }
                gemm_split<4, 1>(w1, bop[t], acc, lane);
                f32x4 h1[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) h1[mb] = silu4(acc[mb]);
                gemm_split<4, 2>(w2, h1, acc2, lane);
                f32x4 eo[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) eo[mb] = silu4(acc2[mb]);
                if (k < n_edges) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) st4(e_out + k * H + 16 * mb + 4 * q, eo[mb]);
                }
                const int rcv = recv_s[k < n_edges ? k : n_edges - 1];
                tile_receiver_sums(eo, wfeat, gsel[tile * 64 + lane], rcv, tile, part, i, q, lane);
            }
        }
    }
}

// ------------------------------------------------------------------ K3: layers 2-4 edge kernel
// e_l = SiLU(W2 SiLU(W_s x_s + W_r x_r + b1 + W_e e_{l-1}) + b2), locs.py:227-235 with the
// node terms P_s = W_s x, P_r = W_r x + b1 gathered as the accumulator's initial value.
// Both contractions run as six bf16 MFMA terms on 3-way split operands (common.h, gemm_split: fp32-equivalent): with
// the fp32 MFMA this kernel kept the matrix pipe -- which for fp32 IS the vector ALU -- busy 71 % of the time at
// 0.32 of the HBM roof (profiles/r02_cfg5shard_pmc_fp32mfma.txt); the bf16 terms run on the matrix pipe proper, beside
// the SiLU / split VALU work of the SIMD's other wave.  The weights' split images are built once per workgroup in LDS
// (2 x 24 KB; two workgroups per CU) and their fragments re-read per tile.  Inputs of the next tile (indices two tiles ahead, gathered rows one
// tile ahead) are in flight while the current tile's 128 MFMAs issue, and the finished tile is
// stored one iteration late, right after the next loads are issued: LLVM waits vmcnt(0) whenever
// loads and stores are both pending (they may retire out of order), so a store issued just before
// the loop-top wait would expose its full write latency on every tile.
constexpr int EDGE_LN_WAVES = 12;        // waves per workgroup of k_edge_layer: one workgroup per CU, 3 waves per SIMD
__global__ void __launch_bounds__(64 * EDGE_LN_WAVES)
k_edge_layer(const float* __restrict__ w_msg0, const float* __restrict__ w_msg2,
             const float* __restrict__ b_msg2, const float* __restrict__ Ps,
             const float* __restrict__ Pr, const float* __restrict__ e_prev,
             const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
             const uint32_t* __restrict__ gsel, float* __restrict__ part,
             float* __restrict__ e_out /* null: the messages themselves are not needed (layer 4) */,
             int64_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* we = smem;                       // split image of W_e = W1[:, 128:192]
    float* w2 = we + SPLIT_WIMG;            // split image of W2
    float* bias = w2 + SPLIT_WIMG;          // [64] b2
    float* wst = bias + H + (threadIdx.x >> 6) * (16 * LDST);     // [waves][16][LDST] tile staging
    for (int idx = threadIdx.x; idx < H * H / 4; idx += 64 * EDGE_LN_WAVES) {
        const int rr = idx >> 4, cc = (idx & 15) * 4;
        stage_split4<4, 2>(we, rr, cc, ld4(w_msg0 + (size_t)rr * (3 * H) + 2 * H + cc));
        stage_split4<4, 2>(w2, rr, cc, ld4(w_msg2 + (size_t)rr * H + cc));
    }
    if (threadIdx.x < H) bias[threadIdx.x] = b_msg2[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t n_tiles = (n_edges + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * EDGE_LN_WAVES;
    f32x4 b2v[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) b2v[mb] = ld4(bias + 16 * mb + 4 * q);

    int64_t tile = (int64_t)blockIdx.x * EDGE_LN_WAVES + wave;
    if (tile >= n_tiles) return;
    auto clampk = [&](int64_t t) { int64_t k = t * 16 + i; return k < n_edges ? k : n_edges - 1; };
    // software pipeline: (s1, r1) = indices of tile+stride; (ps, pr, ev) = rows of the current tile
    int64_t kc = clampk(tile);
    int32_t s0 = send_s[kc], r0 = recv_s[kc];
    f32x4 psv[4], prv[4], ev[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        psv[mb] = ld4(Ps + (int64_t)s0 * H + 16 * mb + 4 * q);
        prv[mb] = ld4(Pr + (int64_t)r0 * H + 16 * mb + 4 * q);
        ev[mb] = ld4(e_prev + kc * H + 16 * mb + 4 * q);
    }
    int64_t kn = clampk(tile + stride < n_tiles ? tile + stride : tile);
    int32_t s1 = send_s[kn], r1 = recv_s[kn];
    int32_t rcur = r0;                      // receiver of this lane's edge in the current tile
    f32x4 eo[4];
    int64_t ko = -1;
    for (; tile < n_tiles; tile += stride) {
        f32x4 acc[4], bop[4], acc2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { acc[mb] = psv[mb] + prv[mb]; bop[mb] = ev[mb]; acc2[mb] = b2v[mb]; }
        // issue the next tile's row gathers and the indices of the tile after it
        const int64_t k = tile * 16 + i;
        const int64_t kn2 = clampk(tile + 2 * stride < n_tiles ? tile + 2 * stride : tile);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            psv[mb] = ld4(Ps + (int64_t)s1 * H + 16 * mb + 4 * q);
            prv[mb] = ld4(Pr + (int64_t)r1 * H + 16 * mb + 4 * q);
            ev[mb] = ld4(e_prev + kn * H + 16 * mb + 4 * q);
        }
        const int32_t rnext = r1;
        const unsigned gs = gsel[tile * 64 + lane];
        kn = kn2;
        s1 = send_s[kn2];
        r1 = recv_s[kn2];
        if (e_out != nullptr && ko >= 0 && ko < n_edges) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) st4(e_out + ko * H + 16 * mb + 4 * q, eo[mb]);
        }
        int z = 0;
        asm volatile("" : "+v"(z));      // opaque offset: keeps the fragment reads inside the loop
        gemm_split<4, 2>(we + z, bop, acc, lane);
        f32x4 h1[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h1[mb] = silu4(acc[mb]);
        asm volatile("" : "+v"(z));
        gemm_split<4, 2>(w2 + z, h1, acc2, lane);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) eo[mb] = silu4(acc2[mb]);
        ko = k;
        tile_receiver_sums(eo, wst, gs, rcur, tile, part, i, q, lane);
        rcur = rnext;
    }
    if (e_out != nullptr && ko >= 0 && ko < n_edges) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) st4(e_out + ko * H + 16 * mb + 4 * q, eo[mb]);
    }
}

// ------------------------------------------------------------------ segmented mean by receiver
// torch_scatter.scatter(e, recv, reduce="mean") (locs.py:236-238) on the receiver-sorted messages:
// one wave per node, lane = column, rows added in tile (= edge) order (deterministic); 8 row loads in flight.
// The edge kernels have already reduced every 16-edge tile per receiver (tile_receiver_sums): a
// node's partial rows are rows node + t0 .. node + t1 of `part` (t = tiles its run touches).
__global__ void __launch_bounds__(256)
k_segment_mean(const float* __restrict__ part, const int32_t* __restrict__ rowptr,
               float* __restrict__ aggr, int64_t n_nodes) {
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    const int ebeg = rowptr[node], eend = rowptr[node + 1];
    const int beg = ebeg >> 4, end = eend > ebeg ? ((eend - 1) >> 4) + 1 : beg;    // tiles of the run
    const float* p = part + (node + (int64_t)beg) * H + lane;
    float s = 0.f;
    int k = beg;
    for (; k + 8 <= end; k += 8, p += 8 * H) {
        const float a0 = p[0], a1 = p[H], a2 = p[2 * H], a3 = p[3 * H], a4 = p[4 * H], a5 = p[5 * H],
                    a6 = p[6 * H], a7 = p[7 * H];
        s += a0; s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;
    }
    for (; k < end; ++k, p += H) s += p[0];
    const float deg = (float)(eend - ebeg > 1 ? eend - ebeg : 1);    // count clamped to >= 1
    aggr[node * H + lane] = s / deg;
}

// Node update of one layer (locs.py:240-241) for a 16-node tile, output rows split over the 4 waves of
// the workgroup (a single wave would chain ~400 MFMAs behind a hundred weight-fragment loads):
//   A: u = SiLU(W3 n + b3), n = x_prev + mean        wave w: rows 32w .. 32w+31 of the 128     -> LDS
//   B: x = n + W4 u + b4                              wave w: rows 16w .. 16w+15                -> LDS, x_out
//   C: next layer's P_s = W_s x, P_r = W_r x + b1     wave w: rows 16w .. of both               (locs.py:233 split)
//      LAST: out MLP (locs.py:160-168,193), rows 16w .. of each hidden layer, then wave 0 globalises
//      (local_to_global.py:12-13) and adds the residual (aether.py:185).
// Every weight fragment a wave needs is requested before its first MFMA.
template <int D, bool LAST>
__global__ void __launch_bounds__(256)
k_node_update(AetherParams P, int layer /*1..4*/, const float* __restrict__ x_prev,
              const float* __restrict__ aggr,
              float* __restrict__ x_out, float* __restrict__ Ps, float* __restrict__ Pr,
              const float* __restrict__ nodeinfo, const float* __restrict__ pos,
              float* __restrict__ out, float* __restrict__ nsave, float* __restrict__ vel_out, float dt,
              int64_t n_nodes, const float* __restrict__ drop1 = nullptr, const float* __restrict__ drop2 = nullptr,
              int* __restrict__ dropword = nullptr) {
    using NI = NodeInfo<D>;
    constexpr int LDUU = 2 * H + 8;
    __shared__ __attribute__((aligned(16))) float ubuf[16 * LDUU];       // u, later o1 | o2 (cols 0-63 | 64-127)
    __shared__ __attribute__((aligned(16))) float xbuf[16 * LDW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const bool ok = node < n_nodes;
    const int64_t nc = ok ? node : n_nodes - 1;
    const float* w3 = layer == 1 ? P.l1_upd_w0 : P.ln_upd_w0[layer - 2];
    const float* b3 = layer == 1 ? P.l1_upd_b0 : P.ln_upd_b0[layer - 2];
    const float* w4 = layer == 1 ? P.l1_upd_w2 : P.ln_upd_w2[layer - 2];
    const float* b4 = layer == 1 ? P.l1_upd_b2 : P.ln_upd_b2[layer - 2];
    // ---- all loads of the kernel
    f32x4 n[4], w3f[2][4], w4f[8], wcf[2][4], u[2], xn, c0v, c1v, w6f[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
        n[mb] = ld4(x_prev + nc * H + 16 * mb + 4 * q) + ld4(aggr + nc * H + 16 * mb + 4 * q);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        u[m] = ld4(b3 + 16 * (2 * wave + m) + 4 * q);
#pragma unroll
        for (int a = 0; a < 4; ++a) w3f[m][a] = ld4(w3 + (size_t)(16 * (2 * wave + m) + i) * H + 16 * a + 4 * q);
    }
    xn = ld4(b4 + 16 * wave + 4 * q);
#pragma unroll
    for (int a = 0; a < 8; ++a) w4f[a] = ld4(w4 + (size_t)(16 * wave + i) * (2 * H) + 16 * a + 4 * q);
    if constexpr (!LAST) {
        const float* w1n = P.ln_msg_w0[layer - 1];            // next layer's W1 [64][192]
        c0v = f32x4{0.f, 0.f, 0.f, 0.f};
        c1v = ld4(P.ln_msg_b0[layer - 1] + 16 * wave + 4 * q);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            wcf[0][a] = ld4(w1n + (size_t)(16 * wave + i) * (3 * H) + 16 * a + 4 * q);
            wcf[1][a] = ld4(w1n + (size_t)(16 * wave + i) * (3 * H) + H + 16 * a + 4 * q);
        }
    } else {
        c0v = ld4(P.out_b0 + 16 * wave + 4 * q);
        c1v = ld4(P.out_b3 + 16 * wave + 4 * q);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            wcf[0][a] = ld4(P.out_w0 + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
            wcf[1][a] = ld4(P.out_w3 + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
            // last Linear has D (2|3) output rows: rows >= D of the 16-row block read row D-1 and are discarded
            if (wave == 0) w6f[a] = ld4(P.out_w6 + (size_t)(i < D ? i : D - 1) * H + 16 * a + 4 * q);
        }
    }
    if (nsave != nullptr && ok) st4(nsave + node * H + 16 * wave + 4 * q, n[wave]);
    // ---- A
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int m = 0; m < 2; ++m) u[m] = mfma16(w3f[m][a][b], n[a][b], u[m]);
#pragma unroll
    for (int m = 0; m < 2; ++m) st4(ubuf + i * LDUU + 16 * (2 * wave + m) + 4 * q, silu4(u[m]));
    __syncthreads();
    // ---- B
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const f32x4 uv = ld4(ubuf + i * LDUU + 16 * a + 4 * q);
#pragma unroll
        for (int b = 0; b < 4; ++b) xn = mfma16(w4f[a][b], uv[b], xn);
    }
    xn += n[wave];
    st4(xbuf + i * LDW + 16 * wave + 4 * q, xn);
    if (ok) st4(x_out + node * H + 16 * wave + 4 * q, xn);
    __syncthreads();
    // ---- C
    f32x4 xv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) xv[a] = ld4(xbuf + i * LDW + 16 * a + 4 * q);
    if constexpr (!LAST) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                c0v = mfma16(wcf[0][a][b], xv[a][b], c0v);
                c1v = mfma16(wcf[1][a][b], xv[a][b], c1v);
            }
        if (ok) {
            st4(Ps + node * H + 16 * wave + 4 * q, c0v);
            st4(Pr + node * H + 16 * wave + 4 * q, c1v);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) c0v = mfma16(wcf[0][a][b], xv[a][b], c0v);
        if (dropword != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *dropword = drop1 != nullptr ? 1 : 0;
        f32x4 o1v = silu4(c0v);
        if (drop1 != nullptr) o1v = o1v * ld4(drop1 + nc * H + 16 * wave + 4 * q);      // nn.Dropout (locs.py:163)
        st4(ubuf + i * LDUU + 16 * wave + 4 * q, o1v);                   // o1 (u is dead: all waves passed B)
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 ov = ld4(ubuf + i * LDUU + 16 * a + 4 * q);
#pragma unroll
            for (int b = 0; b < 4; ++b) c1v = mfma16(wcf[1][a][b], ov[b], c1v);
        }
        f32x4 o2v = silu4(c1v);
        if (drop2 != nullptr) o2v = o2v * ld4(drop2 + nc * H + 16 * wave + 4 * q);      // (locs.py:166)
        st4(ubuf + i * LDUU + H + 16 * wave + 4 * q, o2v);               // o2
        __syncthreads();
        if (wave == 0) {
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 ov = ld4(ubuf + i * LDUU + H + 16 * a + 4 * q);
#pragma unroll
                for (int b = 0; b < 4; ++b) y = mfma16(w6f[a][b], ov[b], y);
            }
            // rows 0..D-1 of y sit in lanes q == 0, registers 0..D-1, for node (lane & 15)
            if (q == 0 && ok) {
                float yl[D];
#pragma unroll
                for (int d = 0; d < D; ++d) yl[d] = y[d] + P.out_b6[d];
                const float* ni = nodeinfo + node * NI::STRIDE;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int b = 0; b < D; ++b) s += ni[NI::R + a * D + b] * yl[b];   // R y
                    const float xnew = pos[node * D + a] + s;
                    out[node * D + a] = xnew;
                    if (vel_out) vel_out[node * D + a] = (xnew - pos[node * D + a]) / dt;
                }
            }
        }
    }
}


}  // namespace
