// wide.h -- hidden_size wider than 64 (the reference accepts any width: nn/state2state/aether.py:143-158,
// experiments/lorentz/main.py:42-43 `--nf`).  Included after backward.h.
//
// The 64-wide kernels (fused.h, streamed.h) keep a whole activation row in one wave's registers; that does not scale
// with the width.  For hidden = 64 m (m >= 2; other widths run zero-padded, as the narrow ones do) the step is computed
// layer by layer on ONE generic kernel:
//
//   k_wgemm   C[M][N] = epilogue(A[M][K] . B[N][K]^T)       nn.Linear form (both operands K-contiguous)
//
// with the epilogues a layer needs (bias, up to two row-gathered addends, SiLU or the product with SiLU' of a saved
// pre-activation, a dropout mask) so that no elementwise pass over an [E, hidden] tensor is left between the GEMMs.
// Arithmetic is the library's: fp32 operands split into two fp16 pieces, three MFMA terms per product, fp32 accumulation
// (common.h, gemm_split; k_wgemm below for how the A operand's range is handled).  Operands are split once per tile while
// they are staged into LDS, in the fragment order stage_split4 defines, and re-read by the four waves (2 x 2 wave tiles of
// 64 x 64).  The backward's data gradients are the same kernel on transposed weight copies (k_wide_prep), its weight
// gradients the generic outer-product kernel of backward.h (k_outer: 64 x 64 pieces, fixed-order partial sums).
// Everything that is not a GEMM (frames, edge features, segmented mean, the last D-row Linear, field net) is either the
// 64-wide path's own kernel (k_node_prep, kb_field, kb_inputs: width passed at run time) or a small kernel below.
#pragma once
#include "common.h"

namespace {

constexpr int WG_BM = 128;                    // rows of A (items) per workgroup
constexpr int WG_BN = 128;                    // rows of B (output features) per workgroup
constexpr int WG_IMG = 2 * 8 * 64 * 4;        // floats of one split tile image: 2 terms x 8 row blocks x 64 lanes x 16 B
constexpr int WG_STAGE = 2 * WG_IMG;          // A image | B image
// ONE stage in LDS (32 KB + a few words), the next k block waits in registers: two workgroups per CU (two waves per SIMD at
// <= 256 registers) instead of one -- with a double-buffered stage a workgroup had the CU to itself and, at K = 128 .. 256
// (4 - 8 k blocks), spent most of its time in its own load latencies (HISTORY R4).
constexpr size_t WG_LDS_BYTES = (size_t)WG_STAGE * 4 + 16;

struct WGemmArgs {
    const float* A; const float* B;           // [M][lda], [N][ldb]; K columns each, K % 32 == 0, rows 16-byte aligned
    const float* bias;                        // [N] or null
    const float* add1; const int32_t* idx1;   // + add1[idx1 ? idx1[row] : row][ldadd] (null: none)
    const float* add2; const int32_t* idx2;
    const float* ds;                          // act == 2: saved pre-activation [M][ldc]
    const float* mask; const int* maskword;   // scale mask [M][ldc] (null: none); maskword != null and *maskword == 0: ignored
    float* pre;                               // value before the activation [M][ldc] (null: not stored)
    float* out;                               // [M][ldc]
    int64_t M;
    int N, K, lda, ldb, ldc, ldadd;           // N % 16 == 0
    int act;                                  // 0: none, 1: SiLU, 2: times SiLU'(ds)
};

__device__ __forceinline__ f32x4 wide_dsilu4(f32x4 z) {
    const f32x4 s = sigmoid4(z);
    return dsilu_from_sigmoid(z, s);
}

// Round 4 (second half): two fp16 pieces per operand, three MFMA terms (common.h; was three bf16 pieces, six terms).  The B
// operand (weights, |w| < 65,504) is split as it is.  The A operand (activations forward, gradients backward: any magnitude)
// carries ONE power-of-two scale per workgroup that only ever shrinks along K: every thread leaves the maximum of the values
// it fetched for the next k block in an LDS word (ds_max on the bit patterns of non-negative floats) in front of the barrier
// the loop takes anyway; behind it everybody derives the same scale -- the first k block sets it (maximum -> 2^13 .. 2^14), a
// later block that would reach 2^15 lowers it and every wave multiplies its accumulators by the ratio (exact).  The epilogue
// divides by the final scale.
__global__ void __launch_bounds__(256, 2)
k_wgemm(const WGemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* smax = reinterpret_cast<unsigned*>(smem + WG_STAGE);    // [2]: maxima of the k blocks, alternating
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int wi = wave & 1, wf = wave >> 1;                      // the wave's 64 x 64 quadrant: items, features
    const int64_t m0 = (int64_t)blockIdx.x * WG_BM;
    const int n0 = (int)blockIdx.y * WG_BN;
    const int nkb = G.K >> 5;
    // staging: float4 number f = tid + 256 j -> (tile row f >> 3, columns 4 (f & 7) ..): 8 threads cover a row's 128 bytes
    f32x4 va[4], vb[4];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j, r = f >> 3, c = (f & 7) * 4;
            const int64_t row = m0 + r;
            const int n = n0 + r;
            va[j] = row < G.M ? ld4(G.A + row * G.lda + 32 * kb + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            vb[j] = n < G.N ? ld4(G.B + (size_t)n * G.ldb + 32 * kb + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto publish_max = [&](int kb) {                               // this thread's part of |A| of block kb -> smax[kb & 1]
        float m = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(va[j][0]), fabsf(va[j][1]))), fmaxf(fabsf(va[j][2]), fabsf(va[j][3])));
        const unsigned wm = wave_max_bits(m);                      // one LDS atomic per wave, not per lane
        if (lane == 0) atomicMax(smax + (kb & 1), wm);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[fb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
    float xs = 0.0f;                                               // the workgroup's A scale (0: not set yet)
    // scale of block kb from the published maximum (workgroup-uniform), accumulators follow; then split + stage
    auto stage = [&](float* buf, int kb) {
        const unsigned mb = smax[kb & 1];
        const unsigned E = (mb >> 23) & 255u;
        if (xs == 0.0f || __uint_as_float(mb) * xs >= 32768.0f) {
            int sh = 140 - (int)E;                                 // max (2^(E-127) ..) -> 2^13 ..
            sh = sh > 40 ? 40 : (sh < -40 ? -40 : sh);
            float ns = __int_as_float((127 + sh) << 23);
            if (xs != 0.0f) {
                ns = ns < xs ? ns : xs;                            // only ever down
                const float ratio = ns / xs;                       // (powers of two: exact)
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) acc[fb][ib] = acc[fb][ib] * ratio;
            }
            xs = ns;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j, r = f >> 3, c = (f & 7) * 4;
            stage_split4<8, 1>(buf, r, c, va[j] * xs);
            stage_split4<8, 1>(buf + WG_IMG, r, c, vb[j]);
        }
    };
    if (tid < 2) smax[tid] = 0u;
    fetch(0);
    lds_barrier();                                                 // smax cleared
    publish_max(0);
    lds_barrier();
    stage(smem, 0);
    lds_barrier();
    for (int kb = 0; kb < nkb; ++kb) {
        const float* buf = smem;
        if (tid == 0) smax[kb & 1] = 0u;                           // block kb's word (read by everybody before the last barrier) is block kb + 2's: two barriers before anybody publishes into it
        if (kb + 1 < nkb) fetch(kb + 1);                           // in flight under this block's MFMAs
        const f16x8* xa = reinterpret_cast<const f16x8*>(buf);
        const f16x8* wb = reinterpret_cast<const f16x8*>(buf + WG_IMG);
        f16x8 xh[4], xl[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int frag = (4 * wi + ib) * 64 + lane;
            xh[ib] = xa[frag]; xl[ib] = xa[512 + frag];
        }
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const int frag = (4 * wf + fb) * 64 + lane;
            const f16x8 wh = wb[frag], wl = wb[512 + frag];
            // small terms first; the four item blocks between two terms of one accumulator hide the MFMA latency
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[fb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[ib], acc[fb][ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[fb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[ib], acc[fb][ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[fb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ib], acc[fb][ib], 0, 0, 0);
        }
        if (kb + 1 < nkb) {
            publish_max(kb + 1);
            lds_barrier();                                         // every wave has its fragments of this block; the maximum is complete
            stage(smem, kb + 1);
            lds_barrier();
        }
    }
    const float inv_xs = xs != 0.0f ? 1.0f / xs : 1.0f;            // (a power of two: exact)
    // ---- epilogue: acc[fb][ib][r] = C[item 16 (4 wi + ib) + i][feature 16 (4 wf + fb) + 4 q + r]
    const bool use_mask = G.mask != nullptr && (G.maskword == nullptr || *G.maskword != 0);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        const int nb = n0 + 16 * (4 * wf + fb);
        if (nb >= G.N) continue;                                    // wave-uniform
        const int n = nb + 4 * q;
        const f32x4 bv = G.bias != nullptr ? ld4(G.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int64_t row = m0 + 16 * (4 * wi + ib) + i;
            if (row >= G.M) continue;
            f32x4 v = acc[fb][ib] * inv_xs + bv;
            if (G.add1 != nullptr) {
                const int64_t r1 = G.idx1 != nullptr ? (int64_t)G.idx1[row] : row;
                v += ld4(G.add1 + r1 * G.ldadd + n);
            }
            if (G.add2 != nullptr) {
                const int64_t r2 = G.idx2 != nullptr ? (int64_t)G.idx2[row] : row;
                v += ld4(G.add2 + r2 * G.ldadd + n);
            }
            if (G.pre != nullptr) st4(G.pre + row * G.ldc + n, v);
            if (G.act == 1) v = silu4(v);
            else if (G.act == 2) v = v * wide_dsilu4(ld4(G.ds + row * G.ldc + n));
            if (use_mask) v = v * ld4(G.mask + row * G.ldc + n);
            if (G.out != nullptr) st4(G.out + row * G.ldc + n, v);
        }
    }
}

// ------------------------------------------------------------------ weight copies the GEMMs read
// transpose == 0: dst[r][c] = src[r][col0 + c]; transpose == 1: dst[c][r] = src[r][col0 + c]   (r < rows, c < cols);
// dst is [dst_rows][dst_ld], zero wherever the source has nothing.
struct WPrepTask { const float* src; float* dst; int rows, cols, src_ld, col0, dst_rows, dst_ld, transpose; };
constexpr int WPREP_MAX_TASKS = 32;
struct WPrepBatch { WPrepTask t[WPREP_MAX_TASKS]; int n_tasks; };

__global__ void __launch_bounds__(256)
k_wide_prep(const WPrepBatch batch, const int n_blocks_x /* = gridDim.x, explicit: a by-value struct kernel that may sit in a
            captured training step consumes no hidden kernel arguments (tools/isa_check.py rule R4) */) {
    const WPrepTask T = batch.t[blockIdx.y];
    const int total = T.dst_rows * T.dst_ld;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += n_blocks_x * 256) {
        const int a = idx / T.dst_ld, b = idx - a * T.dst_ld;
        const int r = T.transpose ? b : a, c = T.transpose ? a : b;
        T.dst[idx] = (r < T.rows && c < T.cols) ? T.src[(size_t)r * T.src_ld + T.col0 + c] : 0.0f;
    }
}

// ------------------------------------------------------------------ layer-1 edge features, one thread per edge
// aether.py:52-100 + geometry.py:76-101, then [rel_feat[recv] | edge_attr_orig] (aether.py:99,177): the feature build of
// k_edge_layer1 (streamed.h) with the rows going to HBM, receiver-sorted order, padded to FPAD columns.
template <int D>
__global__ void __launch_bounds__(256)
k_wide_features(const float* __restrict__ nodeinfo, const float* __restrict__ edge_attr_orig,
                const int32_t* __restrict__ perm, const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
                const float* __restrict__ qattr, float* __restrict__ feat, int64_t n_edges) {
    using NI = NodeInfo<D>;
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_edges) return;
    const float* nj = nodeinfo + (int64_t)send_s[k] * NI::STRIDE;
    const float* nr = nodeinfo + (int64_t)recv_s[k] * NI::STRIDE;
    float njl[NI::STRIDE], nrl[NI::STRIDE];
#pragma unroll
    for (int t = 0; t < NI::STRIDE; t += 4) {
        const f32x4 a = ld4(nj + t), b = ld4(nr + t);
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
    float o[FPAD];
    edge_features<D>(njl, nrl, eal, o);
    float* fr = feat + k * FPAD;
#pragma unroll
    for (int t = 0; t < FPAD; t += 4) st4(fr + t, f32x4{o[t], o[t + 1], o[t + 2], o[t + 3]});
}

// x0 = layer_1.res(rel_feat), rel_feat = [0 | R^T v | R^T f] (aether.py:39-48, locs.py:214-218): 2 D products per output.
template <int D>
__global__ void __launch_bounds__(256)
k_wide_x0(const float* __restrict__ w_res /*[hid][3D]*/, const float* __restrict__ b_res,
          const float* __restrict__ nodeinfo, float* __restrict__ x0, int64_t n_nodes, int hid) {
    using NI = NodeInfo<D>;
    const int per = hid >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t node = idx / per;
    if (node >= n_nodes) return;
    const int c = (int)(idx - node * per) * 4;
    const float* ni = nodeinfo + node * NI::STRIDE;
    float rel[2 * D];
#pragma unroll
    for (int d = 0; d < D; ++d) { rel[d] = ni[NI::CV + d]; rel[D + d] = ni[NI::CF + d]; }
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float s = b_res[c + r];
#pragma unroll
        for (int k = 0; k < 2 * D; ++k) s += w_res[(size_t)(c + r) * 3 * D + D + k] * rel[k];
        v[r] = s;
    }
    st4(x0 + node * hid + c, v);
}

// n = x_prev + mean over the in-edges of e (locs.py:236-240) on the receiver-sorted messages: one wave per node, a lane
// owns four columns (per pass of 256), rows added in edge order (deterministic), four row loads in flight.
__global__ void __launch_bounds__(256)
k_wide_segmean(const float* __restrict__ e, const int32_t* __restrict__ rowptr, const float* __restrict__ x_prev,
               float* __restrict__ n_out, int64_t n_nodes, int hid) {
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    const int beg = rowptr[node], end = rowptr[node + 1];
    const float inv = 1.0f / (float)(end - beg > 1 ? end - beg : 1);
    for (int c = 4 * lane; c < hid; c += 256) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* p = e + (int64_t)beg * hid + c;
        int k = beg;
        for (; k + 4 <= end; k += 4, p += 4 * (int64_t)hid) {
            const f32x4 a0 = ld4(p), a1 = ld4(p + hid), a2 = ld4(p + 2 * (int64_t)hid), a3 = ld4(p + 3 * (int64_t)hid);
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; k < end; ++k, p += hid) s += ld4(p);
        st4(n_out + node * hid + c, ld4(x_prev + node * hid + c) + s * inv);
    }
}

// Last Linear of the out MLP (D rows, locs.py:167), Globalizer (local_to_global.py:12-13) and the residual
// (aether.py:185): one wave per node, lanes over the hidden columns, fixed xor tree.
template <int D>
__global__ void __launch_bounds__(256)
k_wide_out(const float* __restrict__ w7 /*[D][hid]*/, const float* __restrict__ b7, const float* __restrict__ o2,
           const float* __restrict__ nodeinfo, const float* __restrict__ pos, float* __restrict__ out,
           float* __restrict__ vel_out, float dt, int64_t n_nodes, int hid) {
    using NI = NodeInfo<D>;
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    float y[D];
#pragma unroll
    for (int d = 0; d < D; ++d) y[d] = 0.f;
    for (int c = lane; c < hid; c += 64) {
        const float ov = o2[node * hid + c];
#pragma unroll
        for (int d = 0; d < D; ++d) y[d] += w7[(size_t)d * hid + c] * ov;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) y[d] += __shfl_xor(y[d], o, 64);
        y[d] += b7[d];
    }
    if (lane == 0) {
        const float* ni = nodeinfo + node * NI::STRIDE;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < D; ++b) s += ni[NI::R + a * D + b] * y[b];    // R y
            const float xnew = pos[node * D + a] + s;
            out[node * D + a] = xnew;
            if (vel_out) vel_out[node * D + a] = (xnew - pos[node * D + a]) / dt;
        }
    }
}

__global__ void k_wide_set_word(int* __restrict__ w, int v) { *w = v; }

// ------------------------------------------------------------------ backward, elementwise pieces
// dy = R^T g (local_to_global.py:12-13 transposed) [Nn][16], and dL/dpre6 = (W7^T dy) * mask2 * SiLU'(pre6).
template <int D>
__global__ void __launch_bounds__(256)
k_wide_dout(const float* __restrict__ w7 /*[D][hid]*/, const float* __restrict__ nodeinfo,
            const float* __restrict__ g_out, const float* __restrict__ pre6, const float* __restrict__ mask2,
            const int* __restrict__ maskword, float* __restrict__ DY, float* __restrict__ DPO2, int64_t n_nodes, int hid) {
    using NI = NodeInfo<D>;
    const int per = hid >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t node = idx / per;
    if (node >= n_nodes) return;
    const int c = (int)(idx - node * per) * 4;
    const float* ni = nodeinfo + node * NI::STRIDE;
    float dy[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += ni[NI::R + b * D + a] * g_out[node * D + b];
        dy[a] = s;
    }
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < D; ++d) v += ld4(w7 + (size_t)d * hid + c) * own_reg(dy[d]);
    v = v * wide_dsilu4(ld4(pre6 + node * hid + c));
    if (mask2 != nullptr && *maskword != 0) v = v * ld4(mask2 + node * hid + c);
    st4(DPO2 + node * hid + c, v);
    if (c == 0) {
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t + u < D) z[u] = dy[t + u];
            st4(DY + node * 16 + t, z);
        }
    }
}

// dL/dpre2 of an edge MLP: de = dn[recv] / deg[recv] (the mean, locs.py:236-238) + the gradient through the next
// layer's edge input (DE, null for the last layer), times SiLU'(pre2).
__global__ void __launch_bounds__(256)
k_wide_edge_dp2(const float* __restrict__ DN, const float* __restrict__ DE, const float* __restrict__ pre2,
                const int32_t* __restrict__ recv_s, const int32_t* __restrict__ rowptr, float* __restrict__ DP2,
                int64_t n_edges, int hid) {
    const int per = hid >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k = idx / per;
    if (k >= n_edges) return;
    const int c = (int)(idx - k * per) * 4;
    const int r = recv_s[k];
    const int deg = rowptr[r + 1] - rowptr[r];
    const float inv = own_reg(1.0f / (float)(deg > 1 ? deg : 1));
    f32x4 de = ld4(DN + (int64_t)r * hid + c) * inv;
    if (DE != nullptr) de += ld4(DE + k * hid + c);
    st4(DP2 + k * hid + c, de * wide_dsilu4(ld4(pre2 + k * hid + c)));
}

// dP_r[i] = sum_{k: recv = i} G_k (contiguous run), dP_s[j] = sum_{k: send = j} G_k (rows listed by sperm): kb_sum_g
// (backward.h) at any width; wave per node, four columns per lane, fixed order.
__global__ void __launch_bounds__(256)
k_wide_sum_g(const float* __restrict__ G, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ srowptr,
             const int32_t* __restrict__ sperm, float* __restrict__ DPS, float* __restrict__ DPR, int64_t n_nodes,
             int hid) {
    const int lane = threadIdx.x & 63;
    const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    for (int c = 4 * lane; c < hid; c += 256) {
        {
            const int beg = rowptr[node], end = rowptr[node + 1];
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* p = G + (int64_t)beg * hid + c;
            int k = beg;
            for (; k + 4 <= end; k += 4, p += 4 * (int64_t)hid) {
                const f32x4 a0 = ld4(p), a1 = ld4(p + hid), a2 = ld4(p + 2 * (int64_t)hid), a3 = ld4(p + 3 * (int64_t)hid);
                s += a0; s += a1; s += a2; s += a3;
            }
            for (; k < end; ++k, p += hid) s += ld4(p);
            st4(DPR + node * hid + c, s);
        }
        {
            const int beg = srowptr[node], end = srowptr[node + 1];
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            int k = beg;
            for (; k + 4 <= end; k += 4) {
                const int r0 = sperm[k], r1 = sperm[k + 1], r2 = sperm[k + 2], r3 = sperm[k + 3];
                const f32x4 a0 = ld4(G + (int64_t)r0 * hid + c), a1 = ld4(G + (int64_t)r1 * hid + c),
                            a2 = ld4(G + (int64_t)r2 * hid + c), a3 = ld4(G + (int64_t)r3 * hid + c);
                s += a0; s += a1; s += a2; s += a3;
            }
            for (; k < end; ++k) s += ld4(G + (int64_t)sperm[k] * hid + c);
            st4(DPS + node * hid + c, s);
        }
    }
}

}  // namespace
