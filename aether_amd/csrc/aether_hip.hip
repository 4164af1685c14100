// aether_hip.hip -- gfx950 (CDNA4) kernels + C ABI for the Aether state2state step.
//
// What is computed follows the reference (cited per kernel); how it is computed is
// MI355X-first:
//   * every activation tile lives in registers in "MFMA accumulator layout"
//     T[hidden = 16*mb + 4*q + r][item = lane & 15]   (q = lane >> 4, r = register 0..3)
//     which is at once the C/D layout of v_mfma_f32_16x16x4_f32 and (read k = 16a+4q+b)
//     the B-operand layout of the next MFMA, so Linear -> SiLU -> Linear chains never
//     leave the register file (no LDS transpose, no shuffles);
//   * weights are the MFMA A operand: staged once per workgroup in LDS with a padded
//     row (edge kernels, reused by every tile) or read in fragment shape from L2
//     (node kernels, used once per wave);
//   * edges are processed in receiver-sorted order (one stable sort per graph, see
//     aether_graph_build), so the mean over in-edges is a contiguous, deterministic
//     segmented sum -- no float atomics;
//   * layers 2-4 use W1 [x_s | x_r | e] = W_s x_s + W_r x_r + W_e e: the two node terms
//     are computed once per node (P_s, P_r) and gathered as the accumulator's initial
//     value, halving the per-edge MFMA work and never materialising the [E,192] concat
//     (reference: nn/state2state/locs/locs.py:233).
//
// fp32 throughout (v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain).

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/aether_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = AETHER_HIDDEN;     // 64
constexpr int LDW = H + 4;           // padded LDS row (floats) for K = 64 weights
constexpr int FPAD = 32;             // layer-1 feature count padded to two 16-wide k blocks
constexpr int LDF = FPAD + 4;        // padded LDS row for K = 32
constexpr float PI_F = 3.14159274101257324f;       // float(np.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;   // float(2*np.pi)
constexpr float EPS_F = 1e-7f;                     // nn/utils/geometry.py:62

thread_local char g_err[256] = "";

int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

#define HIP_OK(expr)                                                              \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            snprintf(g_err, sizeof(g_err), "%s: %s", #expr, hipGetErrorString(e__)); \
            return AETHER_EHIP;                                                   \
        }                                                                         \
    } while (0)

// ------------------------------------------------------------------ per-kernel HIP-event timing
// Optional (aether_profile_enable): brackets every launch with a pair of events on the
// launch stream so bench.py can report the dominant kernel's average duration.
enum KernelId { K_NODE_PREP = 0, K_EDGE_L1, K_NODE_UPDATE, K_EDGE_LN, K_NODE_LAST, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_node_prep", "k_edge_layer1", "k_node_update",
                                           "k_edge_layer", "k_node_update_last"};
struct ProfSlot { hipEvent_t a, b; int id; };
constexpr int PROF_SLOTS = 8192;
ProfSlot g_prof[PROF_SLOTS];
int g_prof_created = 0, g_prof_used = 0;
bool g_prof_on = false;

struct ProfScope {
    hipStream_t st; int slot = -1;
    ProfScope(int id, hipStream_t s) : st(s) {
        if (!g_prof_on || g_prof_used >= PROF_SLOTS) return;
        if (g_prof_used >= g_prof_created) {
            if (hipEventCreate(&g_prof[g_prof_created].a) != hipSuccess ||
                hipEventCreate(&g_prof[g_prof_created].b) != hipSuccess) return;
            ++g_prof_created;
        }
        slot = g_prof_used++;
        g_prof[slot].id = id;
        (void)hipEventRecord(g_prof[slot].a, st);
    }
    ~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st); }
};

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float silu(float x) {      // torch.nn.SiLU: x * sigmoid(x)
    return x / (1.0f + expf(-x));
}
__device__ __forceinline__ f32x4 silu4(f32x4 v) {
    f32x4 o;
    o[0] = silu(v[0]); o[1] = silu(v[1]); o[2] = silu(v[2]); o[3] = silu(v[3]);
    return o;
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// acc[mb] += W[16mb + i][k] * act[item][k], k = 16a + 4q + b, W rows at stride ldw floats.
// W may be LDS or global; both are read as one 16-byte fragment per (mb, a).
template <int MB, int KB>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ w, int ldw,
                                          const f32x4 (&bop)[KB], f32x4 (&acc)[MB], int i, int q) {
#pragma unroll
    for (int a = 0; a < KB; ++a) {
        f32x4 wv[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) wv[mb] = ld4(w + (16 * mb + i) * ldw + 16 * a + 4 * q);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(wv[mb][b], bop[a][b], acc[mb]);
        }
    }
}

// Cooperative copy of W[rows][cols] (global, row stride src_ld) into LDS [rows][ldw], zero padded.
__device__ __forceinline__ void stage_weight(float* lds, const float* __restrict__ w, int rows,
                                             int cols, int src_ld, int ldw) {
    for (int idx = threadIdx.x; idx < rows * ldw; idx += blockDim.x) {
        int r = idx / ldw, c = idx - r * ldw;
        lds[idx] = (c < cols) ? w[(size_t)r * src_ld + c] : 0.0f;
    }
}

template <int D> struct NodeInfo {
    // [p(D) v(D) f(D) R(D*D row-major) cv(D) cf(D)], padded to a multiple of 4 floats
    static constexpr int P = 0, V = D, F = 2 * D, R = 3 * D, CV = 3 * D + D * D, CF = CV + D;
    static constexpr int STRIDE = (D == 2) ? 16 : 24;
};

// ------------------------------------------------------------------ K0: per-node prep
// field net (aether.py:108-134), frame R from velocity (geometry.py:7-73),
// rel_feat = [0 | R^T v | R^T f] (aether.py:33-50), and x0 = layer_1.res(rel_feat)
// (locs.py:214-218,240).  One thread per node; the weights are wave-uniform (scalar loads).
template <int D>
__global__ void __launch_bounds__(256)
k_node_prep(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel,
            const float* __restrict__ charges, float* __restrict__ nodeinfo,
            float* __restrict__ x0, int64_t n_nodes) {
    using NI = NodeInfo<D>;
    constexpr int FIN = 2 * D + 16;
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_nodes) return;
    float z[FIN];
#pragma unroll
    for (int d = 0; d < D; ++d) { z[d] = x[n * D + d]; z[D + d] = vel[n * D + d]; }
    long ci = (long)(charges[n] + 1.0f);                      // aether.py:122-124 (truncation)
    ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[2 * D + k] = P.field_emb[ci * 16 + k];
    float h1[32], h2[32];
#pragma unroll 4
    for (int o = 0; o < 32; ++o) {
        float s = P.field_b0[o];
#pragma unroll
        for (int k = 0; k < FIN; ++k) s += P.field_w0[o * FIN + k] * z[k];
        h1[o] = silu(s);
    }
#pragma unroll 4
    for (int o = 0; o < 32; ++o) {
        float s = P.field_b2[o];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += P.field_w2[o * 32 + k] * h1[k];
        h2[o] = silu(s);
    }
    float f[D], v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float s = P.field_b4[d];
#pragma unroll
        for (int k = 0; k < 32; ++k) s += P.field_w4[d * 32 + k] * h2[k];
        f[d] = s;
        v[d] = z[D + d];
    }
    // frame (geometry.py:47-73): theta in [0, 2pi), phi = acos(clamp(vz / (|v| + eps)))
    float R[D][D];
    float theta = atan2f(v[1], v[0]);
    if (theta < 0.0f) theta += TWO_PI_F;
    float c = cosf(theta), s = sinf(theta);
    if constexpr (D == 2) {
        R[0][0] = c; R[0][1] = -s; R[1][0] = s; R[1][1] = c;
    } else {
        float rho = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        float cz = v[2] / (rho + EPS_F);
        cz = fminf(fmaxf(cz, -1.0f), 1.0f);
        float phi = acosf(cz);
        float cp = cosf(phi), sp = sinf(phi);
        R[0][0] = cp * c; R[0][1] = -s;  R[0][2] = sp * c;
        R[1][0] = cp * s; R[1][1] = c;   R[1][2] = sp * s;
        R[2][0] = -sp;    R[2][1] = 0.f; R[2][2] = cp;
    }
    float cv[D], cf[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {                             // R^T v, R^T f
        float sv = 0.f, sf = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) { sv += R[b][a] * v[b]; sf += R[b][a] * f[b]; }
        cv[a] = sv; cf[a] = sf;
    }
    float* ni = nodeinfo + n * NI::STRIDE;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ni[NI::P + d] = z[d]; ni[NI::V + d] = v[d]; ni[NI::F + d] = f[d];
        ni[NI::CV + d] = cv[d]; ni[NI::CF + d] = cf[d];
#pragma unroll
        for (int e = 0; e < D; ++e) ni[NI::R + d * D + e] = R[d][e];
    }
    // x0 = W_res [0 | cv | cf] + b_res
    float* xo = x0 + n * H;
#pragma unroll 4
    for (int o = 0; o < H; ++o) {
        float acc = P.l1_res_b[o];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc += P.l1_res_w[o * 3 * D + D + d] * cv[d];
            acc += P.l1_res_w[o * 3 * D + 2 * D + d] * cf[d];
        }
        xo[o] = acc;
    }
}

// ------------------------------------------------------------------ K1: layer-1 edge kernel
// Phase A (one thread per edge): local-frame edge features, aether.py:52-100 +
// geometry.py:76-101, followed by [rel_feat[recv] | edge_attr_orig] (aether.py:99,177).
// Phase B (one wave per 16-edge tile): e1 = SiLU(W2 SiLU(W1 a + b1) + b2), locs.py:206-212.
template <int D>
__device__ __forceinline__ void edge_features(const float* __restrict__ nj,
                                              const float* __restrict__ nir,
                                              const float* __restrict__ ea, float* __restrict__ o) {
    using NI = NodeInfo<D>;
    constexpr int O = D * (D - 1) / 2;
    float rel[D], rrel[D], rv[D], rf[D];
#pragma unroll
    for (int d = 0; d < D; ++d) rel[d] = nj[NI::P + d] - nir[NI::P + d];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) {
            float rba = nir[NI::R + b * D + a];               // (R_i^T)[a][b]
            s0 += rba * rel[b];
            s1 += rba * nj[NI::V + b];
            s2 += rba * nj[NI::F + b];
        }
        rrel[a] = s0; rv[a] = s1; rf[a] = s2;
    }
    auto M = [&](int a, int c) {                              // (R_i^T R_j)[a][c]
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += nir[NI::R + b * D + a] * nj[NI::R + b * D + c];
        return s;
    };
    int k = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = rrel[d];
    if constexpr (D == 2) {
        o[k++] = atan2f(M(1, 0), M(0, 0)) / PI_F;
    } else {
        o[k++] = atan2f(M(1, 0), M(0, 0)) / PI_F;
        o[k++] = asinf(-M(2, 0)) / PI_F;                      // no clamp (geometry.py:93)
        o[k++] = atan2f(M(2, 1), M(2, 2)) / PI_F;
    }
    float d2 = 0.f, r2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { d2 += rel[d] * rel[d]; r2 += rrel[d] * rrel[d]; }
    o[k++] = sqrtf(d2);                                       // |x_j - x_i| (aether.py:71)
    o[k++] = atan2f(rrel[1], rrel[0]);                        // symmetric theta, not normalised
    if constexpr (D == 3) {
        float cz = rrel[2] / (sqrtf(r2) + EPS_F);
        o[k++] = acosf(fminf(fmaxf(cz, -1.0f), 1.0f));
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = rv[d];
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = rf[d];
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = 0.0f;                // rel_feat[recv] = [0 | cv | cf]
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = nir[NI::CV + d];
#pragma unroll
    for (int d = 0; d < D; ++d) o[k++] = nir[NI::CF + d];
    o[k++] = ea[0];
    o[k++] = ea[1];
    static_assert(7 * D + O + 2 <= FPAD, "feature pad");
#pragma unroll
    for (; k < FPAD; ++k) o[k] = 0.0f;
}

template <int D>
__global__ void __launch_bounds__(256)
k_edge_layer1(AetherParams P, const float* __restrict__ nodeinfo,
              const float* __restrict__ edge_attr_orig, const int32_t* __restrict__ perm,
              const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
              float* __restrict__ e_out, float* __restrict__ feat_dbg, int64_t n_edges) {
    using NI = NodeInfo<D>;
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w1 = smem;                       // [64][LDF]
    float* w2 = w1 + H * LDF;               // [64][LDW]
    float* bias = w2 + H * LDW;             // [128]: b1 | b2
    float* feat = bias + 2 * H;             // [256][LDF]
    stage_weight(w1, P.l1_msg_w0, H, F1, F1, LDF);
    stage_weight(w2, P.l1_msg_w2, H, H, H, LDW);
    if (threadIdx.x < H) {
        bias[threadIdx.x] = P.l1_msg_b0[threadIdx.x];
        bias[H + threadIdx.x] = P.l1_msg_b2[threadIdx.x];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t n_chunks = (n_edges + 255) / 256;
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        __syncthreads();                    // weights staged / previous chunk's feat consumed
        {
            int64_t k = chunk * 256 + threadIdx.x;
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
                const float* ea = edge_attr_orig + 2 * (int64_t)perm[k];
                float eal[2] = {ea[0], ea[1]};
                edge_features<D>(njl, nrl, eal, o);
                if (feat_dbg) {
#pragma unroll
                    for (int t = 0; t < FPAD; ++t) feat_dbg[k * FPAD + t] = o[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < FPAD; ++t) o[t] = 0.0f;
            }
            float* fr = feat + threadIdx.x * LDF;
#pragma unroll
            for (int t = 0; t < FPAD; t += 4) st4(fr + t, f32x4{o[t], o[t + 1], o[t + 2], o[t + 3]});
        }
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const int local = wave * 64 + t * 16 + i;
            const int64_t k = chunk * 256 + local;
            if (chunk * 256 + wave * 64 + t * 16 >= n_edges) break;     // wave-uniform
            f32x4 bop[2];
            bop[0] = ld4(feat + local * LDF + 4 * q);
            bop[1] = ld4(feat + local * LDF + 16 + 4 * q);
            f32x4 acc[4], acc2[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                acc[mb] = ld4(bias + 16 * mb + 4 * q);
                acc2[mb] = ld4(bias + H + 16 * mb + 4 * q);
            }
            gemm_tile<4, 2>(w1, LDF, bop, acc, i, q);
            f32x4 h1[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) h1[mb] = silu4(acc[mb]);
            gemm_tile<4, 4>(w2, LDW, h1, acc2, i, q);
            if (k < n_edges) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) st4(e_out + k * H + 16 * mb + 4 * q, silu4(acc2[mb]));
            }
        }
    }
}

// ------------------------------------------------------------------ K3: layers 2-4 edge kernel
// e_l = SiLU(W2 SiLU(W_s x_s + W_r x_r + b1 + W_e e_{l-1}) + b2), locs.py:227-235 with the
// node terms P_s = W_s x, P_r = W_r x + b1 gathered as the accumulator's initial value.
__global__ void __launch_bounds__(256)
k_edge_layer(const float* __restrict__ w_msg0, const float* __restrict__ w_msg2,
             const float* __restrict__ b_msg2, const float* __restrict__ Ps,
             const float* __restrict__ Pr, const float* __restrict__ e_prev,
             const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
             float* __restrict__ e_out, int64_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* we = smem;                       // [64][LDW]  = W1[:, 128:192]
    float* w2 = we + H * LDW;               // [64][LDW]
    float* bias = w2 + H * LDW;             // [64] b2
    stage_weight(we, w_msg0 + 2 * H, H, H, 3 * H, LDW);
    stage_weight(w2, w_msg2, H, H, H, LDW);
    if (threadIdx.x < H) bias[threadIdx.x] = b_msg2[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int64_t n_tiles = (n_edges + 15) / 16;
    const int64_t stride = (int64_t)gridDim.x * 4;
    f32x4 b2v[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) b2v[mb] = ld4(bias + 16 * mb + 4 * q);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += stride) {
        const int64_t k = tile * 16 + i;
        const int64_t kc = k < n_edges ? k : n_edges - 1;
        const int64_t s = send_s[kc], r = recv_s[kc];
        f32x4 acc[4], bop[4], acc2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            acc[mb] = ld4(Ps + s * H + 16 * mb + 4 * q) + ld4(Pr + r * H + 16 * mb + 4 * q);
            bop[mb] = ld4(e_prev + kc * H + 16 * mb + 4 * q);
            acc2[mb] = b2v[mb];
        }
        gemm_tile<4, 4>(we, LDW, bop, acc, i, q);
        f32x4 h1[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) h1[mb] = silu4(acc[mb]);
        gemm_tile<4, 4>(w2, LDW, h1, acc2, i, q);
        if (k < n_edges) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) st4(e_out + k * H + 16 * mb + 4 * q, silu4(acc2[mb]));
        }
    }
}

// ------------------------------------------------------------------ K2: node update kernel
// n = x_prev + mean_{j->i} e (locs.py:236-240); x = n + W4 SiLU(W3 n + b3) + b4 (:241);
// then either the next layer's node terms P_s, P_r, or (LAST) the out MLP (locs.py:160-168),
// globalise (local_to_global.py:12-13) and the residual x + pred (aether.py:185).
// One wave per 16-node tile; weights are read from L2 in fragment shape (used once per wave).
template <int D, bool LAST>
__global__ void __launch_bounds__(64)
k_node_update(AetherParams P, int layer /*1..4*/, const float* __restrict__ x_prev,
              const float* __restrict__ e, const int32_t* __restrict__ rowptr,
              float* __restrict__ x_out, float* __restrict__ Ps, float* __restrict__ Pr,
              const float* __restrict__ nodeinfo, const float* __restrict__ pos,
              float* __restrict__ out, int64_t n_nodes) {
    using NI = NodeInfo<D>;
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4;
    const int64_t node = (int64_t)blockIdx.x * 16 + i;
    const int64_t nc = node < n_nodes ? node : n_nodes - 1;
    const float* w3 = layer == 1 ? P.l1_upd_w0 : P.ln_upd_w0[layer - 2];
    const float* b3 = layer == 1 ? P.l1_upd_b0 : P.ln_upd_b0[layer - 2];
    const float* w4 = layer == 1 ? P.l1_upd_w2 : P.ln_upd_w2[layer - 2];
    const float* b4 = layer == 1 ? P.l1_upd_b2 : P.ln_upd_b2[layer - 2];
    // segmented sum over the node's contiguous run of receiver-sorted edges, in edge order
    const int beg = rowptr[nc], end = rowptr[nc + 1];
    f32x4 n[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) n[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = beg; k < end; ++k) {
        const float* er = e + (int64_t)k * H + 4 * q;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) n[mb] += ld4(er + 16 * mb);
    }
    const float deg = (float)(end - beg > 1 ? end - beg : 1);    // count clamped to >= 1
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) n[mb] = ld4(x_prev + nc * H + 16 * mb + 4 * q) + n[mb] / deg;
    f32x4 u[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) u[mb] = ld4(b3 + 16 * mb + 4 * q);
    gemm_tile<8, 4>(w3, H, n, u, i, q);
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) u[mb] = silu4(u[mb]);
    f32x4 xn[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) xn[mb] = ld4(b4 + 16 * mb + 4 * q);
    gemm_tile<4, 8>(w4, 2 * H, u, xn, i, q);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) xn[mb] += n[mb];
    if (node < n_nodes) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) st4(x_out + node * H + 16 * mb + 4 * q, xn[mb]);
    }
    if constexpr (!LAST) {
        const float* w1n = P.ln_msg_w0[layer - 1];            // next layer's W1 [64][192]
        const float* b1n = P.ln_msg_b0[layer - 1];
        f32x4 ps[4], pr[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            ps[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            pr[mb] = ld4(b1n + 16 * mb + 4 * q);
        }
        gemm_tile<4, 4>(w1n, 3 * H, xn, ps, i, q);
        gemm_tile<4, 4>(w1n + H, 3 * H, xn, pr, i, q);
        if (node < n_nodes) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                st4(Ps + node * H + 16 * mb + 4 * q, ps[mb]);
                st4(Pr + node * H + 16 * mb + 4 * q, pr[mb]);
            }
        }
    } else {
        f32x4 o1[4], o2[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) o1[mb] = ld4(P.out_b0 + 16 * mb + 4 * q);
        gemm_tile<4, 4>(P.out_w0, H, xn, o1, i, q);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { o1[mb] = silu4(o1[mb]); o2[mb] = ld4(P.out_b3 + 16 * mb + 4 * q); }
        gemm_tile<4, 4>(P.out_w3, H, o1, o2, i, q);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) o2[mb] = silu4(o2[mb]);
        // last Linear has D (2|3) output rows: rows >= D of the 16-row block read row D-1
        // (in bounds) and are discarded.
        const int row = i < D ? i : D - 1;
        f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 wv = ld4(P.out_w6 + row * H + 16 * a + 4 * q);
#pragma unroll
            for (int b = 0; b < 4; ++b) y = mfma16(wv[b], o2[a][b], y);
        }
        // rows 0..D-1 of y sit in lanes q == 0, registers 0..D-1, for node (lane & 15)
        if (q == 0 && node < n_nodes) {
            float yl[D];
#pragma unroll
            for (int d = 0; d < D; ++d) yl[d] = y[d] + P.out_b6[d];
            const float* ni = nodeinfo + node * NI::STRIDE;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                float s = 0.f;
#pragma unroll
                for (int b = 0; b < D; ++b) s += ni[NI::R + a * D + b] * yl[b];   // R y
                out[node * D + a] = pos[node * D + a] + s;
            }
        }
    }
}

// ------------------------------------------------------------------ graph build kernels
__global__ void k_graph_keys(const int64_t* __restrict__ send, const int64_t* __restrict__ recv,
                             int64_t n_edges, int64_t n_nodes, int32_t* __restrict__ keys,
                             int32_t* __restrict__ vals, int32_t* __restrict__ bad) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    int64_t s = send[k], r = recv[k];
    if (s < 0 || s >= n_nodes || r < 0 || r >= n_nodes) { atomicOr(bad, 1); r = 0; }
    keys[k] = (int32_t)r;
    vals[k] = (int32_t)k;
}

__global__ void k_graph_finish(const int64_t* __restrict__ send, const int32_t* __restrict__ recv_s,
                               const int32_t* __restrict__ perm, int64_t n_edges, int64_t n_nodes,
                               int32_t* __restrict__ send_s, int32_t* __restrict__ rowptr) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    int64_t s = send[perm[k]];
    send_s[k] = (int32_t)((s < 0 || s >= n_nodes) ? 0 : s);
    int32_t r = recv_s[k];
    int32_t rp = k == 0 ? -1 : recv_s[k - 1];
    for (int32_t n = rp + 1; n <= r; ++n) rowptr[n] = (int32_t)k;      // also covers empty rows
    if (k == n_edges - 1)
        for (int64_t n = r + 1; n <= n_nodes; ++n) rowptr[n] = (int32_t)n_edges;
}

// ------------------------------------------------------------------ host-side layouts
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GraphLayout {
    size_t perm, send_s, recv_s, rowptr, keys, vals, flag, cub, total, cub_bytes;
    GraphLayout(int64_t E, int64_t Nn, bool with_sort_scratch = true) {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
        size_t e4 = (size_t)(E > 0 ? E : 1) * 4;
        perm = take(e4); send_s = take(e4); recv_s = take(e4);
        rowptr = take((size_t)(Nn + 1) * 4);
        keys = take(e4); vals = take(e4); flag = take(256);
        cub_bytes = 0;
        if (with_sort_scratch)      // size query only: no launch, no device access
            (void)hipcub::DeviceRadixSort::SortPairs<int32_t, int32_t>(nullptr, cub_bytes, nullptr, nullptr,
                                                               nullptr, nullptr, (int)E, 0, 32, nullptr);
        cub = take(cub_bytes + 256);
        total = off;
    }
};

struct WsLayout {
    size_t nodeinfo, x[5], ps[3], pr[3], e[4], feat, total;
    WsLayout(int64_t Nn, int64_t E, int D, bool debug_feat) {
        size_t off = 0;
        auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
        nodeinfo = take((size_t)Nn * (D == 2 ? 16 : 24));
        for (auto& v : x) v = take((size_t)Nn * H);
        for (auto& v : ps) v = take((size_t)Nn * H);
        for (auto& v : pr) v = take((size_t)Nn * H);
        for (auto& v : e) v = take((size_t)E * H);
        feat = debug_feat ? take((size_t)E * FPAD) : 0;
        total = off;
    }
};

inline int bits_for(int64_t n) { int b = 1; while (((int64_t)1 << b) < n) ++b; return b; }

template <int D>
int forward_impl(const AetherParams& P, int64_t Nn, int64_t E, const float* x, const float* vel,
                 const float* charges, const float* ea, const char* graph, char* ws, float* out,
                 hipStream_t st) {
    GraphLayout G(E, Nn, false);
    WsLayout W(Nn, E, D, false);
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(graph + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int32_t *perm = gp(G.perm), *send_s = gp(G.send_s), *recv_s = gp(G.recv_s), *rowptr = gp(G.rowptr);
    float* nodeinfo = wp(W.nodeinfo);

    {
        ProfScope ps(K_NODE_PREP, st);
        k_node_prep<D><<<dim3((unsigned)((Nn + 255) / 256)), dim3(256), 0, st>>>(P, x, vel, charges,
                                                                              nodeinfo, wp(W.x[0]), Nn);
    }
    const int64_t n_chunks = (E + 255) / 256;
    const int64_t n_tiles = (E + 15) / 16;
    const unsigned node_grid = (unsigned)((Nn + 15) / 16);
    if (E > 0) {
        const size_t lds1 = (size_t)(H * LDF + H * LDW + 2 * H + 256 * LDF) * 4;
        unsigned g1 = (unsigned)(n_chunks < 2048 ? n_chunks : 2048);
        ProfScope ps(K_EDGE_L1, st);
        k_edge_layer1<D><<<dim3(g1), dim3(256), lds1, st>>>(P, nodeinfo, ea, perm, send_s, recv_s,
                                                           wp(W.e[0]), nullptr, E);
    }
    {
        ProfScope ps(K_NODE_UPDATE, st);
        k_node_update<D, false><<<dim3(node_grid), dim3(64), 0, st>>>(
        P, 1, wp(W.x[0]), wp(W.e[0]), rowptr, wp(W.x[1]), wp(W.ps[0]), wp(W.pr[0]), nodeinfo, x, out, Nn);
    }
    for (int l = 2; l <= 4; ++l) {
        if (E > 0) {
            const size_t lds = (size_t)(2 * H * LDW + H) * 4;
            int64_t wgs = (n_tiles + 3) / 4;
            unsigned g = (unsigned)(wgs < 1024 ? wgs : 1024);
            ProfScope ps(K_EDGE_LN, st);
            k_edge_layer<<<dim3(g), dim3(256), lds, st>>>(P.ln_msg_w0[l - 2], P.ln_msg_w2[l - 2],
                                                         P.ln_msg_b2[l - 2], wp(W.ps[l - 2]),
                                                         wp(W.pr[l - 2]), wp(W.e[l - 2]), send_s, recv_s,
                                                         wp(W.e[l - 1]), E);
        }
        if (l < 4) {
            ProfScope ps(K_NODE_UPDATE, st);
            k_node_update<D, false><<<dim3(node_grid), dim3(64), 0, st>>>(
                P, l, wp(W.x[l - 1]), wp(W.e[l - 1]), rowptr, wp(W.x[l]), wp(W.ps[l - 1]), wp(W.pr[l - 1]),
                nodeinfo, x, out, Nn);
        } else {
            ProfScope ps(K_NODE_LAST, st);
            k_node_update<D, true><<<dim3(node_grid), dim3(64), 0, st>>>(
                P, l, wp(W.x[l - 1]), wp(W.e[l - 1]), rowptr, wp(W.x[l]), nullptr, nullptr, nodeinfo, x,
                out, Nn);
        }
    }
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

const char* aether_version(void) { return "aether_hip 0.1 (gfx950, fp32 MFMA 16x16x4)"; }
const char* aether_last_error(void) { return g_err; }

int aether_profile_enable(int on) {
    g_prof_on = on != 0;
    g_prof_used = 0;
    return AETHER_OK;
}

int aether_profile_kernels(void) { return K_COUNT; }

const char* aether_profile_kernel_name(int id) {
    return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : "";
}

int aether_profile_read(double* total_ms, int64_t* launches, int n) {
    if (!total_ms || !launches || n < K_COUNT) return fail(AETHER_EINVAL, "profile_read: need K_COUNT slots");
    for (int k = 0; k < n; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
    for (int s = 0; s < g_prof_used; ++s) {
        HIP_OK(hipEventSynchronize(g_prof[s].b));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, g_prof[s].a, g_prof[s].b));
        total_ms[g_prof[s].id] += ms;
        launches[g_prof[s].id] += 1;
    }
    g_prof_used = 0;
    return AETHER_OK;
}

size_t aether_graph_bytes(int64_t n_edges, int64_t n_nodes) {
    if (n_edges < 0 || n_nodes <= 0) return 0;
    return GraphLayout(n_edges, n_nodes).total;
}

int aether_graph_build(const int64_t* send, const int64_t* recv, int64_t n_edges, int64_t n_nodes,
                       void* graph, size_t graph_bytes, void* stream) {
    if (n_nodes <= 0 || n_edges < 0 || !graph) return fail(AETHER_EINVAL, "graph_build: bad sizes");
    if (n_edges >= ((int64_t)1 << 31) || n_nodes >= ((int64_t)1 << 31))
        return fail(AETHER_EINVAL, "graph_build: more than 2^31 edges or nodes");
    if (n_edges > 0 && (!send || !recv)) return fail(AETHER_EINVAL, "graph_build: null edge index");
    GraphLayout G(n_edges, n_nodes);
    if (graph_bytes < G.total) return fail(AETHER_ESPACE, "graph_build: graph buffer too small");
    hipStream_t st = (hipStream_t)stream;
    char* g = (char*)graph;
    int32_t* rowptr = (int32_t*)(g + G.rowptr);
    int32_t* flag = (int32_t*)(g + G.flag);
    HIP_OK(hipMemsetAsync(flag, 0, 4, st));
    if (n_edges == 0) {
        HIP_OK(hipMemsetAsync(rowptr, 0, (size_t)(n_nodes + 1) * 4, st));
        HIP_OK(hipStreamSynchronize(st));
        return AETHER_OK;
    }
    unsigned blocks = (unsigned)((n_edges + 255) / 256);
    int32_t *keys = (int32_t*)(g + G.keys), *vals = (int32_t*)(g + G.vals);
    int32_t *recv_s = (int32_t*)(g + G.recv_s), *perm = (int32_t*)(g + G.perm);
    k_graph_keys<<<dim3(blocks), dim3(256), 0, st>>>(send, recv, n_edges, n_nodes, keys, vals, flag);
    size_t cub_bytes = G.cub_bytes;
    // stable LSD radix sort: edges with equal receiver keep ascending edge id
    HIP_OK(hipcub::DeviceRadixSort::SortPairs(g + G.cub, cub_bytes, keys, recv_s, vals, perm,
                                              (int)n_edges, 0, bits_for(n_nodes), st));
    k_graph_finish<<<dim3(blocks), dim3(256), 0, st>>>(send, recv_s, perm, n_edges, n_nodes,
                                                      (int32_t*)(g + G.send_s), rowptr);
    HIP_OK(hipGetLastError());
    int32_t bad = 0;
    HIP_OK(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (bad) return fail(AETHER_EINDEX, "graph_build: edge index outside [0, n_nodes)");
    return AETHER_OK;
}

int aether_graph_perm(const void* graph, int64_t n_edges, int64_t n_nodes, int32_t* perm_out,
                      void* stream) {
    if (!graph || !perm_out) return fail(AETHER_EINVAL, "graph_perm: null pointer");
    GraphLayout G(n_edges, n_nodes, false);
    HIP_OK(hipMemcpyAsync(perm_out, (const char*)graph + G.perm, (size_t)n_edges * 4,
                          hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AETHER_OK;
}

size_t aether_workspace_bytes(int64_t n_nodes, int64_t n_edges, int num_dims, int keep_for_backward) {
    (void)keep_for_backward;
    if (n_nodes <= 0 || n_edges < 0 || (num_dims != 2 && num_dims != 3)) return 0;
    return WsLayout(n_nodes, n_edges, num_dims, false).total;
}

int aether_forward(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                   const float* x, const float* vel, const float* charges,
                   const float* edge_attr_orig, const void* graph, void* workspace,
                   size_t workspace_bytes, float* out, void* stream) {
    if (!params || !x || !vel || !charges || !graph || !workspace || !out)
        return fail(AETHER_EINVAL, "forward: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "forward: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0) return fail(AETHER_EINVAL, "forward: bad sizes");
    if (n_edges > 0 && !edge_attr_orig) return fail(AETHER_EINVAL, "forward: null edge_attr");
    if (workspace_bytes < aether_workspace_bytes(n_nodes, n_edges, num_dims, 0))
        return fail(AETHER_ESPACE, "forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (num_dims == 2)
        return forward_impl<2>(*params, n_nodes, n_edges, x, vel, charges, edge_attr_orig,
                               (const char*)graph, (char*)workspace, out, st);
    return forward_impl<3>(*params, n_nodes, n_edges, x, vel, charges, edge_attr_orig,
                           (const char*)graph, (char*)workspace, out, st);
}

int64_t aether_debug_fetch(const char* name, int num_dims, int64_t n_nodes, int64_t n_edges,
                           const void* workspace, float* dst, void* stream) {
    if (!name || !workspace || !dst) return fail(AETHER_EINVAL, "debug_fetch: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "debug_fetch: num_dims");
    const int D = num_dims, NIS = D == 2 ? 16 : 24;
    WsLayout W(n_nodes, n_edges, D, false);
    const char* ws = (const char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    auto copy2d = [&](const float* src, int64_t rows, int src_ld, int col0, int cols) -> int64_t {
        hipError_t e = hipMemcpy2DAsync(dst, (size_t)cols * 4, src + col0, (size_t)src_ld * 4,
                                        (size_t)cols * 4, (size_t)rows, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { fail(AETHER_EHIP, hipGetErrorString(e)); return AETHER_EHIP; }
        return rows * cols;
    };
    const float* ni = (const float*)(ws + W.nodeinfo);
    if (!strcmp(name, "field")) return copy2d(ni, n_nodes, NIS, 2 * D, D);
    if (!strcmp(name, "R")) return copy2d(ni, n_nodes, NIS, 3 * D, D * D);
    if (!strcmp(name, "canon")) return copy2d(ni, n_nodes, NIS, 3 * D + D * D, 2 * D);
    if (name[0] == 'x' && name[1] >= '0' && name[1] <= '4' && !name[2])
        return copy2d((const float*)(ws + W.x[name[1] - '0']), n_nodes, H, 0, H);
    if (name[0] == 'e' && name[1] >= '1' && name[1] <= '4' && !name[2])
        return copy2d((const float*)(ws + W.e[name[1] - '1']), n_edges, H, 0, H);
    return fail(AETHER_EINVAL, "debug_fetch: unknown name");
}

}  // extern "C"
