// seq2seq Aether, row A8 of SURVEY.md 8a: the field query
//   predict_field (nn/seq2seq/aether.py:86-90) = FourierFeatureMapper (nn/nn/fourier_feature_mapper.py:7-21)
//   + field_net Linear(h,h)-SiLU-Linear(h,h)-SiLU-Linear(h,D) (aether.py:72-78), h = 512.
// Dense fp32 GEMMs over all points of a batch of trajectories (M = h output rows, N = points, K = h):
// MFMA-bound (1.05 MFLOP per point).  Activations are point-major [N][h], which is at once the
// B-operand layout (k = 16a + 4q + b: one 16-byte load feeds four k-steps) and, transposed back by the
// store, the accumulator layout -- the same trick as the rest of the library, so no LDS is needed:
// a wave owns a 64 x 32 output block (4 x 2 MFMA tiles), reads W and X fragments straight from L2 two
// k-groups ahead, and issues 32 MFMAs per six 1-KiB loads.
#pragma once
#include "common.h"

namespace {

constexpr float TWO_PI_S2S = 6.28318530717958647692f;

// gamma[n][k] = sin(2 pi x_n . B[:, k]), gamma[n][half + k] = cos(..)   (fourier_feature_mapper.py:19-21)
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_rff(const float* __restrict__ x, int x_stride, const float* __restrict__ Bm, int half,
          float* __restrict__ gamma, int64_t n_points, int ld = 0 /* row stride of gamma; 0: 2 half */) {
    if (ld == 0) ld = 2 * half;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_points * half) return;
    const int64_t n = idx / half;
    const int k = (int)(idx - n * half);
    float p = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) p = fmaf(TWO_PI_S2S * x[n * x_stride + d], Bm[d * half + k], p);
    gamma[n * ld + k] = sinf(p);
    gamma[n * ld + half + k] = cosf(p);
}

// Y[n][m] = act(sum_k W[m][k] X[n][k] + b[m]); W [M][K] rows at stride ldw (nn.Linear), X [N][K], Y [N][ldy].
// K % 16 == 0.  grid = (ceil(N / (32 NT)), ceil(M / (32 MT))), 4 waves = 2 (m) x 2 (n); a wave owns
// MT x NT MFMA tiles (16 MT rows x 16 NT points) and keeps the fragments of the next PF k-groups in flight
// (4 waves per SIMD at MT = 4: the kernel is bound by the latency of its L2 reads, occupancy matters
// more than a deeper ring or whole-line fragment pairs -- both were measured slower).
// Epilogue: v = act([FiLM per graph](acc + b)) [affine per channel] [* scale[n * sstride]] [+ Y]
// (ACT: 0 none, 1 SiLU, 2 ReLU, 3 tanh, 4 ELU).
// Optional row lists gather X rows / scatter Y rows (edges of one type, compacted on the device).
// KW = 4 (few rows: the loop of K / 16 dependent k-steps is the whole run time): the four waves share ONE
// 16 MT x 16 NT block, wave w takes the k-groups w, w + 4, ...; the partial blocks meet in LDS and wave 0 adds
// them in wave order (deterministic) before the epilogue.  grid = (ceil(N / (16 NT)), ceil(M / (16 MT))).
template <int ACT, int MT, int NT, int PF = 2, int KW = 1>
__global__ void __launch_bounds__(256)
k_s2s_linear(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ X,
             float* __restrict__ Y, int M, int K, int ldw, int64_t N, int ldy,
             const float* __restrict__ scale, int sstride, int accumulate,
             const int64_t* __restrict__ xidx /* row n of X is X[xidx[n]] */,
             const int64_t* __restrict__ yidx /* row n of Y (and of scale) is yidx[n] */,
             const int* __restrict__ n_dev /* non-null: N = *n_dev (compacted row lists) */,
             const float* __restrict__ post_scale /* non-null: v = v * post_scale[m] + post_shift[m] (BatchNorm, eval) */,
             const float* __restrict__ post_shift,
             const float* __restrict__ film_gamma /* non-null: v = (1 + gamma[g][m]) v + beta[g][m] before act, */,
             const float* __restrict__ film_beta  /* g = n / film_rows (FiLM per graph, nn/nn/film.py:58-60)   */,
             int film_rows) {
    if (n_dev != nullptr) N = *n_dev;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = KW == 1 ? (int)blockIdx.y * (32 * MT) + 16 * MT * (wave >> 1) : (int)blockIdx.y * (16 * MT);
    const int64_t n0 = KW == 1 ? (int64_t)blockIdx.x * (32 * NT) + 16 * NT * (wave & 1) : (int64_t)blockIdx.x * (16 * NT);
    if (m0 >= M || n0 >= N) return;                    // the same for every wave of the workgroup when KW > 1
    // rows / points past the end are clamped for the loads and masked at the store
    const float* wrow[MT];
    const float* xrow[NT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + i;
        wrow[t] = W + (size_t)(m < M ? m : M - 1) * ldw + 4 * q;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t n = n0 + 16 * t + i;
        n = n < N ? n : N - 1;
        if (xidx != nullptr) n = xidx[n];
        xrow[t] = X + (size_t)n * K + 4 * q;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
        f32x4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mb + 4 * q + r;
            b4[r] = (bias != nullptr && m < M && (KW == 1 || wave == 0)) ? bias[m] : 0.0f;
        }
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = b4;
    }
    // this wave's k-groups: kg(a) for a < steps (the host launches KW > 1 only with K / 16 >= KW)
    const int steps = KW == 1 ? K >> 4 : ((K >> 4) - wave + KW - 1) / KW;
    auto kg = [&](int a) { return KW == 1 ? a : wave + KW * a; };
    f32x4 wq[PF][MT], xq[PF][NT];                      // ring of prefetched fragments
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int a = kg(p < steps ? p : steps - 1);
#pragma unroll
        for (int t = 0; t < MT; ++t) wq[p][t] = ld4(wrow[t] + 16 * a);
#pragma unroll
        for (int t = 0; t < NT; ++t) xq[p][t] = ld4(xrow[t] + 16 * a);
    }
    for (int a0 = 0; a0 < steps; a0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (a0 + p < steps) {
                f32x4 wv[MT], xv[NT];
#pragma unroll
                for (int t = 0; t < MT; ++t) wv[t] = wq[p][t];
#pragma unroll
                for (int t = 0; t < NT; ++t) xv[t] = xq[p][t];
                const int an = kg(a0 + p + PF < steps ? a0 + p + PF : steps - 1);
#pragma unroll
                for (int t = 0; t < MT; ++t) wq[p][t] = ld4(wrow[t] + 16 * an);
#pragma unroll
                for (int t = 0; t < NT; ++t) xq[p][t] = ld4(xrow[t] + 16 * an);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = mfma16(wv[mb][b], xv[nb][b], acc[mb][nb]);
            }
        }
    }
    if constexpr (KW > 1) {
        __shared__ f32x4 red[KW - 1][MT * NT][64];
        if (wave != 0) {
#pragma unroll
            for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) red[wave - 1][mb * NT + nb][lane] = acc[mb][nb];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 0; w < KW - 1; ++w)
#pragma unroll
            for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                for (int nb = 0; nb < NT; ++nb) acc[mb][nb] += red[w][mb * NT + nb][lane];
    }
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        if (n >= N) continue;
        if (yidx != nullptr) n = yidx[n];
        const size_t g = film_gamma != nullptr ? (size_t)(n / film_rows) * M : 0;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) {
            const int m = m0 + 16 * mb + 4 * q;
            f32x4 v = acc[mb][nb];
            if (film_gamma != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < M) v[r] = (1.0f + film_gamma[g + m + r]) * v[r] + film_beta[g + m + r];
            }
            if (ACT == 1) v = silu4(v);
            if (ACT == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            }
            if (ACT == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanh1(v[r]);
            }
            if (ACT == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu1(v[r]);     // ELU, alpha = 1
            }
            if (post_scale != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < M) v[r] = v[r] * post_scale[m + r] + post_shift[m + r];
            }
            if (scale != nullptr) v = v * scale[(size_t)n * sstride];
            if (m + 3 < M) {
                if (accumulate) v += ld4(Y + (size_t)n * ldy + m);
                st4(Y + (size_t)n * ldy + m, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < M) Y[(size_t)n * ldy + m + r] = v[r] + (accumulate ? Y[(size_t)n * ldy + m + r] : 0.0f);
            }
        }
    }
}

}  // namespace

// =====================================================================================================
// Row A9: augmented local frames of the seq2seq model
//   AugmentedLocalizer.forward            nn/utils/augmented_global_to_local.py:52-68
//   canonicalize_augmented_inputs         nn/utils/canonicalization.py:33-56
//   create_augmented_edge_attr_pos_vel    canonicalization.py:111-140 (2-D), :143-172 (3-D)
//   angle helpers                         nn/utils/geometry.py:7-66,76-127
// Per node: canonical state, frame Rinv and the features of the edge from the virtual origin node
// (pos 0, vel e1, force 0: augmented_global_to_local.py:41).  Per edge j -> i: the features in i's
// frame followed by the receiver's node row.  Element-wise work with a handful of transcendentals per
// item: a thread per node / edge, every output row written contiguously (HBM-bound: 4 (2 NF + 3 D)
// bytes of output per edge against two node rows that stay in L2).
namespace {

constexpr float PI_S2S = 3.14159274101257324219f;         // float32(pi), as the reference's fp32 ops see it
constexpr float TWO_PI_A9 = 6.28318548202514648438f;

template <int D> struct AugDims {
    static constexpr int O = D * (D - 1) / 2, NF = 4 * D + O, RF = 3 * D + NF, EA = NF + RF, EP = D + O;
};

// rho, theta in [0, 2 pi), phi of a 3-vector (geometry.py:37-66, symmetric_theta = False)
__device__ __forceinline__ void spherical3(const float* v, float& rho, float& theta, float& phi) {
    rho = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    theta = atan2f(v[1], v[0]);
    if (theta < 0.0f) theta += TWO_PI_A9;
    phi = acosf(fminf(fmaxf(v[2] / (rho + 1e-7f), -1.0f), 1.0f));
}
__device__ __forceinline__ void rot3(float theta, float phi, float (&R)[3][3]) {      // geometry.py:24-34
    const float c = cosf(theta), s = sinf(theta), cp = cosf(phi), sp = sinf(phi);
    R[0][0] = cp * c; R[0][1] = -s;  R[0][2] = sp * c;
    R[1][0] = cp * s; R[1][1] = c;   R[1][2] = sp * s;
    R[2][0] = -sp;    R[2][1] = 0.f; R[2][2] = cp;
}

// features of the edge j -> i in i's frame; xj, xi = [pos | vel | force]
template <int D>
__device__ __forceinline__ void aug_edge(const float* xj, const float* xi, float* o) {
    if constexpr (D == 2) {
        const float yaw_i = atan2f(xi[3], xi[2]);
        const float c = cosf(yaw_i), s = sinf(yaw_i);                  // r = R(yaw_i)^T = [[c, s], [-s, c]]
        const float dx = xj[0] - xi[0], dy = xj[1] - xi[1];
        float dyaw = atan2f(xj[3], xj[2]) - yaw_i;                     // angle_diff, geometry.py:116-127
        if (dyaw >= PI_S2S) dyaw -= TWO_PI_A9;
        if (dyaw < -PI_S2S) dyaw += TWO_PI_A9;
        float dth = atan2f(dy, dx) - yaw_i;                            // wrap_angles(normalize=True), :108-113
        if (dth <= -PI_S2S) dth += TWO_PI_A9;
        if (dth > PI_S2S) dth -= TWO_PI_A9;
        o[0] = c * dx + s * dy; o[1] = -s * dx + c * dy;
        o[2] = dyaw / PI_S2S;
        o[3] = sqrtf(dx * dx + dy * dy);
        o[4] = dth / PI_S2S;
        o[5] = c * xj[2] + s * xj[3]; o[6] = -s * xj[2] + c * xj[3];
        o[7] = c * xj[4] + s * xj[5]; o[8] = -s * xj[4] + c * xj[5];
    } else {
        float rho, yj, pj, yi, pi_;
        spherical3(xj + 3, rho, yj, pj);
        spherical3(xi + 3, rho, yi, pi_);
        float Ri[3][3], Rj[3][3];
        rot3(yi, pi_, Ri);
        rot3(yj, pj, Rj);
        const float dp[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]};
        // M = Ri^T Rj^T (canonicalization.py:153-154: the sender matrix is transposed as well)
        auto Mel = [&](int a, int b) { return Ri[0][a] * Rj[b][0] + Ri[1][a] * Rj[b][1] + Ri[2][a] * Rj[b][2]; };
        float rdp[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rdp[a] = Ri[0][a] * dp[0] + Ri[1][a] * dp[1] + Ri[2][a] * dp[2];
            o[a] = rdp[a];
            o[9 + a] = Ri[0][a] * xj[3] + Ri[1][a] * xj[4] + Ri[2][a] * xj[5];
            o[12 + a] = Ri[0][a] * xj[6] + Ri[1][a] * xj[7] + Ri[2][a] * xj[8];
        }
        o[3] = atan2f(Mel(1, 0), Mel(0, 0));                           // ZYX, not normalised (geometry.py:87-100)
        o[4] = asinf(-Mel(2, 0));
        o[5] = atan2f(Mel(2, 1), Mel(2, 2));
        float dth, dph;
        spherical3(dp, o[6], dth, dph);                                // node distance
        spherical3(rdp, rho, o[7], o[8]);
    }
}

template <int D>
__global__ void __launch_bounds__(256)
k_s2s_aug_nodes(const float* __restrict__ x, float* __restrict__ rel_feat, float* __restrict__ Rinv,
                int64_t n_nodes, const float* __restrict__ inputs = nullptr, const float* __restrict__ field = nullptr,
                float* __restrict__ ext_out = nullptr) {
    // inputs != null: x[n] = [inputs[n] (2D) | field[n] (D)] is put together here and written to ext_out (k_s2s_extend
    // folded in: one launch less per stage of the variable-N step)
    using A = AugDims<D>;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_nodes) return;
    float xi[3 * D];
    if (inputs != nullptr) {
#pragma unroll
        for (int t = 0; t < 3 * D; ++t) {
            xi[t] = t < 2 * D ? inputs[n * 2 * D + t] : field[n * D + (t - 2 * D)];
            ext_out[n * 3 * D + t] = xi[t];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 3 * D; ++t) xi[t] = x[n * 3 * D + t];
    }
    float row[A::RF];
    float R[D][D];
    if constexpr (D == 2) {
        const float ang = atan2f(xi[3], xi[2]);
        const float c = cosf(ang), s = sinf(ang);
        R[0][0] = c; R[0][1] = -s; R[1][0] = s; R[1][1] = c;
        row[0] = 0.f; row[1] = 0.f; row[2] = sqrtf(xi[2] * xi[2] + xi[3] * xi[3]); row[3] = 0.f;
        row[4] = c * xi[4] + s * xi[5]; row[5] = -s * xi[4] + c * xi[5];
    } else {
        float rho, th, ph;
        spherical3(xi + 3, rho, th, ph);
        rot3(th, ph, R);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            row[a] = 0.f;
            row[3 + a] = R[0][a] * xi[3] + R[1][a] * xi[4] + R[2][a] * xi[5];
            row[6 + a] = R[0][a] * xi[6] + R[1][a] * xi[7] + R[2][a] * xi[8];
        }
    }
    float origin[3 * D];
#pragma unroll
    for (int t = 0; t < 3 * D; ++t) origin[t] = t == D ? 1.0f : 0.0f;
    aug_edge<D>(origin, xi, row + 3 * D);
#pragma unroll
    for (int t = 0; t < A::RF; ++t) rel_feat[n * A::RF + t] = row[t];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Rinv[n * D * D + a * D + b] = R[a][b];
}

// One workgroup = 256 consecutive edges.  Every thread builds its edge's row in LDS; the rows then leave
// as fully coalesced stores (consecutive lanes write consecutive floats): a thread writing its own 96- or
// 156-byte row scatters every store instruction over 64 rows (2.3 TB/s measured vs 4+ this way).
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_aug_edges(const float* __restrict__ x, const int64_t* __restrict__ send, const int64_t* __restrict__ recv,
                const float* __restrict__ rel_feat, int polar, float* __restrict__ edge_attr,
                float* __restrict__ edge_pos, int64_t n_edges, const int64_t* __restrict__ x_send = nullptr,
                const int64_t* __restrict__ x_recv = nullptr /* rows of x, when they differ from send / recv */,
                int out_cols = 0 /* > 0: only the first out_cols columns, rows packed at that stride (k_s2s_pad_rows folded in) */) {
    using A = AugDims<D>;
    constexpr int LDR = A::EA + 1;                                   // odd row stride: conflict-free column writes
    __shared__ float rows[256 * LDR];
    __shared__ float prow[256 * (A::EP + 1)];
    const int64_t e0 = (int64_t)blockIdx.x * 256;
    const int64_t e = e0 + threadIdx.x;
    if (e < n_edges) {
        const int64_t j = send[e], i = recv[e];
        const int64_t jx = x_send ? x_send[e] : j, ix = x_recv ? x_recv[e] : i;
        float xj[3 * D], xi[3 * D], o[A::NF];
#pragma unroll
        for (int t = 0; t < 3 * D; ++t) { xj[t] = x[jx * 3 * D + t]; xi[t] = x[ix * 3 * D + t]; }
        aug_edge<D>(xj, xi, o);
        float* out = rows + threadIdx.x * LDR;
#pragma unroll
        for (int t = 0; t < A::NF; ++t) out[t] = o[t];
#pragma unroll
        for (int t = 0; t < A::RF; ++t) out[A::NF + t] = rel_feat[i * A::RF + t];
        const int p0 = polar ? (D == 2 ? 2 : 3) : 0;                  // augmented_global_to_local.py:19-24
#pragma unroll
        for (int t = 0; t < A::EP; ++t) prow[threadIdx.x * (A::EP + 1) + t] = o[p0 + t];
    }
    __syncthreads();
    const int64_t left = n_edges - e0;
    const int cnt = (int)(left < 256 ? left : 256);
    const int oc = out_cols > 0 ? out_cols : A::EA;
    for (int idx = threadIdx.x; idx < cnt * oc; idx += 256) {
        const int r = idx / oc, c = idx - r * oc;
        edge_attr[e0 * oc + idx] = rows[r * LDR + c];
    }
    for (int idx = threadIdx.x; idx < cnt * A::EP; idx += 256) {
        const int r = idx / A::EP, c = idx - r * A::EP;
        edge_pos[e0 * A::EP + idx] = prow[r * (A::EP + 1) + c];
    }
}

}  // namespace

// =====================================================================================================
// Row A10, decoder half: RecurrentDecoder.forward (nn/seq2seq/aether.py:590-654), SURVEY.md Appendix B.4.
// The Linear layers run through k_s2s_linear; the first message Linear is split per node
// (W1 [h][2h] = [W_recv | W_send], aether.py:597-601: receivers first), so the [E, 2h] concatenation is
// never built: T1 = tanh(A[recv] + S[send]) with A = W_recv hid + b, S = W_send hid.
namespace {

// T[j][:] = tanh(A[recv[e]][:] + S[send[e]][:]), e = list[j] (j < *count); one thread per (row, 4 columns)
__global__ void __launch_bounds__(256)
k_s2s_pair_tanh(const float* __restrict__ A, const float* __restrict__ S, const int64_t* __restrict__ send,
                const int64_t* __restrict__ recv, const int64_t* __restrict__ list, const int* __restrict__ count,
                float* __restrict__ T, int h) {
    const int q4 = h >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)*count * q4) return;
    const int64_t j = idx / q4;
    const int c = (int)(idx - j * q4) * 4;
    const int64_t e = list[j];
    const f32x4 v = ld4(A + (size_t)recv[e] * h + c) + ld4(S + (size_t)send[e] * h + c);
    st4(T + (size_t)j * h + c, f32x4{tanh1(v[0]), tanh1(v[1]), tanh1(v[2]), tanh1(v[3])});
}

// list = the edges whose weight for type k is not zero (one-hot types: the edges of that type; soft
// types: all of them).  One slot range per wave (ballot + one atomic); the order of the list does not
// affect any result (every row is computed on its own and scattered back by edge id).
__global__ void __launch_bounds__(256)
k_s2s_select(const float* __restrict__ edge_w, int K, int k, int64_t n_edges, int64_t* __restrict__ list,
             int* __restrict__ count) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = e < n_edges && edge_w[e * K + k] != 0.0f;
    const unsigned long long mask = __ballot(on);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(count, __popcll(mask));
    base = __shfl(base, 0);
    if (on) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
}

// agg[n][:] = sum over the node's in-edges (order[rowptr[n] .. rowptr[n+1])) of M[edge][:] / max(count, 1):
// torch_scatter mean (aether.py:617,635).  One workgroup per node, a thread per 4 columns, fixed order.
__global__ void __launch_bounds__(128)
k_s2s_segment_mean(const float* __restrict__ Mx, const int64_t* __restrict__ order, const int64_t* __restrict__ rowptr,
                   float* __restrict__ agg, int h, float fixed_div /* > 0: sum / fixed_div (Encoder.edge2node) */,
                   const float* __restrict__ node_div = nullptr /* per-node divisor (batched scenes) */) {
    const int64_t n = blockIdx.x;
    const int64_t beg = rowptr[n], end = rowptr[n + 1];
    for (int c = threadIdx.x * 4; c < h; c += 128 * 4) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int64_t k = beg; k < end; ++k) s += ld4(Mx + (size_t)order[k] * h + c);
        const float cnt = node_div ? node_div[n] : (fixed_div > 0.0f ? fixed_div : (float)(end - beg > 1 ? end - beg : 1));
        st4(agg + (size_t)n * h + c, s / cnt);
    }
}

// dst[r][0..cols_pad) = src[r][0..cols) zero padded (row strides ld_src / cols_pad)
__global__ void __launch_bounds__(256)
k_s2s_pad_rows(const float* __restrict__ src, int cols, int ld_src, float* __restrict__ dst, int cols_pad,
               int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * cols_pad) return;
    const int64_t r = idx / cols_pad;
    const int c = (int)(idx - r * cols_pad);
    dst[idx] = c < cols ? src[r * ld_src + c] : 0.0f;
}

// ext[n] = [inputs[n] (2D) | field[n] (D)]   (aether.py:620)
__global__ void __launch_bounds__(256)
k_s2s_extend(const float* __restrict__ inputs, const float* __restrict__ field, float* __restrict__ ext, int D,
             int64_t n_nodes) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_nodes * 3 * D) return;
    const int64_t n = idx / (3 * D);
    const int c = (int)(idx - n * 3 * D);
    ext[idx] = c < 2 * D ? inputs[n * 2 * D + c] : field[n * D + (c - 2 * D)];
}

// GRU-style gate (aether.py:643-646): r = sigmoid(rp), i = sigmoid(ip), n = tanh(np + r * hh),
// hidden' = (1 - i) * n + i * hidden
__global__ void __launch_bounds__(256)
k_s2s_gate(const float* __restrict__ rp, const float* __restrict__ ip, const float* __restrict__ np_,
           const float* __restrict__ hh, const float* __restrict__ hidden, float* __restrict__ hidden_out,
           int64_t count) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= count) return;
    const float r = sigmoid1(rp[idx]);
    const float i = sigmoid1(ip[idx]);
    const float n = tanh1(np_[idx] + r * hh[idx]);
    hidden_out[idx] = (1.0f - i) * n + i * hidden[idx];
}

// outputs = inputs + [R y_pos | R y_vel]   (Globalizer, local_to_global.py:12-13; aether.py:651-652)
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_globalize(const float* __restrict__ inputs, const float* __restrict__ pred, const float* __restrict__ Rinv,
                float* __restrict__ out, int64_t n_nodes) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_nodes) return;
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk)
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < D; ++b) s += Rinv[n * D * D + a * D + b] * pred[n * 2 * D + chunk * D + b];
            out[n * 2 * D + chunk * D + a] = inputs[n * 2 * D + chunk * D + a] + s;
        }
}

}  // namespace

// =====================================================================================================
// Row A10, prior half: Encoder.single_step_forward (nn/seq2seq/aether.py:384-410), SURVEY.md Appendix B.5.
namespace {

// hw[e][c] = ELU(sum_p W1[c][p] pos[e][p] + b1[c]): the first layer of the hyper-network (pos_size = 3 | 6)
__global__ void __launch_bounds__(256)
k_s2s_pos_hidden(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ pos, int P,
                 float* __restrict__ hw, int h, int64_t n_edges, int relu = 0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * h) return;
    const int64_t e = idx / h;
    const int c = (int)(idx - e * h);
    float s = b1[c];
    for (int p = 0; p < P; ++p) s = fmaf(W1[c * P + p], pos[e * P + p], s);
    hw[idx] = relu ? fmaxf(s, 0.0f) : elu1(s);
}

// M[e][:] += relu(F[e][:]) * w[e * K]   (present messages of the variable-N decoder, aether_dynamicvars.py:831-835)
__global__ void __launch_bounds__(256)
k_s2s_relu_scale_acc(const float* __restrict__ Fm, const float* __restrict__ w, int K, float* __restrict__ Mx, int h,
                     int64_t n_edges) {
    const int q4 = h >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * q4) return;
    const int64_t e = idx / q4;
    const int c = (int)(idx - e * q4) * 4;
    const float we = w[e * K];
    f32x4 v = ld4(Fm + (size_t)e * h + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f) * we;
    st4(Mx + (size_t)e * h + c, ld4(Mx + (size_t)e * h + c) + v);
}

// dst[n][0..16) = [v / max(|v|, 1e-12) (2) | 0 ...], v = x[n][2..4)   (F.normalize of the velocity, padded to a k-group)
__global__ void __launch_bounds__(256)
k_s2s_unit_velocity(const float* __restrict__ x, float* __restrict__ dst, int64_t n_points) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_points) return;
    const float vx = x[n * 4 + 2], vy = x[n * 4 + 3];
    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy), 1e-12f);
    dst[n * 16] = vx / nrm;
    dst[n * 16 + 1] = vy / nrm;
#pragma unroll
    for (int c = 2; c < 16; ++c) dst[n * 16 + c] = 0.0f;
}

// Inputs of the variable-N field net in one launch (aether_dynamicvars.py:64-79): cat[n] = [Fourier features of the position
// (h) | angular_embedding(unit velocity) (h)] -- k_s2s_rff + k_s2s_unit_velocity + a zero-padded weight copy + a K = 2 Linear
// before (four launches of a ~60-launch step).  x [n][4] = [pos | vel]; Bm [2][h / 2]; ang_w [h][2], ang_b [h].
__global__ void __launch_bounds__(256)
k_dyn_field_embed(const float* __restrict__ x, const float* __restrict__ Bm, const float* __restrict__ ang_w,
                  const float* __restrict__ ang_b, int h, float* __restrict__ cat, int64_t n_points) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_points * h) return;
    const int64_t n = idx / h;
    const int j = (int)(idx - n * h), half = h >> 1;
    const float px = x[n * 4], py = x[n * 4 + 1], vx = x[n * 4 + 2], vy = x[n * 4 + 3];
    if (j < half) {
        float p = 0.0f;
        p = fmaf(TWO_PI_S2S * px, Bm[j], p);
        p = fmaf(TWO_PI_S2S * py, Bm[half + j], p);
        cat[n * 2 * h + j] = sinf(p);
        cat[n * 2 * h + half + j] = cosf(p);
    }
    const float nrm = fmaxf(sqrtf(vx * vx + vy * vy), 1e-12f);
    cat[n * 2 * h + h + j] = fmaf(ang_w[2 * j + 1], vy / nrm, fmaf(ang_w[2 * j], vx / nrm, ang_b[j]));
}

// dst[i] = src[i] * s
__global__ void __launch_bounds__(256)
k_s2s_scale(const float* __restrict__ src, float s, float* __restrict__ dst, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) dst[idx] = src[idx] * s;
}

// Anisotropic edge filter (nn/nn/anisotropic_filter.py:34-40):
//   out[e][c] = sum_r ea[e][r] * (b2[r h + c] + sum_k L2[r h + c][k] hw[e][k]),   r < R (24 | 39), c, k < h.
// The [E, R h] filter bank of the reference is never materialised: the contraction runs as one GEMM over
// K' = R h with the B operand formed on the fly, x[(r, k)] = ea[e][r] * hw[e][k] (one hw fragment per
// k-group, scaled by the edge's R feature values).  grid = (ceil(E / (32 NB)), h / 128, splits), 4 waves =
// 2 (c) x 2 (e); a wave owns 64 outputs x 16 NB edges (every L2 fragment feeds 4 NB MFMAs).
// With few edges (the reference's 5-object graphs: 2,560 edges = 80 workgroups on 256 CUs) the k-groups are
// split over gridDim.z; split z writes its partial sums to plane z of `out` (bias in plane 0) and
// k_s2s_sum_planes adds the planes in order.
template <int R, int NB, bool SPLIT>
__global__ void __launch_bounds__(256)
k_s2s_filter(const float* __restrict__ L2w, const float* __restrict__ b2, const float* __restrict__ ea,
             const float* __restrict__ hw, float* __restrict__ out, int h, int64_t n_edges) {
    constexpr int MT = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = (int)blockIdx.y * 128 + 64 * (wave >> 1);
    const int64_t n0 = (int64_t)blockIdx.x * (32 * NB) + 16 * NB * (wave & 1);
    // the edges' R feature values, staged once: evs[r][half][i][nb] (a lane's NB values are one 16-byte read;
    // indexing a per-thread array with the runtime r put it in scratch memory)
    __shared__ float evs[R * 32 * NB];
    for (int idx = threadIdx.x; idx < R * 32 * NB; idx += 256) {
        const int r = idx / (32 * NB), rem = idx - r * (32 * NB);
        const int half = rem / (16 * NB), ii = (rem - half * 16 * NB) / NB, nb = rem % NB;
        int64_t n = (int64_t)blockIdx.x * (32 * NB) + 16 * NB * half + 16 * nb + ii;
        n = n < n_edges ? n : n_edges - 1;
        evs[idx] = ea[(size_t)n * R + r];
    }
    __syncthreads();
    if (n0 >= n_edges) return;
    const int splits = SPLIT ? (int)gridDim.z : 1, z = SPLIT ? (int)blockIdx.z : 0;
    if (SPLIT) out += (size_t)z * n_edges * h;
    static_assert(NB == 4, "one 16-byte LDS read per step holds the NB feature values of a lane");
    const float* evl = evs + 16 * NB * (wave & 1) + NB * i;
    const float* hrow[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        n = n < n_edges ? n : n_edges - 1;
        hrow[nb] = hw + (size_t)n * h + 4 * q;
    }
    // bias term: sum_r ea[e][r] * b2[r h + m]
    f32x4 acc[MT][NB];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int r = 0; r < (SPLIT && z != 0 ? 0 : R); ++r) {
            const f32x4 bv = ld4(b2 + (size_t)r * h + m0 + 16 * mb + 4 * q);
            const f32x4 e4 = *reinterpret_cast<const f32x4*>(evl + r * (32 * NB));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] += bv * own_reg(e4[nb]);
        }
    }
    const float* wbase = L2w + (size_t)(m0 + i) * h + 4 * q;          // row (r h + m0 + 16 mb + i), k-group a
    const int steps = h >> 4;
    const int first = SPLIT ? (steps / splits) * z * R : 0;           // this split's k-groups: steps / splits of them
    // flat step s = a * R + r; the fragments of the next PF steps are in flight (L2 latency ~ 3 steps of
    // MFMAs: 14.0 -> 7.7 ms per prior step at 48,640 edges).  The loop stays rolled: fully unrolling r made
    // the compiler hoist every load (290 - 506 VGPRs, one wave per SIMD, 11.6 ms).
    constexpr int PF = 3;
    const int total = SPLIT ? first + (steps / splits) * R : steps * R;
    f32x4 ring[PF][MT];
    auto wfetch = [&](f32x4 (&dst)[MT], int sidx) {
        const int a = sidx / R, r = sidx - a * R;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) dst[mb] = ld4(wbase + ((size_t)r * h + 16 * mb) * h + 16 * a);
    };
#pragma unroll
    for (int p = 0; p < PF; ++p) wfetch(ring[p], first + p < total ? first + p : total - 1);
    f32x4 hf[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) hf[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = first; s0 < total; s0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int sidx = s0 + p;
            if (sidx < total) {
                const int a = sidx / R, r = sidx - a * R;
                if (r == 0) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) hf[nb] = ld4(hrow[nb] + 16 * a);
                }
                f32x4 wv[MT], xf[NB];
#pragma unroll
                for (int mb = 0; mb < MT; ++mb) wv[mb] = ring[p][mb];
                wfetch(ring[p], sidx + PF < total ? sidx + PF : total - 1);
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(evl + r * (32 * NB));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) xf[nb] = hf[nb] * own_reg(e4[nb]);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma16(wv[mb][b], xf[nb][b], acc[mb][nb]);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int64_t n = n0 + 16 * nb + i;
        if (n >= n_edges) continue;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) st4(out + (size_t)n * h + m0 + 16 * mb + 4 * q, acc[mb][nb]);
    }
}

// out[i] = planes[0][i] + planes[1][i] + ... (fixed order); count % 4 == 0
__global__ void __launch_bounds__(256)
k_s2s_sum_planes(const float* __restrict__ planes, int n_planes, int64_t count, float* __restrict__ out) {
    const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= count) return;
    f32x4 v = ld4(planes + idx);
    for (int z = 1; z < n_planes; ++z) v += ld4(planes + (size_t)z * count + idx);
    st4(out + idx, v);
}

// T[e][:] = ELU(T[e][:] + Ps[send[e]][:] + Pr[recv[e]][:])   (first Linear of mlp4 on [x_send | x_recv | edge])
__global__ void __launch_bounds__(256)
k_s2s_edge_sum_elu(float* __restrict__ T, const float* __restrict__ Ps, const float* __restrict__ Pr,
                   const int64_t* __restrict__ send, const int64_t* __restrict__ recv, int h, int64_t n_edges) {
    const int q4 = h >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * q4) return;
    const int64_t e = idx / q4;
    const int c = (int)(idx - e * q4) * 4;
    f32x4 v = ld4(T + (size_t)e * h + c) + ld4(Ps + (size_t)send[e] * h + c) + ld4(Pr + (size_t)recv[e] * h + c);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = elu1(v[r]);
    st4(T + (size_t)e * h + c, v);
}

// BatchNorm1d in eval mode as a per-channel affine: scale = w / sqrt(var + eps), shift = b - mean * scale
__global__ void __launch_bounds__(256)
k_s2s_bn_affine(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const float sc = w[c] / sqrtf(var[c] + 1e-5f);
    scale[c] = sc;
    shift[c] = b[c] - mean[c] * sc;
}

// Up to three BatchNorm layers in one launch (blockIdx.y = layer; layers without statistics are skipped by the host).
struct BnSets { const float* w[3]; const float* b[3]; const float* mean[3]; const float* var[3]; float* scale[3]; float* shift[3]; };
__global__ void __launch_bounds__(256)
k_s2s_bn_affine_sets(BnSets S, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (c >= n || S.w[j] == nullptr) return;
    const float sc = S.w[j][c] / sqrtf(S.var[j][c] + 1e-5f);
    S.scale[j][c] = sc;
    S.shift[j][c] = S.b[j][c] - S.mean[j][c] * sc;
}

// One LSTM step (torch.nn.LSTM gate order i, f, g, o): gates [E][4R] -> h1, c1
__global__ void __launch_bounds__(256)
k_s2s_lstm_cell(const float* __restrict__ gates, const float* __restrict__ c0, float* __restrict__ h1,
                float* __restrict__ c1, int R, int64_t n_edges) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * R) return;
    const int64_t e = idx / R;
    const int c = (int)(idx - e * R);
    const float* g = gates + (size_t)e * 4 * R;
    const float ig = sigmoid1(g[c]), fg = sigmoid1(g[R + c]);
    const float gg = tanh1(g[2 * R + c]), og = sigmoid1(g[3 * R + c]);
    const float cn = fg * c0[idx] + ig * gg;
    c1[idx] = cn;
    h1[idx] = og * tanh1(cn);
}

// gumbel_softmax(hard=True) with the uniform draw supplied (nn/utils/model_utils.py:58-118)
__global__ void __launch_bounds__(256)
k_s2s_gumbel_hard(const float* __restrict__ logits, const float* __restrict__ uniform, float tau, int K,
                  float* __restrict__ edges, int64_t n_edges) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    float y[4], mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float g = -logf(1e-10f - logf(uniform[e * K + k] + 1e-10f));
        y[k] = (logits[e * K + k] + g) / tau;
        mx = fmaxf(mx, y[k]);
    }
    float sum = 0.0f;
    for (int k = 0; k < K; ++k) { y[k] = expf(y[k] - mx); sum += y[k]; }
    int best = 0;
    for (int k = 0; k < K; ++k) { y[k] = y[k] / sum; if (y[k] > y[best]) best = k; }
    for (int k = 0; k < K; ++k) edges[e * K + k] = ((k == best ? 1.0f : 0.0f) - y[k]) + y[k];
}

}  // namespace
