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
          float* __restrict__ gamma, int64_t n_points) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_points * half) return;
    const int64_t n = idx / half;
    const int k = (int)(idx - n * half);
    float p = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) p = fmaf(TWO_PI_S2S * x[n * x_stride + d], Bm[d * half + k], p);
    gamma[n * 2 * half + k] = sinf(p);
    gamma[n * 2 * half + half + k] = cosf(p);
}

// Y[n][m] = act(sum_k W[m][k] X[n][k] + b[m]); W [M][K] row-major (nn.Linear), X [N][K], Y [N][ldy].
// K % 16 == 0.  grid = (ceil(N / 64), ceil(M / (32 MT))), 4 waves = 2 (m) x 2 (n); a wave owns
// MT x 2 MFMA tiles (16 MT rows x 32 points) and keeps the fragments of the next PF k-groups in flight
// (4 waves per SIMD at MT = 4: the kernel is bound by the latency of its L2 reads, occupancy matters
// more than a deeper ring or whole-line fragment pairs -- both were measured slower).
template <bool SILU, int MT>
__global__ void __launch_bounds__(256)
k_s2s_linear(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ X,
             float* __restrict__ Y, int M, int K, int64_t N, int ldy) {
    constexpr int PF = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = (int)blockIdx.y * (32 * MT) + 16 * MT * (wave >> 1);
    const int64_t n0 = (int64_t)blockIdx.x * 64 + 32 * (wave & 1);
    if (m0 >= M || n0 >= N) return;
    // rows / points past the end are clamped for the loads and masked at the store
    const float* wrow[MT];
    const float* xrow[2];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + i;
        wrow[t] = W + (size_t)(m < M ? m : M - 1) * K + 4 * q;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t n = n0 + 16 * t + i;
        xrow[t] = X + (size_t)(n < N ? n : N - 1) * K + 4 * q;
    }
    f32x4 acc[MT][2];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
        f32x4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int m = m0 + 16 * mb + 4 * q + r; b4[r] = m < M ? bias[m] : 0.0f; }
        acc[mb][0] = b4; acc[mb][1] = b4;
    }
    const int steps = K >> 4;
    f32x4 wq[PF][MT], xq[PF][2];                      // ring of prefetched fragments
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int a = p < steps ? p : steps - 1;
#pragma unroll
        for (int t = 0; t < MT; ++t) wq[p][t] = ld4(wrow[t] + 16 * a);
#pragma unroll
        for (int t = 0; t < 2; ++t) xq[p][t] = ld4(xrow[t] + 16 * a);
    }
    for (int a0 = 0; a0 < steps; a0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (a0 + p < steps) {
                f32x4 wv[MT], xv[2];
#pragma unroll
                for (int t = 0; t < MT; ++t) wv[t] = wq[p][t];
#pragma unroll
                for (int t = 0; t < 2; ++t) xv[t] = xq[p][t];
                const int an = a0 + p + PF < steps ? a0 + p + PF : steps - 1;
#pragma unroll
                for (int t = 0; t < MT; ++t) wq[p][t] = ld4(wrow[t] + 16 * an);
#pragma unroll
                for (int t = 0; t < 2; ++t) xq[p][t] = ld4(xrow[t] + 16 * an);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = mfma16(wv[mb][b], xv[nb][b], acc[mb][nb]);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int64_t n = n0 + 16 * nb + i;
        if (n >= N) continue;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) {
            const int m = m0 + 16 * mb + 4 * q;
            f32x4 v = acc[mb][nb];
            if (SILU) v = silu4(v);
            if (m + 3 < M) {
                st4(Y + (size_t)n * ldy + m, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (m + r < M) Y[(size_t)n * ldy + m + r] = v[r];
            }
        }
    }
}

}  // namespace
