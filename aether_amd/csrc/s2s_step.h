// One autoregressive step of the seq2seq Aether in ~31 launches instead of ~75: field query -> prior
// step -> hard Gumbel sample -> decoder step (nn/seq2seq/aether.py:176-185, :384-410, :590-654), on prepared weights
// (host_s2s_step.inc: aether_s2s_plan_build).  What changes against the per-module entry points (host_seq2seq.inc):
//   * dense layers that share nothing but the launch go into ONE launch (k_s2s_linear_jobs: a table of up to eight
//     independent GEMM jobs, workgroups numbered through the table): the four first-layer products of the hidden-state
//     messages, the two edge types of every message layer, the sender / receiver halves of mlp4, and the four gate
//     pre-activations -- the latter as K-concatenated products [W_in | W_present | W_hidden] . [rel | agg_p | agg_h] on
//     one wide activation row instead of three accumulating launches each;
//   * the local frames are built once per step (the decoder's AugmentedLocalizer sees the same [inputs | field] as the
//     prior's), by one node kernel (extend + canonicalise + padded copies) and one edge kernel;
//   * padded weights, BatchNorm affines and the filter image come from the plan; the scatter targets are cleared by
//     kernels that run anyway; partial planes of the filter GEMM are added while the in-edge sums are taken.
#pragma once
#include "seq2seq.h"

namespace {

constexpr int S2S_MAX_JOBS = 8;
constexpr int S2S_RFG = 32;            // columns of the decoder's wide gate row that hold rel_feat (7D + O <= 24, zero padded)

// Y[n][m] = act(sum_k W[m][k] X[n][k] + b[m] [+ G1[i1[n]][m] + G2[i2[n]][m]]) [affine] [* scale[n]] [+ Y]
struct S2SJob {
    const float* W; const float* bias; const float* X; float* Y;
    const float* scale; const float* post_scale; const float* post_shift;
    const float* g1; const float* g2;            // epilogue gathers: rows g1[i1[n]], g2[i2[n]] (stride M) added before act
    const int64_t* i1; const int64_t* i2;
    const int64_t* xidx; const int64_t* yidx; const int* n_dev;
    const float* W2; const float* X2;            // second K segment: + sum_k W2[m][k] X2[n][k], k < K2 (K-concatenated product)
    const void* Wimg; const void* W2img;         // bf16 x 3 images of W / W2 (k_s2s_gemm_image), or NULL: fp32 MFMA only
    int64_t N;
    int M, K, ldw, ldx, ldy, sstride, act, accumulate, K2, ldw2, ldx2;
    int wg0, gx;                                 // first workgroup of the job, workgroups along n
    // act == 5 (k_s2s_gemm_split only): the rows of W are the LSTM's gates interleaved by unit (image row 4 u + g = gate g of
    // unit u: k_s2s_gemm_image with gate_units = R, bias permuted alike), so a lane's four accumulator rows are (i, f, g, o) of
    // ONE unit: the cell update runs in the epilogue -- c1 = sig(f) c0 + sig(i) tanh(g), h1 = sig(o) tanh(c1) -- and Y is not written
    const float* cell_c0; float* cell_h1; float* cell_c1;       // [N][M / 4]
};
struct S2SJobs { int n; S2SJob j[S2S_MAX_JOBS]; };

// The body of k_s2s_linear (seq2seq.h) with the activation, the row stride of X and the job chosen at run time.
template <int MT, int NT, int PF, int KW, bool TWO>
__global__ void __launch_bounds__(256)
k_s2s_linear_jobs(const S2SJobs jobs) {
    int ji = 0;
#pragma unroll
    for (int t = 1; t < S2S_MAX_JOBS; ++t)
        if (t < jobs.n && (int)blockIdx.x >= jobs.j[t].wg0) ji = t;
    const S2SJob& J = jobs.j[ji];
    const int local = (int)blockIdx.x - J.wg0, bx = local % J.gx, by = local / J.gx;
    int64_t N = J.N;
    if (J.n_dev != nullptr) N = *J.n_dev;
    const int M = J.M, K = J.K, ldw = J.ldw, ldx = J.ldx, ldy = J.ldy;
    const float* __restrict__ W = J.W;
    const float* __restrict__ X = J.X;
    float* __restrict__ Y = J.Y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = KW == 1 ? by * (32 * MT) + 16 * MT * (wave >> 1) : by * (16 * MT);
    const int64_t n0 = KW == 1 ? (int64_t)bx * (32 * NT) + 16 * NT * (wave & 1) : (int64_t)bx * (16 * NT);
    if (m0 >= M || n0 >= N) return;                    // the same for every wave of the workgroup when KW > 1
    const float* wrow[MT];
    const float* xrow[NT];
    const float* wrow2[MT];
    const float* xrow2[NT];
    const int k1g = K >> 4, k2g = (TWO && J.W2 != nullptr) ? J.K2 >> 4 : 0;      // TWO = false: no job of the launch has a second segment
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = m0 + 16 * t + i;
        wrow[t] = W + (size_t)(m < M ? m : M - 1) * ldw + 4 * q;
        wrow2[t] = k2g ? J.W2 + (size_t)(m < M ? m : M - 1) * J.ldw2 + 4 * q - 16 * k1g : wrow[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int64_t n = n0 + 16 * t + i;
        n = n < N ? n : N - 1;
        if (J.xidx != nullptr) n = J.xidx[n];
        xrow[t] = X + (size_t)n * ldx + 4 * q;
        xrow2[t] = k2g ? J.X2 + (size_t)n * J.ldx2 + 4 * q - 16 * k1g : xrow[t];
    }
    auto ldw_ = [&](int t, int a) { return ld4((!TWO || a < k1g ? wrow[t] : wrow2[t]) + 16 * a); };
    auto ldx_ = [&](int t, int a) { return ld4((!TWO || a < k1g ? xrow[t] : xrow2[t]) + 16 * a); };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
        f32x4 b4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mb + 4 * q + r;
            b4[r] = (J.bias != nullptr && m < M && (KW == 1 || wave == 0)) ? J.bias[m] : 0.0f;
        }
#pragma unroll
        for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = b4;
    }
    const int kgroups = k1g + k2g;
    const int steps = KW == 1 ? kgroups : (kgroups - wave + KW - 1) / KW;
    auto kg = [&](int a) { return KW == 1 ? a : wave + KW * a; };
    if (steps > 0) {
        f32x4 wq[PF][MT], xq[PF][NT];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int a = kg(p < steps ? p : steps - 1);
#pragma unroll
            for (int t = 0; t < MT; ++t) wq[p][t] = ldw_(t, a);
#pragma unroll
            for (int t = 0; t < NT; ++t) xq[p][t] = ldx_(t, a);
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
                    for (int t = 0; t < MT; ++t) wq[p][t] = ldw_(t, an);
#pragma unroll
                    for (int t = 0; t < NT; ++t) xq[p][t] = ldx_(t, an);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int mb = 0; mb < MT; ++mb)
#pragma unroll
                            for (int nb = 0; nb < NT; ++nb) acc[mb][nb] = mfma16(wv[mb][b], xv[nb][b], acc[mb][nb]);
                }
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
    const int act = J.act;
#pragma unroll
    for (int nb = 0; nb < NT; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        if (n >= N) continue;
        const int64_t nsrc = n;
        if (J.yidx != nullptr) n = J.yidx[n];
        const float* g1 = J.g1 != nullptr ? J.g1 + (size_t)J.i1[nsrc] * M : nullptr;
        const float* g2 = J.g2 != nullptr ? J.g2 + (size_t)J.i2[nsrc] * M : nullptr;
#pragma unroll
        for (int mb = 0; mb < MT; ++mb) {
            const int m = m0 + 16 * mb + 4 * q;
            f32x4 v = acc[mb][nb];
            if (g1 != nullptr && m + 3 < M) v += ld4(g1 + m) + ld4(g2 + m);
            if (act == 1) v = silu4(v);
            else if (act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            } else if (act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanh1(v[r]);
            } else if (act == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu1(v[r]);
            }
            if (J.post_scale != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < M) v[r] = v[r] * J.post_scale[m + r] + J.post_shift[m + r];
            }
            if (J.scale != nullptr) v = v * J.scale[(size_t)n * J.sstride];
            if (m + 3 < M) {
                if (J.accumulate) v += ld4(Y + (size_t)n * ldy + m);
                st4(Y + (size_t)n * ldy + m, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < M) Y[(size_t)n * ldy + m + r] = v[r] + (J.accumulate ? Y[(size_t)n * ldy + m + r] : 0.0f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Large dense layers (>= 2 K rows) on the matrix pipe: the job-table GEMM as THREE fp16 MFMA terms on operands split into two
// fp16 pieces (common.h; rounds 2-3: six bf16 terms on three pieces).  Weights come as prepared images (the plan), the
// activation rows are split in registers.
//   image of W [M][K] (row stride ldw): [k block K/32][row block M/16][piece 2][lane (i, q)] = W[16 mb + i][32 a + 8 q .. + 8)
// fp16's exponent range: every ROW of activations carries a power-of-two scale s that only ever shrinks while the K loop
// runs -- the first k step sets it from the row's maximum (max -> 2^13..2^14), a later step whose scaled values would reach
// 2^15 lowers it and multiplies the row's accumulators by the ratio (exact); the epilogue divides by the final s.  Every
// value is thus represented to 2^-22 of the largest value its row has shown so far.  Weights are split as they are
// (|w| < 65,504; a lo piece below |w| = 2^-3 is a subnormal: absolute error <= 2^-25).
// gate_units > 0: image row 4 u + g is row g * gate_units + u of W (the four gates of an LSTM unit side by side)
__global__ void __launch_bounds__(256)
k_s2s_gemm_image(const float* __restrict__ W, int M, int K, int ldw, f16x8* __restrict__ img, int gate_units) {
    const int oct = K >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (row, k octet)
    if (idx >= (int64_t)M * oct) return;
    const int m = (int)(idx / oct), o = (int)(idx - (int64_t)m * oct);
    const int ms = gate_units > 0 ? (m & 3) * gate_units + (m >> 2) : m; // source row
    const f32x4 v0 = ld4(W + (size_t)ms * ldw + 8 * o), v1 = ld4(W + (size_t)ms * ldw + 8 * o + 4);
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 a, b;
        split_f16x2(j < 4 ? v0[j] : v1[j - 4], a, b);
        hi[j] = a; lo[j] = b;
    }
    const int a32 = o >> 2, q = o & 3, mb = m >> 4, i = m & 15;
    const size_t frag = ((size_t)a32 * (M >> 4) + mb) * 2;
    img[(frag + 0) * 64 + i + 16 * q] = hi;
    img[(frag + 1) * 64 + i + 16 * q] = lo;
}

// Workgroup = 4 waves (one per SIMD, one workgroup per CU) on a 128 (m) x 64 NB (rows n) tile; a wave owns 128 m x 16 NB rows
// (8 x NB accumulator blocks).  Per 32-wide k step BOTH operands arrive by LDS-DMA in a four-slot ring, requested three steps
// before they are multiplied: the 16 KB of weight fragments (128 m x 32 k x 2 pieces, contiguous in the image; every wave
// reads all of them: one barrier per step) and the wave's own 16 NB rows of X (fp32, 2 KB per row block; read by the wave
// that asked for them: no barrier).  Rounds 2 - 4 (first half) loaded X into registers ONE step ahead: every step then waited
// out a memory round trip (2.2 us per step against 0.35 us of MFMAs: 28 % of the matrix pipe at 48,640 rows, and 24 us for
// a 2,560-row layer whose 160 workgroups each walk 16 dependent steps).
// X in LDS: one DMA instruction moves 16 rows x 64 bytes -- lane L asks for row L & 15, 16-byte chunk L >> 4 of the half row
// and lands at byte 16 L: [chunk 4][row 16] x 16 B, so the read back (lane (i, q): chunks 2 q, 2 q + 1 of row i) is
// consecutive over i.  The rows are scaled and split into two fp16 pieces for step s + 1 UNDER step s's MFMAs: row maxima
// and a speculative split with the current scales sit in the middle of the MFMA stream; only a row whose scale has to
// shrink (first step; rare afterwards) pays for a second split behind it.
// Epilogue and job table as k_s2s_linear_jobs.  Needs M % 128 == 0, K % 32 == 0 (both segments).
constexpr int GS_STAGE = 8 * 2 * 64;             // f16x8 weight fragments per k step (16 KB)
constexpr int GS_NST = 4;
constexpr int GS_XBLK = 2048;                    // bytes of one row block of X per step: 16 rows x 32 k x 4 B
__host__ __device__ constexpr size_t gs_lds_bytes(int nb) { return (size_t)GS_NST * (GS_STAGE * 16 + 4 * nb * GS_XBLK); }
template <int NB>          // row blocks of 16 per wave: 2 (128-row tiles) once those fill the chip twice over, else 1 (64-row tiles)
__global__ void __launch_bounds__(256, 2)      // (2: 256 registers a lane, all of them VGPRs -- at 1 the accumulators moved to AGPRs and
                                              // every step copied the 64 of them out and back around the rescale branch)
k_s2s_gemm_split(const S2SJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];
    f16x8* ring = reinterpret_cast<f16x8*>(gs_smem);
    constexpr int XSTAGE = 4 * NB * GS_XBLK;                  // bytes of X per slot
    constexpr int NDMA = 4 + 2 * NB;                          // DMA instructions per wave and step
    int ji = 0;
#pragma unroll
    for (int t = 1; t < S2S_MAX_JOBS; ++t)
        if (t < jobs.n && (int)blockIdx.x >= jobs.j[t].wg0) ji = t;
    const S2SJob& J = jobs.j[ji];
    // (row tile, column block) of the workgroup: the M / 128 column blocks of a row tile sit in consecutive slots of ONE XCD
    // (workgroups go round-robin over the 8 XCDs), so that the tile's rows of X come from memory once and from that XCD's L2
    // for the other column blocks.  The job's grid is padded to 8 x ceil(gx / 8) row tiles; the extra ones leave below.
    const int local = (int)blockIdx.x - J.wg0, gy = J.M >> 7, slot = local >> 3;
    const int bx = (local & 7) + 8 * (slot / gy), by = slot % gy;
    int64_t N = J.N;
    if (J.n_dev != nullptr) N = *J.n_dev;
    const int M = J.M, ldy = J.ldy;
    float* __restrict__ Y = J.Y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = by * 128;
    const int64_t n0 = (int64_t)bx * (64 * NB) + 16 * NB * wave;
    if ((int64_t)bx * (64 * NB) >= N) return;                // the whole workgroup
    const int s1 = J.K >> 5, s2 = J.W2img != nullptr ? J.K2 >> 5 : 0, S = s1 + s2;
    const int n_mb = M >> 4;
    const f16x8* img1 = reinterpret_cast<const f16x8*>(J.Wimg);
    const f16x8* img2 = reinterpret_cast<const f16x8*>(J.W2img);
    // DMA sources in X: row i of the wave's row blocks (clamped; gathered through xidx), chunk q of a half row
    const float* xr1[NB];
    const float* xr2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        n = n < N ? n : N - 1;
        if (J.xidx != nullptr) n = J.xidx[n];
        xr1[nb] = J.X + (size_t)n * J.ldx + 4 * q;
        xr2[nb] = s2 ? J.X2 + (size_t)n * J.ldx2 + 4 * q : xr1[nb];
    }
    unsigned char* xring = gs_smem + (size_t)GS_NST * GS_STAGE * 16 + wave * (NB * GS_XBLK);      // + slot * XSTAGE
    auto dma = [&](int s) {                                   // both operands of step s -> slot s % NST
        const int slot = s & (GS_NST - 1);
        const f16x8* src = (s < s1 ? img1 + ((size_t)s * n_mb + (m0 >> 4)) * 2 * 64
                                    : img2 + ((size_t)(s - s1) * n_mb + (m0 >> 4)) * 2 * 64) + lane;
        f16x8* dst = ring + slot * GS_STAGE;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fr = wave + 4 * f;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + fr * 64),
                                             (__attribute__((address_space(3))) void*)(dst + fr * 64), 16, 0, 0);
        }
        unsigned char* xd = xring + slot * XSTAGE;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float* p = s < s1 ? xr1[nb] + 32 * s : xr2[nb] + 32 * (s - s1);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                __builtin_amdgcn_global_load_lds(p + 16 * hf, (__attribute__((address_space(3))) void*)(xd + nb * GS_XBLK + hf * 1024),
                                                 16, 0, 0);
        }
    };
    // this lane's two chunks of row i in a row block: half q >> 1, chunks 2 (q & 1) and 2 (q & 1) + 1
    const unsigned xlane = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)xring + (q >> 1) * 1024 +
                           (32 * (q & 1) + i) * 16;
    f32x4 xa[NB][2];                                          // X of the next step: [nb][half of the lane's eight k]
    auto xread = [&](int s) {
        const unsigned base = xlane + (s & (GS_NST - 1)) * XSTAGE;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(xa[nb][0]) : "v"(base), "n"(nb * GS_XBLK));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(xa[nb][1]) : "v"(base), "n"(nb * GS_XBLK + 256));
        }
    };
    f32x4 acc[8][NB];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (J.bias != nullptr) b4 = ld4(J.bias + m0 + 16 * mb + 4 * q);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = b4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // bias / index loads done: the counts below are exact
    dma(0);
    if (S > 1) dma(1);
    if (S > 2) dma(2);
    // Activation scale: ONE power of two per ROW of X (lane (i, q) holds eight k of row 16 nb + i: the four q lanes of a row
    // agree on its maximum through two cross-lane exchanges), 0 = not set yet; the accumulators of a row hold xs x sums.
    // Per row, not per wave: the rows a wave works on may be gathered through per-type lists whose order differs from run to
    // run (atomic appends) -- a scale shared by whichever rows meet in a wave made the result depend on that order (found by
    // tools/s2s_soak.py: 1-ulp differences between runs).
    float xs[NB], xm[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xs[nb] = 0.0f;
    f16x8 xh[NB], xl[NB], xh_n[NB], xl_n[NB];
    // row maxima of the step held in xa (bit patterns: non-negative floats order like unsigned integers; the four q lanes of
    // a row sit in the wave's four rows of 16 lanes: two VALU row swaps, no LDS traffic), and its split with the scales as
    // they stand
    auto x_rowmax = [&]() {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v = 0.0f;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                v = fmaxf(fmaxf(v, fmaxf(fabsf(xa[nb][hh][0]), fabsf(xa[nb][hh][1]))), fmaxf(fabsf(xa[nb][hh][2]), fabsf(xa[nb][hh][3])));
            unsigned u = __float_as_uint(v);
            const auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            u = r16[0] > r16[1] ? r16[0] : r16[1];
            const auto r32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            u = r32[0] > r32[1] ? r32[0] : r32[1];
            xm[nb] = __uint_as_float(u);                          // the row's maximum over this step's 32 k, in all four q lanes
        }
    };
    auto x_split = [&](int nb) {                                  // (natural k order: the halves are 8 q .. + 4, + 4 .. + 8)
        split8(xa[nb][0] * xs[nb], xa[nb][1] * xs[nb], xh_n[nb], xl_n[nb]);
    };
    // behind the step's MFMAs: the first step sets the scales; a row that would reach 2^15 lowers its own and rescales its
    // accumulators (wave-uniform branch, rare); then the split pieces become the current ones
    auto x_settle = [&]() {
        bool redo_lane = false;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) redo_lane = redo_lane || xs[nb] == 0.0f || xm[nb] * xs[nb] >= 32768.0f;
        if (__builtin_amdgcn_ballot_w64(redo_lane) != 0ull) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const unsigned E = (__float_as_uint(xm[nb]) >> 23) & 255u;
                int sh = 140 - (int)E;                            // max (2^(E-127) ..) -> 2^13 ..
                sh = sh > 40 ? 40 : (sh < -40 ? -40 : sh);
                float ns = __int_as_float((127 + sh) << 23);
                float ratio;
                if (xs[nb] == 0.0f) ratio = ns;                   // first time: the accumulators hold the bias
                else if (xm[nb] * xs[nb] >= 32768.0f) { ns = ns < xs[nb] ? ns : xs[nb]; ratio = ns / xs[nb]; }    // (powers of two: exact)
                else { ns = xs[nb]; ratio = 1.0f; }
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) acc[mb][nb] = acc[mb][nb] * ratio;
                xs[nb] = ns;
                split8(xa[nb][0] * ns, xa[nb][1] * ns, xh_n[nb], xl_n[nb]);
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) { xh[nb] = xh_n[nb]; xl[nb] = xl_n[nb]; }
    };
#define GS_XWAIT(cnt)                                                                                             \
    do {                                                                                                          \
        if constexpr (NB == 2)                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1])); \
        else                                                                                                      \
            asm volatile("s_waitcnt lgkmcnt(" #cnt ")" : "+v"(xa[0][0]), "+v"(xa[0][1]));                       \
    } while (0)
#define GS_XTIE()                                                                                                 \
    do {                                                                                                          \
        if constexpr (NB == 2) asm volatile("" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1])); \
        else asm volatile("" : "+v"(xa[0][0]), "+v"(xa[0][1]));                                                 \
    } while (0)
    // X(0): the wave's own requests of step 0 have landed (those of steps 1 and 2 may be in flight: vmcnt counts in order)
    if (S > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
    else if (S > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xread(0);
    GS_XWAIT(0);
    x_rowmax();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) x_split(nb);
    x_settle();
    for (int s = 0; s < S; ++s) {
        // this wave's requests of step s have landed; those of steps s + 1 and s + 2 may be in flight
        if (s + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        else if (s + 1 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                        // weight fragments of step s visible to every wave; slot (s - 1) % NST is free
        if (s + 3 < S) dma(s + 3);
        const bool more = s + 1 < S;                          // (uniform)
        const f16x8* st = ring + (s & (GS_NST - 1)) * GS_STAGE + lane;
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)st;
        // fragments of row block mb + 1 are requested before the MFMAs of row block mb (LDS returns in order: lgkmcnt(2)
        // leaves exactly the newer two outstanding)
        f16x8 w[2][2];
#define GS_READ(buf, mb)                                                                                          \
        do {                                                                                                      \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w[buf][0]) : "v"(base), "n"(((mb) * 2 + 0) * 1024)); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w[buf][1]) : "v"(base), "n"(((mb) * 2 + 1) * 1024)); \
        } while (0)
        GS_READ(0, 0);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int cur = mb & 1;
            if (mb == 3 && more) {
                // half way: the wave's X rows of step s + 1 have landed (steps s + 2, s + 3 may be in flight) -> registers;
                // the reads are older than the two weight reads below, so the wait there covers them
                if (s + 3 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
                else if (s + 2 < S) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                xread(s + 1);
            }
            if (mb < 7) {
                if (cur == 0) GS_READ(1, mb + 1); else GS_READ(0, mb + 1);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w[cur][0]), "+v"(w[cur][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[cur][0]), "+v"(w[cur][1]));
            }
            if (mb == 3 && more) GS_XTIE();
            const f16x8 wh = w[cur][0], wl = w[cur][1];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[nb], acc[mb][nb], 0, 0, 0);
            // VALU work of the next step under this step's MFMAs, a piece per row block
            if (more) {
                if (mb == 4) x_rowmax();
                if (mb == 5) x_split(0);
                if (mb == 6 && NB == 2) x_split(NB - 1);
            }
        }
#undef GS_READ
        if (more) x_settle();
    }
#undef GS_XWAIT
#undef GS_XTIE
    float inv_xs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) inv_xs[nb] = xs[nb] != 0.0f ? 1.0f / xs[nb] : 1.0f;      // (powers of two: exact)
    const int act = J.act;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        if (n >= N) continue;
        const int64_t nsrc = n;
        if (J.yidx != nullptr) n = J.yidx[n];
        const float* g1 = J.g1 != nullptr ? J.g1 + (size_t)J.i1[nsrc] * M : nullptr;
        const float* g2 = J.g2 != nullptr ? J.g2 + (size_t)J.i2[nsrc] * M : nullptr;
        const float sc = J.scale != nullptr ? J.scale[(size_t)n * J.sstride] : 1.0f;
        if (act == 5) {                                       // LSTM cell on the gate pre-activations (rows interleaved by unit)
            const int Ru = M >> 2;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int u = ((m0 + 16 * mb) >> 2) + q;      // rows 4 u .. 4 u + 3
                const f32x4 v = acc[mb][nb] * inv_xs[nb];
                const float ig = sigmoid1(v[0]), fg = sigmoid1(v[1]);
                const float gg = tanh1(v[2]), og = sigmoid1(v[3]);
                const float cn = fg * J.cell_c0[(size_t)n * Ru + u] + ig * gg;
                J.cell_c1[(size_t)n * Ru + u] = cn;
                J.cell_h1[(size_t)n * Ru + u] = og * tanh1(cn);
            }
            continue;
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = m0 + 16 * mb + 4 * q;
            f32x4 v = acc[mb][nb] * inv_xs[nb];
            if (g1 != nullptr) v += ld4(g1 + m) + ld4(g2 + m);
            if (act == 1) v = silu4(v);
            else if (act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            } else if (act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanh1(v[r]);
            } else if (act == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu1(v[r]);
            }
            if (J.post_scale != nullptr) v = v * ld4(J.post_scale + m) + ld4(J.post_shift + m);
            v = v * sc;
            if (J.accumulate) v += ld4(Y + (size_t)n * ldy + m);
            st4(Y + (size_t)n * ldy + m, v);
        }
    }
}

// (A/B: the rounds 2 - 4 structure: X in registers one step ahead, two workgroups per CU)
constexpr int GS1_NST = 3;
template <int NB>          // row blocks of 16 per wave: 2 (128-row tiles) from 16 K rows on, 1 (64-row tiles) for 2 K - 16 K rows
__global__ void __launch_bounds__(256, 2)
k_s2s_gemm_split_r1(const S2SJobs jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];
    f16x8* ring = reinterpret_cast<f16x8*>(gs_smem);
    int ji = 0;
#pragma unroll
    for (int t = 1; t < S2S_MAX_JOBS; ++t)
        if (t < jobs.n && (int)blockIdx.x >= jobs.j[t].wg0) ji = t;
    const S2SJob& J = jobs.j[ji];
    // (row tile, column block) of the workgroup: the M / 128 column blocks of a row tile sit in consecutive slots of ONE XCD
    // (workgroups go round-robin over the 8 XCDs), so that the tile's rows of X come from memory once and from that XCD's L2
    // for the other column blocks.  The job's grid is padded to 8 x ceil(gx / 8) row tiles; the extra ones leave below.
    const int local = (int)blockIdx.x - J.wg0, gy = J.M >> 7, slot = local >> 3;
    const int bx = (local & 7) + 8 * (slot / gy), by = slot % gy;
    int64_t N = J.N;
    if (J.n_dev != nullptr) N = *J.n_dev;
    const int M = J.M, ldy = J.ldy;
    float* __restrict__ Y = J.Y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m0 = by * 128;
    const int64_t n0 = (int64_t)bx * (64 * NB) + 16 * NB * wave;
    if ((int64_t)bx * (64 * NB) >= N) return;                // the whole workgroup
    const int s1 = J.K >> 5, s2 = J.W2img != nullptr ? J.K2 >> 5 : 0, S = s1 + s2;
    const int n_mb = M >> 4;
    const f16x8* img1 = reinterpret_cast<const f16x8*>(J.Wimg);
    const f16x8* img2 = reinterpret_cast<const f16x8*>(J.W2img);
    // rows of X for the wave's two blocks (clamped; gathered through xidx)
    const float* xr1[NB];
    const float* xr2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        n = n < N ? n : N - 1;
        if (J.xidx != nullptr) n = J.xidx[n];
        xr1[nb] = J.X + (size_t)n * J.ldx + 8 * q;
        xr2[nb] = s2 ? J.X2 + (size_t)n * J.ldx2 + 8 * q : xr1[nb];
    }
    auto dma = [&](int s) {                                   // weight fragments of step s -> slot s % NST
        const f16x8* src = (s < s1 ? img1 + ((size_t)s * n_mb + (m0 >> 4)) * 2 * 64
                                    : img2 + ((size_t)(s - s1) * n_mb + (m0 >> 4)) * 2 * 64) + lane;
        f16x8* dst = ring + (s % GS1_NST) * GS_STAGE;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fr = wave + 4 * f;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src + fr * 64),
                                             (__attribute__((address_space(3))) void*)(dst + fr * 64), 16, 0, 0);
        }
    };
    f32x4 xa[NB][2];                                          // the next step's X values: [nb][half]
    // (Plain loads: as inline assembly their destination registers are loop-carried values the compiler knows nothing
    // asynchronous about -- it copied them at the back edge before the data had arrived.  The price: its wait in front of
    // the split is vmcnt(0), which also drains the DMA of step s + 1.)
    auto xload = [&](int s) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float* p = s < s1 ? xr1[nb] + 32 * s : xr2[nb] + 32 * (s - s1);
            xa[nb][0] = ld4(p);
            xa[nb][1] = ld4(p + 4);
        }
    };
    f32x4 acc[8][NB];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) {
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (J.bias != nullptr) b4 = ld4(J.bias + m0 + 16 * mb + 4 * q);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = b4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // bias / index loads done: the counts below are exact
    xload(0);
    dma(0);
    if (S > 1) dma(1);
    // Activation scale: ONE power of two per ROW of X (lane (i, q) holds eight k of row 16 nb + i: the four q lanes of a row
    // agree on its maximum through two cross-lane exchanges), 0 = not set yet; the accumulators of a row hold xs x sums.
    // Per row, not per wave: the rows a wave works on may be gathered through per-type lists whose order differs from run to
    // run (atomic appends) -- a scale shared by whichever rows meet in a wave made the result depend on that order (found by
    // tools/s2s_soak.py: 1-ulp differences between runs).
    float xs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) xs[nb] = 0.0f;
    f16x8 xh[NB], xl[NB];
    // X of a step -> range check, scale, split.  Done for step s + 1 at the END of step s (under the matrix pipe's drain) so
    // that the dependent chain max -> compare -> branch -> multiply -> convert is off the step's critical path.
    auto prepare_x = [&]() {
        float m[NB];
        bool redo_lane = false;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v = 0.0f;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                v = fmaxf(fmaxf(v, fmaxf(fabsf(xa[nb][hh][0]), fabsf(xa[nb][hh][1]))), fmaxf(fabsf(xa[nb][hh][2]), fabsf(xa[nb][hh][3])));
            v = fmaxf(v, __shfl_xor(v, 16));
            v = fmaxf(v, __shfl_xor(v, 32));                      // the row's maximum over this step's 32 k, in all four q lanes
            m[nb] = v;
            redo_lane = redo_lane || xs[nb] == 0.0f || v * xs[nb] >= 32768.0f;
        }
        // the first step sets the scales; a row that would reach 2^15 lowers its own (wave-uniform branch, rare)
        if (__builtin_amdgcn_ballot_w64(redo_lane) != 0ull) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const unsigned E = (__float_as_uint(m[nb]) >> 23) & 255u;
                int sh = 140 - (int)E;                            // max (2^(E-127) ..) -> 2^13 ..
                sh = sh > 40 ? 40 : (sh < -40 ? -40 : sh);
                float ns = __int_as_float((127 + sh) << 23);
                float ratio;
                if (xs[nb] == 0.0f) ratio = ns;                   // first time: the accumulators hold the bias
                else if (m[nb] * xs[nb] >= 32768.0f) { ns = ns < xs[nb] ? ns : xs[nb]; ratio = ns / xs[nb]; }    // (powers of two: exact)
                else { ns = xs[nb]; ratio = 1.0f; }
#pragma unroll
                for (int mb = 0; mb < 8; ++mb) acc[mb][nb] = acc[mb][nb] * ratio;
                xs[nb] = ns;
            }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            split8(xa[nb][0] * xs[nb], xa[nb][1] * xs[nb], xh[nb], xl[nb]);    // (natural k order: the halves are 8 q .. + 4, + 4 .. + 8)
    };
    // X(0) has landed (the DMA loads of steps 0 and 1 were issued behind it: vmcnt counts in order)
    if (S > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    prepare_x();
    for (int s = 0; s < S; ++s) {
        // the fragments of this step have landed; the four DMA loads of step s + 1 may be in flight
        if (s + 1 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                        // fragments visible to every wave; slot (s - 1) % NST is free
        if (s + 1 < S) xload(s + 1);                          // X first, then the DMA (see the vmcnt below)
        if (s + 2 < S) dma(s + 2);
        else if (s + 1 < S) asm volatile("" ::: "memory");
        const f16x8* st = ring + (s % GS1_NST) * GS_STAGE + lane;
        const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)st;
        // fragments of row block mb + 1 are requested before the MFMAs of row block mb (LDS returns in order: lgkmcnt(2)
        // leaves exactly the newer two outstanding)
        f16x8 w[2][2];
#define GS_READ(buf, mb)                                                                                          \
        do {                                                                                                      \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w[buf][0]) : "v"(base), "n"(((mb) * 2 + 0) * 1024)); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w[buf][1]) : "v"(base), "n"(((mb) * 2 + 1) * 1024)); \
        } while (0)
        GS_READ(0, 0);
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int cur = mb & 1;
            if (mb < 7) {
                if (cur == 0) GS_READ(1, mb + 1); else GS_READ(0, mb + 1);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w[cur][0]), "+v"(w[cur][1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[cur][0]), "+v"(w[cur][1]));
            }
            const f16x8 wh = w[cur][0], wl = w[cur][1];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[nb], acc[mb][nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[nb], acc[mb][nb], 0, 0, 0);
        }
#undef GS_READ
        if (s + 1 < S) {
            // X of step s + 1 has landed (the DMA of step s + 2, issued behind it, may be in flight)
            if (s + 2 < S) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prepare_x();
        }
    }
    float inv_xs[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) inv_xs[nb] = xs[nb] != 0.0f ? 1.0f / xs[nb] : 1.0f;      // (powers of two: exact)
    const int act = J.act;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int64_t n = n0 + 16 * nb + i;
        if (n >= N) continue;
        const int64_t nsrc = n;
        if (J.yidx != nullptr) n = J.yidx[n];
        const float* g1 = J.g1 != nullptr ? J.g1 + (size_t)J.i1[nsrc] * M : nullptr;
        const float* g2 = J.g2 != nullptr ? J.g2 + (size_t)J.i2[nsrc] * M : nullptr;
        const float sc = J.scale != nullptr ? J.scale[(size_t)n * J.sstride] : 1.0f;
        if (act == 5) {                                       // LSTM cell on the gate pre-activations (rows interleaved by unit)
            const int Ru = M >> 2;
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                const int u = ((m0 + 16 * mb) >> 2) + q;      // rows 4 u .. 4 u + 3
                const f32x4 v = acc[mb][nb] * inv_xs[nb];
                const float ig = sigmoid1(v[0]), fg = sigmoid1(v[1]);
                const float gg = tanh1(v[2]), og = sigmoid1(v[3]);
                const float cn = fg * J.cell_c0[(size_t)n * Ru + u] + ig * gg;
                J.cell_c1[(size_t)n * Ru + u] = cn;
                J.cell_h1[(size_t)n * Ru + u] = og * tanh1(cn);
            }
            continue;
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            const int m = m0 + 16 * mb + 4 * q;
            f32x4 v = acc[mb][nb] * inv_xs[nb];
            if (g1 != nullptr) v += ld4(g1 + m) + ld4(g2 + m);
            if (act == 1) v = silu4(v);
            else if (act == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            } else if (act == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = tanh1(v[r]);
            } else if (act == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu1(v[r]);
            }
            if (J.post_scale != nullptr) v = v * ld4(J.post_scale + m) + ld4(J.post_shift + m);
            v = v * sc;
            if (J.accumulate) v += ld4(Y + (size_t)n * ldy + m);
            st4(Y + (size_t)n * ldy + m, v);
        }
    }
}

// Node side of the local frames in one launch (last field layer + k_s2s_extend + k_s2s_aug_nodes + two k_s2s_pad_rows):
// a wave per node.  With fh2 != nullptr the wave first finishes the field query, field[n] = W4 fh2[n] + b4 (D rows of he
// weights: 64 lanes x he / 64 products each, butterfly sum); then lane 0 builds ext = [inputs | field], rel_feat (7D + O
// columns), Rinv and the zero-padded copies of rel_feat at the row strides the dense layers read (the prior's res1 and the
// first columns of the decoder's wide gate row).  Also clears the per-type edge counters of the step.
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_node_prep(const float* __restrict__ inputs, const float* __restrict__ field_in, const float* __restrict__ fh2,
                const float* __restrict__ w4, const float* __restrict__ b4, int he, float* __restrict__ field_out,
                float* __restrict__ ext, float* __restrict__ rel_feat, float* __restrict__ Rinv, float* __restrict__ relp,
                int ldp, float* __restrict__ wide, int ldwide, int* __restrict__ counts, int64_t n_nodes) {
    using A = AugDims<D>;
    constexpr int RFp = (A::RF + 15) / 16 * 16;
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x < 8 && counts != nullptr) counts[threadIdx.x] = 0;
    if (n >= n_nodes) return;
    float xi[3 * D];
    if (fh2 != nullptr) {
        float part[D];
#pragma unroll
        for (int dd = 0; dd < D; ++dd) part[dd] = 0.0f;
        for (int k = 4 * lane; k < he; k += 256) {
            const f32x4 xv = ld4(fh2 + (size_t)n * he + k);
#pragma unroll
            for (int dd = 0; dd < D; ++dd) {
                const f32x4 wv = ld4(w4 + (size_t)dd * he + k);
                part[dd] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
            }
        }
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) part[dd] += __shfl_xor(part[dd], off);
            xi[2 * D + dd] = part[dd] + b4[dd];
        }
    } else {
#pragma unroll
        for (int t = 0; t < D; ++t) xi[2 * D + t] = field_in[n * D + t];
    }
    if (lane != 0) return;
#pragma unroll
    for (int t = 0; t < 2 * D; ++t) xi[t] = inputs[n * 2 * D + t];
    if (field_out != nullptr) {
#pragma unroll
        for (int t = 0; t < D; ++t) field_out[n * D + t] = xi[2 * D + t];
    }
#pragma unroll
    for (int t = 0; t < 3 * D; ++t) ext[n * 3 * D + t] = xi[t];
    float row[RFp];
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
    for (int t = A::RF; t < RFp; ++t) row[t] = 0.0f;
#pragma unroll
    for (int t = 0; t < A::RF; ++t) rel_feat[n * A::RF + t] = row[t];
#pragma unroll
    for (int t = 0; t < RFp; ++t) relp[n * ldp + t] = row[t];
    if (wide != nullptr) {                            // S2S_RFG columns: a whole 32-wide k block for the split GEMM
#pragma unroll
        for (int t = 0; t < S2S_RFG; ++t) wide[n * ldwide + t] = t < RFp ? row[t < RFp ? t : 0] : 0.0f;
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Rinv[n * D * D + a * D + b] = R[a][b];
}

// Last layer of the decoder's output MLP + Globalizer + residual in one launch (aether.py:649-652): a wave per node,
// pred = W6 o2[n] + b6 (2D rows of hd weights, butterfly sums), outputs = inputs + [R pred_pos | R pred_vel].
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_out_globalize(const float* __restrict__ o2, const float* __restrict__ w6, const float* __restrict__ b6, int hd,
                    const float* __restrict__ inputs, const float* __restrict__ Rinv, float* __restrict__ out,
                    int64_t n_nodes) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_nodes) return;
    float pred[2 * D];
#pragma unroll
    for (int t = 0; t < 2 * D; ++t) pred[t] = 0.0f;
    for (int k = 4 * lane; k < hd; k += 256) {
        const f32x4 xv = ld4(o2 + (size_t)n * hd + k);
#pragma unroll
        for (int t = 0; t < 2 * D; ++t) {
            const f32x4 wv = ld4(w6 + (size_t)t * hd + k);
            pred[t] += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        }
    }
#pragma unroll
    for (int t = 0; t < 2 * D; ++t) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) pred[t] += __shfl_xor(pred[t], off);
        pred[t] += b6[t];
    }
    if (lane != 0) return;
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk)
#pragma unroll
        for (int a = 0; a < D; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < D; ++b) s += Rinv[n * D * D + a * D + b] * pred[chunk * D + b];
            out[n * 2 * D + chunk * D + a] = inputs[n * 2 * D + chunk * D + a] + s;
        }
}

// k_s2s_aug_edges plus a zero-padded copy of the rows (stride EAp, the decoder's present-message layer).
template <int D>
__global__ void __launch_bounds__(256)
k_s2s_edge_prep(const float* __restrict__ x, const int64_t* __restrict__ send, const int64_t* __restrict__ recv,
                const float* __restrict__ rel_feat, int polar, float* __restrict__ edge_attr, float* __restrict__ eap,
                float* __restrict__ edge_pos, int64_t n_edges) {
    using A = AugDims<D>;
    constexpr int EAp = (A::EA + 31) / 32 * 32;
    constexpr int LDR = A::EA + 1;
    __shared__ float rows[256 * LDR];
    __shared__ float prow[256 * (A::EP + 1)];
    const int64_t e0 = (int64_t)blockIdx.x * 256;
    const int64_t e = e0 + threadIdx.x;
    if (e < n_edges) {
        const int64_t j = send[e], i = recv[e];
        float xj[3 * D], xi[3 * D], o[A::NF];
#pragma unroll
        for (int t = 0; t < 3 * D; ++t) { xj[t] = x[j * 3 * D + t]; xi[t] = x[i * 3 * D + t]; }
        aug_edge<D>(xj, xi, o);
        float* out = rows + threadIdx.x * LDR;
#pragma unroll
        for (int t = 0; t < A::NF; ++t) out[t] = o[t];
#pragma unroll
        for (int t = 0; t < A::RF; ++t) out[A::NF + t] = rel_feat[i * A::RF + t];
        const int p0 = polar ? (D == 2 ? 2 : 3) : 0;
#pragma unroll
        for (int t = 0; t < A::EP; ++t) prow[threadIdx.x * (A::EP + 1) + t] = o[p0 + t];
    }
    __syncthreads();
    const int64_t left = n_edges - e0;
    const int cnt = (int)(left < 256 ? left : 256);
    for (int idx = threadIdx.x; idx < cnt * A::EA; idx += 256) {
        const int r = idx / A::EA, c = idx - r * A::EA;
        edge_attr[e0 * A::EA + idx] = rows[r * LDR + c];
    }
    for (int idx = threadIdx.x; idx < cnt * EAp; idx += 256) {
        const int r = idx / EAp, c = idx - r * EAp;
        eap[e0 * EAp + idx] = c < A::EA ? rows[r * LDR + c] : 0.0f;
    }
    for (int idx = threadIdx.x; idx < cnt * A::EP; idx += 256) {
        const int r = idx / A::EP, c = idx - r * A::EP;
        edge_pos[e0 * A::EP + idx] = prow[r * (A::EP + 1) + c];
    }
}

// eaf[e][:] = sum_z planes[z][e][:] for the node's in-edges (every edge has one receiver: each row is produced once) and
// X0[n][:] = sum over the in-edges / fixed_div: k_s2s_sum_planes + k_s2s_segment_mean of the prior step in one pass.
// (Folding the small res1 layer in as well was measured: 12.6 -> 39.8 us, its strided weight reads serialise.)
__global__ void __launch_bounds__(128)
k_s2s_planes_segsum(const float* __restrict__ planes, int n_planes, int64_t plane_stride, const int64_t* __restrict__ order,
                    const int64_t* __restrict__ rowptr, float* __restrict__ eaf, float* __restrict__ X0, int h,
                    float fixed_div) {
    const int64_t n = blockIdx.x;
    const int64_t beg = rowptr[n], end = rowptr[n + 1];
    for (int c = threadIdx.x * 4; c < h; c += 128 * 4) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int64_t k = beg; k < end; ++k) {
            const int64_t e = order[k];
            f32x4 v = ld4(planes + (size_t)e * h + c);
            for (int z = 1; z < n_planes; ++z) v += ld4(planes + (size_t)z * plane_stride + (size_t)e * h + c);
            if (n_planes > 1 || eaf != planes) st4(eaf + (size_t)e * h + c, v);
            s += v;
        }
        st4(X0 + (size_t)n * h + c, s / fixed_div);
    }
}

// Hard Gumbel sample (k_s2s_gumbel_hard) + the per-type edge lists of the decoder (k_s2s_select for every type) in one
// launch.  counts must be zero (k_s2s_node_prep clears them).  (Computing the last prior layer here, an edge per thread,
// was measured: 4.9 + 8 -> 34.5 us.)
__global__ void __launch_bounds__(256)
k_s2s_gumbel_select(const float* __restrict__ logits, const float* __restrict__ uniform, float tau, int K, int k0,
                    float* __restrict__ edges, int64_t* __restrict__ lists /* [K][n_edges] */, int* __restrict__ counts,
                    int64_t n_edges) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < n_edges) {
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
        for (int k = 0; k < K; ++k) {
            w[k] = ((k == best ? 1.0f : 0.0f) - y[k]) + y[k];
            edges[e * K + k] = w[k];
        }
    }
    const int lane = threadIdx.x & 63;
    for (int k = k0; k < K; ++k) {
        const bool on = e < n_edges && w[k] != 0.0f;
        const unsigned long long mask = __ballot(on);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(counts + k, __popcll(mask));
        base = __shfl(base, 0);
        if (on) lists[(size_t)k * n_edges + base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
    }
}

// Up to three narrow dense layers on the same rows in one launch: Y_g[n][m] = act(b_g[m] + sum_{c < C} W_g[m][c] X[n][c]),
// C <= 8 (the variable-N models feed 6 canonical-state columns to three gates / to mlp1: through the matrix-core GEMM
// that took padded copies of X and of every weight -- five launches).  One thread per (row, 4 outputs); act 0 | 4 (ELU).
struct NarrowLayers { const float* W[3]; const float* b[3]; float* Y[3]; int n; };
__global__ void __launch_bounds__(256)
k_s2s_narrow_layers(NarrowLayers L, const float* __restrict__ X, int ldx, int C, int M, int act, int64_t n_rows) {
    const int q4 = M >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_rows * q4) return;
    const int64_t n = idx / q4;
    const int m = (int)(idx - n * q4) * 4;
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = c < C ? X[n * ldx + c] : 0.0f;
    for (int g = 0; g < L.n; ++g) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv = L.b[g] != nullptr ? L.b[g][m + r] : 0.0f;
            for (int c = 0; c < C; ++c) sv = fmaf(L.W[g][(size_t)(m + r) * C + c], x[c], sv);
            v[r] = act == 4 ? elu1(sv) : sv;
        }
        st4(L.Y[g] + (size_t)n * M + m, v);
    }
}

// The variable-N decoder's version (weights given, not sampled): k_s2s_select of every type and the weights divided by
// the number of active types (k_s2s_scale) in one launch.  lists: type k at lists + k * list_stride; counts must be zero.
__global__ void __launch_bounds__(256)
k_s2s_select_all(const float* __restrict__ edge_w, int K, int k0, float norm, float* __restrict__ ewn,
                 int64_t* __restrict__ lists, size_t list_stride, int* __restrict__ counts, int64_t n_edges) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (e < n_edges)
        for (int k = 0; k < K; ++k) {
            w[k] = edge_w[e * K + k];
            ewn[e * K + k] = w[k] * norm;
        }
    const int lane = threadIdx.x & 63;
    for (int k = k0; k < K; ++k) {
        const bool on = e < n_edges && w[k] != 0.0f;
        const unsigned long long mask = __ballot(on);
        int base = 0;
        if (lane == 0 && mask) base = atomicAdd(counts + k, __popcll(mask));
        base = __shfl(base, 0);
        if (on) lists[(size_t)k * list_stride + base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
    }
}

// T[k][e][:] = tanh(A_k[recv[e]][:] + S_k[send[e]][:]) for every type k whose weight on edge e is not zero (rows by edge
// id: the second message layer gathers them through the type's list), and M1[e][:] = M2[e][:] = 0: k_s2s_pair_tanh of
// all types and the two clears of the scatter targets in one launch.  A, S: [K][n_nodes][h]; T: [K][n_edges][h].
__global__ void __launch_bounds__(256)
k_s2s_pair_tanh_all(const float* __restrict__ A, const float* __restrict__ S, int64_t n_nodes,
                    const int64_t* __restrict__ send, const int64_t* __restrict__ recv, const float* __restrict__ edge_w,
                    int K, int k0, float* __restrict__ T, float* __restrict__ M1, float* __restrict__ M2, int h,
                    int64_t n_edges, int clear_all) {
    // clear_all = 0 (the fused seq2seq step: hard samples, every edge in at most ONE type's list, whose job writes its row
    // of M1 / M2 without reading it): only the rows that no type >= k0 will write are cleared -- none without skip_first.
    // Clearing every row cost 2 x n_edges x h x 4 bytes of writes per step (200 of this kernel's 300 MB at 48,640 x 512)
    // and the same again as reads in the accumulating epilogues.
    const int q4 = h >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * q4) return;
    const int64_t e = idx / q4;
    const int c = (int)(idx - e * q4) * 4;
    const int64_t r = recv[e], s = send[e];
    bool written = false;
    for (int k = k0; k < K; ++k) {
        if (edge_w[e * K + k] == 0.0f) continue;
        written = true;
        const f32x4 v = ld4(A + ((size_t)k * n_nodes + r) * h + c) + ld4(S + ((size_t)k * n_nodes + s) * h + c);
        st4(T + ((size_t)k * n_edges + e) * h + c, f32x4{tanh1(v[0]), tanh1(v[1]), tanh1(v[2]), tanh1(v[3])});
    }
    if (clear_all || !written) {
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        st4(M1 + (size_t)e * h + c, zero);
        st4(M2 + (size_t)e * h + c, zero);
    }
}

// Both in-edge means of the decoder in one launch: blockIdx.y = 0: M1 -> out0, 1: M2 -> out1 (row stride ldo each).
__global__ void __launch_bounds__(128)
k_s2s_segment_mean2(const float* __restrict__ M1, const float* __restrict__ M2, const int64_t* __restrict__ order,
                    const int64_t* __restrict__ rowptr, float* __restrict__ out0, float* __restrict__ out1, int ldo, int h) {
    const int64_t n = blockIdx.x;
    const float* Mx = blockIdx.y == 0 ? M1 : M2;
    float* out = blockIdx.y == 0 ? out0 : out1;
    const int64_t beg = rowptr[n], end = rowptr[n + 1];
    for (int c = threadIdx.x * 4; c < h; c += 128 * 4) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int64_t k = beg; k < end; ++k) s += ld4(Mx + (size_t)order[k] * h + c);
        const float cnt = (float)(end - beg > 1 ? end - beg : 1);
        st4(out + (size_t)n * ldo + c, s / cnt);
    }
}

// out[4 u + g] = a[g R + u] + b[g R + u]: the summed LSTM biases with the gates interleaved by unit (S2SJob act 5)
__global__ void __launch_bounds__(256)
k_s2s_lstm_bias_interleave(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int R) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= 4 * R) return;
    const int src = (m & 3) * R + (m >> 2);
    out[m] = a[src] + b[src];
}

// Plan helpers: dst[r][0 .. ld) = [a[r][0..ca) zero-padded to cap | b[r][0..cb) | c[r][0..cc)] (nullptr parts are skipped)
__global__ void __launch_bounds__(256)
k_s2s_concat_rows(const float* __restrict__ a, int ca, int cap, const float* __restrict__ b, int cb,
                  const float* __restrict__ c, int cc, float* __restrict__ dst, int ld, int64_t rows) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * ld) return;
    const int64_t r = idx / ld;
    const int col = (int)(idx - r * ld);
    float v = 0.0f;
    if (col < cap) v = col < ca ? a[r * ca + col] : 0.0f;
    else if (col < cap + cb) v = b != nullptr ? b[r * cb + (col - cap)] : 0.0f;
    else if (col < cap + cb + cc) v = c != nullptr ? c[r * cc + (col - cap - cb)] : 0.0f;
    dst[idx] = v;
}
// dst = a + b
__global__ void __launch_bounds__(256)
k_s2s_add_vec(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ dst, int n) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < n) dst[idx] = a[idx] + b[idx];
}

}  // namespace
