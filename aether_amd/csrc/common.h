// common.h -- device helpers shared by the streamed and the fused kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "../../include/aether_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int H = AETHER_HIDDEN;     // 64
// LDS row strides (floats).  A fragment read is one ds_read_b128 per lane at
// row*(stride) + 16a + 4q, row = lane & 15, q = lane >> 4; its 16-lane groups mix rows of two q
// values, so the 16-byte slot index (stride/4 * row + q) mod 16 must be distinct inside a group:
// stride = 64 + 8 (slot = 2*row + q: even / odd) is conflict-free, 64 + 4 is 2-way in every group
// (measured: 39 % of LDS cycles were conflict cycles with + 4).
constexpr int LDW = H + 8;           // K = 64 weights / activations
constexpr int FPAD = 32;             // layer-1 feature count padded to two 16-wide k blocks
constexpr int LDF = FPAD + 8;        // K = 32 (slot = 10*row + q)
constexpr int LDST = H + 4;          // wave-private tile staging: conflict-free 16-byte row writes
constexpr float PI_F = 3.14159274101257324f;       // float(np.pi)
constexpr float TWO_PI_F = 6.28318548202514648f;   // float(2*np.pi)
constexpr float EPS_F = 1e-7f;                     // nn/utils/geometry.py:62

// ------------------------------------------------------------------ device helpers
// Workgroup barrier that orders LDS traffic only.  __syncthreads() on gfx950 also drains vmcnt (loads
// and stores share one counter), i.e. it waits for every global load in flight: weight prefetches
// issued before a barrier would be exposed at it.  Use where no thread reads, through global memory,
// what another thread of the workgroup wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// torch.nn.SiLU: x * sigmoid(x) = x / (1 + exp(-x)).  v_exp_f32 / v_rcp_f32 are 1-ulp hardware
// ops; the only extra error is the rounding of x*log2(e) (<= |x| * 9e-8 relative in exp(-x)),
// which is below fp32 resolution of the result for the pre-activation range seen here.  Measured
// against the fp64 reference in tests/test_gpu_parity.py (same 1e-5 scale-relative bar).
__device__ __forceinline__ float silu(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.44269504088896340736f);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
// Four values at once: the three plain multiplies / adds are written on 2-wide vectors so that they
// become v_pk_mul_f32 / v_pk_add_f32 (fp32 MFMA shares the VALU: every instruction saved is
// matrix-pipe time).
__device__ __forceinline__ f32x4 silu4(f32x4 v) {
    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const f32x2 tl = lo * -1.44269504088896340736f, th = hi * -1.44269504088896340736f;
    f32x2 el = {__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])};
    f32x2 eh = {__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])};
    el = el + 1.0f;
    eh = eh + 1.0f;
    const f32x2 rl = {__builtin_amdgcn_rcpf(el[0]), __builtin_amdgcn_rcpf(el[1])};
    const f32x2 rh = {__builtin_amdgcn_rcpf(eh[0]), __builtin_amdgcn_rcpf(eh[1])};
    const f32x2 ol = lo * rl, oh = hi * rh;
    return f32x4{ol[0], ol[1], oh[0], oh[1]};
}
// sigmoid of four values (same hardware ops as silu4); silu = x * s, d silu / dx = s * (1 + x * (1 - s))
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    const f32x2 tl = lo * -1.44269504088896340736f, th = hi * -1.44269504088896340736f;
    f32x2 el = {__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])};
    f32x2 eh = {__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])};
    el = el + 1.0f;
    eh = eh + 1.0f;
    return f32x4{__builtin_amdgcn_rcpf(el[0]), __builtin_amdgcn_rcpf(el[1]), __builtin_amdgcn_rcpf(eh[0]),
                 __builtin_amdgcn_rcpf(eh[1])};
}
__device__ __forceinline__ f32x4 dsilu_from_sigmoid(f32x4 x, f32x4 s) { return s * (1.0f + x * (1.0f - s)); }
// ELU (alpha = 1), tanh and the logistic function of the seq2seq family on the hardware's v_exp_f32 / v_rcp_f32 (1 ulp each)
// instead of the device library's expm1f / tanhf / expf + IEEE division (30 - 50 instructions a value: at 64 values a lane
// the ELU epilogue of a 128 x 128 GEMM tile cost more than the tile's 16 k steps -- tools/gemm_split_variants.py, round 4).
// Where 1 - e^t or 1 - 2 / (1 + e^2x) would cancel (|t| < 1/8, |x| < 1/4) a short Taylor polynomial takes over.  In an fp32
// emulation over [-20, 20] (3 M points, exact exp2): ELU 4.2e-7 and tanh 3.7e-7 relative at most (3 - 4 ulp), 1e-7 absolute;
// the logistic function 9e-8 absolute (relative 1e-6 in the far negative tail, where x log2(e) is rounded before the
// exponential and the value is ~2e-9).
__device__ __forceinline__ float elu1(float v) {
    const float t = fminf(v, 0.0f);
    const float e = __builtin_amdgcn_exp2f(t * 1.44269504088896340736f) - 1.0f;
    float p = fmaf(t, 1.0f / 5040.0f, 1.0f / 720.0f);
    p = fmaf(t, p, 1.0f / 120.0f);
    p = fmaf(t, p, 1.0f / 24.0f);
    p = fmaf(t, p, 1.0f / 6.0f);
    p = fmaf(t, p, 0.5f);
    p = fmaf(t * t, p, t);                                       // t + t^2/2 + .. + t^7/5040: |next term| < 2e-9 |t| at -1/8
    const float neg = t > -0.125f ? p : e;
    return v > 0.0f ? v : neg;
}
__device__ __forceinline__ float tanh1(float x) {
    const float ax = fabsf(x), x2 = x * x;
    const float e = __builtin_amdgcn_exp2f(ax * 2.88539008177792681472f);       // e^(2|x|); inf from |x| > 44: rcp -> 0, tanh -> 1
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
    float p = fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f);
    p = fmaf(x2, p, 2.0f / 15.0f);
    p = fmaf(x2, p, -1.0f / 3.0f);
    p = fmaf(x2 * ax, p, ax);                                    // |x| - |x|^3/3 + 2|x|^5/15 - 17|x|^7/315 + 62|x|^9/2835
    const float m = ax < 0.25f ? p : big;
    return __builtin_copysignf(m, x);
}
__device__ __forceinline__ float sigmoid1(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f));
}
// A scalar the compiler has to hold in a register of its own.  Used for broadcast operands of packed fp32 math
// (vector * scalar): left alone, the compiler folds "element 1 of a pair" into an op_sel modifier on src0 / src1 of
// v_pk_{fma,mul,add}_f32, and that form returns a wrong low half in lanes 48-63 when the SIMD's other wave is issuing
// bf16 MFMAs (measured: tools/micro/pkfma_mfma.hip; tools/isa_check.py rule R3 keeps it out of the library).
__device__ __forceinline__ float own_reg(float x) { asm volatile("" : "+v"(x)); return x; }
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

// ------------------------------------------------------------------ fp32 contractions on the matrix pipe: split operands
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and on the vector ALUs: 32 cycles for 2,048 FLOP,
// and every VALU instruction next to it costs matrix time.  The 16-bit MFMAs (v_mfma_f32_16x16x32_{bf16,f16}) do 16,384
// FLOP in 16 cycles on the matrix pipe proper, accumulating in fp32; products of two 16-bit pieces are exact in fp32.
//
// Rounds 1-3: three bf16 pieces per operand, x = hi + mid + lo + r with |r| <= 2^-24 |x|, and SIX terms
// hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid  (the rest <= 2^-22 relative).
//
// Round 4 (every split GEMM of the library: gemm_split / gemm_split_T below, k_wgemm, k_s2s_gemm_split, k_s2s_filter_split):
// two fp16 pieces, x = hi + lo with hi = fp16(x),
// lo = fp16(x - hi) -- 22 significand bits -- and THREE terms  hi*hi + hi*lo + lo*hi  (lo*lo <= 2^-22 relative): half the
// MFMAs and 2.5 instead of 5 vector instructions per split value, for an error of 2^-22 instead of 2^-24 per product
// (parity bar: 1e-5 = 2^-16.6).  fp16 has a narrow exponent range (normal from 2^-14, top 65,504); it is handled so:
//   * weights are split as they are: a weight's lo piece is a subnormal below |w| = 2^-3 (absolute error <= 2^-25 there),
//     which the fp16 MFMA reads exactly; |w| >= 65,504 cannot be represented (outputs come out non-finite, not wrong);
//   * activations / gradients are checked per wave and GEMM: the wave's max |x| (7 DPP steps + a read-lane) decides between
//     the plain path (2^-6 <= max < 2^15: every value's error is <= 2^-22 of the wave's maximum) and a path that multiplies
//     operand and accumulator by an exact power of two first (max -> 2^13..2^14) and the accumulator back after;
//   * the streaming GEMMs (k_wgemm, k_s2s_gemm_split) keep one such scale per wave / workgroup that only shrinks along K.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- two fp16 pieces
constexpr int SPLIT_WIMG = 2 * 4 * 2 * 64 * 4;          // floats of the split image of a 64 x 64 matrix (2 terms x 8 KB)
__device__ __forceinline__ void split_f16x2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)(x - (float)h);
}
// The lane's B fragment of one 32-deep k block from two accumulator-layout blocks: element j < 4 is hidden unit
// 32 kb + 4 q + j, element j >= 4 is 32 kb + 16 + 4 q + (j - 4) -- a permutation of the natural k order; weight images
// (stage_split4) are written in the same order.  On pairs (the library is built without the SLP vectoriser, build.py):
// v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32 -- five instructions per two values.
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 x = p < 2 ? f32x2{v0[2 * p], v0[2 * p + 1]} : f32x2{v1[2 * p - 4], v1[2 * p - 3]};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        const f32x2 r = x - __builtin_convertvector(h, f32x2);
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
        lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
    }
}
// fp16 x 2 weight image: [term 2][row block mb][k block kb][lane 64] fragments of 8 fp16 (16 bytes); lane (m, q) of
// fragment (mb, kb) holds W[16 mb + m][32 kb + 4 q + j] (j < 4) | W[16 mb + m][32 kb + 16 + 4 q + j].
// `v` = W[row][col .. col + 3] (col a multiple of 4): one half-fragment (8 bytes) per term.
template <int MBN, int KBN>
__device__ __forceinline__ void stage_split4(float* img, int row, int col, const f32x4 v) {
    const int mb = row >> 4, m = row & 15, kb = col >> 5, c5 = col & 31, half = c5 >> 4, qq = (c5 & 15) >> 2;
    f16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 a, b;
        split_f16x2(v[j], a, b);
        h[j] = a; l[j] = b;
    }
    const int frag = (mb * KBN + kb) * 64 + m + 16 * qq;
    constexpr int TERM = MBN * KBN * 64;                        // fragments per term
    *reinterpret_cast<f16x4*>(img + (frag) * 4 + half * 2) = h;
    *reinterpret_cast<f16x4*>(img + (TERM + frag) * 4 + half * 2) = l;
}

// Biased fp32 exponent field of the wave's maximum of m (m >= 0 in every lane): row_shr 1, 2, 4, 8 leave a row's maximum in
// its lane 15, row_bcast 15 / 31 carry it on to lane 63 (out-of-range source lanes read 0, the identity); read-lane 63.
__device__ __forceinline__ unsigned wave_max_bits(float m) {     // bit pattern of the wave's maximum (m >= 0 in every lane)
    int v = __float_as_int(m);                                  // non-negative floats order like their bit patterns
#define AETHER_DPP_MAX(CTRL) { const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); v = o > v ? o : v; }
    AETHER_DPP_MAX(0x111) AETHER_DPP_MAX(0x112) AETHER_DPP_MAX(0x114) AETHER_DPP_MAX(0x118) AETHER_DPP_MAX(0x142) AETHER_DPP_MAX(0x143)
#undef AETHER_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_max_exponent(float m) { return (wave_max_bits(m) >> 23) & 255u; }
// What a GEMM's activation operand needs before it is split into fp16 pieces: nothing (biased exponent of the wave's maximum
// in [121, 141], i.e. 2^-6 <= max < 2^15; or 0: all zeros), or a power-of-two scale that puts the maximum into 2^13..2^14
// (exponent clamped to +-40).  Wave-uniform.
struct SplitScale { bool on; float s, inv_s; };
template <int N>
__device__ __forceinline__ SplitScale split_scale_of(const f32x4 (&act)[N]) {
    float m = 0.0f;
#pragma unroll
    for (int b = 0; b < N; ++b) m = fmaxf(fmaxf(m, fmaxf(fabsf(act[b][0]), fabsf(act[b][1]))), fmaxf(fabsf(act[b][2]), fabsf(act[b][3])));
    // The common case costs two compares: no lane at or above 2^15, some lane at or above 2^-6 (or nothing but zeros).  Only
    // the other case needs the maximum itself (the DPP reduction).
    const bool big = __builtin_amdgcn_ballot_w64(m >= 32768.0f) != 0ull;
    const bool some = __builtin_amdgcn_ballot_w64(m >= 0.015625f) != 0ull;
    const bool any = __builtin_amdgcn_ballot_w64(m > 0.0f) != 0ull;
    SplitScale r;
    r.on = big || (any && !some);
    r.s = 1.0f; r.inv_s = 1.0f;
    if (r.on) {
        const unsigned E = wave_max_exponent(m);
        int sh = 140 - (int)E;                                  // max (2^(E-127) ..) -> 2^13 ..
        sh = sh > 40 ? 40 : (sh < -40 ? -40 : sh);
        r.s = __int_as_float((127 + sh) << 23);
        r.inv_s = __int_as_float((127 - sh) << 23);
    }
    return r;
}

// (xh / xl: the operand's pieces, kept for callers that stage them: fused_bwd.h)
template <int MBN, int KBN>
__device__ __forceinline__ void gemm_split_core(const f16x8* __restrict__ w, const f32x4 (&act)[2 * KBN], f32x4 (&acc)[MBN], int lane,
                                                f16x8 (&xh)[KBN], f16x8 (&xl)[KBN]) {
    constexpr int TERM = MBN * KBN * 64;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        split8(act[2 * kb], act[2 * kb + 1], xh[kb], xl[kb]);
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            const int frag = (mb * KBN + kb) * 64 + lane;
            const f16x8 wh = w[frag], wl = w[TERM + frag];
            // small terms first
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[kb], acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[kb], acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[kb], acc[mb], 0, 0, 0);
        }
    }
}
// acc[mb] += W[16 mb + i][k] * act[item][k] over KBN 32-deep k blocks; act = 2 KBN accumulator-layout blocks.
// Returns the scale it applied to the operand (1 on the plain path) and the operand's pieces (of the SCALED values).
template <int MBN, int KBN>
__device__ __forceinline__ SplitScale gemm_split_keep(const float* __restrict__ img, const f32x4 (&act)[2 * KBN], f32x4 (&acc)[MBN],
                                                      int lane, f16x8 (&xh)[KBN], f16x8 (&xl)[KBN]) {
    const SplitScale sc = split_scale_of(act);
    f32x4 x[2 * KBN];
#pragma unroll
    for (int b = 0; b < 2 * KBN; ++b) x[b] = act[b];
    if (sc.on) {                                                // (wave-uniform)
#pragma unroll
        for (int b = 0; b < 2 * KBN; ++b) x[b] = x[b] * sc.s;
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) acc[mb] = acc[mb] * sc.s;
    }
    gemm_split_core<MBN, KBN>(reinterpret_cast<const f16x8*>(img), x, acc, lane, xh, xl);
    if (sc.on) {
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) acc[mb] = acc[mb] * sc.inv_s;
    }
    return sc;
}
template <int MBN, int KBN>
__device__ __forceinline__ void gemm_split(const float* __restrict__ img, const f32x4 (&act)[2 * KBN], f32x4 (&acc)[MBN],
                                           int lane) {
    f16x8 xh[KBN], xl[KBN];
    (void)gemm_split_keep<MBN, KBN>(img, act, acc, lane, xh, xl);
}

// The same product with the weight fragments of ONE 16-row block held in registers (k_fused's node phase: a wave owns a row
// block of W3 / W4 / W_s / W_r and applies it to one or two node tiles): wh / wl = the image's fragments (mb, kb = 0 .. KBN).
template <int KBN>
__device__ __forceinline__ f32x4 gemm_split_regs(const f16x8 (&wh)[KBN], const f16x8 (&wl)[KBN], const f32x4 (&act)[2 * KBN], f32x4 acc) {
    const SplitScale sc = split_scale_of(act);
    f32x4 x[2 * KBN];
#pragma unroll
    for (int b = 0; b < 2 * KBN; ++b) x[b] = act[b];
    if (sc.on) {                                                // (wave-uniform)
#pragma unroll
        for (int b = 0; b < 2 * KBN; ++b) x[b] = x[b] * sc.s;
        acc = acc * sc.s;
    }
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        f16x8 xh, xl;
        split8(x[2 * kb], x[2 * kb + 1], xh, xl);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[kb], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kb], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[kb], xh, acc, 0, 0, 0);
    }
    if (sc.on) acc = acc * sc.inv_s;
    return acc;
}
// the lane's fragments (mb, kb = 0 .. KBN) of both terms from an image of MBN row blocks in global memory
template <int MBN, int KBN>
__device__ __forceinline__ void load_split_frags(const float* __restrict__ img, int mb, int lane, f16x8 (&wh)[KBN], f16x8 (&wl)[KBN]) {
    // wave-uniform row block -> scalar base + one 32-bit lane offset (no 64-bit address pair per fragment)
    const f16x8* w = reinterpret_cast<const f16x8*>(img) + __builtin_amdgcn_readfirstlane(mb) * (KBN * 64);
    constexpr int TERM = MBN * KBN * 64;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        wh[kb] = w[kb * 64 + lane];
        wl[kb] = w[TERM + kb * 64 + lane];
    }
}

// The TRANSPOSED product from the same split image (round 4): acc[ob] += W^T[16 ob + i'][m] * act[item][m], i.e.
// out[k] = sum_m W[m][k] act[m] for the backward's W2^T dpre2 / W_e^T G, without a second (transposed) copy of the weights.
// `img` is the image gemm_split reads (MBN = 4 row blocks of W, KBN k blocks); the contraction now runs over W's ROWS, 32 per
// block (act = 4 accumulator-layout blocks), the outputs are W's columns, OB = 2 KBN blocks of 16.
// The A fragment a lane needs -- 8 values of ONE column from 8 different rows -- lies in 8 different fragments of the image:
// ds_read_b64_tr_b16 gathers it.  Per group of 16 lanes (lanes 16 g .. 16 g + 15; g = this lane's q, which selects rows
// 4 g .. 4 g + 3 of a 16-row block) the instruction reads 4 rows x 16 columns of 16-bit values and returns them column-major:
// lane 4 r + p of the group supplies the address of row r, columns 4 p .. 4 p + 3 (8 bytes: one half-fragment of the image,
// lane slot (4 g + r) + 16 p, half = ob & 1); lane c receives column c of the four rows.  Two reads (row blocks 2 mp and
// 2 mp + 1) make the 8-element fragment in gemm_split's k order.  Addresses of the four p lanes are 256 bytes apart: a
// 4-way bank conflict per read (16 instead of 4 LDS cycles) -- 32 reads per 64 x 64 product.
// EXEC must be all ones (the gather crosses lanes).
// the four transposed reads of block BLK = (mp, ob): rows 32 mp .. of W (two 16-row blocks), columns 16 ob .. of W
template <int KBN, int BLK>
__device__ __forceinline__ void split_T_issue(unsigned base, u32x2 (&lo)[2], u32x2 (&hi)[2]) {
    constexpr int TERM_BYTES = 4 * KBN * 64 * 16, OB = 2 * KBN, mp = BLK / OB, ob = BLK % OB;
#pragma unroll
    for (int t = 0; t < 2; ++t)
        // fragment (mb, kb) of term t starts at ((mb KBN + kb) 64) 16 bytes; kb = ob >> 1, half = ob & 1
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(lo[t]), "=&v"(hi[t])     // early clobber: the first read's data may land before the second one issues
                     : "v"(base), "n"(t * TERM_BYTES + (((2 * mp) * KBN + (ob >> 1)) * 64) * 16 + (ob & 1) * 8),
                       "n"(t * TERM_BYTES + (((2 * mp + 1) * KBN + (ob >> 1)) * 64) * 16 + (ob & 1) * 8)
                     : "memory");
}
template <bool LAST>
__device__ __forceinline__ void split_T_consume(u32x2 (&lo)[2], u32x2 (&hi)[2], const f16x8 xh, const f16x8 xl, f32x4& acc) {
    // LDS returns in order: lgkmcnt(4) leaves exactly the next block's four reads outstanding
    if constexpr (LAST)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
    else
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
    f16x8 w[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const u32x4 v = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
        w[t] = __builtin_bit_cast(f16x8, v);
    }
    const f16x8 wh = w[0], wl = w[1];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc, 0, 0, 0);
}
template <int KBN, int BLK>
__device__ __forceinline__ void split_T_step(unsigned base, u32x2 (&lo)[2][2], u32x2 (&hi)[2][2], const f16x8 (&xh)[2],
                                             const f16x8 (&xl)[2], f32x4 (&acc)[2 * KBN]) {
    constexpr int OB = 2 * KBN, NBLK = 2 * OB, mp = BLK / OB, ob = BLK % OB;
    if constexpr (BLK + 1 < NBLK) split_T_issue<KBN, BLK + 1>(base, lo[(BLK + 1) & 1], hi[(BLK + 1) & 1]);
    split_T_consume<BLK + 1 == NBLK>(lo[BLK & 1], hi[BLK & 1], xh[mp], xl[mp], acc[ob]);
    if constexpr (BLK + 1 < NBLK) split_T_step<KBN, BLK + 1>(base, lo, hi, xh, xl, acc);
}
// (every caller passes zero accumulators: they are not rescaled on the way in).  Returns the operand's scale and pieces.
template <int KBN>
__device__ __forceinline__ SplitScale gemm_split_T_keep(const float* __restrict__ img, const f32x4 (&act)[4], f32x4 (&acc)[2 * KBN],
                                                        int lane, f16x8 (&xh)[2], f16x8 (&xl)[2]) {
    const int g = lane >> 4, r = (lane & 15) >> 2, p = lane & 3;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)img + ((4 * g + r) + 16 * p) * 16;
    const SplitScale sc = split_scale_of(act);
    f32x4 x[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) x[b] = act[b];
    if (sc.on) {                                                // (wave-uniform; gradients are usually far below 2^-6)
#pragma unroll
        for (int b = 0; b < 4; ++b) x[b] = x[b] * sc.s;
    }
    split8(x[0], x[1], xh[0], xl[0]);
    split8(x[2], x[3], xh[1], xl[1]);
    // software pipeline over the blocks (mp, ob): the four reads of block n + 1 are in flight while block n's three MFMAs issue
    u32x2 lo[2][2], hi[2][2];
    split_T_issue<KBN, 0>(base, lo[0], hi[0]);
    split_T_step<KBN, 0>(base, lo, hi, xh, xl, acc);
    if (sc.on) {
#pragma unroll
        for (int ob = 0; ob < 2 * KBN; ++ob) acc[ob] = acc[ob] * sc.inv_s;
    }
    return sc;
}
template <int KBN>
__device__ __forceinline__ void gemm_split_T(const float* __restrict__ img, const f32x4 (&act)[4], f32x4 (&acc)[2 * KBN],
                                             int lane) {
    f16x8 xh[2], xl[2];
    (void)gemm_split_T_keep<KBN>(img, act, acc, lane, xh, xl);
}

// The product on operand pieces that exist already (k_fused keeps a message tile as its pieces: they serve the receiver sums
// of this layer and the first GEMM of the next): sc = the scale the pieces were made with.
template <int MBN, int KBN>
__device__ __forceinline__ void gemm_split_pieces(const float* __restrict__ img, const f16x8 (&xh)[KBN], const f16x8 (&xl)[KBN],
                                                  const SplitScale sc, f32x4 (&acc)[MBN], int lane) {
    constexpr int TERM = MBN * KBN * 64;
    const f16x8* w = reinterpret_cast<const f16x8*>(img);
    if (sc.on) {                                                // (wave-uniform)
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) acc[mb] = acc[mb] * own_reg(sc.s);
    }
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb)
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) {
            const int frag = (mb * KBN + kb) * 64 + lane;
            const f16x8 wh = w[frag], wl = w[TERM + frag];
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[kb], acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[kb], acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[kb], acc[mb], 0, 0, 0);
        }
    if (sc.on) {
#pragma unroll
        for (int mb = 0; mb < MBN; ++mb) acc[mb] = acc[mb] * own_reg(sc.inv_s);
    }
}
// the pieces of an operand tile (accumulator layout, 2 KBN blocks) and the scale they carry
template <int KBN>
__device__ __forceinline__ SplitScale split_tile(const f32x4 (&act)[2 * KBN], f16x8 (&xh)[KBN], f16x8 (&xl)[KBN]) {
    const SplitScale sc = split_scale_of(act);
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        if (sc.on) split8(act[2 * kb] * own_reg(sc.s), act[2 * kb + 1] * own_reg(sc.s), xh[kb], xl[kb]);
        else split8(act[2 * kb], act[2 * kb + 1], xh[kb], xl[kb]);
    }
    return sc;
}

// ------------------------------------------------------------------ staged tiles as fp16 pieces, read transposed (round 4)
// A staged tensor of a tile ([16 rows][64 features]) lies in LDS as the two fp16 pieces its GEMM splits it into anyway:
// [piece 2][16 rows][FB_RH bytes] -- 4,352 bytes, what fp32 rows [16][LDST] take.  Products over the tile's 16 rows read the
// pieces TRANSPOSED (ds_read_b64_tr_b16: lane (c, q) of a 16-lane group gets rows 4 q .. 4 q + 3 of column c) as A / B
// fragments of v_mfma_f32_16x16x32_f16 with K = the 16 rows in k slots 8 q + j, j < 4 (slots j >= 4 are zero).
constexpr int FB_RH = 2 * LDST;                // bytes of a staged row: 64 halves + 4 halves of padding (136)
constexpr int FB_PLANE = 16 * FB_RH;          // bytes of one piece of a tile
static_assert(2 * FB_PLANE == 16 * LDST * 4, "the two pieces take the room of the fp32 rows");
template <int KBN>
__device__ __forceinline__ void fb_stage_pieces(float* arr, int i, int q, const f16x8 (&xh)[KBN], const f16x8 (&xl)[KBN]) {
    char* row = reinterpret_cast<char*>(arr) + i * FB_RH + 8 * q;
#pragma unroll
    for (int kb = 0; kb < KBN; ++kb) {
        const u32x4 h = __builtin_bit_cast(u32x4, xh[kb]), l = __builtin_bit_cast(u32x4, xl[kb]);
        // elements j < 4: features 32 kb + 4 q + j; j >= 4: 32 kb + 16 + 4 q + (j - 4)
        *reinterpret_cast<u32x2*>(row + 64 * kb) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(row + 64 * kb + 32) = u32x2{h[2], h[3]};
        *reinterpret_cast<u32x2*>(row + FB_PLANE + 64 * kb) = u32x2{l[0], l[1]};
        *reinterpret_cast<u32x2*>(row + FB_PLANE + 64 * kb + 32) = u32x2{l[2], l[3]};
    }
}
// the lane's part of a transposed 4 x 16 read: rows 4 q + r, 8-byte column group p (lane = 16 q + 4 r + p)
__device__ __forceinline__ unsigned fb_tr_lane_offset(int lane) {
    const int q = lane >> 4, c = lane & 15;
    return (unsigned)((4 * q + (c >> 2)) * FB_RH + 8 * (c & 3));
}
__device__ __forceinline__ unsigned fb_lds_addr(const float* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ f16x8 fb_frag(const u32x2 v) { return __builtin_bit_cast(f16x8, u32x4{v[0], v[1], 0u, 0u}); }
#define FB_TR(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(dst) : "v"(base), "n"(off) : "memory")
// 0 / 1 fragment of four incidence bits (bit j: k slot j): fp16 1.0 = 0x3C00
__device__ __forceinline__ f16x8 fb_inc_frag(unsigned nib) {
    const unsigned lo = ((nib & 1u) | ((nib & 2u) << 15)) * 0x3C00u, hi = (((nib >> 2) & 1u) | ((nib & 8u) << 13)) * 0x3C00u;
    return __builtin_bit_cast(f16x8, u32x4{lo, hi, 0u, 0u});
}


// Cooperative copy of W[rows][cols] (global, row stride src_ld) into LDS [rows][ldw], zero padded.
__device__ __forceinline__ void stage_weight(float* lds, const float* __restrict__ w, int rows,
                                             int cols, int src_ld, int ldw) {
    for (int idx = threadIdx.x; idx < rows * ldw; idx += blockDim.x) {
        int r = idx / ldw, c = idx - r * ldw;
        lds[idx] = (c < cols) ? w[(size_t)r * src_ld + c] : 0.0f;
    }
}

// Vectorised copy of a [64][64] weight slice (row stride src_ld floats, 16-byte aligned rows) into
// LDS [64][LDW]: all global loads are issued before the first LDS store.
template <int THREADS>
__device__ __forceinline__ void stage_weight64(float* lds, const float* __restrict__ w, int src_ld) {
    constexpr int PER = (H * H / 4 + THREADS - 1) / THREADS;
    f32x4 v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = threadIdx.x + THREADS * j, r = idx >> 4, c = (idx & 15) * 4;
        if (idx < H * H / 4) v[j] = ld4(w + (size_t)r * src_ld + c);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int idx = threadIdx.x + THREADS * j, r = idx >> 4, c = (idx & 15) * 4;
        if (idx < H * H / 4) st4(lds + r * LDW + c, v[j]);
    }
}

// Rollout extras of one step (aether_rollout): `qattr` != nullptr makes the kernels derive
// edge_attr_orig = [q_send * q_recv, |x_send - x_recv|] themselves (the runner's per-batch prep,
// experiments/lorentz/main.py:243-246) instead of reading it; `vel_out` != nullptr also writes the
// next velocity (x_next - x) / dt next to the output; `ext_field` != nullptr replaces the built-in field net
// by a precomputed per-node field [n_nodes][D] (the dynamic-field variant, aether_dynamic_field).
struct StepExtras {
    const float* qattr; float* vel_out; float dt; const float* ext_field; bool skip_e4 = false;
    // training with dropout_prob > 0 (locs.py:160-168: nn.Dropout after the two SiLUs of the out MLP): the scale masks
    // [n_nodes][64] (0 or 1 / (1 - p)) the caller wrote into the workspace, or null; dropword (device): 1 when this forward
    // applied them -- aether_backward reads it
    const float* drop1 = nullptr; const float* drop2 = nullptr; int* dropword = nullptr;
};

template <int D> struct NodeInfo {
    // [p(D) v(D) f(D) R(D*D row-major) cv(D) cf(D)], padded to a multiple of 4 floats
    static constexpr int P = 0, V = D, F = 2 * D, R = 3 * D, CV = 3 * D + D * D, CF = CV + D;
    static constexpr int STRIDE = (D == 2) ? 16 : 24;
};


// Frame R from velocity (geometry.py:7-73): theta in [0, 2pi), phi = acos(clamp(vz/(|v|+eps)));
// canonical velocity / force cv = R^T v, cf = R^T f (aether.py:33-50).
template <int D>
__device__ __forceinline__ void node_frame(const float (&v)[D], const float (&f)[D], float (&R)[D][D],
                                           float (&cv)[D], float (&cf)[D]) {
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
#pragma unroll
    for (int a = 0; a < D; ++a) {                             // R^T v, R^T f
        float sv = 0.f, sf = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) { sv += R[b][a] * v[b]; sf += R[b][a] * f[b]; }
        cv[a] = sv; cf[a] = sf;
    }
}

// Local-frame edge features for edge j -> i (aether.py:52-100, geometry.py:76-101), followed by
// [rel_feat[recv] | edge_attr_orig] (aether.py:99,177); `nj` / `nir` are NodeInfo records.
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


}  // namespace
