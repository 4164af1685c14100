// Anisotropic edge filter of the seq2seq / variable-N prior step (nn/nn/anisotropic_filter.py:34-40), third version:
//   out[e][c] = sum_r ea[e][r] * (b2[r h + c] + sum_k L2[r h + c][k] hw[e][k]),   hw[e][k] = act(W1[k] . pos[e] + b1[k])
// evaluated in the reference's own order -- Z_r = L2_r hw first, then the weighted sum over the R features -- which
// keeps the matrix-core operand the same for every r: the hidden row hw[e][:] is split ONCE per k slab into pieces held in
// registers, the weights come as a prepared image (k_s2s_filter_images, once per weight version).
//
// Round 4: fp16 x 2 pieces, three MFMA terms per product (rounds 2-3: bf16 x 3 pieces, six terms).  x = hi + lo with
// hi = fp16(x), lo = fp16(x - hi) carries 22 significand bits; hi.hi + hi.lo + lo.hi drops only lo.lo (2^-22 relative, the
// six bf16 terms dropped 2^-24): still two orders inside the 1e-5 parity bar, for HALF the matrix-pipe work and 2/3 of the
// image bytes -- this kernel is matrix-pipe bound (0.69 busy in round 2).  fp16's narrow exponent range is handled with
// exact power-of-two scales: the weight image is scaled by f_w (max |L2| -> 2^13..2^14, one number per image, in the image's
// trailer), every hidden row by its own f_e (k_s2s_filter_bimg: row maximum -> 2^13..2^14), so both pieces of every value
// within 2^-18 of its row's / tensor's maximum are normal fp16 numbers; the product of the inverse scales multiplies the
// edge feature ea[e][r] before it weights Z_r (four multiplies per step).  Smaller values lose relative, not absolute,
// precision (fp16 subnormals: 2^-24 of the scaled maximum).
// The first version (seq2seq.h, k_s2s_filter) formed x[(r, k)] = ea[e][r] * hw[e][k] on the fly for the fp32 MFMA.
//
// Workgroup = 8 waves = 2 (c halves) x 4 (edge quarters) on a 64 (c) x 256 (edges) output tile, two waves per SIMD.  The
// k range of a unit is walked in slabs of 64; a wave owns 32 c x 64 edges and holds the B fragments of its edges for BOTH
// 32-wide k blocks of the slab (64 edges x 64 k x 2 pieces = 64 registers), loaded once per slab from the prepared image
// of the hidden rows, so Z_r accumulates over the whole slab before it is weighted.  Per feature r the workgroup needs the
// weight fragments of 64 c x 64 k (16 KB, contiguous in the image): they arrive by LDS-DMA two iterations ahead in a
// three-slot ring (48 KB), one workgroup barrier per r.  A wave runs 48 MFMAs per r (first term of each tile on a zero
// accumulator) and adds out += ea[:, r] * Z_r (32 FMAs per lane).
// LDS: ring 48 KB + feature values R x 1 KB + bias slab R x 256 B.
// Units = (256-edge tile, 64-wide c block, split z): split z covers h / splits consecutive k (a multiple of 64) and writes
// plane z of `out` (bias in plane 0); k_s2s_sum_planes adds the planes in order (deterministic).  Persistent workgroups,
// one per CU; the workgroups of an XCD walk the same (c block, split) pairs side by side, one edge tile each.
#pragma once
#include "common.h"

namespace {

constexpr int FILT_STAGE = 2 * 4 * 2 * 64;      // f16x8 fragments of one step: 2 k blocks x 4 row blocks x 2 pieces (16 KB)
constexpr int FILT_NST = 3;                     // ring slots: a step is requested FILT_NST iterations before it is multiplied
constexpr int FILT_TRAILER = 64;                // floats behind the image's fragments: [0] = max |L2| (scale derived from it)

__host__ __device__ constexpr size_t filt_lds_bytes(int R) {
    return (size_t)(FILT_NST * FILT_STAGE) * 16 + (size_t)R * 256 * 4 + (size_t)R * 64 * 4;
}
// fragments (16 bytes each) of a weight image / of the hidden rows' image; the hidden rows' buffer carries one float per
// (padded) edge behind its fragments: 1 / f_e
__host__ __device__ constexpr size_t filt_image_frags(int R, int h) { return (size_t)R * h * (h >> 3) * 2; }
__host__ __device__ constexpr size_t filt_image_bytes(int R, int h) { return filt_image_frags(R, h) * 16 + FILT_TRAILER * 4; }
__host__ __device__ constexpr size_t filt_bimg_frags(int64_t E, int h) { return (size_t)((E + 15) / 16 * 16) * (h >> 3) * 2; }
__host__ __device__ constexpr size_t filt_bimg_bytes(int64_t E, int h) { return filt_bimg_frags(E, h) * 16 + (size_t)((E + 15) / 16 * 16) * 4; }

// The power of two f with |x| f <= 2^14 for every |x| <= maxabs (fp16 tops out at 65,504); exponent clamped to +-40 (a
// tensor of 1e-30s needs no help), 1 for a zero / non-finite maximum.
__device__ __forceinline__ float fp16_scale_for(float maxabs) {
    if (!(maxabs > 0.0f) || !(maxabs < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(maxabs, &e);                  // maxabs = m 2^e, m in [0.5, 1)
    int s = 14 - e;
    s = s > 40 ? 40 : (s < -40 ? -40 : s);
    return ldexpf(1.0f, s);
}

// max |x| of a tensor in two launches (weight preparation: once per weight version; no atomics, nothing to initialise):
// FILT_TRAILER - 1 workgroups leave their maxima in trailer[1 ..], one wave folds them into trailer[0]
__global__ void __launch_bounds__(1024)
k_s2s_absmax(const float* __restrict__ x, int64_t n4 /* float4 count */, float* __restrict__ trailer) {
    float m = 0.0f;
    for (int64_t idx = (int64_t)blockIdx.x * 1024 + threadIdx.x; idx < n4; idx += (int64_t)(FILT_TRAILER - 1) * 1024) {
        const f32x4 v = ld4(x + 4 * idx);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    __shared__ float red[1024];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) trailer[1 + blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(64)
k_s2s_absmax_fold(float* __restrict__ trailer) {
    float m = threadIdx.x >= 1 && threadIdx.x < FILT_TRAILER ? trailer[threadIdx.x] : 0.0f;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if (threadIdx.x == 0) trailer[0] = m;
}

// image[((r (h/32) + a32) (h/16) + mb) 2 + term][lane (i, q)] = pieces of f_w L2[r h + 16 mb + i][32 a32 + 8 q .. + 8);
// f_w from the trailer's maximum (k_s2s_absmax, launched in front of this kernel)
__global__ void __launch_bounds__(256)
k_s2s_filter_images(const float* __restrict__ L2w, int R, int h, f16x8* __restrict__ img) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (row, k octet)
    const int oct = h >> 3;
    if (idx >= (int64_t)R * h * oct) return;
    const float fw = fp16_scale_for(reinterpret_cast<const float*>(img + filt_image_frags(R, h))[0]);
    const int64_t row = idx / oct;
    const int o = (int)(idx - row * oct);
    const int r = (int)(row / h), c = (int)(row - (int64_t)r * h);
    const f32x4 v0 = ld4(L2w + row * h + 8 * o), v1 = ld4(L2w + row * h + 8 * o + 4);
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 a, b;
        split_f16x2((j < 4 ? v0[j] : v1[j - 4]) * fw, a, b);
        hi[j] = a; lo[j] = b;
    }
    const int a32 = o >> 2, q = o & 3, mb = c >> 4, i = c & 15;
    const size_t frag = (((size_t)r * (h >> 5) + a32) * (h >> 4) + mb) * 2;
    img[(frag + 0) * 64 + i + 16 * q] = hi;
    img[(frag + 1) * 64 + i + 16 * q] = lo;
}
// both launches of a weight image's preparation
inline void filter_images_launch(const float* L2w, int R, int h, void* image, hipStream_t st) {
    f16x8* img = reinterpret_cast<f16x8*>(image);
    float* trailer = reinterpret_cast<float*>(img + filt_image_frags(R, h));
    k_s2s_absmax<<<dim3(FILT_TRAILER - 1), dim3(1024), 0, st>>>(L2w, (int64_t)R * h * h / 4, trailer);
    k_s2s_absmax_fold<<<dim3(1), dim3(64), 0, st>>>(trailer);
    const int64_t items = (int64_t)R * h * (h / 8);
    k_s2s_filter_images<<<dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st>>>(L2w, R, h, img);
}

// B operand of the filter GEMM: the hidden rows of the hyper-network, hw[e][k] = act(W1[k] . pos[e] + b1[k]), scaled by the
// row's f_e, split into fp16 pieces and laid out as MFMA fragments: bimg[((e / 16) (h / 32) + a32) 2 + term][lane (i, q)] =
// pieces of f_e hw[16 (e / 16) + i][32 a32 + 8 q .. + 8); 1 / f_e behind the fragments.  One workgroup per block of 16 edges
// (blockIdx.x): thread (og, i) walks the k octets og, og + 16, .. of edge i twice -- the row maximum first, then the pieces
// (the values are 6 FMAs each: cheaper to recompute than to keep).  Edges past the end repeat the last one.
template <int P>
__device__ __forceinline__ void filter_bimg_body(const float* __restrict__ pos, const float* __restrict__ W1,
                                                 const float* __restrict__ b1, int relu, int h, int64_t n_edges,
                                                 f16x8* __restrict__ bimg) {
    __shared__ float smax[16][17];
    const int oct = h >> 3;
    const int64_t blk = blockIdx.x;
    const int i = threadIdx.x & 15, og = threadIdx.x >> 4;
    int64_t e = blk * 16 + i;
    e = e < n_edges ? e : n_edges - 1;
    float pe[P];
#pragma unroll
    for (int p = 0; p < P; ++p) pe[p] = pos[e * P + p];
    auto value = [&](int k) {
        float sv = b1[k];
#pragma unroll
        for (int p = 0; p < P; ++p) sv = fmaf(W1[k * P + p], pe[p], sv);
        return relu ? fmaxf(sv, 0.0f) : elu1(sv);
    };
    // the thread's values: octets og, og + 16, .. (h <= 1024: at most 8 of them stay in registers; beyond, recomputed)
    constexpr int KEEP = 8;
    float val[KEEP][8];
    float mx = 0.0f;
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const int o = og + 16 * it;
        if (o < oct) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { val[it][u] = value(8 * o + u); mx = fmaxf(mx, fabsf(val[it][u])); }
        }
    }
    for (int o = og + 16 * KEEP; o < oct; o += 16)
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, fabsf(value(8 * o + u)));
    smax[og][i] = mx;
    __syncthreads();
    mx = smax[0][i];
#pragma unroll
    for (int g = 1; g < 16; ++g) mx = fmaxf(mx, smax[g][i]);
    const float fe = fp16_scale_for(mx);
    float* einv = reinterpret_cast<float*>(bimg + filt_bimg_frags(n_edges, h));
    if (og == 0) einv[blk * 16 + i] = 1.0f / fe;
    auto emit = [&](int o, const float (&v)[8]) {
        f16x8 hi, lo;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            _Float16 a, b;
            split_f16x2(v[u] * fe, a, b);
            hi[u] = a; lo[u] = b;
        }
        const int a32 = o >> 2, q = o & 3;
        f16x8* dst = bimg + ((blk * (h >> 5) + a32) * 2) * 64 + i + 16 * q;
        dst[0] = hi;
        dst[64] = lo;
    };
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        const int o = og + 16 * it;
        if (o < oct) emit(o, val[it]);
    }
    for (int o = og + 16 * KEEP; o < oct; o += 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = value(8 * o + u);
        emit(o, v);
    }
}
inline dim3 filter_bimg_grid(int64_t E, int n_types = 1) { return dim3((unsigned)((E + 15) / 16), (unsigned)n_types); }
template <int P>
__global__ void __launch_bounds__(256)
k_s2s_filter_bimg(const float* __restrict__ pos, const float* __restrict__ W1, const float* __restrict__ b1, int relu,
                  int h, int64_t n_edges, f16x8* __restrict__ bimg) {
    filter_bimg_body<P>(pos, W1, b1, relu, h, n_edges, bimg);
}

// Several filters over the same edges in one launch (the variable-N decoder has one per edge type): blockIdx.y picks the
// filter's pointers; the work of a filter is exactly that of its own launch.
constexpr int FILT_MAX_TYPES = 4;
struct FilterTypes {
    const f16x8* img[FILT_MAX_TYPES];       // prepared images of the filter banks
    const float* b2[FILT_MAX_TYPES];
    const float* w0[FILT_MAX_TYPES];        // first hyper-network layer (k_s2s_filter_bimg)
    const float* b0[FILT_MAX_TYPES];
    f16x8* bimg[FILT_MAX_TYPES];            // B-operand images, one buffer per filter
    float* out[FILT_MAX_TYPES];             // [splits][n_edges][h] planes (or the result itself when splits == 1)
};
template <int P>
__global__ void __launch_bounds__(256)
k_s2s_filter_bimg_types(const float* __restrict__ pos, FilterTypes T, int relu, int h, int64_t n_edges) {
    const int t = blockIdx.y;
    filter_bimg_body<P>(pos, T.w0[t], T.b0[t], relu, h, n_edges, T.bimg[t]);
}

// s_waitcnt vmcnt(4 n): all but the youngest n steps' LDS-DMA loads (four per step and wave of the early group) have landed
__device__ __forceinline__ void filt_wait_steps(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
}
static_assert(FILT_NST >= 3 && FILT_NST <= 6, "filt_wait_steps covers up to five steps in flight");

// The nine LDS reads of a step of the filter GEMM -- the feature values, then the [mb][kb][term] fragments of the wave's c half
// (fragment 8 kb + 2 mb + term, 1 KiB each, from the half's base) -- and the wait that closes them.  Issue order = the order
// the MFMAs consume them.  Early-clobber outputs (isa_check R5: a destination must not double as a later read's address).
// Free functions: clang rejects inline asm on captured variables inside the kernel's generic lambdas.
#define FILT_RD(KB, MB, T) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(w[KB][MB][T]) : "v"(base), "n"((8 * KB + 2 * MB + T) * 1024))
__device__ __forceinline__ void filt_issue_reads(unsigned base, unsigned eaddr, f32x4& e4, f16x8 (&w)[2][2][2]) {
    asm volatile("ds_read_b128 %0, %1" : "=&v"(e4) : "v"(eaddr));
    FILT_RD(0, 0, 0); FILT_RD(0, 0, 1); FILT_RD(1, 0, 0); FILT_RD(1, 0, 1);
    FILT_RD(0, 1, 0); FILT_RD(0, 1, 1); FILT_RD(1, 1, 0); FILT_RD(1, 1, 1);
}
#undef FILT_RD
__device__ __forceinline__ void filt_reads_done(f32x4& e4, f16x8 (&w)[2][2][2]) {      // every LDS read issued so far has landed
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(e4), "+v"(w[0][0][0]), "+v"(w[0][0][1]), "+v"(w[0][1][0]), "+v"(w[0][1][1]), "+v"(w[1][0][0]),
                   "+v"(w[1][0][1]), "+v"(w[1][1][0]), "+v"(w[1][1][1]));
}

template <int R>
__device__ __forceinline__ void filter_split_body(const f16x8* __restrict__ img, const float* __restrict__ b2,
                                                  const float* __restrict__ ea, const f16x8* __restrict__ bimg,
                                                  float* __restrict__ out, int h, int64_t n_edges, int splits, int rs, int n_wgs) {
    // n_wgs = workgroups along x, as an explicit argument: the kernels below consume NO implicit (hidden) kernel argument
    // -- gridDim.x would be hidden_block_count_x.
    // rs > 1 (few edges, variable-N steps): the R features are divided over rs units as well -- plane z = zr * splits + zk
    // holds the k range zk of features [zr R / rs, (zr + 1) R / rs); at 200 edges a unit otherwise walks all 15 features
    // alone (15 iterations of 48 MFMAs per wave on 16 of the 256 CUs)
    extern __shared__ __attribute__((aligned(16))) unsigned char filt_smem[];
    f16x8* ring = reinterpret_cast<f16x8*>(filt_smem);                          // [slot NST][kb 2][mb 4][term 2][lane]
    float* evs = reinterpret_cast<float*>(ring + FILT_NST * FILT_STAGE);        // [r][eq 4][i 16][nb 4]
    float* b2s = evs + R * 256;                                                 // [r][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int chalf = wave >> 2, eq = wave & 3;
    const bool late = chalf != 0;                              // the SIMD's second wave: multiplies first (see the step loop)
    const int n_eb = (int)((n_edges + 255) >> 8), n_cb = h >> 6;
    const int n_pairs = n_cb * splits * rs;
    const int nr = R / rs;                                     // features of a unit
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, n_slots = (n_wgs + 7 - xcd) >> 3;
    const int my_pairs = (n_pairs - xcd + 7) >> 3;             // pairs xcd, xcd + 8, ..
    const int n_a32 = h >> 5, n_mb = h >> 4;
    const int slabs = (h / splits) >> 6;                       // 64-wide k slabs of a unit
    const int IT = slabs * nr;                                 // iterations of a unit: (slab, r)
    const int64_t n_eb16 = (n_edges + 15) >> 4;
    // image strides in fragments of 64 lanes: k block to k block, one r step, and the jump from (slab, R - 1) to (slab + 1, 0)
    const int64_t kb_stride = (int64_t)n_mb * 2 * 64;
    const int64_t r_stride = (int64_t)n_a32 * kb_stride;
    const int64_t slab_jump = 2 * kb_stride - (int64_t)(nr - 1) * r_stride;
    const float fw = fp16_scale_for(reinterpret_cast<const float*>(img + filt_image_frags(R, h))[0]);
    const float inv_fw = 1.0f / fw;
    const float* einv = reinterpret_cast<const float*>(bimg + filt_bimg_frags(n_edges, h));

    for (int unit = slot; unit < my_pairs * n_eb; unit += n_slots) {
        const int pair = xcd + 8 * (unit / n_eb);
        const int64_t e0 = (int64_t)(unit % n_eb) * 256;
        const int c0 = (pair % n_cb) * 64, z = pair / n_cb;
        const int zk = z % splits, r_lo = (z / splits) * nr;
        const int kbase = zk * (h / splits);

        // DMA cursor: step (slab 0, r 0); the four waves of the EARLY group (below) move the step's 16 fragments, wave eq the
        // fragments eq, eq + 4, eq + 8, eq + 12 (fragment f = 8 kb + 2 mb + term sits at kb * kb_stride + (2 mb + term) * 64)
        const f16x8* dsrc = img + ((size_t)(kbase >> 5) * n_mb + (c0 >> 4)) * 2 * 64 + lane + (int64_t)r_lo * r_stride;
        int dr = 0, dslot = 0, dleft = IT;
        auto dma_next = [&]() {                                // the next step -> its slot; advances the cursor
            if (dleft <= 0) return;
            f16x8* dst = ring + dslot * FILT_STAGE;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int fr = eq + 4 * f, kb = f >> 1, rest = fr - 8 * kb;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(dsrc + kb * kb_stride + rest * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + fr * 64), 16, 0, 0);
            }
            --dleft;
            dslot = dslot == FILT_NST - 1 ? 0 : dslot + 1;
            if (++dr == nr) { dr = 0; dsrc += slab_jump; } else dsrc += r_stride;
        };
        // unit prologue: everything it reads from global memory in a few rounds of loads -- a batch of a thread's loads is
        // issued before the first value is used (as a plain loop the compiler waits for every load in turn: 20 dependent
        // round trips, ~30 us per unit, as much as the unit's 39 GEMM steps at 2,560 edges -- found in round 4's PMC pass)
        f32x4 esc4;                                            // 1 / (f_e f_w) of this lane's four edges (16 nb + i of the wave's 64)
        {
            constexpr int NJ = (R * 256 + 511) / 512, NB2 = (R * 64 + 511) / 512, BATCH = 10;
            // an opaque copy of the thread index: what is derived from it below is recomputed per unit instead of being hoisted
            // out of the unit loop and kept (= spilled) across the GEMM steps
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int64_t n = e0 + 64 * eq + 16 * nb + i;
                n = n < n_edges ? n : n_edges - 1;
                esc4[nb] = einv[n];
            }
            float bv[NB2];
            if (zk == 0) {
#pragma unroll
                for (int j = 0; j < NB2; ++j) {
                    const int idx = tid + 512 * j;
                    bv[j] = idx < R * 64 ? b2[(size_t)(idx >> 6) * h + c0 + (idx & 63)] : 0.0f;
                }
            }
#pragma unroll
            for (int j0 = 0; j0 < NJ; j0 += BATCH) {           // feature values of the tile's edges (coalesced reads)
                float ev[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const int idx = tid + 512 * (j0 + j), nl = idx / R;
                    int64_t n = e0 + nl;
                    n = n < n_edges ? n : n_edges - 1;
                    ev[j] = (j0 + j < NJ && idx < R * 256) ? ea[(size_t)n * R + (idx - nl * R)] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < BATCH; ++j) {
                    const int idx = tid + 512 * (j0 + j), nl = idx / R, r = idx - nl * R;
                    if (j0 + j < NJ && idx < R * 256) evs[r * 256 + (nl & 192) + 4 * (nl & 15) + ((nl >> 4) & 3)] = ev[j];
                }
            }
            if (zk == 0) {
#pragma unroll
                for (int j = 0; j < NB2; ++j) {
                    const int idx = tid + 512 * j;
                    if (idx < R * 64) b2s[idx] = bv[j];
                }
            }
            esc4 = esc4 * inv_fw;
        }

        f32x4 outv[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) outv[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* evl = evs + 64 * eq + 4 * i;
        __syncthreads();                                       // staged values visible, every global load of the prologue done
        if (!late) { dma_next(); dma_next(); }
        if (zk == 0) {                                         // bias term: sum_r ea[e][r] * b2[r h + c]
            for (int r = r_lo; r < r_lo + nr; ++r) {
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(evl + r * 256);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(b2s + r * 64 + 32 * chalf + 16 * mb + 4 * q);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) outv[mb][nb] += bv * own_reg(e4[nb]);
                }
            }
        }

        f16x8 xh[2][4], xl[2][4];                              // B fragments [k block][nb]: edges 16 nb + i, k = 32 kb + 8 q + u
        auto build_b = [&](int slab) {                         // 16 coalesced 1 KB loads, once per slab
            const int a32 = (kbase + 64 * slab) >> 5;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int64_t blk = (e0 >> 4) + 4 * eq + nb;
                blk = blk < n_eb16 ? blk : n_eb16 - 1;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f16x8* src = bimg + ((blk * n_a32 + a32 + kb) * 2) * 64 + lane;
                    xh[kb][nb] = src[0]; xl[kb][nb] = src[64];
                }
            }
            // consume the loads HERE: left pending, the compiler's wait for them lands in front of the first MFMA of the
            // iteration as vmcnt(0) -- behind the DMA just issued for a later step, whose latency it then exposes
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+v"(xh[kb][nb]), "+v"(xl[kb][nb]));
        };
        // Software pipeline over the steps (round 4): the fragment reads of step it + 1 are issued BEFORE the MFMAs of step it
        // (two register sets), so the matrix pipe does not stand while eight waves pull 72 KB out of LDS behind every barrier.
        // Step s: LDS-DMA issued in iteration s - NST, awaited at the top of iteration s - 1, read into registers during
        // iteration s - 1, multiplied in iteration s.  Slot s % NST is free again once every wave's reads of it have completed:
        // lgkmcnt(0) in front of the barrier of iteration s, behind which step s + NST is requested into it.  (NST = 6 measured:
        // no faster -- the ring does not wait for memory.)
        // Fragment reads as inline assembly: the compiler waits for EVERY outstanding LDS-DMA (vmcnt(0)) before an LDS read it
        // can see -- it cannot tell the ring slots apart.  [kb][mb][term] of this wave's c half: fragment 8 kb + 2 (2 chalf + mb)
        // + term.  Early-clobber outputs (isa_check R5: a destination must not double as a later read's address).
        f16x8 w0[2][2][2], w1[2][2][2];                        // two register sets of [kb][mb][term]
        f32x4 e40, e41;
        // The two waves of a SIMD (wave w and w + 4: the two c halves) run the step in OPPOSITE order.  In lock step -- a
        // barrier per step -- both did their bookkeeping (waits, LDS-DMA, fragment reads, the 32 weighting FMAs: ~1,200 cycles
        // with the matrix pipe idle), then both their 48 MFMAs (2 x 768 cycles): 2,700 cycles per step, measured with the
        // timing-only variants of tools/filter_variants.py (no MFMAs: 86 of 177 us).  Now the EARLY group (c half 0) does the
        // step's bookkeeping first -- it also moves the whole workgroup's LDS-DMA -- and multiplies last; the LATE group
        // multiplies first: one wave's MFMAs run under the other's bookkeeping.  (The LDS-DMA split evenly over both groups,
        // the late one requesting after its MFMAs, with a fourth ring slot: measured slower, 0.482 against 0.474 ms per step.)
        auto issue_reads = [&](f16x8 (&wn)[2][2][2], f32x4& en, int slot_n, int r_n) {
            const f16x8* stp = ring + slot_n * FILT_STAGE + (4 * chalf) * 64 + lane;
            filt_issue_reads((unsigned)(size_t)(__attribute__((address_space(3))) const char*)stp,
                             (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(evl + r_n * 256), en, wn);
        };
        auto multiply = [&](f16x8 (&wc)[2][2][2], const f32x4 ec) {
            const f32x4 es = ec * esc4;                        // ea[:, r] with the operands' scales undone (powers of two: exact)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {                   // Z_r rows 16 mb .. + 16 of this wave over the slab's 64 k
                f32x4 tmp[4];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f16x8 wh = wc[kb][mb][0], wl = wc[kb][mb][1];
                    // small terms first; four independent accumulators per term
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[kb][nb], kb == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[kb][nb], tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[kb][nb], tmp[nb], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) outv[mb][nb] += tmp[nb] * own_reg(es[nb]);       // out += ea[:, r] * Z_r
            }
        };
        int cur_r = r_lo, slot_n = 0;                          // feature of step it; ring slot of step it + 1 (running counters)
        auto step_body = [&](auto late_tag, f16x8 (&wc)[2][2][2], f32x4& ec, f16x8 (&wn)[2][2][2], f32x4& en, int it) {
            constexpr bool late = decltype(late_tag)::value;   // (shadows the wave's flag: one loop per group below)
            if (cur_r == r_lo) build_b(it / nr);
            // step it + 1's fragments have landed (the early group's own loads; the steps behind it may still be in flight) ...
            if (!late) {
                if constexpr (FILT_NST == 3) {
                    if (it + 2 < IT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else {
                    const int behind = IT - 2 - it;
                    filt_wait_steps(behind < 0 ? 0 : (behind < FILT_NST - 2 ? behind : FILT_NST - 2));
                }
            }
            filt_reads_done(ec, wc);                           // ... and this wave's reads of step it (slot it % NST) are complete
            lds_barrier();                                     // ... for every wave
            const int r_n = cur_r + 1 == r_lo + nr ? r_lo : cur_r + 1;
            slot_n = slot_n == FILT_NST - 1 ? 0 : slot_n + 1;
            const bool more = it + 1 < IT;
            if (!late) {
                dma_next();                                    // step it + NST -> slot it % NST
                if (more) issue_reads(wn, en, slot_n, r_n);
                multiply(wc, ec);
            } else {
                multiply(wc, ec);
                if (more) issue_reads(wn, en, slot_n, r_n);
            }
            cur_r = r_n;
        };
        if (!late) {
#pragma unroll
            for (int j = 2; j < FILT_NST; ++j) dma_next();     // (steps 0 and 1 were requested above: NST in flight)
            filt_wait_steps(IT - 1 < FILT_NST - 1 ? IT - 1 : FILT_NST - 1);   // step 0 has landed; the later ones may be in flight
        }
        lds_barrier();                                         // step 0 has landed for every wave
        issue_reads(w0, e40, 0, r_lo);
        if (!late) {
            for (int it = 0; it < IT; it += 2) {
                step_body(std::false_type{}, w0, e40, w1, e41, it);
                if (it + 1 < IT) step_body(std::false_type{}, w1, e41, w0, e40, it + 1);
            }
        } else {
            for (int it = 0; it < IT; it += 2) {
                step_body(std::true_type{}, w0, e40, w1, e41, it);
                if (it + 1 < IT) step_body(std::true_type{}, w1, e41, w0, e40, it + 1);
            }
        }
        float* dst = out + (size_t)z * n_edges * h;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int64_t n = e0 + 64 * eq + 16 * nb + i;
            if (n >= n_edges) continue;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) st4(dst + (size_t)n * h + c0 + 32 * chalf + 16 * mb + 4 * q, outv[mb][nb]);
        }
        __syncthreads();                                       // ring and staged values are free for the next unit; stores issued
    }
}
template <int R>
__global__ void __launch_bounds__(512)             // two waves per SIMD: <= 256 registers each, no AGPR allocation
k_s2s_filter_split(const f16x8* __restrict__ img, const float* __restrict__ b2, const float* __restrict__ ea,
                   const f16x8* __restrict__ bimg, float* __restrict__ out, int h, int64_t n_edges, int splits, int rs,
                   int n_wgs /* = gridDim.x */) {
    filter_split_body<R>(img, b2, ea, bimg, out, h, n_edges, splits, rs, n_wgs);
}
template <int R>
__global__ void __launch_bounds__(512)
k_s2s_filter_split_types(FilterTypes T, const float* __restrict__ ea, int h, int64_t n_edges, int splits, int rs,
                         int n_wgs /* = gridDim.x */) {
    const int t = blockIdx.y;
    filter_split_body<R>(T.img[t], T.b2[t], ea, T.bimg[t], T.out[t], h, n_edges, splits, rs, n_wgs);
}

// The variable-N decoder's edge messages from the present state (aether_dynamicvars.py:827-835) out of the filters' planes:
//   M[e][c] = sum_k w[e][k] relu(sum_z planes_k[z][e][c]),  k = k0 .. K-1 in order, planes in order
// (k_s2s_sum_planes + k_s2s_relu_scale_acc of every type, same order of additions, one launch; M is written, not added to).
__global__ void __launch_bounds__(256)
k_dyn_filter_combine(FilterTypes T, int n_types, int n_planes, const float* __restrict__ w /* [n_edges][K], column k0 + t */,
                     int K, int k0, float* __restrict__ M, int h, int64_t n_edges) {
    const int q4 = h >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_edges * q4) return;
    const int64_t e = idx / q4;
    const int c = (int)(idx - e * q4) * 4;
    const size_t count = (size_t)n_edges * h, at = (size_t)e * h + c;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < n_types; ++t) {
        f32x4 v = ld4(T.out[t] + at);
        for (int z = 1; z < n_planes; ++z) v += ld4(T.out[t] + (size_t)z * count + at);
        const float we = w[e * K + k0 + t];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f) * we;
        acc = acc + v;
    }
    st4(M + at, acc);
}

}  // namespace
