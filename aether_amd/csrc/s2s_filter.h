// Anisotropic edge filter of the seq2seq / variable-N prior step (nn/nn/anisotropic_filter.py:34-40), second version:
//   out[e][c] = sum_r ea[e][r] * (b2[r h + c] + sum_k L2[r h + c][k] hw[e][k]),   hw[e][k] = act(W1[k] . pos[e] + b1[k])
// evaluated in the reference's own order -- Z_r = L2_r hw first, then the weighted sum over the R features -- which
// keeps the matrix-core operand the same for every r: the hidden row hw[e][:] is split ONCE per k slab into three bf16
// pieces held in registers, the weights come as a prepared bf16 x 3 image (k_s2s_filter_images, once per weight version),
// and every fp32 product runs as six v_mfma_f32_16x16x32_bf16 terms (common.h: same error level as the fp32 MFMA, 2.6 x
// its rate).  The first version (seq2seq.h, k_s2s_filter) formed x[(r, k)] = ea[e][r] * hw[e][k] on the fly for the fp32
// MFMA: it sits at 0.7-0.8 of THAT roof (472 us at 2,560 edges x 39 features, 4.9 ms at 48,640 x 24).
//
// Workgroup = 8 waves = 2 (c halves) x 4 (edge quarters) on a 64 (c) x 256 (edges) output tile, two waves per SIMD.  The
// k range of a unit is walked in slabs of 64; a wave owns 32 c x 64 edges and holds the B fragments of its edges for BOTH
// 32-wide k blocks of the slab (64 edges x 64 k x 3 pieces = 96 registers), loaded once per slab from the prepared image
// of the hidden rows (k_s2s_filter_bimg: the first hyper-network layer, split; it replaces k_s2s_pos_hidden), so Z_r
// accumulates over the whole slab before it is weighted.  Per feature r the workgroup needs the weight fragments of
// 64 c x 64 k (24 KB, contiguous in the image): they arrive by LDS-DMA two iterations ahead in a three-slot ring (72 KB),
// one workgroup barrier per r.  A wave runs 96 MFMAs per r (first term of each tile on a zero accumulator) and adds
// out += ea[:, r] * Z_r (32 FMAs per lane).  Measured at 48,640 edges x 24 features (612 GFLOP): 2.45 ms = 250 TFLOP/s
// fp32-equivalent, 1.5 PFLOP/s of bf16 MFMA work; the fp32-MFMA version took 4.9 ms.  PMC: the matrix pipe is busy 0.69 of
// the cycles at the 2.08 GHz the chip holds under this load (DESIGN.md 4.8a has the split of the other 31 %).
// LDS: ring 72 KB + feature values R x 1 KB + bias slab R x 256 B.
// Units = (256-edge tile, 64-wide c block, split z): split z covers h / splits consecutive k (a multiple of 64) and writes
// plane z of `out` (bias in plane 0); k_s2s_sum_planes adds the planes in order (deterministic).  Persistent workgroups,
// one per CU; the workgroups of an XCD walk the same (c block, split) pairs side by side, one edge tile each.
#pragma once
#include "common.h"

namespace {

constexpr int FILT_STAGE = 2 * 4 * 3 * 64;      // bf16x8 fragments of one step: 2 k blocks x 4 row blocks x 3 pieces (24 KB)
constexpr int FILT_NST = 3;

__host__ __device__ constexpr size_t filt_lds_bytes(int R) {
    return (size_t)(FILT_NST * FILT_STAGE) * 16 + (size_t)R * 256 * 4 + (size_t)R * 64 * 4;
}

// image[((r (h/32) + a32) (h/16) + mb) 3 + term][lane (i, q)] = pieces of L2[r h + 16 mb + i][32 a32 + 8 q .. + 8)
__global__ void __launch_bounds__(256)
k_s2s_filter_images(const float* __restrict__ L2w, int R, int h, bf16x8* __restrict__ img) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;         // (row, k octet)
    const int oct = h >> 3;
    if (idx >= (int64_t)R * h * oct) return;
    const int64_t row = idx / oct;
    const int o = (int)(idx - row * oct);
    const int r = (int)(row / h), c = (int)(row - (int64_t)r * h);
    const f32x4 v0 = ld4(L2w + row * h + 8 * o), v1 = ld4(L2w + row * h + 8 * o + 4);
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 a, b, d;
        split_bf16x3(j < 4 ? v0[j] : v1[j - 4], a, b, d);
        hi[j] = a; mid[j] = b; lo[j] = d;
    }
    const int a32 = o >> 2, q = o & 3, mb = c >> 4, i = c & 15;
    const size_t frag = (((size_t)r * (h >> 5) + a32) * (h >> 4) + mb) * 3;
    img[(frag + 0) * 64 + i + 16 * q] = hi;
    img[(frag + 1) * 64 + i + 16 * q] = mid;
    img[(frag + 2) * 64 + i + 16 * q] = lo;
}

// B operand of the filter GEMM: the hidden rows of the hyper-network, hw[e][k] = act(W1[k] . pos[e] + b1[k]), split into
// bf16 pieces and laid out as MFMA fragments: bimg[((e / 16) (h / 32) + a32) 3 + term][lane (i, q)] = pieces of
// hw[16 (e / 16) + i][32 a32 + 8 q .. + 8).  One thread per (edge, k octet); edges past the end repeat the last one.
template <int P>
__device__ __forceinline__ void filter_bimg_body(const float* __restrict__ pos, const float* __restrict__ W1,
                                                 const float* __restrict__ b1, int relu, int h, int64_t n_edges,
                                                 bf16x8* __restrict__ bimg) {
    const int oct = h >> 3;
    const int64_t padded = (n_edges + 15) & ~(int64_t)15;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= padded * oct) return;
    // consecutive threads: the 16 edges of a block, then the octets -> 256-byte runs of the image
    const int64_t blk = idx / (16 * oct);
    const int rem = (int)(idx - blk * 16 * oct), o = rem >> 4, i = rem & 15;
    int64_t e = blk * 16 + i;
    e = e < n_edges ? e : n_edges - 1;
    float pe[P];
#pragma unroll
    for (int p = 0; p < P; ++p) pe[p] = pos[e * P + p];
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = 8 * o + u;
        float sv = b1[k];
#pragma unroll
        for (int p = 0; p < P; ++p) sv = fmaf(W1[k * P + p], pe[p], sv);
        sv = sv > 0.0f ? sv : (relu ? 0.0f : expm1f(sv));
        __bf16 ph, pm, pl;
        split_bf16x3(sv, ph, pm, pl);
        hi[u] = ph; mid[u] = pm; lo[u] = pl;
    }
    const int a32 = o >> 2, q = o & 3;
    bf16x8* dst = bimg + ((blk * (h >> 5) + a32) * 3) * 64 + i + 16 * q;
    dst[0] = hi;
    dst[64] = mid;
    dst[128] = lo;
}
template <int P>
__global__ void __launch_bounds__(256)
k_s2s_filter_bimg(const float* __restrict__ pos, const float* __restrict__ W1, const float* __restrict__ b1, int relu,
                  int h, int64_t n_edges, bf16x8* __restrict__ bimg) {
    filter_bimg_body<P>(pos, W1, b1, relu, h, n_edges, bimg);
}

// Several filters over the same edges in one launch (the variable-N decoder has one per edge type): blockIdx.y picks the
// filter's pointers; the work of a filter is exactly that of its own launch.
constexpr int FILT_MAX_TYPES = 4;
struct FilterTypes {
    const bf16x8* img[FILT_MAX_TYPES];      // prepared images of the filter banks
    const float* b2[FILT_MAX_TYPES];
    const float* w0[FILT_MAX_TYPES];        // first hyper-network layer (k_s2s_filter_bimg)
    const float* b0[FILT_MAX_TYPES];
    bf16x8* bimg[FILT_MAX_TYPES];           // B-operand images, one buffer per filter
    float* out[FILT_MAX_TYPES];             // [splits][n_edges][h] planes (or the result itself when splits == 1)
};
template <int P>
__global__ void __launch_bounds__(256)
k_s2s_filter_bimg_types(const float* __restrict__ pos, FilterTypes T, int relu, int h, int64_t n_edges) {
    const int t = blockIdx.y;
    filter_bimg_body<P>(pos, T.w0[t], T.b0[t], relu, h, n_edges, T.bimg[t]);
}

template <int R>
__device__ __forceinline__ void filter_split_body(const bf16x8* __restrict__ img, const float* __restrict__ b2,
                                                  const float* __restrict__ ea, const bf16x8* __restrict__ bimg,
                                                  float* __restrict__ out, int h, int64_t n_edges, int splits, int rs, int n_wgs) {
    // n_wgs = workgroups along x, as an explicit argument: the kernels below consume NO implicit (hidden) kernel argument
    // -- gridDim.x would be hidden_block_count_x.
    // rs > 1 (few edges, variable-N steps): the R features are divided over rs units as well -- plane z = zr * splits + zk
    // holds the k range zk of features [zr R / rs, (zr + 1) R / rs); at 200 edges a unit otherwise walks all 15 features
    // alone (15 iterations of 96 MFMAs per wave on 16 of the 256 CUs: 35 us of a 0.43 ms step)
    extern __shared__ __attribute__((aligned(16))) unsigned char filt_smem[];
    bf16x8* ring = reinterpret_cast<bf16x8*>(filt_smem);                        // [slot 3][kb 2][mb 4][term 3][lane]
    float* evs = reinterpret_cast<float*>(ring + FILT_NST * FILT_STAGE);        // [r][eq 4][i 16][nb 4]
    float* b2s = evs + R * 256;                                                 // [r][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int chalf = wave >> 2, eq = wave & 3;
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
    const int64_t kb_stride = (int64_t)n_mb * 3 * 64;
    const int64_t r_stride = (int64_t)n_a32 * kb_stride;
    const int64_t slab_jump = 2 * kb_stride - (int64_t)(nr - 1) * r_stride;

    for (int unit = slot; unit < my_pairs * n_eb; unit += n_slots) {
        const int pair = xcd + 8 * (unit / n_eb);
        const int64_t e0 = (int64_t)(unit % n_eb) * 256;
        const int c0 = (pair % n_cb) * 64, z = pair / n_cb;
        const int zk = z % splits, r_lo = (z / splits) * nr;
        const int kbase = zk * (h / splits);

        // DMA cursor: step (slab 0, r 0); the wave moves fragments wave, wave + 8, wave + 16 of the step's 24
        // (fragment f = 12 kb + 3 mb + term sits at kb * kb_stride + (3 mb + term) * 64 of the image)
        const bf16x8* dsrc = img + ((size_t)(kbase >> 5) * n_mb + (c0 >> 4)) * 3 * 64 + lane + (int64_t)r_lo * r_stride;
        int dr = 0, dslot = 0, dleft = IT;
        auto dma_next = [&]() {                                // the next step -> its slot; advances the cursor
            if (dleft <= 0) return;
            bf16x8* dst = ring + dslot * FILT_STAGE;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int fr = wave + 8 * f, kb = fr >= 12 ? 1 : 0, rest = fr - 12 * kb;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(dsrc + kb * kb_stride + rest * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + fr * 64), 16, 0, 0);
            }
            --dleft;
            dslot = dslot == FILT_NST - 1 ? 0 : dslot + 1;
            if (++dr == nr) { dr = 0; dsrc += slab_jump; } else dsrc += r_stride;
        };
        // unit prologue: everything it reads from global memory in one round of loads
        for (int idx = tid; idx < R * 256; idx += 512) {       // feature values of the tile's edges (coalesced reads)
            const int nl = idx / R, r = idx - nl * R;
            int64_t n = e0 + nl;
            n = n < n_edges ? n : n_edges - 1;
            evs[r * 256 + (nl & 192) + 4 * (nl & 15) + ((nl >> 4) & 3)] = ea[(size_t)n * R + r];
        }
        if (zk == 0)
            for (int idx = tid; idx < R * 64; idx += 512) b2s[idx] = b2[(size_t)(idx >> 6) * h + c0 + (idx & 63)];

        f32x4 outv[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) outv[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* evl = evs + 64 * eq + 4 * i;
        __syncthreads();                                       // staged values visible, every global load of the prologue done
        dma_next();
        dma_next();
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

        bf16x8 xh[2][4], xm[2][4], xl[2][4];                   // B fragments [k block][nb]: edges 16 nb + i, k = 32 kb + 8 q + u
        auto build_b = [&](int slab) {                         // 24 coalesced 1 KB loads, once per slab
            const int a32 = (kbase + 64 * slab) >> 5;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int64_t blk = (e0 >> 4) + 4 * eq + nb;
                blk = blk < n_eb16 ? blk : n_eb16 - 1;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8* src = bimg + ((blk * n_a32 + a32 + kb) * 3) * 64 + lane;
                    xh[kb][nb] = src[0]; xm[kb][nb] = src[64]; xl[kb][nb] = src[128];
                }
            }
            // consume the loads HERE: left pending, the compiler's wait for them lands in front of the first MFMA of the
            // iteration as vmcnt(0) -- behind the DMA just issued for a later step, whose latency it then exposes
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) asm volatile("" : "+v"(xh[kb][nb]), "+v"(xm[kb][nb]), "+v"(xl[kb][nb]));
        };
        for (int it = 0; it < IT; ++it) {
            const int slab = it / nr, r = r_lo + it - slab * nr;
            if (r == r_lo) build_b(slab);
            // this step's fragments have landed (the three loads of step it + 1 may still be in flight) ...
            if (it + 1 < IT) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();                                     // ... for every wave; slot (it - 1) % NST is free
            dma_next();                                        // step it + 2
            // Fragment reads as inline assembly: the compiler waits for EVERY outstanding LDS-DMA (vmcnt(0)) before an LDS
            // read it can see -- it cannot tell the ring slots apart -- which would expose the latency of the step just
            // requested.  [kb][mb][term] of this wave's c half: fragment 12 kb + 3 (2 chalf + mb) + term
            bf16x8 w[2][2][3];
            f32x4 e4;
            {
                const bf16x8* st = ring + (it % FILT_NST) * FILT_STAGE + (6 * chalf) * 64 + lane;
                const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)st;
                const unsigned eaddr = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(evl + r * 256);
                // issue order = consumption order (e4, then (mb, kb) groups of three); LDS returns in order, so
                // lgkmcnt(9 - 3 g) in front of group g leaves exactly the later groups outstanding
                asm volatile("ds_read_b128 %0, %1" : "=v"(e4) : "v"(eaddr));
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[kb][mb][t]) : "v"(base), "n"((12 * kb + 3 * mb + t) * 1024));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {                   // Z_r rows 16 mb .. + 16 of this wave over the slab's 64 k
                f32x4 tmp[4];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (mb == 0 && kb == 0) asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(w[0][0][0]), "+v"(w[0][0][1]), "+v"(w[0][0][2]), "+v"(e4));
                    else if (mb == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(w[1][0][0]), "+v"(w[1][0][1]), "+v"(w[1][0][2]));
                    else if (kb == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w[0][1][0]), "+v"(w[0][1][1]), "+v"(w[0][1][2]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[1][1][0]), "+v"(w[1][1][1]), "+v"(w[1][1][2]));
                    const bf16x8 wh = w[kb][mb][0], wm = w[kb][mb][1], wl = w[kb][mb][2];
                    // small terms first; four independent accumulators per term
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm[kb][nb], kb == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh[kb][nb], tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[kb][nb], tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh[kb][nb], tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm[kb][nb], tmp[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tmp[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[kb][nb], tmp[nb], 0, 0, 0);
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) outv[mb][nb] += tmp[nb] * own_reg(e4[nb]);       // out += ea[:, r] * Z_r
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
k_s2s_filter_split(const bf16x8* __restrict__ img, const float* __restrict__ b2, const float* __restrict__ ea,
                   const bf16x8* __restrict__ bimg, float* __restrict__ out, int h, int64_t n_edges, int splits, int rs,
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
