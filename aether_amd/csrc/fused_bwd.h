// fused_bwd.h -- parameter gradients of the Aether step for groups of small graphs: the GNN part of the backward
// (four layers of node-update / edge-MLP / gather backward, locs.py:227-243 differentiated) in ONE launch per step,
// one workgroup per fused-forward workgroup (same FusedWG descriptors, same local edge order, fused.h).
//
// What stays on chip, against the layer-by-layer kernels of backward.h (kb_node / kb_edge / kb_gather + k_outer):
//   * G = dL/dpre1, h, dL/dpre2 of an edge tile never leave the owning wave: the four GEMMs of the tile
//     (recompute pre1 / pre2, back through W2 and W_e) chain in registers, and the two edge-level weight-gradient
//     products of the layer (dW2 += dpre2 (x) h, dW_e += G (x) e_prev; layer 1: dW1 += G (x) features) are
//     accumulated in the wave's registers over its tiles (the [E,64] row tensors of backward.h: 560 MB per step at
//     cfg2) -- one partial per workgroup and layer goes to memory, k_fb_reduce adds them in workgroup order;
//   * the gradient with respect to a message tile (W_e^T G of layer l+1) stays in the registers of the wave that owns
//     the tile, as the forward keeps the messages;
//   * sums of G over a node's in-edges (dP_r) and out-edges (dP_s) run on the matrix core against the tile's 0/1
//     incidence matrices (built from the sender / receiver slots), accumulated over the wave's tiles, added over the
//     waves in wave order through LDS: fixed order, no atomics, bit-stable;
//   * node-level work (update MLP backward, dx_{l-1} = dn + W_s^T dP_s + W_r^T dP_r) splits its output rows over the
//     four waves; its row tensors (2,560 nodes) still go to memory for the node-level weight-gradient products, which
//     stay with k_outer (they are ~5 % of the bytes the edge-level ones were).
// Split mode (two workgroups per graph): dP_s of a node needs the G rows of its out-edges in BOTH workgroups.  A
// workgroup walks its tiles in reverse local order -- the tiles with the partner's senders first -- and every wave
// publishes its partial sums for the partner's nodes as soon as it is through with them (sc1 stores, drained, one
// agent-scope add per wave on the workgroup's counter: MI355X_MICROARCH.md, hand-offs measured with sc1 loads, row 3);
// the partner picks them up after its own edge phase, one tile or more later.
//
// 4 waves per workgroup, one per SIMD, up to 512 registers each: the two 64 x 64 accumulators (128 registers), the
// incidence sums (64) and the carried message gradients (16 per tile) do not fit beside a second wave, and the fp32
// MFMA shares its SIMD's vector ALUs: a second wave would not overlap with it anyway.
#pragma once
#include "common.h"
#include "fused.h"
#include "backward.h"

namespace {

constexpr int FB_THREADS = 256;
constexpr int FB_SA = LDST;                   // staging row stride (floats): 16-byte row writes of a tile are conflict-free
constexpr int FB_PART = 2 * H * H + 2 * H;    // floats per (workgroup, layer) partial: dW2 | dW_e (dW1) | db2 | db1
constexpr int FB_MAX_WGS = 1024;              // workgroups whose partials the workspace holds (beyond: backward.h path)

struct FbLds {                                // offsets in floats
    // Weight images in MFMA-fragment order: [row block mb][k block a][lane 64] x 16 bytes -- what lane (i, q) reads for
    // (mb, a) is W[16 mb + i][16 a + 4 q .. + 3].  One fragment = 1 KiB contiguous = one LDS-DMA wave instruction with
    // per-lane source addresses (fb_stage_frags), and the fragment reads are lane-linear: no padding, no conflicts.
    // W_e and W2 -- the two recompute GEMMs of a tile -- are the forward's split (3 x bf16) images, copied as they lie
    // in the workspace (k_prepare_weights; common.h gemm_split): 16 KB each.
    static constexpr int WE = 0;                                   // W_e   split image (layer 1: W1, K = 32: half of it)
    static constexpr int W2 = WE + SPLIT_WIMG;                     // W2    split image
    static constexpr int W2T = W2 + SPLIT_WIMG;                    // G staging of the tile rounds (rounds 1-3: W2^T, W_e^T fragments)
    static constexpr int WEND = 2 * 4 * FUSED_MAX_NODES * LDST;    // the region of images + G staging also holds the dumps below
    static_assert(WEND >= W2T + 4 * 16 * FB_SA, "images + G staging inside the dump region");
    static_assert(2 * 4 * FUSED_MAX_NODES * LDST <= WEND, "per-wave dumps alias the weight images");
    static constexpr int BIAS = WEND;                              // [64] b2 | [64] b1 (layer 1)
    static constexpr int PSB = BIAS + 2 * H;                       // [32][LDW]  P_s (visible slots) -> total dP_s
    static constexpr int PRB = PSB + FUSED_MAX_NODES * LDW;        // [32][LDW]  P_r (own slots)     -> total dP_r
    static constexpr int DNS = PRB + FUSED_MAX_NODES * LDW;        // [32][LDW]  dn  (own slots)
    static constexpr int INVD = DNS + FUSED_MAX_NODES * LDW;       // [32]       1 / max(in-degree, 1)
    static constexpr int STG = INVD + FUSED_MAX_NODES;             // [4 waves][3][16][FB_SA] tile staging
    static constexpr int STG_SIZE = 4 * 3 * 16 * FB_SA;
    static constexpr int DXS = STG;                                // [32][LDW]  dx  (own slots)   } node phase only:
    static constexpr int DPU = DXS + FUSED_MAX_NODES * LDW;        // [32][LDU]  dpre_u            } alias the staging
    static constexpr int SCL = STG + STG_SIZE;                      // [4 waves][4]: 1/(s_d2 s_h), 1/(s_G s_ep), 1/s_d2, 1/s_G of the round's tiles
    static constexpr int TOTAL = SCL + 16;
    // round 4: G of a tile gets staging rows of its own (d2, h, e_prev and G of the round's four tiles stay staged until every
    // wave has added its row quarter of both products): [4 waves][16][FB_SA] in what used to hold W2^T (no longer staged)
    static constexpr int STG_G = W2T;
    static_assert(4 * 16 * FB_SA <= WEND - W2T, "G staging fits the former transposed-image region");
    // per-wave dumps of the incidence sums alias the (then dead) weight images: [4 waves][32][LDST] each
    static constexpr int DUMP_S = 0;
    static constexpr int DUMP_R = 4 * FUSED_MAX_NODES * LDST;
    static_assert(DPU + FUSED_MAX_NODES * LDU <= TOTAL, "node-phase buffers alias the staging");
    static_assert(DUMP_R + 4 * FUSED_MAX_NODES * LDST <= WEND, "dumps alias the weight images");
    static_assert(TOTAL * 4 <= 160 * 1024, "LDS budget");
};

// Everything layer l of the backward reads and writes: one contiguous record, fetched once at the start of the layer
// (the kernel-argument segment is memory: pointers indexed by a run-time layer cost a scalar-cache round trip EACH
// when they are fetched where they are used).
struct FbLayer {
    const float* e_prev;       // messages of layer l-1, receiver-sorted rows (layer 1: the edge features [E][FPAD])
    const float* n;            // n_l = x_{l-1} (res) + mean, saved by the forward
    const float* ps; const float* pr;     // P_s, P_r of layer l (layers 2-4)
    const float* msg_w0;       // layer 1: W1 [64][F1]; layers 2-4: [64][192] = W_s | W_r | W_e
    const float* msg_w2; const float* msg_b2;
    const float* upd_w0; const float* upd_b0;            // W3 [128][64], b3
    const float* w4t; const float* w3t; const float* w2t; const float* w0t;     // transposed copies (k_transpose)
    const float* img_e; const float* img_2;  // the forward's split images of W_e (layer 1: W1) and W2 (fused.h, fused_wimg_offset)
    float* U; float* DPU; float* DPS; float* DPR;        // row tensors of the node-level weight-gradient products
    float* DXout;              // dL/dx_{l-1} (layers 2-4)
};

struct FbArgs {
    // graph view
    const int32_t* rowptr; const int32_t* send_s; const int32_t* recv_s; const FusedWG* wgdesc;
    const int32_t* lorder; const int4* ledge; const int32_t* nrange;
    FbLayer layer[4];
    const float* msg_b0_1;     // layer 1 only (layers 2-4: b1 sits in P_r)
    int f1;
    const float* dx4;          // in: dL/dx_4 (kb_out)
    float* DN1;                // dL/dn_1: layer_1.res and the field backward
    float* DA;                 // [E][FPAD]: dL/d(layer-1 edge features)
    float* DE;                 // [E][64] scratch: message gradients between layers when they do not fit registers (ROUNDS > 3)
    float* partial;            // [n_wgs][4 layers][FB_PART]
    float* xchg;               // split mode: [3 layers][n_wgs][4 waves][32][64] partial dP_s rows
    int* flags;                // split mode: per workgroup, waves that have published (4 per layer)
    int* errword;
    int n_wgs;
    float* stamps;             // diagnostic build only: [n_wgs][FUSED_STAMPS] microseconds since entry
};

// acc[mb][nb] += sum_k X[k][4 i' + mb] * Y[k][NB i + nb] over the 16 staged rows k (i' = MFMA row = 4q + r).
template <int NB>
__device__ __forceinline__ void fb_outer16(const float* __restrict__ sa, const float* __restrict__ sb,
                                           f32x4 (&acc)[4][NB], int i, int q) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 av = ld4(sa + (4 * s4 + q) * FB_SA + 4 * i);
        float bv[NB];
        if constexpr (NB == 4) {
            const f32x4 b4 = ld4(sb + (4 * s4 + q) * FB_SA + 4 * i);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bv[nb] = b4[nb];
        } else {
            const f32x2 b2 = *reinterpret_cast<const f32x2*>(sb + (4 * s4 + q) * FB_SA + 2 * i);
            bv[0] = b2[0]; bv[1] = b2[1];
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma16(av[mb], bv[nb], acc[mb][nb]);
    }
}

// Row-partitioned form (round 4): the wave with index w owns the rows m = 4 i' + w of the product (a quarter, interleaved)
// for EVERY tile of the workgroup -- acc[nb] += sum_k X[k][4 i' + w] * Y[k][NB i + nb] -- so the four waves' accumulators are
// disjoint pieces of the workgroup's sum: no cross-wave reduction, 32 accumulator registers instead of 128.
// `colsum` (optional): += sum_k X[k][4 i' + w] in every column -- the bias gradient of the rows this wave owns, as one more
// MFMA per k step against a column of ones (no per-lane sums, no cross-lane reduction at the end of the layer).
template <int NB, bool COLSUM>
__device__ __forceinline__ void fb_outer16_q(const float* __restrict__ sa, const float* __restrict__ sb, f32x4 (&acc)[NB],
                                             f32x4& colsum, int i, int q, int w) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const float av = sa[(4 * s4 + q) * FB_SA + 4 * i + w];
        if constexpr (COLSUM) colsum = mfma16(av, 1.0f, colsum);
        float bv[NB];
        if constexpr (NB == 4) {
            const f32x4 b4 = ld4(sb + (4 * s4 + q) * FB_SA + 4 * i);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bv[nb] = b4[nb];
        } else {
            const f32x2 b2 = *reinterpret_cast<const f32x2*>(sb + (4 * s4 + q) * FB_SA + 2 * i);
            bv[0] = b2[0]; bv[1] = b2[1];
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma16(av, bv[nb], acc[nb]);
    }
}

// ------------------------------------------------------------------ round 4 (second half): the staged tiles as fp16 pieces
// A staged tensor of a tile ([16 edges][64 features]) lies in LDS as the two fp16 pieces the tile's GEMMs split it into
// anyway (gemm_split_keep / gemm_split_T_keep: pieces of the SCALED values, the wave's power-of-two scale beside them):
// [piece 2][16 rows][FB_RH halves] -- 4,352 bytes, exactly what the fp32 rows [16][FB_SA] took.  The weight-gradient and
// incidence products read them TRANSPOSED (ds_read_b64_tr_b16: lane (c, q) of a 16-lane group gets rows 4 q .. 4 q + 3 of
// column c) as A / B fragments of v_mfma_f32_16x16x32_f16 with K = the tile's 16 edges in k slots 8 q + j, j < 4 (slots
// j >= 4 are zero), three terms per product (two where one operand is a 0/1 matrix), and add tmp x 1 / (s_x s_y) to the
// fp32 accumulators: ~1/3 of the matrix time of the fp32 MFMA form (16 x 32 cycles per 16 x 64 product and tile), which
// ran on the vector ALU.
// (FB_RH, FB_PLANE, fb_stage_pieces, fb_tr_lane_offset, fb_lds_addr, fb_frag, FB_TR, fb_inc_frag: common.h -- fused.h uses them too)
// acc[nb] += c_xy * sum_k X[k][m0 + 4 q + r] Y[k][16 nb + i] over the 16 staged rows; colsum += c_x * sum_k X[k][m0 + 4 q + r]
// xa / ya: LDS byte addresses of the arrays' hi planes + the lane's offset (xa also + 2 m0).  In two halves -- the reads, and
// the wait + MFMAs -- so that the caller can have the next product's reads in flight: LATER = reads issued after this
// product's (LDS returns in order).
template <int NB> struct FbOuterRegs { u32x2 xr[2], yr[NB][2]; };
template <int NB>
__device__ __forceinline__ void fb_outer16_issue(unsigned xa, unsigned ya, FbOuterRegs<NB>& R) {
    FB_TR(R.xr[0], xa, 0); FB_TR(R.xr[1], xa, FB_PLANE);
    FB_TR(R.yr[0][0], ya, 0); FB_TR(R.yr[0][1], ya, FB_PLANE);
    FB_TR(R.yr[1][0], ya, 32); FB_TR(R.yr[1][1], ya, FB_PLANE + 32);
    if constexpr (NB == 4) {
        FB_TR(R.yr[2][0], ya, 64); FB_TR(R.yr[2][1], ya, FB_PLANE + 64);
        FB_TR(R.yr[3][0], ya, 96); FB_TR(R.yr[3][1], ya, FB_PLANE + 96);
    }
}
template <int NB, bool COLSUM, int LATER>
__device__ __forceinline__ void fb_outer16_consume(FbOuterRegs<NB>& R, f32x4 (&acc)[NB], f32x4& colsum, float c_xy, float c_x) {
    static_assert(LATER == 0 || LATER == 6 || LATER == 10, "reads of the product issued behind this one");
    if constexpr (NB == 4) {
        if constexpr (LATER == 0)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.xr[0]), "+v"(R.xr[1]), "+v"(R.yr[0][0]), "+v"(R.yr[0][1]), "+v"(R.yr[1][0]), "+v"(R.yr[1][1]),
                                                  "+v"(R.yr[2][0]), "+v"(R.yr[2][1]), "+v"(R.yr[3][0]), "+v"(R.yr[3][1]) :: "memory");
        else if constexpr (LATER == 6)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(R.xr[0]), "+v"(R.xr[1]), "+v"(R.yr[0][0]), "+v"(R.yr[0][1]), "+v"(R.yr[1][0]), "+v"(R.yr[1][1]),
                                                  "+v"(R.yr[2][0]), "+v"(R.yr[2][1]), "+v"(R.yr[3][0]), "+v"(R.yr[3][1]) :: "memory");
        else
            asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(R.xr[0]), "+v"(R.xr[1]), "+v"(R.yr[0][0]), "+v"(R.yr[0][1]), "+v"(R.yr[1][0]), "+v"(R.yr[1][1]),
                                                   "+v"(R.yr[2][0]), "+v"(R.yr[2][1]), "+v"(R.yr[3][0]), "+v"(R.yr[3][1]) :: "memory");
    } else {
        static_assert(LATER == 0, "the narrow product is the second of its tile");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(R.xr[0]), "+v"(R.xr[1]), "+v"(R.yr[0][0]), "+v"(R.yr[0][1]), "+v"(R.yr[1][0]), "+v"(R.yr[1][1]) :: "memory");
    }
    const f16x8 xh = fb_frag(R.xr[0]), xl = fb_frag(R.xr[1]);
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    c_xy = own_reg(c_xy); c_x = own_reg(c_x);                  // (broadcast operands of packed fp32 math: rule R3)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const f16x8 yh = fb_frag(R.yr[nb][0]), yl = fb_frag(R.yr[nb][1]);
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, zero, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, t, 0, 0, 0);
        acc[nb] = acc[nb] + t * c_xy;
    }
    if constexpr (COLSUM) {
        const f16x8 ones = __builtin_bit_cast(f16x8, u32x4{0x3C003C00u, 0x3C003C00u, 0u, 0u});       // 1.0 in the four live k slots
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, ones, zero, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, ones, t, 0, 0, 0);
        colsum = colsum + t * c_x;
    }
}
// acc[mb] += W[16 mb + i][k] * act[item][k] with W as a fragment-ordered LDS image of KB k blocks per row block.
template <int MB, int KB>
__device__ __forceinline__ void fb_gemm(const float* __restrict__ img, const f32x4 (&bop)[KB], f32x4 (&acc)[MB], int lane) {
#pragma unroll
    for (int a = 0; a < KB; ++a) {
        f32x4 wv[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) wv[mb] = ld4(img + ((mb * KB + a) * 64 + lane) * 4);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = mfma16(wv[mb][b], bop[a][b], acc[mb]);
        }
    }
}

// LDS-DMA of a [16 MB][16 KB] weight block (row stride src_ld floats, 16-byte aligned rows) into a fragment-ordered
// image: fragment f = mb * KB + a is one global_load_lds_dwordx4 of the wave that owns it (f mod 4 == wave); no
// registers, nothing to wait for until the image is read (s_waitcnt vmcnt(0) + barrier by the caller).
template <int MB, int KB>
__device__ __forceinline__ void fb_stage_frags(float* img, const float* __restrict__ w, int src_ld, int wave, int lane) {
    const int i = lane & 15, q = lane >> 4;
#pragma unroll
    for (int f0 = 0; f0 < MB * KB; f0 += 4) {
        const int f = f0 + wave;
        if (f < MB * KB) {
            const int mb = f / KB, a = f % KB;
            const float* src = w + (size_t)(16 * mb + i) * src_ld + 16 * a + 4 * q;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(img + f * 256), 16, 0, 0);
        }
    }
}

// LDS-DMA of a prepared image that already lies in fragment order: NFRAG contiguous 1 KiB fragments, one wave instruction each.
template <int NFRAG>
__device__ __forceinline__ void fb_stage_image(float* img, const float* __restrict__ src, int wave, int lane) {
#pragma unroll
    for (int f0 = 0; f0 < NFRAG; f0 += 4) {
        const int f = f0 + wave;
        if (NFRAG % 4 == 0 || f < NFRAG)
            __builtin_amdgcn_global_load_lds(src + f * 256 + lane * 4, (__attribute__((address_space(3))) void*)(img + f * 256), 16, 0, 0);
    }
}

#ifdef AETHER_FUSED_STAMPS
#define FB_STAMP(id)                                                                             \
    do {                                                                                         \
        if (tid == 0 && blockIdx.x < 4096)                                                       \
            A.stamps[blockIdx.x * FUSED_STAMPS + (id)] = (float)(wall_clock64() - t_entry) * 0.01f; \
    } while (0)
#define FB_WSTAMP(id)                                                                            \
    do {                                                                                         \
        if (lane == 0 && blockIdx.x < 4096)                                                      \
            A.stamps[blockIdx.x * FUSED_STAMPS + 64 + wave * 100 + (id)] = (float)(wall_clock64() - t_entry) * 0.01f; \
    } while (0)
#else
#define FB_STAMP(id)
#define FB_WSTAMP(id)
#endif

template <int ROUNDS>
__global__ void __launch_bounds__(FB_THREADS)
k_fused_bwd(FbArgs A) {
    using L = FbLds;
    constexpr int NWV = 4;
    constexpr bool DE_REGS = ROUNDS <= 3;     // message gradients carried in registers (else: through the A.DE scratch)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wE = smem + L::WE;
    float* w2 = smem + L::W2;
    float* bias = smem + L::BIAS;
    float* psb = smem + L::PSB;
    float* prb = smem + L::PRB;
    float* dns = smem + L::DNS;
    float* invd = smem + L::INVD;
    float* dxs = smem + L::DXS;
    float* dpus = smem + L::DPU;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
#ifdef AETHER_FUSED_STAMPS
    const unsigned long long t_entry = wall_clock64();
#endif
    const FusedWG wg = A.wgdesc[blockIdx.x];
    const int vb = wg.vb, nv = wg.ve - wg.vb;
    const int nb = wg.nb, n = wg.ne - wg.nb, off = nb - vb;
    const int eb = wg.eb, m = wg.m;                      // (row pointers kept with the descriptor)
    const int n_tiles = (m + 15) >> 4;
    const int na = wg.na;
    const bool split = wg.partner >= 0;
    bool gave_up = false;          // thread 0: a bounded wait on the partner workgroup timed out
    const int nbk = n > 16 ? 2 : 1;                      // node tiles of the own range
    float* sa = smem + L::STG + wave * (3 * 16 * FB_SA);     // dpre2, then G
    float* sb = sa + 16 * FB_SA;                               // h
    float* sc = sb + 16 * FB_SA;                               // e_prev (layer 1: the edge features)

    // ---------------------------------------------------------------- per-tile structure (rounds r: tile 4r + wave)
    int ke[ROUNDS], sl[ROUNDS], rl[ROUNDS];
    unsigned inc[ROUNDS];                    // bits 0-7: sender incidence (edge 4 q + j: bit j + 4 * node block), 8-15: receiver incidence
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int local = 16 * (NWV * r + wave) + i;
        const bool ok = local < m;
        // one record per edge in local order (fused.h, k_graph_lorder): {sorted position, sender, receiver, original edge};
        // a padding row borrows the slots of the workgroup's first edge (its contributions are masked by ke < 0)
        const int4 le = m > 0 ? A.ledge[eb + (ok ? local : 0)] : make_int4(eb, vb, nb, 0);
        ke[r] = ok ? le.x : -1;
        sl[r] = le.y - vb;
        rl[r] = le.z - nb;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e_lane = (lane & 48) + 4 * q + j;             // lane of this 16-group that holds edge 4 q + j (k slot 8 q + j)
            const int es = __shfl(sl[r], e_lane), er = __shfl(rl[r], e_lane), ek = __shfl(ke[r], e_lane);
            if (ek >= 0) {
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2) {
                    if (es == 16 * b2 + i) bits |= 1u << (j + 4 * b2);
                    if (er == 16 * b2 + i) bits |= 1u << (8 + j + 4 * b2);
                }
            }
        }
        inc[r] = bits;
    }
    if (tid < FUSED_MAX_NODES) {
        float v = 1.0f;
        if (tid < n) {
            const int4 nr = *reinterpret_cast<const int4*>(A.nrange + 4 * (int64_t)(nb + tid));
            const int deg = (nr.y - nr.x) + (nr.w - nr.z);
            v = 1.0f / (float)(deg > 1 ? deg : 1);
        }
        invd[tid] = v;
    }
    // dx_4 of the own nodes -> LDS (rows of unused slots zero)
    for (int idx = tid; idx < FUSED_MAX_NODES * 16; idx += FB_THREADS) {
        const int s = idx >> 4, c = (idx & 15) * 4;
        st4(dxs + s * LDW + c, s < n ? ld4(A.dx4 + (int64_t)(nb + s) * H + c) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    __syncthreads();

    FB_STAMP(0);
    f32x4 de[DE_REGS ? ROUNDS : 1][4];       // dL/d(message tile) carried from layer l+1 to layer l
    int published = 0;                       // layers whose partial dP_s rows this workgroup's waves have published

    // Weight fragments of a layer's node update backward (W3, W4^T: rows 32 wave ..; W3^T: rows 16 wave ..).  They depend
    // on nothing the kernel computes: requested for layer l - 1 as soon as layer l's accumulators have been dumped, they
    // arrive under the reduction phases instead of costing the next node phase a memory round trip.
    f32x4 nw3f[2][4], nw4f[2][4], nw3tf[8], nb3v[2];
    // part 0: W3 + b3, part 1: W4^T, part 2: W3^T (8 KB per wave each); part < 0: all three.  In parts they are issued one per
    // round of tiles: 25 loads in one go stood a wave ~2 us in its memory instructions (every workgroup asks at the same time)
    auto issue_node_frags = [&](const int l, const int part) {
        const FbLayer Ln = A.layer[l - 1];
        int tid_o = threadIdx.x;
        asm volatile("" : "+v"(tid_o));
        const int ln = tid_o & 63, wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), ii = ln & 15, qq = ln >> 4;
        if (part < 0 || part == 0) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int mb = 2 * wv + mm;
                nb3v[mm] = ld4(Ln.upd_b0 + 16 * mb + 4 * qq);
#pragma unroll
                for (int a = 0; a < 4; ++a) nw3f[mm][a] = ld4(Ln.upd_w0 + (size_t)(16 * mb + ii) * H + 16 * a + 4 * qq);
            }
        }
        if (part < 0 || part == 1) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int a = 0; a < 4; ++a) nw4f[mm][a] = ld4(Ln.w4t + (size_t)(16 * (2 * wv + mm) + ii) * H + 16 * a + 4 * qq);
        }
        if (part < 0 || part == 2) {
#pragma unroll
            for (int a = 0; a < 8; ++a) nw3tf[a] = ld4(Ln.w3t + (size_t)(16 * wv + ii) * (2 * H) + 16 * a + 4 * qq);
        }
    };
    issue_node_frags(4, -1);

    // ================================================================ one layer of the backward
    auto layer = [&](auto first_tag, const int l) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int NBE = FIRST ? 2 : 4;                       // 16-wide blocks of the edge-side product operand
        const FbLayer Lp = A.layer[l - 1];
        // Per-layer copies of the thread coordinates behind an opaque barrier: every address derived from them is
        // recomputed in this layer instead of being hoisted out of the layer loop and kept (= spilled) for the whole
        // kernel; a spill reload waits for all global loads in flight (vmcnt counts in order).
        int tid_o = threadIdx.x;
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int i = lane & 15, q = lane >> 4;
        float* sa = smem + L::STG + wave * (3 * 16 * FB_SA);
        float* sb = sa + 16 * FB_SA;
        float* sc = sb + 16 * FB_SA;
        float* sd = smem + L::STG_G + wave * (16 * FB_SA);
        // ------------------------------------------------------------ node update backward (locs.py:240-241)
        // x_l = n + W4 silu(W3 n + b3) + b4:  dpre_u = (W4^T dx) * silu'(pre_u),  dn = dx + W3^T dpre_u
        {
            // Issue order = wait order (vmcnt counts in order): LDS-DMA of the edge weights, then the small row loads
            // whose data is needed first, then the node-phase weight fragments.  One memory round trip for all of it.
            if constexpr (FIRST) {
                fb_stage_image<8>(wE, Lp.img_e, wave, lane);               // W1 padded to K = 32: 2 terms x 4 fragments
            } else {
                fb_stage_image<16>(wE, Lp.img_e, wave, lane);
            }
            fb_stage_image<16>(w2, Lp.img_2, wave, lane);
            // (round 4: no transposed copies -- W2^T dpre2 and W_e^T G read these two images transposed, gemm_split_T)
            f32x4 pv[2][2];
            float bv0 = 0.0f, bv1 = 0.0f;
            if constexpr (FIRST) {
                if (tid < H) bv1 = A.msg_b0_1[tid];
            } else {
                // P_s of the visible nodes, P_r of the own nodes (saved by the forward): 2 x 2 rows of 16 float4 per thread
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int idx = tid + FB_THREADS * j, sl0 = idx >> 4, c = (idx & 15) * 4;
                    pv[j][0] = ld4(Lp.ps + (int64_t)(vb + (sl0 < nv ? sl0 : 0)) * H + c);
                    pv[j][1] = ld4(Lp.pr + (int64_t)(nb + (sl0 < n ? sl0 : 0)) * H + c);
                }
            }
            if (tid < H) bv0 = Lp.msg_b2[tid];
            f32x4 (&w3f)[2][4] = nw3f, (&w4f)[2][4] = nw4f, (&w3tf)[8] = nw3tf, (&b3v)[2] = nb3v;   // requested a phase ago
            if constexpr (FIRST) {
                if (tid < H) bias[H + tid] = bv1;
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int idx = tid + FB_THREADS * j, sl0 = idx >> 4, c = (idx & 15) * 4;
                    st4(psb + sl0 * LDW + c, pv[j][0]);
                    st4(prb + sl0 * LDW + c, pv[j][1]);
                }
            }
            if (tid < H) bias[tid] = bv0;
            FB_STAMP(6 + 8 * (4 - l));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t < nbk) {
                    const int slot = 16 * t + i;
                    const int64_t node = nb + (slot < n ? slot : 0);
                    f32x4 nt[4], dxv[4], pu[2], du[2];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        nt[a] = ld4(Lp.n + node * H + 16 * a + 4 * q);
                        dxv[a] = ld4(dxs + slot * LDW + 16 * a + 4 * q);
                    }
                    f32x4 dn = ld4(dxs + slot * LDW + 16 * wave + 4 * q);      // dx rows 16 * wave .. (stage B)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) { pu[mm] = b3v[mm]; du[mm] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm) {
                                pu[mm] = mfma16(w3f[mm][a][b], nt[a][b], pu[mm]);
                                du[mm] = mfma16(w4f[mm][a][b], dxv[a][b], du[mm]);
                            }
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int mb = 2 * wave + mm;
                        const f32x4 sg = sigmoid4(pu[mm]);
                        const f32x4 u = pu[mm] * sg;
                        const f32x4 dpu = du[mm] * dsilu_from_sigmoid(pu[mm], sg);
                        if (slot < n) {
                            st4(Lp.U + node * 2 * H + 16 * mb + 4 * q, u);
                            st4(Lp.DPU + node * 2 * H + 16 * mb + 4 * q, dpu);
                        }
                        st4(dpus + slot * LDU + 16 * mb + 4 * q, dpu);
                    }
                    FB_STAMP(7 + 8 * (4 - l));
                    lds_barrier();                                // dpre_u of this node tile complete
#pragma unroll
                    for (int a = 0; a < 8; ++a) {
                        const f32x4 bv = ld4(dpus + slot * LDU + 16 * a + 4 * q);
#pragma unroll
                        for (int b = 0; b < 4; ++b) dn = mfma16(w3tf[a][b], bv[b], dn);
                    }
                    st4(dns + slot * LDW + 16 * wave + 4 * q, dn);
                    if (FIRST && slot < n) st4(A.DN1 + node * H + 16 * wave + 4 * q, dn);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA fragments have landed
            lds_barrier();         // dn, weights, P rows, bias complete; dxs / dpus dead: staging rows are free
        }
        FB_STAMP(1 + 8 * (4 - l));

        // ------------------------------------------------------------ edge tiles, last tile first
        // dW2 += dpre2 (x) h;  dW_e += G (x) e_prev   (dW1 += G (x) features): this wave's row quarter (m = 4 i' + wave) over ALL
        // tiles of the workgroup (fb_outer16_q) -- final for the workgroup, written out without a cross-wave reduction
        f32x4 aw2[4], awe[NBE];
        f32x4 dps[2][4], dpr[2][4];            // sum of G over out-edges (visible slots) / in-edges (own slots)
        f32x4 db2 = f32x4{0.f, 0.f, 0.f, 0.f}, db1 = db2;      // bias gradients of this wave's rows (fb_outer16_q's column sums)
#pragma unroll
        for (int nbx = 0; nbx < 4; ++nbx) aw2[nbx] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nbx = 0; nbx < NBE; ++nbx) awe[nbx] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int nbx = 0; nbx < 4; ++nbx) { dps[b2][nbx] = f32x4{0.f, 0.f, 0.f, 0.f}; dpr[b2][nbx] = f32x4{0.f, 0.f, 0.f, 0.f}; }

        // Split mode: the wave's partial dP_s rows (all visible slots; the partner reads its own nodes' rows, which
        // are final once the wave is through its tiles with partner senders) -> xchg, then one add on the counter.
        auto publish = [&]() {
            float* dst = A.xchg + (((size_t)(4 - l) * A.n_wgs + blockIdx.x) * NWV + wave) * (FUSED_MAX_NODES * H);
            // transpose through the wave's staging rows: sa[node][h] (32 rows x 64 fit the 2 x 16 x FB_SA floats)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int nbx = 0; nbx < 4; ++nbx)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) sa[(16 * b2 + 4 * q + r4) * LDST + 16 * nbx + i] = dps[b2][nbx][r4];
            __builtin_amdgcn_wave_barrier();
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, FUSED_MAX_NODES * H * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = lane + 64 * j, row = idx >> 4, c = (idx & 15) * 4;
                const f32x4 v = ld4(sa + row * LDST + c);
                u32x4 bits;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) bits[r4] = __float_as_uint(v[r4]);
                __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (row * H + c) * 4, 0, 16);     // aux 16 = sc1
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_fetch_add(A.flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_wave_barrier();
        };
        static_assert(FUSED_MAX_NODES * LDST <= 3 * 16 * FB_SA, "publish transposes through the wave's staging rows");

        // e_prev rows of a tile (layer 1: its feature rows), requested one tile ahead: the single wave of a SIMD has
        // nothing else to hide a memory round trip behind
        auto load_ep = [&](int rr, f32x4 (&dst)[4]) {
            const int kr = ke[rr];
            const int64_t kc = kr >= 0 ? kr : eb;
            if constexpr (FIRST) {
                dst[0] = ld4(Lp.e_prev + kc * FPAD + 4 * q);
                dst[1] = ld4(Lp.e_prev + kc * FPAD + 16 + 4 * q);
            } else {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) dst[mb] = ld4(Lp.e_prev + kc * H + 16 * mb + 4 * q);
            }
        };
        f32x4 epn[4];
        {
            bool have = false;                                 // the wave's first tile = its highest existing round
#pragma unroll
            for (int r = ROUNDS - 1; r >= 0; --r)
                if (!have && NWV * r + wave < n_tiles) { load_ep(r, epn); have = true; }
        }
        // The NEXT layer's node-phase weight fragments (100 KB per workgroup) are requested during the tile phase, a third per
        // round (below): they arrive under the 28 us of edge tiles.  Requested at the end of the layer (round 3) they met every
        // other workgroup's -- all 256 run their layers in step -- and the wave stood 3-4 us in its memory instructions; the
        // row partition of the weight-gradient accumulators (96 registers less) makes room for them during the tiles.
        bool need_publish = !FIRST && split;
#pragma unroll
        for (int r = ROUNDS - 1; r >= 0; --r) {
            const int tile = NWV * r + wave;
            if (need_publish && tile < n_tiles && 16 * (tile + 1) <= na) {         // no partner sender from here on
                publish();
                need_publish = false;
            }
            if (l > 1) {           // (rounds: ROUNDS - 1 .. 0 -> parts 0 ..; with fewer than three rounds the rest follows the loop)
                const int part = ROUNDS - 1 - r;
                if (part < 3) issue_node_frags(l - 1, part);
            }
            if (tile < n_tiles) {
                FB_WSTAMP(20 * (4 - l) + 4 * r);
                // Opaque copies: whatever is derived from them (a dozen LDS / global addresses per tile) is recomputed
                // here instead of being hoisted out of the layer loop for all tiles at once -- hoisted, those values
                // spill, and every spill reload waits for all global loads in flight (vmcnt counts in order).
                int ker = ke[r], slr = sl[r], rlr = rl[r];
                unsigned incr = inc[r];
                asm volatile("" : "+v"(ker), "+v"(slr), "+v"(rlr), "+v"(incr));
                const bool ok = ker >= 0;
                const int64_t kc = ok ? ker : eb;
                const float vm = ok ? 1.0f : 0.0f;
                // ---- forward recompute: pre1, h = silu(pre1), pre2.  Operands of the weight-gradient products go to the
                // wave's staging rows as soon as they exist, so that their registers die with the GEMM that reads them.
                f32x4 ep[4], p1[4], ds1[4], hh[4], p2[4];
#pragma unroll
                for (int mb = 0; mb < (FIRST ? 2 : 4); ++mb) ep[mb] = epn[mb] * vm;
                if (r > 0) load_ep(r > 0 ? r - 1 : 0, epn);      // tile r exists, so does tile r - 1
                // (each operand's fp16 pieces go to the wave's staging rows as its GEMM makes them: fb_stage_pieces)
                SplitScale s_ep;
                if constexpr (FIRST) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(bias + H + 16 * mb + 4 * q);
                    f32x4 b2[2] = {ep[0], ep[1]};
                    f16x8 ph[1], pl[1];
                    s_ep = gemm_split_keep<4, 1>(wE, b2, p1, lane, ph, pl);
                    fb_stage_pieces<1>(sc, i, q, ph, pl);
                } else {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        p1[mb] = ld4(psb + slr * LDW + 16 * mb + 4 * q) + ld4(prb + rlr * LDW + 16 * mb + 4 * q);
                    f16x8 ph[2], pl[2];
                    s_ep = gemm_split_keep<4, 2>(wE, ep, p1, lane, ph, pl);
                    fb_stage_pieces<2>(sc, i, q, ph, pl);
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(90);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const f32x4 sg = sigmoid4(p1[mb]);
                    hh[mb] = p1[mb] * sg;
                    ds1[mb] = dsilu_from_sigmoid(p1[mb], sg);
                    p2[mb] = ld4(bias + 16 * mb + 4 * q);
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(91);
                SplitScale s_h;
                {
                    f16x8 ph[2], pl[2];
                    s_h = gemm_split_keep<4, 2>(w2, hh, p2, lane, ph, pl);
                    fb_stage_pieces<2>(sb, i, q, ph, pl);
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(92);
                // ---- de = dn[recv] / deg (+ gradient through the next layer's edge input); back through both Linears
                f32x4 d2[4], dh[4], g[4];
                const float sc_deg = invd[rlr] * vm;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    f32x4 dev = ld4(dns + rlr * LDW + 16 * mb + 4 * q) * sc_deg;
                    if (l < 4) {
                        if constexpr (DE_REGS) dev += de[r][mb];
                        else dev += ld4(A.DE + kc * H + 16 * mb + 4 * q) * vm;       // (ROUNDS > 3: [E][64] scratch)
                    }
                    d2[mb] = dev * dsilu_from_sigmoid(p2[mb], sigmoid4(p2[mb]));
                    dh[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(93);
                SplitScale s_d2;
                {
                    f16x8 ph[2], pl[2];
                    s_d2 = gemm_split_T_keep<2>(w2, d2, dh, lane, ph, pl);       // W2^T dpre2 on the matrix pipe, from the forward's image
                    fb_stage_pieces<2>(sa, i, q, ph, pl);
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(94);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) g[mb] = dh[mb] * ds1[mb];
                // ---- gradient into this layer's edge input (its GEMM also makes G's pieces for the staging rows)
                SplitScale s_g;
                if constexpr (FIRST) {
                    f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                    f16x8 ph[2], pl[2];
                    s_g = gemm_split_T_keep<1>(wE, g, da, lane, ph, pl);            // W1^T G
                    fb_stage_pieces<2>(sd, i, q, ph, pl);
                    if (ok) { st4(A.DA + kc * FPAD + 4 * q, da[0]); st4(A.DA + kc * FPAD + 16 + 4 * q, da[1]); }
                } else {
                    f32x4 dep[4];
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) dep[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                    f16x8 ph[2], pl[2];
                    s_g = gemm_split_T_keep<2>(wE, g, dep, lane, ph, pl);           // W_e^T G
                    fb_stage_pieces<2>(sd, i, q, ph, pl);
                    if constexpr (DE_REGS) {
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) de[r][mb] = dep[mb];
                    } else if (ok) {
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb) st4(A.DE + kc * H + 16 * mb + 4 * q, dep[mb]);
                    }
                }
                if (lane == 0) {                               // this tile's scales for the round's products
                    float* scl = smem + L::SCL + 4 * wave;
                    scl[0] = s_d2.inv_s * s_h.inv_s; scl[1] = s_g.inv_s * s_ep.inv_s; scl[2] = s_d2.inv_s; scl[3] = s_g.inv_s;
                }
                __builtin_amdgcn_wave_barrier();
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(96);
                if constexpr (!FIRST) {
                    // out[node][h] += sum_edge Inc[node][edge] * G[edge][h]: A = the lane's incidence bits as 0 / 1 fp16, B = G's
                    // staged pieces read transposed; two terms (A is exact), then the tile's 1 / s_G
                    const unsigned ga = fb_lds_addr(sd) + fb_tr_lane_offset(lane);
                    u32x2 gr[4][2];
                    FB_TR(gr[0][0], ga, 0); FB_TR(gr[0][1], ga, FB_PLANE);
                    FB_TR(gr[1][0], ga, 32); FB_TR(gr[1][1], ga, FB_PLANE + 32);
                    FB_TR(gr[2][0], ga, 64); FB_TR(gr[2][1], ga, FB_PLANE + 64);
                    FB_TR(gr[3][0], ga, 96); FB_TR(gr[3][1], ga, FB_PLANE + 96);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(gr[0][0]), "+v"(gr[0][1]), "+v"(gr[1][0]), "+v"(gr[1][1]), "+v"(gr[2][0]),
                                                          "+v"(gr[2][1]), "+v"(gr[3][0]), "+v"(gr[3][1]));
                    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float cg = own_reg(s_g.inv_s);                       // (broadcast operand of packed fp32 math: rule R3)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        if (b2 == 0 || nv > 16) {
                            const f16x8 sv = fb_inc_frag((incr >> (4 * b2)) & 15u);
#pragma unroll
                            for (int nbx = 0; nbx < 4; ++nbx) {
                                f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(sv, fb_frag(gr[nbx][1]), zero, 0, 0, 0);
                                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(sv, fb_frag(gr[nbx][0]), t, 0, 0, 0);
                                dps[b2][nbx] = dps[b2][nbx] + t * cg;
                            }
                        }
                        if (b2 == 0 || n > 16) {
                            const f16x8 rv = fb_inc_frag((incr >> (8 + 4 * b2)) & 15u);
#pragma unroll
                            for (int nbx = 0; nbx < 4; ++nbx) {
                                f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(rv, fb_frag(gr[nbx][1]), zero, 0, 0, 0);
                                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(rv, fb_frag(gr[nbx][0]), t, 0, 0, 0);
                                dpr[b2][nbx] = dpr[b2][nbx] + t * cg;
                            }
                        }
                    }
                }
                if (l == 3 && r == ROUNDS - 1) FB_WSTAMP(98);
            }
            // ---- the round's four tiles are staged (d2 | h | e_prev per wave, G beside them): every wave adds its row quarter
            // of both weight-gradient products over all of them (workgroup-uniform conditions: every wave takes the barriers)
            if (NWV * r < n_tiles) {
                lds_barrier();
#pragma unroll
                for (int t = 0; t < NWV; ++t) {
                    if (NWV * r + t < n_tiles) {
                        const float* ta = smem + L::STG + t * (3 * 16 * FB_SA);
                        const float* td = smem + L::STG_G + t * (16 * FB_SA);
                        const float* scl = smem + L::SCL + 4 * t;
                        const unsigned lo_ = fb_tr_lane_offset(lane), xo = lo_ + 32u * (unsigned)wave;      // rows m = 16 wave ..
                        const float c0 = scl[0], c1 = scl[1], c2 = scl[2], c3 = scl[3];
                        FbOuterRegs<4> R2;
                        FbOuterRegs<NBE> Re;
                        fb_outer16_issue<4>(fb_lds_addr(ta) + xo, fb_lds_addr(ta + 16 * FB_SA) + lo_, R2);                  // both products' reads in flight
                        fb_outer16_issue<NBE>(fb_lds_addr(td) + xo, fb_lds_addr(ta + 2 * 16 * FB_SA) + lo_, Re);
                        fb_outer16_consume<4, true, 2 + 2 * NBE>(R2, aw2, db2, c0, c2);                                   // dW2 += dpre2 (x) h, db2
                        fb_outer16_consume<NBE, FIRST, 0>(Re, awe, db1, c1, c3);                                          // dW_e += G (x) e_prev
                    }
                }
                lds_barrier();         // staging rows are free for the next round (and for publish)
            }
        }
        if (need_publish) publish();
        if (l > 1) {
#pragma unroll
            for (int part = ROUNDS; part < 3; ++part) issue_node_frags(l - 1, part);
        }
        FB_WSTAMP(20 * (4 - l) + 16);

        // ------------------------------------------------------------ reductions over the workgroup
        lds_barrier();             // every wave is through its tiles: weight images and P rows are dead
        FB_STAMP(2 + 8 * (4 - l));
        if constexpr (!FIRST) {
            // (1) incidence sums: per-wave dumps -> totals in wave order (+ the partner's four partials)
            float* dump_s = smem + L::DUMP_S + wave * (FUSED_MAX_NODES * LDST);
            float* dump_r = smem + L::DUMP_R + wave * (FUSED_MAX_NODES * LDST);
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int nbx = 0; nbx < 4; ++nbx)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        dump_s[(16 * b2 + 4 * q + r4) * LDST + 16 * nbx + i] = dps[b2][nbx][r4];
                        dump_r[(16 * b2 + 4 * q + r4) * LDST + 16 * nbx + i] = dpr[b2][nbx][r4];
                    }
            if (split && tid == 0) {         // the partner's partials of this layer: 4 waves x (layers so far)
                const int want = NWV * (5 - l);
                unsigned spins = 0;
                unsigned long long t_first = 0;
                while (__hip_atomic_load(A.flags + wg.partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if ((++spins & 1023u) == 0) {              // bounded by time (5 s), as in fused.h
                        const unsigned long long now = wall_clock64();
                        if (t_first == 0) t_first = now;
                        if (now - t_first <= 500000000ull) continue;
                        // partner not resident: report, and poison this workgroup's
                        if (A.errword) __hip_atomic_store(A.errword, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        gave_up = true;                        // partial sums at exit (gradients come out NaN, not stale)
                        break;
                    }
                }
            }
            lds_barrier();
            const float* px = A.xchg + ((size_t)(4 - l) * A.n_wgs + (split ? wg.partner : 0)) * NWV * (FUSED_MAX_NODES * H);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(px), 0, NWV * FUSED_MAX_NODES * H * 4, 0x00020000);
            for (int idx = tid; idx < FUSED_MAX_NODES * 16; idx += FB_THREADS) {
                const int s = idx >> 4, c = (idx & 15) * 4;         // own slot s
                f32x4 ts = f32x4{0.f, 0.f, 0.f, 0.f}, tr = f32x4{0.f, 0.f, 0.f, 0.f};
                if (s < n) {
#pragma unroll
                    for (int w = 0; w < NWV; ++w) {
                        ts += ld4(smem + L::DUMP_S + w * (FUSED_MAX_NODES * LDST) + (off + s) * LDST + c);
                        tr += ld4(smem + L::DUMP_R + w * (FUSED_MAX_NODES * LDST) + s * LDST + c);
                    }
                    if (split) {
#pragma unroll
                        for (int w = 0; w < NWV; ++w) {
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                                rsrc, ((w * FUSED_MAX_NODES + (off + s)) * H + c) * 4, 0, 16);          // aux 16 = sc1
                            f32x4 f;
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) f[r4] = __uint_as_float(v[r4]);
                            ts += f;
                        }
                    }
                    st4(Lp.DPS + (int64_t)(nb + s) * H + c, ts);
                    st4(Lp.DPR + (int64_t)(nb + s) * H + c, tr);
                }
                st4(psb + s * LDW + c, ts);        // totals as B operands of the dx GEMM (own slots)
                st4(prb + s * LDW + c, tr);
            }
            lds_barrier();
        }
        FB_STAMP(3 + 8 * (4 - l));
        // (2) + (3).  Order = what the in-order memory counter wants (loads and stores share vmcnt): first every LOAD the rest
        // of the layer and the next node phase need, then the partial's stores (fire and forget: a wait for the loads above
        // leaves them outstanding), then the dx GEMM.
        f32x4 wsf[4], wrf[4];
        if constexpr (!FIRST) {
            const float* w1t = Lp.w0t;                    // [192][64]: W_s^T | W_r^T | W_e^T
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                wsf[a] = ld4(w1t + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
                wrf[a] = ld4(w1t + (size_t)H * H + (size_t)(16 * wave + i) * H + 16 * a + 4 * q);
            }
        }
        // (2) edge-level weight gradients: every wave holds a row quarter (rows m = 4 i' + wave) of the workgroup's two sums and
        // of the bias sums -- straight to the partial, no reduction over the waves (round 3: two LDS passes per layer)
        {
            float* dst = A.partial + ((size_t)blockIdx.x * 4 + (l - 1)) * FB_PART;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int mrow = 16 * wave + 4 * q + r4;       // accumulator block nb holds columns 16 nb + i
#pragma unroll
                for (int nbx = 0; nbx < 4; ++nbx) dst[mrow * H + 16 * nbx + i] = aw2[nbx][r4];
                if constexpr (FIRST) {
                    dst[H * H + mrow * FPAD + i] = awe[0][r4];
                    dst[H * H + mrow * FPAD + 16 + i] = awe[1][r4];
                } else {
#pragma unroll
                    for (int nbx = 0; nbx < 4; ++nbx) dst[H * H + mrow * H + 16 * nbx + i] = awe[nbx][r4];
                }
                if (i == 0) {                             // (every column of the ones product holds the sum)
                    dst[2 * H * H + mrow] = db2[r4];
                    dst[2 * H * H + H + mrow] = FIRST ? db1[r4] : 0.0f;
                }
            }
        }
        FB_STAMP(4 + 8 * (4 - l));
        // (3) dx_{l-1} = dn_l + W_s^T dP_s + W_r^T dP_r   (locs.py:233 split, transposed): rows 16 * wave ..
        if constexpr (!FIRST) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (t >= nbk) continue;
                const int slot = 16 * t + i;
                f32x4 dx = ld4(dns + slot * LDW + 16 * wave + 4 * q);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const f32x4 bs = ld4(psb + slot * LDW + 16 * a + 4 * q);
                    const f32x4 br = ld4(prb + slot * LDW + 16 * a + 4 * q);
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        dx = mfma16(wsf[a][b], bs[b], dx);
                        dx = mfma16(wrf[a][b], br[b], dx);
                    }
                }
                st4(dxs + slot * LDW + 16 * wave + 4 * q, dx);
                if (slot < n) st4(Lp.DXout + (int64_t)(nb + slot) * H + 16 * wave + 4 * q, dx);
            }
        }
        lds_barrier();             // dx_{l-1} complete; reduction buffers consumed: weight images are free again
        FB_STAMP(5 + 8 * (4 - l));
    };

#pragma unroll 1
    for (int l = 4; l >= 2; --l) layer(std::integral_constant<bool, false>{}, l);
    layer(std::integral_constant<bool, true>{}, 1);
    (void)published;
    if (split && tid == 0)         // the partner's counter has been consumed for the last time: re-arm it
        __hip_atomic_store(A.flags + wg.partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (split) {
        __syncthreads();           // every wave's stores of the partials have completed (vmcnt drained): the NaN lands last
        if (tid == 0 && gave_up) {
            for (int l = 0; l < 4; ++l) A.partial[((size_t)blockIdx.x * 4 + l) * FB_PART] = __builtin_nanf("");
        }
    }
}

// Sum of the workgroups' partials in workgroup order: element e of (layer, FB_PART).  1024 threads = 256 elements x 4
// groups; a group adds its contiguous quarter of the workgroups in order, the four sums are combined in order.
struct FbReduceArgs {
    const float* partial; int n_wgs;
    float* w2[4]; float* b2[4];
    float* we[4];              // layer 1: dW1 [64][f1]; layers 2-4: into msg_w0 + 2H (row stride 3H)
    float* b1;                 // layer 1 only
    int f1;
};
__device__ __forceinline__ void fb_reduce_body(const FbReduceArgs& R, int bx, int l /* layer index 0..3 */) {
    // 64 elements x 16 groups per block: a group adds its contiguous sixteenth of the workgroups in order, the sixteen
    // sums are combined in order -- a fixed tree; short dependent chains and 4 x 130 blocks to fill the chip
    const int grp = threadIdx.x >> 6, t = threadIdx.x & 63, e = bx * 64 + t;
    const bool valid = e < FB_PART;
    const int per = (R.n_wgs + 15) / 16;
    const int c0 = grp * per, c1 = c0 + per < R.n_wgs ? c0 + per : R.n_wgs;
    float s = 0.0f;
    if (valid) {
        const float* src = R.partial + (size_t)l * FB_PART + e;
        int w = c0;
        for (; w + 8 <= c1; w += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(w + u) * 4 * FB_PART];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; w < c1; ++w) s += src[(size_t)w * 4 * FB_PART];
    }
    __shared__ float red[16][64];
    red[grp][t] = s;
    __syncthreads();
    if (grp != 0 || !valid) return;
    float tot = red[0][t];
#pragma unroll
    for (int g = 1; g < 16; ++g) tot += red[g][t];
    if (e < H * H) {
        R.w2[l][e] = tot;
    } else if (e < 2 * H * H) {
        const int o = e - H * H;
        if (l == 0) {
            const int mrow = o / FPAD, c = o % FPAD;
            if (o < H * FPAD && c < R.f1) R.we[0][mrow * R.f1 + c] = tot;
        } else {
            R.we[l][(o >> 6) * (3 * H) + (o & 63)] = tot;
        }
    } else if (e < 2 * H * H + H) {
        R.b2[l][e - 2 * H * H] = tot;
    } else if (l == 0) {
        R.b1[e - 2 * H * H - H] = tot;
    }
}
constexpr int FB_REDUCE_GX = (FB_PART + 63) / 64;
__global__ void __launch_bounds__(1024)
k_fb_reduce(FbReduceArgs R) { fb_reduce_body(R, (int)blockIdx.x, (int)blockIdx.y); }

// k_outer_reduce (the node-level products' partials, backward.h) and k_fb_reduce in one launch: they write different
// gradient tensors and neither reads what the other writes.  Blocks [0, outer_blocks): (bx, task) of k_outer_reduce.
constexpr int OUTER_REDUCE_GX = 64 * 64 / 256 + 1;
__global__ void __launch_bounds__(1024)
k_reduce_both(OuterBatch batch, const float* __restrict__ partial, FbReduceArgs R, int outer_blocks) {
    const int b = (int)blockIdx.x;
    if (b < outer_blocks) {
        outer_reduce_body(batch, partial, b % OUTER_REDUCE_GX, b / OUTER_REDUCE_GX);
    } else {
        const int c = b - outer_blocks;
        fb_reduce_body(R, c % FB_REDUCE_GX, c / FB_REDUCE_GX);
    }
}

}  // namespace
