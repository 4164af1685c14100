// edge_acc.h -- edge MLP backward of one layer for LARGE graphs with the layer's edge-level weight gradients
// accumulated on chip (round 3).  Included after backward.h and fused_bwd.h.
//
// backward.h's kb_edge writes, per edge and layer, the rows G = dL/dpre1, h and dpre2 (768 B) for k_outer to multiply
// afterwards (which reads them again, with e_prev: 1 KB): at the 33.5 M-edge shard of BASELINE config 5 that is 34 GB of
// HBM traffic per layer for the two products and 24 ms of a 106 ms training step (profiles/r02_cfg5shard_bench.json).
// kb_edge_acc keeps the tile's four operand tiles in wave-private LDS rows and accumulates
//     dW2 += dpre2 (x) h          dW_e += G (x) e_prev   (layer 1: dW1 += G (x) features)         db2 += dpre2, db1 += G
// in registers over all tiles of the wave (fused_bwd.h's block scheme, fb_outer16), as k_fused_bwd does for small graphs;
// only G (read by the node-sum kernels) and the message gradient DE / DA still go to memory.  One 4-wave workgroup per CU
// (the accumulators need the AGPR half of the register file); at the end the four waves add up in wave order through LDS
// and write ONE partial per workgroup; k_edge_acc_reduce adds the partials in workgroup order (fixed 16-group tree):
// no atomics, bit-stable.  Used when the deferred all-in-one product launch does not apply (E > outer_defer_max_edges).
#pragma once

namespace {

constexpr int EA_REGS = 128 + 32;                   // accumulator registers per lane: dW2 64 | dW_e 64 | db2 16 | db1 16
constexpr int EA_PART = EA_REGS * 64;               // floats per workgroup partial, [register][lane]
constexpr int EA_STG = 2 * 16 * FB_SA;              // staging floats per wave: one operand pair at a time (dpre2 | h, then G | e_prev)

template <bool FIRST>
__global__ void __launch_bounds__(256)
kb_edge_acc(const float* __restrict__ w_in /*FIRST: W1 [64][F1] else msg_w0 [64][192]*/, int f1,
            const float* __restrict__ b_in, const float* __restrict__ w2g, const float* __restrict__ b2g,
            const float* __restrict__ w_in_t /*FIRST: W1^T [32][64] else W_e^T [64][64]*/,
            const float* __restrict__ w2t, const float* __restrict__ Ps, const float* __restrict__ Pr,
            const float* __restrict__ e_prev, const float* __restrict__ feat,
            const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
            const int32_t* __restrict__ rowptr, const float* __restrict__ DN, float* __restrict__ DE,
            int have_de, float* __restrict__ G, float* __restrict__ DA, float* __restrict__ partial,
            const float* __restrict__ img_in /*split image of W_e (FIRST: of W1, K padded to 32)*/,
            const float* __restrict__ img_2 /*split image of W2*/, int64_t n_edges) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // All four products of a tile run as six bf16 terms on split images (common.h gemm_split: fp32-equivalent, 96
    // matrix-pipe instructions of 16 cycles for 128 fp32 MFMAs of 32 that would issue on the vector ALU): the two recompute
    // products on the forward's images, the two transposed ones (dh = W2^T dpre2, de = W_in^T G) on images of the transposed
    // copies built here.  (Round 3, first version: the transposed products as fp32 MFMAs -- 128 of the tile's 256.)
    float* wi = smem;                  // split image of W_in [SPLIT_WIMG] (FIRST: half of it)
    float* w2 = wi + SPLIT_WIMG;       // split image of W2
    float* w2ts = w2 + SPLIT_WIMG;     // split image of W2^T
    float* wit = w2ts + SPLIT_WIMG;    // split image of W_in^T (FIRST: W1^T, 32 rows: half of it)
    float* stg = wit + SPLIT_WIMG;     // [4 waves][2][16][FB_SA]
    for (int idx = threadIdx.x; idx < (FIRST ? SPLIT_WIMG / 2 : SPLIT_WIMG) / 4; idx += 256) st4(wi + 4 * idx, ld4(img_in + 4 * idx));
    for (int idx = threadIdx.x; idx < SPLIT_WIMG / 4; idx += 256) st4(w2 + 4 * idx, ld4(img_2 + 4 * idx));
    for (int idx = threadIdx.x; idx < H * H / 4; idx += 256) {
        const int r = idx >> 4, c = (idx & 15) * 4;
        stage_split4<4, 2>(w2ts, r, c, ld4(w2t + (size_t)r * H + c));
        if (FIRST) {
            if (r < FPAD) stage_split4<2, 2>(wit, r, c, ld4(w_in_t + (size_t)r * H + c));
        } else {
            stage_split4<4, 2>(wit, r, c, ld4(w_in_t + (size_t)r * H + c));
        }
    }
    (void)w_in; (void)f1; (void)w2g;      // (kept in the signature: same argument list as kb_edge)
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    float* sa = stg + wave * EA_STG;   // dpre2, then G
    float* sb = sa + 16 * FB_SA;       // h, then e_prev / features
    const int64_t tiles = (n_edges + 15) >> 4;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + wave;
    f32x4 accW2[4][4], bs2[4], bs1[4];
    f32x4 accWe[4][FIRST ? 2 : 4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        bs2[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        bs1[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 4; ++b) accW2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < (FIRST ? 2 : 4); ++b) accWe[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto clampk = [&](int64_t tt) { const int64_t k = 16 * tt + i; return k < n_edges ? k : n_edges - 1; };
    if (t < tiles) {
        // Software pipeline, two stages deep (one wave per SIMD: nobody else hides a tile's memory round trip): the ROWS of
        // tile t + stride and the INDICES of tile t + 2 stride are requested while tile t computes.
        f32x4 nbop[4], nps[4], npr[4], ndn[4], ndev[4];      // (P_s and P_r rows stay apart: adding them here would wait for them)
        int ndeg;
        auto request_rows = [&](int64_t kk, int32_t ss, int32_t rr) {
            if (FIRST) {
                nbop[0] = ld4(feat + kk * FPAD + 4 * q);
                nbop[1] = ld4(feat + kk * FPAD + 16 + 4 * q);
            } else {
                load_tile64(nbop, e_prev, kk, H, q);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    nps[mb] = ld4(Ps + (int64_t)ss * H + 16 * mb + 4 * q);
                    npr[mb] = ld4(Pr + (int64_t)rr * H + 16 * mb + 4 * q);
                }
            }
            load_tile64(ndn, DN, rr, H, q);
            if (have_de) load_tile64(ndev, DE, kk, H, q);
            ndeg = rowptr[rr + 1] - rowptr[rr];
        };
        {
            const int64_t k0 = clampk(t);
            request_rows(k0, send_s[k0], recv_s[k0]);
        }
        int64_t kn = clampk(t + stride < tiles ? t + stride : t);
        int32_t s_n = send_s[kn], r_n = recv_s[kn];
        for (; t < tiles; t += stride) {
            int z = 0;
            asm volatile("" : "+v"(z));      // opaque offset: weight fragments are re-read from LDS per tile
            const int64_t k = 16 * t + i;
            const bool ok = k < n_edges;
            // ---- this tile's rows: requested one iteration ago
            f32x4 bop[4], p1[4], dnv[4], dev[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) { bop[mb] = nbop[mb]; dnv[mb] = ndn[mb]; dev[mb] = ndev[mb]; }
            if (FIRST) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(b_in + 16 * mb + 4 * q);
            } else {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) p1[mb] = nps[mb] + npr[mb];
            }
            const float inv = 1.0f / (float)(ndeg > 1 ? ndeg : 1);
            // ---- the next tile's rows and the indices of the tile after it
            request_rows(kn, s_n, r_n);
            kn = clampk(t + 2 * stride < tiles ? t + 2 * stride : t);
            s_n = send_s[kn];
            r_n = recv_s[kn];
            // ---- forward recompute: pre1, h = silu(pre1), pre2
            f32x4 p2[4], h[4], sg1[4];
            if (FIRST) {
                f32x4 b2[2] = {bop[0], bop[1]};
                gemm_split<4, 1>(wi + z, b2, p1, lane);
            } else {
                gemm_split<4, 2>(wi + z, bop, p1, lane);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                sg1[mb] = sigmoid4(p1[mb]);
                h[mb] = p1[mb] * sg1[mb];
                p2[mb] = ld4(b2g + 16 * mb + 4 * q);
            }
            gemm_split<4, 2>(w2 + z, h, p2, lane);
            // ---- de = dn[recv] / deg (+ gradient through the next layer's edge input); back through both Linears
            f32x4 d2[4], dh[4], g[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 de = dnv[mb] * inv;
                if (have_de) de += dev[mb];
                d2[mb] = de * dsilu_from_sigmoid(p2[mb], sigmoid4(p2[mb]));
                if (!ok) d2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};             // rows past the end contribute nothing to the products
                dh[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            gemm_split<4, 2>(w2ts + z, d2, dh, lane);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) g[mb] = dh[mb] * dsilu_from_sigmoid(p1[mb], sg1[mb]);
            if (FIRST) {
                f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                gemm_split<2, 2>(wit + z, g, da, lane);
                if (ok) { st4(DA + k * FPAD + 4 * q, da[0]); st4(DA + k * FPAD + 16 + 4 * q, da[1]); }
            } else {
                f32x4 dep[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) dep[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                gemm_split<4, 2>(wit + z, g, dep, lane);
                if (ok) store_tile64(DE, k, H, q, dep);
            }
            if (ok) store_tile64(G, k, H, q, g);
            // ---- the layer's edge-level weight gradients, on chip: one operand pair in the wave's staging rows at a time
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                st4(sa + i * FB_SA + 16 * mb + 4 * q, d2[mb]);
                st4(sb + i * FB_SA + 16 * mb + 4 * q, h[mb]);
                bs2[mb] += d2[mb];
                bs1[mb] += g[mb];
            }
            __builtin_amdgcn_wave_barrier();
            fb_outer16<4>(sa, sb, accW2, i, q);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                st4(sa + i * FB_SA + 16 * mb + 4 * q, g[mb]);          // (g of a row past the end is 0: dh = W2^T 0)
                if (!FIRST || mb < 2) st4(sb + i * FB_SA + 16 * mb + 4 * q, ok ? bop[mb] : f32x4{0.f, 0.f, 0.f, 0.f});
            }
            __builtin_amdgcn_wave_barrier();
            fb_outer16<FIRST ? 2 : 4>(sa, sb, accWe, i, q);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- four waves -> one partial: every wave parks its accumulators [register][lane] in LDS (the weights are dead),
    // thread t adds the four copies of its elements in wave order
    __syncthreads();
    float* red = smem;                                       // [4 waves][EA_PART] = 160 KB > LDS: two halves of 80 registers
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        float* mine = red + wave * (EA_PART / 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (half == 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) mine[((a * 4 + b) * 4 + rr) * 64 + lane] = accW2[a][b][rr];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) mine[(64 + a * 4 + rr) * 64 + lane] = bs2[a][rr];
            } else {
#pragma unroll
                for (int b = 0; b < (FIRST ? 2 : 4); ++b)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) mine[((a * 4 + b) * 4 + rr) * 64 + lane] = accWe[a][b][rr];
                if (FIRST) {
#pragma unroll
                    for (int b = 2; b < 4; ++b)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) mine[((a * 4 + b) * 4 + rr) * 64 + lane] = 0.0f;
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) mine[(64 + a * 4 + rr) * 64 + lane] = bs1[a][rr];
            }
        }
        __syncthreads();
        float* dst = partial + (size_t)blockIdx.x * EA_PART + half * (EA_PART / 2);
        for (int e = threadIdx.x; e < EA_PART / 2; e += 256)
            dst[e] = ((red[e] + red[EA_PART / 2 + e]) + red[2 * (EA_PART / 2) + e]) + red[3 * (EA_PART / 2) + e];
    }
}

// Sum of the workgroups' partials (workgroup order, fixed 16-group tree) and the scatter into the gradient tensors.
// Partial layout per workgroup: half 0 = [dW2 64 regs | db2 16 regs][lane], half 1 = [dW_e 64 regs | db1 16 regs][lane];
// register (mb, nb, r) of lane (i, q) of a <4, NB> product is element (m = 4 (4 q + r) + mb, n = NB i + nb) (fb_outer16);
// a bias register (mb, r) of lane (i, q) is hidden unit 16 mb + 4 q + r summed over the lane's edges: add the 16 lanes i.
struct EdgeAccOut {
    float* w2; float* b2;              // [64][64], [64]
    float* we; int ldwe; int ncols;    // dW_e -> we[m * ldwe + n], n < ncols (layer 1: [64][F1], NB = 2; else columns 128..191 of [64][192])
    float* b1;                         // layer 1 only (layers 2-4: the bias gradient comes with dP_r)
    int nb_e;                          // 2 (layer 1) or 4
};
__global__ void __launch_bounds__(1024)
k_edge_acc_reduce(const float* __restrict__ partial, int n_wgs, EdgeAccOut O) {
    // blockIdx.x: 64 consecutive elements of a partial; 16 groups of 64 threads take every 16th workgroup... in order:
    // group g adds workgroups [g * per, (g + 1) * per) sequentially, the 16 sums are combined in order
    const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + t;
    const int per = (n_wgs + 15) / 16;
    const int w0 = grp * per, w1 = w0 + per < n_wgs ? w0 + per : n_wgs;
    float s = 0.0f;
    for (int w = w0; w < w1; ++w) s += partial[(size_t)w * EA_PART + e];
    __shared__ float red[16][64];
    red[grp][t] = s;
    __syncthreads();
    if (grp != 0) return;
    float tot = red[0][t];
#pragma unroll
    for (int g = 1; g < 16; ++g) tot += red[g][t];
    const int half = e / (EA_PART / 2), f = e % (EA_PART / 2);
    const int reg = f >> 6, lane = f & 63, i = lane & 15, q = lane >> 4;
    if (reg < 64) {
        const int mb = reg >> 4, nb = (reg >> 2) & 3, r = reg & 3;
        const int m = 4 * (4 * q + r) + mb;
        if (half == 0) {
            O.w2[m * H + 4 * i + nb] = tot;
        } else if (nb < O.nb_e) {
            const int n = O.nb_e * i + nb;
            if (n < O.ncols) O.we[(size_t)m * O.ldwe + n] = tot;
        }
        return;
    }
    // bias registers: the 16 lanes i of a (register, q) hold partial sums of the same hidden unit
    red[0][t] = tot;                     // (group 0 only from here on: no other thread touches red)
    __builtin_amdgcn_wave_barrier();
    if (i == 0) {
        float b = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) b += red[0][16 * q + j];
        const int br = reg - 64, unit = 16 * (br >> 2) + 4 * q + (br & 3);
        if (half == 0) O.b2[unit] = b;
        else if (O.b1) O.b1[unit] = b;
    }
}


// ------------------------------------------------------------------ round 4: two waves per SIMD
// kb_edge_acc runs ONE wave per SIMD (450 VGPR + 194 AGPR: two 64 x 64 accumulators, four split weight images' worth of
// LDS): a tile is a dependent chain of ~7 us with the matrix pipe 18 % and the vector ALU 50 % busy, and nobody to fill
// the gaps.  kb_edge_acc8 is the same arithmetic with what round 4 learnt in k_fused_bwd:
//   * the transposed products (W2^T dpre2, W_in^T G) read the forward's split images transposed (gemm_split_T): two images
//     in LDS instead of four;
//   * the weight-gradient accumulators are partitioned by output rows: a workgroup works in ROUNDS of eight tiles (one per
//     wave); after a round's tiles are staged, wave w adds the rows 16 (w & 3) .. of the product over the four tiles
//     4 (w >> 2) .. of the round -- 16 accumulator registers per product instead of 64, nothing to reduce over the waves
//     but the two tile halves, which go out as two partials per workgroup;
//   * bias sums ride along as one more MFMA per k step against a column of ones.
// 8 waves per workgroup, one workgroup per CU, <= 256 registers: two waves per SIMD.
//
// NW = 4 (option edge_acc = 3, the default): the same kernel as TWO workgroups of four waves per CU.  Eight waves of one
// workgroup take their barriers together, so a SIMD's two waves are always in the SAME phase (both in their GEMMs, then both
// in their weight-gradient products): nothing overlaps.  Two independent workgroups drift apart.
// Round 4, second half: the staged tensors are the fp16 pieces their GEMMs make (fused_bwd.h: fb_stage_pieces), the
// weight-gradient products run on the 16-bit matrix pipe from transposed reads with per-tile power-of-two scales
// (fb_outer16_issue / fb_outer16_consume) instead of fp32 MFMAs on the vector ALU: 12.2 -> 11.4 ms per launch at the
// 33.5 M-edge shard.  LDS per workgroup: 2 x 16 KB images + 4 x 8.5 KB staging = 67 KB.
constexpr int EA8_WAVES = 8;
constexpr int EA8_WAVE_STG = 2 * 16 * FB_SA;     // staging floats per wave: one operand pair at a time, as fp16 pieces
__host__ __device__ constexpr size_t ea8_lds_bytes(int nw) { return (size_t)(2 * SPLIT_WIMG + nw * EA8_WAVE_STG + 2 * nw) * 4; }
template <bool FIRST, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1)
kb_edge_acc8(const float* __restrict__ b_in, const float* __restrict__ b2g, const float* __restrict__ Ps,
             const float* __restrict__ Pr, const float* __restrict__ e_prev, const float* __restrict__ feat,
             const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s, const int32_t* __restrict__ rowptr,
             const float* __restrict__ DN, float* __restrict__ DE, int have_de, float* __restrict__ G,
             float* __restrict__ DA, float* __restrict__ partial /* [n_wgs][2][FB_PART] */,
             const float* __restrict__ img_in /*split image of W_e (FIRST: of W1, K padded to 32)*/,
             const float* __restrict__ img_2 /*split image of W2*/, int64_t n_edges, int n_wgs) {
    constexpr int NBE = FIRST ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wi = smem;                  // split image of W_in (FIRST: half of it is used)
    float* w2 = wi + SPLIT_WIMG;       // split image of W2
    float* stg = w2 + SPLIT_WIMG;      // [NW waves][2 arrays][2 fp16 pieces][16][FB_RH bytes] (fused_bwd.h: fb_stage_pieces)
    constexpr int EA8_STG = EA8_WAVE_STG;
    float* scl = stg + NW * EA8_STG;   // [NW][2]: the staged pair's 1 / (s_x s_y), 1 / s_x
    static_assert(NW == 4 || NW == 8, "one or two groups of four row quarters");
    for (int idx = threadIdx.x; idx < (FIRST ? SPLIT_WIMG / 2 : SPLIT_WIMG) / 4; idx += 64 * NW) st4(wi + 4 * idx, ld4(img_in + 4 * idx));
    for (int idx = threadIdx.x; idx < SPLIT_WIMG / 4; idx += 64 * NW) st4(w2 + 4 * idx, ld4(img_2 + 4 * idx));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    float* sa = stg + wave * EA8_STG;   // dpre2, then G
    float* sb = sa + 16 * FB_SA;        // h, then e_prev / features
    const int wq = wave & 3, whalf = wave >> 2;          // accumulator rows 16 wq .., tiles 4 whalf .. of a round
    const int64_t tiles = (n_edges + 15) >> 4;
    const int64_t stride = (int64_t)n_wgs * NW;
    f32x4 accW2[4], accWe[NBE], bs2 = f32x4{0.f, 0.f, 0.f, 0.f}, bs1 = bs2;
#pragma unroll
    for (int b = 0; b < 4; ++b) accW2[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NBE; ++b) accWe[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto clampk = [&](int64_t tt) { const int64_t k = 16 * tt + i; return k < n_edges ? k : n_edges - 1; };
    // the HBM stream of the next round (e_prev / features and the incoming message gradient) is requested a round ahead;
    // the gathered rows (P_s, P_r, dn: L2 hits) when the tile starts -- the SIMD's other wave covers that round trip
    f32x4 nbop[4], ndev[4];
    int64_t t0 = (int64_t)blockIdx.x * NW;
    {
        const int64_t k0 = clampk(t0 + wave < tiles ? t0 + wave : tiles - 1);
        if (FIRST) { nbop[0] = ld4(feat + k0 * FPAD + 4 * q); nbop[1] = ld4(feat + k0 * FPAD + 16 + 4 * q); }
        else load_tile64(nbop, e_prev, k0, H, q);
        if (have_de) load_tile64(ndev, DE, k0, H, q);
    }
    for (; t0 < tiles; t0 += stride) {                   // rounds: workgroup-uniform
        const int64_t t = t0 + wave;
        const bool active = t < tiles;                   // wave-uniform
        f32x4 g[4], bop[4];
        SplitScale s_ep{false, 1.0f, 1.0f}, s_g{false, 1.0f, 1.0f};
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { g[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; bop[mb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (active) {
            const int64_t k = 16 * t + i;
            const bool ok = k < n_edges;
            const int64_t kc = ok ? k : n_edges - 1;
            const int32_t s = send_s[kc], r = recv_s[kc];
            const int deg = rowptr[r + 1] - rowptr[r];
            f32x4 p1[4], dnv[4], dev[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) { bop[mb] = nbop[mb]; dev[mb] = ndev[mb]; }
            if (FIRST) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) p1[mb] = ld4(b_in + 16 * mb + 4 * q);
            } else {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    p1[mb] = ld4(Ps + (int64_t)s * H + 16 * mb + 4 * q) + ld4(Pr + (int64_t)r * H + 16 * mb + 4 * q);
            }
            load_tile64(dnv, DN, r, H, q);
            {   // next round's stream
                const int64_t tn = t + stride < tiles ? t + stride : t;
                const int64_t kn = clampk(tn);
                if (FIRST) { nbop[0] = ld4(feat + kn * FPAD + 4 * q); nbop[1] = ld4(feat + kn * FPAD + 16 + 4 * q); }
                else load_tile64(nbop, e_prev, kn, H, q);
                if (have_de) load_tile64(ndev, DE, kn, H, q);
            }
            const float inv = own_reg(1.0f / (float)(deg > 1 ? deg : 1));
            // ---- forward recompute: pre1, h = silu(pre1), pre2
            f32x4 p2[4], h[4], sg1[4];
            if (FIRST) {
                f32x4 b2[2] = {bop[0], bop[1]};
                f16x8 ph[1], pl[1];
                s_ep = gemm_split_keep<4, 1>(wi, b2, p1, lane, ph, pl);
            } else {
                f16x8 ph[2], pl[2];
                s_ep = gemm_split_keep<4, 2>(wi, bop, p1, lane, ph, pl);
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                sg1[mb] = sigmoid4(p1[mb]);
                h[mb] = p1[mb] * sg1[mb];
                p2[mb] = ld4(b2g + 16 * mb + 4 * q);
            }
            SplitScale s_h;
            {   // h's pieces go to the staging rows as its GEMM makes them
                f16x8 ph[2], pl[2];
                s_h = gemm_split_keep<4, 2>(w2, h, p2, lane, ph, pl);
                fb_stage_pieces<2>(sb, i, q, ph, pl);
            }
            // ---- de = dn[recv] / deg (+ gradient through the next layer's edge input); back through both Linears
            f32x4 d2[4], dh[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 de = dnv[mb] * inv;
                if (have_de) de += dev[mb];
                d2[mb] = de * dsilu_from_sigmoid(p2[mb], sigmoid4(p2[mb]));
                if (!ok) d2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};             // rows past the end contribute nothing to the products
                dh[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                f16x8 ph[2], pl[2];
                const SplitScale s_d2 = gemm_split_T_keep<2>(w2, d2, dh, lane, ph, pl);        // W2^T dpre2 from the forward's image
                fb_stage_pieces<2>(sa, i, q, ph, pl);
                if (lane == 0) { scl[2 * wave] = s_d2.inv_s * s_h.inv_s; scl[2 * wave + 1] = s_d2.inv_s; }
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) g[mb] = dh[mb] * dsilu_from_sigmoid(p1[mb], sg1[mb]);
            if (FIRST) {
                f32x4 da[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                f16x8 ph[2], pl[2];
                s_g = gemm_split_T_keep<1>(wi, g, da, lane, ph, pl);    // W1^T G
                if (ok) { st4(DA + k * FPAD + 4 * q, da[0]); st4(DA + k * FPAD + 16 + 4 * q, da[1]); }
            } else {
                f32x4 dep[4];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) dep[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                f16x8 ph[2], pl[2];
                s_g = gemm_split_T_keep<2>(wi, g, dep, lane, ph, pl);   // W_e^T G
                if (ok) store_tile64(DE, k, H, q, dep);
            }
            if (ok) store_tile64(G, k, H, q, g);
            if (!ok) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) bop[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- dW2 += dpre2 (x) h, db2: the round's tiles are staged; this wave's row quarter over its four tiles
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (t0 + 4 * whalf + tt < tiles) {
                const float* ta = stg + (4 * whalf + tt) * EA8_STG;
                const float c0 = scl[2 * (4 * whalf + tt)], c1 = scl[2 * (4 * whalf + tt) + 1];
                const unsigned lo_ = fb_tr_lane_offset(lane);
                FbOuterRegs<4> R;
                fb_outer16_issue<4>(fb_lds_addr(ta) + lo_ + 32u * (unsigned)wq, fb_lds_addr(ta + 16 * FB_SA) + lo_, R);
                fb_outer16_consume<4, true, 0>(R, accW2, bs2, c0, c1);
            }
        }
        lds_barrier();
        // ---- dW_e += G (x) e_prev (layer 1: dW1 += G (x) features, db1)
        if (active) {                      // (the pieces are made again from the live tiles, with the scales their GEMMs used: 40 vector
                                           //  instructions per tensor against 32 registers held across the first product)
            f16x8 ph[2], pl[2];
            split8(g[0] * s_g.s, g[1] * s_g.s, ph[0], pl[0]);          // (g of a row past the end is 0: dh = W2^T 0)
            split8(g[2] * s_g.s, g[3] * s_g.s, ph[1], pl[1]);
            fb_stage_pieces<2>(sa, i, q, ph, pl);
            split8(bop[0] * s_ep.s, bop[1] * s_ep.s, ph[0], pl[0]);
            if constexpr (FIRST) {
                const f16x8 h1[1] = {ph[0]}, l1[1] = {pl[0]};
                fb_stage_pieces<1>(sb, i, q, h1, l1);
            } else {
                split8(bop[2] * s_ep.s, bop[3] * s_ep.s, ph[1], pl[1]);
                fb_stage_pieces<2>(sb, i, q, ph, pl);
            }
            if (lane == 0) { scl[2 * wave] = s_g.inv_s * s_ep.inv_s; scl[2 * wave + 1] = s_g.inv_s; }
        }
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (t0 + 4 * whalf + tt < tiles) {
                const float* ta = stg + (4 * whalf + tt) * EA8_STG;
                const float c0 = scl[2 * (4 * whalf + tt)], c1 = scl[2 * (4 * whalf + tt) + 1];
                const unsigned lo_ = fb_tr_lane_offset(lane);
                FbOuterRegs<NBE> R;
                fb_outer16_issue<NBE>(fb_lds_addr(ta) + lo_ + 32u * (unsigned)wq, fb_lds_addr(ta + 16 * FB_SA) + lo_, R);
                fb_outer16_consume<NBE, FIRST, 0>(R, accWe, bs1, c0, c1);
            }
        }
        lds_barrier();
    }
    // ---- two partials per workgroup (tile halves 0 / 1), each wave its row quarter: [dW2 64 x 64 | dW_e 64 x 64 (64 x 32) | db2 | db1]
    float* dst = partial + ((size_t)blockIdx.x * (NW / 4) + whalf) * FB_PART;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int mrow = 16 * wq + 4 * q + r4;           // accumulator block nb holds columns 16 nb + i
#pragma unroll
        for (int nbx = 0; nbx < 4; ++nbx) dst[mrow * H + 16 * nbx + i] = accW2[nbx][r4];
        if constexpr (FIRST) {
            dst[H * H + mrow * FPAD + i] = accWe[0][r4];
            dst[H * H + mrow * FPAD + 16 + i] = accWe[1][r4];
        } else {
#pragma unroll
            for (int nbx = 0; nbx < 4; ++nbx) dst[H * H + mrow * H + 16 * nbx + i] = accWe[nbx][r4];
        }
        if (i == 0) {
            dst[2 * H * H + mrow] = bs2[r4];
            dst[2 * H * H + H + mrow] = FIRST ? bs1[r4] : 0.0f;
        }
    }
}

// Sum of kb_edge_acc8's partials ([n_parts][FB_PART], part order = workgroup order, fixed 16-group tree) into the gradient
// tensors.  Element e < H*H: dW2[m][n]; < 2 H*H: dW_e (layer 1: [64][FPAD] rows, columns < f1); then db2, db1.
__global__ void __launch_bounds__(1024)
k_edge_acc8_reduce(const float* __restrict__ partial, int n_parts, EdgeAccOut O) {
    const int t = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + t;
    const bool valid = e < FB_PART;
    const int per = (n_parts + 15) / 16;
    const int w0 = grp * per, w1 = w0 + per < n_parts ? w0 + per : n_parts;
    float s = 0.0f;
    if (valid)
        for (int w = w0; w < w1; ++w) s += partial[(size_t)w * FB_PART + e];
    __shared__ float red[16][64];
    red[grp][t] = s;
    __syncthreads();
    if (grp != 0 || !valid) return;
    float tot = red[0][t];
#pragma unroll
    for (int g = 1; g < 16; ++g) tot += red[g][t];
    if (e < H * H) {
        O.w2[e] = tot;
    } else if (e < 2 * H * H) {
        const int o = e - H * H;
        if (O.nb_e == 2) {                               // layer 1: [64][FPAD]
            const int mrow = o / FPAD, c = o % FPAD;
            if (o < H * FPAD && c < O.ncols) O.we[(size_t)mrow * O.ldwe + c] = tot;
        } else {
            O.we[(size_t)(o >> 6) * O.ldwe + (o & 63)] = tot;
        }
    } else if (e < 2 * H * H + H) {
        O.b2[e - 2 * H * H] = tot;
    } else if (O.b1 != nullptr) {
        O.b1[e - 2 * H * H - H] = tot;
    }
}

}  // namespace
