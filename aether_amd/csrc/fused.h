// fused.h -- the whole Aether step for one *group* of small graphs in ONE launch.
//
// A group is a contiguous node range [nb, ne) whose in-edges all have senders inside the range
// (one or several whole graphs; built by aether_graph_build), with at most FUSED_MAX_NODES nodes
// and FUSED_MAX_EDGES edges.  One workgroup of NW waves owns the group from the field query to the
// output; nothing but the inputs, the weights and the D output floats per node touches HBM:
//   * every edge message tile (16 edges x 64) stays in the owning wave's registers across the four
//     layers (it is the next layer's MFMA B operand as it stands);
//   * node state (x, n, P_s, P_r) and the per-layer edge weights live in LDS (padded rows);
//   * the mean over in-edges needs no workgroup barrier inside a layer: the wave parks its tile in
//     16 private LDS rows and multiplies it, transposed, by the tile's 0/1 receiver-segment matrix
//     on the matrix core (16 extra MFMAs, built once per tile from the graph structure); one
//     partial row per (receiver, tile) goes to LDS and the node phase adds a node's partial rows in
//     tile order.  Fixed order everywhere: deterministic, no atomics.  (A masked DPP butterfly was
//     measured 2-4x slower here: the kernel is VALU-issue-bound, not MFMA-bound.)
//   * node-level GEMMs are split along their output rows over the waves; their weights come from L2
//     in MFMA fragment shape (each is used once per group and layer), all issued before the barrier
//     that ends the edge phase.
// References: see common.h / streamed.h; the arithmetic per stage is identical to the streamed path.
#pragma once
#include "common.h"

namespace {

constexpr int FUSED_MAX_NODES = 32;          // two 16-node MFMA tiles
constexpr int FUSED_MAX_EDGES = 384;         // 24 tiles (N=20 fully connected: 380 edges)
constexpr int FUSED_MAX_TILES = FUSED_MAX_EDGES / 16;
constexpr int LDU = 2 * H + 8;               // padded LDS row for the 128-wide update hidden

// The edge MLP's two 64 x 64 contractions run as six bf16 MFMA terms on split operands (common.h, gemm_split); the
// weight images (3 x 8 KB each instead of 18 KB of padded fp32) fit every variant since the layer-1 features of the
// 17-24 tile one are built two rounds at a time.
template <int ROUNDS> constexpr bool fused_split_gemm() { return true; }
constexpr int FUSED_WIMG = SPLIT_WIMG;                   // floats of a split image of a 64 x 64 matrix (16 KB)
template <int NW, int ROUNDS> struct FusedLds {          // offsets in floats
    static constexpr int WSZ = fused_split_gemm<ROUNDS>() ? FUSED_WIMG : H * LDW;
    static constexpr int WA = 0;                                   // W_e  (layer 1: W1): [64][LDW] fp32 (ld LDF) | split image
    static constexpr int WB = WA + WSZ;                            // W2
    static constexpr int BIAS = WB + WSZ;                          // [128]      b1 | b2
    static constexpr int XBUF = BIAS + 2 * H;                      // [32][LDW]  x_{l-1} / x_l
    static constexpr int NBUF = XBUF + FUSED_MAX_NODES * LDW;      // [32][LDW]  n = x + mean
    static constexpr int PS = NBUF + FUSED_MAX_NODES * LDW;        // [32][LDW]  W_s x
    static constexpr int PR = PS + FUSED_MAX_NODES * LDW;          // [32][LDW]  W_r x + b1
    static constexpr int NINFO = PR + FUSED_MAX_NODES * LDW;       // [32][24]   NodeInfo records
    // one row per (receiver, tile).  A split workgroup walks its edges in two receiver-sorted runs (own senders,
    // then the partner's: see FusedWG), whose rows are kept apart by an offset of n: 2 * 32 + 16 rows for up to
    // 16 tiles; with 17-24 tiles (ROUNDS == 3) only unsplit workgroups are built: 32 + 24 rows.
    static constexpr int PART_ROWS = ROUNDS == 3 ? FUSED_MAX_NODES + FUSED_MAX_TILES : 2 * FUSED_MAX_NODES + 16;
    static constexpr int PART = NINFO + FUSED_MAX_NODES * 24;      // [PART_ROWS][LDW]  per-(receiver, tile) sums
    static constexpr int ARRIVED = PART + PART_ROWS * LDW;         // [4] ints: split mode, layer whose partner rows are in LDS
    static constexpr int SCRATCH = ARRIVED + 4;                    // aliased by the regions below
    static constexpr int FEAT_ROWS = 16 * (ROUNDS < 2 ? ROUNDS : 2);           // per wave: two rounds of features at a time
    static constexpr int SCRATCH_SIZE =
        NW * FEAT_ROWS * LDF > NW * 16 * LDST ? NW * FEAT_ROWS * LDF : NW * 16 * LDST;
    static constexpr int TOTAL = SCRATCH + SCRATCH_SIZE;
    static constexpr int FIELD_Z = SCRATCH;                            // [32][24]  p | v | emb
    static constexpr int FIELD_H1 = FIELD_Z + FUSED_MAX_NODES * 24;    // [32][32]
    static constexpr int FIELD_H2 = FIELD_H1 + FUSED_MAX_NODES * 32;   // [32][32]
    static constexpr int FIELD_F = FIELD_H2 + FUSED_MAX_NODES * 32;    // [32][4]
    static constexpr int FIELD_W = FIELD_F + FUSED_MAX_NODES * 4;      // field-net parameters (2,080 floats)
    static constexpr int FW0 = FIELD_W;                                // [32][22]
    static constexpr int FW2 = FW0 + 32 * 22;                          // [32][32]
    static constexpr int FW4 = FW2 + 32 * 32;                          // [3][32]
    static constexpr int FB = FW4 + 3 * 32;                            // b0[32] b2[32] b4[4]
    static constexpr int FEMB = FB + 68;                               // [3][16]
    static constexpr int FIELD_END = FEMB + 48;
    static constexpr int FEAT = SCRATCH;                               // [NW waves][FEAT_ROWS][LDF]
    static constexpr int WSTAGE = SCRATCH;                             // [NW waves][16][LDST] tile staging
    static constexpr int UBUF = SCRATCH;                               // [32][LDU]
    static constexpr int OBUF1 = SCRATCH;                              // [32][LDW]
    static constexpr int OBUF2 = SCRATCH + FUSED_MAX_NODES * LDW;      // [32][LDW]
    static_assert(FIELD_END <= TOTAL, "field scratch");
    static_assert(UBUF + FUSED_MAX_NODES * LDU <= TOTAL, "ubuf");
    static_assert(OBUF2 + FUSED_MAX_NODES * LDW <= TOTAL, "obuf");
    static_assert(TOTAL * 4 <= 160 * 1024, "LDS budget");
};

// Split (3 x bf16) images of the edge-MLP weights in global memory, in the exact order the kernel keeps them in LDS
// (stage_split4): per layer l = 1..4 image A (layer 1: W1 padded to K = 32, FUSED_WIMG / 2 floats; layers 2-4: W_e)
// and image B (W2).  k_prepare_weights (aether_hip.hip) writes them once per weight version; k_fused copies them with LDS-DMA.
// Round 4: the node phase's weights as split images too (its GEMMs moved from the fp32 MFMA to the matrix pipe): per layer
// W3 [128][64] (stage_split4<8, 2>), W4 [64][128] (<4, 4>), and the NEXT layer's W_s, W_r (<4, 2> each; layer 4: out_w0, out_w3).
constexpr int FUSED_NIMG_W3 = 2 * 8 * 2 * 64 * 4, FUSED_NIMG_W4 = 2 * 4 * 4 * 64 * 4, FUSED_NIMG_WS = 2 * 4 * 2 * 64 * 4;
constexpr int FUSED_NIMG_LAYER = FUSED_NIMG_W3 + FUSED_NIMG_W4 + 2 * FUSED_NIMG_WS;
constexpr int FUSED_WIMG_SET = 8 * FUSED_WIMG + 4 * FUSED_NIMG_LAYER;      // floats reserved (layer 1's image A uses half of its slot)
constexpr int FUSED_SPLIT_BLOCKS = 8 + 4 * 6;            // blocks of k_prepare_weights that write images (512 threads each)
__device__ __host__ constexpr int fused_wimg_offset(int layer, int which) { return ((layer - 1) * 2 + which) * FUSED_WIMG; }
// which: 0 W3, 1 W4, 2 W_s (next layer / out_w0), 3 W_r (next layer / out_w3)
__device__ __host__ constexpr int fused_nimg_offset(int layer, int which) {
    return 8 * FUSED_WIMG + (layer - 1) * FUSED_NIMG_LAYER +
           (which == 0 ? 0 : which == 1 ? FUSED_NIMG_W3 : which == 2 ? FUSED_NIMG_W3 + FUSED_NIMG_W4 : FUSED_NIMG_W3 + FUSED_NIMG_W4 + FUSED_NIMG_WS);
}

__device__ __forceinline__ void split_weights_block(const AetherParams& P, int f1, float* __restrict__ wimg, int block, int tid) {
    if (block >= 8) {                                     // node-phase images: six blocks per layer
        const int nbk = block - 8, layer = nbk / 6 + 1, part = nbk % 6;
        if (part < 2) {                                   // W3 rows 64 part ..: 1,024 float4
            const float* w3 = layer == 1 ? P.l1_upd_w0 : P.ln_upd_w0[layer - 2];
            float* img = wimg + fused_nimg_offset(layer, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + 512 * j, rr = 64 * part + (idx >> 4), cc = (idx & 15) * 4;
                stage_split4<8, 2>(img, rr, cc, ld4(w3 + (size_t)rr * H + cc));
            }
        } else if (part < 4) {                            // W4 [64][128], rows 32 (part - 2) ..
            const float* w4 = layer == 1 ? P.l1_upd_w2 : P.ln_upd_w2[layer - 2];
            float* img = wimg + fused_nimg_offset(layer, 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + 512 * j, rr = 32 * (part - 2) + (idx >> 5), cc = (idx & 31) * 4;
                stage_split4<4, 4>(img, rr, cc, ld4(w4 + (size_t)rr * (2 * H) + cc));
            }
        } else {                                          // W_s / W_r of the next layer (layer 4: out_w0 / out_w3)
            const float* src = layer < 4 ? P.ln_msg_w0[layer - 1] + (part == 4 ? 0 : H) : (part == 4 ? P.out_w0 : P.out_w3);
            const int ld = layer < 4 ? 3 * H : H;
            float* img = wimg + fused_nimg_offset(layer, part - 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = tid + 512 * j, rr = idx >> 4, cc = (idx & 15) * 4;
                stage_split4<4, 2>(img, rr, cc, ld4(src + (size_t)rr * ld + cc));
            }
        }
        return;
    }
    const int layer = block / 2 + 1, which = block & 1;
    float* img = wimg + fused_wimg_offset(layer, which);
    if (which == 0 && layer == 1) {                       // W1 [64][f1] -> K padded to 32: one float4 per thread
        const int r = tid >> 3, c0 = (tid & 7) * 4;
        f32x4 v;
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = c0 + b < f1 ? P.l1_msg_w0[r * f1 + c0 + b] : 0.0f;
        stage_split4<4, 1>(img, r, c0, v);
        return;
    }
    const float* src;
    int ld;
    if (which == 0) { src = P.ln_msg_w0[layer - 2] + 2 * H; ld = 3 * H; }             // W_e = W1[:, 128:192]
    else { src = layer == 1 ? P.l1_msg_w2 : P.ln_msg_w2[layer - 2]; ld = H; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + 512 * j, rr = idx >> 4, cc = (idx & 15) * 4;
        stage_split4<4, 2>(img, rr, cc, ld4(src + (size_t)rr * ld + cc));
    }
}

// Optional global copies of the intermediates (same layout as the streamed path's workspace), so
// that the parity tests and (later) the backward can read them.
struct FusedDebug {
    float* nodeinfo; float* x[5]; float* e[4];
    float* n[4]; float* ps[3]; float* pr[3]; float* feat;     // saved for the backward
    int* flags;             // [workgroups] split mode: layer whose P_s rows this workgroup has published (lives in the
                            // graph buffer: zero when no launch is in flight, every launch re-arms what it consumed)
    int* errword;           // host-mapped word: set when a bounded wait on the partner workgroup gave up
    const float* wimg;      // split weight images (k_split_weights); used by the split-GEMM variants
    float* stamps;          // [groups][FUSED_STAMPS] diagnostic build only
    StepExtras step;        // rollout: derived edge attributes, next velocity
};
constexpr int FUSED_STAMPS = 512;

// In-kernel phase stamps: only in the diagnostic build (-DAETHER_FUSED_STAMPS).  Thread 0 of each
// workgroup records the 100 MHz wall clock (microseconds since entry) at phase boundaries; lane 0 of
// each wave records shader-clock cycles around the pieces of its layer-2 tiles.  The stamps go to a
// buffer nothing else reads.
#ifdef AETHER_FUSED_STAMPS
#define FUSED_STAMP(id)                                                                          \
    do {                                                                                         \
        if (tid == 0 && blockIdx.x < 4096)                                                       \
            dbg.stamps[blockIdx.x * FUSED_STAMPS + (id)] = (float)(wall_clock64() - t_entry) * 0.01f; \
    } while (0)
#define FUSED_WSTAMP(layer_, r_, k_)                                                              \
    do {                                                                                         \
        if ((layer_) == 2 && lane == 0 && blockIdx.x < 4096 && wave < 16)                        \
            dbg.stamps[blockIdx.x * FUSED_STAMPS + 64 + wave * 24 + (r_) * 8 + (k_)] =            \
                (float)(__builtin_amdgcn_s_memtime() - c_entry);                                 \
    } while (0)
#else
#define FUSED_STAMP(id)
#define FUSED_WSTAMP(layer_, r_, k_)
#endif

// One workgroup's share of a group: it sees nodes [vb, ve) (whole graphs: every sender of its edges)
// and owns nodes [nb, ne) -- their in-edges, their node updates, their outputs.  Unsplit: own = visible.
// Split (two workgroups per group, when there are fewer groups than half the CUs): the partner owns
// the rest and the two exchange their P_s rows once per layer through global memory.
// A workgroup walks its m in-edges in a LOCAL order (lorder[eb + local] = offset into the receiver-sorted
// range): first the `na` edges whose sender it owns, then the edges from the partner's nodes, each run
// receiver-sorted (unsplit: na = m, the identity).  Tiles of the first run need nothing from the partner, so
// the forward starts with them while the partner's rows are in flight -- and the backward starts with the
// second run, whose sender-side sums the partner waits for.
struct FusedWG { int vb, ve, nb, ne, tile0, partner, na, eb, m, pad1, pad2, pad3; };   // eb, m: first in-edge (sorted position) and in-edge count of [nb, ne)
struct FusedTile { int wg, t; };             // workgroup descriptor index, tile index within the workgroup

// Local edge order of every workgroup (one block per workgroup; m <= FUSED_MAX_EDGES <= blockDim.x) and the
// local index ranges of every own node's in-edges in the two runs: nrange[node] = {a_beg, a_end, b_beg, b_end}.
__global__ void __launch_bounds__(512)
k_graph_lorder(FusedWG* __restrict__ wgdesc, const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
               const int32_t* __restrict__ perm, const int32_t* __restrict__ rowptr, int32_t* __restrict__ lorder,
               int4* __restrict__ ledge, int32_t* __restrict__ nrange) {
    __shared__ int wsum[2][8];
    __shared__ int cnt_a[FUSED_MAX_NODES + 1], cnt_b[FUSED_MAX_NODES + 1];
    const FusedWG wg = wgdesc[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int eb = rowptr[wg.nb], m = rowptr[wg.ne] - eb, n = wg.ne - wg.nb;
    const bool valid = tid < m;
    const bool split = wg.partner >= 0;
    const int snd = valid ? send_s[eb + tid] : -1;
    const bool own = valid && (!split || (snd >= wg.nb && snd < wg.ne));
    const bool oth = valid && !own;
    const unsigned long long bo = __ballot(own), bt = __ballot(oth);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) { wsum[0][wave] = __popcll(bo); wsum[1][wave] = __popcll(bt); }
    if (tid <= FUSED_MAX_NODES) { cnt_a[tid] = 0; cnt_b[tid] = 0; }
    __syncthreads();
    int base_o = 0, base_t = 0, na = 0;
    for (int w = 0; w < 8; ++w) {
        if (w < wave) { base_o += wsum[0][w]; base_t += wsum[1][w]; }
        na += wsum[0][w];
    }
    // ledge: the same order with everything the forward kernel's feature phase looks up per edge in one 16-byte record
    // {sorted position, sender, receiver, original edge}: one round trip instead of three dependent ones (k_fused, P2)
    if (valid) {
        const int pos = own ? base_o + __popcll(bo & below) : na + base_t + __popcll(bt & below);
        lorder[eb + pos] = tid;
        ledge[eb + pos] = make_int4(eb + tid, snd, recv_s[eb + tid], perm[eb + tid]);
    }
    if (tid == 0) wgdesc[blockIdx.x].na = na;
    // per-node run lengths (thread per own node, a handful of edges each), then a serial prefix
    if (tid < n) {
        int ca = 0, cb = 0;
        for (int k = rowptr[wg.nb + tid]; k < rowptr[wg.nb + tid + 1]; ++k) {
            const int sd = send_s[k];
            if (!split || (sd >= wg.nb && sd < wg.ne)) ++ca; else ++cb;
        }
        cnt_a[tid] = ca; cnt_b[tid] = cb;
    }
    __syncthreads();
    if (tid == 0) {
        int pa = 0, pb = na;
        for (int s = 0; s < n; ++s) {
            int32_t* o = nrange + 4 * (int64_t)(wg.nb + s);
            o[0] = pa; o[1] = pa + cnt_a[s]; o[2] = pb; o[3] = pb + cnt_b[s];
            pa += cnt_a[s]; pb += cnt_b[s];
        }
    }
}

// Per-tile structure of a workgroup's edge list in LOCAL order, built once with the graph: for every lane of
// the tile's wave the 0/1 column of the segment matrix it feeds to the matrix core (tsel) and the
// partial rows its four result registers go to (tdst, one byte each, 0xFF = none).  A segment = consecutive
// rows with one receiver inside one run; its partial row is receiver + tile (+ n in the second run).
__global__ void __launch_bounds__(64)
k_graph_tiles(const FusedTile* __restrict__ tdesc, const FusedWG* __restrict__ wgdesc,
              const int32_t* __restrict__ recv_s, const int32_t* __restrict__ rowptr,
              const int32_t* __restrict__ lorder, uint32_t* __restrict__ tsel, uint32_t* __restrict__ tdst) {
    const FusedTile T = tdesc[blockIdx.x];
    const FusedWG wg = wgdesc[T.wg];
    const int eb = rowptr[wg.nb], m = rowptr[wg.ne] - eb, n = wg.ne - wg.nb;
    const int na = wg.na;
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    const int local = 16 * T.t + i;
    const bool valid = local < m;
    const int rcv = valid ? recv_s[eb + lorder[eb + local]] - wg.nb : -1;
    // smask bit j: row j starts a new segment; segment ids count up in row order; padding rows belong to no segment.
    const int prev = __shfl_up(rcv, 1, 16);
    const unsigned smask = (unsigned)__ballot(q == 0 && i > 0 && (rcv != prev || local == na)) & 0xFFFFu;
    const unsigned vmask = (unsigned)__ballot(q == 0 && valid) & 0xFFFFu;
    unsigned sb = 0, dp = 0;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int edge = 4 * s4 + q;
        const int seg_of_edge = __popc(smask & ((2u << edge) - 1u));
        if (((vmask >> edge) & 1u) && seg_of_edge == i) sb |= 1u << s4;
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int seg = 4 * q + r4;
        unsigned mm = smask;                 // first row of segment `seg`: its start bit (segment 0: row 0)
        int s0 = 0;
        bool exists = true;
        for (int t = 0; t < seg; ++t) {
            if (mm == 0) { exists = false; break; }
            s0 = __ffs(mm) - 1;
            mm &= mm - 1;
        }
        const int node = __shfl(rcv, (lane & 48) + s0);           // owner-local receiver of row s0
        const int second = (wg.partner >= 0 && 16 * T.t + s0 >= na) ? n : 0;
        const unsigned row = (exists && ((vmask >> s0) & 1u)) ? (unsigned)(node + T.t + second) : 0xFFu;
        dp |= row << (8 * r4);
    }
    tsel[(size_t)blockIdx.x * 64 + lane] = sb;
    tdst[(size_t)blockIdx.x * 64 + lane] = dp;
}

// The same structure for the whole receiver-sorted edge list (tile t = sorted positions 16t..16t+15),
// one word per lane, for the streamed edge kernels (streamed.h::tile_receiver_sums):
// bits 0-3 segment-matrix column; per result register r4: first row of segment 4q+r4 (4 bits), valid (1 bit).
__global__ void __launch_bounds__(64)
k_graph_gtiles(const int32_t* __restrict__ recv_s, int64_t n_edges, uint32_t* __restrict__ gsel) {
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    const int64_t k = (int64_t)blockIdx.x * 16 + i;
    const bool valid = k < n_edges;
    const int rcv = valid ? recv_s[k] : -1;
    const int prev = __shfl_up(rcv, 1, 16);
    const unsigned smask = (unsigned)__ballot(q == 0 && i > 0 && rcv != prev) & 0xFFFFu;
    const unsigned vmask = (unsigned)__ballot(q == 0 && valid) & 0xFFFFu;
    unsigned w = 0;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int edge = 4 * s4 + q;
        const int seg_of_edge = __popc(smask & ((2u << edge) - 1u));
        if (((vmask >> edge) & 1u) && seg_of_edge == i) w |= 1u << s4;
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int seg = 4 * q + r4;
        unsigned mm = smask;
        int s0 = 0;
        bool exists = true;
        for (int t = 0; t < seg; ++t) {
            if (mm == 0) { exists = false; break; }
            s0 = __ffs(mm) - 1;
            mm &= mm - 1;
        }
        const unsigned ok = (exists && ((vmask >> s0) & 1u)) ? 16u : 0u;
        w |= ((unsigned)s0 | ok) << (4 + 5 * r4);
    }
    gsel[(size_t)blockIdx.x * 64 + lane] = w;
}

#ifndef AETHER_R3_DEFER
#define AETHER_R3_DEFER 0      // diagnostic: 1 = three-tile variants request the node-phase weights after the tile loop (spill-free, slower)
#endif
template <int D, int NW, int ROUNDS, bool KEEP>
__global__ void __launch_bounds__(NW * 64)
k_fused(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel,
        const float* __restrict__ charges, const float* __restrict__ edge_attr_orig,
        const int32_t* __restrict__ perm, const int32_t* __restrict__ send_s,
        const int32_t* __restrict__ recv_s, const int32_t* __restrict__ rowptr,
        const FusedWG* __restrict__ wgdesc, const uint32_t* __restrict__ tsel,
        const uint32_t* __restrict__ tdst, const int4* __restrict__ ledge,
        const int32_t* __restrict__ nrange, FusedDebug dbg, float* __restrict__ out) {
    constexpr bool keep = KEEP;      // inference build carries none of the save-for-backward stores
    using NI = NodeInfo<D>;
    using L = FusedLds<NW, ROUNDS>;
    constexpr int THREADS = NW * 64;
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    constexpr int FIN = 2 * D + 16;
    static_assert(NW == 8, "waves 0-7 run the node phase (12 and 16 waves per workgroup were measured: DESIGN.md 4.1)");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wA = smem + L::WA;
    float* wB = smem + L::WB;
    float* bias = smem + L::BIAS;
    float* xbuf = smem + L::XBUF;
    float* nbuf = smem + L::NBUF;
    float* psb = smem + L::PS;
    float* prb = smem + L::PR;
    float* ninfo = smem + L::NINFO;
    float* part = smem + L::PART;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const FusedWG wg = wgdesc[blockIdx.x];
    const int vb = wg.vb, nv = wg.ve - wg.vb;            // visible nodes (slots of ninfo / psb)
    const int nb = wg.nb, ne = wg.ne;                    // own nodes (slots of xbuf / nbuf / prb / part)
    const int n = ne - nb, off = nb - vb;
    const int eb = wg.eb, m = wg.m;                      // (= rowptr[nb], rowptr[ne] - rowptr[nb]: kept with the descriptor)
    const int n_tiles = (m + 15) >> 4;
    const int na = wg.na;                                // edges [0, na) of the local order have own senders
    volatile int* arrived = reinterpret_cast<volatile int*>(smem + L::ARRIVED);
    if (tid == 0) { arrived[0] = 0; arrived[1] = 0; }    // ordered before any use by the prologue's barriers
#ifdef AETHER_FUSED_STAMPS
    const unsigned long long t_entry = wall_clock64();
    const unsigned long long c_entry = __builtin_amdgcn_s_memtime();
#endif
    FUSED_STAMP(0);

    // ---------------------------------------------------------------- P0: layer-1 weights, loads only
    // (W1 [64][F1] zero padded to [64][LDF], W2, biases); they land in LDS after the prologue, so their
    // latency hides behind the field net.
    constexpr bool SPLITG = fused_split_gemm<ROUNDS>();
    constexpr int P0A = SPLITG ? 1 : (H * LDF + THREADS - 1) / THREADS;
    constexpr int P0B = SPLITG ? 1 : (H * H / 4 + THREADS - 1) / THREADS;
    float p0a[P0A];
    f32x4 p0b[P0B];
    float p0bias = 0.0f;
    // Split-GEMM variants: a layer's two weight images are copied global -> LDS by LDS-DMA, one 1 KiB fragment per wave
    // instruction (images are lane-linear in both places), no registers, no VALU; `first` skips the unused half of
    // layer 1's image A.  The caller waits (vmcnt) before the barrier that precedes the first read.
    auto dma_images = [&](int layer_) {
        const float* ga = dbg.wimg + fused_wimg_offset(layer_, 0);
        const float* gb = dbg.wimg + fused_wimg_offset(layer_, 1);
        const int na_frag = layer_ == 1 ? 8 : 16, total = na_frag + 16;       // 1 KiB fragments: 2 terms x 4 (x 2)
        for (int f = wave; f < total; f += NW) {
            const float* src = f < na_frag ? ga + f * 256 : gb + (f - na_frag) * 256;
            float* dst = f < na_frag ? wA + f * 256 : wB + (f - na_frag) * 256;
            __builtin_amdgcn_global_load_lds(src + lane * 4, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    if constexpr (SPLITG) {
        dma_images(1);
    } else {
#pragma unroll
        for (int j = 0; j < P0A; ++j) {
            const int idx = tid + THREADS * j, r = idx / LDF, c = idx - r * LDF;
            p0a[j] = (idx < H * LDF && c < F1) ? P.l1_msg_w0[r * F1 + c] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < P0B; ++j) {
            const int idx = tid + THREADS * j;
            if (idx < H * H / 4) p0b[j] = ld4(P.l1_msg_w2 + (size_t)(idx >> 4) * H + (idx & 15) * 4);
        }
    }
    if (tid < 2 * H) p0bias = tid < H ? P.l1_msg_b0[tid] : P.l1_msg_b2[tid - H];
    FUSED_STAMP(1);

    // ---------------------------------------------------------------- edge indices of P2, requested during P1
    // P2 needs, per lane, dependent rounds of global loads before it can read a node record: in rounds 1 - 4 (first half)
    // row pointers -> local order -> sender / receiver / original edge -> edge attributes, issued in P2 (1.8 us per workgroup).
    // Now the descriptor carries the row pointers, the graph one 16-byte record per edge in local order (ledge,
    // k_graph_lorder), and every wave asks for its records and attributes during P1 -- only two of the eight waves work
    // there (<= 32 visible nodes), behind their own inputs.
    constexpr int FR = ROUNDS < 2 ? ROUNDS : 2;              // rounds of the first feature pass (lane >> 4)
    const int f_local = 16 * (NW * (lane >> 4) + wave) + (lane & 15);
    const bool f_have = lane < 16 * FR && f_local < m;       // this lane builds the features of edge f_local
    int4 f_e = make_int4(0, 0, 0, 0);                        // {sorted position, sender, receiver, original edge}
    f32x2 f_ea = {0.0f, 0.0f};                               // its two edge attributes (q_i q_j: the first only)
    int4 t_e[ROUNDS];                                        // per (round, lane i): the tile's edge i
    unsigned t_sel[ROUNDS], t_dst[ROUNDS];
    auto index_stage_a = [&]() {
        if (f_have) f_e = ledge[eb + f_local];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int tile = NW * r + wave, local = 16 * tile + i;
            t_e[r] = m > 0 ? ledge[eb + (local < m ? local : 0)] : make_int4(0, vb, nb, 0);
            const bool have = tile < n_tiles;
            t_sel[r] = have ? tsel[(size_t)(wg.tile0 + tile) * 64 + lane] : 0u;
            t_dst[r] = have ? tdst[(size_t)(wg.tile0 + tile) * 64 + lane] : 0xFFFFFFFFu;
        }
    };
    // (returns its result: written through the capture, the two floats were kept in scratch memory -- 8 bytes per thread
    // stored and re-loaded, + 1 MB of WRITE_SIZE per launch in the first version)
    auto index_stage_b = [&]() -> f32x2 {
        f32x2 v = {0.0f, 0.0f};
        if (f_have) {
            if (dbg.step.qattr) {          // main.py:243-246: q_i q_j (the distance is computed with the features)
                v[0] = dbg.step.qattr[f_e.y] * dbg.step.qattr[f_e.z];
            } else {
                const float* ea = edge_attr_orig + 2 * (int64_t)f_e.w;
                v[0] = ea[0]; v[1] = ea[1];
            }
        }
        return v;
    };

    // ---------------------------------------------------------------- P1: field net, frames, x0 on the matrix core
    // aether.py:108-134 (field), geometry.py:7-73 + aether.py:33-50 (frames), locs.py:214-218 (x0).
    // Wave t owns the visible nodes 16t .. 16t+15 as the 16 columns of its MFMA tiles; the three
    // Linear layers chain in accumulator layout, lanes q == 0 end up with the node's force and build
    // its frame, and x0 = res(rel_feat) is one more tile product: no workgroup barrier in between,
    // every global load (inputs, 2,080 field parameters, res weights) issued up front.
    {
        const int vtiles = (nv + 15) >> 4;
        if (wave < vtiles) {
            const int node = 16 * wave + i;                    // visible slot
            const bool live = node < nv;
            const int64_t g = vb + (live ? node : 0);
            float pz[D], vz[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { pz[d] = x[g * D + d]; vz[d] = vel[g * D + d]; }
            const float ch = charges[g];
            // layer-1 operands: B = z[k][node], A = W0[16mb + i][k], k = 4s + q (FIN = 2D + 16 <= 24)
            float zc[6][3], a1[2][6];
#pragma unroll
            for (int s4 = 0; s4 < 6; ++s4) {
                const int k = 4 * s4 + q;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    zc[s4][c] = (k >= 2 * D && k < FIN) ? P.field_emb[c * 16 + (k - 2 * D)] : 0.0f;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) a1[mb][s4] = k < FIN ? P.field_w0[(16 * mb + i) * FIN + k] : 0.0f;
            }
            f32x4 w2f[2][2], w4f[2], acc1[2], acc2[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                acc1[mb] = ld4(P.field_b0 + 16 * mb + 4 * q);
                acc2[mb] = ld4(P.field_b2 + 16 * mb + 4 * q);
#pragma unroll
                for (int a = 0; a < 2; ++a) w2f[mb][a] = ld4(P.field_w2 + (16 * mb + i) * 32 + 16 * a + 4 * q);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
                w4f[a] = i < D ? ld4(P.field_w4 + i * 32 + 16 * a + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 acc3 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc3[r] = (4 * q + r < D) ? P.field_b4[4 * q + r] : 0.0f;
            // x0 operands: A = W_res[16mb + i][D + k], k = 4s + q < 2D (the first D inputs of rel_feat are zero)
            float ar[4][2];
            f32x4 acc0[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                acc0[mb] = ld4(P.l1_res_b + 16 * mb + 4 * q);
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4)
                    ar[mb][s4] = 4 * s4 + q < 2 * D ? P.l1_res_w[(16 * mb + i) * 3 * D + D + 4 * s4 + q] : 0.0f;
            }
            index_stage_a();                                    // behind this wave's own inputs (in-order returns: they do not wait for it)
            long ci = (long)(ch + 1.0f);                        // charge_to_index: (q + 1).long(), aether.py:127-129
            ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
#pragma unroll
            for (int s4 = 0; s4 < 6; ++s4) {
                const int k = 4 * s4 + q;
                float zk = ci == 0 ? zc[s4][0] : (ci == 1 ? zc[s4][1] : zc[s4][2]);
#pragma unroll
                for (int d = 0; d < D; ++d) { zk = k == d ? pz[d] : zk; zk = k == D + d ? vz[d] : zk; }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc1[mb] = mfma16(a1[mb][s4], zk, acc1[mb]);
            }
            f32x4 hh[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) hh[mb] = silu4(acc1[mb]);
            f_ea = index_stage_b();                                    // (the records are back by now; measured: asked for earlier, nothing gained)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) acc2[mb] = mfma16(w2f[mb][a][b], hh[a][b], acc2[mb]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) hh[mb] = silu4(acc2[mb]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc3 = mfma16(w4f[a][b], hh[a][b], acc3);
            // lanes q == 0: acc3[d] = force component d of node i -> frame + NodeInfo record
            if (q == 0 && live) {
                float f[D], R[D][D], cv[D], cf[D];
#pragma unroll
                for (int d = 0; d < D; ++d)
                    f[d] = dbg.step.ext_field ? dbg.step.ext_field[(int64_t)(vb + node) * D + d] : acc3[d];
                node_frame<D>(vz, f, R, cv, cf);
                float* ni = ninfo + node * 24;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    ni[NI::P + d] = pz[d]; ni[NI::V + d] = vz[d]; ni[NI::F + d] = f[d];
                    ni[NI::CV + d] = cv[d]; ni[NI::CF + d] = cf[d];
#pragma unroll
                    for (int e = 0; e < D; ++e) ni[NI::R + d * D + e] = R[d][e];
                }
                if (keep && node >= off && node < off + n) {
                    float* gni = dbg.nodeinfo + (int64_t)(vb + node) * NI::STRIDE;
#pragma unroll
                    for (int t = 0; t < NI::STRIDE; ++t) gni[t] = t < NI::CF + D ? ni[t] : 0.0f;
                }
            }
            __builtin_amdgcn_wave_barrier();
            // x0 = W_res[:, D:] [cv | cf] + b (own nodes only; cv, cf are contiguous in the record)
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
                const int k = 4 * s4 + q;
                const float rk = (live && k < 2 * D) ? ninfo[node * 24 + NI::CV + k] : 0.0f;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc0[mb] = mfma16(ar[mb][s4], rk, acc0[mb]);
            }
            const int own = node - off;
            if (own >= 0 && own < n) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    st4(xbuf + own * LDW + 16 * mb + 4 * q, acc0[mb]);
                    if (keep) st4(dbg.x[0] + (int64_t)(nb + own) * H + 16 * mb + 4 * q, acc0[mb]);
                }
            }
        } else {
            index_stage_a();
            f_ea = index_stage_b();
            // rows of unused node slots are zero
            for (int idx = tid - 64 * vtiles; idx < (FUSED_MAX_NODES - n) * (H / 4); idx += THREADS - 64 * vtiles)
                st4(xbuf + (n + (idx >> 4)) * LDW + (idx & 15) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        // P0, second half: the layer-1 weights go to LDS
        if constexpr (SPLITG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's LDS-DMA fragments have landed
        } else {
#pragma unroll
            for (int j = 0; j < P0A; ++j) {
                const int idx = tid + THREADS * j;
                if (idx < H * LDF) wA[idx] = p0a[j];
            }
#pragma unroll
            for (int j = 0; j < P0B; ++j) {
                const int idx = tid + THREADS * j;
                if (idx < H * H / 4) st4(wB + (idx >> 4) * LDW + (idx & 15) * 4, p0b[j]);
            }
        }
        if (tid < 2 * H) bias[tid] = p0bias;
        lds_barrier();
    }
    FUSED_STAMP(2);

    // ---------------------------------------------------------------- P2: edge features -> B operands
    // Wave w owns tiles {w, w+NW, ..}.  Per tile, 16 lanes each build the features of one edge into
    // the wave's scratch rows; then every lane reads its B fragments back.  Per-tile constants of
    // the graph structure (sender / receiver slot, receiver-segment index) are computed once here.
    int sl[ROUNDS], rl[ROUNDS];              // sender slot (visible numbering), receiver slot (own numbering)
    unsigned selbits[ROUNDS];                // bit s4: S[seg = i][edge = 4*s4 + q] of the tile
    unsigned destpack[ROUNDS];               // byte r4: partial row of segment 4q + r4, 0xFF = none
    f32x4 e[ROUNDS][4];                      // message tiles, MFMA accumulator layout
    int ke[KEEP ? ROUNDS : 1];               // KEEP: receiver-sorted position of the lane's edge (row of the saved tensors)
    {
        float* scratch = smem + L::FEAT + wave * (L::FEAT_ROWS * LDF);
#pragma unroll
        for (int r0 = 0; r0 < ROUNDS; r0 += 2) {               // two rounds (32 scratch rows) per pass
            constexpr int PASS = 2;
            const int nr_pass = ROUNDS - r0 < PASS ? ROUNDS - r0 : PASS;
            if (lane < 16 * nr_pass) {
                const int r = r0 + (lane >> 4), ii = lane & 15;
                const int local = 16 * (NW * r + wave) + ii;
                float o[FPAD];
                if (local < m) {
                    // position in the receiver-sorted edge list, sender, receiver (first pass: requested during P1)
                    const int4 le = r0 == 0 ? f_e : ledge[eb + local];
                    const int k = le.x, ks = le.y, kr = le.z;
                    const float* nj = ninfo + (ks - vb) * 24;
                    const float* nr = ninfo + (kr - vb) * 24;
                    float njl[NI::STRIDE], nrl[NI::STRIDE];
#pragma unroll
                    for (int t = 0; t < NI::STRIDE; ++t) { njl[t] = nj[t]; nrl[t] = nr[t]; }
                    float eal[2];
                    if (dbg.step.qattr) {          // main.py:243-246: [q_i q_j, sqrt(sum((x_i - x_j)^2))]
                        float d2 = 0.0f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const float df = njl[NI::P + d] - nrl[NI::P + d];
                            d2 += df * df;
                        }
                        eal[0] = r0 == 0 ? f_ea[0] : dbg.step.qattr[ks] * dbg.step.qattr[kr];
                        eal[1] = sqrtf(d2);
                    } else if (r0 == 0) {
                        eal[0] = f_ea[0]; eal[1] = f_ea[1];
                    } else {
                        const float* ea = edge_attr_orig + 2 * (int64_t)le.w;
                        eal[0] = ea[0]; eal[1] = ea[1];
                    }
                    edge_features<D>(njl, nrl, eal, o);
                } else {
#pragma unroll
                    for (int t = 0; t < FPAD; ++t) o[t] = 0.0f;
                }
#pragma unroll
                for (int t = 0; t < FPAD; t += 4)
                    st4(scratch + lane * LDF + t, f32x4{o[t], o[t + 1], o[t + 2], o[t + 3]});
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int rr = 0; rr < PASS; ++rr) {
                const int r = r0 + rr;
                if (r < ROUNDS) {
                    e[r][0] = ld4(scratch + (16 * rr + i) * LDF + 4 * q);   // features as bop[0..1]
                    e[r][1] = ld4(scratch + (16 * rr + i) * LDF + 16 + 4 * q);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this pass's features sit in registers ...
            __builtin_amdgcn_wave_barrier();                       // ... before the next pass overwrites the rows
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            e[r][2] = f32x4{0.f, 0.f, 0.f, 0.f};
            e[r][3] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int local = 16 * (NW * r + wave) + i;
            sl[r] = t_e[r].y - vb;                             // (requested during P1)
            rl[r] = t_e[r].z - nb;
            if constexpr (KEEP) {
                ke[r] = local < m ? t_e[r].x : -1;
                // the saved feature rows, from the B-operand registers: 16 rows x 64 contiguous bytes per store and all 64 lanes
                // (written by the lane that built the row they went out as eight 16-byte pieces per lane, half the wave idle)
                if (ke[r] >= 0) {
                    st4(dbg.feat + (int64_t)ke[r] * FPAD + 4 * q, e[r][0]);
                    st4(dbg.feat + (int64_t)ke[r] * FPAD + 16 + 4 * q, e[r][1]);
                }
            }
            selbits[r] = t_sel[r];
            destpack[r] = t_dst[r];
        }
        lds_barrier();       // feature scratch (aliases SCRATCH) is dead from here on
    }
    FUSED_STAMP(3);

    // node-sum ownership (step 1 of the node phase): thread -> (node slot, 4 columns).  The in-edge
    // sum of a node arrives as per-(receiver, tile) partial rows `part[node + tile]`.
    // Two runs of tiles per node (own senders / partner's senders, see FusedWG); the second run's rows are offset by n.
    const int aslot = (tid >> 4) & (FUSED_MAX_NODES - 1), ac4 = (tid & 15) * 4;
    int at0 = 0, at1 = 0, bt0 = 0, bt1 = 0;
    float adeg = 1.0f;
    if (aslot < n) {
        const int4 nr = *reinterpret_cast<const int4*>(nrange + 4 * (int64_t)(nb + aslot));
        at0 = nr.x >> 4;
        at1 = nr.y > nr.x ? ((nr.y - 1) >> 4) + 1 : at0;
        bt0 = (nr.z >> 4) + n;
        bt1 = nr.w > nr.z ? ((nr.w - 1) >> 4) + 1 + n : bt0;
        const int deg = (nr.y - nr.x) + (nr.w - nr.z);
        adeg = (float)(deg > 1 ? deg : 1);
    }

    // next layer's W_s / W_r fragments (after layer 4: out_w0 / out_w3): fp16 x 2 pieces of the wave's row block, k blocks 0, 1
    f16x8 wsh[2], wsl[2], wrh[2], wrl[2];
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        // ------------------------------------------------------------ edge tiles (locs.py:227-238)
        // No workgroup barrier in here.  A wave with K tiles runs them through one branch-free block
        // (front = first Linear + SiLU, back = second Linear + SiLU + per-receiver sums) so that the
        // compiler can overlap one tile's VALU / LDS tail with the next tile's MFMAs.
        const float* w3 = layer == 1 ? P.l1_upd_w0 : P.ln_upd_w0[layer - 2];
        const float* b3 = layer == 1 ? P.l1_upd_b0 : P.ln_upd_b0[layer - 2];
        const float* w4 = layer == 1 ? P.l1_upd_w2 : P.ln_upd_w2[layer - 2];
        const float* b4 = layer == 1 ? P.l1_upd_b2 : P.ln_upd_b2[layer - 2];
        // work split: step 2 has 16 (row block, node tile) units, step 3 has 8, step 4 has 16
        const int mb2 = wave & 7;                         // step 2: rows 16*mb2.. of the 128
        const int mb3 = wave & 3, tn3 = (wave >> 2) & 1;  // steps 3, 4: rows 16*mb3.. of node tile tn3
        const bool act3 = wave < 8;                       // steps 2 - 4 / out MLP: waves 0-7
        // step 4 with one node tile (split mode): waves 0-3 compute P_s, waves 4-7 P_r, tile 0
        const bool one_tile = n <= 16;
        const int tn4 = one_tile ? 0 : tn3;
        const bool do_s = act3 && (one_tile ? wave < 4 : true);
        const bool do_r = act3 && (one_tile ? wave >= 4 : true);
        // Every L2 load of the node phase is issued BEFORE the wave's last edge tile, so that the
        // ~130 KB of weights a workgroup needs per layer (all 256 workgroups ask at the same moment)
        // stream in under the tile's MFMAs: next layer's edge weights (W_e = W1[:, 128:192], W2: they
        // go to LDS once every wave has left the edge tiles), then W3 / W4 / next-layer W_s, W_r fragments.
        constexpr int STG = SPLITG ? 1 : (H * H / 4 + THREADS - 1) / THREADS;      // float4 per thread per staged matrix
        f16x8 w3h[2], w3l[2], w4h[4], w4l[4];              // the wave's row block of W3 (k = 64) and of W4 (k = 128), split pieces
        f32x4 stA[STG], stB[STG];
        float b2n = 0.0f;
        // layer 4 has no next edge layer: wsv / wrv carry the out-MLP fragments (out_w0, out_w3) instead
        // part 0: staged matrices; 1: W3; 2, 3: W4 halves; 4: W_s / W_r.  A wave spreads the parts over
        // its last tile (a burst of ~20 loads per wave blocks at issue until the L2 returns drain).
        auto issue_loads = [&](int part) {
            if (part == 0 && layer < 4) {
                if constexpr (!SPLITG) {
                    const float* w1n = P.ln_msg_w0[layer - 1];
#pragma unroll
                    for (int j = 0; j < STG; ++j) {
                        const int idx = tid + THREADS * j, rr = idx >> 4, cc = (idx & 15) * 4;
                        if (idx < H * H / 4) {
                            stA[j] = ld4(w1n + (size_t)rr * (3 * H) + 2 * H + cc);
                            stB[j] = ld4(P.ln_msg_w2[layer - 1] + (size_t)rr * H + cc);
                        }
                    }
                }
                if (tid < H) b2n = P.ln_msg_b2[layer - 1][tid];
            }
            // (round 4: fragments of the prepared fp16 x 2 images, fused_nimg_offset -- same bytes per weight as fp32)
            if (part == 1 && act3) load_split_frags<8, 2>(dbg.wimg + fused_nimg_offset(layer, 0), mb2, lane, w3h, w3l);
            if ((part == 2 || part == 3) && act3 && 16 * tn3 < n) {
                const f16x8* w = reinterpret_cast<const f16x8*>(dbg.wimg + fused_nimg_offset(layer, 1)) +
                                 __builtin_amdgcn_readfirstlane(mb3) * (4 * 64);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int kb = 2 * (part - 2) + kk;
                    w4h[kb] = w[kb * 64 + lane];
                    w4l[kb] = w[4 * 4 * 64 + kb * 64 + lane];
                }
            }
            if (part == 4 && (layer < 4 ? true : act3)) {
                if (layer == 4 || do_s) load_split_frags<4, 2>(dbg.wimg + fused_nimg_offset(layer, 2), mb3, lane, wsh, wsl);
                if (layer == 4 || do_r) load_split_frags<4, 2>(dbg.wimg + fused_nimg_offset(layer, 3), mb3, lane, wrh, wrl);
            }
        };
        // Split mode, layers 2-4: the partner workgroup's P_s rows are in flight when the edge phase starts.  The
        // workgroup walks the tiles whose senders it owns first (local order, FusedWG); the LAST wave -- it has the
        // fewest tiles, and its only one needs the partner -- polls the partner's flag, copies the rows into LDS and
        // sets an LDS word; a wave waits on that word only before its first tile with a partner sender.  No
        // workgroup barrier, and nobody waits while there is work that does not need the rows.
        const bool xch = wg.partner >= 0 && layer > 1;
        auto front_gemm = [&](int r, f32x4 (&acc)[4], bool last) {
            if (last) issue_loads(0);
            if (layer == 1) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = ld4(bias + 16 * mb + 4 * q);
                f32x4 bop[2] = {e[r][0], e[r][1]};
                if constexpr (SPLITG) gemm_split<4, 1>(wA, bop, acc, lane);
                else gemm_tile<4, 2>(wA, LDF, bop, acc, i, q);
            } else {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    acc[mb] = ld4(prb + rl[r] * LDW + 16 * mb + 4 * q) + ld4(psb + sl[r] * LDW + 16 * mb + 4 * q);
                if constexpr (SPLITG) gemm_split<4, 2>(wA, e[r], acc, lane);
                else gemm_tile<4, 4>(wA, LDW, e[r], acc, i, q);
            }
        };
        auto receive_partner_rows = [&]() {
            // second half of the hand-off (first half: end of the previous node phase), run by ONE wave: lane 0
            // polls the partner's flag (bounded: a wait that gives up reports through dbg.errword), then the wave
            // reads the partner's rows with sc1 (L1-bypassing) 16-byte buffer loads -- the only loads of those
            // bytes in this launch -- into LDS and publishes them to the other waves through an LDS word
            // (cdna_hip_programming.md Guideline 16: "the other waves load after ... an LDS word it then sets").
            if (lane == 0) {
                unsigned spins = 0;
                unsigned long long t_first = 0;
                while (__hip_atomic_load(dbg.flags + wg.partner, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < layer - 1) {
                    __builtin_amdgcn_s_sleep(2);
                    // bounded by TIME (5 s of the 100 MHz wall clock), not by a spin count: when the device is shared
                    // (another process, another stream) the partner may legitimately be scheduled late -- a count of
                    // 2^22 polls gave up in a two-process rehearsal on one GPU
                    if ((++spins & 1023u) == 0) {
                        const unsigned long long now = wall_clock64();
                        if (t_first == 0) t_first = now;
                        if (now - t_first <= 500000000ull) continue;
                        // partner not resident: give up, never hang -- and say so:
                        // the host-mapped word for the next entry point / aether_check_async_error, and an LDS word that
                        // turns this workgroup's outputs into NaN (a launch that finished on stale rows must not look valid)
                        if (dbg.errword) __hip_atomic_store(dbg.errword, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        arrived[1] = 1;
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int np = nv - n;                              // partner rows = visible slots outside [off, off + n)
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dbg.ps[layer - 2] + (int64_t)vb * H, 0, nv * H * 4, 0x00020000);
            for (int idx = lane; idx < np * 16; idx += 64) {
                int slot = idx >> 4;
                if (slot >= off) slot += n;
                const int c = (idx & 15) * 4;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (slot * H + c) * 4, 0, 16);   // aux 16 = sc1
                f32x4 f;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) f[r4] = __uint_as_float(v[r4]);
                st4(psb + slot * LDW + c, f);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // rows are in LDS before the word says so
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) arrived[0] = layer;
        };
        auto wait_partner_rows = [&]() {
            while (__builtin_amdgcn_readfirstlane(arrived[0]) < layer) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        };
        auto front_act = [&](int r, f32x4 (&acc)[4], f32x4 (&h1)[4], bool last) {
            if (last && !(AETHER_R3_DEFER && ROUNDS >= 3)) issue_loads(1);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) h1[mb] = silu4(acc[mb]);
        };
        auto back = [&](int r, const f32x4 (&h1)[4], bool last) {
            const int tile = NW * r + wave;
            f32x4 acc2[4];
            if (last && !(AETHER_R3_DEFER && ROUNDS >= 3)) issue_loads(2);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc2[mb] = ld4(bias + H + 16 * mb + 4 * q);
            if constexpr (SPLITG) gemm_split<4, 2>(wB, h1, acc2, lane);
            else gemm_tile<4, 4>(wB, LDW, h1, acc2, i, q);
            if (last && !(AETHER_R3_DEFER && ROUNDS >= 3)) issue_loads(3);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) e[r][mb] = silu4(acc2[mb]);
            if constexpr (KEEP) {
                if (ke[r] >= 0 && dbg.e[layer - 1] != nullptr) {          // (layer 4 under AETHER_FLAG_BACKWARD_ONLY: not kept)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
                        st4(dbg.e[layer - 1] + (int64_t)ke[r] * H + 16 * mb + 4 * q, e[r][mb]);
                }
            }
            // Per-receiver sums of the tile on the matrix core: park the tile in 16 LDS rows (one set
            // per round, private to the wave), read it back transposed and multiply by the 0/1 segment
            // matrix, out[seg][h] = sum_edge S[seg][edge] * E[edge][h]  (k runs in edge order).
            float* wst = smem + L::WSTAGE + wave * (16 * LDST);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) st4(wst + i * LDST + 16 * mb + 4 * q, e[r][mb]);
            __builtin_amdgcn_wave_barrier();
            f32x4 red[4];
#pragma unroll
            for (int nbk = 0; nbk < 4; ++nbk) red[nbk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float sel = (selbits[r] >> s4) & 1u ? 1.0f : 0.0f;
                const float* erow = wst + (4 * s4 + q) * LDST + i;
#pragma unroll
                for (int nbk = 0; nbk < 4; ++nbk) red[nbk] = mfma16(sel, erow[16 * nbk], red[nbk]);
            }
            // lane (h = 16 nbk + i, q) register r4 holds segment 4q + r4
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const unsigned row = (destpack[r] >> (8 * r4)) & 0xFFu;
                if (row != 0xFFu) {
                    float* dst = part + row * LDW + i;
                    dst[0] = red[0][r4]; dst[16] = red[1][r4]; dst[32] = red[2][r4]; dst[48] = red[3][r4];
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (last && !(AETHER_R3_DEFER && ROUNDS >= 3)) issue_loads(4);
        };
        {
            const int nvalid = n_tiles > wave ? (n_tiles - wave + NW - 1) / NW : 0;    // wave-uniform
            if (nvalid == 0) {
#pragma unroll
                for (int part = 0; part < 5; ++part) issue_loads(part);
            }
            if (xch && wave == NW - 1) receive_partner_rows();
            bool have_rows = !xch || wave == NW - 1;
#pragma unroll
            for (int r = 0; r < ROUNDS; ++r) {
                if (r < nvalid) {
                    const bool last = r == nvalid - 1;
                    FUSED_WSTAMP(layer, r, 0);
                    if (!have_rows && 16 * (NW * r + wave + 1) > na) {      // first tile with a partner sender
                        wait_partner_rows();
                        have_rows = true;
                    }
                    FUSED_WSTAMP(layer, r, 1);
                    f32x4 acc[4], h1[4];
                    front_gemm(r, acc, last);
                    front_act(r, acc, h1, last);
                    FUSED_WSTAMP(layer, r, 2);
                    back(r, h1, last);
                    FUSED_WSTAMP(layer, r, 5);
                }
            }
            if constexpr (AETHER_R3_DEFER && ROUNDS >= 3) {
                if (nvalid > 0) { issue_loads(1); issue_loads(2); issue_loads(3); issue_loads(4); }
            }
        }
        FUSED_STAMP(4 + 8 * (layer - 1) + 2);
        // ------------------------------------------------------------ node phase (locs.py:240-241)
        lds_barrier();       // all partial rows are published; wA / wB are idle
        FUSED_STAMP(4 + 8 * (layer - 1) + 3);
        // step 1: n = x_prev + (sum of the node's partial rows, in tile order) / max(deg, 1)
        if (tid < FUSED_MAX_NODES * 16) {
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t = at0; t < at1; ++t) sum += ld4(part + (aslot + t) * LDW + ac4);
            for (int t = bt0; t < bt1; ++t) sum += ld4(part + (aslot + t) * LDW + ac4);
            const f32x4 nv = ld4(xbuf + aslot * LDW + ac4) + sum / adeg;
            st4(nbuf + aslot * LDW + ac4, nv);
            if (keep && aslot < n) st4(dbg.n[layer - 1] + (int64_t)(nb + aslot) * H + ac4, nv);
        }
        if (layer < 4) {       // next layer's edge weights -> LDS
            if constexpr (SPLITG) {
                dma_images(layer + 1);       // in flight under the node phase; waited for before its last barrier
            } else {                         // (loads were issued above)
#pragma unroll
                for (int j = 0; j < STG; ++j) {
                    const int idx = tid + THREADS * j, rr = idx >> 4, cc = (idx & 15) * 4;
                    if (idx < H * H / 4) {
                        st4(wA + rr * LDW + cc, stA[j]);
                        st4(wB + rr * LDW + cc, stB[j]);
                    }
                }
            }
            if (tid < H) bias[H + tid] = b2n;
        }
        lds_barrier();       // n complete
        FUSED_STAMP(4 + 8 * (layer - 1) + 7);
        // step 2: u = SiLU(W3 n + b3): rows 16*mb2.. of u for both node tiles
        if (act3) {
            float* ubuf = smem + L::UBUF;
            const f32x4 bv = ld4(b3 + 16 * mb2 + 4 * q);
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                if (16 * tn < n) {
                    f32x4 nv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) nv[a] = ld4(nbuf + (16 * tn + i) * LDW + 16 * a + 4 * q);
                    const f32x4 acc = gemm_split_regs<2>(w3h, w3l, nv, bv);
                    st4(ubuf + (16 * tn + i) * LDU + 16 * mb2 + 4 * q, silu4(acc));
                }
            }
        }
        lds_barrier();       // u complete
        FUSED_STAMP(4 + 8 * (layer - 1) + 4);

        // step 3: x = n + W4 u + b4: rows 16*mb3.. of node tile tn3
        if (act3 && 16 * tn3 < n) {
            const float* ubuf = smem + L::UBUF;
            f32x4 acc = ld4(b4 + 16 * mb3 + 4 * q);
#pragma unroll
            for (int hk = 0; hk < 2; ++hk) {                   // k halves of 64 (each with its own range check): fewer live registers
                f32x4 uv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) uv[a] = ld4(ubuf + (16 * tn3 + i) * LDU + 16 * (4 * hk + a) + 4 * q);
                const f16x8 wh2[2] = {w4h[2 * hk], w4h[2 * hk + 1]}, wl2[2] = {w4l[2 * hk], w4l[2 * hk + 1]};
                acc = gemm_split_regs<2>(wh2, wl2, uv, acc);
            }
            acc += ld4(nbuf + (16 * tn3 + i) * LDW + 16 * mb3 + 4 * q);
            st4(xbuf + (16 * tn3 + i) * LDW + 16 * mb3 + 4 * q, acc);
            if (keep && 16 * tn3 + i < n)
                st4(dbg.x[layer] + (int64_t)(nb + 16 * tn3 + i) * H + 16 * mb3 + 4 * q, acc);
        }
        lds_barrier();
        FUSED_STAMP(4 + 8 * (layer - 1) + 5);
        // step 4: next layer's node terms P_s = W_s x, P_r = W_r x + b1 (locs.py:233 split)
        if (layer < 4) {
            const float* b1n = P.ln_msg_b0[layer - 1];
            if (16 * tn4 < n) {
                f32x4 xv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) xv[a] = ld4(xbuf + (16 * tn4 + i) * LDW + 16 * a + 4 * q);
                if (do_s) {
                    const f32x4 accs = gemm_split_regs<2>(wsh, wsl, xv, f32x4{0.f, 0.f, 0.f, 0.f});
                    if (16 * tn4 + i < n) {      // sender rows live in the visible numbering
                        st4(psb + (off + 16 * tn4 + i) * LDW + 16 * mb3 + 4 * q, accs);
                        float* gps = dbg.ps[layer - 1] + (int64_t)(nb + 16 * tn4 + i) * H + 16 * mb3 + 4 * q;
                        if (wg.partner >= 0) {       // write-through (sc1) payload: no release fence needed
                            // ONE 16-byte sc1 store per lane (aux 16): an 8-byte sc1 store is a fabric write of its
                            // own, 2.7 x the time per byte (MI355X_MICROARCH.md, visibility price list)
                            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
                                dbg.ps[layer - 1] + (int64_t)nb * H, 0, n * H * 4, 0x00020000);
                            u32x4 bits;
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) bits[r4] = __float_as_uint(accs[r4]);
                            __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, ((16 * tn4 + i) * H + 16 * mb3 + 4 * q) * 4, 0, 16);
                        } else if (keep) {
                            st4(gps, accs);
                        }
                    }
                }
                if (do_r) {
                    const f32x4 accr = gemm_split_regs<2>(wrh, wrl, xv, ld4(b1n + 16 * mb3 + 4 * q));
                    st4(prb + (16 * tn4 + i) * LDW + 16 * mb3 + 4 * q, accr);
                    if (keep && 16 * tn4 + i < n)
                        st4(dbg.pr[layer - 1] + (int64_t)(nb + 16 * tn4 + i) * H + 16 * mb3 + 4 * q, accr);
                }
            }
            // First half of the hand-off of the own P_s rows to the partner workgroup
            // (cdna_hip_programming.md Guideline 16, form R1 with write-through payload): every byte of
            // the payload was stored sc1 (16-byte buffer stores above); every wave drains its stores
            // before the barrier, then one lane raises the flag.  The partner picks the rows up after
            // the first GEMM of its next edge phase (receive_partner_rows), as this workgroup does with
            // the partner's: the flag's flight time hides behind those 64 MFMAs.
            if (wg.partner >= 0 || SPLITG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // hand-off stores; LDS-DMA images
            lds_barrier();   // P_s / P_r and the staged weights are visible to the next edge tiles
            if (wg.partner >= 0 && tid == 0)
                __hip_atomic_store(dbg.flags + blockIdx.x, layer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FUSED_STAMP(4 + 8 * (layer - 1) + 6);
        }
    }
    if (wg.partner >= 0 && tid == 0)       // the partner's last flag value has been consumed: re-arm it
        __hip_atomic_store(dbg.flags + wg.partner, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---------------------------------------------------------------- out MLP + globalise + residual
    // locs.py:160-168,193; local_to_global.py:12-13; aether.py:185
    {
        float* o1 = smem + L::OBUF1;
        float* o2 = smem + L::OBUF2;
        const int mb = wave & 3, tn = (wave >> 2) & 1;
        const bool act = wave < 8 && 16 * tn < n;
        f32x4 w6v[4];                                    // last Linear: rows >= D of the block are discarded
        if (wave < 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a) w6v[a] = ld4(P.out_w6 + (i < D ? i : D - 1) * H + 16 * a + 4 * q);
        }
        if (act) {
            f32x4 xv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xv[a] = ld4(xbuf + (16 * tn + i) * LDW + 16 * a + 4 * q);
            const f32x4 acc = gemm_split_regs<2>(wsh, wsl, xv, ld4(P.out_b0 + 16 * mb + 4 * q));
            f32x4 v = silu4(acc);
            if constexpr (KEEP) {
                if (dbg.step.drop1 != nullptr && 16 * tn + i < n)
                    v = v * ld4(dbg.step.drop1 + (int64_t)(nb + 16 * tn + i) * H + 16 * mb + 4 * q);
            }
            st4(o1 + (16 * tn + i) * LDW + 16 * mb + 4 * q, v);
        }
        if constexpr (KEEP) {
            if (blockIdx.x == 0 && tid == 0 && dbg.step.dropword != nullptr) *dbg.step.dropword = dbg.step.drop1 != nullptr ? 1 : 0;
        }
        lds_barrier();
        if (act) {
            f32x4 xv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) xv[a] = ld4(o1 + (16 * tn + i) * LDW + 16 * a + 4 * q);
            const f32x4 acc = gemm_split_regs<2>(wrh, wrl, xv, ld4(P.out_b3 + 16 * mb + 4 * q));
            f32x4 v = silu4(acc);
            if constexpr (KEEP) {
                if (dbg.step.drop2 != nullptr && 16 * tn + i < n)
                    v = v * ld4(dbg.step.drop2 + (int64_t)(nb + 16 * tn + i) * H + 16 * mb + 4 * q);
            }
            st4(o2 + (16 * tn + i) * LDW + 16 * mb + 4 * q, v);
        }
        lds_barrier();
        if (wave < 2 && 16 * wave < n) {
            const int tn2 = wave;
            f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 xv = ld4(o2 + (16 * tn2 + i) * LDW + 16 * a + 4 * q);
#pragma unroll
                for (int b = 0; b < 4; ++b) y = mfma16(w6v[a][b], xv[b], y);
            }
            const int node = 16 * tn2 + i;
            if (q == 0 && node < n) {
                float yl[D];
#pragma unroll
                for (int d = 0; d < D; ++d) yl[d] = y[d] + P.out_b6[d];
                const float* ni = ninfo + (off + node) * 24;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    float s = 0.f;
#pragma unroll
                    for (int b = 0; b < D; ++b) s += ni[NI::R + a * D + b] * yl[b];
                    float xn = ni[NI::P + a] + s;
                    if (arrived[1]) xn = __builtin_nanf("");          // a hand-off wait gave up somewhere in this workgroup
                    out[(int64_t)(nb + node) * D + a] = xn;
                    if (dbg.step.vel_out)
                        dbg.step.vel_out[(int64_t)(nb + node) * D + a] = (xn - ni[NI::P + a]) / dbg.step.dt;
                }
            }
        }
    }
    FUSED_STAMP(40);
#ifdef AETHER_FUSED_STAMPS
    if (tid == 0 && blockIdx.x < 4096)      // total cycles, for the clock estimate
        dbg.stamps[blockIdx.x * FUSED_STAMPS + 41] = (float)(__builtin_amdgcn_s_memtime() - c_entry);
#endif
}

}  // namespace
