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

#include "common.h"
#include "streamed.h"
#include "fused.h"
#include "backward.h"
#include "fused_bwd.h"
#include "edge_acc.h"
#include "input_grad.h"
#include "wide.h"
#include "seq2seq.h"
#include "s2s_filter.h"
#include "s2s_step.h"
#include "dynfield.h"
#include "s2s_dynfield.h"
#include "knn.h"
#include "dyn_step.h"
#include "sim.h"
#include "train_step.h"

#include <mutex>
#include <utility>
#include <vector>

namespace {

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
enum KernelId { K_NODE_PREP = 0, K_EDGE_L1, K_NODE_UPDATE, K_EDGE_LN, K_NODE_LAST, K_FUSED, KB_OUT, KB_NODE,
                KB_EDGE, KB_GATHER, KB_FIELD, KB_OUTER, K_SEGMEAN, KB_FUSED, KB_FBRED, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"k_node_prep", "k_edge_layer1", "k_node_update", "k_edge_layer",
                                           "k_node_update_last", "k_fused", "kb_out", "kb_node", "kb_edge",
                                           "kb_gather", "kb_field", "k_outer+reduce", "k_segment_mean", "k_fused_bwd",
                                           "k_fb_reduce"};
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

// Kernels that use more than 64 KiB of dynamic LDS need an explicit opt-in, once per (kernel, device).
int ensure_dynamic_lds(const void* kernel, size_t bytes) {
    struct Entry { const void* kernel; int dev; size_t bytes; };
    static std::mutex mu;
    static std::vector<Entry> done;
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (auto& d : done)
        if (d.kernel == kernel && d.dev == dev) {
            if (d.bytes >= bytes) return AETHER_OK;
            HIP_OK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            d.bytes = bytes;
            return AETHER_OK;
        }
    HIP_OK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.push_back({kernel, dev, bytes});
    return AETHER_OK;
}

// A kernel that gives up a bounded wait (the split-mode hand-off, fused.h) cannot return a status: it sets a
// word in host-mapped memory instead.  Every entry point that launches looks at it first, so a timeout turns
// into an AETHER_EHIP return of the NEXT call (aether_check_async_error asks explicitly after a sync).
int* g_async_host = nullptr;
int* g_async_dev = nullptr;
int* async_error_word() {
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return;
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipHostFree(h); return; }
        g_async_host = (int*)h;
        g_async_dev = (int*)d;
    });
    return g_async_dev;
}
int take_async_error() {
    if (g_async_host && *(volatile int*)g_async_host) {
        const int code = *(volatile int*)g_async_host;
        *(volatile int*)g_async_host = 0;
        if (code == 2)
            return fail(AETHER_EHIP, "an earlier aether_dyn_step found a different number of non-zero mask entries than "
                                     "n_present: its outputs were filled with NaN");
        return fail(AETHER_EHIP, "an earlier launch gave up waiting for its partner workgroup (split-mode hand-off): "
                                 "the results of that launch are invalid");
    }
    return AETHER_OK;
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

__global__ void k_iota(int32_t* __restrict__ v, int64_t n) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) v[k] = (int32_t)k;
}

// rowptr of a sorted key array (covers empty rows)
__global__ void k_rowptr(const int32_t* __restrict__ keys_sorted, int64_t n, int64_t n_rows,
                         int32_t* __restrict__ rowptr) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int32_t r = keys_sorted[k];
    int32_t rp = k == 0 ? -1 : keys_sorted[k - 1];
    for (int32_t j = rp + 1; j <= r; ++j) rowptr[j] = (int32_t)k;
    if (k == n - 1)
        for (int64_t j = r + 1; j <= n_rows; ++j) rowptr[j] = (int32_t)n;
}

// Component detection: an edge (s, r) "crosses" every boundary n with min < n <= max.
// diff is a difference array; its inclusive prefix sum is the number of crossing edges.
__global__ void k_graph_cross(const int64_t* __restrict__ send, const int64_t* __restrict__ recv,
                              int64_t n_edges, int64_t n_nodes, int32_t* __restrict__ diff) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    int64_t s = send[k], r = recv[k];
    if (s < 0 || s >= n_nodes || r < 0 || r >= n_nodes || s == r) return;
    int64_t a = s < r ? s : r, b = s < r ? r : s;
    atomicAdd(diff + a + 1, 1);
    atomicAdd(diff + b + 1, -1);
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

// Does an edge index equal the one a graph view was built from?  (send[perm[k]], recv[perm[k]]) against the sorted copies.
__global__ void k_graph_match(const int64_t* __restrict__ send, const int64_t* __restrict__ recv, int64_t n_edges,
                              const int32_t* __restrict__ perm, const int32_t* __restrict__ send_s,
                              const int32_t* __restrict__ recv_s, int* __restrict__ differs) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    const int64_t e = perm[k];
    if (send[e] != (int64_t)send_s[k] || recv[e] != (int64_t)recv_s[k]) *differs = 1;
}

// ------------------------------------------------------------------ host-side layouts
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int64_t FUSED_TABLE_MAX_EDGES = 4 << 20;
int g_fused_split = 1;        // aether_set_option("fused_split", 0|1): two workgroups per group when CUs are idle
int g_fused_pair_stride = 8;  // aether_set_option("fused_pair_stride", 1|8): distance of the two workgroups of a split group
                              // (8: both on one XCD -- workgroup b runs on XCD b % 8 -- 47.25 -> 46.56 us per launch at cfg2; a
                              // placement hint only: the hand-off protocol is the cross-XCD one either way)

struct GraphLayout {
    size_t perm, send_s, recv_s, rowptr, gsel, wgdesc, tdesc, tsel, tdst, lorder, ledge, nrange, hflags, sperm, srowptr, keys,
        vals, diff, cross, flag, cub, total, cub_bytes;
    int64_t max_wgs, max_tiles;
    GraphLayout(int64_t E, int64_t Nn, bool with_sort_scratch = true) {
        size_t off = 0;
        auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
        size_t e4 = (size_t)(E > 0 ? E : 1) * 4;
        perm = take(e4); send_s = take(e4); recv_s = take(e4);
        rowptr = take((size_t)(Nn + 1) * 4);
        gsel = take((size_t)((E + 15) / 16) * 64 * 4 + 4);     // tile structure for tile_receiver_sums
        // fused-path tables; skipped for graphs far beyond what the fused kernel serves
        const bool tables = E <= FUSED_TABLE_MAX_EDGES;
        max_wgs = tables ? 2 * Nn : 0;
        max_tiles = tables ? (E + 15) / 16 + 2 * Nn : 0;
        wgdesc = take((size_t)max_wgs * sizeof(FusedWG) + 4);
        tdesc = take((size_t)max_tiles * sizeof(FusedTile) + 4);
        tsel = take((size_t)max_tiles * 64 * 4 + 4);
        tdst = take((size_t)max_tiles * 64 * 4 + 4);
        lorder = take(tables ? e4 : 4);         // local edge order of every fused workgroup (FusedWG)
        ledge = take(tables ? 4 * e4 : 16);     // ... and {sorted position, sender, receiver, original edge} in that order
        nrange = take(tables ? (size_t)Nn * 16 : 4);                 // per node: local index ranges of its two runs
        hflags = take((size_t)(2 * max_wgs + 64) * 4);               // split-mode hand-off flags: forward | backward
        sperm = take(e4);                       // receiver-sorted positions grouped by sender (stable)
        srowptr = take((size_t)(Nn + 1) * 4);
        keys = take(e4); vals = take(e4);
        diff = take((size_t)(Nn + 2) * 4); cross = take((size_t)(Nn + 2) * 4);
        flag = take(256);
        cub_bytes = 0;
        if (with_sort_scratch) {    // size queries only: no launch, no device access
            (void)hipcub::DeviceRadixSort::SortPairs<int32_t, int32_t>(nullptr, cub_bytes, nullptr, nullptr,
                                                               nullptr, nullptr, (int)E, 0, 32, nullptr);
            size_t scan_bytes = 0;
            (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr,
                                                   (int)(Nn + 1), nullptr);
            if (scan_bytes > cub_bytes) cub_bytes = scan_bytes;
        }
        cub = take(cub_bytes + 256);
        total = off;
    }
};

constexpr int OUTER_MAX_CHUNKS = 256;      // workgroups (= partials) per task
// Up to this many edges the backward keeps the operands of every layer's weight gradients alive and
// multiplies all of them in one launch at the end (12 more [E, 64] buffers: 3 KiB per edge).
int64_t g_outer_defer_max_edges = 1 << 20;      // aether_set_option("outer_defer_max_edges", n)
int g_edge_acc = 3;                             // aether_set_option("edge_acc", 0..3; 3 = kb_edge_acc8<4 waves>, two workgroups per CU): above that threshold, the edge-level weight
                                                // gradients are accumulated inside the edge kernel (edge_acc.h)
int g_outer_tiles_per_wave = 0;                 // aether_set_option("outer_tiles_per_wave", n): 16-row tiles per wave of k_outer (0: by task size)
int g_linear_small_wgs = 128;                   // aether_set_option("linear_small_wgs", n): below n workgroups, 16 x 32 blocks
int g_linear_kwaves = 4;                        // aether_set_option("linear_kwaves", 1 | 4): waves of a workgroup that split a small layer's k-groups
int g_filter_wg_target = 768;                   // aether_set_option("filter_wg_target", n): k-splits of the first-version filter kernel (variable-N steps)
int g_gemm_split = 1;                           // aether_set_option("gemm_split", 0 | 1 | 2 | 3): fp16 x 2 GEMM for the >= 128-workgroup layers of the fused seq2seq step (2 / 3: force a kernel structure)
int g_dyn_filter_v1 = 0;                        // aether_set_option("dyn_filter_v1", 0 auto | 1 always | 2 never): first-version filter kernel in the variable-N steps
int g_dyn_filter_v1_edges = 0;                  // auto: below this many edges (measured: no size where the first version wins)
int g_filter_rsplits = 0;                       // aether_set_option("filter_rsplits", 0 auto | 1 | 3 | 5 | 15): feature split of the 15-feature filter GEMM
int g_filter_wgs = 256;                         // workgroups of k_s2s_filter_split: one per CU
int g_filter_splits = 0;                        // aether_set_option("filter_splits", n): k-splits of the filter GEMM, 0 = by balance


struct WsLayout {
    // forward (always)
    size_t nodeinfo, x[5], ps[3], pr[3], e[4], aggr, part, stamps, flags, velbuf[2], wimg, fwd_total;
    // saved by the forward under KEEP_INTERMEDIATES for the backward
    size_t n[4], feat, drop, dropword;      // drop: [2][n_nodes][64] dropout scale masks (caller-written), dropword: applied?
    // backward temporaries
    // Operands of the weight-gradient outer products are per layer when `defer` (all of them are then
    // multiplied in ONE launch at the end of the backward); otherwise the layers share one set.
    size_t DXl[5], Ul[4], DPUl[4], Gl[4], H1l[4], DP2l[4], DPSl[4], DPRl[4];
    size_t wt, DN, O1, O2, DPO1, DPO2, DY, DE, DA, RELF, Z, H1f, H2f, DPH1, DPH2, DF, DZE, ONEHOT, partial;
    size_t fpart, xchg;             // fused backward: per-workgroup weight-gradient partials, split-mode dP_s exchange
    size_t fpart_wgs, xchg_wgs;     // workgroups they are sized for
    bool defer;
    size_t partial_cap;             // partials (OUTER_PART floats each)
    size_t total;
    WsLayout(int64_t Nn, int64_t E, int D, bool training) {
        size_t off = 0;
        auto take = [&](size_t floats) { size_t o = off; off = align_up(off + floats * 4, 256); return o; };
        const size_t nn = (size_t)Nn, ee = (size_t)E;
        nodeinfo = take(nn * (D == 2 ? 16 : 24));
        for (auto& v : x) v = take(nn * H);
        for (auto& v : ps) v = take(nn * H);
        for (auto& v : pr) v = take(nn * H);
        for (auto& v : e) v = take(ee * H);
        aggr = take(nn * H);
        part = take((nn + (ee + 15) / 16 + 1) * H);   // per-(receiver, tile) sums of the streamed edge kernels
        stamps = take((size_t)4096 * FUSED_STAMPS);
        flags = take(2 * nn + 64);              // split-mode hand-off flags, one int per workgroup
        for (auto& v : velbuf) v = take(nn * 4);    // aether_rollout: velocities of the steps, ping-pong
        wimg = take(FUSED_WIMG_SET);                // split (3 x bf16) images of the edge-MLP weights (k_split_weights)
        fwd_total = off;
        for (auto& v : n) v = take(nn * H);
        feat = take(ee * FPAD);
        drop = take(nn * 2 * H); dropword = take(64);
        defer = E <= g_outer_defer_max_edges;
        const int sets = defer ? 4 : 1;
        for (int k = 0; k < 5; ++k) DXl[k] = (defer || k < 2) ? take(nn * H) : DXl[k - 2];
        for (int k = 0; k < 4; ++k) {
            const bool own = k < sets;
            Ul[k] = own ? take(nn * 2 * H) : Ul[0];
            DPUl[k] = own ? take(nn * 2 * H) : DPUl[0];
            DPSl[k] = own ? take(nn * H) : DPSl[0];
            DPRl[k] = own ? take(nn * H) : DPRl[0];
            Gl[k] = own ? take(ee * H) : Gl[0];
            // (not deferred + edge_acc: h and dpre2 never leave the edge kernel -- 2 x 8.6 GB less at config 5's shard)
            const bool rows_needed = defer || !g_edge_acc;
            H1l[k] = own ? take(rows_needed ? ee * H : 64) : H1l[0];
            DP2l[k] = own ? take(rows_needed ? ee * H : 64) : DP2l[0];
        }
        wt = take((size_t)160 * 1024); DN = take(nn * H);
        O1 = take(nn * H); O2 = take(nn * H); DPO1 = take(nn * H); DPO2 = take(nn * H); DY = take(nn * 16);
        DE = take(ee * H); DA = take(ee * FPAD);
        RELF = take(nn * 16);
        Z = take(nn * 32); H1f = take(nn * 32); H2f = take(nn * 32); DPH1 = take(nn * 32); DPH2 = take(nn * 32);
        DF = take(nn * 16); DZE = take(nn * 16); ONEHOT = take(nn * 16);
        {   // at most 9 edge-level and 48 node-level tasks (64 x 64 pieces) share the partial buffer
            auto chunks_of = [](size_t rows) {
                size_t c = ((rows + 15) / 16 + 3) / 4;
                return c < 1 ? (size_t)1 : (c > (size_t)OUTER_MAX_CHUNKS ? (size_t)OUTER_MAX_CHUNKS : c);
            };
            partial_cap = 9 * chunks_of(ee) + 48 * chunks_of(nn);     // 64 x 64 pieces
            partial = take(partial_cap * OUTER_PART);
        }
        // fused backward (fused_bwd.h): one partial per workgroup and layer; a workgroup owns >= 1 node, split
        // graphs have at most one workgroup per CU (<= 512 covers every CU count this library will see)
        fpart_wgs = 2 * nn < (size_t)FB_MAX_WGS ? 2 * nn : (size_t)FB_MAX_WGS;
        fpart = take(fpart_wgs * 4 * FB_PART);
        xchg_wgs = 2 * nn < 512 ? 2 * nn : 512;
        xchg = take(xchg_wgs * 3 * 4 * FUSED_MAX_NODES * H);
        total = training ? off : fwd_total;
    }
};

inline int bits_for(int64_t n) { int b = 1; while (((int64_t)1 << b) < n) ++b; return b; }


// Layout of the transposed weight copies the backward kernels read (one region of the workspace), as tasks.
template <int D>
BwdWT transposed_weights(const AetherParams& P, float* base, TransposeBatch& TB) {
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    BwdWT WT;
    size_t off = 0;
    TB.n_tasks = 0;
    auto add = [&](const float* src, int rows, int cols, int src_ld, int col0, int ldd, int cols_pad) {
        float* dst = base + off;
        TB.t[TB.n_tasks++] = TransposeTask{src, dst, rows, cols, src_ld, col0, ldd, cols_pad};
        off += (size_t)cols_pad * ldd;
        return (const float*)dst;
    };
    WT.out_w0t = add(P.out_w0, H, H, H, 0, H, H);
    WT.out_w3t = add(P.out_w3, H, H, H, 0, H, H);
    WT.out_w6t = add(P.out_w6, D, H, H, 0, 16, H);
    for (int l = 1; l <= 4; ++l) {
        const float* w3 = l == 1 ? P.l1_upd_w0 : P.ln_upd_w0[l - 2];
        const float* w4 = l == 1 ? P.l1_upd_w2 : P.ln_upd_w2[l - 2];
        const float* w2 = l == 1 ? P.l1_msg_w2 : P.ln_msg_w2[l - 2];
        WT.upd_w2t[l - 1] = add(w4, H, 2 * H, 2 * H, 0, H, 2 * H);          // [128][64]
        WT.upd_w0t[l - 1] = add(w3, 2 * H, H, H, 0, 2 * H, H);              // [64][128]
        WT.msg_w2t[l - 1] = add(w2, H, H, H, 0, H, H);
        if (l == 1) WT.msg_w0t[0] = add(P.l1_msg_w0, H, F1, F1, 0, H, FPAD); // [32][64]
        else WT.msg_w0t[l - 1] = add(P.ln_msg_w0[l - 2], H, 3 * H, 3 * H, 0, H, 3 * H);   // [192][64]
    }
    return WT;
}

// Everything derived from the weights alone, in ONE launch in front of the step: the split (3 x bf16) images of the
// edge-MLP matrices the fused forward copies into LDS (blocks 0-7) and, when the intermediates are kept for a
// backward, the transposed copies its kernels read (the remaining blocks).
__global__ void __launch_bounds__(512)
k_prepare_weights(AetherParams P, int f1, float* __restrict__ wimg, TransposeBatch TB, int split_blocks) {
    if ((int)blockIdx.x < split_blocks) {
        split_weights_block(P, f1, wimg, (int)blockIdx.x, threadIdx.x);
        return;
    }
    const int b = (int)blockIdx.x - split_blocks;
    const TransposeTask T = TB.t[b >> 1];                      // two blocks per task
    const int total = T.cols_pad * T.ldd;
    for (int idx = (b & 1) * 512 + threadIdx.x; idx < total; idx += 1024) {
        const int c = idx / T.ldd, r = idx - c * T.ldd;
        T.dst[idx] = (c < T.cols && r < T.rows) ? T.src[(size_t)r * T.src_ld + T.col0 + c] : 0.0f;
    }
}

template <int D>
int prepare_weights(const AetherParams& P, char* ws, bool split_images, bool transposes, int64_t Nn, int64_t E, hipStream_t st) {
    if (!split_images && !transposes) return AETHER_OK;
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    WsLayout W(Nn, E, D, transposes);
    TransposeBatch TB;
    TB.n_tasks = 0;
    if (transposes) (void)transposed_weights<D>(P, reinterpret_cast<float*>(ws + W.wt), TB);
    const int sb = split_images ? FUSED_SPLIT_BLOCKS : 0;
    k_prepare_weights<<<dim3((unsigned)(sb + 2 * TB.n_tasks)), dim3(512), 0, st>>>(P, F1, reinterpret_cast<float*>(ws + W.wimg),
                                                                                    TB, sb);
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

template <int D, int NW, int ROUNDS, bool KEEP>
int fused_launch(const AetherParams& P, const float* x, const float* vel, const float* charges,
                 const float* ea, const int32_t* perm, const int32_t* send_s, const int32_t* recv_s,
                 const int32_t* rowptr, const FusedWG* wgdesc, const uint32_t* tsel, const uint32_t* tdst,
                 const int4* ledge, const int32_t* nrange, int n_groups, const FusedDebug& dbg, float* out,
                 hipStream_t st) {
    auto kern = k_fused<D, NW, ROUNDS, KEEP>;
    constexpr size_t lds = (size_t)FusedLds<NW, ROUNDS>::TOTAL * 4;
    if (ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds)) return AETHER_EHIP;
    ProfScope ps(K_FUSED, st);
    kern<<<dim3((unsigned)n_groups), dim3(NW * 64), lds, st>>>(P, x, vel, charges, ea, perm, send_s, recv_s,
                                                              rowptr, wgdesc, tsel, tdst, ledge, nrange, dbg, out);
    return AETHER_OK;
}

template <int D>
int fused_impl(const AetherParams& P, int64_t Nn, int64_t E, const AetherGraphInfo& info, const float* x,
               const float* vel, const float* charges, const float* ea, const char* graph, char* ws,
               float* out, bool keep, bool ws_reused, StepExtras step, hipStream_t st, bool weights_prepared = false) {
    GraphLayout G(E, Nn, false);
    WsLayout W(Nn, E, D, keep);
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(graph + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    FusedDebug dbg;
    dbg.nodeinfo = wp(W.nodeinfo);
    for (int k = 0; k < 5; ++k) dbg.x[k] = wp(W.x[k]);
    for (int k = 0; k < 4; ++k) dbg.e[k] = wp(W.e[k]);
    if (step.skip_e4) dbg.e[3] = nullptr;       // training: the backward never reads the last layer's messages
    for (int k = 0; k < 4; ++k) dbg.n[k] = wp(W.n[k]);
    for (int k = 0; k < 3; ++k) { dbg.ps[k] = wp(W.ps[k]); dbg.pr[k] = wp(W.pr[k]); }
    dbg.feat = wp(W.feat);
    dbg.stamps = wp(W.stamps);
    // hand-off flags live with the graph view: zeroed when it is built, re-armed by every launch that used them,
    // never touched by anything else -- no memset per call, whatever else the caller does with the workspace
    dbg.flags = reinterpret_cast<int*>(const_cast<char*>(graph) + G.hflags);
    dbg.errword = async_error_word();
    dbg.wimg = wp(W.wimg);
    dbg.step = step;
    const FusedWG* wgd = reinterpret_cast<const FusedWG*>(graph + G.wgdesc);
    const uint32_t* tsel = reinterpret_cast<const uint32_t*>(graph + G.tsel);
    const uint32_t* tdst = reinterpret_cast<const uint32_t*>(graph + G.tdst);
    (void)ws_reused;
    const int tiles = (info.max_group_edges + 15) / 16;
    // split-GEMM variants: 3 x bf16 images of the eight edge-MLP matrices; training: the backward's transposed copies
    if (prepare_weights<D>(P, ws, !weights_prepared, keep, Nn, E, st)) return AETHER_EHIP;
    int rc;
#define AETHER_FUSED_CASE(NWV, R)                                                                     \
    rc = keep ? fused_launch<D, NWV, R, true>(P, x, vel, charges, ea, gp(G.perm), gp(G.send_s),       \
                                              gp(G.recv_s), gp(G.rowptr), wgd, tsel, tdst,            \
                                              reinterpret_cast<const int4*>(graph + G.ledge), gp(G.nrange), info.n_groups, dbg, out, st) \
              : fused_launch<D, NWV, R, false>(P, x, vel, charges, ea, gp(G.perm), gp(G.send_s),      \
                                               gp(G.recv_s), gp(G.rowptr), wgd, tsel, tdst,           \
                                               reinterpret_cast<const int4*>(graph + G.ledge), gp(G.nrange), info.n_groups, dbg, out, st)
    if (tiles <= 8) { AETHER_FUSED_CASE(8, 1); }
    else if (tiles <= 16) { AETHER_FUSED_CASE(8, 2); }
    else { AETHER_FUSED_CASE(8, 3); }
#undef AETHER_FUSED_CASE
    if (rc != AETHER_OK) return rc;
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

template <int D>
int streamed_impl(const AetherParams& P, int64_t Nn, int64_t E, const float* x, const float* vel,
                  const float* charges, const float* ea, const char* graph, char* ws, float* out,
                  bool keep, StepExtras step, hipStream_t st) {
    GraphLayout G(E, Nn, false);
    WsLayout W(Nn, E, D, keep);
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(graph + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int32_t *perm = gp(G.perm), *send_s = gp(G.send_s), *recv_s = gp(G.recv_s), *rowptr = gp(G.rowptr);
    float* nodeinfo = wp(W.nodeinfo);
    // the backward's transposed copies -- and the split images too: aether_backward picks the fused backward from the
    // graph alone (fused_backward_applies), whichever forward ran, and that kernel stages its recompute GEMMs from them
    if (prepare_weights<D>(P, ws, keep, keep, Nn, E, st)) return AETHER_EHIP;

    {
        ProfScope ps(K_NODE_PREP, st);
        k_node_prep<D><<<dim3((unsigned)((Nn + 15) / 16)), dim3(64), 0, st>>>(P, x, vel, charges,
                                                                              nodeinfo, wp(W.x[0]), step.ext_field, Nn);
    }
    const int64_t n_chunks = (E + 255) / 256;
    const int64_t n_tiles = (E + 15) / 16;
    const unsigned node_grid = (unsigned)((Nn + 15) / 16);
    const uint32_t* gsel = reinterpret_cast<const uint32_t*>(graph + G.gsel);
    if (E > 0) {
        const size_t lds1 = (size_t)(SPLIT_WIMG / 2 + SPLIT_WIMG + 2 * H + 4 * 64 * LDF) * 4;
        if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_edge_layer1<D>), lds1)) return AETHER_EHIP;
        const int64_t n_wg1 = ((E + 63) / 64 + 3) / 4;
        unsigned g1 = (unsigned)(n_wg1 < 512 ? n_wg1 : 512);              // 2 workgroups per CU
        ProfScope ps(K_EDGE_L1, st);
        k_edge_layer1<D><<<dim3(g1), dim3(256), lds1, st>>>(P, nodeinfo, ea, perm, send_s, recv_s, gsel,
                                                           wp(W.part), wp(W.e[0]), keep ? wp(W.feat) : nullptr,
                                                           step.qattr, E);
    }
    auto seg_mean = [&](int l) {
        ProfScope ps(K_SEGMEAN, st);
        k_segment_mean<<<dim3((unsigned)((Nn + 3) / 4)), dim3(256), 0, st>>>(wp(W.part), rowptr, wp(W.aggr), Nn);
    };
    seg_mean(1);
    {
        ProfScope ps(K_NODE_UPDATE, st);
        k_node_update<D, false><<<dim3(node_grid), dim3(256), 0, st>>>(
        P, 1, wp(W.x[0]), wp(W.aggr), wp(W.x[1]), wp(W.ps[0]), wp(W.pr[0]), nodeinfo, x, out,
        keep ? wp(W.n[0]) : nullptr, nullptr, 1.0f, Nn);
    }
    for (int l = 2; l <= 4; ++l) {
        if (E > 0) {
            const size_t lds = (size_t)(2 * SPLIT_WIMG + H + EDGE_LN_WAVES * 16 * LDST) * 4;
            if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_edge_layer), lds)) return AETHER_EHIP;
            int64_t wgs = (n_tiles + EDGE_LN_WAVES - 1) / EDGE_LN_WAVES;
            unsigned g = (unsigned)(wgs < 256 ? wgs : 256);                   // one 12-wave workgroup per CU
            ProfScope ps(K_EDGE_LN, st);
            // layer 4's messages are only needed as receiver sums (locs.py:190-193) unless kept
            float* eo = (l == 4 && (!keep || step.skip_e4)) ? nullptr : wp(W.e[l - 1]);
            k_edge_layer<<<dim3(g), dim3(64 * EDGE_LN_WAVES), lds, st>>>(P.ln_msg_w0[l - 2], P.ln_msg_w2[l - 2], P.ln_msg_b2[l - 2],
                                                         wp(W.ps[l - 2]), wp(W.pr[l - 2]), wp(W.e[l - 2]), send_s,
                                                         recv_s, gsel, wp(W.part), eo, E);
        }
        seg_mean(l);
        if (l < 4) {
            ProfScope ps(K_NODE_UPDATE, st);
            k_node_update<D, false><<<dim3(node_grid), dim3(256), 0, st>>>(
                P, l, wp(W.x[l - 1]), wp(W.aggr), wp(W.x[l]), wp(W.ps[l - 1]), wp(W.pr[l - 1]),
                nodeinfo, x, out, keep ? wp(W.n[l - 1]) : nullptr, nullptr, 1.0f, Nn);
        } else {
            ProfScope ps(K_NODE_LAST, st);
            k_node_update<D, true><<<dim3(node_grid), dim3(256), 0, st>>>(
                P, l, wp(W.x[l - 1]), wp(W.aggr), wp(W.x[l]), nullptr, nullptr, nodeinfo, x,
                out, keep ? wp(W.n[l - 1]) : nullptr, step.vel_out, step.dt, Nn, step.drop1, step.drop2, step.dropword);
        }
    }
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

// ------------------------------------------------------------------ backward orchestration
struct OuterList {
    std::vector<OuterTask> tasks;
    // Products wider than 64 x 64 are entered as 64 x 64 pieces (column slices of A / B, the matching
    // block of C): every task is then of the <4,4> class and one k_outer + one k_outer_reduce launch
    // serves a whole backward (the 128-wide classes cost two more launch pairs, ~36 us per step at cfg2).
    void add(const float* A, int lda, int M, const float* B, int ldb, int N, int64_t rows, float* C, int ldc,
             float* bias) {
        for (int m0 = 0; m0 < M; m0 += 64)
            for (int n0 = 0; n0 < N; n0 += 64) {
                OuterTask t;
                t.A = A + m0; t.B = B + n0; t.C = C + (size_t)m0 * ldc + n0;
                t.bias = (bias != nullptr && n0 == 0) ? bias + m0 : nullptr;
                t.lda = lda; t.ldb = ldb; t.ldc = ldc;
                t.M = M - m0 < 64 ? M - m0 : 64;
                t.N = N - n0 < 64 ? N - n0 : 64;
                t.chunks = 1; t.part0 = 0;
                t.rows = rows;
                tasks.push_back(t);
            }
    }
};

// Multiplies every task of the list (all of them 64 x 64 or smaller: OuterList::add): one k_outer and
// one k_outer_reduce launch per OUTER_MAX_TASKS tasks; then empties the list.
int run_outer(OuterList& L, float* partial, size_t partial_cap, hipStream_t st, const FbReduceArgs* fb = nullptr) {
    if (L.tasks.empty()) {
        if (fb != nullptr) k_fb_reduce<<<dim3(FB_REDUCE_GX, 4), dim3(1024), 0, st>>>(*fb);
        return AETHER_OK;
    }
    size_t next_part = 0;
    for (OuterTask& t : L.tasks) {
        int64_t tiles = (t.rows + 15) / 16;
        // row tiles per wave: 1 for the edge-level products (measured), 4 for node-level ones, whose
        // 35 KB partial per workgroup otherwise outweighs their operands (29 -> 22 us per step at cfg2)
        const int tpw = g_outer_tiles_per_wave > 0 ? g_outer_tiles_per_wave : (t.rows > 16384 ? 1 : 4);
        int64_t chunks = (tiles + 4 * tpw - 1) / (4 * tpw);
        if (chunks < 1) chunks = 1;
        if (chunks > OUTER_MAX_CHUNKS) chunks = OUTER_MAX_CHUNKS;
        t.chunks = (int)chunks;
        t.part0 = (int)next_part;
        next_part += (size_t)chunks;
        if (t.M > 64 || t.N > 64) return fail(AETHER_EINVAL, "k_outer: task wider than 64 x 64");
    }
    if (next_part > partial_cap) return fail(AETHER_EINVAL, "k_outer: partial buffer too small");
    ProfScope ps(KB_OUTER, st);
    for (size_t k0 = 0; k0 < L.tasks.size(); k0 += OUTER_MAX_TASKS) {
        OuterBatch b;
        b.n_tasks = (int)(L.tasks.size() - k0 < (size_t)OUTER_MAX_TASKS ? L.tasks.size() - k0 : (size_t)OUTER_MAX_TASKS);
        int max_chunks = 1;
        for (int k = 0; k < b.n_tasks; ++k) {
            b.t[k] = L.tasks[k0 + k];
            if (b.t[k].chunks > max_chunks) max_chunks = b.t[k].chunks;
        }
        const size_t lds = (size_t)4 * 16 * (64 + 16 + 64 + 16) * sizeof(float);
        k_outer<4, 4><<<dim3((unsigned)max_chunks, (unsigned)b.n_tasks), dim3(256), lds, st>>>(b, partial);
        const bool last = k0 + OUTER_MAX_TASKS >= L.tasks.size();
        if (fb != nullptr && last) {         // the fused backward's edge-level partials ride along (one launch less)
            const int outer_blocks = OUTER_REDUCE_GX * b.n_tasks;
            k_reduce_both<<<dim3((unsigned)(outer_blocks + FB_REDUCE_GX * 4)), dim3(1024), 0, st>>>(b, partial, *fb, outer_blocks);
        } else {
            k_outer_reduce<<<dim3((unsigned)OUTER_REDUCE_GX, (unsigned)b.n_tasks), dim3(1024), 0, st>>>(b, partial);
        }
    }
    L.tasks.clear();
    return AETHER_OK;
}

template <int D>
int backward_impl(const AetherParams& P, const AetherParams& Gr, int64_t Nn, int64_t E, const float* x,
                  const float* vel, const float* charges, const char* graph, char* ws, const float* g_out,
                  hipStream_t st, float* grad_field = nullptr) {
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    constexpr int FIN = 2 * D + 16;
    GraphLayout G(E, Nn, false);
    WsLayout W(Nn, E, D, true);
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(graph + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int32_t *send_s = gp(G.send_s), *recv_s = gp(G.recv_s), *rowptr = gp(G.rowptr);
    const int32_t *sperm = gp(G.sperm), *srowptr = gp(G.srowptr);
    float* partial = wp(W.partial);
    OuterList L;                     // pending weight-gradient products
    auto flush = [&]() -> int { return W.defer ? AETHER_OK : run_outer(L, partial, W.partial_cap, st); };
    const int64_t ntile = (Nn + 15) / 16, etile = (E + 15) / 16;
    const unsigned ngrid = (unsigned)ntile;
    const unsigned egrid = (unsigned)((etile + 3) / 4 < 512 ? (etile + 3) / 4 : 512);    // 2 workgroups per CU
    auto optin = [&](const void* k, size_t lds) -> int { return ensure_dynamic_lds(k, lds); };
    // edge-level weight gradients inside the edge kernel (edge_acc.h): one 4-wave workgroup per CU, one partial each
    const unsigned agrid = (unsigned)((etile + 3) / 4 < 256 ? (etile + 3) / 4 : 256);
    const bool acc_path = g_edge_acc && !W.defer && E > 0 && (size_t)agrid * EA_PART <= W.partial_cap * (size_t)OUTER_PART;
    // ---- transposed weight copies (one launch)
    TransposeBatch TB;
    const BwdWT WT = transposed_weights<D>(P, wp(W.wt), TB);     // written by the forward (prepare_weights)
    // ---- out MLP
    {
        { ProfScope ps(KB_OUT, st);
        kb_out<D><<<dim3(ngrid), dim3(256), 0, st>>>(P, WT, wp(W.x[4]), wp(W.nodeinfo), g_out, wp(W.DXl[4]), wp(W.O1),
                                                   wp(W.O2), wp(W.DPO1), wp(W.DPO2), wp(W.DY), Nn, wp(W.drop),
                                                   wp(W.drop) + (size_t)Nn * H, reinterpret_cast<const int*>(ws + W.dropword)); }
        L.add(wp(W.DPO1), H, H, wp(W.x[4]), H, H, Nn, Gr.out_w0, H, Gr.out_b0);
        L.add(wp(W.DPO2), H, H, wp(W.O1), H, H, Nn, Gr.out_w3, H, Gr.out_b3);
        L.add(wp(W.DY), 16, D, wp(W.O2), H, H, Nn, Gr.out_w6, H, Gr.out_b6);
        if (flush()) return AETHER_EHIP;
    }
    for (int l = 4; l >= 1; --l) {
        float* dx_cur = wp(W.DXl[l]);        // dL/dx_l; kb_gather writes dL/dx_{l-1}
        float* dx_nxt = wp(W.DXl[l - 1]);
        float *bU = wp(W.Ul[l - 1]), *bDPU = wp(W.DPUl[l - 1]), *bG = wp(W.Gl[l - 1]), *bH1 = wp(W.H1l[l - 1]);
        float *bDP2 = wp(W.DP2l[l - 1]), *bDPS = wp(W.DPSl[l - 1]), *bDPR = wp(W.DPRl[l - 1]);
        const float* w3 = l == 1 ? P.l1_upd_w0 : P.ln_upd_w0[l - 2];
        const float* b3 = l == 1 ? P.l1_upd_b0 : P.ln_upd_b0[l - 2];
        float* gw3 = l == 1 ? Gr.l1_upd_w0 : Gr.ln_upd_w0[l - 2];
        float* gb3 = l == 1 ? Gr.l1_upd_b0 : Gr.ln_upd_b0[l - 2];
        float* gw4 = l == 1 ? Gr.l1_upd_w2 : Gr.ln_upd_w2[l - 2];
        float* gb4 = l == 1 ? Gr.l1_upd_b2 : Gr.ln_upd_b2[l - 2];
        const float* w2 = l == 1 ? P.l1_msg_w2 : P.ln_msg_w2[l - 2];
        const float* b2 = l == 1 ? P.l1_msg_b2 : P.ln_msg_b2[l - 2];
        float* gw2 = l == 1 ? Gr.l1_msg_w2 : Gr.ln_msg_w2[l - 2];
        float* gb2 = l == 1 ? Gr.l1_msg_b2 : Gr.ln_msg_b2[l - 2];
        // ---- node update: dx_l -> dn_l
        { ProfScope ps(KB_NODE, st);
        kb_node<<<dim3(ngrid), dim3(256), 0, st>>>(w3, b3, WT.upd_w2t[l - 1], WT.upd_w0t[l - 1], wp(W.n[l - 1]),
                                                 dx_cur, wp(W.DN), bU, bDPU, Nn); }
        L.add(dx_cur, H, H, bU, 2 * H, 2 * H, Nn, gw4, 2 * H, gb4);
        L.add(bDPU, 2 * H, 2 * H, wp(W.n[l - 1]), H, H, Nn, gw3, H, gb3);
        // ---- edge MLP
        if (E > 0) {
            const size_t lds = (size_t)(4 * H * LDW) * 4;
            ProfScope* pse = new ProfScope(KB_EDGE, st);
            if (acc_path && g_edge_acc >= 2 && (size_t)512 * FB_PART <= W.partial_cap * (size_t)OUTER_PART) {       // (<= 512 partials)
                // round 4: two waves per SIMD, transposed products from the forward's images, accumulators partitioned by
                // output rows (edge_acc.h, kb_edge_acc8): 2 = one workgroup of eight waves per CU, 3 = two of four
                EdgeAccOut O{};
                O.w2 = gw2; O.b2 = gb2;
                int n_parts = 0;
                auto launch8 = [&](auto nw_tag) -> bool {
                    constexpr int NW = decltype(nw_tag)::value;
                    const size_t lds_8 = ea8_lds_bytes(NW);
                    const int64_t want = (etile + NW - 1) / NW, cap = 256 * (8 / NW);
                    const unsigned g8 = (unsigned)(want < cap ? want : cap);
                    n_parts = (int)g8 * (NW / 4);
                    if (l == 1) {
                        if (optin(reinterpret_cast<const void*>(kb_edge_acc8<true, NW>), lds_8)) return false;
                        kb_edge_acc8<true, NW><<<dim3(g8), dim3(64 * NW), lds_8, st>>>(
                            P.l1_msg_b0, b2, nullptr, nullptr, nullptr, wp(W.feat), send_s, recv_s, rowptr, wp(W.DN), wp(W.DE), 1, bG,
                            wp(W.DA), partial, wp(W.wimg) + fused_wimg_offset(1, 0), wp(W.wimg) + fused_wimg_offset(1, 1), E, (int)g8);
                    } else {
                        if (optin(reinterpret_cast<const void*>(kb_edge_acc8<false, NW>), lds_8)) return false;
                        kb_edge_acc8<false, NW><<<dim3(g8), dim3(64 * NW), lds_8, st>>>(
                            nullptr, b2, wp(W.ps[l - 2]), wp(W.pr[l - 2]), wp(W.e[l - 2]), nullptr, send_s, recv_s, rowptr, wp(W.DN),
                            wp(W.DE), l < 4 ? 1 : 0, bG, nullptr, partial, wp(W.wimg) + fused_wimg_offset(l, 0),
                            wp(W.wimg) + fused_wimg_offset(l, 1), E, (int)g8);
                    }
                    return true;
                };
                const bool ok8 = g_edge_acc == 2 ? launch8(std::integral_constant<int, 8>{}) : launch8(std::integral_constant<int, 4>{});
                if (!ok8) return AETHER_EHIP;
                if (l == 1) { O.we = Gr.l1_msg_w0; O.ldwe = F1; O.ncols = F1; O.b1 = Gr.l1_msg_b0; O.nb_e = 2; }
                else { O.we = Gr.ln_msg_w0[l - 2] + 2 * H; O.ldwe = 3 * H; O.ncols = H; O.b1 = nullptr; O.nb_e = 4; }
                k_edge_acc8_reduce<<<dim3((FB_PART + 63) / 64), dim3(1024), 0, st>>>(partial, n_parts, O);
            } else if (acc_path) {
                // large graphs: the two edge-level products of the layer accumulate inside the edge kernel (edge_acc.h)
                const size_t lds_a = (size_t)(4 * SPLIT_WIMG + 4 * EA_STG) * 4;      // four split images + a staging pair per wave
                EdgeAccOut O{};
                O.w2 = gw2; O.b2 = gb2;
                if (l == 1) {
                    if (optin(reinterpret_cast<const void*>(kb_edge_acc<true>), lds_a)) return AETHER_EHIP;
                    kb_edge_acc<true><<<dim3(agrid), dim3(256), lds_a, st>>>(
                        P.l1_msg_w0, F1, P.l1_msg_b0, w2, b2, WT.msg_w0t[0], WT.msg_w2t[0], nullptr, nullptr, nullptr,
                        wp(W.feat), send_s, recv_s, rowptr, wp(W.DN), wp(W.DE), 1, bG, wp(W.DA), partial,
                        wp(W.wimg) + fused_wimg_offset(1, 0), wp(W.wimg) + fused_wimg_offset(1, 1), E);
                    O.we = Gr.l1_msg_w0; O.ldwe = F1; O.ncols = F1; O.b1 = Gr.l1_msg_b0; O.nb_e = 2;
                } else {
                    if (optin(reinterpret_cast<const void*>(kb_edge_acc<false>), lds_a)) return AETHER_EHIP;
                    kb_edge_acc<false><<<dim3(agrid), dim3(256), lds_a, st>>>(
                        P.ln_msg_w0[l - 2], 0, nullptr, w2, b2, WT.msg_w0t[l - 1] + 2 * H * H, WT.msg_w2t[l - 1],
                        wp(W.ps[l - 2]), wp(W.pr[l - 2]), wp(W.e[l - 2]), nullptr, send_s, recv_s, rowptr, wp(W.DN),
                        wp(W.DE), l < 4 ? 1 : 0, bG, nullptr, partial, wp(W.wimg) + fused_wimg_offset(l, 0),
                        wp(W.wimg) + fused_wimg_offset(l, 1), E);
                    O.we = Gr.ln_msg_w0[l - 2] + 2 * H; O.ldwe = 3 * H; O.ncols = H; O.b1 = nullptr; O.nb_e = 4;
                }
                k_edge_acc_reduce<<<dim3(EA_PART / 64), dim3(1024), 0, st>>>(partial, (int)agrid, O);
            } else if (l == 1) {
                if (optin(reinterpret_cast<const void*>(kb_edge<true>), lds)) return AETHER_EHIP;
                kb_edge<true><<<dim3(egrid), dim3(256), lds, st>>>(
                    P.l1_msg_w0, F1, P.l1_msg_b0, w2, b2, WT.msg_w0t[0], WT.msg_w2t[0], nullptr, nullptr, nullptr,
                    wp(W.feat), send_s, recv_s, rowptr, wp(W.DN), wp(W.DE), 1, bG, bH1, bDP2,
                    wp(W.DA), E);
            } else {
                if (optin(reinterpret_cast<const void*>(kb_edge<false>), lds)) return AETHER_EHIP;
                kb_edge<false><<<dim3(egrid), dim3(256), lds, st>>>(
                    P.ln_msg_w0[l - 2], 0, nullptr, w2, b2, WT.msg_w0t[l - 1] + 2 * H * H, WT.msg_w2t[l - 1],
                    wp(W.ps[l - 2]), wp(W.pr[l - 2]), wp(W.e[l - 2]), nullptr, send_s, recv_s, rowptr, wp(W.DN),
                    wp(W.DE), l < 4 ? 1 : 0, bG, bH1, bDP2, nullptr, E);
            }
            delete pse;
            if (!acc_path) {
                L.add(bDP2, H, H, bH1, H, H, E, gw2, H, gb2);
                if (l == 1) L.add(bG, H, H, wp(W.feat), FPAD, F1, E, Gr.l1_msg_w0, F1, Gr.l1_msg_b0);
                else L.add(bG, H, H, wp(W.e[l - 2]), H, H, E, Gr.ln_msg_w0[l - 2] + 2 * H, 3 * H, nullptr);
            }
        } else {
            HIP_OK(hipMemsetAsync(gw2, 0, (size_t)H * H * 4, st));
            HIP_OK(hipMemsetAsync(gb2, 0, (size_t)H * 4, st));
            HIP_OK(hipMemsetAsync(bG, 0, 256, st));
            if (l == 1) {
                HIP_OK(hipMemsetAsync(Gr.l1_msg_w0, 0, (size_t)H * F1 * 4, st));
                HIP_OK(hipMemsetAsync(Gr.l1_msg_b0, 0, (size_t)H * 4, st));
            } else {
                L.add(bDPS, H, H, bDPS, H, H, 0, Gr.ln_msg_w0[l - 2] + 2 * H, 3 * H, nullptr);
            }
        }
        if (l >= 2) {
            // ---- sums of G onto nodes, then dx_{l-1}
            { ProfScope ps(KB_GATHER, st);
            if (E <= 64 * Nn) {
                kb_gather<<<dim3(ngrid), dim3(1024), 0, st>>>(bG, rowptr, srowptr, sperm, WT.msg_w0t[l - 1],
                                                             wp(W.DN), bDPS, bDPR, dx_nxt, Nn);
            } else {
                kb_sum_g<<<dim3((unsigned)((Nn + 3) / 4)), dim3(256), 0, st>>>(bG, rowptr, srowptr, sperm, bDPS, bDPR, Nn);
                kb_gather_rows<<<dim3(ngrid), dim3(64), 0, st>>>(WT.msg_w0t[l - 1], bDPS, bDPR, wp(W.DN), dx_nxt, Nn);
            } }
            L.add(bDPS, H, H, wp(W.x[l - 1]), H, H, Nn, Gr.ln_msg_w0[l - 2], 3 * H, nullptr);
            L.add(bDPR, H, H, wp(W.x[l - 1]), H, H, Nn, Gr.ln_msg_w0[l - 2] + H, 3 * H, Gr.ln_msg_b0[l - 2]);
            if (flush()) return AETHER_EHIP;
        } else {
            // ---- res + field net
            { ProfScope ps(KB_FIELD, st);
            if (grad_field != nullptr)
                kb_field<D, true><<<dim3((unsigned)((Nn + 7) / 8)), dim3(256), 0, st>>>(
                    P, x, vel, charges, wp(W.nodeinfo), wp(W.DA), wp(W.DN), rowptr, recv_s, srowptr, sperm,
                    wp(W.RELF), wp(W.Z), wp(W.H1f), wp(W.H2f), wp(W.DPH1), wp(W.DPH2), wp(W.DF), wp(W.DZE),
                    wp(W.ONEHOT), grad_field, Nn);
            else
                kb_field<D, false><<<dim3((unsigned)((Nn + 7) / 8)), dim3(256), 0, st>>>(
                    P, x, vel, charges, wp(W.nodeinfo), wp(W.DA), wp(W.DN), rowptr, recv_s, srowptr, sperm,
                    wp(W.RELF), wp(W.Z), wp(W.H1f), wp(W.H2f), wp(W.DPH1), wp(W.DPH2), wp(W.DF), wp(W.DZE),
                    wp(W.ONEHOT), nullptr, Nn); }
            L.add(wp(W.DN), H, H, wp(W.RELF), 16, 3 * D, Nn, Gr.l1_res_w, 3 * D, Gr.l1_res_b);
            if (grad_field == nullptr) {
                L.add(wp(W.DF), 16, D, wp(W.H2f), 32, 32, Nn, Gr.field_w4, 32, Gr.field_b4);
                L.add(wp(W.DPH2), 32, 32, wp(W.H1f), 32, 32, Nn, Gr.field_w2, 32, Gr.field_b2);
                L.add(wp(W.DPH1), 32, 32, wp(W.Z), 32, FIN, Nn, Gr.field_w0, FIN, Gr.field_b0);
                L.add(wp(W.ONEHOT), 16, 3, wp(W.DZE), 16, 16, Nn, Gr.field_emb, 16, nullptr);
            }
        }
    }
    if (run_outer(L, partial, W.partial_cap, st)) return AETHER_EHIP;
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

int g_fused_backward = 1;      // aether_set_option("fused_backward", 0|1): one-launch GNN backward for small-graph groups

template <int D>
bool fused_backward_applies(const AetherGraphInfo& info, int64_t Nn, int64_t E) {
    if (!g_fused_backward || info.n_groups <= 0 || E <= 0) return false;
    if (info.max_group_edges > 12 * 16) return false;         // three tiles per wave (split N=20 graphs: 12 tiles); beyond: backward.h
    WsLayout W(Nn, E, D, true);
    if (!W.defer) return false;                               // per-layer row tensors must exist
    if ((size_t)info.n_groups > W.fpart_wgs) return false;
    if ((info.reserved & 1) && (size_t)info.n_groups > W.xchg_wgs) return false;
    return true;
}

// The GNN part of the backward as ONE launch (fused_bwd.h) between the out-MLP kernel and the field kernel; the
// node-level weight-gradient products stay with k_outer, the edge-level ones come out of k_fb_reduce.
template <int D>
int backward_fused_impl(const AetherParams& P, const AetherParams& Gr, int64_t Nn, int64_t E, const AetherGraphInfo& info,
                        const float* x, const float* vel, const float* charges, const char* graph, char* ws,
                        const float* g_out, hipStream_t st, float* grad_field = nullptr) {
    constexpr int F1 = 7 * D + D * (D - 1) / 2 + 2;
    constexpr int FIN = 2 * D + 16;
    GraphLayout G(E, Nn, false);
    WsLayout W(Nn, E, D, true);
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(graph + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const int32_t *recv_s = gp(G.recv_s), *rowptr = gp(G.rowptr);
    const int32_t *sperm = gp(G.sperm), *srowptr = gp(G.srowptr);
    OuterList L;
    const unsigned ngrid = (unsigned)((Nn + 15) / 16);
    // ---- transposed weight copies (one launch)
    TransposeBatch TB;
    const BwdWT WT = transposed_weights<D>(P, wp(W.wt), TB);     // written by the forward (prepare_weights)
    // ---- out MLP
    { ProfScope ps(KB_OUT, st);
    kb_out<D><<<dim3(ngrid), dim3(256), 0, st>>>(P, WT, wp(W.x[4]), wp(W.nodeinfo), g_out, wp(W.DXl[4]), wp(W.O1),
                                               wp(W.O2), wp(W.DPO1), wp(W.DPO2), wp(W.DY), Nn, wp(W.drop),
                                               wp(W.drop) + (size_t)Nn * H, reinterpret_cast<const int*>(ws + W.dropword)); }
    L.add(wp(W.DPO1), H, H, wp(W.x[4]), H, H, Nn, Gr.out_w0, H, Gr.out_b0);
    L.add(wp(W.DPO2), H, H, wp(W.O1), H, H, Nn, Gr.out_w3, H, Gr.out_b3);
    L.add(wp(W.DY), 16, D, wp(W.O2), H, H, Nn, Gr.out_w6, H, Gr.out_b6);
    // ---- the four GNN layers: one launch
    {
        FbArgs A;
        A.rowptr = rowptr; A.send_s = gp(G.send_s); A.recv_s = recv_s;
        A.wgdesc = reinterpret_cast<const FusedWG*>(graph + G.wgdesc);
        A.lorder = gp(G.lorder); A.ledge = reinterpret_cast<const int4*>(graph + G.ledge); A.nrange = gp(G.nrange);
        for (int k = 0; k < 4; ++k) {                           // layer k + 1
            FbLayer& Y = A.layer[k];
            Y.e_prev = k == 0 ? wp(W.feat) : wp(W.e[k - 1]);
            Y.n = wp(W.n[k]);
            Y.ps = k == 0 ? nullptr : wp(W.ps[k - 1]);
            Y.pr = k == 0 ? nullptr : wp(W.pr[k - 1]);
            Y.msg_w0 = k == 0 ? P.l1_msg_w0 : P.ln_msg_w0[k - 1];
            Y.msg_w2 = k == 0 ? P.l1_msg_w2 : P.ln_msg_w2[k - 1];
            Y.msg_b2 = k == 0 ? P.l1_msg_b2 : P.ln_msg_b2[k - 1];
            Y.upd_w0 = k == 0 ? P.l1_upd_w0 : P.ln_upd_w0[k - 1];
            Y.upd_b0 = k == 0 ? P.l1_upd_b0 : P.ln_upd_b0[k - 1];
            Y.w4t = WT.upd_w2t[k]; Y.w3t = WT.upd_w0t[k]; Y.w2t = WT.msg_w2t[k]; Y.w0t = WT.msg_w0t[k];
            Y.img_e = wp(W.wimg) + fused_wimg_offset(k + 1, 0);      // written by the forward of this step (prepare_weights)
            Y.img_2 = wp(W.wimg) + fused_wimg_offset(k + 1, 1);
            Y.U = wp(W.Ul[k]); Y.DPU = wp(W.DPUl[k]); Y.DPS = wp(W.DPSl[k]); Y.DPR = wp(W.DPRl[k]);
            Y.DXout = k == 0 ? nullptr : wp(W.DXl[k]);         // layer k + 1 produces dL/dx_k
        }
        A.msg_b0_1 = P.l1_msg_b0;
        A.f1 = F1;
        A.dx4 = wp(W.DXl[4]);
        A.DN1 = wp(W.DN);
        A.DA = wp(W.DA);
        A.DE = wp(W.DE);
        A.partial = wp(W.fpart);
        A.xchg = wp(W.xchg);
        A.flags = reinterpret_cast<int*>(const_cast<char*>(graph) + G.hflags) + G.max_wgs + 32;
        A.errword = async_error_word();
        A.n_wgs = info.n_groups;
        A.stamps = wp(W.stamps);
        const int tiles = (info.max_group_edges + 15) / 16;
        const size_t lds = (size_t)FbLds::TOTAL * 4;
        ProfScope ps(KB_FUSED, st);
#define AETHER_FB_CASE(R)                                                                                   \
        do {                                                                                                \
            if (ensure_dynamic_lds(reinterpret_cast<const void*>(k_fused_bwd<R>), lds)) return AETHER_EHIP; \
            k_fused_bwd<R><<<dim3((unsigned)info.n_groups), dim3(FB_THREADS), lds, st>>>(A);              \
        } while (0)
        if (tiles <= 4) AETHER_FB_CASE(1);
        else if (tiles <= 8) AETHER_FB_CASE(2);
        else AETHER_FB_CASE(3);
#undef AETHER_FB_CASE
    }
    // ---- node-level weight-gradient products of the layers
    for (int l = 4; l >= 1; --l) {
        float* gw3 = l == 1 ? Gr.l1_upd_w0 : Gr.ln_upd_w0[l - 2];
        float* gb3 = l == 1 ? Gr.l1_upd_b0 : Gr.ln_upd_b0[l - 2];
        float* gw4 = l == 1 ? Gr.l1_upd_w2 : Gr.ln_upd_w2[l - 2];
        float* gb4 = l == 1 ? Gr.l1_upd_b2 : Gr.ln_upd_b2[l - 2];
        L.add(wp(W.DXl[l]), H, H, wp(W.Ul[l - 1]), 2 * H, 2 * H, Nn, gw4, 2 * H, gb4);
        L.add(wp(W.DPUl[l - 1]), 2 * H, 2 * H, wp(W.n[l - 1]), H, H, Nn, gw3, H, gb3);
        if (l >= 2) {
            L.add(wp(W.DPSl[l - 1]), H, H, wp(W.x[l - 1]), H, H, Nn, Gr.ln_msg_w0[l - 2], 3 * H, nullptr);
            L.add(wp(W.DPRl[l - 1]), H, H, wp(W.x[l - 1]), H, H, Nn, Gr.ln_msg_w0[l - 2] + H, 3 * H, Gr.ln_msg_b0[l - 2]);
        }
    }
    // ---- res + field net
    { ProfScope ps(KB_FIELD, st);
    if (grad_field != nullptr)
        kb_field<D, true><<<dim3((unsigned)((Nn + 7) / 8)), dim3(256), 0, st>>>(
            P, x, vel, charges, wp(W.nodeinfo), wp(W.DA), wp(W.DN), rowptr, recv_s, srowptr, sperm,
            wp(W.RELF), wp(W.Z), wp(W.H1f), wp(W.H2f), wp(W.DPH1), wp(W.DPH2), wp(W.DF), wp(W.DZE),
            wp(W.ONEHOT), grad_field, Nn);
    else
        kb_field<D, false><<<dim3((unsigned)((Nn + 7) / 8)), dim3(256), 0, st>>>(
            P, x, vel, charges, wp(W.nodeinfo), wp(W.DA), wp(W.DN), rowptr, recv_s, srowptr, sperm,
            wp(W.RELF), wp(W.Z), wp(W.H1f), wp(W.H2f), wp(W.DPH1), wp(W.DPH2), wp(W.DF), wp(W.DZE),
            wp(W.ONEHOT), nullptr, Nn); }
    L.add(wp(W.DN), H, H, wp(W.RELF), 16, 3 * D, Nn, Gr.l1_res_w, 3 * D, Gr.l1_res_b);
    if (grad_field == nullptr) {
        L.add(wp(W.DF), 16, D, wp(W.H2f), 32, 32, Nn, Gr.field_w4, 32, Gr.field_b4);
        L.add(wp(W.DPH2), 32, 32, wp(W.H1f), 32, 32, Nn, Gr.field_w2, 32, Gr.field_b2);
        L.add(wp(W.DPH1), 32, 32, wp(W.Z), 32, FIN, Nn, Gr.field_w0, FIN, Gr.field_b0);
        L.add(wp(W.ONEHOT), 16, 3, wp(W.DZE), 16, 16, Nn, Gr.field_emb, 16, nullptr);
    }
    // ---- edge-level weight gradients: the workgroups' partials, added in workgroup order -- in the launch that adds up
    // the node-level products' partials
    {
        FbReduceArgs R;
        R.partial = wp(W.fpart);
        R.n_wgs = info.n_groups;
        for (int k = 0; k < 4; ++k) {
            R.w2[k] = k == 0 ? Gr.l1_msg_w2 : Gr.ln_msg_w2[k - 1];
            R.b2[k] = k == 0 ? Gr.l1_msg_b2 : Gr.ln_msg_b2[k - 1];
            R.we[k] = k == 0 ? Gr.l1_msg_w0 : Gr.ln_msg_w0[k - 1] + 2 * H;
        }
        R.b1 = Gr.l1_msg_b0;
        R.f1 = F1;
        if (run_outer(L, wp(W.partial), W.partial_cap, st, &R)) return AETHER_EHIP;
    }
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

#include "wide_impl.inc"

}  // namespace

// =================================================================== C ABI
extern "C" {

const char* aether_version(void) { return "aether_hip 0.7 (gfx950; state2state fused + streamed + wide + fused backward, loss + AdamW, seq2seq fused step, variable-N steps, kNN, simulators; dense layers as split-fp16 matrix-core GEMMs)"; }
const char* aether_last_error(void) { return g_err; }

#include "host_seq2seq.inc"
#include "host_s2s_step.inc"
#include "host_dynamicvars.inc"
#include "host_dyn_step.inc"
#include "host_sim.inc"
#include "host_train.inc"

int aether_set_option(const char* name, int value) {
    if (!name) return fail(AETHER_EINVAL, "set_option: null name");
    if (!strcmp(name, "fused_split")) {      // takes effect at the next aether_graph_build
        g_fused_split = value != 0;
        return AETHER_OK;
    }
    if (!strcmp(name, "fused_pair_stride")) {     // takes effect at the next aether_graph_build
        if (value != 1 && value != 8) return fail(AETHER_EINVAL, "set_option: fused_pair_stride is 1 or 8");
        g_fused_pair_stride = (int)value;
        return AETHER_OK;
    }
    if (!strcmp(name, "fused_backward")) {   // 0: layer-by-layer backward kernels even for small-graph groups
        g_fused_backward = value != 0;
        return AETHER_OK;
    }
    if (!strcmp(name, "edge_acc")) {                // changes aether_workspace_bytes(): set before sizing workspaces
        g_edge_acc = value < 0 ? 0 : (value > 3 ? 3 : value);      // 0: row tensors + k_outer, 1: kb_edge_acc (round 3), 2 / 3: kb_edge_acc8<8 / 4 waves>
        return AETHER_OK;
    }
    if (!strcmp(name, "outer_defer_max_edges")) {   // changes aether_workspace_bytes(): set before sizing workspaces
        if (value < 0) return fail(AETHER_EINVAL, "set_option: outer_defer_max_edges must be >= 0");
        g_outer_defer_max_edges = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "outer_tiles_per_wave")) {
        if (value < 0 || value > 64) return fail(AETHER_EINVAL, "set_option: outer_tiles_per_wave must be 0..64");
        g_outer_tiles_per_wave = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "linear_small_wgs")) {
        g_linear_small_wgs = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "linear_kwaves")) {
        if (value != 1 && value != 4) return fail(AETHER_EINVAL, "set_option: linear_kwaves must be 1 or 4");
        g_linear_kwaves = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "gemm_split")) {
        if (value < 0 || value > 3) return fail(AETHER_EINVAL, "set_option: gemm_split must be 0, 1, 2 or 3");
        g_gemm_split = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "dyn_filter_v1")) {
        if (value < 0 || value > 2) return fail(AETHER_EINVAL, "set_option: dyn_filter_v1 must be 0, 1 or 2");
        g_dyn_filter_v1 = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "dyn_filter_v1_edges")) { g_dyn_filter_v1_edges = value; return AETHER_OK; }
    if (!strcmp(name, "filter_rsplits")) {
        if (value != 0 && 15 % value != 0) return fail(AETHER_EINVAL, "set_option: filter_rsplits is 0 (auto), 1, 3, 5 or 15");
        g_filter_rsplits = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "filter_wgs")) {
        if (value < 8 || value % 8 != 0) return fail(AETHER_EINVAL, "set_option: filter_wgs must be a multiple of 8");
        g_filter_wgs = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "filter_splits")) {           // changes the seq2seq / variable-N prior workspace sizes
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(AETHER_EINVAL, "set_option: filter_splits must be 0 (automatic), 1, 2, 4 or 8");
        g_filter_splits = value;
        return AETHER_OK;
    }
    if (!strcmp(name, "filter_wg_target")) {        // changes the seq2seq / variable-N prior and decoder workspace sizes
        if (value < 1) return fail(AETHER_EINVAL, "set_option: filter_wg_target must be >= 1");
        g_filter_wg_target = value;
        return AETHER_OK;
    }
    return fail(AETHER_EINVAL, "set_option: unknown option");
}

int aether_check_async_error(void) { return take_async_error(); }

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
                       void* graph, size_t graph_bytes, AetherGraphInfo* info, void* stream) {
    if (n_nodes <= 0 || n_edges < 0 || !graph || !info) return fail(AETHER_EINVAL, "graph_build: bad sizes");
    memset(info, 0, sizeof(*info));
    info->n_nodes = n_nodes;
    info->n_edges = n_edges;
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
        HIP_OK(hipMemsetAsync(g + G.srowptr, 0, (size_t)(n_nodes + 1) * 4, st));
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
    k_graph_gtiles<<<dim3((unsigned)((n_edges + 15) / 16)), dim3(64), 0, st>>>(recv_s, n_edges,
                                                                              (uint32_t*)(g + G.gsel));
    // sender lists for the backward: stable sort of the receiver-sorted positions by sender
    {
        int32_t* send_s = (int32_t*)(g + G.send_s);
        k_iota<<<dim3(blocks), dim3(256), 0, st>>>(vals, n_edges);
        size_t cb = G.cub_bytes;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(g + G.cub, cb, send_s, keys, vals, (int32_t*)(g + G.sperm),
                                                  (int)n_edges, 0, bits_for(n_nodes), st));
        k_rowptr<<<dim3(blocks), dim3(256), 0, st>>>(keys, n_edges, n_nodes, (int32_t*)(g + G.srowptr));
    }
    HIP_OK(hipGetLastError());
    int32_t bad = 0;
    HIP_OK(hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (bad) return fail(AETHER_EINDEX, "graph_build: edge index outside [0, n_nodes)");

    // ---- groups for the fused kernel: contiguous node ranges closed under the edge relation ----
    int32_t* diff = (int32_t*)(g + G.diff);
    int32_t* cross = (int32_t*)(g + G.cross);
    HIP_OK(hipMemsetAsync(diff, 0, (size_t)(n_nodes + 2) * 4, st));
    k_graph_cross<<<dim3(blocks), dim3(256), 0, st>>>(send, recv, n_edges, n_nodes, diff);
    size_t scan_bytes = G.cub_bytes;
    HIP_OK(hipcub::DeviceScan::InclusiveSum(g + G.cub, scan_bytes, diff, cross, (int)(n_nodes + 1), st));
    std::vector<int32_t> h_cross((size_t)n_nodes + 1), h_rowptr((size_t)n_nodes + 1);
    HIP_OK(hipMemcpyAsync(h_cross.data(), cross, (size_t)(n_nodes + 1) * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(h_rowptr.data(), rowptr, (size_t)(n_nodes + 1) * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    // components = maximal ranges with no crossing edge at their boundaries
    const int cap_n = FUSED_MAX_NODES, cap_e = FUSED_MAX_EDGES;
    bool ok = true;
    int64_t max_cn = 0, max_ce = 0;
    for (int64_t c0 = 0; c0 < n_nodes && ok;) {
        int64_t c1 = c0 + 1;
        while (c1 < n_nodes && h_cross[c1] != 0) ++c1;
        int64_t cn = c1 - c0, ce = h_rowptr[c1] - h_rowptr[c0];
        if (cn > cap_n || ce > cap_e) ok = false;
        if (cn > max_cn) max_cn = cn;
        if (ce > max_ce) max_ce = ce;
        c0 = c1;
    }
    if (ok) {
        // pack whole components into groups; aim for >= 256 groups (one per CU) before filling them
        int64_t tgt_n = (n_nodes + 255) / 256, tgt_e = (n_edges + 255) / 256;
        if (tgt_n < max_cn) tgt_n = max_cn;
        if (tgt_e < max_ce) tgt_e = max_ce;
        if (tgt_n > cap_n) tgt_n = cap_n;
        if (tgt_e > cap_e) tgt_e = cap_e;
        std::vector<int32_t> grp;
        grp.push_back(0);
        int64_t gs = 0, mgn = 0, mge = 0;
        for (int64_t c0 = 0; c0 < n_nodes;) {
            int64_t c1 = c0 + 1;
            while (c1 < n_nodes && h_cross[c1] != 0) ++c1;
            if (c0 > gs && (c1 - gs > tgt_n || h_rowptr[c1] - h_rowptr[gs] > tgt_e)) {
                grp.push_back((int32_t)c0);
                gs = c0;
            }
            if (c1 - gs > mgn) mgn = c1 - gs;
            if (h_rowptr[c1] - h_rowptr[gs] > mge) mge = h_rowptr[c1] - h_rowptr[gs];
            c0 = c1;
        }
        grp.push_back((int32_t)n_nodes);
        const int n_grp = (int)grp.size() - 1;
        if (G.max_wgs >= 2 * (int64_t)n_grp) {
            // Workgroup descriptors.  With fewer groups than half the CUs, each group is split over two
            // workgroups by receiver range (balanced by in-edge count); see FusedWG.
            int cus = 256, dev = 0;       // (hipGetDeviceProperties takes ~100 ms per call: one attribute instead)
            if (hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
                cus = 256;
            bool split = g_fused_split && 2 * n_grp <= cus;
            for (int k = 0; k < n_grp && split; ++k) if (grp[k + 1] - grp[k] < 2) split = false;
            std::vector<FusedWG> wgs;
            std::vector<FusedTile> tiles;
            mgn = 0; mge = 0;
            auto add_wg = [&](int vb, int ve, int nb, int ne, int partner) {
                const int m = h_rowptr[ne] - h_rowptr[nb];
                FusedWG w = {vb, ve, nb, ne, (int)tiles.size(), partner, 0, (int)h_rowptr[nb], m, 0, 0, 0};
                for (int t = 0; t < (m + 15) / 16; ++t) tiles.push_back(FusedTile{(int)wgs.size(), t});
                if (ne - nb > mgn) mgn = ne - nb;
                if (m > mge) mge = m;
                wgs.push_back(w);
            };
            for (int pass = 0; pass < 2; ++pass) {
            wgs.clear(); tiles.clear(); mgn = 0; mge = 0;
            auto split_point = [&](int a, int b) {
                const int64_t tot = h_rowptr[b] - h_rowptr[a];
                int nm = a + 1;
                int64_t best = -1;
                for (int c = a + 1; c < b; ++c) {
                    int64_t left = h_rowptr[c] - h_rowptr[a], d = left * 2 > tot ? left * 2 - tot : tot - left * 2;
                    if (best < 0 || d < best) { best = d; nm = c; }
                }
                return nm;
            };
            // Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8: tools/micro/xcc_map.hip).
            // With option "fused_pair_stride" = 8 the two halves of a group sit 8 apart (blocks of 16: the eight first
            // halves, then the eight second halves), i.e. on the SAME XCD and behind the same L2; 1 = adjacent (different XCDs).
            const int pstride = split ? g_fused_pair_stride : 1;
            for (int k0 = 0; k0 < n_grp; k0 += pstride) {
                const int kn = std::min(pstride, n_grp - k0);
                if (!split) { add_wg(grp[k0], grp[k0 + 1], grp[k0], grp[k0 + 1], -1); continue; }
                const int base = (int)wgs.size();
                for (int j = 0; j < kn; ++j) {
                    const int a = grp[k0 + j], b = grp[k0 + j + 1];
                    add_wg(a, b, a, split_point(a, b), base + kn + j);
                }
                for (int j = 0; j < kn; ++j) {
                    const int a = grp[k0 + j], b = grp[k0 + j + 1];
                    add_wg(a, b, split_point(a, b), b, base + j);
                }
            }
            // a split workgroup keeps two runs of partial rows in LDS (FusedLds::PART_ROWS): at most 16 tiles
            if (split && mge > 16 * 16) { split = false; continue; }
            break;
            }
            if ((int64_t)tiles.size() <= G.max_tiles) {
                HIP_OK(hipMemcpyAsync(g + G.wgdesc, wgs.data(), wgs.size() * sizeof(FusedWG), hipMemcpyHostToDevice, st));
                HIP_OK(hipMemsetAsync(g + G.hflags, 0, (size_t)(2 * G.max_wgs + 64) * 4, st));
                k_graph_lorder<<<dim3((unsigned)wgs.size()), dim3(512), 0, st>>>(
                    (FusedWG*)(g + G.wgdesc), (const int32_t*)(g + G.send_s), recv_s, (const int32_t*)(g + G.perm), rowptr,
                    (int32_t*)(g + G.lorder), (int4*)(g + G.ledge), (int32_t*)(g + G.nrange));
                if (!tiles.empty()) {
                    HIP_OK(hipMemcpyAsync(g + G.tdesc, tiles.data(), tiles.size() * sizeof(FusedTile),
                                          hipMemcpyHostToDevice, st));
                    k_graph_tiles<<<dim3((unsigned)tiles.size()), dim3(64), 0, st>>>(
                        (const FusedTile*)(g + G.tdesc), (const FusedWG*)(g + G.wgdesc), recv_s, rowptr,
                        (const int32_t*)(g + G.lorder), (uint32_t*)(g + G.tsel), (uint32_t*)(g + G.tdst));
                }
                HIP_OK(hipStreamSynchronize(st));       // host vectors must outlive the copies
                info->n_groups = (int32_t)wgs.size();
                info->max_group_nodes = (int32_t)mgn;
                info->max_group_edges = (int32_t)mge;
                info->reserved = split ? 1 : 0;
            }
        }
    }
    return AETHER_OK;
}

int aether_graph_matches(const int64_t* send, const int64_t* recv, int64_t n_edges, int64_t n_nodes, const void* graph,
                         void* stream) {
    if (!graph || n_edges < 0 || n_nodes <= 0) return fail(AETHER_EINVAL, "graph_matches: bad arguments");
    if (n_edges == 0) return 1;
    if (!send || !recv) return fail(AETHER_EINVAL, "graph_matches: null edge index");
    static int* h_flag = nullptr;
    static int* d_flag = nullptr;
    if (!h_flag) {
        void* h = nullptr;
        HIP_OK(hipHostMalloc(&h, 64, hipHostMallocMapped));
        void* d = nullptr;
        HIP_OK(hipHostGetDevicePointer(&d, h, 0));
        h_flag = (int*)h; d_flag = (int*)d;
    }
    GraphLayout G(n_edges, n_nodes, false);
    const char* g = (const char*)graph;
    hipStream_t st = (hipStream_t)stream;
    *(volatile int*)h_flag = 0;
    k_graph_match<<<dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, st>>>(
        send, recv, n_edges, (const int32_t*)(g + G.perm), (const int32_t*)(g + G.send_s), (const int32_t*)(g + G.recv_s), d_flag);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(st));
    return *(volatile int*)h_flag ? 0 : 1;
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
    if (n_nodes <= 0 || n_edges < 0 || (num_dims != 2 && num_dims != 3)) return 0;
    return WsLayout(n_nodes, n_edges, num_dims, keep_for_backward != 0).total;
}

size_t aether_dropout_mask_offset(int64_t n_nodes, int64_t n_edges, int num_dims) {
    if (n_nodes <= 0 || n_edges < 0 || (num_dims != 2 && num_dims != 3)) return 0;
    return WsLayout(n_nodes, n_edges, num_dims, true).drop;
}

static int forward_common(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                          const float* x, const float* vel, const float* charges,
                          const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                          void* workspace, size_t workspace_bytes, float* out, int flags, void* stream,
                          const float* field) {
    if (!params || !x || !vel || !charges || !graph || !info || !workspace || !out)
        return fail(AETHER_EINVAL, "forward: null pointer");
    if (take_async_error()) return AETHER_EHIP;
    if (info->n_nodes != n_nodes || info->n_edges != n_edges)
        return fail(AETHER_EINVAL, "forward: graph info does not match n_nodes / n_edges");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "forward: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0) return fail(AETHER_EINVAL, "forward: bad sizes");
    if (n_edges > 0 && !edge_attr_orig) return fail(AETHER_EINVAL, "forward: null edge_attr");
    if (workspace_bytes <
        aether_workspace_bytes(n_nodes, n_edges, num_dims, (flags & AETHER_FLAG_KEEP_INTERMEDIATES) ? 1 : 0))
        return fail(AETHER_ESPACE, "forward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool fused = info->n_groups > 0 && n_edges > 0 && !(flags & AETHER_FLAG_FORCE_STREAMED);
    if ((flags & AETHER_FLAG_FORCE_FUSED) && !fused)
        return fail(AETHER_EINVAL, "forward: fused path requested but the graph has no groups");
    const bool keep = (flags & AETHER_FLAG_KEEP_INTERMEDIATES) != 0;
    const bool reused = (flags & AETHER_FLAG_WORKSPACE_REUSED) != 0;
    const bool prepared = (flags & AETHER_FLAG_WEIGHTS_PREPARED) != 0;
    StepExtras no_extras{nullptr, nullptr, 1.0f, field, (flags & AETHER_FLAG_BACKWARD_ONLY) != 0};
    if ((flags & AETHER_FLAG_DROPOUT) && !keep)
        return fail(AETHER_EINVAL, "forward: AETHER_FLAG_DROPOUT needs AETHER_FLAG_KEEP_INTERMEDIATES (the masks live in the training workspace)");
    if (keep) {
        const WsLayout W(n_nodes, n_edges, num_dims, true);
        float* masks = reinterpret_cast<float*>((char*)workspace + W.drop);
        if (flags & AETHER_FLAG_DROPOUT) { no_extras.drop1 = masks; no_extras.drop2 = masks + (size_t)n_nodes * H; }
        no_extras.dropword = reinterpret_cast<int*>((char*)workspace + W.dropword);
    }
    if (fused) {
        if (num_dims == 2)
            return fused_impl<2>(*params, n_nodes, n_edges, *info, x, vel, charges, edge_attr_orig,
                                 (const char*)graph, (char*)workspace, out, keep, reused, no_extras, st, prepared);
        return fused_impl<3>(*params, n_nodes, n_edges, *info, x, vel, charges, edge_attr_orig,
                             (const char*)graph, (char*)workspace, out, keep, reused, no_extras, st, prepared);
    }
    if (num_dims == 2)
        return streamed_impl<2>(*params, n_nodes, n_edges, x, vel, charges, edge_attr_orig,
                                (const char*)graph, (char*)workspace, out, keep, no_extras, st);
    return streamed_impl<3>(*params, n_nodes, n_edges, x, vel, charges, edge_attr_orig,
                            (const char*)graph, (char*)workspace, out, keep, no_extras, st);
}

int aether_forward(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                   const float* x, const float* vel, const float* charges,
                   const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                   void* workspace, size_t workspace_bytes, float* out, int flags, void* stream) {
    return forward_common(params, num_dims, n_nodes, n_edges, x, vel, charges, edge_attr_orig, graph, info, workspace,
                          workspace_bytes, out, flags, stream, nullptr);
}

int aether_forward_field(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges,
                         const float* x, const float* vel, const float* charges, const float* field,
                         const float* edge_attr_orig, const void* graph, const AetherGraphInfo* info,
                         void* workspace, size_t workspace_bytes, float* out, int flags, void* stream) {
    if (!field) return fail(AETHER_EINVAL, "forward_field: null field");
    return forward_common(params, num_dims, n_nodes, n_edges, x, vel, charges, edge_attr_orig, graph, info, workspace,
                          workspace_bytes, out, flags, stream, field);
}

int aether_dynamic_field(const AetherDynFieldParams* p, int num_dims, int64_t n_graphs, int nodes_per_graph,
                         const float* x, const float* vel, const float* charges, float* field, void* stream) {
    if (!p || !x || !vel || !charges || !field) return fail(AETHER_EINVAL, "dynamic_field: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "dynamic_field: num_dims must be 2 or 3");
    if (n_graphs <= 0 || nodes_per_graph <= 0 || nodes_per_graph > DYNFIELD_MAX_NODES)
        return fail(AETHER_EINVAL, "dynamic_field: 1..2048 nodes per graph");
    if (n_graphs >= ((int64_t)1 << 31)) return fail(AETHER_EINVAL, "dynamic_field: too many graphs");
    hipStream_t st = (hipStream_t)stream;
    if (num_dims == 2) k_dynfield<2><<<dim3((unsigned)n_graphs), dim3(256), 0, st>>>(*p, x, vel, charges, field, nodes_per_graph);
    else k_dynfield<3><<<dim3((unsigned)n_graphs), dim3(256), 0, st>>>(*p, x, vel, charges, field, nodes_per_graph);
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

static int rollout_common(const AetherParams* params, const AetherDynFieldParams* dyn, int nodes_per_graph, float* field_buf,
                          int num_dims, int64_t n_nodes, int64_t n_edges, const float* x0, const float* vel0,
                          const float* charges, const void* graph, const AetherGraphInfo* info, void* workspace,
                          size_t workspace_bytes, float* trajectory, int steps, float dt, int flags, void* stream) {
    if (!params || !x0 || !vel0 || !charges || !graph || !info || !workspace || !trajectory)
        return fail(AETHER_EINVAL, "rollout: null pointer");
    if (take_async_error()) return AETHER_EHIP;
    if (info->n_nodes != n_nodes || info->n_edges != n_edges)
        return fail(AETHER_EINVAL, "rollout: graph info does not match n_nodes / n_edges");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "rollout: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0 || steps < 0) return fail(AETHER_EINVAL, "rollout: bad sizes");
    if (!(dt != 0.0f)) return fail(AETHER_EINVAL, "rollout: dt must be non-zero");
    if (flags & AETHER_FLAG_KEEP_INTERMEDIATES) return fail(AETHER_EINVAL, "rollout: inference only");
    if (workspace_bytes < aether_workspace_bytes(n_nodes, n_edges, num_dims, 0))
        return fail(AETHER_ESPACE, "rollout: workspace too small");
    if (dyn && (!field_buf || nodes_per_graph <= 0 || n_nodes % nodes_per_graph != 0))
        return fail(AETHER_EINVAL, "rollout: n_nodes must be a multiple of nodes_per_graph (and a field buffer given)");
    hipStream_t st = (hipStream_t)stream;
    const bool fused = info->n_groups > 0 && n_edges > 0 && !(flags & AETHER_FLAG_FORCE_STREAMED);
    if ((flags & AETHER_FLAG_FORCE_FUSED) && !fused)
        return fail(AETHER_EINVAL, "rollout: fused path requested but the graph has no groups");
    WsLayout W(n_nodes, n_edges, num_dims, false);
    char* ws = (char*)workspace;
    const size_t stride = (size_t)n_nodes * num_dims;
    for (int t = 0; t < steps; ++t) {
        const float* x = t == 0 ? x0 : trajectory + (size_t)(t - 1) * stride;
        const float* v = t == 0 ? vel0 : reinterpret_cast<const float*>(ws + W.velbuf[(t - 1) & 1]);
        float* out = trajectory + (size_t)t * stride;
        if (dyn) {              // the dynamic-field model: LatentFieldNetwork on the current state, then the step
            int rc = aether_dynamic_field(dyn, num_dims, n_nodes / nodes_per_graph, nodes_per_graph, x, v, charges, field_buf, stream);
            if (rc != AETHER_OK) return rc;
        }
        StepExtras ex{charges, reinterpret_cast<float*>(ws + W.velbuf[t & 1]), dt, dyn ? field_buf : nullptr};
        const bool reused = t > 0 || (flags & AETHER_FLAG_WORKSPACE_REUSED);   // the step before re-armed the flags
        const bool prepared = t > 0 || (flags & AETHER_FLAG_WEIGHTS_PREPARED);  // one weight conversion per rollout
        int rc;
        if (fused)
            rc = num_dims == 2 ? fused_impl<2>(*params, n_nodes, n_edges, *info, x, v, charges, nullptr, (const char*)graph,
                                               ws, out, false, reused, ex, st, prepared)
                               : fused_impl<3>(*params, n_nodes, n_edges, *info, x, v, charges, nullptr, (const char*)graph,
                                               ws, out, false, reused, ex, st, prepared);
        else
            rc = num_dims == 2 ? streamed_impl<2>(*params, n_nodes, n_edges, x, v, charges, nullptr, (const char*)graph, ws,
                                                  out, false, ex, st)
                               : streamed_impl<3>(*params, n_nodes, n_edges, x, v, charges, nullptr, (const char*)graph, ws,
                                                  out, false, ex, st);
        if (rc != AETHER_OK) return rc;
    }
    return AETHER_OK;
}

int aether_rollout(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges, const float* x0,
                   const float* vel0, const float* charges, const void* graph, const AetherGraphInfo* info,
                   void* workspace, size_t workspace_bytes, float* trajectory, int steps, float dt,
                   int flags, void* stream) {
    return rollout_common(params, nullptr, 0, nullptr, num_dims, n_nodes, n_edges, x0, vel0, charges, graph, info, workspace,
                          workspace_bytes, trajectory, steps, dt, flags, stream);
}

int aether_rollout_dynamic_field(const AetherParams* params, const AetherDynFieldParams* dyn_params, int num_dims,
                                 int64_t n_nodes, int64_t n_edges, int nodes_per_graph, const float* x0, const float* vel0,
                                 const float* charges, const void* graph, const AetherGraphInfo* info, void* workspace,
                                 size_t workspace_bytes, float* field_scratch, float* trajectory, int steps, float dt,
                                 int flags, void* stream) {
    if (!dyn_params) return fail(AETHER_EINVAL, "rollout_dynamic_field: null pointer");
    return rollout_common(params, dyn_params, nodes_per_graph, field_scratch, num_dims, n_nodes, n_edges, x0, vel0, charges,
                          graph, info, workspace, workspace_bytes, trajectory, steps, dt, flags, stream);
}

int aether_backward(const AetherParams* params, const AetherParams* grads, int num_dims, int64_t n_nodes,
                    int64_t n_edges, const float* x, const float* vel, const float* charges,
                    const void* graph, const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                    const float* grad_out, void* stream) {
    if (!params || !grads || !x || !vel || !charges || !graph || !info || !workspace || !grad_out)
        return fail(AETHER_EINVAL, "backward: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "backward: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0 || info->n_nodes != n_nodes || info->n_edges != n_edges)
        return fail(AETHER_EINVAL, "backward: bad sizes");
    if (workspace_bytes < aether_workspace_bytes(n_nodes, n_edges, num_dims, 1))
        return fail(AETHER_ESPACE, "backward: workspace too small (forward must run with KEEP_INTERMEDIATES)");
    hipStream_t st = (hipStream_t)stream;
    if (take_async_error()) return AETHER_EHIP;
    if (num_dims == 2) {
        if (fused_backward_applies<2>(*info, n_nodes, n_edges))
            return backward_fused_impl<2>(*params, *grads, n_nodes, n_edges, *info, x, vel, charges, (const char*)graph,
                                          (char*)workspace, grad_out, st);
        return backward_impl<2>(*params, *grads, n_nodes, n_edges, x, vel, charges, (const char*)graph,
                                (char*)workspace, grad_out, st);
    }
    if (fused_backward_applies<3>(*info, n_nodes, n_edges))
        return backward_fused_impl<3>(*params, *grads, n_nodes, n_edges, *info, x, vel, charges, (const char*)graph,
                                      (char*)workspace, grad_out, st);
    return backward_impl<3>(*params, *grads, n_nodes, n_edges, x, vel, charges, (const char*)graph,
                            (char*)workspace, grad_out, st);
}

int aether_backward_inputs(const AetherParams* params, int num_dims, int64_t n_nodes, int64_t n_edges, const float* x,
                           const float* vel, const float* charges, const void* graph, const AetherGraphInfo* info,
                           void* workspace, size_t workspace_bytes, const float* out, const float* grad_out, float* grad_x,
                           float* grad_vel, float* grad_edge_attr, const float* field_input_grad, void* stream) {
    if (!params || !x || !vel || !charges || !graph || !info || !workspace || !out || !grad_out || !grad_x || !grad_vel)
        return fail(AETHER_EINVAL, "backward_inputs: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "backward_inputs: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0 || info->n_nodes != n_nodes || info->n_edges != n_edges)
        return fail(AETHER_EINVAL, "backward_inputs: bad sizes");
    if (workspace_bytes < aether_workspace_bytes(n_nodes, n_edges, num_dims, 1))
        return fail(AETHER_ESPACE, "backward_inputs: workspace too small (the workspace of the forward / aether_backward pair)");
    hipStream_t st = (hipStream_t)stream;
    if (take_async_error()) return AETHER_EHIP;
    GraphLayout G(n_edges, n_nodes, false);
    WsLayout W(n_nodes, n_edges, num_dims, true);
    const char* g = (const char*)graph;
    char* ws = (char*)workspace;
    auto gp = [&](size_t off) { return reinterpret_cast<const int32_t*>(g + off); };
    auto wp = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    const dim3 grid((unsigned)((n_nodes + 7) / 8));
    if (num_dims == 2)
        kb_inputs<2><<<grid, dim3(256), 0, st>>>(*params, x, vel, charges, wp(W.nodeinfo), out, grad_out, wp(W.DA), wp(W.DN),
                                               wp(W.DF), gp(G.rowptr), gp(G.send_s), gp(G.recv_s), gp(G.srowptr),
                                               gp(G.sperm), grad_x, grad_vel, n_nodes, field_input_grad);
    else
        kb_inputs<3><<<grid, dim3(256), 0, st>>>(*params, x, vel, charges, wp(W.nodeinfo), out, grad_out, wp(W.DA), wp(W.DN),
                                               wp(W.DF), gp(G.rowptr), gp(G.send_s), gp(G.recv_s), gp(G.srowptr),
                                               gp(G.sperm), grad_x, grad_vel, n_nodes, field_input_grad);
    if (grad_edge_attr && n_edges > 0) {
        const int D = num_dims, col0 = 7 * D + D * (D - 1) / 2;
        kb_edge_attr_grad<<<dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, st>>>(wp(W.DA), gp(G.perm), col0, grad_edge_attr,
                                                                                       n_edges);
    }
    HIP_OK(hipGetLastError());
    return AETHER_OK;
}

int aether_backward_field(const AetherParams* params, const AetherParams* grads, int num_dims, int64_t n_nodes,
                          int64_t n_edges, const float* x, const float* vel, const float* charges,
                          const void* graph, const AetherGraphInfo* info, void* workspace, size_t workspace_bytes,
                          const float* grad_out, float* grad_field, void* stream) {
    if (!params || !grads || !x || !vel || !charges || !graph || !info || !workspace || !grad_out || !grad_field)
        return fail(AETHER_EINVAL, "backward_field: null pointer");
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "backward_field: num_dims must be 2 or 3");
    if (n_nodes <= 0 || n_edges < 0 || info->n_nodes != n_nodes || info->n_edges != n_edges)
        return fail(AETHER_EINVAL, "backward_field: bad sizes");
    if (workspace_bytes < aether_workspace_bytes(n_nodes, n_edges, num_dims, 1))
        return fail(AETHER_ESPACE, "backward_field: workspace too small (forward must run with KEEP_INTERMEDIATES)");
    hipStream_t st = (hipStream_t)stream;
    if (take_async_error()) return AETHER_EHIP;
    if (num_dims == 2) {
        if (fused_backward_applies<2>(*info, n_nodes, n_edges))
            return backward_fused_impl<2>(*params, *grads, n_nodes, n_edges, *info, x, vel, charges, (const char*)graph,
                                          (char*)workspace, grad_out, st, grad_field);
        return backward_impl<2>(*params, *grads, n_nodes, n_edges, x, vel, charges, (const char*)graph,
                                (char*)workspace, grad_out, st, grad_field);
    }
    if (fused_backward_applies<3>(*info, n_nodes, n_edges))
        return backward_fused_impl<3>(*params, *grads, n_nodes, n_edges, *info, x, vel, charges, (const char*)graph,
                                      (char*)workspace, grad_out, st, grad_field);
    return backward_impl<3>(*params, *grads, n_nodes, n_edges, x, vel, charges, (const char*)graph,
                            (char*)workspace, grad_out, st, grad_field);
}

size_t aether_dynamic_field_backward_workspace_bytes(int num_dims, int64_t n_graphs) {
    if ((num_dims != 2 && num_dims != 3) || n_graphs <= 0) return 0;
    const size_t total = num_dims == 2 ? DynOff<2>::total : DynOff<3>::total;
    return (size_t)n_graphs * total * sizeof(float) + 256;
}

int aether_dynamic_field_backward(const AetherDynFieldParams* p, const AetherDynFieldParams* grads, int num_dims,
                                  int64_t n_graphs, int nodes_per_graph, const float* x, const float* vel,
                                  const float* charges, const float* grad_field, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return aether_dynamic_field_backward_inputs(p, grads, num_dims, n_graphs, nodes_per_graph, x, vel, charges, grad_field,
                                                workspace, workspace_bytes, nullptr, stream);
}

int aether_dynamic_field_backward_inputs(const AetherDynFieldParams* p, const AetherDynFieldParams* grads, int num_dims,
                                         int64_t n_graphs, int nodes_per_graph, const float* x, const float* vel,
                                         const float* charges, const float* grad_field, void* workspace,
                                         size_t workspace_bytes, float* grad_field_inputs, void* stream) {
    if (!p || !grads || !x || !vel || !charges || !grad_field || !workspace)
        return fail(AETHER_EINVAL, "dynamic_field_backward: null pointer");
    {
        const float* const* gp = reinterpret_cast<const float* const*>(grads);
        for (size_t k = 0; k < sizeof(AetherDynFieldParams) / sizeof(const float*); ++k)
            if (!gp[k]) return fail(AETHER_EINVAL, "dynamic_field_backward: null gradient pointer");
    }
    if (num_dims != 2 && num_dims != 3) return fail(AETHER_EINVAL, "dynamic_field_backward: num_dims must be 2 or 3");
    if (n_graphs <= 0 || nodes_per_graph <= 0 || nodes_per_graph > DYNFIELD_MAX_NODES)
        return fail(AETHER_EINVAL, "dynamic_field_backward: 1..2048 nodes per graph");
    if (n_graphs >= ((int64_t)1 << 31)) return fail(AETHER_EINVAL, "dynamic_field_backward: too many graphs");
    if (workspace_bytes < aether_dynamic_field_backward_workspace_bytes(num_dims, n_graphs))
        return fail(AETHER_ESPACE, "dynamic_field_backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
    if (num_dims == 2) {
        kb_dynfield<2><<<dim3((unsigned)n_graphs), dim3(256), 0, st>>>(*p, x, vel, charges, grad_field, partial, nodes_per_graph,
                                                                       grad_field_inputs);
        k_dynfield_reduce<2><<<dim3((DynOff<2>::total + 31) / 32), dim3(256), 0, st>>>(partial, n_graphs, *grads);
    } else {
        kb_dynfield<3><<<dim3((unsigned)n_graphs), dim3(256), 0, st>>>(*p, x, vel, charges, grad_field, partial, nodes_per_graph,
                                                                       grad_field_inputs);
        k_dynfield_reduce<3><<<dim3((DynOff<3>::total + 31) / 32), dim3(256), 0, st>>>(partial, n_graphs, *grads);
    }
    HIP_OK(hipGetLastError());
    return AETHER_OK;
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
    if (!strcmp(name, "stamps"))
        return copy2d((const float*)(ws + W.stamps), 4096, FUSED_STAMPS, 0, FUSED_STAMPS);
    if (name[0] == 'x' && name[1] >= '0' && name[1] <= '4' && !name[2])
        return copy2d((const float*)(ws + W.x[name[1] - '0']), n_nodes, H, 0, H);
    if (name[0] == 'e' && name[1] >= '1' && name[1] <= '4' && !name[2])
        return copy2d((const float*)(ws + W.e[name[1] - '1']), n_edges, H, 0, H);
    if (!strcmp(name, "efeat"))      // layer-1 edge features [E][FPAD] (kept with KEEP_INTERMEDIATES), sorted order
        return copy2d((const float*)(ws + W.feat), n_edges, FPAD, 0, FPAD);
    return fail(AETHER_EINVAL, "debug_fetch: unknown name");
}

#include "host_wide.inc"

}  // extern "C"
