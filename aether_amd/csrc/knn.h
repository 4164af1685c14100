// k-nearest-neighbour edge builder for variable-N scenes (first half):
//   Encoder.knn_edges        nn/dynamicvars/aether_dynamicvars.py:559-586 (repeated in the other *_dynamicvars.py)
//   get_knn_graph_info       experiments/ind/single_ind_data.py:186-217
// Per scene (time step): every present object (mask != 0) lists its k nearest present objects by 2-D
// distance, nearest first; edges come out scene by scene, object by object, in the compacted numbering
// (present objects of all scenes counted consecutively).  The reference builds an [S, N, N] distance
// matrix, a full topk and boolean-mask filters on the host side of every forward; here it is three launches
// of integer / compare work: select (one workgroup per scene, positions and masks in LDS, a sorted
// top-16 list per thread in registers), a scan over scenes, and the write-out.  Traffic: the positions once,
// 16 bytes per edge out.
#pragma once
#include "common.h"

namespace {

constexpr int KNN_MAX_K = 16;
constexpr int KNN_MAX_OBJECTS = 8192;          // positions + masks + compact index of a scene in LDS

// exclusive prefix sums of val[0..n) (LDS) over the workgroup, in place; returns the total.
// part: LDS scratch of blockDim.x ints.  Per-thread chunk sums, an inclusive scan across each wave's lanes (shuffles), the
// (at most 16) wave totals added by every thread: no serial pass of one thread over blockDim.x LDS words (that pass alone
// took 5-7 us per call, four calls per variable-N prediction step).
__device__ inline int block_exclusive_scan(int* val, int n, int* part, const int nt_fixed = 0 /* the workgroup's thread
        count when the caller knows it: blockDim.x is an implicit (hidden) kernel argument, which by-value-struct kernels of
        captured steps must not consume (tools/isa_check.py rule R4) */) {
    const int tid = threadIdx.x, nt = nt_fixed > 0 ? nt_fixed : (int)blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, n_waves = (nt + 63) >> 6;
    const int chunk = (n + nt - 1) / nt;
    const int beg = min(tid * chunk, n), end = min(beg + chunk, n);
    int s = 0;
    for (int i = beg; i < end; ++i) s += val[i];
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int wave_off = 0, total = 0;
    for (int w = 0; w < n_waves; ++w) {
        const int v = part[w];
        wave_off += w < wave ? v : 0;
        total += v;
    }
    int run = wave_off + incl - s;
    for (int i = beg; i < end; ++i) { const int v = val[i]; val[i] = run; run += v; }
    __syncthreads();
    return total;
}

// nbr[s][i][0..k): neighbours of object i, nearest first (local object ids); cnt[s][i]: how many are real.
// scene_nodes[s], scene_edges[s] (zeroed by the host): present objects and edges of the scene.
__global__ void __launch_bounds__(256)
k_knn_select(const float* __restrict__ x, int x_stride, const float* __restrict__ masks, int N, int k,
             int* __restrict__ nbr, int* __restrict__ cnt, int64_t* __restrict__ scene_nodes,
             int64_t* __restrict__ scene_edges) {
    extern __shared__ float knn_lds[];
    float* px = knn_lds;                         // [N]
    float* py = px + N;                          // [N]
    float* pm = py + N;                          // [N]
    int* red = reinterpret_cast<int*>(pm + N);   // [257]
    const int64_t s = blockIdx.x;
    const int tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) {
        const float* row = x + ((size_t)s * N + j) * x_stride;
        px[j] = row[0]; py[j] = row[1];
        pm[j] = masks[(size_t)s * N + j];
    }
    __syncthreads();
    int my_nodes = 0, my_edges = 0;
    // gridDim.y workgroups share a scene: each takes the objects i = blockIdx.y * 256 + tid (+ 256 gridDim.y ...)
    for (int i = (int)blockIdx.y * 256 + tid; i < N; i += 256 * (int)gridDim.y) {
        int c = 0;
        if (pm[i] != 0.0f) {
            ++my_nodes;
            float bd[KNN_MAX_K];
            int bi[KNN_MAX_K];
#pragma unroll
            for (int p = 0; p < KNN_MAX_K; ++p) { bd[p] = INFINITY; bi[p] = -1; }
            const float xi = px[i], yi = py[i];
            for (int j = 0; j < N; ++j) {
                if (j == i || pm[j] == 0.0f) continue;
                const float dx = __fsub_rn(xi, px[j]), dy = __fsub_rn(yi, py[j]);
                float cd = __fsqrt_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)));
                int ci = j;
                if (!(cd < bd[KNN_MAX_K - 1])) continue;
                bool shifting = false;                           // stable insertion: equal distances keep index order
#pragma unroll
                for (int p = 0; p < KNN_MAX_K; ++p) {
                    if (shifting || cd < bd[p]) {                // once placed, everything behind moves down one slot
                        const float td = bd[p]; const int ti = bi[p];
                        bd[p] = cd; bi[p] = ci; cd = td; ci = ti;
                        shifting = true;
                    }
                }
            }
            int* out = nbr + ((size_t)s * N + i) * k;
#pragma unroll
            for (int p = 0; p < KNN_MAX_K; ++p)
                if (p < k) { out[p] = bi[p]; c += (bi[p] >= 0) ? 1 : 0; }
        }
        cnt[(size_t)s * N + i] = c;
        my_edges += c;
    }
    // two block reductions (nodes, edges)
    red[tid] = my_nodes;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    const int nodes = red[0];
    __syncthreads();
    red[tid] = my_edges;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    if (tid == 0) {                                              // integer sums: the order of the atomics does not matter
        atomicAdd(reinterpret_cast<unsigned long long*>(scene_nodes + s), (unsigned long long)nodes);
        atomicAdd(reinterpret_cast<unsigned long long*>(scene_edges + s), (unsigned long long)red[0]);
    }
}

// node_off[s], edge_off[s] = exclusive prefix sums over scenes; totals = {edges, nodes}.  One workgroup.
__global__ void __launch_bounds__(1024)
k_knn_scan(const int64_t* __restrict__ scene_nodes, const int64_t* __restrict__ scene_edges, int64_t n_scenes,
           int64_t* __restrict__ node_off, int64_t* __restrict__ edge_off, int64_t* __restrict__ totals) {
    __shared__ int64_t pn[1025], pe[1025];
    const int tid = threadIdx.x;
    const int64_t chunk = (n_scenes + 1023) / 1024;
    const int64_t beg = min((int64_t)tid * chunk, n_scenes), end = min(beg + chunk, n_scenes);
    int64_t sn = 0, se = 0;
    for (int64_t i = beg; i < end; ++i) { sn += scene_nodes[i]; se += scene_edges[i]; }
    pn[tid] = sn; pe[tid] = se;
    __syncthreads();
    if (tid == 0) {
        int64_t rn = 0, re = 0;
        const int active = (int)((n_scenes + chunk - 1) / (chunk > 0 ? chunk : 1));     // threads that own scenes (one scene: 1, not 1,024 serial steps)
        for (int t = 0; t < (active < 1024 ? active : 1024); ++t) {
            const int64_t vn = pn[t], ve = pe[t];
            pn[t] = rn; pe[t] = re; rn += vn; re += ve;
        }
        totals[0] = re; totals[1] = rn;
    }
    __syncthreads();
    int64_t rn = pn[tid], re = pe[tid];
    for (int64_t i = beg; i < end; ++i) {
        node_off[i] = rn; edge_off[i] = re;
        rn += scene_nodes[i]; re += scene_edges[i];
    }
}

// send[e] = node_off[s] + compact(i), recv[e] = node_off[s] + compact(neighbour), e from edge_off[s] on,
// object by object, nearest first   (aether_dynamicvars.py:577-584)
__global__ void __launch_bounds__(256)
k_knn_write(const float* __restrict__ masks, const int* __restrict__ nbr, const int* __restrict__ cnt, int N, int k,
            const int64_t* __restrict__ node_off, const int64_t* __restrict__ edge_off, int64_t* __restrict__ send,
            int64_t* __restrict__ recv, const int64_t* __restrict__ scene_sums = nullptr /* one scene: {edges, nodes} */,
            int64_t* __restrict__ totals = nullptr) {
    extern __shared__ float knn_lds[];
    int* compact = reinterpret_cast<int*>(knn_lds);   // [N] -> exclusive count of present objects
    int* eoff = compact + N;                            // [N] -> exclusive count of edges
    int* part = eoff + N;                               // [257]
    const int64_t s = blockIdx.x;
    const int tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) {
        compact[j] = masks[(size_t)s * N + j] != 0.0f ? 1 : 0;
        eoff[j] = cnt[(size_t)s * N + j];
    }
    __syncthreads();
    block_exclusive_scan(compact, N, part);
    block_exclusive_scan(eoff, N, part);
    // one scene (node_off == null): offsets are zero and the totals are the scene's sums -- no scan launch
    const int64_t nb = node_off ? node_off[s] : 0, eb = edge_off ? edge_off[s] : 0;
    if (totals != nullptr && tid == 0) { totals[0] = scene_sums[0]; totals[1] = scene_sums[1]; }
    for (int idx = tid; idx < N * k; idx += 256) {
        const int i = idx / k, r = idx - i * k;
        if (r >= cnt[(size_t)s * N + i]) continue;
        const int j = nbr[((size_t)s * N + i) * k + r];
        send[eb + eoff[i] + r] = nb + compact[i];
        recv[eb + eoff[i] + r] = nb + compact[j];
    }
}

}  // namespace
