// dyn_step.h -- the glue of one prediction step of the variable-N model:
//   AetherDynamicVars.predict_future's loop body, nn/dynamicvars/aether_dynamicvars.py:245-273 -> :64-79 (field),
//   :672-699 (prior step with its own kNN graph and the per-pair LSTM slots), :133-145 (sampling), :775-870 (decoder).
// The three stages (aether_dyn_field, aether_dyn_prior_step, aether_dyn_decoder_step) and aether_knn_edges exist; between
// them the Python module ran ~60 torch launches per step (mask -> index list, gathers, scatters, a stable sort for the
// receiver CSR, slot arithmetic).  Here that is five small kernels, so aether_dyn_step is ONE C call per step.
// All of it is index / copy work on a few hundred objects: single-workgroup kernels where a scan is needed, no atomics on
// floats, results independent of scheduling.
#pragma once
#include "common.h"
#include "knn.h"

namespace {

constexpr int DYN_MAX_OBJECTS = KNN_MAX_OBJECTS;

// Present objects of the scene: idx[c] = row of the c-th non-zero mask entry (ascending), cidx[row] = c or -1;
// cur_in[c] = state[idx[c]] (4 floats), cur_h[c] = hidden[idx[c]] (h floats), rowptr_dec[c] = c * deg (c <= n_present).
// status[0] = the number of present objects found.  When it differs from n_present (host-supplied) the async error word is
// set to 2 and the rows past the found count repeat row 0 (memory-safe; aether_dyn_step's finish kernel then writes NaN).
__global__ void __launch_bounds__(256)
k_dyn_present(const float* __restrict__ state, const float* __restrict__ mask, const float* __restrict__ hidden, int n_max,
              int n_present, int h, int deg, int64_t* __restrict__ idx, int* __restrict__ cidx, float* __restrict__ cur_in,
              float* __restrict__ cur_h, int64_t* __restrict__ rowptr_dec, int* __restrict__ status, int* __restrict__ errword,
              int64_t* __restrict__ knn_sums /* [2]: zeroed for the kNN builder */, int* __restrict__ dec_counts /* [64]: zeroed
              for the decoder's per-type edge lists (two memset nodes less per step) */) {
    extern __shared__ int dyn_lds[];
    if (threadIdx.x < 2) knn_sums[threadIdx.x] = 0;
    if (threadIdx.x < 64) dec_counts[threadIdx.x] = 0;
    int* flag = dyn_lds;                 // [n_max] -> exclusive count of present objects
    int* rows = flag + n_max;            // [n_present] rows of the present objects
    int* part = rows + n_present;        // [257]
    const int tid = threadIdx.x;
    for (int j = tid; j < n_max; j += 256) flag[j] = mask[j] != 0.0f ? 1 : 0;
    for (int c = tid; c < n_present; c += 256) rows[c] = 0;
    __syncthreads();
    const int found = block_exclusive_scan(flag, n_max, part);
    for (int j = tid; j < n_max; j += 256) {
        const bool present = mask[j] != 0.0f;
        const int c = flag[j];
        cidx[j] = present && c < n_present ? c : -1;
        if (present && c < n_present) rows[c] = j;
    }
    __syncthreads();
    if (tid == 0) {
        status[0] = found;
        if (found != n_present && errword) *errword = 2;
    }
    for (int c = tid; c < n_present; c += 256) idx[c] = rows[c];
    for (int c = tid; c <= n_present; c += 256) rowptr_dec[c] = (int64_t)c * deg;
    for (int t = tid; t < n_present * 4; t += 256) cur_in[t] = state[(size_t)rows[t >> 2] * 4 + (t & 3)];
    const int h4 = h >> 2;
    for (int t = tid; t < n_present * h4; t += 256) {
        const int c = t / h4, o = t - c * h4;
        st4(cur_h + (size_t)c * h + 4 * o, ld4(hidden + (size_t)rows[c] * h + 4 * o));
    }
}

// (order, rowptr) = edge ids grouped by receiver, stable (torch.argsort(recv, stable=True) + the CSR offsets): counts by
// integer LDS atomics, a scan, then chunks of 256 edges in id order -- an edge's place is its receiver's row start + the
// edges of that receiver in earlier chunks + those before it in its own chunk.  One workgroup; n <= 8192 objects.
// When the mask disagreed with n_present (status[0] != n_nodes) the kNN graph it was built from has other sizes than the
// buffers' users assume: its first n_edges entries are clamped into range here, so that the stages -- whose results are
// discarded (k_dyn_finish writes NaN) -- read inside their arrays.
__global__ void __launch_bounds__(256)
k_dyn_csr(int64_t* __restrict__ send, int64_t* __restrict__ recv, int64_t n_edges, int n_nodes, const int* __restrict__ status,
          int64_t* __restrict__ order, int64_t* __restrict__ rowptr) {
    extern __shared__ int dyn_lds[];
    int* start = dyn_lds;                // [n_nodes] counts -> row starts -> running fill position
    int* part = start + n_nodes;         // [257]
    int* chunk = part + 257;             // [256] receivers of the current chunk
    const int tid = threadIdx.x;
    for (int j = tid; j < n_nodes; j += 256) start[j] = 0;
    if (status[0] != n_nodes) {
        for (int64_t e = tid; e < n_edges; e += 256) {
            const int64_t a = send[e], b = recv[e];
            send[e] = a < 0 ? 0 : (a >= n_nodes ? n_nodes - 1 : a);
            recv[e] = b < 0 ? 0 : (b >= n_nodes ? n_nodes - 1 : b);
        }
    }
    __syncthreads();
    for (int64_t e = tid; e < n_edges; e += 256) {
        const int64_t r = recv[e];
        if (r >= 0 && r < n_nodes) atomicAdd(start + (int)r, 1);
    }
    __syncthreads();
    const int total = block_exclusive_scan(start, n_nodes, part);
    for (int j = tid; j < n_nodes; j += 256) rowptr[j] = start[j];
    if (tid == 0) rowptr[n_nodes] = total;
    __syncthreads();
    for (int64_t base = 0; base < n_edges; base += 256) {
        const int64_t e = base + tid;
        int r = -1;
        if (e < n_edges) { const int64_t rr = recv[e]; r = rr >= 0 && rr < n_nodes ? (int)rr : -1; }
        chunk[tid] = r;
        __syncthreads();
        int before = 0, after = 0;
        if (r >= 0) {
            for (int t = 0; t < 256; ++t) {
                const bool same = chunk[t] == r;
                before += (same && t < tid) ? 1 : 0;
                after += (same && t > tid) ? 1 : 0;
            }
            order[start[r] + before] = e;
        }
        __syncthreads();
        if (r >= 0 && after == 0) start[r] += before + 1;      // the last edge of r in this chunk moves r's fill position
        __syncthreads();
    }
}

// LSTM state rows of the caller's edges: one slot per ordered pair of objects (aether_dynamicvars.py:680-686),
// and -- riding along -- ext_full[row] = [state[row] (4) | field of the row (2), zero when absent]: the un-compacted array
// the decoder's edge features read with compacted indices (the reference's quirk, :823);
// slot = gs (n_max - 1) + gr - (gr >= gs) with gs, gr the un-compacted rows of the edge's ends; gathers h0, c0.
__global__ void __launch_bounds__(256)
k_dyn_slots_gather(const int64_t* __restrict__ gsend, const int64_t* __restrict__ grecv, const int64_t* __restrict__ node_inds,
                   int n_present, int n_max, int64_t n_edges, int R, const float* __restrict__ prior_h,
                   const float* __restrict__ prior_c, int64_t* __restrict__ slot, float* __restrict__ h0,
                   float* __restrict__ c0, const float* __restrict__ state = nullptr, const float* __restrict__ field_c = nullptr,
                   const int* __restrict__ cidx = nullptr, float* __restrict__ ext_full = nullptr) {
    const int r4 = R >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ext_full != nullptr && t < (int64_t)n_max * 6) {          // k_dyn_ext's work rides along (one launch less)
        const int row = (int)(t / 6), col = (int)(t - (int64_t)row * 6);
        float v;
        if (col < 4) v = state[(size_t)row * 4 + col];
        else { const int c = cidx[row]; v = c >= 0 ? field_c[(size_t)c * 2 + col - 4] : 0.0f; }
        ext_full[t] = v;
    }
    if (t >= n_edges * r4) return;
    const int64_t e = t / r4;
    const int o = (int)(t - e * r4);
    int64_t a = gsend[e], b = grecv[e];
    a = a < 0 ? 0 : (a >= n_present ? n_present - 1 : a);          // (out-of-range ids are refused on the host side of the
    b = b < 0 ? 0 : (b >= n_present ? n_present - 1 : b);          //  module; clamped here to stay inside the arrays)
    const int64_t gs = node_inds[a], gr = node_inds[b];
    int64_t s = gs * (n_max - 1) + gr - (gr >= gs ? 1 : 0);
    const int64_t n_slots = (int64_t)n_max * (n_max - 1);
    s = s < 0 ? 0 : (s >= n_slots ? n_slots - 1 : s);
    if (o == 0) slot[e] = s;
    st4(h0 + e * R + 4 * o, ld4(prior_h + s * R + 4 * o));
    st4(c0 + e * R + 4 * o, ld4(prior_c + s * R + 4 * o));
}

__global__ void __launch_bounds__(256)
k_dyn_slots_scatter(const int64_t* __restrict__ slot, const float* __restrict__ h1, const float* __restrict__ c1,
                    int64_t n_edges, int R, float* __restrict__ prior_h, float* __restrict__ prior_c) {
    const int r4 = R >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_edges * r4) return;
    const int64_t e = t / r4;
    const int o = (int)(t - e * r4);
    const int64_t s = slot[e];
    st4(prior_h + s * R + 4 * o, ld4(h1 + e * R + 4 * o));
    st4(prior_c + s * R + 4 * o, ld4(c1 + e * R + 4 * o));
}

// k_dyn_slots_scatter and the hard Gumbel sample of the edge types (k_s2s_gumbel_hard, nn/utils/model_utils.py:58-118) in one
// launch: both only need the prior step's outputs.
__global__ void __launch_bounds__(256)
k_dyn_scatter_sample(const int64_t* __restrict__ slot, const float* __restrict__ h1, const float* __restrict__ c1,
                     int64_t n_edges, int R, float* __restrict__ prior_h, float* __restrict__ prior_c,
                     const float* __restrict__ logits, const float* __restrict__ uniform, float tau, int K,
                     float* __restrict__ edges, const int* __restrict__ status = nullptr, int n_present = 0) {
    const int r4 = R >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_edges * r4) {
        const int64_t e = t / r4;
        const int o = (int)(t - e * r4);
        const int64_t s = slot[e];
        f32x4 hv = ld4(h1 + e * R + 4 * o), cv = ld4(c1 + e * R + 4 * o);
        if (status != nullptr && status[0] != n_present) {       // the mask disagreed with n_present: the stages ran on clamped
            const float poison = __int_as_float(0x7fc00000);     // index lists -- their rows are garbage: NaN, as the outputs
            hv = f32x4{poison, poison, poison, poison}; cv = hv;
        }
        st4(prior_h + s * R + 4 * o, hv);
        st4(prior_c + s * R + 4 * o, cv);
    }
    if (t < n_edges) {                         // the arithmetic of k_s2s_gumbel_hard, statement for statement
        const int64_t e = t;
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
        for (int k = 0; k < K; ++k) edges[e * K + k] = ((k == best ? 1.0f : 0.0f) - y[k]) + y[k];
    }
}

// prediction [n_max][4] = the decoder's rows at the present objects, zero elsewhere (:866-868); hidden[idx[c]] = new_h[c].
// NaN everywhere when the mask disagreed with n_present (status[0] != n_present).
__global__ void __launch_bounds__(256)
k_dyn_finish(const float* __restrict__ out_c, const float* __restrict__ new_h, const int* __restrict__ cidx,
             const int64_t* __restrict__ idx, const int* __restrict__ status, int n_max, int n_present, int h,
             float* __restrict__ prediction, float* __restrict__ hidden) {
    const bool bad = status[0] != n_present;
    const float poison = __int_as_float(0x7fc00000);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int h4 = h >> 2;
    if (t < (int64_t)n_max * 4) {
        const int row = (int)(t >> 2), c = cidx[row];
        prediction[t] = bad ? poison : (c >= 0 ? out_c[(size_t)c * 4 + (t & 3)] : 0.0f);
    }
    if (t < (int64_t)n_present * h4) {
        const int c = (int)(t / h4), o = (int)(t - (int64_t)c * h4);
        f32x4 v = ld4(new_h + (size_t)c * h + 4 * o);
        if (bad) v = f32x4{poison, poison, poison, poison};
        st4(hidden + (size_t)idx[c] * h + 4 * o, v);
    }
}

// The state a prediction step sees (aether_dynamicvars.py:264): observed objects get their ground truth, the others the
// model's own last prediction.  state = observed * truth + (1 - observed) * last, as the reference computes it.
__global__ void __launch_bounds__(256)
k_dyn_mix(const float* __restrict__ truth, const float* __restrict__ last, const float* __restrict__ observed, int n_max,
          float* __restrict__ state) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_max * 4) return;
    const float o = observed[t >> 2];
    state[t] = o * truth[t] + (1.0f - o) * last[t];
}


// =================================================================== B scenes per call (round 4; BASELINE config 4 is batch = 64)
// aether_dyn_step_batched: the same step for B scenes at once -- the present objects of all scenes are numbered
// consecutively (scene-major), every stage (field, kNN + prior step, sampling, decoder) runs once over all of them, as
// predict_future_batched of the module did with ~60 torch launches and two host round trips per step in between.
// The per-scene sizes travel BY VALUE in the kernel arguments (the host knows them: the data set's node_inds), so what
// the host validated is what the kernels index with; offsets are prefix sums every workgroup forms itself in LDS.
// These kernels take a by-value struct and may sit in captured graphs: they consume no hidden kernel arguments
// (no gridDim / blockDim reads; tools/isa_check.py rule R4).
constexpr int DYN_MAX_SCENES = 256;
struct DynScenes {
    int n_scenes, n_max, knn_k;                  // scenes, object rows per scene, k of the encoder's kNN graph
    short n[DYN_MAX_SCENES];                     // present objects per scene (0, or >= 2)
    unsigned char deg[DYN_MAX_SCENES];           // in-degree of the caller's graph per scene (columns of edge2node)
};

// Exclusive prefix sums over the scenes, in LDS: nodes, kNN / caller edges (n * min(k, n - 1)), edge2node entries (n * deg).
// off: [3][DYN_MAX_SCENES + 1] ints; 256 threads.
__device__ inline void dyn_scene_offsets(const DynScenes& S, int* off, int* part) {
    const int tid = threadIdx.x;
    int* nb = off; int* eb = off + (DYN_MAX_SCENES + 1); int* ob = off + 2 * (DYN_MAX_SCENES + 1);
    if (tid < DYN_MAX_SCENES) {
        const int n = tid < S.n_scenes ? S.n[tid] : 0;
        const int k = n >= 2 ? min(S.knn_k, n - 1) : 0;
        nb[tid] = n; eb[tid] = n * k; ob[tid] = n * (tid < S.n_scenes ? S.deg[tid] : 0);
    }
    __syncthreads();
    const int tn = block_exclusive_scan(nb, DYN_MAX_SCENES, part, 256);
    const int te = block_exclusive_scan(eb, DYN_MAX_SCENES, part, 256);
    const int to = block_exclusive_scan(ob, DYN_MAX_SCENES, part, 256);
    if (tid == 0) { nb[DYN_MAX_SCENES] = tn; eb[DYN_MAX_SCENES] = te; ob[DYN_MAX_SCENES] = to; }
    __syncthreads();
}
// scene of item t in a prefix array (first s with base[s + 1] > t); base[n_scenes .. DYN_MAX_SCENES] = total
__device__ inline int dyn_scene_of(const int* base, int n_scenes, int t) {
    int lo = 0, hi = n_scenes - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (base[mid + 1] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// One workgroup per scene: k_dyn_present for scene s.  idx[c] = flat row s * n_max + j of the c-th present object (c in the
// consecutive numbering of all scenes), cidx[flat row] = c or -1, cur_in / cur_h gathered, rowptr_dec / div_node of the
// decoder's aggregation (aether_dynamicvars.py:853-860: rows of edge2node, divisor n - 1).  status[0] += found,
// status[1] |= (found != n[s]) -- both zeroed by the host before the launch; a mismatch also sets the async error word.
__global__ void __launch_bounds__(256)
k_dynb_present(const DynScenes S, const float* __restrict__ state, const float* __restrict__ mask,
               const float* __restrict__ hidden, int h, int64_t* __restrict__ idx, int* __restrict__ cidx,
               float* __restrict__ cur_in, float* __restrict__ cur_h, int64_t* __restrict__ rowptr_dec,
               float* __restrict__ div_node, int* __restrict__ status, int* __restrict__ errword,
               int64_t* __restrict__ knn_sums /* [2 * n_scenes], zeroed here */, int* __restrict__ dec_counts /* [64] */) {
    extern __shared__ int dyn_lds[];
    int* off = dyn_lds;                                    // [3][257]
    int* part = off + 3 * (DYN_MAX_SCENES + 1);            // [257]
    int* flag = part + 257;                                // [n_max]
    int* rows = flag + S.n_max;                            // [n_max]
    const int tid = threadIdx.x, s = blockIdx.x, n_max = S.n_max;
    dyn_scene_offsets(S, off, part);
    const int n_s = S.n[s], deg = S.deg[s];
    const int nbase = off[s], obase = off[2 * (DYN_MAX_SCENES + 1) + s];
    if (tid < 2) knn_sums[2 * s + tid] = 0;
    if (s == 0 && tid < 64) dec_counts[tid] = 0;
    const float* m = mask + (size_t)s * n_max;
    for (int j = tid; j < n_max; j += 256) { flag[j] = m[j] != 0.0f ? 1 : 0; rows[j] = 0; }
    __syncthreads();
    const int found = block_exclusive_scan(flag, n_max, part, 256);
    for (int j = tid; j < n_max; j += 256) {
        const bool present = m[j] != 0.0f;
        const int c = flag[j];
        cidx[(size_t)s * n_max + j] = present && c < n_s ? nbase + c : -1;
        if (present && c < n_s) rows[c] = j;
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(status, found);
        if (found != n_s) { atomicOr(status + 1, 1); if (errword) *errword = 2; }
        if (s == S.n_scenes - 1) rowptr_dec[off[DYN_MAX_SCENES]] = off[2 * (DYN_MAX_SCENES + 1) + DYN_MAX_SCENES];
    }
    const float dv = (float)(n_s > 1 ? n_s - 1 : 1);
    for (int c = tid; c < n_s; c += 256) {
        idx[nbase + c] = (int64_t)s * n_max + rows[c];
        rowptr_dec[nbase + c] = (int64_t)obase + (int64_t)c * deg;
        div_node[nbase + c] = dv;
    }
    for (int t = tid; t < n_s * 4; t += 256)
        cur_in[(size_t)nbase * 4 + t] = state[((size_t)s * n_max + rows[t >> 2]) * 4 + (t & 3)];
    const int h4 = h >> 2;
    for (int t = tid; t < n_s * h4; t += 256) {
        const int c = t / h4, o = t - c * h4;
        st4(cur_h + (size_t)(nbase + c) * h + 4 * o, ld4(hidden + ((size_t)s * n_max + rows[c]) * h + 4 * o));
    }
}

// One workgroup per scene: k_dyn_csr on the scene's slice of the kNN graph (its edges are contiguous, scene by scene:
// knn.h); order holds global edge ids, rowptr global offsets.  A mismatch (status[1]) clamps the ids into the scene's range.
__global__ void __launch_bounds__(256)
k_dynb_csr(const DynScenes S, int64_t* __restrict__ send, int64_t* __restrict__ recv, const int* __restrict__ status,
           int64_t* __restrict__ order, int64_t* __restrict__ rowptr) {
    extern __shared__ int dyn_lds[];
    int* off = dyn_lds;                                    // [3][257]
    int* part = off + 3 * (DYN_MAX_SCENES + 1);            // [257]
    int* chunk = part + 257;                               // [256]
    int* start = chunk + 256;                              // [n of the scene]
    const int tid = threadIdx.x, s = blockIdx.x;
    dyn_scene_offsets(S, off, part);
    const int n = S.n[s], nbase = off[s];
    const int ebase = off[(DYN_MAX_SCENES + 1) + s], E = off[(DYN_MAX_SCENES + 1) + s + 1] - ebase;
    if (s == S.n_scenes - 1 && tid == 0) rowptr[off[DYN_MAX_SCENES]] = off[(DYN_MAX_SCENES + 1) + DYN_MAX_SCENES];
    if (n == 0) return;
    for (int j = tid; j < n; j += 256) start[j] = 0;
    if (status[1] != 0) {
        for (int e = tid; e < E; e += 256) {
            const int64_t a = send[ebase + e] - nbase, b = recv[ebase + e] - nbase;
            send[ebase + e] = nbase + (a < 0 ? 0 : (a >= n ? n - 1 : a));
            recv[ebase + e] = nbase + (b < 0 ? 0 : (b >= n ? n - 1 : b));
        }
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        const int64_t r = recv[ebase + e] - nbase;
        if (r >= 0 && r < n) atomicAdd(start + (int)r, 1);
    }
    __syncthreads();
    block_exclusive_scan(start, n, part, 256);
    for (int j = tid; j < n; j += 256) rowptr[nbase + j] = (int64_t)ebase + start[j];
    __syncthreads();
    for (int base = 0; base < E; base += 256) {
        const int e = base + tid;
        int r = -1;
        if (e < E) { const int64_t rr = recv[ebase + e] - nbase; r = rr >= 0 && rr < n ? (int)rr : -1; }
        chunk[tid] = r;
        __syncthreads();
        int before = 0, after = 0;
        if (r >= 0) {
            for (int t = 0; t < 256; ++t) {
                const bool same = chunk[t] == r;
                before += (same && t < tid) ? 1 : 0;
                after += (same && t > tid) ? 1 : 0;
            }
            order[ebase + start[r] + before] = (int64_t)ebase + e;
        }
        __syncthreads();
        if (r >= 0 && after == 0) start[r] += before + 1;
        __syncthreads();
    }
}

// The caller's graphs, scene-local numbering -> the arrays the batched stages take (decoder.forward_batched's torch glue):
//   send / recv   = local id + the scene's first consecutive number            (compacted, all scenes)
//   ssend / srecv = local id + s * n_max                                       (the reference's quirk, :823: un-compacted
//                                                                               rows indexed with compacted ids)
//   slot          = s n_max (n_max - 1) + gs (n_max - 1) + gr - (gr >= gs), gs / gr the object rows of the ends (:680-686)
//   h0 / c0       = the LSTM state rows of the slots;   ext_full[flat row] = [state | field or 0]
// and (second part of the grid's work) order[j] = edge2node[j] + the scene's first edge.
__global__ void __launch_bounds__(256)
k_dynb_index(const DynScenes S, const int64_t* __restrict__ gsend, const int64_t* __restrict__ grecv,
             const int64_t* __restrict__ node_inds /* scene-local rows, concatenated; null: from idx */,
             const int64_t* __restrict__ idx, const int64_t* __restrict__ e2n, int R, const float* __restrict__ prior_h,
             const float* __restrict__ prior_c, int64_t* __restrict__ send, int64_t* __restrict__ recv,
             int64_t* __restrict__ ssend, int64_t* __restrict__ srecv, int64_t* __restrict__ slot, float* __restrict__ h0,
             float* __restrict__ c0, int64_t* __restrict__ order, const float* __restrict__ state,
             const float* __restrict__ field_c, const int* __restrict__ cidx, float* __restrict__ ext_full) {
    __shared__ int off[3 * (DYN_MAX_SCENES + 1)];
    __shared__ int part[257];
    dyn_scene_offsets(S, off, part);
    const int* nb = off; const int* eb = off + (DYN_MAX_SCENES + 1); const int* ob = off + 2 * (DYN_MAX_SCENES + 1);
    const int n_total = nb[DYN_MAX_SCENES], E = eb[DYN_MAX_SCENES], O = ob[DYN_MAX_SCENES];
    const int r4 = R >> 2, n_max = S.n_max;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < (int64_t)S.n_scenes * n_max * 6) {
        const int64_t row = t / 6;
        const int col = (int)(t - row * 6);
        float v;
        if (col < 4) v = state[row * 4 + col];
        else { const int c = cidx[row]; v = c >= 0 ? field_c[(size_t)c * 2 + col - 4] : 0.0f; }
        ext_full[t] = v;
    }
    if (t < O) {
        const int s = dyn_scene_of(ob, S.n_scenes, (int)t);
        const int Es = eb[s + 1] - eb[s];
        int64_t v = e2n[t];
        v = v < 0 ? 0 : (v >= Es ? Es - 1 : v);                  // (refused on the host side of the module; clamped here)
        order[t] = v + eb[s];
    }
    if (t >= (int64_t)E * r4) return;
    const int e = (int)(t / r4), o = (int)(t - (int64_t)e * r4);
    const int s = dyn_scene_of(eb, S.n_scenes, e);
    const int n = S.n[s], nbase = nb[s];
    int64_t a = gsend[e], b = grecv[e];
    a = a < 0 ? 0 : (a >= n ? n - 1 : a);
    b = b < 0 ? 0 : (b >= n ? n - 1 : b);
    int64_t gs, gr;
    if (node_inds != nullptr) { gs = node_inds[nbase + a]; gr = node_inds[nbase + b]; }
    else { gs = idx[nbase + a] - (int64_t)s * n_max; gr = idx[nbase + b] - (int64_t)s * n_max; }
    gs = gs < 0 ? 0 : (gs >= n_max ? n_max - 1 : gs);
    gr = gr < 0 ? 0 : (gr >= n_max ? n_max - 1 : gr);
    const int64_t per = (int64_t)n_max * (n_max - 1);
    int64_t sl = gs * (n_max - 1) + gr - (gr >= gs ? 1 : 0);
    sl = (sl < 0 ? 0 : (sl >= per ? per - 1 : sl)) + (int64_t)s * per;
    if (o == 0) {
        slot[e] = sl;
        send[e] = nbase + a; recv[e] = nbase + b;
        ssend[e] = (int64_t)s * n_max + a; srecv[e] = (int64_t)s * n_max + b;
    }
    st4(h0 + (size_t)e * R + 4 * o, ld4(prior_h + sl * R + 4 * o));
    st4(c0 + (size_t)e * R + 4 * o, ld4(prior_c + sl * R + 4 * o));
    (void)n_total;
}

// prediction [B * n_max][4] = the decoder's rows at the present objects, zero elsewhere (:866-868); hidden[idx[c]] =
// new_h[c].  NaN everywhere when a scene's mask disagreed with its n (status[1]).
__global__ void __launch_bounds__(256)
k_dynb_finish(const float* __restrict__ out_c, const float* __restrict__ new_h, const int* __restrict__ cidx,
              const int64_t* __restrict__ idx, const int* __restrict__ status, int64_t n_rows, int n_total, int h,
              float* __restrict__ prediction, float* __restrict__ hidden) {
    const bool bad = status[1] != 0;
    const float poison = __int_as_float(0x7fc00000);
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int h4 = h >> 2;
    if (t < n_rows * 4) {
        const int64_t row = t >> 2;
        const int c = cidx[row];
        prediction[t] = bad ? poison : (c >= 0 ? out_c[(size_t)c * 4 + (t & 3)] : 0.0f);
    }
    if (t < (int64_t)n_total * h4) {
        const int c = (int)(t / h4), o = (int)(t - (int64_t)c * h4);
        f32x4 v = ld4(new_h + (size_t)c * h + 4 * o);
        if (bad) v = f32x4{poison, poison, poison, poison};
        st4(hidden + (size_t)idx[c] * h + 4 * o, v);
    }
}

}  // namespace
