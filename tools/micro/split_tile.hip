// split_tile.hip -- prototype: the edge-tile MLP (Linear 64->64, SiLU, Linear 64->64, SiLU on a 16-edge tile held in
// registers) on the fp32 MFMA (v_mfma_f32_16x16x4_f32, what fused.h runs today) against the same contraction as SIX
// bf16 MFMA terms on operands split into three bf16 pieces each (x = hi + mid + lo, 24 significant bits; products
// hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; fp32 accumulate).  Prints the error of both against an fp64 host
// evaluation and the time per tile at 2 waves per SIMD on every CU.  Diagnostic / design study, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/split_tile.hip -o gpurun_out/split_tile && gpurun_out/split_tile
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int H = 64, LDW = H + 8;

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 silu4(f32x4 v) {
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float e = __builtin_amdgcn_exp2f(v[k] * -1.44269504088896340736f);
        o[k] = v[k] * __builtin_amdgcn_rcpf(1.0f + e);
    }
    return o;
}

// ---------------------------------------------------------------- fp32 MFMA version (as fused.h)
__global__ void __launch_bounds__(512)
k_f32(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b, const float* __restrict__ x,
      float* __restrict__ y, int iters) {
    __shared__ __attribute__((aligned(16))) float wa[H * LDW], wb[H * LDW], bias[2 * H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    for (int idx = tid; idx < H * H; idx += 512) {
        wa[(idx >> 6) * LDW + (idx & 63)] = w1[idx];
        wb[(idx >> 6) * LDW + (idx & 63)] = w2[idx];
    }
    if (tid < 2 * H) bias[tid] = b[tid];
    __syncthreads();
    const size_t tile = (size_t)blockIdx.x * 8 + wave;
    f32x4 e[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) e[mb] = ld4(x + (tile * 16 + i) * H + 16 * mb + 4 * q);
    for (int it = 0; it < iters; ++it) {
        int z = 0;
        asm volatile("" : "+v"(z));
        f32x4 acc[4], h1[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = ld4(bias + 16 * mb + 4 * q);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 wv[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) wv[mb] = ld4(wa + z + (16 * mb + i) * LDW + 16 * a + 4 * q);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mb][bb], e[a][bb], acc[mb], 0, 0, 0);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { h1[mb] = silu4(acc[mb]); acc[mb] = ld4(bias + H + 16 * mb + 4 * q); }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 wv[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) wv[mb] = ld4(wb + z + (16 * mb + i) * LDW + 16 * a + 4 * q);
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[mb][bb], h1[a][bb], acc[mb], 0, 0, 0);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) e[mb] = silu4(acc[mb]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) st4(y + (tile * 16 + i) * H + 16 * mb + 4 * q, e[mb]);
}

// ---------------------------------------------------------------- 3 x bf16 split version
// Two accumulator blocks (8 fp32 values of a lane: hidden 32kb + 4q + r and 32kb + 16 + 4q + r) -> the lane's B fragment
// of k-block kb in three bf16 pieces.  The k order inside the block is a permutation; the weights use the same one.
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xv = j < 4 ? v0[j] : v1[j - 4];
        const __bf16 h = (__bf16)xv;
        const float r1 = xv - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[j] = h; mid[j] = m; lo[j] = (__bf16)r2;
    }
}

// LDS image of one 64 x 64 weight matrix: [term 3][mb 4][kb 2][lane 64] fragments of 8 bf16 (16 bytes).
constexpr int WFRAG = 3 * 4 * 2 * 64;     // in units of bf16x8
__device__ __forceinline__ void stage_split(bf16x8* lds, const float* __restrict__ w, int tid) {
    // 512 threads: thread -> (mb, kb, lane)
    const int lane = tid & 63, mk = tid >> 6, mb = mk >> 1, kb = mk & 1, m = lane & 15, q = lane >> 4;
    const f32x4 v0 = ld4(w + (16 * mb + m) * H + 32 * kb + 4 * q);
    const f32x4 v1 = ld4(w + (16 * mb + m) * H + 32 * kb + 16 + 4 * q);
    bf16x8 hi, mid, lo;
    split8(v0, v1, hi, mid, lo);
    lds[(0 * 8 + mk) * 64 + lane] = hi;
    lds[(1 * 8 + mk) * 64 + lane] = mid;
    lds[(2 * 8 + mk) * 64 + lane] = lo;
}

template <int TERMS>
__device__ __forceinline__ void gemm_split(const bf16x8* __restrict__ wl, const f32x4 (&x)[4], f32x4 (&acc)[4], int lane) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        bf16x8 xh, xm, xl;
        split8(x[2 * kb], x[2 * kb + 1], xh, xm, xl);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const bf16x8 wh = wl[(0 * 8 + 2 * mb + kb) * 64 + lane];
            const bf16x8 wm = wl[(1 * 8 + 2 * mb + kb) * 64 + lane];
            const bf16x8 wlo = wl[(2 * 8 + 2 * mb + kb) * 64 + lane];
            // small terms first
            if (TERMS >= 6) {
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, acc[mb], 0, 0, 0);
            }
            if (TERMS >= 3) {
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, acc[mb], 0, 0, 0);
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, acc[mb], 0, 0, 0);
            }
            acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[mb], 0, 0, 0);
        }
    }
}

template <int TERMS, int SKEW = 0>
__global__ void __launch_bounds__(512)
k_split(const float* __restrict__ w1, const float* __restrict__ w2, const float* __restrict__ b, const float* __restrict__ x,
        float* __restrict__ y, int iters) {
    __shared__ __attribute__((aligned(16))) bf16x8 wa[WFRAG], wb[WFRAG];
    __shared__ __attribute__((aligned(16))) float bias[2 * H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    stage_split(wa, w1, tid);
    stage_split(wb, w2, tid);
    if (tid < 2 * H) bias[tid] = b[tid];
    __syncthreads();
    const size_t tile = (size_t)blockIdx.x * 8 + wave;
    f32x4 e[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) e[mb] = ld4(x + (tile * 16 + i) * H + 16 * mb + 4 * q);
    if (SKEW > 0 && wave >= 4) {        // stagger the second wave of every SIMD (MI355X_MICROARCH.md, two waves per SIMD, item 9)
        for (int k = 0; k < SKEW; ++k) __builtin_amdgcn_s_sleep(8);
    }
    for (int it = 0; it < iters; ++it) {
        int z = 0;
        asm volatile("" : "+v"(z));     // opaque: fragments are re-read from LDS per tile, as in the real kernel
        f32x4 acc[4], h1[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = ld4(bias + 16 * mb + 4 * q);
        gemm_split<TERMS>(wa + z, e, acc, lane);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) { h1[mb] = silu4(acc[mb]); acc[mb] = ld4(bias + H + 16 * mb + 4 * q); }
        gemm_split<TERMS>(wb + z, h1, acc, lane);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) e[mb] = silu4(acc[mb]);
    }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) st4(y + (tile * 16 + i) * H + 16 * mb + 4 * q, e[mb]);
}

static double silu_d(double z) { return z / (1.0 + std::exp(-z)); }

int main() {
    const int WGS = 256, TILES = WGS * 8, ROWS = TILES * 16;
    std::vector<float> w1(H * H), w2(H * H), b(2 * H), x((size_t)ROWS * H);
    srand(7);
    auto u = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto& v : w1) v = u() * 0.25f;
    for (auto& v : w2) v = u() * 0.25f;
    for (auto& v : b) v = u() * 0.1f;
    for (auto& v : x) v = u() * 3.0f;
    float *dw1, *dw2, *db, *dx, *dy;
    CHECK(hipMalloc(&dw1, H * H * 4)); CHECK(hipMalloc(&dw2, H * H * 4)); CHECK(hipMalloc(&db, 2 * H * 4));
    CHECK(hipMalloc(&dx, x.size() * 4)); CHECK(hipMalloc(&dy, x.size() * 4));
    CHECK(hipMemcpy(dw1, w1.data(), H * H * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw2, w2.data(), H * H * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), 2 * H * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    // fp64 reference of one iteration, first 64 tiles
    const int CR = 64 * 16;
    std::vector<double> ref((size_t)CR * H);
    for (int r = 0; r < CR; ++r) {
        double h1[H];
        for (int o = 0; o < H; ++o) {
            double s = b[o];
            for (int k = 0; k < H; ++k) s += (double)w1[o * H + k] * x[(size_t)r * H + k];
            h1[o] = silu_d(s);
        }
        for (int o = 0; o < H; ++o) {
            double s = b[H + o];
            for (int k = 0; k < H; ++k) s += (double)w2[o * H + k] * h1[k];
            ref[(size_t)r * H + o] = silu_d(s);
        }
    }
    std::vector<float> y(x.size());
    auto err = [&](const char* name) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        double mx = 0, sc = 0, rms = 0;
        for (size_t k = 0; k < ref.size(); ++k) { mx = fmax(mx, fabs(y[k] - ref[k])); sc = fmax(sc, fabs(ref[k])); rms += (y[k] - ref[k]) * (y[k] - ref[k]); }
        printf("%-26s max|err| / max|ref| = %.3e   rms err = %.3e  (scale %.3f)\n", name, mx / sc, sqrt(rms / ref.size()), sc);
    };
    hipLaunchKernelGGL(k_f32, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, 1); err("fp32 MFMA 16x16x4");
    hipLaunchKernelGGL(k_split<6>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, 1); err("3 x bf16, 6 terms");
    hipLaunchKernelGGL(k_split<3>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, 1); err("3 x bf16, 3 terms");
    hipLaunchKernelGGL(k_split<1>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, 1); err("plain bf16 (1 term)");
    // timing
    const int IT = 200;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int k = 0; k < 3; ++k) launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        for (int k = 0; k < 5; ++k) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us_per_iter = ms * 1e3 / 5 / IT;
        printf("%-26s %.3f us per tile-iteration per wave (2 waves per SIMD: %.0f cycles per tile at 2.2 GHz)\n", name,
               us_per_iter, us_per_iter * 2200.0 / 2.0);
    };
    timeit("fp32 MFMA 16x16x4", [&] { hipLaunchKernelGGL(k_f32, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("3 x bf16, 6 terms", [&] { hipLaunchKernelGGL(k_split<6>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("6 terms, skew 1 x 512 cyc", [&] { hipLaunchKernelGGL((k_split<6, 1>), dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("6 terms, skew 2 x 512 cyc", [&] { hipLaunchKernelGGL((k_split<6, 2>), dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("6 terms, skew 3 x 512 cyc", [&] { hipLaunchKernelGGL((k_split<6, 3>), dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("6 terms, skew 5 x 512 cyc", [&] { hipLaunchKernelGGL((k_split<6, 5>), dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("3 x bf16, 3 terms", [&] { hipLaunchKernelGGL(k_split<3>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    timeit("plain bf16 (1 term)", [&] { hipLaunchKernelGGL(k_split<1>, dim3(WGS), dim3(512), 0, 0, dw1, dw2, db, dx, dy, IT); });
    return 0;
}
