// mfma_chain.hip -- issue cost of v_mfma_f32_16x16x32_bf16 when consecutive instructions accumulate into the SAME tuple
// (gemm_split's order: six terms of one row block in a row) against interleaved accumulators.  Cycles per MFMA for 1 and
// 2 waves per SIMD.  Diagnostic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_chain.hip -o tools/micro/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int NACC>
__global__ void __launch_bounds__(512) k(const bf16x8* ab, float* out, int iters, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const bf16x8 A = ab[lane], B = ab[64 + lane];
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 24 MFMAs per iteration in every variant
        if constexpr (NACC == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                             : "+v"(acc[c]) : "v"(A), "v"(B));
        } else if constexpr (NACC == 2) {
#pragma unroll
            for (int c = 0; c < 4; c += 2)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1"
                             : "+v"(acc[c]), "+v"(acc[c + 1]) : "v"(A), "v"(B));
        } else {
#pragma unroll
            for (int r = 0; r < 6; ++r)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t"
                             "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %5, %3"
                             : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(A), "v"(B));
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = acc[c] * 1e-6f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}

template <typename K>
static void run(const char* name, K kern, int threads, const bf16x8* ab, float* out, unsigned long long* cyc, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, ab, out, 10, cyc);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, ab, out, iters, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c;
    CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = 24.0 * iters;                         // MFMAs per wave
    const int wps = threads / 256;                         // waves per SIMD
    printf("%-44s %d wave(s)/SIMD: %.1f s_memtime ticks per MFMA per wave, %.2f us total -> %.1f ns per MFMA per SIMD\n", name, wps,
           (double)c / n, ms * 1e3, ms * 1e6 / (n * wps));
}

int main() {
    bf16x8* ab; float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&ab, 128 * 16)); CHECK(hipMemset(ab, 0, 128 * 16));
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 8));
    const int iters = 20000;
    for (int threads : {256, 512}) {
        run("six in a row on one accumulator", k<1>, threads, ab, out, cyc, iters);
        run("two accumulators alternating", k<2>, threads, ab, out, cyc, iters);
        run("four accumulators round-robin", k<4>, threads, ab, out, cyc, iters);
    }
    return 0;
}
