// Microbenchmark: does VALU work overlap with v_mfma_f32_16x16x4_f32 on gfx950?
// variant 0: MFMA only; 1: MFMA + N independent v_fma in the same wave; 2: MFMA + N v_exp (trans);
// block = 64*W threads (W waves per SIMD when 4*W waves per CU... we launch 256 blocks of 256*W threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
    f32x4 acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = a + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32; ++g) {          // 128 MFMAs per iteration, NV valu per 4 MFMAs
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 1) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
                if (KIND == 2) v[j & 7] = __builtin_amdgcn_exp2f(v[j & 7]) * 0.5f;
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, int KIND> void run(const char* name, int waves_per_simd, float* d) {
    int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(256), block(256 * waves_per_simd);
    k<NV, KIND><<<grid, block>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, KIND><<<grid, block>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = 128.0 * iters * waves_per_simd;              // per SIMD
    double us = ms * 1e3;
    printf("%-28s waves/SIMD=%d  NV=%2d  time=%8.1f us  ns per MFMA per SIMD=%6.2f  (32 cyc @2.4GHz = 13.3 ns)  TF=%.1f\n",
           name, waves_per_simd, NV, us, us * 1e3 / mfma, 1024.0 * mfma * 2048.0 / (us * 1e-6) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("mfma only", w, d);
        run<2, 1>("mfma + 2 fma per 4 mfma", w, d);
        run<5, 1>("mfma + 5 fma per 4 mfma", w, d);
        run<10, 1>("mfma + 10 fma per 4 mfma", w, d);
        run<20, 1>("mfma + 20 fma per 4 mfma", w, d);
        run<2, 2>("mfma + 2 exp+mul per 4 mfma", w, d);
        run<5, 2>("mfma + 5 exp+mul per 4 mfma", w, d);
    }
    return 0;
}
