// pkfma_mfma.hip -- follow-up to agpr_hazard.hip P4: a pair of packed fp32 FMAs with op_sel modifiers
// gives wrong results when the OTHER wave of the same SIMD is issuing matrix instructions.  This program maps the
// condition: which form of the VALU sequence (SEQ), next to which neighbour (NEIGH), fails in which lanes / halves.
// Waves 0-3 of a 512-thread workgroup run the sequence on operands loaded from global memory and check it against
// scalar FMAs (relative 1e-4: the failure is a whole product missing); waves 4-7 (second wave of the same SIMDs) run
// the neighbour loop.  Diagnostic, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/pkfma_mfma.hip -o tools/micro/pkfma_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// NEIGH: 0 idle | 1 bf16 MFMA 16x16x32 chains | 2 fp32 MFMA 16x16x4 chains | 3 v_fma_f32 loop | 4 v_pk_fma_f32 loop
//        5 bf16 MFMA 32x32x16 chains
template <int NEIGH>
__device__ __forceinline__ void neighbour(const bf16x8* __restrict__ ab, int iters, float* sink, int lane) {
    if constexpr (NEIGH == 0) return;
    const bf16x8 A = ab[lane], B = ab[64 + lane];
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    float s = (float)lane, t = 1.0f;
    f32x2 p = {1.0f, 2.0f}, qv = {0.5f, 0.25f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big = {};
    for (int it = 0; it < iters; ++it) {
        if constexpr (NEIGH == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7"
                             : "+v"(acc[c]) : "v"(A), "v"(B));
        } else if constexpr (NEIGH == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 7"
                             : "+v"(acc[c]) : "v"(s), "v"(t));
        } else if constexpr (NEIGH == 3) {
            asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\t"
                         "v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1"
                         : "+v"(s) : "v"(t));
        } else if constexpr (NEIGH == 4) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\t"
                         "v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1"
                         : "+v"(p) : "v"(qv));
        } else {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15"
                         : "+v"(big) : "v"(A), "v"(B));
        }
        if ((it & 31) == 31) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = acc[c] * 1e-3f;
            s = s * 1e-3f + 1.0f; p = p * 1e-3f; big = big * 1e-3f;
        }
    }
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + s + p[0] + big[0] == 12345.678f) sink[0] = 1.0f;
}

// SEQ: 0 the compiled sequence | 1 second FMA without op_sel (pre-swizzled operand) | 2 an s_nop 0 after every instruction
//      3 only the second FMA (accumulator from a v_mov pair) | 4 second FMA into a fresh destination | 5 scalar v_fma_f32
//      6 the compiled sequence with v_pk_mul/v_pk_add instead of the second FMA | 7 first FMA only (op_sel_hi:[1,0,0])
template <int SEQ, int NEIGH>
__global__ void __launch_bounds__(512)
k_seq(const bf16x8* __restrict__ ab, unsigned long long* __restrict__ bad, int iters, const float* __restrict__ gtab,
      float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) { neighbour<NEIGH>(ab, iters * 2, sink, lane); return; }
    unsigned long long nlo = 0, nhi = 0;
    unsigned slot = wave * 31 + blockIdx.x * 7;
    for (int it = 0; it < iters; ++it) {
        slot = (slot * 5 + 1) & 4095;
        const float* src = gtab + ((size_t)slot * 64 + lane) * 8;     // per lane: R00 R01 | R10 R11 | rel0 rel1 | y0 y1
        f32x2 r0, r1, rel, y, d, z, e;
        float t;
#define LOADS "global_load_dwordx2 %0, %8, off\n\tglobal_load_dwordx2 %2, %8, off offset:16\n\t" \
              "global_load_dwordx2 %3, %8, off offset:24\n\tglobal_load_dwordx2 %1, %8, off offset:8\n\ts_waitcnt vmcnt(0)\n\t"
#define OUTS : "=&v"(r0), "=&v"(r1), "=&v"(rel), "=&v"(y), "=&v"(d), "=&v"(z), "=&v"(t), "=&v"(e) : "v"(src), "v"(lane) : "memory"
        if constexpr (SEQ == 0)
            asm volatile(LOADS "v_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\t"
                         "v_mov_b32_e32 %6, %9\n\tv_pk_fma_f32 %4, %1, %2, %4 op_sel:[0,1,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 1)
            asm volatile(LOADS "v_mov_b32_e32 %7, %2\n\tv_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\t"
                         "v_mov_b32_e32 %6, %9\n\tv_pk_fma_f32 %4, %1, %2, %4 op_sel:[0,1,0]\n\ts_nop 4" OUTS);   // placeholder, replaced below
        else if constexpr (SEQ == 2)
            asm volatile(LOADS "v_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\ts_nop 0\n\t"
                         "v_mov_b32_e32 %6, %9\n\ts_nop 0\n\tv_pk_fma_f32 %4, %1, %2, %4 op_sel:[0,1,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 3)
            asm volatile(LOADS "v_pk_mul_f32 %4, %0, %2 op_sel_hi:[1,0]\n\ts_nop 4\n\tv_pk_fma_f32 %4, %1, %2, %4 op_sel:[0,1,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 4)
            asm volatile(LOADS "v_pk_fma_f32 %7, %0, %2, 0 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\t"
                         "v_mov_b32_e32 %6, %9\n\tv_pk_fma_f32 %4, %1, %2, %7 op_sel:[0,1,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 5)
            asm volatile(LOADS "v_mul_f32 %4, %0, %2\n\tv_mov_b32_e32 %6, %9\n\ts_nop 4" OUTS);   // placeholder, replaced below
        else if constexpr (SEQ == 6)
            asm volatile(LOADS "v_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\tv_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\t"
                         "v_mov_b32_e32 %6, %9\n\tv_pk_mul_f32 %7, %1, %2 op_sel:[0,1]\n\tv_pk_add_f32 %4, %4, %7\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 7)
            asm volatile(LOADS "v_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 8)        // the modifier on src0: operands swapped (the product is commutative)
            asm volatile(LOADS "v_pk_mul_f32 %4, %0, %2 op_sel_hi:[1,0]\n\ts_nop 4\n\tv_pk_fma_f32 %4, %2, %1, %4 op_sel:[1,0,0]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 9)        // the modifier on src2 only: d = (R00 rel0 + y1, R01 rel1 + y1)
            asm volatile(LOADS "v_pk_fma_f32 %4, %0, %2, %3 op_sel:[0,0,1]\n\ts_nop 4" OUTS);
        else if constexpr (SEQ == 10)       // broadcast of the LOW half to both lanes (op_sel_hi cleared on src1)
            asm volatile(LOADS "v_pk_mul_f32 %4, %0, %2 op_sel_hi:[1,0]\n\ts_nop 4\n\tv_pk_fma_f32 %4, %1, %2, %4 op_sel_hi:[1,0,1]\n\ts_nop 4" OUTS);
        else                                // v_pk_add_f32 with the modifier: d = (R00 + rel1, R01 + rel1)
            asm volatile(LOADS "v_pk_add_f32 %4, %0, %2 op_sel:[0,1]\n\ts_nop 4" OUTS);
        float w0, w1;
        if constexpr (SEQ == 7) { w0 = r0[0] * rel[0]; w1 = r0[1] * rel[0]; }
        else if constexpr (SEQ == 9) { w0 = __builtin_fmaf(r0[0], rel[0], y[1]); w1 = __builtin_fmaf(r0[1], rel[1], y[1]); }
        else if constexpr (SEQ == 10) { w0 = __builtin_fmaf(r1[0], rel[0], r0[0] * rel[0]); w1 = __builtin_fmaf(r1[1], rel[0], r0[1] * rel[0]); }
        else if constexpr (SEQ == 11) { w0 = r0[0] + rel[1]; w1 = r0[1] + rel[1]; }
        else { w0 = __builtin_fmaf(r1[0], rel[1], r0[0] * rel[0]); w1 = __builtin_fmaf(r1[1], rel[1], r0[1] * rel[0]); }
        if (__builtin_fabsf(d[0] - w0) > 1e-4f * (__builtin_fabsf(w0) + 1.0f)) ++nlo;
        if (__builtin_fabsf(d[1] - w1) > 1e-4f * (__builtin_fabsf(w1) + 1.0f)) ++nhi;
    }
    // counters: [quad of lanes 0..3][half lo/hi]
    if (nlo) atomicAdd(bad + (lane >> 4) * 2, nlo);
    if (nhi) atomicAdd(bad + (lane >> 4) * 2 + 1, nhi);
}

template <typename K>
static void run(const char* name, K kernel, const bf16x8* d_ab, unsigned long long* d_bad, int blocks, int iters, const float* d_g, float* d_sink) {
    CHECK(hipMemset(d_bad, 0, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), 0, 0, d_ab, d_bad, iters, d_g, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long b[8];
    CHECK(hipMemcpy(b, d_bad, 64, hipMemcpyDeviceToHost));
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += b[k];
    const double lane_trials = (double)blocks * 256 * iters;
    printf("%-64s wrong %10llu of %.2e lane-results (%.1e); by lanes 0-15/16-31/32-47/48-63 lo|hi: %llu|%llu %llu|%llu %llu|%llu %llu|%llu  %.0f ms\n",
           name, tot, 2 * lane_trials, tot / (2 * lane_trials), b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], ms);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    std::vector<uint16_t> ab(2 * 64 * 8);
    for (auto& v : ab) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    std::vector<float> g((size_t)4096 * 64 * 8);
    for (auto& v : g) v = rnd() * 8.0f;
    bf16x8* d_ab; unsigned long long* d_bad; float *d_g, *d_sink;
    CHECK(hipMalloc(&d_ab, ab.size() * 2)); CHECK(hipMalloc(&d_bad, 64)); CHECK(hipMalloc(&d_g, g.size() * 4)); CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemcpy(d_ab, ab.data(), ab.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_g, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    printf("pkfma_mfma: %d blocks x 4 checking waves x %d iterations\n", blocks, iters);
#define RUN(SEQ, NEIGH, text) run(text, k_seq<SEQ, NEIGH>, d_ab, d_bad, blocks, iters, d_g, d_sink)
    RUN(0, 0, "compiled sequence | neighbour idle");
    RUN(0, 1, "compiled sequence | neighbour bf16 MFMA 16x16x32");
    RUN(0, 5, "compiled sequence | neighbour bf16 MFMA 32x32x16");
    RUN(0, 2, "compiled sequence | neighbour fp32 MFMA 16x16x4");
    RUN(0, 3, "compiled sequence | neighbour v_fma_f32 loop");
    RUN(0, 4, "compiled sequence | neighbour v_pk_fma_f32 loop");
    RUN(2, 1, "s_nop 0 after every instruction | bf16 MFMA 16x16x32");
    RUN(3, 1, "v_pk_mul, 5 idle, second FMA op_sel:[0,1,0] alone | bf16 MFMA");
    RUN(4, 1, "second FMA into a fresh destination | bf16 MFMA");
    RUN(6, 1, "second FMA as v_pk_mul op_sel + v_pk_add | bf16 MFMA");
    RUN(7, 1, "first FMA only (op_sel_hi:[1,0,0]) | bf16 MFMA");
    RUN(8, 1, "modifier on src0 (op_sel:[1,0,0]) | bf16 MFMA");
    RUN(9, 1, "modifier on src2 (op_sel:[0,0,1]) | bf16 MFMA");
    RUN(10, 1, "low half to both lanes (op_sel_hi:[1,0,1]) | bf16 MFMA");
    RUN(11, 1, "v_pk_add_f32 op_sel:[0,1] | bf16 MFMA");
    return 0;
}
