// split_T.hip -- gemm_split_T (common.h): the transposed product read from the forward's split image with
// ds_read_b64_tr_b16, against the fp32 MFMA chain on a transposed fp32 copy (fb_gemm's arithmetic) and an fp64 host sum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/micro/split_T.hip -o tools/micro/split_T
#include "../../aether_amd/csrc/common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// W [64][64] (KBN = 2) or [64][32] (KBN = 1); act [16 items][64]; out [16 items][32 KBN] = act . W   (out[item][k] = sum_m W[m][k] act[item][m])
template <int KBN>
__global__ void __launch_bounds__(64) k_test(const float* __restrict__ W, const float* __restrict__ act, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float img[3 * 4 * KBN * 64 * 4];
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    constexpr int C = 32 * KBN;
    for (int idx = lane; idx < 64 * C / 4; idx += 64) {
        const int rr = idx / (C / 4), cc = (idx % (C / 4)) * 4;
        stage_split4<4, KBN>(img, rr, cc, ld4(W + rr * C + cc));
    }
    __syncthreads();
    f32x4 a[4], acc[2 * KBN];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) a[mb] = ld4(act + i * 64 + 16 * mb + 4 * q);
#pragma unroll
    for (int ob = 0; ob < 2 * KBN; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_split_T<KBN>(img, a, acc, lane);
#pragma unroll
    for (int ob = 0; ob < 2 * KBN; ++ob) st4(out + i * C + 16 * ob + 4 * q, acc[ob]);
}

template <int KBN>
int run() {
    constexpr int C = 32 * KBN;
    std::vector<float> W(64 * C), A(16 * 64), O(16 * C);
    srand(7 + KBN);
    for (auto& v : W) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : A) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 3.f;
    float *dW, *dA, *dO;
    CHECK(hipMalloc(&dW, W.size() * 4)); CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dO, O.size() * 4));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    k_test<KBN><<<1, 64>>>(dW, dA, dO);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int it = 0; it < 16; ++it)
        for (int k = 0; k < C; ++k) {
            double s = 0;
            for (int m = 0; m < 64; ++m) s += (double)W[m * C + k] * (double)A[it * 64 + m];
            worst = fmax(worst, fabs(s - (double)O[it * C + k]));
            scale = fmax(scale, fabs(s));
        }
    printf("KBN=%d ([64][%d] image): max |gemm_split_T - fp64| = %.3e, scale %.3f -> %.2e scale-relative\n", KBN, C, worst, scale, worst / scale);
    return worst / scale > 1e-6;
}

int main() { return run<2>() | run<1>(); }
