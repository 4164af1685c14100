// The rest of the runner's training step around forward and backward (experiments/lorentz/main.py:164,289-292):
// MSE loss with its gradient, and AdamW over every parameter tensor, one launch each.  torch spends 4 + 3 launches
// (≈ 45 us of a 300 us captured step at cfg2) on them.
#pragma once
#include "common.h"

namespace {

// loss = mean((pred - target)^2), dpred = 2 (pred - target) / n   (nn.MSELoss and its backward seed, main.py:86,289-290).
// Up to 256 workgroups; partial sums meet in `scratch` ([256] floats + one int counter, zero at the first launch) and
// the last workgroup to finish adds them in index order (fixed order: bit-stable), then re-arms the counter.
__global__ void __launch_bounds__(256)
k_mse_loss_grad(const float* __restrict__ pred, const float* __restrict__ target, int64_t n, float* __restrict__ loss,
                float* __restrict__ dpred, float* __restrict__ scratch) {
    __shared__ float red[256];
    __shared__ int last;
    const float inv = 1.0f / (float)n;
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = pred[i] - target[i];
        dpred[i] = 2.0f * d * inv;
        s += d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (gridDim.x == 1) {                      // small inputs: no round trips through memory
        if (threadIdx.x == 0) *loss = red[0] * inv;
        return;
    }
    int* counter = reinterpret_cast<int*>(scratch + 256);
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + blockIdx.x, red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    red[threadIdx.x] = threadIdx.x < gridDim.x ? __hip_atomic_load(scratch + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *loss = red[0] * inv;
        *counter = 0;
    }
}

constexpr int ADAMW_MAX_TENSORS = 64;
struct AdamWTable {
    float* param[ADAMW_MAX_TENSORS];
    const float* grad[ADAMW_MAX_TENSORS];
    float* exp_avg[ADAMW_MAX_TENSORS];
    float* exp_avg_sq[ADAMW_MAX_TENSORS];
    int numel[ADAMW_MAX_TENSORS];
    int block0[ADAMW_MAX_TENSORS + 1];        // first workgroup of every tensor (1,024 elements per workgroup)
    int n;
};

// torch.optim.AdamW (decoupled weight decay, no amsgrad, no maximize), the arithmetic of its fused kernel:
//   p -= lr wd p;  m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// for every tensor of the table (up to 64) in ONE launch.  `step` (device scalar, the number of steps taken so far) and `lr` live in
// device memory, so a captured launch follows the counter and a learning-rate schedule.  The last workgroup to finish
// writes step + 1 (every workgroup has read it by then) and re-arms the counter.
__global__ void __launch_bounds__(256)
k_adamw(const AdamWTable T, float* __restrict__ step, const float* __restrict__ lr_dev, int* __restrict__ counter,
        double beta1, double beta2, double log_beta1, double log_beta2, float eps, double weight_decay, float grad_scale,
        int bump) {
    __shared__ float sh[3];
    if (threadIdx.x == 0) {
        // 1 - beta^t = -expm1(t log beta): no cancellation at small t, and no double-precision pow (microseconds on the
        // critical path of a 10 us launch); log beta arrives rounded from the host's double
        const float t = *step + 1.0f, lr = *lr_dev;
        const float bc1 = -expm1f(t * (float)log_beta1), bc2 = -expm1f(t * (float)log_beta2);
        sh[0] = lr / bc1;
        sh[1] = sqrtf(bc2);
        sh[2] = 1.0f - lr * (float)weight_decay;
    }
    __syncthreads();
    const float step_size = sh[0], bc2_sqrt = sh[1], decay = sh[2];
    // 1 - beta in double, then rounded (torch passes python doubles: 1.0f - 0.999f is off by 5e-5 relative)
    const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    // which tensor: one table entry per lane and a ballot (a serial scan of 47 kernel-argument loads cost 3 us)
    const int lane = threadIdx.x & 63;
    const unsigned long long mine = __ballot(lane < T.n && (int)blockIdx.x >= T.block0[lane]);
    const int ti = __popcll(mine) - 1;
    const int base = ((int)blockIdx.x - T.block0[ti]) * 1024;
    const int n = T.numel[ti];
    float* __restrict__ p = T.param[ti];
    const float* __restrict__ g = T.grad[ti];
    float* __restrict__ m = T.exp_avg[ti];
    float* __restrict__ v = T.exp_avg_sq[ti];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = base + r * 256 + (int)threadIdx.x;
        if (i < n) {
            const float gi = g[i] * grad_scale;      // 1 / world after a sum all-reduce (data-parallel); x 1.0f is exact
            float pi = p[i] * decay;
            const float mi = m[i] + omb1 * (gi - m[i]);
            const float vi = b2 * v[i] + omb2 * gi * gi;
            m[i] = mi;
            v[i] = vi;
            const float denom = sqrtf(vi) / bc2_sqrt + eps;
            pi -= step_size * mi / denom;
            p[i] = pi;
        }
    }
    if (!bump) return;                        // more than 64 tensors: only the last launch of a step counts it
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(counter, 1) == (int)gridDim.x - 1) {
            *step = *step + 1.0f;
            *counter = 0;
        }
    }
}

}  // namespace
