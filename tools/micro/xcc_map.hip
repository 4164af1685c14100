// Which XCD does workgroup b of a 1-D grid land on?  (hwreg XCC_ID, gfx940+)   hipcc --offload-arch=gfx950 -O3 xcc_map.hip -o xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID[3:0]
        const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((32 - 1) << 11));     // HW_REG_HW_ID
        out[2 * blockIdx.x] = (int)xcc;
        out[2 * blockIdx.x + 1] = (int)hwid;
    }
    // keep the workgroup resident for a while so that the whole grid is co-resident
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
int main() {
    for (int threads : {512, 256}) for (int grid : {256, 512, 1024}) {
        int* d; hipMalloc(&d, grid * 2 * sizeof(int));
        hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d);
        std::vector<int> h(grid * 2);
        hipMemcpy(h.data(), d, grid * 2 * sizeof(int), hipMemcpyDeviceToHost);
        int ok = 0, okdiv = 0;
        for (int b = 0; b < grid; ++b) { ok += h[2 * b] == (b % 8); okdiv += h[2 * b] == (b / (grid / 8)) % 8; }
        printf("threads %d grid %d: xcc == b %% 8 for %d of %d workgroups (blocked mapping would give %d); first 24:", threads, grid, ok, grid, okdiv);
        for (int b = 0; b < 24; ++b) printf(" %d", h[2 * b]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
