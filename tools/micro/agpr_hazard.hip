// agpr_hazard.hip -- isolates the two instruction patterns that distinguish the corrupting build of k_edge_layer1
//  from the clean one, each in a loop of its own with every wait written by hand:
//
//   P1  an LDS load whose destination is an accumulator register, consumed as SrcC by a bf16 MFMA right after the
//       s_waitcnt that covers it (optionally with younger LDS loads still in flight, as the compiled code has them);
//   P2  a bf16 MFMA whose destination tuple partially overlaps its SrcC tuple (a[2:5] <- a[4:7]: the register
//       allocator's "sliding" tuples), optionally with LDS loads returning while it executes.
//
// Every trial is checked bit for bit against the same operation done the plain way (VGPR accumulator, disjoint or
// identical tuples).  Prints trials and mismatches per pattern.  Diagnostic, not product code.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/agpr_hazard.hip -o tools/micro/agpr_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

constexpr int SLOTS = 128;          // LDS table: SLOTS x 64 lanes x 16 B = 128 KB
constexpr int LDS_BYTES = SLOTS * 64 * 16 + 8192;   // + slack for the filler loads' offsets

__device__ __forceinline__ bool same(f32x4 a, f32x4 b) {
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    return x[0] == y[0] && x[1] == y[1] && x[2] == y[2] && x[3] == y[3];
}


// The reference result: the same MFMA with its accumulator in VGPRs (loaded by a plain, compiler-counted LDS load that
// is forced into VGPRs), fully padded.
__device__ __forceinline__ f32x4 mfma_vgpr(bf16x8 A, bf16x8 B, f32x4 c) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15" : "+v"(c) : "v"(A), "v"(B));
    return c;
}
__device__ __forceinline__ f32x4 lds_vgpr(const float* p) {
    f32x4 c = *reinterpret_cast<const f32x4*>(p);
    asm volatile("" : "+v"(c));
    return c;
}

// MODE 0: consume right after lgkmcnt(0).  MODE 1: three younger LDS loads (VGPR destinations) in flight, wait
// lgkmcnt(3) -- the accumulator load is then the oldest and only it has to be back.  MODE 2: as 0, with 8 idle states
// between the wait and the MFMA.  ZERO_AB: A = B = 0, so D must equal the loaded C exactly.
template <int MODE>
__global__ void __launch_bounds__(256)
k_p1(const float* __restrict__ init, const bf16x8* __restrict__ ab, unsigned long long* __restrict__ bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int idx = threadIdx.x; idx < SLOTS * 64 * 4; idx += 256) tab[idx] = init[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8 A = ab[lane], B = ab[64 + lane];
    unsigned long long nbad = 0;
    unsigned slot = wave * 31 + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        slot = (slot * 5 + 1) & (SLOTS - 1);
        const unsigned addr = (slot * 64 + lane) * 16;           // LDS byte address (dynamic LDS starts at 0)
        const f32x4 c = lds_vgpr(tab + (slot * 64 + lane) * 4);
        const f32x4 want = mfma_vgpr(A, B, c);
        f32x4 acc, t0, t1, t2;
        if constexpr (MODE == 0) {
            asm volatile("ds_read_b128 %0, %1\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"
                         "s_nop 15\n\ts_nop 7"
                         : "=&a"(acc) : "v"(addr), "v"(A), "v"(B) : "memory");
        } else if constexpr (MODE == 1) {
            asm volatile("ds_read_b128 %0, %4\n\t"
                         "ds_read_b128 %1, %4 offset:1024\n\t"
                         "ds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072\n\t"
                         "s_waitcnt lgkmcnt(3)\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\ts_nop 7"
                         : "=&a"(acc), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(addr), "v"(A), "v"(B) : "memory");
        } else {
            asm volatile("ds_read_b128 %0, %1\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"
                         "s_nop 15\n\ts_nop 7"
                         : "=&a"(acc) : "v"(addr), "v"(A), "v"(B) : "memory");
        }
        if (!same(acc, want)) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// P2: D tuple a[2:5], SrcC tuple a[4:7].  MODE 0: nothing else in flight.  MODE 1: two LDS loads (VGPR destinations)
// issued just before the MFMA return while it executes.  MODE 2: the control -- identical tuples a[4:7] <- a[4:7].
template <int MODE>
__global__ void __launch_bounds__(256)
k_p2(const float* __restrict__ init, const bf16x8* __restrict__ ab, unsigned long long* __restrict__ bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int idx = threadIdx.x; idx < SLOTS * 64 * 4; idx += 256) tab[idx] = init[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8 A = ab[lane], B = ab[64 + lane];
    unsigned long long nbad = 0;
    unsigned slot = wave * 31 + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        slot = (slot * 5 + 1) & (SLOTS - 1);
        const unsigned addr = (slot * 64 + lane) * 16;
        const f32x4 c = lds_vgpr(tab + (slot * 64 + lane) * 4);
        const f32x4 want = mfma_vgpr(A, B, c);
        float d0, d1, d2, d3;
        f32x4 t0, t1;
        if constexpr (MODE == 0) {
            asm volatile("v_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\t"
                         "v_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\t"
                         "v_accvgpr_write_b32 a2, %7\n\tv_accvgpr_write_b32 a3, %6\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_16x16x32_bf16 a[2:5], %8, %9, a[4:7]\n\t"
                         "s_nop 15\n\ts_nop 7\n\t"
                         "v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3\n\t"
                         "v_accvgpr_read_b32 %2, a4\n\tv_accvgpr_read_b32 %3, a5\n\ts_nop 1"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                         : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(A), "v"(B)
                         : "a2", "a3", "a4", "a5", "a6", "a7", "memory");
        } else if constexpr (MODE == 1) {
            asm volatile("v_accvgpr_write_b32 a4, %6\n\tv_accvgpr_write_b32 a5, %7\n\t"
                         "v_accvgpr_write_b32 a6, %8\n\tv_accvgpr_write_b32 a7, %9\n\t"
                         "v_accvgpr_write_b32 a2, %9\n\tv_accvgpr_write_b32 a3, %8\n\t"
                         "s_nop 7\n\t"
                         "ds_read_b128 %4, %12\n\t"
                         "ds_read_b128 %5, %12 offset:4096\n\t"
                         "v_mfma_f32_16x16x32_bf16 a[2:5], %10, %11, a[4:7]\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\ts_nop 7\n\t"
                         "v_accvgpr_read_b32 %0, a2\n\tv_accvgpr_read_b32 %1, a3\n\t"
                         "v_accvgpr_read_b32 %2, a4\n\tv_accvgpr_read_b32 %3, a5\n\ts_nop 1"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(t0), "=&v"(t1)
                         : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(A), "v"(B), "v"(addr)
                         : "a2", "a3", "a4", "a5", "a6", "a7", "memory");
        } else {
            asm volatile("v_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\t"
                         "v_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_16x16x32_bf16 a[4:7], %8, %9, a[4:7]\n\t"
                         "s_nop 15\n\ts_nop 7\n\t"
                         "v_accvgpr_read_b32 %0, a4\n\tv_accvgpr_read_b32 %1, a5\n\t"
                         "v_accvgpr_read_b32 %2, a6\n\tv_accvgpr_read_b32 %3, a7\n\ts_nop 1"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                         : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(A), "v"(B)
                         : "a4", "a5", "a6", "a7", "memory");
        }
        if (!same(f32x4{d0, d1, d2, d3}, want)) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

// P3: the register allocator's sequence as compiled into the corrupting kernel -- four chains of six MFMAs whose result
// tuples slide by two registers (a[16:19], a[18:21], a[20:23], a[22:25]), the first MFMA of chains 2-4 taking its SrcC
// from a tuple an LDS load filled (a[20:23], a[24:27], a[28:31]), halves read out between the chains with the
// compiler's idle-state counts, LDS loads of the next operands in flight.  LOADS = 1: accumulators come from LDS loads
// (as compiled); LOADS = 0: the same values written with v_accvgpr_write (no memory load lands in an AGPR).
template <int LOADS>
__global__ void __launch_bounds__(256)
k_p3(const float* __restrict__ init, const bf16x8* __restrict__ ab, unsigned long long* __restrict__ bad, int iters, float* dbg) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    for (int idx = threadIdx.x; idx < SLOTS * 64 * 4; idx += 256) tab[idx] = init[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8 A = ab[lane], B = ab[64 + lane];
    unsigned long long nbad = 0;
    unsigned slot = wave * 31 + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        slot = (slot * 5 + 1) & (SLOTS - 1);
        const unsigned s0 = slot, s1 = (slot + 17) & (SLOTS - 1), s2 = (slot + 41) & (SLOTS - 1), s3 = (slot + 77) & (SLOTS - 1);
        const unsigned a0 = (s0 * 64 + lane) * 16, a1 = (s1 * 64 + lane) * 16, a2 = (s2 * 64 + lane) * 16, a3 = (s3 * 64 + lane) * 16;
        f32x4 c[4] = {lds_vgpr(tab + (s0 * 64 + lane) * 4), lds_vgpr(tab + (s1 * 64 + lane) * 4),
                      lds_vgpr(tab + (s2 * 64 + lane) * 4), lds_vgpr(tab + (s3 * 64 + lane) * 4)};
        f32x4 want[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            want[ch] = c[ch];
#pragma unroll
            for (int r = 0; r < 6; ++r) want[ch] = mfma_vgpr(A, B, want[ch]);
        }
        float g[16];
        f32x4 w0, w1, w2;
#define FIVE(D) "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " D "\n\t" "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " D "\n\t" \
               "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " D "\n\t" "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " D "\n\t" \
               "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " D "\n\t"
#define CHAIN(D, C) "v_mfma_f32_16x16x32_bf16 " D ", %[A], %[B], " C "\n\t" FIVE(D)
        if constexpr (LOADS == 1) {
            asm volatile("ds_read_b128 a[16:19], %[a0]\n\t"
                         "ds_read_b128 a[20:23], %[a1]\n\t"
                         "ds_read_b128 a[24:27], %[a2]\n\t"
                         "ds_read_b128 a[28:31], %[a3]\n\t"
                         "ds_read_b128 %[w0], %[a0] offset:256\n\t"
                         "ds_read_b128 %[w1], %[a1] offset:512\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_mfma_f32_16x16x32_bf16 a[16:19], %[A], %[B], a[16:19]\n\t"
                         FIVE("a[16:19]")
                         "ds_read_b128 %[w2], %[a2] offset:768\n\t"
                         "s_nop 7\n\t"
                         "v_accvgpr_read_b32 %[g3], a19\n\tv_accvgpr_read_b32 %[g2], a18\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[18:21]", "a[20:23]")
                         "ds_read_b128 %[w0], %[a3] offset:1024\n\t"
                         "s_nop 6\n\t"
                         "v_accvgpr_read_b32 %[g7], a21\n\tv_accvgpr_read_b32 %[g6], a20\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[20:23]", "a[24:27]")
                         "ds_read_b128 %[w1], %[a0] offset:1280\n\t"
                         "s_nop 6\n\t"
                         "v_accvgpr_read_b32 %[g11], a23\n\tv_accvgpr_read_b32 %[g10], a22\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[22:25]", "a[28:31]")
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\t"
                         "v_accvgpr_read_b32 %[g5], a19\n\tv_accvgpr_read_b32 %[g4], a18\n\t"
                         "v_accvgpr_read_b32 %[g1], a17\n\tv_accvgpr_read_b32 %[g0], a16\n\t"
                         "v_accvgpr_read_b32 %[g13], a23\n\tv_accvgpr_read_b32 %[g12], a22\n\t"
                         "v_accvgpr_read_b32 %[g9], a21\n\tv_accvgpr_read_b32 %[g8], a20\n\t"
                         "v_accvgpr_read_b32 %[g15], a25\n\tv_accvgpr_read_b32 %[g14], a24\n\ts_nop 1"
                         : [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [g2] "=&v"(g[2]), [g3] "=&v"(g[3]), [g4] "=&v"(g[4]), [g5] "=&v"(g[5]),
                           [g6] "=&v"(g[6]), [g7] "=&v"(g[7]), [g8] "=&v"(g[8]), [g9] "=&v"(g[9]), [g10] "=&v"(g[10]),
                           [g11] "=&v"(g[11]), [g12] "=&v"(g[12]), [g13] "=&v"(g[13]), [g14] "=&v"(g[14]), [g15] "=&v"(g[15]),
                           [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2)
                         : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [A] "v"(A), [B] "v"(B)
                         : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29",
                           "a30", "a31", "memory");
        } else {
            asm volatile("v_accvgpr_write_b32 a16, %[c00]\n\tv_accvgpr_write_b32 a17, %[c01]\n\tv_accvgpr_write_b32 a18, %[c02]\n\tv_accvgpr_write_b32 a19, %[c03]\n\t"
                         "v_accvgpr_write_b32 a20, %[c10]\n\tv_accvgpr_write_b32 a21, %[c11]\n\tv_accvgpr_write_b32 a22, %[c12]\n\tv_accvgpr_write_b32 a23, %[c13]\n\t"
                         "v_accvgpr_write_b32 a24, %[c20]\n\tv_accvgpr_write_b32 a25, %[c21]\n\tv_accvgpr_write_b32 a26, %[c22]\n\tv_accvgpr_write_b32 a27, %[c23]\n\t"
                         "v_accvgpr_write_b32 a28, %[c30]\n\tv_accvgpr_write_b32 a29, %[c31]\n\tv_accvgpr_write_b32 a30, %[c32]\n\tv_accvgpr_write_b32 a31, %[c33]\n\t"
                         "ds_read_b128 %[w0], %[a0] offset:256\n\t"
                         "ds_read_b128 %[w1], %[a1] offset:512\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         "v_mfma_f32_16x16x32_bf16 a[16:19], %[A], %[B], a[16:19]\n\t"
                         FIVE("a[16:19]")
                         "ds_read_b128 %[w2], %[a2] offset:768\n\t"
                         "s_nop 7\n\t"
                         "v_accvgpr_read_b32 %[g3], a19\n\tv_accvgpr_read_b32 %[g2], a18\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[18:21]", "a[20:23]")
                         "ds_read_b128 %[w0], %[a3] offset:1024\n\t"
                         "s_nop 6\n\t"
                         "v_accvgpr_read_b32 %[g7], a21\n\tv_accvgpr_read_b32 %[g6], a20\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[20:23]", "a[24:27]")
                         "ds_read_b128 %[w1], %[a0] offset:1280\n\t"
                         "s_nop 6\n\t"
                         "v_accvgpr_read_b32 %[g11], a23\n\tv_accvgpr_read_b32 %[g10], a22\n\t"
                         "s_waitcnt lgkmcnt(1)\n\t"
                         CHAIN("a[22:25]", "a[28:31]")
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_nop 15\n\t"
                         "v_accvgpr_read_b32 %[g5], a19\n\tv_accvgpr_read_b32 %[g4], a18\n\t"
                         "v_accvgpr_read_b32 %[g1], a17\n\tv_accvgpr_read_b32 %[g0], a16\n\t"
                         "v_accvgpr_read_b32 %[g13], a23\n\tv_accvgpr_read_b32 %[g12], a22\n\t"
                         "v_accvgpr_read_b32 %[g9], a21\n\tv_accvgpr_read_b32 %[g8], a20\n\t"
                         "v_accvgpr_read_b32 %[g15], a25\n\tv_accvgpr_read_b32 %[g14], a24\n\ts_nop 1"
                         : [g0] "=&v"(g[0]), [g1] "=&v"(g[1]), [g2] "=&v"(g[2]), [g3] "=&v"(g[3]), [g4] "=&v"(g[4]), [g5] "=&v"(g[5]),
                           [g6] "=&v"(g[6]), [g7] "=&v"(g[7]), [g8] "=&v"(g[8]), [g9] "=&v"(g[9]), [g10] "=&v"(g[10]),
                           [g11] "=&v"(g[11]), [g12] "=&v"(g[12]), [g13] "=&v"(g[13]), [g14] "=&v"(g[14]), [g15] "=&v"(g[15]),
                           [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2)
                         : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [A] "v"(A), [B] "v"(B),
                           [c00] "v"(c[0][0]), [c01] "v"(c[0][1]), [c02] "v"(c[0][2]), [c03] "v"(c[0][3]),
                           [c10] "v"(c[1][0]), [c11] "v"(c[1][1]), [c12] "v"(c[1][2]), [c13] "v"(c[1][3]),
                           [c20] "v"(c[2][0]), [c21] "v"(c[2][1]), [c22] "v"(c[2][2]), [c23] "v"(c[2][3]),
                           [c30] "v"(c[3][0]), [c31] "v"(c[3][1]), [c32] "v"(c[3][2]), [c33] "v"(c[3][3])
                         : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29",
                           "a30", "a31", "memory");
        }
        bool ok = true;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) ok = ok && same(f32x4{g[4 * ch], g[4 * ch + 1], g[4 * ch + 2], g[4 * ch + 3]}, want[ch]);
        if (!ok) ++nbad;
        if (it == 0 && blockIdx.x == 0 && threadIdx.x < 64 && dbg) {
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int r = 0; r < 4; ++r) { dbg[(threadIdx.x * 16 + 4 * ch + r) * 2] = g[4 * ch + r]; dbg[(threadIdx.x * 16 + 4 * ch + r) * 2 + 1] = want[ch][r]; }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}


typedef float f32x2_t __attribute__((ext_vector_type(2)));

// P4: the instruction the corrupting builds actually get wrong (tools/hazard_variants.py forensic): the second of two
// packed FMAs that build feature 0, "v_pk_fma_f32 D, A2, B, D op_sel:[0,1,0]", came out as if its LOW product were 0 in
// lanes 48-63.  Waves 0-3 of a 512-thread workgroup run that sequence on freshly loaded operands and check it against
// scalar FMAs; waves 4-7 (the second wave of the same SIMDs) run what the other wave of the SIMD may have been running
// in the kernel: NEIGH 0 nothing, 1 chains of bf16 MFMAs with AGPR accumulators, 2 the same with VGPR accumulators.
template <int NEIGH>
__global__ void __launch_bounds__(512)
k_p4(const float* __restrict__ init, const bf16x8* __restrict__ ab, unsigned long long* __restrict__ bad, int iters,
     const float* __restrict__ gtab, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long nbad = 0;
    if (wave >= 4) {
        if constexpr (NEIGH == 0) return;
        const bf16x8 A = ab[lane], B = ab[64 + lane];
        f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (NEIGH == 1)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7"
                                 : "+a"(acc[c]) : "v"(A), "v"(B));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7"
                                 : "+v"(acc[c]) : "v"(A), "v"(B));
            }
            if ((it & 63) == 63) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = acc[c] * 1e-3f;
            }
        }
        if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.678f) sink[0] = 1.0f;
        return;
    }
    unsigned slot = wave * 31 + blockIdx.x * 7;
    for (int it = 0; it < iters; ++it) {
        slot = (slot * 5 + 1) & 4095;
        const float* src = gtab + ((size_t)slot * 64 + lane) * 8;     // per lane: R00 R01 | R10 R11 | rel0 rel1 | y0 y1
        f32x2_t r0, r1, rel, y, d, z;
        float t;
        asm volatile("global_load_dwordx2 %0, %7, off\n\t"
                     "global_load_dwordx2 %2, %7, off offset:16\n\t"
                     "global_load_dwordx2 %3, %7, off offset:24\n\t"
                     "global_load_dwordx2 %1, %7, off offset:8\n\t"          // R10 R11 last, as in the kernel (the receiver's row, offset 32)
                     "s_waitcnt vmcnt(0)\n\t"
                     "v_pk_fma_f32 %4, %0, %2, 0 op_sel_hi:[1,0,0]\n\t"      // d = (R00 rel0, R01 rel0)
                     "v_pk_fma_f32 %5, %3, %0, 0 op_sel_hi:[0,1,0]\n\t"      // an unrelated packed FMA, as in the kernel
                     "v_mov_b32_e32 %6, %8\n\t"
                     "v_pk_fma_f32 %4, %1, %2, %4 op_sel:[0,1,0]\n\t"        // d += (R10 rel1, R11 rel1)
                     "s_nop 4"
                     : "=&v"(r0), "=&v"(r1), "=&v"(rel), "=&v"(y), "=&v"(d), "=&v"(z), "=&v"(t)
                     : "v"(src), "v"(lane) : "memory");
        const float w0 = __builtin_fmaf(r1[0], rel[1], r0[0] * rel[0]);
        const float w1 = __builtin_fmaf(r1[1], rel[1], r0[1] * rel[0]);
        // (the kernel's failure is a whole product missing, so a relative 1e-4 separates it from last-bit contraction noise)
        if (__builtin_fabsf(d[0] - w0) > 1e-4f * (__builtin_fabsf(w0) + 1.0f) ||
            __builtin_fabsf(d[1] - w1) > 1e-4f * (__builtin_fabsf(w1) + 1.0f)) {
            if (nbad == 0 && blockIdx.x == 0 && threadIdx.x == 0 && it < 4) {
                sink[1] = d[0]; sink[2] = w0; sink[3] = d[1]; sink[4] = w1; sink[5] = r0[0]; sink[6] = r0[1]; sink[7] = r1[0];
                sink[8] = r1[1]; sink[9] = rel[0]; sink[10] = rel[1];
            }
            ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

static float* g_dbg = nullptr;
template <typename K, typename... Extra>
static void run_t(int threads, const char* name, K kernel, const float* d_init, const bf16x8* d_ab, unsigned long long* d_bad,
                  int blocks, int iters, Extra... extra) {
    CHECK(hipMemset(d_bad, 0, 8));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), threads == 256 ? LDS_BYTES : 0, 0, d_init, d_ab, d_bad, iters, extra...);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    const double trials = (double)blocks * 4 * iters;
    printf("%-58s trials %.3e  mismatches %llu  (%.2e per trial)  %.1f ms\n", name, trials, bad, bad / trials, ms);
    fflush(stdout);
}

template <typename K, typename... Extra>
static void run(const char* name, K kernel, const float* d_init, const bf16x8* d_ab, unsigned long long* d_bad, int blocks, int iters,
                Extra... extra) {
    run_t(256, name, kernel, d_init, d_ab, d_bad, blocks, iters, extra...);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int blocks = argc > 2 ? atoi(argv[2]) : 512;
    std::vector<float> init(SLOTS * 64 * 4);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : init) v = rnd();
    std::vector<uint16_t> ab(2 * 64 * 8);
    for (auto& v : ab) { float f = rnd(); uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    float* d_init; bf16x8* d_ab; unsigned long long* d_bad;
    CHECK(hipMalloc(&d_init, init.size() * 4)); CHECK(hipMalloc(&d_ab, ab.size() * 2)); CHECK(hipMalloc(&d_bad, 8));
    CHECK(hipMemcpy(d_init, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_ab, ab.data(), ab.size() * 2, hipMemcpyHostToDevice));
    printf("agpr_hazard: %d blocks x 4 waves x %d iterations per pattern\n", blocks, iters);
    run("P1.0 LDS->AGPR, MFMA SrcC right after lgkmcnt(0)", k_p1<0>, d_init, d_ab, d_bad, blocks, iters);
    run("P1.1 LDS->AGPR, three younger loads in flight, lgkmcnt(3)", k_p1<1>, d_init, d_ab, d_bad, blocks, iters);
    run("P1.2 LDS->AGPR, lgkmcnt(0) + 8 idle states", k_p1<2>, d_init, d_ab, d_bad, blocks, iters);
    run("P2.0 dst a[2:5] / SrcC a[4:7]", k_p2<0>, d_init, d_ab, d_bad, blocks, iters);
    run("P2.1 dst a[2:5] / SrcC a[4:7], LDS loads returning", k_p2<1>, d_init, d_ab, d_bad, blocks, iters);
    run("P2.2 control: dst = SrcC = a[4:7]", k_p2<2>, d_init, d_ab, d_bad, blocks, iters);
    CHECK(hipMalloc(&g_dbg, 64 * 16 * 2 * 4));
    CHECK(hipMemset(g_dbg, 0, 64 * 16 * 2 * 4));
    run("P3.1 compiled sliding-tuple sequence, accumulators by LDS load", k_p3<1>, d_init, d_ab, d_bad, blocks, iters / 4, g_dbg);
    {
        std::vector<float> h(64 * 16 * 2);
        CHECK(hipMemcpy(h.data(), g_dbg, h.size() * 4, hipMemcpyDeviceToHost));
        int shown = 0;
        for (int l = 0; l < 64 && shown < 6; ++l)
            for (int k = 0; k < 16; ++k)
                if (h[(l * 16 + k) * 2] != h[(l * 16 + k) * 2 + 1] && shown < 6) {
                    printf("   lane %d chain %d reg %d: got %.6f want %.6f\n", l, k / 4, k % 4, h[(l * 16 + k) * 2], h[(l * 16 + k) * 2 + 1]);
                    ++shown;
                }
        int nb[16] = {0};
        for (int l = 0; l < 64; ++l) for (int k = 0; k < 16; ++k) nb[k] += h[(l * 16 + k) * 2] != h[(l * 16 + k) * 2 + 1];
        printf("   first trial of wave 0, wrong lanes per (chain, reg):");
        for (int k = 0; k < 16; ++k) printf(" %d", nb[k]);
        printf("\n");
    }
    run("P3.0 same sequence, accumulators by v_accvgpr_write", k_p3<0>, d_init, d_ab, d_bad, blocks, iters / 4, (float*)nullptr);
    {
        std::vector<float> g((size_t)4096 * 64 * 8);
        for (auto& v : g) v = rnd() * 8.0f;
        float *d_g, *d_sink;
        CHECK(hipMalloc(&d_g, g.size() * 4)); CHECK(hipMalloc(&d_sink, 64)); CHECK(hipMemset(d_sink, 0, 64));
        CHECK(hipMemcpy(d_g, g.data(), g.size() * 4, hipMemcpyHostToDevice));
        run_t(512, "P4.0 packed-FMA pair on loaded operands, second wave idle", k_p4<0>, d_init, d_ab, d_bad, blocks, iters, (const float*)d_g, d_sink);
        {
            float h[16];
            CHECK(hipMemcpy(h, d_sink, 64, hipMemcpyDeviceToHost));
            printf("   P4 first mismatch: d0 %.9g want %.9g | d1 %.9g want %.9g | R00 %.9g R01 %.9g R10 %.9g R11 %.9g rel %.9g %.9g\n",
                   h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10]);
        }
        run_t(512, "P4.1 ... second wave: bf16 MFMA chains, AGPR accumulators", k_p4<1>, d_init, d_ab, d_bad, blocks, iters, (const float*)d_g, d_sink);
        run_t(512, "P4.2 ... second wave: bf16 MFMA chains, VGPR accumulators", k_p4<2>, d_init, d_ab, d_bad, blocks, iters, (const float*)d_g, d_sink);
    }
    return 0;
}
