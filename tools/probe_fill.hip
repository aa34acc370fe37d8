// L2/MALL/HBM -> LDS fill-rate probe (dev tool, not product): how many GB/s can `buffer_load_dwordx4 ... lds`
// deliver per CU as a function of workgroups per CU, bytes in flight and the footprint of the source?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
// each wave issues PIECES 1-KiB DMAs per stage; NST stages in flight; ITERS stages per workgroup
template <int PIECES, int NST>
__global__ __launch_bounds__(256) void fill_kernel(const void* src, long bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long pa = (unsigned long long)src;
    i32x4 rs; rs.x = (int)(unsigned)pa; rs.y = (int)((unsigned)(pa >> 32) & 0xffffu); rs.z = (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes); rs.w = 0x00020000;
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const long stage_bytes = 4L * PIECES * 1024;
    const long nstage_src = bytes / stage_bytes;
    long s = ((long)blockIdx.x * 7919) % nstage_src;
    auto issue = [&](int st) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const unsigned off = (unsigned)(s * stage_bytes + (wave * PIECES + p) * 1024 + lane * 16);
            dma16(rs, off, lds0 + st * (unsigned)stage_bytes + (wave * PIECES + p) * 1024);
        }
        s += gridDim.x; if (s >= nstage_src) s -= nstage_src;
    };
    for (int i = 0; i < NST - 1; ++i) issue(i);
    for (int it = 0; it < iters; ++it) {
        issue((it + NST - 1) % NST);
        if (NST == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PIECES) : "memory");
        else if (NST == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[blockIdx.x & 63];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int PIECES, int NST>
int run(const void* src, long bytes, int wg_per_cu, unsigned* sink, const char* what) {
    auto k = fill_kernel<PIECES, NST>;
    const int smem = NST * 4 * PIECES * 1024;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    const int grid = 256 * wg_per_cu, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<<<grid, 256, smem>>>(src, bytes, 50, sink);
    CK(hipEventRecord(e0));
    k<<<grid, 256, smem>>>(src, bytes, iters, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double total = (double)grid * iters * 4.0 * PIECES * 1024;
    printf("%-6s src %6ld MB  %d WG/CU  stage %2d KB x %d stages (%3d KB in flight/WG): %7.1f GB/s per CU, %6.2f TB/s total\n",
           what, bytes >> 20, wg_per_cu, 4 * PIECES, NST, 4 * PIECES * (NST - 1), total / ms / 1e6 / 256, total / ms / 1e9);
    return 0;
}
int main() {
    unsigned* sink; CK(hipMalloc(&sink, 4096 * 4));
    long sizes[3] = {2L << 20, 64L << 20, 1L << 30};
    const char* names[3] = {"L2", "MALL", "HBM"};
    for (int z = 0; z < 3; ++z) {
        void* src; CK(hipMalloc(&src, sizes[z])); CK(hipMemset(src, 1, sizes[z]));
        for (int w = 1; w <= 4; w *= 2) {
            if (run<4, 2>(src, sizes[z], w, sink, names[z])) return 1;       // 16 KB stages, 16 KB in flight
            if (run<8, 2>(src, sizes[z], w, sink, names[z])) return 1;       // 32 KB stages, 32 KB in flight
            if (w <= 2 && run<8, 3>(src, sizes[z], w, sink, names[z])) return 1;   // 64 KB in flight
            if (w == 1 && run<8, 4>(src, sizes[z], w, sink, names[z])) return 1;   // 96 KB in flight
        }
        CK(hipFree(src));
    }
    return 0;
}
