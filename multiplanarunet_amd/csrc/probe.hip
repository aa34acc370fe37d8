// Measured machine peaks for bench.py's roofline objects ("a measured MFMA micro-benchmark +
// stream-triad and quote both"). Measurement aids, not product kernels.
#include "kernels.h"

namespace mpu {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// every wave issues `iters` x 8 independent v_mfma_f32_32x32x16_bf16 (4 accumulators x 2): the matrix pipe's
// back-to-back rate, no memory traffic. FLOPs = waves * iters * 8 * 2*32*32*16.
__global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + (threadIdx.x & 3)); b[j] = (short)(0x3c00 + (blockIdx.x & 7)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;            // keeps the accumulators alive; never true
}

// a[i] = b[i] + s * c[i], 16 bytes per lane: 2 reads + 1 write per element
__global__ __launch_bounds__(256) void probe_triad_kernel(float4* __restrict__ a, const float4* __restrict__ b,
                                                          const float4* __restrict__ c, long n4, float s) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

// out[i] = x[3i] + x[3i+1] + x[3i+2]: every lane reads 12 contiguous bytes (the K = 3 gather width of the fused
// back-mapping), the wave 768 contiguous bytes: 12 n bytes read exactly once. Calibrates FETCH_SIZE for 12-byte accesses.
__global__ __launch_bounds__(256) void probe_gather12_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v[3];
        __builtin_memcpy(v, x + 3 * i, 12);
        out[i] = v[0] + v[1] + v[2];
    }
}

}  // namespace
}  // namespace mpu

using namespace mpu;

extern "C" {

// 12 n bytes read (each once, 12 per lane), 4 n written.
int mpu_probe_gather12(const float* d_x, float* d_out, int64_t n, void* stream) {
    MPU_REQUIRE(d_x && d_out && n > 0, "mpu_probe_gather12: bad argument");
    long blocks = (n + 255) / 256; if (blocks > 256L * 32) blocks = 256L * 32;
    probe_gather12_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(d_x, d_out, n);
    return launch_ok();
}

// Launches the MFMA probe on `blocks` workgroups of 4 waves; *flops = the bf16 FLOPs it executes.
int mpu_probe_mfma_bf16(int32_t blocks, int32_t iters, float* d_sink, double* flops, void* stream) {
    MPU_REQUIRE(blocks > 0 && iters > 0 && d_sink, "mpu_probe_mfma_bf16: bad argument");
    probe_mfma_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, d_sink);
    if (flops) *flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
    return launch_ok();
}

// a = b + s*c over n floats (n % 4 == 0); HBM bytes moved = 12 * n.
int mpu_probe_stream_triad(float* d_a, const float* d_b, const float* d_c, int64_t n, void* stream) {
    MPU_REQUIRE(d_a && d_b && d_c && n > 0 && n % 4 == 0, "mpu_probe_stream_triad: bad argument");
    long blocks = (n / 4 + 255) / 256; if (blocks > 256L * 32) blocks = 256L * 32;
    probe_triad_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((float4*)d_a, (const float4*)d_b,
                                                                                      (const float4*)d_c, n / 4, 1.5f);
    return launch_ok();
}

}  // extern "C"
