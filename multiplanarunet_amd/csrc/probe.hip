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

// The same loop with RANDOM operands (round 4): four A and four B fragments of pseudo-random bf16 values (random sign and
// mantissa, exponents 2^-2 .. 2^1) rotate through the MFMAs, so consecutive instructions present different bits to the
// multiplier arrays -- as a convolution on real activations does. The constant-operand probe above barely toggles the
// datapath and holds ~2.2 GHz; under random operands the part is POWER-limited and clocks far lower (clock sampler:
// mpu_probe_clock), which is the ceiling a real MFMA kernel works under.
__global__ __launch_bounds__(256) void probe_mfma_random_kernel(int iters, float* sink) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a[4], b[4];
    unsigned h = (unsigned)(threadIdx.x * 2654435761u) ^ (unsigned)(blockIdx.x * 40503u + 12345u);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            a[k][j] = (short)(((h >> 16) & 0x807f) | (0x3e80 + ((h >> 8) & 0x3) * 0x80));
            h = h * 1664525u + 1013904223u;
            b[k][j] = (short)(((h >> 16) & 0x807f) | (0x3e80 + ((h >> 8) & 0x3) * 0x80));
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i + 2 * u + 1) & 3], acc[i], 0, 0, 0);
        if ((it & 63) == 63) {                   // keep the sums finite: damp the accumulators now and then
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 0.0009765625f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) sink[0] = s;
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

// dst = src, 16 bytes per lane, UNROLL float4s per thread with all loads in flight before the first store: the "float4 copy"
// the guide quotes at 6.29 TB/s (MI355X_MICROARCH.md, chip table). VARIANT 0: default cache policy, one shot (a workgroup per
// 256 * UNROLL float4s); 1: non-temporal loads and stores (streamed once: do not keep the lines); 2: default policy, grid-stride
// over 256 * 32 workgroups (the triad probe's shape). Settles whether this pool's boxes reach that rate.
// (n4 is a multiple of 256 * UNROLL: no tail guards -- a guarded store drags its load under the branch and serialises the
// four round trips, the "serialised loads" of DESIGN section 5.)
template <int VARIANT, int UNROLL>
__global__ __launch_bounds__(256) void probe_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, long n4) {
    const long stride = VARIANT == 2 ? (long)gridDim.x * blockDim.x * UNROLL : n4;
    for (long base = ((long)blockIdx.x * UNROLL) * blockDim.x + threadIdx.x; base < n4; base += stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float4* p = src + base + (long)u * blockDim.x;
            if (VARIANT == 1) {
                v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
                v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
            } else v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            float4* q = dst + base + (long)u * blockDim.x;
            if (VARIANT == 1) {
                __builtin_nontemporal_store(v[u].x, &q->x); __builtin_nontemporal_store(v[u].y, &q->y);
                __builtin_nontemporal_store(v[u].z, &q->z); __builtin_nontemporal_store(v[u].w, &q->w);
            } else *q = v[u];
        }
    }
}

// Read-once in PERMUTED runs (round 5): the n floats are cut into runs of RUN contiguous bytes (a power of two, 128 ... 4096) and
// the runs are visited in a scattered order (run r of the walk = source run (r * odd constant) mod nruns, a bijection for a
// power-of-two run count). Inside a run the lanes read contiguous 16-byte pieces, so every request is as coalesced as a copy's;
// only the ORDER in which DRAM pages are opened differs from a stream. Calibrates what the gather kernels (fused back-mapping:
// 128-byte lines of six prediction volumes in brick order) can expect from HBM: the stream figure is not their bound.
// A thread sums 8 pieces and writes one float: 3 % write traffic.
template <int UNROLL>
__global__ __launch_bounds__(256) void probe_permuted_read_kernel(const float4* __restrict__ src, float* __restrict__ out, long n4,
                                                                  int run_shift4 /* log2(run bytes / 16) */, long run_mask) {
    const long base = ((long)blockIdx.x * UNROLL) * 256 + threadIdx.x;
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const long e = base + (long)u * 256;
        const long r = e >> run_shift4, w = e & ((1L << run_shift4) - 1);
        const long rp = (r * 0x9E3779B1L) & run_mask;
        v[u] = src[(rp << run_shift4) + w];
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
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

// One wave samples the shader clock while other kernels run: (s_memtime, s_memrealtime) pairs `naps` x s_sleep 127 apart
// (s_memrealtime ticks at a constant 100 MHz; s_memtime counts shader cycles). Round 4: under the dense MFMA + LDS
// kernels this part runs well below its 2.4 GHz maximum (1.5 GHz inside conv_halo16), so a cycle count is not a time.
__global__ __launch_bounds__(64) void probe_clock_kernel(unsigned long long* out, int n, int naps) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        out[2 * i] = __builtin_amdgcn_s_memtime();
        out[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(127);
    }
}

}  // namespace
}  // namespace mpu

using namespace mpu;

extern "C" {

// n (shader cycles, 100-MHz ticks) pairs into d_samples [2 n] u64, about naps x 8 k cycles apart; launch it on a SIDE stream
// next to the work whose clock is wanted.
int mpu_probe_clock(uint64_t* d_samples, int32_t n, int32_t naps, void* stream) {
    MPU_REQUIRE(d_samples && n > 0 && naps >= 0, "mpu_probe_clock: bad argument");
    probe_clock_kernel<<<dim3(1), dim3(64), 0, (hipStream_t)stream>>>((unsigned long long*)d_samples, n, naps);
    return launch_ok();
}

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

// ... with pseudo-random bf16 operands (see probe_mfma_random_kernel): the power-limited ceiling of real data
int mpu_probe_mfma_bf16_random(int32_t blocks, int32_t iters, float* d_sink, double* flops, void* stream) {
    MPU_REQUIRE(blocks > 0 && iters > 0 && d_sink, "mpu_probe_mfma_bf16_random: bad argument");
    probe_mfma_random_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(iters, d_sink);
    if (flops) *flops = (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
    return launch_ok();
}

// dst = src over n floats (n % 4096 == 0); HBM bytes moved = 8 * n. variant 0: default policy, one shot; 1: non-temporal; 2: default
// policy, grid-stride.
int mpu_probe_stream_copy(float* d_dst, const float* d_src, int64_t n, int32_t variant, void* stream) {
    constexpr int U = 4;
    MPU_REQUIRE(d_dst && d_src && n > 0 && n % (4 * 256 * U) == 0 && variant >= 0 && variant <= 2,
                "mpu_probe_stream_copy: bad argument (n must be a multiple of 4096)");
    const long n4 = n / 4;
    const long blocks = n4 / (256L * U);
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) probe_copy_kernel<0, U><<<dim3((unsigned)blocks), dim3(256), 0, st>>>((float4*)d_dst, (const float4*)d_src, n4);
    else if (variant == 1) probe_copy_kernel<1, U><<<dim3((unsigned)blocks), dim3(256), 0, st>>>((float4*)d_dst, (const float4*)d_src, n4);
    else probe_copy_kernel<2, U><<<dim3((unsigned)(blocks < 256L * 32 ? blocks : 256L * 32)), dim3(256), 0, st>>>((float4*)d_dst, (const float4*)d_src, n4);
    return launch_ok();
}

// Sum of n floats read once in permuted runs of run_bytes (128 ... 4096, a power of two; run_bytes >= 4 n: one run = a stream);
// n a power of two >= 2^13; d_out holds n / 32 floats. HBM bytes read = 4 n.
int mpu_probe_permuted_read(const float* d_src, float* d_out, int64_t n, int32_t run_bytes, void* stream) {
    MPU_REQUIRE(d_src && d_out && n >= 8192 && (n & (n - 1)) == 0, "mpu_probe_permuted_read: n must be a power of two >= 8192");
    MPU_REQUIRE(run_bytes >= 16 && (run_bytes & (run_bytes - 1)) == 0, "mpu_probe_permuted_read: run_bytes must be a power of two >= 16");
    constexpr int U = 8;
    const long n4 = n / 4;
    int sh = 0; while ((16L << sh) < run_bytes && (1L << sh) < n4) ++sh;
    const long nruns = n4 >> sh;
    probe_permuted_read_kernel<U><<<dim3((unsigned)(n4 / (256L * U))), dim3(256), 0, (hipStream_t)stream>>>(
        (const float4*)d_src, d_out, n4, sh, nruns - 1);
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
