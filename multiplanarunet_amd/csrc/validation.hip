// Epoch-end validation counting (mpunet/callbacks/validation.py:115-125): argmax of the class scores fused with
// the per-class TP / relevant / selected counts. HBM-bound (4K + 1 bytes per pixel read once, nothing written but
// 3K integers); integer work, so the result does not depend on the order of the adds: per-thread register
// counters -> wave shuffle reduction -> LDS -> one 64-bit atomic per counter and workgroup.
#include "kernels.h"

namespace mpu {
namespace {

constexpr int VC_MAXK = 16;

template <int K>
__global__ __launch_bounds__(256) void validation_count_kernel(const float* __restrict__ pred, const uint8_t* __restrict__ y,
                                                               long n, unsigned long long* __restrict__ counts) {
    unsigned tp[K], rel[K], sel[K];
#pragma unroll
    for (int c = 0; c < K; ++c) { tp[c] = 0; rel[c] = 0; sel[c] = 0; }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* p = pred + i * K;
        float v[K];
#pragma unroll
        for (int c = 0; c < K; ++c) v[c] = p[c];
        // np.argmax: first maximum; a NaN counts as the maximum (first NaN wins)
        int best = 0; float bv = v[0];
#pragma unroll
        for (int c = 1; c < K; ++c) {
            const bool take = (v[c] > bv) || (v[c] != v[c] && bv == bv);
            bv = take ? v[c] : bv; best = take ? c : best;
        }
        const int t = y[i];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            rel[c] += (t == c); sel[c] += (best == c); tp[c] += (t == c && best == c);
        }
    }
    __shared__ unsigned long long sh[3 * K];
    if (threadIdx.x < 3 * K) sh[threadIdx.x] = 0ull;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K; ++c) {
        unsigned a = tp[c], b = rel[c], d = sel[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); d += __shfl_down(d, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sh[c], (unsigned long long)a);
            atomicAdd(&sh[K + c], (unsigned long long)b);
            atomicAdd(&sh[2 * K + c], (unsigned long long)d);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * K && sh[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sh[threadIdx.x]);
}

template <int K>
int launch_vc(const float* pred, const uint8_t* y, long n, unsigned long long* counts, hipStream_t st) {
    // per-thread 32-bit counters: a thread sees at most n / (grid * 256) + 1 pixels
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    validation_count_kernel<K><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(pred, y, n, counts);
    return launch_ok();
}

}  // namespace
}  // namespace mpu

using namespace mpu;

extern "C" int mpu_validation_count(const float* d_pred, const uint8_t* d_y, int64_t n, int32_t n_classes,
                                    int64_t* d_counts, void* stream) {
    MPU_REQUIRE(d_pred && d_y && d_counts, "mpu_validation_count: null argument");
    MPU_REQUIRE(n >= 0 && n_classes >= 1 && n_classes <= VC_MAXK, "mpu_validation_count: need 1 <= n_classes <= 16");
    if (n == 0) return MPU_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* c = (unsigned long long*)d_counts;
    switch (n_classes) {
#define VC_CASE(KK) case KK: return launch_vc<KK>(d_pred, d_y, (long)n, c, st);
        VC_CASE(1) VC_CASE(2) VC_CASE(3) VC_CASE(4) VC_CASE(5) VC_CASE(6) VC_CASE(7) VC_CASE(8)
        VC_CASE(9) VC_CASE(10) VC_CASE(11) VC_CASE(12) VC_CASE(13) VC_CASE(14) VC_CASE(15) VC_CASE(16)
#undef VC_CASE
    }
    return fail(MPU_EINVAL, "%s", "mpu_validation_count: bad n_classes");
}
