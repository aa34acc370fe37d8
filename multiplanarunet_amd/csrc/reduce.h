// Second stage of the deterministic two-stage column reductions (BN statistics, BN backward sums,
// bias gradients): partial[k][stat][col] summed over k in a fixed order, in double.
#pragma once
#include <hip/hip_runtime.h>

namespace mpu {

// block = FIN_COLS columns x FIN_KL k-lanes (256 threads)
constexpr int FIN_COLS = 4, FIN_KL = 64;

// out[s] (valid on the k-lane-0 thread of each column) = sum_k partial[k*stride + s*stat_stride + col].
// All loads of a round (16 per statistic) are issued before the first add: one memory round trip per
// 1024 partial rows.
template <int NS>
__device__ __forceinline__ void partial_sums(const float* __restrict__ partial, int nblk, long stride, long stat_stride,
                                             int col, bool valid, double* red /*[256]*/, double* out /*[NS]*/) {
    const int kl = threadIdx.x / FIN_COLS;
    double s[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = 0.0;
    if (valid) {
        for (int base = 0; base < nblk; base += 16 * FIN_KL) {
            float v[NS][16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = base + kl + u * FIN_KL;
#pragma unroll
                for (int i = 0; i < NS; ++i) {       // clamped, unconditional load: a predicated one is waited for on the spot
                    const float x = partial[(long)(k < nblk ? k : nblk - 1) * stride + i * stat_stride + col];
                    v[i][u] = k < nblk ? x : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) s[i] += (double)v[i][u];
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        red[threadIdx.x] = s[i];
        __syncthreads();
        double t = 0.0;
        if (kl == 0)
            for (int j = 0; j < FIN_KL; ++j) t += red[j * FIN_COLS + (threadIdx.x % FIN_COLS)];
        out[i] = t;
        __syncthreads();
    }
}

// Same for partials stored column-major, partial[(s*C + col)*nblk + k] (the conv epilogues' BN statistics): thread
// (col = tid / FIN_KL, k-lane = tid % FIN_KL) reads 64 consecutive floats per wave. out[] valid on k-lane 0.
template <int NS>
__device__ __forceinline__ void partial_sums_colmajor(const float* __restrict__ partial, int nblk, int C, int col,
                                                      bool valid, double* red /*[256]*/, double* out /*[NS]*/) {
    const int kl = threadIdx.x % FIN_KL, cl = threadIdx.x / FIN_KL;
    double s[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = 0.0;
    if (valid) {
        for (int base = 0; base < nblk; base += 16 * FIN_KL) {
            float v[NS][16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = base + kl + u * FIN_KL;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float x = partial[((long)i * C + col) * nblk + (k < nblk ? k : nblk - 1)];
                    v[i][u] = k < nblk ? x : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) s[i] += (double)v[i][u];
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        red[threadIdx.x] = s[i];
        __syncthreads();
        double t = 0.0;
        if (kl == 0)
            for (int j = 0; j < FIN_KL; ++j) t += red[cl * FIN_KL + j];
        out[i] = t;
        __syncthreads();
    }
}

// bias gradient: out[c] = sum_k partial[k][c]; one block of this body handles FIN_COLS columns
__device__ __forceinline__ void colsum_finalize_block(int blk, const float* __restrict__ partial, int nblk, int C,
                                                      float* __restrict__ out, double* red) {
    const int c = blk * FIN_COLS + (threadIdx.x % FIN_COLS);
    double s;
    partial_sums<1>(partial, nblk, C, 0, c, c < C, red, &s);
    if (c < C && threadIdx.x < FIN_COLS) out[c] = (float)s;
}

}  // namespace mpu
