// Fusion-model training step (reference: mpunet/bin/train_fusion.py:327-362,
// mpunet/models/fusion_model.py:9-39, mpunet/evaluate/loss_functions.py:207-246).
//
//   p[n,:] = softmax_k( sum_v W[v,k] x[n,v,k] + b[k] )
//   loss   = mean_n ( 1 - mean_k( 2 w y p / (w (p + y) + 1e-6) ) ) + 1e-6 mean(W^2) + 1e-6 mean(b^2)
//
// with y the one-hot of the integer target. The reference's generalized Dice loss is evaluated PER POINT
// (its reduction dims are empty for [N,K] predictions), so its class weights 1/ref_vol ('simple'), 1/ref_vol^2
// ('square') and 1 ('uniform') all come out as 1: infinite weights (classes absent from the point) are replaced
// by the largest finite weight of the batch, which is 1. Only the true class c contributes:
//   L_n = 1 - (2/K) p_c / (1 + p_c + 1e-6),   dL_n/dz_k = -(2/K) (1 + 1e-6) / (1 + p_c + 1e-6)^2 * p_c (d_ck - p_k)
// Two deterministic stages: per-block partial sums of the gradient (fp32 in-thread, fp64 across threads), then
// one block combines them in a fixed order, adds the regulariser and applies Keras Adam to the V*K + K parameters.
#include "kernels.h"
#include "../../include/mpunet_hip.h"

namespace mpu {
namespace {

constexpr int FT_MAXV = 16;
constexpr int FT_MAX_BLOCKS = 256;

template <int K>
__global__ __launch_bounds__(256) void fusion_grad_kernel(const float* __restrict__ x, const uint8_t* __restrict__ y, long n,
                                                          int V, const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ partial /*[blk][V*K + K + 1]*/) {
    __shared__ float sW[FT_MAXV * K + K];
    for (int i = threadIdx.x; i < V * K + K; i += 256) sW[i] = i < V * K ? W[i] : b[i - V * K];
    __syncthreads();
    const int nacc = V * K + K + 1;
    float gW[FT_MAXV][K];                                       // dL/dW, fp32 per thread
    float gb[K];
    float ls = 0.f;
#pragma unroll
    for (int v = 0; v < FT_MAXV; ++v)
#pragma unroll
        for (int k = 0; k < K; ++k) gW[v][k] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) gb[k] = 0.f;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
        const float* xp = x + t * V * K;
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = 0.f;
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = z[k] + sW[v * K + k] * xp[v * K + k];
        float mx = -3.4e38f;
#pragma unroll
        for (int k = 0; k < K; ++k) { z[k] = z[k] + sW[V * K + k]; mx = fmaxf(mx, z[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - mx); s += z[k]; }
        const float inv = 1.f / s;
        const int c = y[t];
        float pc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { z[k] *= inv; if (k == c) pc = z[k]; }
        if (c < K) {                                            // a target outside [0,K) has an all-zero one-hot: L_n = 1
            const float den = 1.f + pc + 1e-6f;
            ls += 1.f - (2.f / K) * pc / den;
            const float dLdp = -(2.f / K) * (1.f + 1e-6f) / (den * den);
            float dz[K];
#pragma unroll
            for (int k = 0; k < K; ++k) dz[k] = dLdp * pc * ((k == c ? 1.f : 0.f) - z[k]);
#pragma unroll
            for (int v = 0; v < FT_MAXV; ++v)
                if (v < V) {
#pragma unroll
                    for (int k = 0; k < K; ++k) gW[v][k] += dz[k] * xp[v * K + k];
                }
#pragma unroll
            for (int k = 0; k < K; ++k) gb[k] += dz[k];
        } else {
            ls += 1.f;
        }
    }
    // block reduction: butterfly over the 64 lanes of each wave (fixed order), then the 4 waves through LDS
    auto wave_sum = [](float v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ float wsum[4][FT_MAXV * K + K + 1];
#pragma unroll
    for (int v = 0; v < FT_MAXV; ++v)
        if (v < V) {
#pragma unroll
            for (int k = 0; k < K; ++k) { const float t = wave_sum(gW[v][k]); if (lane == 0) wsum[wave][v * K + k] = t; }
        }
#pragma unroll
    for (int k = 0; k < K; ++k) { const float t = wave_sum(gb[k]); if (lane == 0) wsum[wave][V * K + k] = t; }
    { const float t = wave_sum(ls); if (lane == 0) wsum[wave][V * K + K] = t; }
    __syncthreads();
    for (int a = threadIdx.x; a < nacc; a += 256)
        partial[(long)blockIdx.x * nacc + a] =
            (float)((double)wsum[0][a] + (double)wsum[1][a] + (double)wsum[2][a] + (double)wsum[3][a]);
}

__global__ __launch_bounds__(256) void fusion_update_kernel(const float* __restrict__ partial, int nblk, int V, int K, long n,
                                                            float* W, float* b, float* m, float* v2, double alpha,
                                                            float b1, float b2, float eps, int apply, float* grads_out,
                                                            float* loss_out) {
    const int np = V * K + K, nacc = np + 1;
    // (one block.) The regulariser of the reported loss is read BEFORE any thread updates W / b (it used to race
    // with the updates of the same pass: a loss mixing pre- and post-update weights, ~1e-6, nondeterministic)
    double rw = 0.0, rb = 0.0;
    if (threadIdx.x == np % 256 && loss_out) {
        for (int i = 0; i < V * K; ++i) rw += (double)W[i] * (double)W[i];
        for (int i = 0; i < K; ++i) rb += (double)b[i] * (double)b[i];
    }
    __syncthreads();
    for (int a = threadIdx.x; a < nacc; a += 256) {
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += (double)partial[(long)k * nacc + a];
        if (a == np) {                                          // loss: data term + regulariser
            if (loss_out) *loss_out = (float)(s / (double)n + 1e-6 * rw / (V * K) + 1e-6 * rb / K);
            continue;
        }
        float* p = a < V * K ? &W[a] : &b[a - V * K];
        const double regd = a < V * K ? 2e-6 * (double)*p / (V * K) : 2e-6 * (double)*p / K;
        const float g = (float)(s / (double)n + regd);
        if (grads_out) grads_out[a] = g;
        if (apply) {                                            // Keras Adam (TF ApplyAdam form)
            const float mm = m[a] + (g - m[a]) * (1.f - b1);
            const float vv = v2[a] + (g * g - v2[a]) * (1.f - b2);
            m[a] = mm; v2[a] = vv;
            *p = *p - (mm * (float)alpha) / (sqrtf(vv) + eps);
        }
    }
}

// ---- data-parallel form: the step in two halves with the replica SUM between them -----------
// stage 2a: column sums of the block partials in fp64, fixed order -> sums[V*K + K + 1] (gradient SUMS of the data term and the
// summed per-point loss over THIS rank's points) and sums[nacc] = the rank's point count. The host all-reduces the nacc + 1 doubles.
__global__ __launch_bounds__(256) void fusion_sums_kernel(const float* __restrict__ partial, int nblk, int nacc, long n,
                                                          double* __restrict__ sums) {
    for (int a = threadIdx.x; a <= nacc; a += 256) {
        if (a == nacc) { sums[a] = (double)n; continue; }
        double s = 0.0;
        for (int k = 0; k < nblk; ++k) s += (double)partial[(long)k * nacc + a];
        sums[a] = s;
    }
}
// stage 2b: exactly fusion_update_kernel's arithmetic on the (all-reduced) sums; n = sums[nacc] = the points of all ranks
__global__ __launch_bounds__(256) void fusion_apply_kernel(const double* __restrict__ sums, int V, int K, float* W, float* b,
                                                           float* m, float* v2, double alpha, float b1, float b2, float eps,
                                                           int apply, float* grads_out, float* loss_out) {
    const int np = V * K + K, nacc = np + 1;
    const double n = sums[nacc];
    // the reported loss uses the weights BEFORE this step's update (thread `np` used to read W / b while the other
    // threads of the same pass were updating them): every thread that will report reads first, then the block synchronises
    double rw = 0.0, rb = 0.0;
    if (threadIdx.x == np % 256 && loss_out) {
        for (int i = 0; i < V * K; ++i) rw += (double)W[i] * (double)W[i];
        for (int i = 0; i < K; ++i) rb += (double)b[i] * (double)b[i];
    }
    __syncthreads();
    for (int a = threadIdx.x; a < nacc; a += 256) {
        const double s = sums[a];
        if (a == np) {
            if (loss_out) *loss_out = (float)(s / n + 1e-6 * rw / (V * K) + 1e-6 * rb / K);
            continue;
        }
        float* p = a < V * K ? &W[a] : &b[a - V * K];
        const double regd = a < V * K ? 2e-6 * (double)*p / (V * K) : 2e-6 * (double)*p / K;
        const float g = (float)(s / n + regd);
        if (grads_out) grads_out[a] = g;
        if (apply) {
            const float mm = m[a] + (g - m[a]) * (1.f - b1);
            const float vv = v2[a] + (g * g - v2[a]) * (1.f - b2);
            m[a] = mm; v2[a] = vv;
            *p = *p - (mm * (float)alpha) / (sqrtf(vv) + eps);
        }
    }
}

int launch_fusion_grad(const float* d_x, const uint8_t* d_y, long n, int V, int K, const float* d_W, const float* d_b,
                       float* d_workspace, long* blocks_out, hipStream_t st) {
    long blocks = (n + 255) / 256; if (blocks > FT_MAX_BLOCKS) blocks = FT_MAX_BLOCKS;
    *blocks_out = blocks;
    if (blocks == 0) return MPU_OK;                              // a rank without points in this batch
#define MPU_FT_CASE(KK) case KK: fusion_grad_kernel<KK><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_x, d_y, n, V, d_W, d_b, d_workspace); break;
    switch (K) {
        MPU_FT_CASE(1) MPU_FT_CASE(2) MPU_FT_CASE(3) MPU_FT_CASE(4) MPU_FT_CASE(5) MPU_FT_CASE(6) MPU_FT_CASE(7) MPU_FT_CASE(8)
        default: return fail(MPU_EINVAL, "%s", "mpu_fusion_train: bad class count");
    }
#undef MPU_FT_CASE
    return launch_ok();
}

}  // namespace
}  // namespace mpu

using namespace mpu;

extern "C" {

int64_t mpu_fusion_train_workspace_floats(int32_t n_views, int32_t n_classes) {
    return (int64_t)FT_MAX_BLOCKS * (n_views * n_classes + n_classes + 1);
}

int mpu_fusion_train_step(const float* d_x, const uint8_t* d_y, int64_t n, int32_t n_views, int32_t n_classes,
                          float* d_W, float* d_b, float* d_adam_m, float* d_adam_v, int64_t t, double lr,
                          double beta1, double beta2, double eps, float* d_workspace, float* d_grads_out,
                          float* d_loss_out, void* stream) {
    MPU_REQUIRE(d_x && d_y && d_W && d_b && d_workspace, "mpu_fusion_train_step: null argument");
    MPU_REQUIRE(n >= 1 && n_views >= 1 && n_views <= FT_MAXV && n_classes >= 1 && n_classes <= 8,
                "mpu_fusion_train_step: need n >= 1, 1 <= views <= 16, 1 <= classes <= 8");
    MPU_REQUIRE(t == 0 || (d_adam_m && d_adam_v && t >= 1), "mpu_fusion_train_step: Adam state missing");
    hipStream_t st = (hipStream_t)stream;
    long blocks = 0;
    int rc = launch_fusion_grad(d_x, d_y, (long)n, n_views, n_classes, d_W, d_b, d_workspace, &blocks, st);
    if (rc) return rc;
    const double alpha = t >= 1 ? lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t)) : 0.0;
    fusion_update_kernel<<<1, 256, 0, st>>>(d_workspace, (int)blocks, n_views, n_classes, (long)n, d_W, d_b, d_adam_m,
                                            d_adam_v, alpha, (float)beta1, (float)beta2, (float)eps, t >= 1 ? 1 : 0,
                                            d_grads_out, d_loss_out);
    return launch_ok();
}

int mpu_fusion_grad_sums(const float* d_x, const uint8_t* d_y, int64_t n, int32_t n_views, int32_t n_classes,
                         const float* d_W, const float* d_b, float* d_workspace, double* d_sums, void* stream) {
    MPU_REQUIRE(d_W && d_b && d_workspace && d_sums && (n == 0 || (d_x && d_y)), "mpu_fusion_grad_sums: null argument");
    MPU_REQUIRE(n >= 0 && n_views >= 1 && n_views <= FT_MAXV && n_classes >= 1 && n_classes <= 8,
                "mpu_fusion_grad_sums: need n >= 0, 1 <= views <= 16, 1 <= classes <= 8");
    hipStream_t st = (hipStream_t)stream;
    long blocks = 0;
    int rc = launch_fusion_grad(d_x, d_y, (long)n, n_views, n_classes, d_W, d_b, d_workspace, &blocks, st);
    if (rc) return rc;
    fusion_sums_kernel<<<1, 256, 0, st>>>(d_workspace, (int)blocks, n_views * n_classes + n_classes + 1, (long)n, d_sums);
    return launch_ok();
}

int mpu_fusion_apply_sums(const double* d_sums, int32_t n_views, int32_t n_classes, float* d_W, float* d_b,
                          float* d_adam_m, float* d_adam_v, int64_t t, double lr, double beta1, double beta2, double eps,
                          float* d_grads_out, float* d_loss_out, void* stream) {
    MPU_REQUIRE(d_sums && d_W && d_b, "mpu_fusion_apply_sums: null argument");
    MPU_REQUIRE(n_views >= 1 && n_views <= FT_MAXV && n_classes >= 1 && n_classes <= 8, "mpu_fusion_apply_sums: bad shape");
    MPU_REQUIRE(t == 0 || (d_adam_m && d_adam_v && t >= 1), "mpu_fusion_apply_sums: Adam state missing");
    const double alpha = t >= 1 ? lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t)) : 0.0;
    fusion_apply_kernel<<<1, 256, 0, (hipStream_t)stream>>>(d_sums, n_views, n_classes, d_W, d_b, d_adam_m, d_adam_v, alpha,
                                                            (float)beta1, (float)beta2, (float)eps, t >= 1 ? 1 : 0, d_grads_out,
                                                            d_loss_out);
    return launch_ok();
}

}  // extern "C"
