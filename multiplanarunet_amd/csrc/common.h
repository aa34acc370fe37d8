// Shared helpers for libmpunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mpunet_hip.h"

namespace mpu {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, const char* a = "", long b = 0, long c = 0) {
    snprintf(g_err, sizeof(g_err), fmt, a, b, c);
    return code;
}

#define MPU_CHECK_HIP(expr)                                                       \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return mpu::fail(MPU_EHIP, "%s: HIP error %ld", hipGetErrorString(e_), (long)e_); \
    } while (0)

#define MPU_REQUIRE(cond, msg)                                                    \
    do { if (!(cond)) return mpu::fail(MPU_EINVAL, "%s", msg); } while (0)

inline int launch_ok() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MPU_EHIP, "%s: launch failed (%ld)", hipGetErrorString(e), (long)e);
    return MPU_OK;
}

// bf16 <-> f32 (round-to-nearest-even, as torch.bfloat16)
typedef uint16_t bf16_t;
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {      // branch-free RNE; NaN stays NaN
    const uint32_t u = __float_as_uint(f);
    const uint32_t r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    const uint32_t nan = (u >> 16) | 0x40u;
    return (bf16_t)(((u & 0x7fffffffu) > 0x7f800000u) ? nan : r);
}
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace mpu
