// Shared helpers for libmpunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
// gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even, NaN stays NaN)
typedef __bf16 mpu_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float mpu_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    const mpu_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mpu_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

// ---- split-bf16 products (round 6, dtype "bf16x3": f32 storage, three bf16 MFMAs per product) ------------------------------
// x = hi + lo + O(2^-17 |x|) with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi) (x - hi is exact in fp32);
// a * b ~ hi_a hi_b + hi_a lo_b + lo_a hi_b, the dropped lo_a lo_b <= 2^-18 |a b|: products good to ~2^-16 relative, accumulated
// in fp32 by the matrix pipe at three bf16 MFMAs (3 x 32 cycles) per sixteen k-values instead of eight exact-f32 MFMAs
// (8 x 64 cycles). The eight f32 of a lane are two 16-byte LDS chunks p, q; any k-order is fine as long as both operands
// of a product use the same one.
typedef short mpu_s16x8 __attribute__((ext_vector_type(8)));
typedef float mpu_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void x3_split(const uint4& p, const uint4& q, mpu_s16x8& hi, mpu_s16x8& lo) {
    const uint32_t x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = __uint_as_float(x[2 * e]), x1 = __uint_as_float(x[2 * e + 1]);
        h[e] = f32x2_to_bf16x2(x0, x1);
        const float h0 = __uint_as_float(h[e] << 16), h1 = __uint_as_float(h[e] & 0xffff0000u);
        l[e] = f32x2_to_bf16x2(x0 - h0, x1 - h1);
    }
    hi = __builtin_bit_cast(mpu_s16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(mpu_s16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// Weights are split ONCE, when the packed operand copies are refreshed (launch_x3_words): a packed 32-bit word then holds
// bf16 hi in its low and bf16 lo in its high half, and a fragment of eight of them is regrouped with eight byte permutes
// instead of the ~28 conversions of x3_split.
__device__ __forceinline__ uint32_t x3_word(float x) {
    const uint32_t h = f32x2_to_bf16x2(x, 0.f) & 0xffffu;
    const uint32_t l = f32x2_to_bf16x2(x - __uint_as_float(h << 16), 0.f) & 0xffffu;
    return h | (l << 16);
}
__device__ __forceinline__ void x3_unpack(const uint4& p, const uint4& q, mpu_s16x8& hi, mpu_s16x8& lo) {
    const uint32_t w[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x05040100u);      // low halves: (w1.lo << 16) | w0.lo
        l[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x07060302u);      // high halves
    }
    hi = __builtin_bit_cast(mpu_s16x8, make_uint4(h[0], h[1], h[2], h[3]));
    lo = __builtin_bit_cast(mpu_s16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// c += a * b from the split operands, smallest terms first
__device__ __forceinline__ void x3_mma(const mpu_s16x8& ahi, const mpu_s16x8& alo, const mpu_s16x8& bhi, const mpu_s16x8& blo,
                                       mpu_f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, c, 0, 0, 0);
}

// "done once" flags of per-kernel attributes (hipFuncSetAttribute is per DEVICE): one bit per device of this process, so a
// process that drives several GPUs sets the attribute on each of them. Two steps: first_use_on_device() only TESTS the
// bit, mark_used_on_device() sets it AFTER the attribute calls have succeeded -- a second host thread on the same device either
// sees the bit (attributes in place) or repeats the idempotent calls itself; it can never launch ahead of them.
inline unsigned long long device_bit() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 0;
    return 1ull << dev;
}
inline bool first_use_on_device(unsigned long long& mask) {
    const unsigned long long bit = device_bit();
    return bit == 0 || (__atomic_load_n(&mask, __ATOMIC_ACQUIRE) & bit) == 0;
}
inline void mark_used_on_device(unsigned long long& mask) {
    const unsigned long long bit = device_bit();
    if (bit) __atomic_fetch_or(&mask, bit, __ATOMIC_RELEASE);
}
// compute units of the CURRENT device (cached per device, not per process)
inline int device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (!n) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace mpu
