// NHWC implicit-GEMM convolution kernels for gfx950 MFMA.
//
//   conv_igemm_kernel : forward conv and data-gradient (same kernel, different
//                       packed weights). D[n][m] = sum_k W[n][k] * X[m][k] with
//                       n = output channel (MFMA rows), m = output pixel (MFMA
//                       cols), k = (tap, input channel). Modes:
//                         CONV3   3x3 SAME stride 1               (unet.py:120-179)
//                         UPCONV2 UpSampling2D(2) + 2x2 SAME conv (unet.py:159-163;
//                                 TF SAME for k=2 pads 0 top/left, 1 bottom/right)
//                         CONV3S2 3x3 stride 2 pad 1: the data-gradient of UPCONV2
//                                 with tap-combined weights
//                         CONV1   1x1
//   wgrad_igemm_kernel: weight gradient dW[tap][ci][co] = sum_m X[m@tap][ci] dZ[m][co]
//                       split over the pixel dimension, deterministic second stage.
//
// Tiling: 256 threads = 4 waves (64 lanes each); K rows of 128 B (64 bf16 / 32 f32)
// staged global -> VGPR -> LDS (rows padded to 144 B: conflict-free ds_read_b128),
// double buffered, one barrier per K step, next tile's global loads in flight
// under the MFMAs. bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32
// (exact f32). The k-order inside a row is permuted identically for both
// operands, which leaves the dot product unchanged.
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<CONV3>   { static constexpr int NTAPS = 9, KW = 3; };
template <> struct ModeTraits<UPCONV2> { static constexpr int NTAPS = 4, KW = 2; };
template <> struct ModeTraits<CONV3S2> { static constexpr int NTAPS = 9, KW = 3; };
template <> struct ModeTraits<CONV1>   { static constexpr int NTAPS = 1, KW = 1; };

// input pixel (iy,ix) read by output pixel (oy,ox) at tap (ky,kx); false = zero padding
template <int MODE>
__device__ __forceinline__ bool tap_src(int oy, int ox, int ky, int kx, int Ho, int Wo, int& iy, int& ix) {
    if (MODE == CONV3) {
        iy = oy + ky - 1; ix = ox + kx - 1;
        return (unsigned)iy < (unsigned)Ho && (unsigned)ix < (unsigned)Wo;
    } else if (MODE == UPCONV2) {
        const int uy = oy + ky, ux = ox + kx;
        iy = uy >> 1; ix = ux >> 1;
        return uy < Ho && ux < Wo;
    } else if (MODE == CONV3S2) {
        iy = 2 * oy + ky - 1; ix = 2 * ox + kx - 1;
        return (unsigned)iy < (unsigned)(2 * Ho) && (unsigned)ix < (unsigned)(2 * Wo);
    } else {
        iy = oy; ix = ox;
        return true;
    }
}
template <int MODE> __device__ __forceinline__ int in_h(int Ho) {
    return MODE == UPCONV2 ? Ho / 2 : (MODE == CONV3S2 ? Ho * 2 : Ho);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// XCD-aware bijective remap: consecutive logical tiles land on one XCD (one L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


template <typename T, int MODE, int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    constexpr int EPC = 16 / sizeof(T);          // elements per 16-B chunk
    constexpr int BKE = 128 / sizeof(T);         // elements per K row
    constexpr int LROW = 144;
    constexpr int NW_ROWS = BN / 32, NP_ROWS = BM / 32;
    constexpr int TN = WN / 32, TM = WM / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NTAPS = ModeTraits<MODE>::NTAPS, KW = ModeTraits<MODE>::KW;
    constexpr int STAGE = (BN + BM) * LROW;
    static_assert((BN / WN) * (BM / WM) == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (logical % tiles_n) * BN;
    const long m0 = (long)(logical / tiles_n) * BM;
    const int Cin = a.C0 + a.C1;
    const int nchunks = (Cin + BKE - 1) / BKE;
    const int nit = NTAPS * nchunks;
    const int Hi = in_h<MODE>(a.Ho), Wi = in_h<MODE>(a.Wo);
    const long M = (long)a.B * a.Ho * a.Wo;
    const T* in0 = (const T*)a.in0; const T* in1 = (const T*)a.in1; const T* wp = (const T*)a.w;

    const int ck = tid & 7, r0 = tid >> 3;
    int pb[NP_ROWS], py[NP_ROWS], px[NP_ROWS];
#pragma unroll
    for (int i = 0; i < NP_ROWS; ++i) {
        const long m = m0 + r0 + 32 * i;
        if (m < M) {
            const int ox = (int)(m % a.Wo); const long t = m / a.Wo;
            const int oy = (int)(t % a.Ho); const int b = (int)(t / a.Ho);
            pb[i] = b * Hi * Wi; py[i] = oy; px[i] = ox;
        } else { pb[i] = -1; py[i] = 0; px[i] = 0; }
    }

    uint4 wreg[NW_ROWS], preg[NP_ROWS];
    auto gload = [&](int tap, int cc) {
        const int ch = cc * BKE + ck * EPC;
        const bool chv = ch < Cin;
#pragma unroll
        for (int i = 0; i < NW_ROWS; ++i) {
            const int n = n0 + r0 + 32 * i;
            wreg[i] = make_uint4(0, 0, 0, 0);
            if (chv && n < a.Cout)
                wreg[i] = *(const uint4*)(wp + (long)tap * a.w_tap_stride + (long)n * a.w_row_stride + ch);
        }
        const T* src; int cs, Cs;
        if (ch < a.C0) { src = in0; cs = ch; Cs = a.C0; } else { src = in1; cs = ch - a.C0; Cs = a.C1; }
        const int ky = tap / KW, kx = tap % KW;
#pragma unroll
        for (int i = 0; i < NP_ROWS; ++i) {
            int iy, ix;
            const bool v = tap_src<MODE>(py[i], px[i], ky, kx, a.Ho, a.Wo, iy, ix);
            preg[i] = make_uint4(0, 0, 0, 0);
            if (v && chv && pb[i] >= 0)
                preg[i] = *(const uint4*)(src + ((long)pb[i] + (long)iy * Wi + ix) * Cs + cs);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* base = smem + buf * STAGE + r0 * LROW + ck * 16;
#pragma unroll
        for (int i = 0; i < NW_ROWS; ++i) *(uint4*)(base + i * 32 * LROW) = wreg[i];
#pragma unroll
        for (int i = 0; i < NP_ROWS; ++i) *(uint4*)(base + BN * LROW + i * 32 * LROW) = preg[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const unsigned char* Wb = smem + buf * STAGE + (wn * WN + (lane & 31)) * LROW + (lane >> 5) * 16;
        const unsigned char* Pb = smem + buf * STAGE + BN * LROW + (wm * WM + (lane & 31)) * LROW + (lane >> 5) * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) af[i] = *(const uint4*)(Wb + i * 32 * LROW + s * 32);
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[j] = *(const uint4*)(Pb + j * 32 * LROW + s * 32);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) Mma<T>::run(af[i], bf[j], acc[i][j]);
        }
    };

    int tap = 0, cc = 0;
    gload(0, 0);
    lstore(0);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nit;
        if (more) {
            if (++cc == nchunks) { cc = 0; ++tap; }
            gload(tap, cc);
        }
        compute(buf);
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane holds pixel m = col, 4 consecutive channels per register quad
    T* out = (T*)a.out; const T* mask = (const T*)a.mask;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const long m = m0 + wm * WM + j * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wn * WN + i * 32 + 8 * q + 4 * (lane >> 5);
                if (n >= a.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][4 * q + e];
                    if (a.bias) v[e] += a.bias[n + e];
                    if (a.relu) v[e] = fmaxf(v[e], 0.f);
                }
                const long o = m * a.Cout + n;
                if (sizeof(T) == 2) {
                    if (mask) {
                        const uint2 mk = *(const uint2*)((const bf16_t*)mask + o);
                        const bf16_t mm[4] = {(bf16_t)(mk.x & 0xffff), (bf16_t)(mk.x >> 16),
                                              (bf16_t)(mk.y & 0xffff), (bf16_t)(mk.y >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!(bf16_to_f32(mm[e]) > 0.f)) v[e] = 0.f;
                    }
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                    *(uint2*)((bf16_t*)out + o) = pk;
                } else {
                    if (mask) {
                        const float4 mk = *(const float4*)((const float*)mask + o);
                        if (!(mk.x > 0.f)) v[0] = 0.f;
                        if (!(mk.y > 0.f)) v[1] = 0.f;
                        if (!(mk.z > 0.f)) v[2] = 0.f;
                        if (!(mk.w > 0.f)) v[3] = 0.f;
                    }
                    *(float4*)((float*)out + o) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------- //
// weight gradient
// ------------------------------------------------------------------------- //

// output tile BCI x BCO per workgroup, 4 waves as 2x2, K step = 32 pixels.
template <typename T, int MODE, int BCI, int BCO>
__global__ __launch_bounds__(256) void wgrad_igemm_kernel(WgradArgs a) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int KP = 32;                                  // pixels per K step
    constexpr int RSX = BCI * sizeof(T) + 64, RSZ = BCO * sizeof(T) + 64;
    constexpr int CPRX = BCI / EPC, CPRZ = BCO / EPC;      // chunks per row
    constexpr int NLX = KP * CPRX / 256, NLZ = KP * CPRZ / 256;
    constexpr int WCI = BCI / 2, WCO = BCO / 2, TI = WCI / 32, TJ = WCO / 32;
    constexpr int KW = ModeTraits<MODE>::KW;
    constexpr int STAGE = KP * (RSX + RSZ);
    static_assert(NLX >= 1 && NLZ >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int Cin = a.C0 + a.C1;
    const int tiles_co = (a.Cout + BCO - 1) / BCO;
    const int ci0 = (blockIdx.x / tiles_co) * BCI, co0 = (blockIdx.x % tiles_co) * BCO;
    const int tap = blockIdx.y, ky = tap / KW, kx = tap % KW;
    const long M = (long)a.B * a.Ho * a.Wo;
    const long mbeg = (long)blockIdx.z * a.mchunk;
    const long mend = (mbeg + a.mchunk < M) ? mbeg + a.mchunk : M;
    const int Hi = in_h<MODE>(a.Ho), Wi = in_h<MODE>(a.Wo);
    const T* x0 = (const T*)a.x0; const T* x1 = (const T*)a.x1; const T* dz = (const T*)a.dz;

    uint4 xreg[NLX], zreg[NLZ];
    auto gload = [&](long mb) {
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int q = tid + 256 * i, row = q / CPRX, c = q % CPRX;
            const long m = mb + row;
            const int ch = ci0 + c * EPC;
            xreg[i] = make_uint4(0, 0, 0, 0);
            if (m < mend && ch < Cin) {
                const int ox = (int)(m % a.Wo); const long t = m / a.Wo;
                const int oy = (int)(t % a.Ho); const int b = (int)(t / a.Ho);
                int iy, ix;
                if (tap_src<MODE>(oy, ox, ky, kx, a.Ho, a.Wo, iy, ix)) {
                    const T* src; int cs, Cs;
                    if (ch < a.C0) { src = x0; cs = ch; Cs = a.C0; } else { src = x1; cs = ch - a.C0; Cs = a.C1; }
                    xreg[i] = *(const uint4*)(src + (((long)b * Hi + iy) * Wi + ix) * Cs + cs);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NLZ; ++i) {
            const int q = tid + 256 * i, row = q / CPRZ, c = q % CPRZ;
            const long m = mb + row;
            const int ch = co0 + c * EPC;
            zreg[i] = make_uint4(0, 0, 0, 0);
            if (m < mend && ch < a.Cout) zreg[i] = *(const uint4*)(dz + m * a.Cout + ch);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* xb = smem + buf * STAGE;
        unsigned char* zb = xb + KP * RSX;
#pragma unroll
        for (int i = 0; i < NLX; ++i) {
            const int q = tid + 256 * i;
            *(uint4*)(xb + (q / CPRX) * RSX + (q % CPRX) * 16) = xreg[i];
        }
#pragma unroll
        for (int i = 0; i < NLZ; ++i) {
            const int q = tid + 256 * i;
            *(uint4*)(zb + (q / CPRZ) * RSZ + (q % CPRZ) * 16) = zreg[i];
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const unsigned char* xb = smem + buf * STAGE;
        const unsigned char* zb = xb + KP * RSX;
        if constexpr (sizeof(T) == 2) {
            // transposed fragment reads: lane gets 4 consecutive pixels (k) of one channel
            const int krow = 8 * (lane >> 5) + ((lane & 15) >> 2);
            const int ccol = 16 * ((lane >> 4) & 1) + (lane & 3) * 4;
#pragma unroll
            for (int s = 0; s < KP / 16; ++s) {
                s16x8 af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const unsigned char* p = xb + (s * 16 + krow) * RSX + (wi * WCI + i * 32 + ccol) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * RSX));
                    af[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const unsigned char* p = zb + (s * 16 + krow) * RSZ + (wj * WCO + j * 32 + ccol) * 2;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * RSZ));
                    bf[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < KP; k += 2) {
                float af[TI], bf[TJ];
                const int row = k + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TI; ++i)
                    af[i] = *(const float*)(xb + row * RSX + (wi * WCI + i * 32 + (lane & 31)) * 4);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    bf[j] = *(const float*)(zb + row * RSZ + (wj * WCO + j * 32 + (lane & 31)) * 4);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nit = (int)((mend - mbeg + KP - 1) / KP);
    if (nit > 0) {
        gload(mbeg);
        lstore(0);
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const bool more = it + 1 < nit;
            if (more) gload(mbeg + (long)(it + 1) * KP);
            compute(it & 1);
            if (more) lstore((it & 1) ^ 1);
            __syncthreads();
        }
    }
    // D[row = ci][col = co]: lane col = lane&31, rows (r&3)+8*(r>>2)+4*(lane>>5)
    float* P = a.partial + ((long)blockIdx.z * gridDim.y + tap) * (long)Cin * a.Cout;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int co = co0 + wj * WCO + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wi * WCI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (ci < Cin && co < a.Cout) P[(long)ci * a.Cout + co] = acc[i][j][r];
            }
        }
}

// second stage: dW[e] = sum_s partial[s][e]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit,
                                                           long n, float* __restrict__ dW) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < ksplit; ++k) s += partial[(long)k * n + e];
        dW[e] = s;
    }
}

// ------------------------------------------------------------------------- //
// host-side launchers (internal C++ API used by unet.hip and the op-level ABI)
// ------------------------------------------------------------------------- //
template <typename T, int MODE, int BN, int BM, int WN, int WM>
static int launch_conv_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int SMEM = 2 * (BN + BM) * 144;
    auto kern = conv_igemm_kernel<T, MODE, BN, BM, WN, WM>;
    static bool attr_set = false;
    if (!attr_set) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        attr_set = true;
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long tiles = (long)cdiv(a.Cout, BN) * cdiv(M, BM);
    if (prof_on()) {
        const int taps = MODE == UPCONV2 ? 4 : (MODE == CONV1 ? 1 : 9);
        prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * taps * (a.C0 + a.C1), st);
    }
    kern<<<dim3((unsigned)tiles), dim3(256), SMEM, st>>>(a);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

template <typename T, int MODE>
static int launch_conv_mode(const ConvArgs& a, hipStream_t st) {
    const long M = (long)a.B * a.Ho * a.Wo;
    const long t128 = (long)cdiv(a.Cout, 128) * cdiv(M, 128);
    const long t64x128 = (long)cdiv(a.Cout, 64) * cdiv(M, 128);
    if (a.Cout > 64 && t128 >= 384) return launch_conv_cfg<T, MODE, 128, 128, 64, 64>(a, st);
    if (t64x128 >= 384 || a.Cout <= 64) {
        if (M >= 128 * 64) return launch_conv_cfg<T, MODE, 64, 128, 64, 32>(a, st);
    }
    return launch_conv_cfg<T, MODE, 64, 64, 32, 32>(a, st);
}

int launch_conv(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
#define MPU_CONV_CASE(TT)                                                          \
    switch (mode) {                                                                \
        case CONV3: return launch_conv_mode<TT, CONV3>(a, st);                     \
        case UPCONV2: return launch_conv_mode<TT, UPCONV2>(a, st);                 \
        case CONV3S2: return launch_conv_mode<TT, CONV3S2>(a, st);                 \
        case CONV1: return launch_conv_mode<TT, CONV1>(a, st);                     \
        default: return fail(MPU_EINVAL, "%s", "conv: bad mode");                  \
    }
    if (dtype == MPU_BF16) { MPU_CONV_CASE(bf16_t) }
    if (dtype == MPU_F32) { MPU_CONV_CASE(float) }
#undef MPU_CONV_CASE
    return fail(MPU_EINVAL, "%s", "conv: bad dtype");
}

// workspace (floats) the split-K partials of a wgrad call need
long wgrad_partial_elems(int mode, int Cin, int Cout, long M, int* ksplit_out, int* mchunk_out) {
    const int ntaps = mode == UPCONV2 ? 4 : (mode == CONV1 ? 1 : 9);
    const int bc = (Cin >= 128 && Cout >= 128) ? 128 : 64;
    const long tiles = (long)cdiv(Cin, bc) * cdiv(Cout, bc) * ntaps;
    long ks = (1024 + tiles - 1) / tiles;                 // aim at ~1024 workgroups
    const long maxks = (M + 255) / 256;                   // at least 256 pixels per split
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    long mchunk = ((M + ks - 1) / ks + 31) / 32 * 32;
    ks = (M + mchunk - 1) / mchunk;
    if (ksplit_out) *ksplit_out = (int)ks;
    if (mchunk_out) *mchunk_out = (int)mchunk;
    return ks * ntaps * (long)Cin * Cout;
}

template <typename T, int MODE>
static int launch_wgrad_mode(WgradArgs a, float* dW, hipStream_t st) {
    const int Cin = a.C0 + a.C1;
    const int ntaps = ModeTraits<MODE>::NTAPS;
    const long n = (long)ntaps * Cin * a.Cout;
    if (prof_on())
        prof_begin(PROF_WGRAD, a.flops > 0 ? a.flops : 2.0 * a.B * a.Ho * a.Wo * (double)n, st);
    bool big = false;
    if constexpr (sizeof(T) == 2) {
        if (Cin >= 128 && a.Cout >= 128) {
            big = true;
            dim3 g((unsigned)(cdiv(Cin, 128) * cdiv(a.Cout, 128)), ntaps, a.ksplit);
            wgrad_igemm_kernel<T, MODE, 128, 128><<<g, dim3(256), 0, st>>>(a);
        }
    }
    if (!big) {
        dim3 g((unsigned)(cdiv(Cin, 64) * cdiv(a.Cout, 64)), ntaps, a.ksplit);
        wgrad_igemm_kernel<T, MODE, 64, 64><<<g, dim3(256), 0, st>>>(a);
    }
    if (prof_on()) prof_end(st);
    int rc = launch_ok();
    if (rc) return rc;
    long blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    wgrad_reduce_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(a.partial, a.ksplit, n, dW);
    return launch_ok();
}

int launch_wgrad(int dtype, int mode, const WgradArgs& a, float* dW, hipStream_t st) {
#define MPU_WG_CASE(TT)                                                            \
    switch (mode) {                                                                \
        case CONV3: return launch_wgrad_mode<TT, CONV3>(a, dW, st);                \
        case UPCONV2: return launch_wgrad_mode<TT, UPCONV2>(a, dW, st);            \
        case CONV1: return launch_wgrad_mode<TT, CONV1>(a, dW, st);                \
        default: return fail(MPU_EINVAL, "%s", "wgrad: bad mode");                 \
    }
    if (dtype == MPU_BF16) { MPU_WG_CASE(bf16_t) }
    if (dtype == MPU_F32) { MPU_WG_CASE(float) }
#undef MPU_WG_CASE
    return fail(MPU_EINVAL, "%s", "wgrad: bad dtype");
}

}  // namespace mpu
