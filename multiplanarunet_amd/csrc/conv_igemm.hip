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
#include <stdlib.h>
#include "kernels.h"
#include "reduce.h"

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
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
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
    // K steps: for every tap, the 128-byte channel rows of source 0, then those of source 1
    // (a row never straddles the two concat sources, so the source is wave-uniform)
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    const int nit = NTAPS * nchunks;
    const int Hi = in_h<MODE>(a.Ho), Wi = in_h<MODE>(a.Wo);
    const long M = (long)a.B * a.Ho * a.Wo;
    constexpr unsigned OOB = 0xfffffff0u;           // buffer loads beyond num_records return 0
    const long npix = (long)a.B * Hi * Wi;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.in0, 0, (int)(npix * a.C0 * (long)sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.in1 ? a.in1 : a.in0), 0, (int)(a.in1 ? npix * a.C1 * (long)sizeof(T) : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.w, 0, (int)(a.w_elems * (long)sizeof(T)), 0x00020000);

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
    unsigned wrow[NW_ROWS];                          // byte offset of this thread's weight rows
#pragma unroll
    for (int i = 0; i < NW_ROWS; ++i) {
        const int n = n0 + r0 + 32 * i;
        wrow[i] = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * (long)sizeof(T)) : OOB;
    }

    auto gload = [&](uint4 (&wr)[NW_ROWS], uint4 (&pr)[NP_ROWS], int tap, int cc) {
        const bool s1 = cc >= nch0;
        const int cbase = (s1 ? cc - nch0 : cc) * BKE;
        const int Cs = s1 ? a.C1 : a.C0;
        const int ch = cbase + ck * EPC;
        const bool chv = ch < Cs;
        const unsigned wk = (unsigned)(((long)tap * a.w_tap_stride + (s1 ? a.C0 : 0) + ch) * (long)sizeof(T));
#pragma unroll
        for (int i = 0; i < NW_ROWS; ++i) {
            const unsigned off = (chv && wrow[i] != OOB) ? wrow[i] + wk : OOB;
            wr[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
        }
        const int ky = tap / KW, kx = tap % KW;
#pragma unroll
        for (int i = 0; i < NP_ROWS; ++i) {
            int iy, ix;
            const bool v = tap_src<MODE>(py[i], px[i], ky, kx, a.Ho, a.Wo, iy, ix) && chv && pb[i] >= 0;
            const unsigned off = v ? (unsigned)(((pb[i] + iy * Wi + ix) * Cs + ch) * (int)sizeof(T)) : OOB;
            pr[i] = s1 ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0))
                       : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs0, off, 0, 0));
        }
    };
    auto lstore = [&](const uint4 (&wr)[NW_ROWS], const uint4 (&pr)[NP_ROWS], int buf) {
        unsigned char* base = smem + buf * STAGE + r0 * LROW + ck * 16;
#pragma unroll
        for (int i = 0; i < NW_ROWS; ++i) *(uint4*)(base + i * 32 * LROW) = wr[i];
#pragma unroll
        for (int i = 0; i < NP_ROWS; ++i) *(uint4*)(base + BN * LROW + i * 32 * LROW) = pr[i];
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

    // Software pipeline, prefetch distance 2: tile t is consumed from LDS[t&1] while tile t+1
    // waits in one register set and the loads of tile t+2 are issued into the other.
    uint4 wA[NW_ROWS], pA[NP_ROWS], wB[NW_ROWS], pB[NP_ROWS];
    int tapN = 0, ccN = 0;
    auto advance = [&]() { if (++ccN == nchunks) { ccN = 0; ++tapN; } };
    gload(wA, pA, tapN, ccN); advance();
    if (nit > 1) { gload(wB, pB, tapN, ccN); advance(); }
    lstore(wA, pA, 0);
    __syncthreads();
    for (int it = 0; it < nit; it += 2) {
        if (it + 2 < nit) { gload(wA, pA, tapN, ccN); advance(); }
        compute(0);
        if (it + 1 < nit) lstore(wB, pB, 1);
        __syncthreads();
        if (it + 1 >= nit) break;
        if (it + 3 < nit) { gload(wB, pB, tapN, ccN); advance(); }
        compute(1);
        if (it + 2 < nit) lstore(wA, pA, 0);
        __syncthreads();
    }

    // epilogue. The last loop iteration ended with a barrier, so the staging LDS is free:
    //   1. bias of this tile -> LDS; 2. +bias, ReLU, convert, write the [BM][BN] tile to LDS
    //   (row = pixel); 3. 16-byte coalesced row stores (+ coalesced ReLU-mask loads).
    constexpr int OROW = BN * (int)sizeof(T) + 16;          // padded output row
    static_assert(BM * OROW + 3 * BN * 4 <= 2 * STAGE, "epilogue tile must fit the staging LDS");
    float* sbias = (float*)(smem + BM * OROW);
    if (tid < BN) {
        const bool nv = n0 + tid < a.Cout;
        sbias[tid] = (a.bias && nv) ? a.bias[n0 + tid] : 0.f;
        sbias[BN + tid] = (a.post_scale && nv) ? a.post_scale[n0 + tid] : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && nv) ? a.post_shift[n0 + tid] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int ml = wm * WM + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * WN + i * 32 + 8 * q + 4 * (lane >> 5);
                const float4 bq = *(const float4*)(sbias + nl);
                float v[4] = {acc[i][j][4 * q] + bq.x, acc[i][j][4 * q + 1] + bq.y,
                              acc[i][j][4 * q + 2] + bq.z, acc[i][j][4 * q + 3] + bq.w};
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (a.post_scale) {
                    const float4 sq = *(const float4*)(sbias + BN + nl), hq = *(const float4*)(sbias + 2 * BN + nl);
                    v[0] = v[0] * sq.x + hq.x; v[1] = v[1] * sq.y + hq.y;
                    v[2] = v[2] * sq.z + hq.z; v[3] = v[3] * sq.w + hq.w;
                }
                unsigned char* dst = smem + ml * OROW + nl * (int)sizeof(T);
                if (sizeof(T) == 2) {
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(v[0], v[1]);
                    pk.y = f32x2_to_bf16x2(v[2], v[3]);
                    *(uint2*)dst = pk;
                } else {
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
    __syncthreads();
    {
        constexpr int CPRO = BN * (int)sizeof(T) / 16;       // 16-byte chunks per output row
        T* out = (T*)a.out; const T* mask = (const T*)a.mask;
        for (int idx = tid; idx < BM * CPRO; idx += 256) {
            const int row = idx / CPRO, c = idx % CPRO;
            const long m = m0 + row;
            const int n = n0 + c * EPC;
            if (m >= M || n >= a.Cout) continue;
            uint4 val = *(const uint4*)(smem + row * OROW + c * 16);
            const long o = m * a.Cout + n;
            if (mask) {
                const uint4 mk = *(const uint4*)(mask + o);
                if (sizeof(T) == 2) {
                    // bf16 > 0  <=>  sign bit clear and magnitude non-zero
                    auto keep = [](uint32_t mw, uint32_t vw) {
                        const uint32_t lo = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                        const uint32_t hi = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                        return vw & (lo | hi);
                    };
                    val.x = keep(mk.x, val.x); val.y = keep(mk.y, val.y);
                    val.z = keep(mk.z, val.z); val.w = keep(mk.w, val.w);
                } else {
                    if (!(__uint_as_float(mk.x) > 0.f)) val.x = 0;
                    if (!(__uint_as_float(mk.y) > 0.f)) val.y = 0;
                    if (!(__uint_as_float(mk.z) > 0.f)) val.z = 0;
                    if (!(__uint_as_float(mk.w) > 0.f)) val.w = 0;
                }
            }
            *(uint4*)(out + o) = val;
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

// second stage: dW[e] = sum_s partial[s][e]  (fixed order: deterministic); 16-B loads, 4 splits in flight
// Blocks past `main_blocks` finish the fused bias gradient (db = sum of the wgrad kernel's column-sum partials).
struct DbFin { const float* partial; float* db; int nshare, C, main_blocks; int il4_cout = 0; int cout = 0, cin_job = 0, cin_total = 0, ci_base = 0; };

// Write the sum of one 16-byte partial column.
//  * il4_cout == 0: the partials are [tap][ci][co] like dW: float4 e holds four consecutive co.
//  * il4_cout = Cout (wgrad_taps): the partials are [tap][ci / 4][co][4 ci] -- a lane of that kernel holds four consecutive
//    input channels of one output channel, so its accumulator leaves as ONE 16-byte store; e = (tap * Cin/4 + ci/4) * Cout
//    + co, the four values belong to four consecutive rows of dW. Consecutive threads write consecutive co.
//  * cin_job > 0 (one source of a concat layer as its own job): the job's taps hold cin_job rows each, which are rows
//    [ci_base, ci_base + cin_job) of dW's cin_total-row taps.
__device__ __forceinline__ void store_dw_sum(float* __restrict__ dW, long e, const float4& s, int il4_cout, int cout, int cin_job,
                                             int cin_total, int ci_base) {
    if (il4_cout) {
        const unsigned row = (unsigned)e / (unsigned)il4_cout, co = (unsigned)e - row * (unsigned)il4_cout;   // row = tap * cin_job/4 + ci/4
        long drow = (long)row * 4;
        if (cin_job) { const unsigned c4 = (unsigned)cin_job >> 2, tap = row / c4, cib = row - tap * c4; drow = (long)tap * cin_total + ci_base + 4L * cib; }
        float* d = dW + drow * il4_cout + co;
        d[0] = s.x; d[il4_cout] = s.y; d[2 * (long)il4_cout] = s.z; d[3 * (long)il4_cout] = s.w;
        return;
    }
    if (!cin_job) { reinterpret_cast<float4*>(dW)[e] = s; return; }
    const unsigned el = (unsigned)e * 4u, row = el / (unsigned)cout, co = el - row * (unsigned)cout;         // row = tap * cin_job + ci
    const unsigned tap = row / (unsigned)cin_job, ci = row - tap * (unsigned)cin_job;
    *reinterpret_cast<float4*>(dW + ((long)tap * cin_total + ci_base + ci) * cout + co) = s;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int ksplit, long n,
                                                           float* __restrict__ dW, DbFin f) {
    if ((int)blockIdx.x >= f.main_blocks) {
        __shared__ double dred[256];
        colsum_finalize_block((int)blockIdx.x - f.main_blocks, f.partial, f.nshare, f.C, f.db, dred);
        return;
    }
    const long n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)f.main_blocks * 256) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= ksplit; k += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[(long)(k + u) * n4 + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        if (k < ksplit) {       // tail in ONE round trip: clamped unconditional loads, the surplus ones add +0 (a select,
            float4 v[3];        // not a branch: the compiler sinks loads into branches and waits for each on the spot)
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = p4[(long)(k + u < ksplit ? k + u : k) * n4 + e];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool on = k + u < ksplit;
                s.x += on ? v[u].x : 0.f; s.y += on ? v[u].y : 0.f; s.z += on ? v[u].z : 0.f; s.w += on ? v[u].w : 0.f;
            }
        }
        store_dw_sum(dW, e, s, f.il4_cout, f.cout, f.cin_job, f.cin_total, f.ci_base);
    }
}

// many splits of a small weight tensor (full-resolution layers): 4 k-lanes per 16-B column, fixed-order LDS combine
__global__ __launch_bounds__(256) void wgrad_reduce_kl4_kernel(const float* __restrict__ partial, int ksplit, long n,
                                                               float* __restrict__ dW, DbFin f) {
    if ((int)blockIdx.x >= f.main_blocks) {
        __shared__ double dred[256];
        colsum_finalize_block((int)blockIdx.x - f.main_blocks, f.partial, f.nshare, f.C, f.db, dred);
        return;
    }
    __shared__ float4 red[256];
    const long n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(partial);
    const int col = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
        int k = kl;
        for (; k + 12 < ksplit; k += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[(long)(k + 4 * u) * n4 + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        if (k < ksplit) {       // tail in one round trip (clamped loads, surplus ones add +0 through a select)
            float4 v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = p4[(long)(k + 4 * u < ksplit ? k + 4 * u : k) * n4 + e];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool on = k + 4 * u < ksplit;
                s.x += on ? v[u].x : 0.f; s.y += on ? v[u].y : 0.f; s.z += on ? v[u].z : 0.f; s.w += on ? v[u].w : 0.f;
            }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && e < n4) {
#pragma unroll
        for (int j = 1; j < 4; ++j) { const float4 v = red[j * 64 + col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        store_dw_sum(dW, e, s, f.il4_cout, f.cout, f.cin_job, f.cin_total, f.ci_base);
    }
}

// every recorded reduction in one launch: block -> job by the table's block ranges; bodies as above
struct ReduceTable { int njobs, _pad; ReduceJob job[REDUCE_MAX_JOBS]; };
__global__ __launch_bounds__(256) void wgrad_reduce_all_kernel(ReduceTable t) {
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < t.njobs; ++k) if ((int)blockIdx.x >= t.job[k].blk_begin) j = k;
    const ReduceJob& q = t.job[j];
    const int b = (int)blockIdx.x - q.blk_begin;
    if (b >= q.main_blocks) {
        __shared__ double dred[256];
        colsum_finalize_block(b - q.main_blocks, q.db_partial, q.nshare, q.C, q.db, dred);
        return;
    }
    const long n4 = q.n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(q.partial);
    const int ksplit = q.ksplit;
    if (!q.kl4) {
        for (long e = (long)b * 256 + threadIdx.x; e < n4; e += (long)q.main_blocks * 256) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= ksplit; k += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = p4[(long)(k + u) * n4 + e];
#pragma unroll
                for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            if (k < ksplit) {
                float4 v[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) v[u] = p4[(long)(k + u < ksplit ? k + u : k) * n4 + e];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const bool on = k + u < ksplit;
                    s.x += on ? v[u].x : 0.f; s.y += on ? v[u].y : 0.f; s.z += on ? v[u].z : 0.f; s.w += on ? v[u].w : 0.f;
                }
            }
            store_dw_sum(q.dW, e, s, q.il4_cout, q.cout, q.cin_job, q.cin_total, q.ci_base);
        }
        return;
    }
    __shared__ float4 red[256];
    const int col = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const long e = (long)b * 64 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
        int k = kl;
        for (; k + 12 < ksplit; k += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[(long)(k + 4 * u) * n4 + e];
#pragma unroll
            for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        if (k < ksplit) {
            float4 v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = p4[(long)(k + 4 * u < ksplit ? k + 4 * u : k) * n4 + e];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool on = k + 4 * u < ksplit;
                s.x += on ? v[u].x : 0.f; s.y += on ? v[u].y : 0.f; s.z += on ? v[u].z : 0.f; s.w += on ? v[u].w : 0.f;
            }
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && e < n4) {
#pragma unroll
        for (int jj = 1; jj < 4; ++jj) { const float4 v = red[jj * 64 + col]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        store_dw_sum(q.dW, e, s, q.il4_cout, q.cout, q.cin_job, q.cin_total, q.ci_base);
    }
}

// which: WG_TAPS = the jobs whose partials come from wgrad_taps (ci-interleaved layout), WG_GLDS = all others; the jobs
// not taken stay in the queue (renumbered), so two flushes with complementary masks run every job exactly once
int flush_wgrad_reduces(ReduceQueue& q, hipStream_t st, int which) {
    if (q.njobs == 0) return MPU_OK;
    ReduceTable t; t.njobs = 0; t._pad = 0;
    int nblocks = 0, kept = 0, kept_blocks = 0;
    for (int k = 0; k < q.njobs; ++k) {
        const bool is_taps = q.job[k].il4_cout > 0;
        if (which & (is_taps ? WG_TAPS : WG_GLDS)) {
            ReduceJob& j = t.job[t.njobs++];
            j = q.job[k]; j.blk_begin = nblocks; nblocks += j.main_blocks + j.db_blocks;
        } else {
            ReduceJob j = q.job[k];
            j.blk_begin = kept_blocks; kept_blocks += j.main_blocks + j.db_blocks;
            q.job[kept++] = j;
        }
    }
    q.njobs = kept; q.nblocks = kept_blocks;
    if (t.njobs == 0 || nblocks == 0) return MPU_OK;
    launch_k(wgrad_reduce_all_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, t);
    return launch_ok();
}

// every recorded weight-gradient kernel: the wgrad_taps jobs as one launch, the wgrad_glds jobs as another
int flush_wgrad_group(int dtype, WgradGroup& g, hipStream_t st, int which) {
    int rc = MPU_OK;
    if (g.ntaps && (which & WG_TAPS)) {
        if (prof_on()) prof_begin(PROF_WGRAD, g.taps_flops, st);
        rc = launch_wgrad_taps_group(g.taps, g.ntaps, st);
        if (prof_on()) prof_end(st);
        if (sched_log_on()) sched_note("wgrad-group taps jobs=%d", g.ntaps);
        g.ntaps = 0; g.taps_flops = 0;
    }
    if (!rc && g.nglds && (which & WG_GLDS)) {
        if (prof_on()) prof_begin(PROF_WGRAD, g.glds_flops, st);
        rc = launch_wgrad_glds_group(dtype, g.glds, g.nglds, st);
        if (prof_on()) prof_end(st);
        if (sched_log_on()) sched_note("wgrad-group glds jobs=%d", g.nglds);
        g.nglds = 0; g.glds_flops = 0;
    }
    return rc;
}

// MPU_CONV_IMPL=regs selects the register-staged kernel of this file; default is the LDS-DMA
// kernel of conv_glds.hip (same tiling, same results).
static int conv_impl() {
    const int impl = (int)env(ENV_CONV_IMPL);
    return impl;
}

// ------------------------------------------------------------------------- //
// host-side launchers (internal C++ API used by unet.hip and the op-level ABI)
// ------------------------------------------------------------------------- //
template <typename T, int MODE, int BN, int BM, int WN, int WM>
static int launch_conv_cfg(const ConvArgs& a_in, hipStream_t st) {
    constexpr int SMEM = 2 * (BN + BM) * 144;
    auto kern = conv_igemm_kernel<T, MODE, BN, BM, WN, WM>;
    ConvArgs a = a_in;
    if (a.w_elems <= 0) a.w_elems = (ModeTraits<MODE>::NTAPS - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long tiles = (long)cdiv(a.Cout, BN) * cdiv(M, BM);
    {   // 32-bit buffer offsets: every operand must stay below 2 GiB
        constexpr int NT = ModeTraits<MODE>::NTAPS;
        const long hi = MODE == UPCONV2 ? a.Ho / 2 : (MODE == CONV3S2 ? a.Ho * 2 : a.Ho);
        const long wi = MODE == UPCONV2 ? a.Wo / 2 : (MODE == CONV3S2 ? a.Wo * 2 : a.Wo);
        const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
        if ((long)a.B * hi * wi * cmax * (long)sizeof(T) >= (1L << 31) ||
            ((NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride) * (long)sizeof(T) >= (1L << 31))
            return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    }
    if (prof_on()) {
        const int taps = MODE == UPCONV2 ? 4 : (MODE == CONV1 ? 1 : 9);
        prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * taps * (a.C0 + a.C1), st);
    }
    launch_k(kern, dim3((unsigned)tiles), dim3(256), SMEM, st, a);
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

static int halo_on() {
    return (int)env(ENV_CONV_HALO);
}

static int launch_conv_impl(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    if (a.stats_rows) *a.stats_rows = 0;          // set by the schedules that produce the fused BN statistics
    if (a.x3 && (conv_impl() != 1 || dtype != MPU_F32 || mode == CONV1))   // (packed hi | lo words: the LDS-DMA f32 kernels only)
        return fail(MPU_EUNSUPPORTED, "%s", "conv: split-bf16 products (f32x3) need the LDS-DMA kernels, f32 tensors and a 3x3 / 2x2 layer");
    auto note = [&](const char* sched) {
        if (sched_log_on())
            sched_note("conv %s mode=%d B=%d H=%d W=%d Cin=%d Cout=%d dgrad=%d pool=%d head=%d", sched, mode, a.B, a.Ho, a.Wo,
                       a.C0 + a.C1, a.Cout, a.bias ? 0 : 1, (a.pooled_done && *a.pooled_done) ? 1 : 0,
                       (a.head_done && *a.head_done) ? 1 : 0);
    };
    if (conv_impl() == 1) {
        if (halo_on()) {
            const int c = try_conv_c8(dtype, mode, a, st);
            if (c != 0) { note("c8"); return c < 0 ? c : MPU_OK; }
            const int w = try_conv_ws(dtype, mode, a, st);
            if (w != 0) { note("ws"); return w < 0 ? w : MPU_OK; }
            const int x = try_conv_halo16(dtype, mode, a, st);
            if (x != 0) { note("halo16p"); return x < 0 ? x : MPU_OK; }
            const int h = try_conv_halo(dtype, mode, a, st);
            if (h != 0) { note(h == 2 ? "halo8" : "halo"); return h < 0 ? h : MPU_OK; }
        }
        if (halo_on()) {
            const int k = try_conv_deepk(dtype, mode, a, st);
            if (k != 0) { note("deepk"); return k < 0 ? k : MPU_OK; }
        }
        const int rc = launch_conv_glds(dtype, mode, a, st);
        note(last_glds_schedule());
        return rc;
    }
    note("regs");
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

int launch_conv(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    const int rc = launch_conv_impl(dtype, mode, a, st);
    // accumulator mode of the fused statistics (ConvArgs.stats_acc): the schedule that produced them added them to the
    // accumulator instead of writing its rows -- the caller sees "-1 rows"
    if (!rc && a.stats_acc && a.stats_rows && *a.stats_rows > 0) *a.stats_rows = -1;
    return rc;
}

// workspace (floats) the split-K partials of a wgrad call need
long wgrad_partial_elems(int mode, int Cin, int Cout, long M, int* ksplit_out, int* mchunk_out, bool grouped) {
    const int ntaps = mode == UPCONV2 ? 4 : (mode == CONV1 ? 1 : 9);
    const int bc = (Cin >= 128 && Cout >= 128) ? 128 : 64;
    const long tiles = (long)cdiv(Cin, bc) * cdiv(Cout, bc) * ntaps;
    constexpr long target = 512, nosplit = 384;
    // stand-alone launch: aim at ~512 workgroups (2 per CU: measured best); inside a grouped launch (many jobs, several waves
    // of workgroups): ~256 per job -- half the fp32 partial copies (round 3 sweep: 2.77 -> 2.70 ms per step; 128: 2.74)
    const long tgt = grouped ? (target + 1) / 2 : target;
    long ks = tiles >= nosplit ? 1 : (tgt + tiles - 1) / tiles;
                                                          // no split (and no reduce pass) once the tile grid fills the chip
    const long maxks = (M + 511) / 512;                   // at least 512 pixels per split
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    long mchunk = ((M + ks - 1) / ks + 31) / 32 * 32;
    ks = (M + mchunk - 1) / mchunk;
    if (ksplit_out) *ksplit_out = (int)ks;
    if (mchunk_out) *mchunk_out = (int)mchunk;
    // + fused bias-gradient partials: [ks][ntaps * ci-tiles][Cout]
    long elems = ks * (ntaps * (long)Cin * Cout + (long)ntaps * cdiv(Cin, 64) * Cout);
    // the all-taps kernel (wgrad_taps.hip) keeps one partial copy per pixel strip
    if ((mode == CONV3 || mode == UPCONV2) && (long)Cin * Cout <= TAPS_MAX_CICO) {
        const long te = (long)(TAPS_MAX_WGS / (cdiv(Cin, 64) * cdiv(Cout, 64))) * ((long)ntaps * Cin * Cout + Cout);
        if (te > elems) elems = te;
    }
    return elems;
}

template <typename T, int MODE>
static int launch_wgrad_mode(WgradArgs a, float* dW, hipStream_t st, ReduceQueue* rq, WgradGroup* grp) {
    const int dt_ = sizeof(T) == 2 ? MPU_BF16 : MPU_F32;
    // A concat layer whose first source is not a multiple of 64 channels (complexity_factor 2: 96 | 96, 184 | 184, ...)
    // cannot be tiled over both sources by the LDS-DMA kernels (a 64-channel tile would straddle them): each source runs
    // as its own job with its own partials, and the reduction writes its rows of dW (ReduceJob::cin_job).
    if (a.C1 > 0 && a.C0 % 64 != 0 && a.cin_total == 0 && conv_impl() == 1 && dt_ == MPU_BF16 && rq && MODE == CONV3 &&
        rq->njobs + 2 <= REDUCE_MAX_JOBS) {
        WgradArgs h[2] = {a, a};
        h[0].C1 = 0; h[0].x1 = nullptr; h[0].cin_total = a.C0 + a.C1; h[0].ci_base = 0;
        h[1].x0 = a.x1; h[1].C0 = a.C1; h[1].C1 = 0; h[1].x1 = nullptr; h[1].cin_total = a.C0 + a.C1; h[1].ci_base = a.C0;
        h[1].db = nullptr;                                       // the bias gradient (sum of dz) belongs to the layer once
        bool ok = true;
        long need[2];
        for (int k = 0; k < 2; ++k) {
            const bool grouped = grp != nullptr;
            ok = ok && (wgrad_taps_plan(dt_, MODE, a.B, a.Ho, a.Wo, h[k].C0, 0, a.Cout, grouped).use || wgrad_glds_supported(dt_, MODE, h[k]));
            wgrad_partial_elems(MODE, h[k].C0, a.Cout, (long)a.B * a.Ho * a.Wo, &h[k].ksplit, &h[k].mchunk, grouped);
            need[k] = wgrad_scratch_need(dt_, MODE, a.B, a.Ho, a.Wo, h[k].C0, 0, a.Cout);
            h[k].flops = a.flops * ((double)h[k].C0 / (a.C0 + a.C1));
        }
        if (ok && a.partial_cap > 0 && need[0] + need[1] <= a.partial_cap) {   // (unknown capacity: keep the one-job fallback)
            h[1].partial = a.partial + need[0];
            h[0].partial_cap = need[0]; h[1].partial_cap = need[1];
            for (int k = 0; k < 2; ++k) { const int rc = launch_wgrad_mode<T, MODE>(h[k], dW, st, rq, grp); if (rc) return rc; }
            return MPU_OK;
        }
    }
    const int Cin = a.C0 + a.C1;
    if (conv_impl() == 1) {                      // first layer: 1-2 image channels in 8-channel records
        const int c8 = try_wgrad_c8(dt_, MODE, a, dW, st);
        if (c8 != 0) {
            if (sched_log_on()) sched_note("wgrad c8 mode=%d B=%d H=%d W=%d Cin=%d Cout=%d ksplit=1", MODE, a.B, a.Ho, a.Wo, Cin, a.Cout);
            return c8 < 0 ? c8 : MPU_OK;
        }
    }
    TapsPlan taps; taps.use = 0;
    const bool can_group = grp && rq && rq->njobs < REDUCE_MAX_JOBS && conv_impl() == 1 && dt_ == MPU_BF16;
    if (conv_impl() == 1) taps = wgrad_taps_plan(dt_, MODE, a.B, a.Ho, a.Wo, a.C0, a.C1, a.Cout, can_group && grp->ntaps < TAPS_GROUP_MAX);
    if (taps.use) a.ksplit = (taps.nstrips + 1) / 2;   // one partial copy per pair of pixel strips
    // the LDS-DMA kernels also sum dz over the pixels (bias gradient) when they handle the shape
    a.fuse_db = (a.db && conv_impl() == 1 && (taps.use || wgrad_glds_supported(dt_, MODE, a))) ? 1 : 0;
    if (a.db && !a.fuse_db) {
        int rc0 = launch_colsum(dt_, a.dz, (long)a.B * a.Ho * a.Wo, a.Cout, a.colsum_scratch, a.db, st);
        if (rc0) return rc0;
    }
    const long n_ = (long)ModeTraits<MODE>::NTAPS * Cin * a.Cout;
    if (a.partial_cap > 0) {                     // the chosen plan must fit the caller's region (a neighbour layer's scratch follows it)
        const long rows = taps.use ? a.ksplit : (long)a.ksplit * ModeTraits<MODE>::NTAPS * cdiv(Cin, 64);
        const long used = (long)a.ksplit * n_ + (a.db ? rows * a.Cout : 0);
        if (used > a.partial_cap) {
            snprintf(g_err, sizeof(g_err), "wgrad: plan needs %ld floats of scratch, the region holds %ld (mode %d B=%d %dx%d %d->%d%s)",
                     used, a.partial_cap, MODE, a.B, a.Ho, a.Wo, Cin, a.Cout, grp ? ", grouped" : "");
            return MPU_EINVAL;
        }
    }
    a.db_partial = a.partial + (long)a.ksplit * n_;   // tail of the workspace
    const bool strided = a.cin_total > 0;         // one source of a concat layer: always through the reduction (row remap)
    if (a.ksplit == 1 && !strided) a.partial = dW;   // single split: the kernel's output IS the weight gradient
    const int ntaps = ModeTraits<MODE>::NTAPS;
    const long n = (long)ntaps * Cin * a.Cout;
    const bool will_defer = grp && rq && rq->njobs < REDUCE_MAX_JOBS && conv_impl() == 1 && dt_ == MPU_BF16 &&
                            ((taps.use && grp->ntaps < TAPS_GROUP_MAX) || (!taps.use && grp->nglds < GLDS_GROUP_MAX && wgrad_glds_grid(MODE, a) > 0));
    if (prof_on() && !will_defer)
        prof_begin(PROF_WGRAD, a.flops > 0 ? a.flops : 2.0 * a.B * a.Ho * a.Wo * (double)n, st);
    bool big = false;
    int g_ = 0;
    bool deferred = false;                       // recorded in the group instead of launched (needs the deferred reduction too)
    if (grp && rq && rq->njobs < REDUCE_MAX_JOBS && conv_impl() == 1 && dt_ == MPU_BF16) {
        if (taps.use && grp->ntaps < TAPS_GROUP_MAX) {
            TapsGroupJob& j = grp->taps[grp->ntaps++];
            j.a = a; j.p = taps; j.mode = MODE; j.blk_begin = 0;
            grp->taps_flops += a.flops > 0 ? a.flops : 2.0 * a.B * a.Ho * a.Wo * (double)n;
            deferred = true; g_ = 1;
        } else if (!taps.use && grp->nglds < GLDS_GROUP_MAX && wgrad_glds_grid(MODE, a) > 0) {
            GldsGroupJob& j = grp->glds[grp->nglds++];
            j.a = a; j.mode = MODE; j.blk_begin = 0;
            grp->glds_flops += a.flops > 0 ? a.flops : 2.0 * a.B * a.Ho * a.Wo * (double)n;
            deferred = true; g_ = 1;
        }
    }
    if (!deferred) {                             // (a deferred job is timed as part of its grouped launch at the flush)
        if (taps.use) { g_ = launch_wgrad_taps(MODE, a, taps, st); if (g_) return g_; g_ = 1; }
        else if (conv_impl() == 1) g_ = try_wgrad_glds(dt_, MODE, a, st);
    }
    if (g_ < 0) return g_;
    if (sched_log_on())
        sched_note("wgrad %s mode=%d B=%d H=%d W=%d Cin=%d Cout=%d ksplit=%d%s", taps.use ? "taps" : (g_ == 1 ? "glds" : "regs"),
                   MODE, a.B, a.Ho, a.Wo, Cin, a.Cout, a.ksplit, deferred ? " grouped" : "");
    if (g_ == 1) big = true;                      // launched by an LDS-DMA kernel
    else
    if constexpr (sizeof(T) == 2) {
        if (Cin >= 128 && a.Cout >= 128) {
            big = true;
            dim3 g((unsigned)(cdiv(Cin, 128) * cdiv(a.Cout, 128)), ntaps, a.ksplit);
            launch_k(wgrad_igemm_kernel<T, MODE, 128, 128>, g, dim3(256), 0, st, a);
        }
    }
    if (!big) {
        dim3 g((unsigned)(cdiv(Cin, 64) * cdiv(a.Cout, 64)), ntaps, a.ksplit);
        launch_k(wgrad_igemm_kernel<T, MODE, 64, 64>, g, dim3(256), 0, st, a);
    }
    if (prof_on() && !deferred) prof_end(st);
    int rc = launch_ok();
    if (rc) return rc;
    DbFin f; f.partial = nullptr; f.db = nullptr; f.nshare = 0; f.C = 0; f.main_blocks = 0;
    f.il4_cout = taps.use ? a.Cout : 0;                        // wgrad_taps writes ci-interleaved partial columns
    f.cout = a.Cout;
    if (strided) { f.cin_job = Cin; f.cin_total = a.cin_total; f.ci_base = a.ci_base; }
    int db_blocks = 0;
    if (a.fuse_db) {         // bias gradient: sum the [ksplit * taps * ci-tiles][Cout] partials of the LDS-DMA kernel
        const bool big128 = sizeof(T) == 2 && Cin >= 128 && a.Cout >= 128 && (a.C1 == 0 || a.C0 % 128 == 0);
        f.partial = a.db_partial; f.db = a.db; f.C = a.Cout;
        f.nshare = taps.use ? a.ksplit : a.ksplit * ntaps * cdiv(Cin, big128 ? 128 : 64);
        db_blocks = cdiv(a.Cout, FIN_COLS);
    }
    const long n4 = n / 4;
    const bool kl4 = a.ksplit >= 8 && n4 <= 64L * 8192;
    if (a.ksplit > 1 || strided) {
        if (kl4) f.main_blocks = (int)((n4 + 63) / 64);
        else { long blocks = (n4 + 255) / 256; if (blocks > 4096) blocks = 4096; f.main_blocks = (int)blocks; }
    }
    if (rq && (a.ksplit > 1 || strided || db_blocks) && rq->njobs < REDUCE_MAX_JOBS) {      // deferred: one launch for many layers
        ReduceJob& j = rq->job[rq->njobs++];
        j.partial = a.partial; j.dW = dW; j.n = (a.ksplit > 1 || strided) ? n : 0; j.ksplit = a.ksplit; j.kl4 = kl4 ? 1 : 0;
        j.cout = a.Cout; j.cin_job = f.cin_job; j.cin_total = f.cin_total; j.ci_base = f.ci_base;
        j.db_partial = f.partial; j.db = f.db; j.nshare = f.nshare; j.C = f.C;
        j.blk_begin = rq->nblocks; j.main_blocks = (a.ksplit > 1 || strided) ? f.main_blocks : 0; j.db_blocks = db_blocks; j.il4_cout = f.il4_cout;
        rq->nblocks += j.main_blocks + j.db_blocks;
        return MPU_OK;
    }
    if (a.ksplit == 1 && !strided) return db_blocks ? launch_colsum_finalize(a.db_partial, f.nshare, a.Cout, a.db, st) : MPU_OK;
    if (kl4)
        launch_k(wgrad_reduce_kl4_kernel, dim3((unsigned)(f.main_blocks + db_blocks)), dim3(256), 0, st, a.partial, a.ksplit, n, dW, f);
    else
        launch_k(wgrad_reduce_kernel, dim3((unsigned)(f.main_blocks + db_blocks)), dim3(256), 0, st, a.partial, a.ksplit, n, dW, f);
    return launch_ok();
}

// Floats ONE weight-gradient job writes into its scratch region under one schedule decision (mirrors launch_wgrad_mode):
// the K-split / strip partials of dW + the bias-gradient partial rows. grouped: the plan a job inside a grouped launch takes
// (fewer strips / K splits per job -- but also a lower taps threshold, so a layer that runs as wgrad_glds on its own can run
// as wgrad_taps inside a group and the other way round: the two plans have DIFFERENT layouts and either can be the larger).
long wgrad_job_floats(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, bool grouped) {
    const int Cin = C0 + C1;
    const int ntaps = mode == UPCONV2 ? 4 : (mode == CONV1 ? 1 : 9);
    const long M = (long)B * H * W, n = (long)ntaps * Cin * Cout;
    int ks = 1, mchunk = 0;
    wgrad_partial_elems(mode, Cin, Cout, M, &ks, &mchunk, grouped);
    TapsPlan taps; taps.use = 0;
    if (conv_impl() == 1) taps = wgrad_taps_plan(dtype, mode, B, H, W, C0, C1, Cout, grouped);
    if (taps.use) ks = (taps.nstrips + 1) / 2;
    const long nshare = taps.use ? ks : (long)ks * ntaps * cdiv(Cin, 64);     // (128-wide tiles: half of this)
    return (long)ks * n + nshare * Cout;
}

// scratch of one layer: the LARGER of the stand-alone and the grouped plan (the region was sized with the
// stand-alone plan and used with the grouped one, which overflowed into the next layer's region on shapes outside the
// BASELINE configs, e.g. bf16 B=2 40x40 128->128). launch_wgrad_mode checks the chosen plan against partial_cap.
long wgrad_scratch_need(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, int c0_logical) {
    const long a0 = wgrad_job_floats(dtype, mode, B, H, W, C0, C1, Cout, false);
    const long a1 = wgrad_job_floats(dtype, mode, B, H, W, C0, C1, Cout, true);
    long need = a0 > a1 ? a0 : a1;
    if (C1 > 0 && C0 % 64 != 0 && conv_impl() == 1 && dtype == MPU_BF16 && mode == CONV3)   // two jobs (launch_wgrad_mode): both regions
        return wgrad_scratch_need(dtype, mode, B, H, W, C0, 0, Cout) + wgrad_scratch_need(dtype, mode, B, H, W, C1, 0, Cout) +
               ((need + 63) / 64 * 64 + 64);                                                  // (+ the unsplit need: the fallback still fits)
    // the first-layer schedule (tried first) keeps one compact row per strip of image rows: its own layout and size
    const long c8 = conv_impl() == 1 ? wgrad_c8_scratch_floats(dtype, mode, B, H, W, C0, C1, c0_logical, Cout) : 0;
    if (c8 > need) need = c8;
    return (need + 63) / 64 * 64 + 64;
}

int launch_wgrad(int dtype, int mode, const WgradArgs& a, float* dW, hipStream_t st, ReduceQueue* rq, WgradGroup* grp) {
#define MPU_WG_CASE(TT)                                                            \
    switch (mode) {                                                                \
        case CONV3: return launch_wgrad_mode<TT, CONV3>(a, dW, st, rq, grp);            \
        case UPCONV2: return launch_wgrad_mode<TT, UPCONV2>(a, dW, st, rq, grp);        \
        case CONV1: return launch_wgrad_mode<TT, CONV1>(a, dW, st, rq, grp);            \
        default: return fail(MPU_EINVAL, "%s", "wgrad: bad mode");                 \
    }
    if (dtype == MPU_BF16) { MPU_WG_CASE(bf16_t) }
    if (dtype == MPU_F32) { MPU_WG_CASE(float) }
#undef MPU_WG_CASE
    return fail(MPU_EINVAL, "%s", "wgrad: bad dtype");
}

}  // namespace mpu
