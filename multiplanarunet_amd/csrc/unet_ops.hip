// Memory-bound U-Net kernels around the MFMA convolutions: weight packing,
// BatchNormalization (train statistics / apply / backward), 2x2 max-pool and its
// gradient, bias-gradient column sums, the 1x1 softmax head with the Keras
// sparse-CE gradient, and Adam. All activations are [M][C] (NHWC flattened),
// C a multiple of 8; every kernel moves 16 B per lane.
//
// Reference semantics: mpunet/models/unet.py:114-216 (layer order), Keras
// defaults restated in SURVEY.md section 8a rows a6/a7, oracle/unet_ref.py.
#include <cmath>
#include "kernels.h"
#include "reduce.h"

namespace mpu {

template <typename T> struct Vec;   // one 16-byte chunk of T as floats
template <> struct Vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 t = *(const uint4*)p;
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = f32x2_to_bf16x2(v[2 * i], v[2 * i + 1]);
        *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

constexpr int COEF_LDS_C = 1024;      // per-channel coefficient tables up to this many channels are staged in LDS
template <int NTAB>
__device__ __forceinline__ const float* stage_coeffs(const float* __restrict__ k, int C, float* lds) {
    if (C > COEF_LDS_C) return nullptr;
    for (int i = threadIdx.x; i < NTAB * C; i += 256) lds[i] = k[i];
    __syncthreads();
    return lds;
}
template <int N>
__device__ __forceinline__ void coef_load(const float* lds, const float* __restrict__ glob, int off, float* out) {
    if (lds) {
#pragma unroll
        for (int i = 0; i < N; i += 4) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(lds + off + i);
    } else {
#pragma unroll
        for (int i = 0; i < N; i += 4) *reinterpret_cast<float4*>(out + i) = *reinterpret_cast<const float4*>(glob + off + i);
    }
}

static inline int ew_grid(long work) {
    long b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ------------------------------------------------------------------------- //
// weight packing: fp32 master (Keras HWIO = [tap][ci][co]) -> MFMA operands
// ------------------------------------------------------------------------- //
// forward operand [tap][co][ci]: 32x32 tiles transposed through LDS (coalesced reads along co,
// coalesced writes along ci)
template <typename T>
__global__ __launch_bounds__(256) void pack_fwd_kernel(int ntaps, const float* __restrict__ W, int Cin, int Cout,
                                                       T* __restrict__ wf) {
    __shared__ float tile[32][33];
    const int tci = (Cin + 31) / 32, tco = (Cout + 31) / 32;
    const long ntiles = (long)ntaps * tci * tco;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tap = (int)(t / (tci * tco)); const int r = (int)(t % (tci * tco));
        const int ci0 = (r / tco) * 32, co0 = (r % tco) * 32;
        const float* src = W + (long)tap * Cin * Cout;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ci0 + ty + 8 * k, co = co0 + tx;
            tile[ty + 8 * k][tx] = (ci < Cin && co < Cout) ? src[(long)ci * Cout + co] : 0.f;
        }
        __syncthreads();
        T* dst = wf + (long)tap * Cin * Cout;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = co0 + ty + 8 * k, ci = ci0 + tx;
            if (ci < Cin && co < Cout) dst[(long)co * Cin + ci] = from_f32<T>(tile[tx][ty + 8 * k]);
        }
        __syncthreads();
    }
}

// data-gradient operand [tap'][ci][co] (same element order as the Keras kernel: coalesced)
template <typename T>
__global__ void pack_dgrad_kernel(int mode, const float* __restrict__ W, int Cin, int Cout, T* __restrict__ wd) {
    const long per_tap = (long)Cin * Cout;
    if (mode == CONV1) {
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per_tap; e += (long)gridDim.x * blockDim.x)
            wd[e] = from_f32<T>(W[e]);
        return;
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < 9 * per_tap;
         e += (long)gridDim.x * blockDim.x) {
        const int tp = (int)(e / per_tap); const long r = e % per_tap;
        float v;
        if (mode == CONV3) {
            v = W[(long)(8 - tp) * per_tap + r];                 // 180-degree rotated taps
        } else {                                                  // UPCONV2 -> 3x3 stride-2 combined taps
            const int dy = tp / 3 - 1, dx = tp % 3 - 1;           // S(-1)={1}, S(0)={0,1}, S(1)={0}
            v = 0.f;
            for (int ky = 0; ky < 2; ++ky) {
                if ((dy == -1 && ky != 1) || (dy == 1 && ky != 0)) continue;
                for (int kx = 0; kx < 2; ++kx) {
                    if ((dx == -1 && kx != 1) || (dx == 1 && kx != 0)) continue;
                    v += W[(long)(ky * 2 + kx) * per_tap + r];
                }
            }
        }
        wd[e] = from_f32<T>(v);
    }
}

// ---- all layers of a model in ONE launch ------------------------------------------------------
// unit = one 32x32 forward tile, or one 1024-element chunk of the data-gradient operand; the job
// table travels in the kernel arguments (no device-side table to keep in sync).
template <typename T>
__device__ __forceinline__ void pack_fwd_tile(const PackJob& j, int t, const float* __restrict__ params, T* packed,
                                              float (*tile)[65]) {
    // 64x64 tile: 16-byte reads along co, transpose through LDS, 16-byte writes along ci
    constexpr int N = Vec<T>::N;
    const int Cin = j.Cin, Cout = j.Cout;
    const int tci = (Cin + 63) / 64, tco = (Cout + 63) / 64;
    const int tap = t / (tci * tco); const int r = t % (tci * tco);
    const int ci0 = (r / tco) * 64, co0 = (r % tco) * 64;
    const float* src = params + j.w + (long)tap * Cin * Cout;
    {
        const int ty = threadIdx.x >> 4, tx4 = (threadIdx.x & 15) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cil = ty + 16 * k, ci = ci0 + cil, co = co0 + tx4;
            // clamped, unconditional load + select (Cout % 8 == 0): the four loads of a thread are in flight together
            const bool in = ci < Cin && co < Cout;
            float4 v = *reinterpret_cast<const float4*>(src + (long)(ci < Cin ? ci : Cin - 1) * Cout + (co < Cout ? co : Cout - 4));
            if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
            tile[cil][tx4] = v.x; tile[cil][tx4 + 1] = v.y; tile[cil][tx4 + 2] = v.z; tile[cil][tx4 + 3] = v.w;
        }
    }
    __syncthreads();
    T* dst = packed + j.wf + (long)tap * Cin * Cout;
    constexpr int GPR = 64 / N, RPP = 256 / GPR;                  // 16-byte groups per co row, co rows per pass
#pragma unroll
    for (int pass = 0; pass < 64 / RPP; ++pass) {
        const int col = threadIdx.x / GPR + pass * RPP, cil = (threadIdx.x % GPR) * N;
        const int co = co0 + col, ci = ci0 + cil;
        if (ci < Cin && co < Cout) {                              // Cin % 8 == 0: the whole vector is in range
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = tile[cil + e][col];
            Vec<T>::store(dst + (long)co * Cin + ci, v);
        }
    }
}

template <typename T>
__device__ __forceinline__ void pack_dgrad_chunk(const PackJob& j, int chunk, const float* __restrict__ params, T* packed) {
    const long per_tap = (long)j.Cin * j.Cout;                 // multiple of 64: 8 consecutive elements never straddle taps
    const long e = (long)chunk * 2048 + threadIdx.x * 8;
    if (e >= 9 * per_tap) return;
    const int tp = (int)(e / per_tap); const long r = e % per_tap;
    const float* W = params + j.w;
    float v[8];
    auto load8 = [&](const float* p, float* o) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    };
    if (j.mode == CONV3) {
        load8(W + (long)(8 - tp) * per_tap + r, v);            // 180-degree rotated taps
    } else {                                                    // UPCONV2 -> 3x3 stride-2 combined taps
        const int dy = tp / 3 - 1, dx = tp % 3 - 1;             // S(-1)={1}, S(0)={0,1}, S(1)={0}
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        for (int ky = 0; ky < 2; ++ky) {
            if ((dy == -1 && ky != 1) || (dy == 1 && ky != 0)) continue;
            for (int kx = 0; kx < 2; ++kx) {
                if ((dx == -1 && kx != 1) || (dx == 1 && kx != 0)) continue;
                float u[8];
                load8(W + (long)(ky * 2 + kx) * per_tap + r, u);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += u[i];
            }
        }
    }
    T* dst = packed + j.wd + e;
    constexpr int N = Vec<T>::N;
#pragma unroll
    for (int h = 0; h < 8 / N; ++h) {
        float w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = v[h * N + i];
        Vec<T>::store(dst + h * N, w);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(PackTable tab, const float* __restrict__ params, T* packed) {
    __shared__ float tile[64][65];
    int ji = 0;
    while (ji + 1 < tab.njobs && (int)blockIdx.x >= tab.job[ji + 1].unit_begin) ++ji;
    const PackJob& j = tab.job[ji];
    const int u = (int)blockIdx.x - j.unit_begin;
    if (u < j.fwd_units) pack_fwd_tile<T>(j, u, params, packed, tile);
    else pack_dgrad_chunk<T>(j, u - j.fwd_units, params, packed);
}

int launch_pack_all(int dtype, PackTable& tab, const float* params, void* packed, hipStream_t st) {
    int units = 0;
    for (int i = 0; i < tab.njobs; ++i) {
        PackJob& j = tab.job[i];
        const int ntaps = j.mode == UPCONV2 ? 4 : 9;
        j.unit_begin = units;
        j.fwd_units = ntaps * cdiv(j.Cin, 64) * cdiv(j.Cout, 64);
        units += j.fwd_units + (int)cdiv(9L * j.Cin * j.Cout, 2048L);
    }
    if (units == 0) return MPU_OK;
    if (dtype == MPU_BF16) pack_all_kernel<bf16_t><<<units, 256, 0, st>>>(tab, params, (bf16_t*)packed);
    else pack_all_kernel<float><<<units, 256, 0, st>>>(tab, params, (float*)packed);
    return launch_ok();
}

// dtype "bf16x3": packed f32 operands -> (bf16 hi | bf16 lo << 16) words, in place, after every refresh of the packed copies
// (common.h: x3_word / x3_unpack). n = 32-bit words.
__global__ __launch_bounds__(256) void x3_words_kernel(uint32_t* __restrict__ buf, long n) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 + 4 <= n) {
        uint4 v = *reinterpret_cast<uint4*>(buf + i4);
        v.x = x3_word(__uint_as_float(v.x)); v.y = x3_word(__uint_as_float(v.y));
        v.z = x3_word(__uint_as_float(v.z)); v.w = x3_word(__uint_as_float(v.w));
        *reinterpret_cast<uint4*>(buf + i4) = v;
    } else {
        for (long i = i4; i < n; ++i) buf[i] = x3_word(__uint_as_float(buf[i]));
    }
}
int launch_x3_words(void* buf, long n, hipStream_t st) {
    if (n <= 0) return MPU_OK;
    x3_words_kernel<<<(unsigned)((n + 1023) / 1024), 256, 0, st>>>((uint32_t*)buf, n);
    return launch_ok();
}

int launch_pack_weights(int dtype, int mode, const float* W, int Cin, int Cout, void* wf, void* wd, hipStream_t st) {
    const int ntaps = mode == UPCONV2 ? 4 : (mode == CONV1 ? 1 : 9);
    long tiles = (long)ntaps * cdiv(Cin, 32) * cdiv(Cout, 32);
    if (tiles > 4096) tiles = 4096;
    const long n = 9L * Cin * Cout;
    if (dtype == MPU_BF16) {
        pack_fwd_kernel<bf16_t><<<(unsigned)tiles, 256, 0, st>>>(ntaps, W, Cin, Cout, (bf16_t*)wf);
        if (wd) pack_dgrad_kernel<bf16_t><<<ew_grid(n), 256, 0, st>>>(mode, W, Cin, Cout, (bf16_t*)wd);
    } else {
        pack_fwd_kernel<float><<<(unsigned)tiles, 256, 0, st>>>(ntaps, W, Cin, Cout, (float*)wf);
        if (wd) pack_dgrad_kernel<float><<<ew_grid(n), 256, 0, st>>>(mode, W, Cin, Cout, (float*)wd);
    }
    return launch_ok();
}

// one thread = one group of 8 output channels of one pixel (Cpad % 8 == 0): clamped unconditional loads (a predicated
// scalar load per element is waited for on the spot), 16-byte stores
template <typename T>
__global__ void cast_pad_kernel(const float* __restrict__ x, long M, int Cin, int Cpad, T* __restrict__ out, long long* zero_p,
                                long zero_n) {
    // (training forward: the step's first launch also zeroes the BatchNorm accumulators of both passes -- ConvArgs.stats_acc)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += (long)gridDim.x * blockDim.x) zero_p[i] = 0;
    const int gpp = Cpad >> 3;
    const long n = M * gpp;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long m = e / gpp; const int c0 = (int)(e - m * gpp) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            const float t = x[m * Cin + (c < Cin ? c : Cin - 1)];
            v[i] = c < Cin ? t : 0.f;
        }
        constexpr int N = Vec<T>::N;
#pragma unroll
        for (int h = 0; h < 8 / N; ++h) {
            float w[N];
#pragma unroll
            for (int i = 0; i < N; ++i) w[i] = v[h * N + i];
            Vec<T>::store(out + e * 8 + h * N, w);
        }
    }
}
int launch_cast_pad(int dtype, const float* x, long M, int Cin, int Cpad, void* out, hipStream_t st, long long* zero_p, long zero_n) {
    if (Cpad % 8 || Cin < 1 || Cin > Cpad) return fail(MPU_EINVAL, "%s", "cast_pad: padded channel count must be a multiple of 8");
    if (!zero_p) zero_n = 0;
    if (dtype == MPU_BF16) cast_pad_kernel<bf16_t><<<ew_grid(M * Cpad / 8), 256, 0, st>>>(x, M, Cin, Cpad, (bf16_t*)out, zero_p, zero_n);
    else cast_pad_kernel<float><<<ew_grid(M * Cpad / 8), 256, 0, st>>>(x, M, Cin, Cpad, (float*)out, zero_p, zero_n);
    return launch_ok();
}

// ------------------------------------------------------------------------- //
// per-channel reductions over the M rows of [M][C]: two deterministic stages
//   OP 0: (sum x, sum x^2)                      BN statistics
//   OP 1: (sum dn, sum dn*(x-mean)*invstd)      BN backward
//   OP 2: (sum x)                               bias gradient
// partial layout [nblk][NS][C]
// ------------------------------------------------------------------------- //
constexpr int CR_THREADS = 512;      // column-reduction block size
// SPLIT (dtype "bf16x3", OP 2 on the f32 dz of a conv): the pass that sums the bias gradient also writes the three bf16 planes
// hi | hi | lo of its input for the weight-gradient kernels (see split3_kernel) -- dz is read once instead of twice
template <typename T, int OP, bool SPLIT = false>
__global__ __launch_bounds__(CR_THREADS) void colreduce_kernel(const T* __restrict__ a, const T* __restrict__ b, long M, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        int rows_per_blk, float* __restrict__ partial, long long* acc_out,
                                                        float acc_scale0, float acc_scale1, bf16_t* __restrict__ sp_out = nullptr,
                                                        int sp_two = 0 /* 1: two stored planes hi | lo */) {
    static_assert(!SPLIT || (OP == 2 && Vec<T>::N == 4), "SPLIT: the f32 bias-gradient pass");
    constexpr int N = Vec<T>::N;
    constexpr int NS = OP == 2 ? 1 : 2;
    __shared__ float red[CR_THREADS * N * NS];
    __shared__ float red2[(256 * N * NS > CR_THREADS) ? 256 * N * NS : CR_THREADS];   // [Q][V], Q*V <= max(CR_THREADS, V)
    const int cpr = C / N;
    int TX = 1; while (TX < cpr && TX < 256) TX <<= 1;
    const int TY = CR_THREADS / TX;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const long r_beg = (long)blockIdx.x * rows_per_blk;
    long r_end = r_beg + rows_per_blk; if (r_end > M) r_end = M;
    for (int cg = 0; cg < cpr; cg += TX) {
        const int c = cg + tx;
        float s0[N], s1[N];
#pragma unroll
        for (int i = 0; i < N; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        if (c < cpr) {
            float mu[N], is[N];
            if (OP == 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) { mu[i] = mean[c * N + i]; is[i] = invstd[c * N + i]; }
            }
            auto split_store = [&](const float* v, long e) {      // e: element index of v[0] in the [M][C] tensor
                const long n = M * C;
                const uint32_t h0 = f32x2_to_bf16x2(v[0], v[1]), h1 = f32x2_to_bf16x2(v[2], v[3]);
                const uint32_t l0 = f32x2_to_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = f32x2_to_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                *reinterpret_cast<uint2*>(sp_out + e) = make_uint2(h0, h1);
                if (sp_two) { *reinterpret_cast<uint2*>(sp_out + n + e) = make_uint2(l0, l1); return; }
                *reinterpret_cast<uint2*>(sp_out + n + e) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(sp_out + 2 * n + e) = make_uint2(l0, l1);
            };
            auto accum = [&](const float* va, const float* vb) {
                if (OP == 0) {
#pragma unroll
                    for (int i = 0; i < N; ++i) { s0[i] += va[i]; s1[i] += va[i] * va[i]; }
                } else if (OP == 1) {
#pragma unroll
                    for (int i = 0; i < N; ++i) { s0[i] += va[i]; s1[i] += va[i] * ((vb[i] - mu[i]) * is[i]); }
                } else {
#pragma unroll
                    for (int i = 0; i < N; ++i) s0[i] += va[i];
                }
            };
            long r = r_beg + ty;
            for (; r + 3L * TY < r_end; r += 4L * TY) {       // 4 (8) independent 16-B loads in flight per thread
                float va[4][N], vb[4][N];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    Vec<T>::load(a + (r + (long)u * TY) * C + (long)c * N, va[u]);
                    if (OP == 1) Vec<T>::load(b + (r + (long)u * TY) * C + (long)c * N, vb[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    accum(va[u], vb[u]);
                    if (SPLIT) split_store(va[u], (r + (long)u * TY) * C + (long)c * N);
                }
            }
            for (; r < r_end; r += TY) {
                float va[N], vb[N];
                Vec<T>::load(a + r * C + (long)c * N, va);
                if (OP == 1) Vec<T>::load(b + r * C + (long)c * N, vb);
                accum(va, vb);
                if (SPLIT) split_store(va, r * C + (long)c * N);
            }
        }
        // reduce over ty through LDS
#pragma unroll
        for (int i = 0; i < N; ++i) {
            red[(threadIdx.x * NS + 0) * N + i] = s0[i];
            if (NS == 2) red[(threadIdx.x * NS + 1) * N + i] = s1[i];
        }
        __syncthreads();
        {   // two-level block reduction over the TY row-lanes: V = TX*NS*N values per row-lane; every thread sums a
            // slice of the row-lanes of one value, then one thread per value combines the Q slices (fixed order)
            const int V = TX * NS * N;
            const int Q = CR_THREADS / V > 0 ? CR_THREADS / V : 1;
            const int per = (TY + Q - 1) / Q;
            for (int idx = threadIdx.x; idx < V * Q; idx += CR_THREADS) {
                const int v = idx % V, q = idx / V;
                double acc = 0.0;
                for (int y = q * per; y < (q + 1) * per && y < TY; ++y) acc += (double)red[y * V + v];
                red2[q * V + v] = (float)acc;
            }
            __syncthreads();
            for (int v = threadIdx.x; v < V; v += CR_THREADS) {
                double accd = 0.0;
                for (int q = 0; q < Q; ++q) accd += (double)red2[q * V + v];
                const float acc = (float)accd;
                const int tx2 = v / (NS * N), s = (v / N) % NS, i = v % N;
                const int c2 = cg + tx2;
                if (c2 < cpr) {
                    if (acc_out) stats_acc_add(acc_out, C, s, c2 * N + i, acc, s ? acc_scale1 : acc_scale0);    // (accumulator mode, kernels.h)
                    else partial[((long)blockIdx.x * NS + s) * C + c2 * N + i] = acc;
                }
            }
        }
        __syncthreads();
    }
}

static int red_blocks(long M, int C, int* rows_per_blk) {
    long rpb = (M + RED_MAX_BLOCKS - 1) / RED_MAX_BLOCKS;
    int TX = 1; while (TX < C / 8 && TX < 256) TX <<= 1;      // as in the kernel (f32: C/4, only more rows per thread)
    const long min_rows = 4L * (CR_THREADS / TX);             // >= 4 rows per thread
    if (rpb < min_rows) rpb = min_rows;
    *rows_per_blk = (int)rpb;
    return (int)((M + rpb - 1) / rpb);
}

template <int OP>
static int launch_colreduce(int dtype, const void* a, const void* b, long M, int C, const float* mean,
                            const float* invstd, float* partial, int* nblk_out, hipStream_t st, long long* acc = nullptr,
                            const float* acc_scale = nullptr) {
    int rpb; const int nblk = red_blocks(M, C, &rpb);
    *nblk_out = nblk;
    const float s0 = acc ? acc_scale[0] : 0.f, s1 = acc ? acc_scale[1] : 0.f;
    if (dtype == MPU_BF16)
        colreduce_kernel<bf16_t, OP><<<nblk, CR_THREADS, 0, st>>>((const bf16_t*)a, (const bf16_t*)b, M, C, mean, invstd, rpb, partial, acc, s0, s1);
    else
        colreduce_kernel<float, OP><<<nblk, CR_THREADS, 0, st>>>((const float*)a, (const float*)b, M, C, mean, invstd, rpb, partial, acc, s0, s1);
    return launch_ok();
}


// The arithmetic shared by the separate and the folded kernels lives in these helpers with EXPLICIT fused operations / contraction
// off: left to -ffp-contract=fast the compiler fuses `a * b + c * d + e` differently in different kernels (measured: the folded
// backward pass differed from the separate one in the last bit of most gradients until these were pinned).
__device__ __forceinline__ float bn_affine(float v, float sc, float sh) { return fmaf(v, sc, sh); }
__device__ __forceinline__ float bn_bwd_affine(float g, float v, float k1, float k2, float k3) { return fmaf(k1, g, fmaf(k2, v, k3)); }
struct BnFwdCoef { float mean, invstd, scale, shift, mmean, mvar; };
__device__ __forceinline__ BnFwdCoef bn_fwd_coeffs(double s, double ss, long M, float gamma, float beta, float mm, float mv,
                                                   float eps, float mom) {
#pragma clang fp contract(off)
    BnFwdCoef r;
    const double mu = s / (double)M;
    double var = ss / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    r.invstd = (float)(1.0 / sqrt(var + (double)eps));
    r.scale = gamma * r.invstd;
    r.mean = (float)mu;
    r.shift = beta - (float)mu * r.scale;
    const double ub = var * ((double)M / (double)(M > 1 ? M - 1 : 1));
    r.mmean = mm * mom + (float)mu * (1.f - mom);
    r.mvar = mv * mom + (float)ub * (1.f - mom);
    return r;
}
__device__ __forceinline__ void bn_bwd_coeffs(double s, double sx, long M, float gamma, float mean, float invstd,
                                              float& k1, float& k2, float& k3) {
#pragma clang fp contract(off)
    const double sc = (double)gamma * (double)invstd;
    const double mdn = s / (double)M, mdx = sx / (double)M;
    const double k2d = -sc * mdx * (double)invstd;
    k1 = (float)sc; k2 = (float)k2d; k3 = (float)(-sc * mdn - k2d * (double)mean);
}

// COLMAJOR: partial is [2][C][nblk] (written by the conv epilogues), else [nblk][2][C] (colreduce)
template <bool COLMAJOR>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partial, int nblk, int C, long M,
                                         const float* gamma, const float* beta, float* mmean, float* mvar,
                                         float* mean, float* invstd, float* scale, float* shift, float eps, float mom) {
    __shared__ double red[256];
    const int c = blockIdx.x * FIN_COLS + (COLMAJOR ? threadIdx.x / FIN_KL : threadIdx.x % FIN_COLS);
    double st[2];
    if (COLMAJOR) {
        partial_sums_colmajor<2>(partial, nblk, C, c, c < C, red, st);
        if (c >= C || threadIdx.x % FIN_KL) return;
    } else {
        partial_sums<2>(partial, nblk, 2L * C, C, c, c < C, red, st);
        if (c >= C || threadIdx.x >= FIN_COLS) return;
    }
    const BnFwdCoef k = bn_fwd_coeffs(st[0], st[1], M, gamma[c], beta[c], mmean[c], mvar[c], eps, mom);
    mean[c] = k.mean; invstd[c] = k.invstd; scale[c] = k.scale; shift[c] = k.shift;
    mmean[c] = k.mmean; mvar[c] = k.mvar;
}

int launch_bn_stats(int dtype, const void* x, long M, int C, float* partial, const float* gamma, const float* beta,
                    float* mmean, float* mvar, float* mean, float* invstd, float* scale, float* shift,
                    float eps, float momentum, int ready_rows, hipStream_t st) {
    int nblk = ready_rows;                       // > 0: `partial` already holds that many rows from the conv epilogue
    if (nblk <= 0) {
        int rc = launch_colreduce<0>(dtype, x, nullptr, M, C, nullptr, nullptr, partial, &nblk, st);
        if (rc) return rc;
    }
    if (ready_rows > 0)
        bn_stats_finalize_kernel<true><<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, M, gamma, beta, mmean, mvar,
                                                                          mean, invstd, scale, shift, eps, momentum);
    else
        bn_stats_finalize_kernel<false><<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, M, gamma, beta, mmean, mvar,
                                                                           mean, invstd, scale, shift, eps, momentum);
    return launch_ok();
}

__global__ void bn_infer_coeffs_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                       int C, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.f / sqrtf(mv[c] + eps));
    scale[c] = sc; shift[c] = beta[c] - mm[c] * sc;
}
int launch_bn_infer_coeffs(const float* gamma, const float* beta, const float* mm, const float* mv, int C, float eps,
                           float* scale, float* shift, hipStream_t st) {
    bn_infer_coeffs_kernel<<<cdiv(C, 128), 128, 0, st>>>(gamma, beta, mm, mv, C, eps, scale, shift);
    return launch_ok();
}

// BatchNormalization finalize FOLDED into the apply pass (round 5) for producers that leave at most 64 partial rows per channel
// (conv_deepk: 32; conv_halo8 on 64-filter tiles: 64): a workgroup owns a 64-channel slab x a pixel group, its first 128 threads
// sum the rows of (channel, statistic) themselves -- strictly in row order, in double: for <= 64 rows that IS the order of
// bn_stats_finalize_kernel (one row per k-lane, the k-lanes combined in order, the absent ones adding +0), so the coefficients
// are the same bits -- and the pixel group 0 of a slab also writes mean / 1/std / scale / shift and the moving statistics, as
// the finalize launch did. The pixels of the first pass are requested before the rows, gamma / beta at kernel entry. Rounds 3's
// attempt at this lost 0.7 % with 256 rows (the serial reduction in front of every workgroup); with <= 64 rows it replaces a
// 5.2-us launch by ~1 us inside the next one.
constexpr int BN_FOLD_MAX_ROWS = 64;

// rows of one (statistic, channel): partial[(st * C + c) * nblk + k], nblk % 4 == 0, nblk <= 64; sequential double sum
__device__ __forceinline__ double bn_fold_rowsum(const float* __restrict__ p, int nblk) {
    float4 v[BN_FOLD_MAX_ROWS / 4];
#pragma unroll
    for (int u = 0; u < BN_FOLD_MAX_ROWS / 4; ++u) {          // unconditional (clamped) loads: all in flight together
        const float4 t = *(const float4*)(p + (4 * u < nblk ? 4 * u : 0));
        const bool on = 4 * u < nblk;
        v[u] = make_float4(on ? t.x : 0.f, on ? t.y : 0.f, on ? t.z : 0.f, on ? t.w : 0.f);
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < BN_FOLD_MAX_ROWS / 4; ++u) { s += (double)v[u].x; s += (double)v[u].y; s += (double)v[u].z; s += (double)v[u].w; }
    return s;
}

// accumulator mode (round 6, ConvArgs.stats_acc): the eight per-XCD fixed-point sums of one (statistic, channel); exact
__device__ __forceinline__ double bn_acc_sum(const long long* __restrict__ acc, int C, int st, int c, float inv_scale) {
    long long v[BN_ACC_ROWS];
#pragma unroll
    for (int r = 0; r < BN_ACC_ROWS; ++r) v[r] = acc[((long)r * 2 + st) * C + c];          // all eight loads in flight
    long long t = 0;
#pragma unroll
    for (int r = 0; r < BN_ACC_ROWS; ++r) t += v[r];
    return (double)t * (double)inv_scale;
}

template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_fold_apply_kernel(const T* __restrict__ x, int B, int H, int W, int C,
                                                            const float* __restrict__ partial, int nblk,
                                                            const long long* __restrict__ acc, float inv0, float inv1, long M,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* mmean, float* mvar, float* mean, float* invstd, float* scale,
                                                            float* shift, float eps, float mom, T* __restrict__ y,
                                                            T* __restrict__ pooled, int nslab, int ppw) {
    constexpr int N = Vec<T>::N, TPP = 64 / N, PPP = 256 / TPP;  // threads per pixel (64 channels), pixels per pass
    __shared__ double red[2][64];
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    const int slab = (int)blockIdx.x % nslab, g = (int)blockIdx.x / nslab;
    const int c0 = slab * 64, tc = (threadIdx.x % TPP) * N, tp = threadIdx.x / TPP;
    const int Hp = H / 2, Wp = W / 2;
    const long Q = POOL ? (long)B * Hp * Wp : (long)B * H * W;   // work items: pixels, or pooled pixels (2 x 2 source pixels each)
    const long q0 = (long)g * ppw;
    const int npass = ppw / PPP;
    auto src_off = [&](long q, int d) -> long {                  // element offset of source pixel d (POOL: 0..3) of work item q
        if (!POOL) return q * C + c0 + tc;
        const int px = (int)(q % Wp); long t = q / Wp;
        const int py = (int)(t % Hp); const int b = (int)(t / Hp);
        return ((((long)b * H + 2 * py + (d >> 1)) * W + 2 * px + (d & 1)) * C) + c0 + tc;
    };
    constexpr int ND = POOL ? 4 : 1, NP = POOL ? 1 : 4;          // source pixels per item; passes per workgroup (all requested up front)
    float xv[NP][ND][N];
#pragma unroll
    for (int it = 0; it < NP; ++it) {                            // every pixel of the workgroup is requested BEFORE the rows
        const long q = q0 + (long)it * PPP + tp;
        const long qc = (it < npass && q < Q) ? q : (q0 < Q ? q0 : Q - 1);
#pragma unroll
        for (int d = 0; d < ND; ++d) Vec<T>::load(x + src_off(qc, d), xv[it][d]);
    }
    float g_ = 0.f, b_ = 0.f, mm_ = 0.f, mv_ = 0.f;
    if (threadIdx.x < 64) {
        g_ = gamma[c0 + threadIdx.x]; b_ = beta[c0 + threadIdx.x];
        mm_ = mmean[c0 + threadIdx.x]; mv_ = mvar[c0 + threadIdx.x];
    }
    if (threadIdx.x < 128) {
        const int cl = threadIdx.x & 63, st = threadIdx.x >> 6;
        red[st][cl] = acc ? bn_acc_sum(acc, C, st, c0 + cl, st ? inv1 : inv0)
                          : bn_fold_rowsum(partial + ((long)st * C + c0 + cl) * nblk, nblk);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                      // the arithmetic of bn_stats_finalize_kernel, expression for expression
        const int c = c0 + threadIdx.x;
        const BnFwdCoef k = bn_fwd_coeffs(red[0][threadIdx.x], red[1][threadIdx.x], M, g_, b_, mm_, mv_, eps, mom);
        coef[0][threadIdx.x] = k.scale; coef[1][threadIdx.x] = k.shift;
        if (g == 0) {
            mean[c] = k.mean; invstd[c] = k.invstd; scale[c] = k.scale; shift[c] = k.shift;
            mmean[c] = k.mmean; mvar[c] = k.mvar;
        }
    }
    __syncthreads();
    float sc[N], sh[N];
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        *reinterpret_cast<float4*>(sc + i) = *reinterpret_cast<const float4*>(&coef[0][tc + i]);
        *reinterpret_cast<float4*>(sh + i) = *reinterpret_cast<const float4*>(&coef[1][tc + i]);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const long q = q0 + (long)it * PPP + tp;
        if (it < npass && q < Q) {
            float mx[N];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
#pragma unroll
                for (int i = 0; i < N; ++i) xv[it][d][i] = bn_affine(xv[it][d][i], sc[i], sh[i]);
                Vec<T>::store(y + src_off(q, d), xv[it][d]);
                if (POOL) {                                      // the pooled value is the max of the STORED (rounded) values
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float r = to_f32<T>(from_f32<T>(xv[it][d][i]));
                        mx[i] = d == 0 ? r : fmaxf(mx[i], r);
                    }
                }
            }
            if (POOL) Vec<T>::store(pooled + q * C + c0 + tc, mx);
        }
    }
}

// 1 = launched (finalize + apply in one pass), 0 = not suited (the caller runs the two launches)
// acc != NULL (nblk = -1): the producer added its sums to the fixed-point accumulator (scales acc_scale[2]) instead of writing rows
bool bn_fold_shape_ok(int C, int H, int W, bool pooled) { return !(C & 63) && !(pooled && ((H | W) & 1)); }
int launch_bn_fold_fwd(int dtype, const void* x, int B, int H, int W, int C, const float* partial, int nblk,
                       const float* gamma, const float* beta, float* mmean, float* mvar, float* mean, float* invstd,
                       float* scale, float* shift, float eps, float momentum, void* y, void* pooled, hipStream_t st,
                       const long long* acc, const float* acc_scale) {
    if (acc) { if (C & 63) return fail(MPU_EINVAL, "%s", "bn_fold: accumulator mode needs a multiple of 64 channels"); }
    else if (env(ENV_BN_FOLD) == 0 || nblk <= 0 || nblk > BN_FOLD_MAX_ROWS || (nblk & 3) || (C & 63)) return 0;
    const float inv0 = acc ? 1.f / acc_scale[0] : 0.f, inv1 = acc ? 1.f / acc_scale[1] : 0.f;
    if (pooled && ((H | W) & 1)) return 0;
    const long M = (long)B * H * W;
    const long Q = pooled ? M / 4 : M;
    const int nslab = C / 64;
    const int ppp = dtype == MPU_BF16 ? 32 : 16;
    const int maxp = pooled ? 1 : 4;                             // passes a workgroup keeps in registers (the kernel's NP)
    long ppw = (Q * nslab + 255) / 256;                          // about one workgroup per CU, more when the pixels do not fit
    ppw = (ppw + ppp - 1) / ppp * ppp;
    if (ppw < ppp) ppw = ppp;
    if (ppw > (long)maxp * ppp) ppw = (long)maxp * ppp;
    const long groups = (Q + ppw - 1) / ppw;
    if (groups * nslab > (1L << 20)) return 0;
    const dim3 grid((unsigned)(groups * nslab)), blk(256);
#define MPU_BNF(TT)                                                                                                       \
    if (pooled) bn_fold_apply_kernel<TT, true><<<grid, blk, 0, st>>>((const TT*)x, B, H, W, C, partial, nblk, acc, inv0, inv1, M, gamma, beta, mmean, mvar, \
                                                                    mean, invstd, scale, shift, eps, momentum, (TT*)y, (TT*)pooled, nslab, (int)ppw); \
    else bn_fold_apply_kernel<TT, false><<<grid, blk, 0, st>>>((const TT*)x, B, H, W, C, partial, nblk, acc, inv0, inv1, M, gamma, beta, mmean, mvar, \
                                                               mean, invstd, scale, shift, eps, momentum, (TT*)y, nullptr, nslab, (int)ppw);
    if (dtype == MPU_BF16) { MPU_BNF(bf16_t) } else { MPU_BNF(float) }
#undef MPU_BNF
    if (sched_log_on()) sched_note("bn_fold fwd C=%d rows=%d pool=%d grid=%ld", C, nblk, pooled ? 1 : 0, groups * nslab);
    const int rc = launch_ok();
    return rc ? rc : 1;
}

// y = x*scale + shift ; POOL: also 2x2 max of y (taken after the affine: gamma may be negative)
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int B, int H, int W, int C,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       T* __restrict__ y, T* __restrict__ pooled) {
    constexpr int N = Vec<T>::N;
    __shared__ __attribute__((aligned(16))) float kc[2 * COEF_LDS_C];
    const float* lds = nullptr;                              // scale -> kc[0..C), shift -> kc[C..2C)
    if (C <= COEF_LDS_C) {
        for (int i = threadIdx.x; i < C; i += 256) { kc[i] = scale[i]; kc[C + i] = shift[i]; }
        __syncthreads();
        lds = kc;
    }
    const int cpr = C / N;
    if (!POOL) {
        const long total = (long)B * H * W * cpr;
        const long stride = (long)gridDim.x * 256;
        auto one = [&](long e, float (&v)[N]) {
            const int c = (int)(e % cpr);
            float sc[N], sh[N];
            coef_load<N>(lds, scale, c * N, sc);
            if (lds) coef_load<N>(lds, scale, C + c * N, sh); else coef_load<N>(nullptr, shift, c * N, sh);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = bn_affine(v[i], sc[i], sh[i]);
            Vec<T>::store(y + e * N, v);
        };
        long e = (long)blockIdx.x * 256 + threadIdx.x;
        for (; e + stride < total; e += 2 * stride) {
            float v0[N], v1[N];
            Vec<T>::load(x + e * N, v0); Vec<T>::load(x + (e + stride) * N, v1);
            one(e, v0); one(e + stride, v1);
        }
        if (e < total) { float v0[N]; Vec<T>::load(x + e * N, v0); one(e, v0); }
    } else {
        const int Hp = H / 2, Wp = W / 2;
        const long total = (long)B * Hp * Wp * cpr;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const int c = (int)(e % cpr); long t = e / cpr;
            const int px = (int)(t % Wp); t /= Wp;
            const int py = (int)(t % Hp); const int b = (int)(t / Hp);
            float sc[N], sh[N], mx[N], v[4][N];
            long o[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                o[d] = ((((long)b * H + 2 * py + (d >> 1)) * W + 2 * px + (d & 1)) * cpr + c) * N;
                Vec<T>::load(x + o[d], v[d]);
            }
            coef_load<N>(lds, scale, c * N, sc);
            if (lds) coef_load<N>(lds, scale, C + c * N, sh); else coef_load<N>(nullptr, shift, c * N, sh);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int i = 0; i < N; ++i) v[d][i] = bn_affine(v[d][i], sc[i], sh[i]);
                Vec<T>::store(y + o[d], v[d]);
                // the pooled value is the max of the STORED (rounded) values
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float r = to_f32<T>(from_f32<T>(v[d][i]));
                    mx[i] = d == 0 ? r : fmaxf(mx[i], r);
                }
            }
            Vec<T>::store(pooled + e * N, mx);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, int B, int H, int W, int C, T* __restrict__ pooled) {
    constexpr int N = Vec<T>::N;
    const int cpr = C / N, Hp = H / 2, Wp = W / 2;
    const long total = (long)B * Hp * Wp * cpr;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % cpr); long t = e / cpr;
        const int px = (int)(t % Wp); t /= Wp;
        const int py = (int)(t % Hp); const int b = (int)(t / Hp);
        float mx[N];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float v[N];
            Vec<T>::load(x + ((((long)b * H + 2 * py + (d >> 1)) * W + 2 * px + (d & 1)) * cpr + c) * N, v);
#pragma unroll
            for (int i = 0; i < N; ++i) mx[i] = d == 0 ? v[i] : fmaxf(mx[i], v[i]);
        }
        Vec<T>::store(pooled + e * N, mx);
    }
}
int launch_maxpool(int dtype, const void* x, int B, int H, int W, int C, void* pooled, hipStream_t st) {
    const long work = (long)B * (H / 2) * (W / 2) * C / 8;
    if (dtype == MPU_BF16) maxpool_kernel<bf16_t><<<ew_grid(work), 256, 0, st>>>((const bf16_t*)x, B, H, W, C, (bf16_t*)pooled);
    else maxpool_kernel<float><<<ew_grid(work), 256, 0, st>>>((const float*)x, B, H, W, C, (float*)pooled);
    return launch_ok();
}

int launch_bn_apply(int dtype, const void* x, int B, int H, int W, int C, const float* scale, const float* shift,
                    void* y, void* pooled, hipStream_t st) {
    const long work = (long)B * H * W * C / 8 / (pooled ? 4 : 1);
#define MPU_BNA(TT)                                                                                      \
    if (pooled) bn_apply_kernel<TT, true><<<ew_grid(work), 256, 0, st>>>((const TT*)x, B, H, W, C, scale, shift, (TT*)y, (TT*)pooled); \
    else bn_apply_kernel<TT, false><<<ew_grid(work), 256, 0, st>>>((const TT*)x, B, H, W, C, scale, shift, (TT*)y, nullptr);
    if (dtype == MPU_BF16) { MPU_BNA(bf16_t) } else { MPU_BNA(float) }
#undef MPU_BNA
    return launch_ok();
}

#define RC_(expr) do { int rc__ = (expr); if (rc__) return rc__; } while (0)
// BN backward finalize: dgamma, dbeta and the per-channel coefficients of
// dx = k1*dn + k2*x + k3  (dx = scale*(dn - mean(dn) - xhat*mean(dn*xhat)))
// COLMAJOR: partial is [2][C][nblk] (written by a conv epilogue), else [nblk][2][C] (colreduce, maxpool_bwd_add)
template <bool COLMAJOR>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, long M,
                                       const float* gamma, const float* mean, const float* invstd,
                                       float* dgamma, float* dbeta, float* coeffs) {
    __shared__ double red[256];
    const int c = blockIdx.x * FIN_COLS + (COLMAJOR ? threadIdx.x / FIN_KL : threadIdx.x % FIN_COLS);
    double st[2];
    if (COLMAJOR) {
        partial_sums_colmajor<2>(partial, nblk, C, c, c < C, red, st);
        if (c >= C || threadIdx.x % FIN_KL) return;
    } else {
        partial_sums<2>(partial, nblk, 2L * C, C, c, c < C, red, st);
        if (c >= C || threadIdx.x >= FIN_COLS) return;
    }
    const double s = st[0], sx = st[1];
    dgamma[c] = (float)sx; dbeta[c] = (float)s;
    float k1, k2, k3;
    bn_bwd_coeffs(s, sx, M, gamma[c], mean[c], invstd[c], k1, k2, k3);
    coeffs[c] = k1; coeffs[C + c] = k2; coeffs[2 * C + c] = k3;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dn, const T* __restrict__ x, long M, int C,
                                                           const float* __restrict__ k, T* __restrict__ dz) {
    constexpr int N = Vec<T>::N;
    __shared__ __attribute__((aligned(16))) float kc[3 * COEF_LDS_C];
    const float* lds = stage_coeffs<3>(k, C, kc);
    const int cpr = C / N;
    const long total = M * cpr;
    const long stride = (long)gridDim.x * 256;
    auto one = [&](long e, const float* g, const float* v) {
        const int c = (int)(e % cpr);
        float k1[N], k2[N], k3[N], o[N];
        coef_load<N>(lds, k, c * N, k1); coef_load<N>(lds, k, C + c * N, k2); coef_load<N>(lds, k, 2 * C + c * N, k3);
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = v[i] > 0.f ? bn_bwd_affine(g[i], v[i], k1[i], k2[i], k3[i]) : 0.f;
        Vec<T>::store(dz + e * N, o);
    };
    long e = (long)blockIdx.x * 256 + threadIdx.x;
    for (; e + stride < total; e += 2 * stride) {
        float g0[N], v0[N], g1[N], v1[N];
        Vec<T>::load(dn + e * N, g0); Vec<T>::load(x + e * N, v0);
        Vec<T>::load(dn + (e + stride) * N, g1); Vec<T>::load(x + (e + stride) * N, v1);
        one(e, g0, v0); one(e + stride, g1, v1);
    }
    if (e < total) {
        float g0[N], v0[N];
        Vec<T>::load(dn + e * N, g0); Vec<T>::load(x + e * N, v0);
        one(e, g0, v0);
    }
}

// BatchNorm backward with the finalize folded in (see bn_fold_apply_kernel): the producer of dn (a data-gradient epilogue) left
// <= 64 column-major partial rows of (sum dn, sum dn * xhat) per channel
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_fold_kernel(const T* __restrict__ dn, const T* __restrict__ x, long M, int C,
                                                          const float* __restrict__ partial, int nblk,
                                                          const long long* __restrict__ acc, float inv0, float inv1,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                          float* coeffs, T* __restrict__ dz, int nslab, int ppw) {
    constexpr int N = Vec<T>::N, TPP = 64 / N, PPP = 256 / TPP;
    __shared__ double red[2][64];
    __shared__ __attribute__((aligned(16))) float coef[3][64];
    const int slab = (int)blockIdx.x % nslab, g = (int)blockIdx.x / nslab;
    const int c0 = slab * 64, tc = (threadIdx.x % TPP) * N, tp = threadIdx.x / TPP;
    const long q0 = (long)g * ppw;
    const int npass = ppw / PPP;
    constexpr int NP = 4;                                        // passes per workgroup, all requested up front
    float gv[NP][N], xv[NP][N];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const long q = q0 + (long)it * PPP + tp;
        const long qc = (it < npass && q < M) ? q : (q0 < M ? q0 : M - 1);
        Vec<T>::load(dn + qc * C + c0 + tc, gv[it]); Vec<T>::load(x + qc * C + c0 + tc, xv[it]);
    }
    float g_ = 0.f, mu_ = 0.f, is_ = 0.f;
    if (threadIdx.x < 64) { g_ = gamma[c0 + threadIdx.x]; mu_ = mean[c0 + threadIdx.x]; is_ = invstd[c0 + threadIdx.x]; }
    if (threadIdx.x < 128) {
        const int cl = threadIdx.x & 63, st = threadIdx.x >> 6;
        red[st][cl] = acc ? bn_acc_sum(acc, C, st, c0 + cl, st ? inv1 : inv0)
                          : bn_fold_rowsum(partial + ((long)st * C + c0 + cl) * nblk, nblk);
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                      // bn_bwd_finalize_kernel, expression for expression
        const int c = c0 + threadIdx.x;
        const double s = red[0][threadIdx.x], sx = red[1][threadIdx.x];
        float k1f, k2f, k3f;
        bn_bwd_coeffs(s, sx, M, g_, mu_, is_, k1f, k2f, k3f);
        coef[0][threadIdx.x] = k1f; coef[1][threadIdx.x] = k2f; coef[2][threadIdx.x] = k3f;
        if (g == 0) {
            dgamma[c] = (float)sx; dbeta[c] = (float)s;
            coeffs[c] = k1f; coeffs[C + c] = k2f; coeffs[2 * C + c] = k3f;
        }
    }
    __syncthreads();
    float k1[N], k2[N], k3[N];
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        *reinterpret_cast<float4*>(k1 + i) = *reinterpret_cast<const float4*>(&coef[0][tc + i]);
        *reinterpret_cast<float4*>(k2 + i) = *reinterpret_cast<const float4*>(&coef[1][tc + i]);
        *reinterpret_cast<float4*>(k3 + i) = *reinterpret_cast<const float4*>(&coef[2][tc + i]);
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const long q = q0 + (long)it * PPP + tp;
        if (it < npass && q < M) {
            float o[N];
#pragma unroll
            for (int i = 0; i < N; ++i) o[i] = xv[it][i] > 0.f ? bn_bwd_affine(gv[it][i], xv[it][i], k1[i], k2[i], k3[i]) : 0.f;
            Vec<T>::store(dz + q * C + c0 + tc, o);
        }
    }
}

int launch_bn_backward(int dtype, const void* dn, const void* x, long M, int C, float* partial, const float* gamma,
                       const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coeffs, void* dz,
                       int ready_rows, int ready_colmajor, hipStream_t st, long long* acc, const float* acc_scale) {
    int nblk = ready_rows;                       // > 0: the producer of dn already wrote that many partial rows; -1: it added
    int rc = 0;                                  //      its sums to the accumulator `acc` (which is offered, zeroed, whenever non-NULL)
    const bool use_acc = acc && !(C & 63) && (nblk == -1 || nblk <= 0);
    if (nblk == -1 && !use_acc) return fail(MPU_EINVAL, "%s", "bn_backward: accumulator-mode sums without an accumulator");
    if (use_acc && nblk != -1)                   // no producer had the sums: the column reduction adds them to the accumulator itself
        RC_(launch_colreduce<1>(dtype, dn, x, M, C, mean, invstd, partial, &nblk, st, acc, acc_scale));
    if (use_acc || (env(ENV_BN_FOLD) != 0 && ready_colmajor && nblk > 0 && nblk <= BN_FOLD_MAX_ROWS && !(nblk & 3) && !(C & 63))) {
        const int nslab = C / 64, ppp = dtype == MPU_BF16 ? 32 : 16;       // finalize folded into the apply pass
        long ppw = (M * nslab + 255) / 256;
        ppw = (ppw + ppp - 1) / ppp * ppp;
        if (ppw < ppp) ppw = ppp;
        if (ppw > 4L * ppp) ppw = 4L * ppp;                     // (the kernel's NP passes)
        const long groups = (M + ppw - 1) / ppw;
        if (groups * nslab <= (1L << 20) || use_acc) {
            if (groups * nslab > (1L << 20)) return fail(MPU_EUNSUPPORTED, "%s", "bn_backward: tensor too large for the folded kernel");
            const dim3 grid((unsigned)(groups * nslab)), blk(256);
            const long long* ac = use_acc ? acc : nullptr;
            const float inv0 = use_acc ? 1.f / acc_scale[0] : 0.f, inv1 = use_acc ? 1.f / acc_scale[1] : 0.f;
            if (dtype == MPU_BF16)
                bn_bwd_fold_kernel<bf16_t><<<grid, blk, 0, st>>>((const bf16_t*)dn, (const bf16_t*)x, M, C, partial, nblk, ac, inv0, inv1, gamma, mean, invstd,
                                                                 dgamma, dbeta, coeffs, (bf16_t*)dz, nslab, (int)ppw);
            else
                bn_bwd_fold_kernel<float><<<grid, blk, 0, st>>>((const float*)dn, (const float*)x, M, C, partial, nblk, ac, inv0, inv1, gamma, mean, invstd,
                                                                dgamma, dbeta, coeffs, (float*)dz, nslab, (int)ppw);
            if (sched_log_on()) sched_note("bn_fold bwd C=%d rows=%d grid=%ld", C, use_acc ? -1 : nblk, groups * nslab);
            return launch_ok();
        }
    }
    if (nblk <= 0) { ready_colmajor = 0; rc = launch_colreduce<1>(dtype, dn, x, M, C, mean, invstd, partial, &nblk, st); }
    if (rc) return rc;
    if (ready_colmajor)
        bn_bwd_finalize_kernel<true><<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, M, gamma, mean, invstd, dgamma, dbeta, coeffs);
    else
        bn_bwd_finalize_kernel<false><<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, M, gamma, mean, invstd, dgamma, dbeta, coeffs);
    rc = launch_ok();
    if (rc) return rc;
    const long work = M * C / 8 / 2;
    if (dtype == MPU_BF16)
        bn_bwd_apply_kernel<bf16_t><<<ew_grid(work), 256, 0, st>>>((const bf16_t*)dn, (const bf16_t*)x, M, C, coeffs, (bf16_t*)dz);
    else
        bn_bwd_apply_kernel<float><<<ew_grid(work), 256, 0, st>>>((const float*)dn, (const float*)x, M, C, coeffs, (float*)dz);
    return launch_ok();
}

// STATS: also the BatchNorm-backward sums of the level's BN over the STORED dn (sum dn, sum dn * xhat, xhat from the BN
// input x): the grid is sized so that a thread keeps its channel chunk over all its elements ((gridDim.x * 256) % cpr
// == 0); per-block partial row [2][C] in the layout of colreduce<1>, which this replaces for the encoder levels.
// RECOMP (round 6, with STATS): the post-BatchNorm tensor n is NOT read -- its stored values are recomputed from the BatchNorm
// input x the kernel loads anyway (n = rounding of bn_affine(x, scale, shift), the expression and the rounding of the forward
// pass: the same bits, hence the same arg-max) -- and dn is NOT written: maxpool_bwd_bn_fold_kernel below recomputes it.
template <typename T, bool STATS, bool RECOMP = false>
__global__ __launch_bounds__(256) void maxpool_bwd_add_kernel(const T* __restrict__ n, const T* __restrict__ dskip,
                                                              const T* __restrict__ dp, int B, int H, int W, int C,
                                                              T* __restrict__ dn, const T* __restrict__ x,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ partial, long long* acc_out, float acc_scale0,
                                                              float acc_scale1, const float* __restrict__ scale = nullptr,
                                                              const float* __restrict__ shift = nullptr) {
    static_assert(!RECOMP || STATS, "RECOMP needs the BatchNorm input");
    constexpr int N = Vec<T>::N;
    const int cpr = C / N, Hp = H / 2, Wp = W / 2;
    const long total = (long)B * Hp * Wp * cpr;
    float s0[N], s1[N], mu[N], is[N], sc[RECOMP ? N : 1], sh[RECOMP ? N : 1];
    if (STATS) {
        const int c = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cpr);
#pragma unroll
        for (int i = 0; i < N; ++i) { s0[i] = 0.f; s1[i] = 0.f; mu[i] = mean[c * N + i]; is[i] = invstd[c * N + i]; }
        if (RECOMP) {
#pragma unroll
            for (int i = 0; i < N; ++i) { sc[i] = scale[c * N + i]; sh[i] = shift[c * N + i]; }
        }
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % cpr); long t = e / cpr;
        const int px = (int)(t % Wp); t /= Wp;
        const int py = (int)(t % Hp); const int b = (int)(t / Hp);
        float g[N];
        Vec<T>::load(dp + e * N, g);
        float v[4][N];
        long o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            o[d] = ((((long)b * H + 2 * py + (d >> 1)) * W + 2 * px + (d & 1)) * cpr + c) * N;
            if (!RECOMP) Vec<T>::load(n + o[d], v[d]);
        }
        float xv[4][N];
        if (STATS) {
#pragma unroll
            for (int d = 0; d < 4; ++d) Vec<T>::load(x + o[d], xv[d]);
        }
        if (RECOMP) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < N; ++i) v[d][i] = to_f32<T>(from_f32<T>(bn_affine(xv[d][i], sc[i], sh[i])));
        }
        int arg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int a = 0; float m = v[0][i];
#pragma unroll
            for (int d = 1; d < 4; ++d) if (v[d][i] > m) { m = v[d][i]; a = d; }
            arg[i] = a;
        }
        float sk[4][N];                                          // the four skip-gradient pieces: one round trip, not one
        if (dskip) {                                             // per window position (a load inside the loop below is
#pragma unroll                                                   // waited for together with the previous store)
            for (int d = 0; d < 4; ++d) Vec<T>::load(dskip + o[d], sk[d]);
        } else {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int i = 0; i < N; ++i) sk[d][i] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float s[N];
#pragma unroll
            for (int i = 0; i < N; ++i) s[i] = sk[d][i] + (arg[i] == d ? g[i] : 0.f);
            if (!RECOMP) Vec<T>::store(dn + o[d], s);
            if (STATS) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float r = to_f32<T>(from_f32<T>(s[i]));        // the value colreduce would read back
                    s0[i] += r; s1[i] += r * ((xv[d][i] - mu[i]) * is[i]);
                }
            }
        }
    }
    if (STATS) {      // block reduction over the 256 / cpr threads that share a channel chunk (fixed order)
        __shared__ float red[256 * N * 2];
        const int nl = 256 / cpr;
#pragma unroll
        for (int i = 0; i < N; ++i) { red[(threadIdx.x * N + i) * 2] = s0[i]; red[(threadIdx.x * N + i) * 2 + 1] = s1[i]; }
        __syncthreads();
        for (int v = threadIdx.x; v < C * 2; v += 256) {
            const int st2 = v / C, col = v - st2 * C, cg = col / N, i = col - cg * N;
            double acc = 0.0;
            for (int rl = 0; rl < nl; ++rl) acc += (double)red[((rl * cpr + cg) * N + i) * 2 + st2];
            if (acc_out) stats_acc_add(acc_out, C, st2, col, (float)acc, st2 ? acc_scale1 : acc_scale0);
            else partial[((long)blockIdx.x * 2 + st2) * C + col] = (float)acc;
        }
    }
}

int launch_maxpool_bwd_add(int dtype, const void* n, const void* dskip, const void* dp, int B, int H, int W, int C,
                           void* dn, hipStream_t st) {
    const long work = (long)B * (H / 2) * (W / 2) * C / 8;
    if (dtype == MPU_BF16)
        maxpool_bwd_add_kernel<bf16_t, false><<<ew_grid(work), 256, 0, st>>>((const bf16_t*)n, (const bf16_t*)dskip, (const bf16_t*)dp, B, H, W, C, (bf16_t*)dn, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f);
    else
        maxpool_bwd_add_kernel<float, false><<<ew_grid(work), 256, 0, st>>>((const float*)n, (const float*)dskip, (const float*)dp, B, H, W, C, (float*)dn, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f);
    return launch_ok();
}

// Same, plus the BN-backward partial sums of dn against the BN input x (see the kernel). *rows = partial rows written
// ([rows][2][C]); 0 = shape not suited (plain kernel launched, the caller runs the column reduction).
// Workgroups of the pool-backward kernels with fused sums (both forms: the SAME partition of the elements over threads, so that their
// per-thread fp32 partial sums -- and with them every bit downstream -- agree). The kernels hold 147-160 registers: three workgroups
// per CU are resident, a cap of 1024 ran 1.33 rounds (R6ax: levels 0 / 1 41.0 -> 38.9 us, 26.1 -> 23.5 us at 768).
static long pool_bwd_blocks(long work) {
    const long cap = env(ENV_POOL_BWD_BLOCKS) > 0 ? env(ENV_POOL_BWD_BLOCKS) : 3L * device_cu_count();
    const long blocks = (work + 255) / 256;
    return blocks > cap ? cap : blocks;
}

int launch_maxpool_bwd_add_stats(int dtype, const void* n, const void* dskip, const void* dp, int B, int H, int W, int C,
                                 void* dn, const void* x, const float* mean, const float* invstd, float* partial,
                                 long partial_cap, int* rows, hipStream_t st, long long* acc, const float* acc_scale) {
    const bool on = env(ENV_FUSED_BN_BWD) != 0;
    const int N = dtype == MPU_BF16 ? 8 : 4, cpr = C / N;
    const long work = (long)B * (H / 2) * (W / 2) * cpr;
    const long blocks = pool_bwd_blocks(work);
    *rows = 0;
    if (!on || C % N || cpr < 1 || cpr > 256 || 256 % cpr || blocks * 2 * C > partial_cap)
        return launch_maxpool_bwd_add(dtype, n, dskip, dp, B, H, W, C, dn, st);
    if (dtype == MPU_BF16)
        maxpool_bwd_add_kernel<bf16_t, true><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)n, (const bf16_t*)dskip, (const bf16_t*)dp, B, H, W, C, (bf16_t*)dn, (const bf16_t*)x, mean, invstd, partial, acc, acc ? acc_scale[0] : 0.f, acc ? acc_scale[1] : 0.f);
    else
        maxpool_bwd_add_kernel<float, true><<<(unsigned)blocks, 256, 0, st>>>((const float*)n, (const float*)dskip, (const float*)dp, B, H, W, C, (float*)dn, (const float*)x, mean, invstd, partial, acc, acc ? acc_scale[0] : 0.f, acc ? acc_scale[1] : 0.f);
    *rows = acc ? -1 : (int)blocks;
    return launch_ok();
}

// Round 6, second pass of the encoder levels' backward step without the dn tensor: recomputes what maxpool_bwd_add_kernel<RECOMP>
// summed -- dn = rounding of (skip gradient + un-pooled gradient at the arg-max of the recomputed n) -- and applies the BatchNorm
// backward to it: dz = [x > 0] * (k1 * dn + k2 * x + k3), coefficients from the accumulators in the prologue (the arithmetic of
// bn_bwd_fold_kernel, expression for expression: the same bits as the two-tensor form). A workgroup covers whole pixels
// (256 % (C / N) == 0), so its coefficient table holds all C channels: [5][C] floats of LDS.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_bn_fold_kernel(const T* __restrict__ x, const T* __restrict__ dskip,
                                                                  const T* __restrict__ dp, int B, int H, int W, int C,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const long long* __restrict__ acc, float inv0, float inv1, long M,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, float* dgamma, float* dbeta,
                                                                  float* coeffs, T* __restrict__ dz) {
    constexpr int N = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float tab[];          // [5][C]: k1, k2, k3, scale, shift
    for (int c = threadIdx.x; c < C; c += 256) {
        const double s = bn_acc_sum(acc, C, 0, c, inv0), sx = bn_acc_sum(acc, C, 1, c, inv1);
        float k1f, k2f, k3f;
        bn_bwd_coeffs(s, sx, M, gamma[c], mean[c], invstd[c], k1f, k2f, k3f);
        tab[c] = k1f; tab[C + c] = k2f; tab[2 * C + c] = k3f; tab[3 * C + c] = scale[c]; tab[4 * C + c] = shift[c];
        if (blockIdx.x == 0) {
            dgamma[c] = (float)sx; dbeta[c] = (float)s;
            coeffs[c] = k1f; coeffs[C + c] = k2f; coeffs[2 * C + c] = k3f;
        }
    }
    __syncthreads();
    const int cpr = C / N, Hp = H / 2, Wp = W / 2;
    const long total = (long)B * Hp * Wp * cpr;
    const int cc = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cpr);     // (gridDim.x * 256) % cpr == 0: the same chunk for all items
    float k1[N], k2[N], k3[N], sc[N], sh[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        k1[i] = tab[cc * N + i]; k2[i] = tab[C + cc * N + i]; k3[i] = tab[2 * C + cc * N + i];
        sc[i] = tab[3 * C + cc * N + i]; sh[i] = tab[4 * C + cc * N + i];
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % cpr); long t = e / cpr;
        const int px = (int)(t % Wp); t /= Wp;
        const int py = (int)(t % Hp); const int b = (int)(t / Hp);
        float g[N];
        Vec<T>::load(dp + e * N, g);
        float xv[4][N], sk[4][N];
        long o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            o[d] = ((((long)b * H + 2 * py + (d >> 1)) * W + 2 * px + (d & 1)) * cpr + c) * N;
            Vec<T>::load(x + o[d], xv[d]);
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) Vec<T>::load(dskip + o[d], sk[d]);
        int arg[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            int a = 0; float m = to_f32<T>(from_f32<T>(bn_affine(xv[0][i], sc[i], sh[i])));
#pragma unroll
            for (int d = 1; d < 4; ++d) {
                const float v = to_f32<T>(from_f32<T>(bn_affine(xv[d][i], sc[i], sh[i])));
                if (v > m) { m = v; a = d; }
            }
            arg[i] = a;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float q[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const float r = to_f32<T>(from_f32<T>(sk[d][i] + (arg[i] == d ? g[i] : 0.f)));       // the stored dn of the two-tensor form
                q[i] = xv[d][i] > 0.f ? bn_bwd_affine(r, xv[d][i], k1[i], k2[i], k3[i]) : 0.f;
            }
            Vec<T>::store(dz + o[d], q);
        }
    }
}

// Both passes of an encoder level's backward step without the dn tensor (accumulator mode; see the two kernels). 1 = launched,
// 0 = shape not suited (the caller runs launch_maxpool_bwd_add_stats + launch_bn_backward).
int launch_maxpool_bwd_bn(int dtype, const void* dskip, const void* dp, int B, int H, int W, int C, const void* x,
                          const float* mean, const float* invstd, const float* scale, const float* shift, const float* gamma,
                          float* dgamma, float* dbeta, float* coeffs, void* dz, long long* acc, const float* acc_scale, hipStream_t st) {
    const int N = dtype == MPU_BF16 ? 8 : 4, cpr = C / N;
    if (!acc || !dskip || C % N || (C & 63) || cpr < 1 || cpr > 256 || 256 % cpr || (H & 1) || (W & 1) || 5L * C * 4 > 48 * 1024) return 0;
    const long work = (long)B * (H / 2) * (W / 2) * cpr;
    const long blocks = pool_bwd_blocks(work);
    const long M = (long)B * H * W;
    const float inv0 = 1.f / acc_scale[0], inv1 = 1.f / acc_scale[1];
    const unsigned lds = (unsigned)(5L * C * 4);
    if (dtype == MPU_BF16) {
        maxpool_bwd_add_kernel<bf16_t, true, true><<<(unsigned)blocks, 256, 0, st>>>(nullptr, (const bf16_t*)dskip, (const bf16_t*)dp, B, H, W, C, nullptr, (const bf16_t*)x, mean, invstd, nullptr, acc, acc_scale[0], acc_scale[1], scale, shift);
        maxpool_bwd_bn_fold_kernel<bf16_t><<<(unsigned)blocks, 256, lds, st>>>((const bf16_t*)x, (const bf16_t*)dskip, (const bf16_t*)dp, B, H, W, C, scale, shift, acc, inv0, inv1, M, gamma, mean, invstd, dgamma, dbeta, coeffs, (bf16_t*)dz);
    } else {
        maxpool_bwd_add_kernel<float, true, true><<<(unsigned)blocks, 256, 0, st>>>(nullptr, (const float*)dskip, (const float*)dp, B, H, W, C, nullptr, (const float*)x, mean, invstd, nullptr, acc, acc_scale[0], acc_scale[1], scale, shift);
        maxpool_bwd_bn_fold_kernel<float><<<(unsigned)blocks, 256, lds, st>>>((const float*)x, (const float*)dskip, (const float*)dp, B, H, W, C, scale, shift, acc, inv0, inv1, M, gamma, mean, invstd, dgamma, dbeta, coeffs, (float*)dz);
    }
    if (sched_log_on()) sched_note("bn_fold bwd C=%d rows=-1 pool=1 grid=%ld", C, blocks);
    const int rc = launch_ok();
    return rc ? rc : 1;
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* out) {
    __shared__ double red[256];
    colsum_finalize_block(blockIdx.x, partial, nblk, C, out, red);
}
// dtype "bf16x3" weight gradients (round 6): dW = sum_p x[p] dz[p] with x = x_hi + x_lo, dz = dz_hi + dz_lo is, without the
// lo lo term, sum_p x_hi dz_hi + x_lo dz_hi + x_hi dz_lo -- ONE reduction over THREE TIMES the pixels. So the f32 tensors are
// split once into three bf16 planes stacked along the batch axis (x: hi | lo | hi, dz: hi | hi | lo) and the bf16 weight-
// gradient kernels (wgrad_taps / wgrad_glds, grouped launches and all) run unchanged on a batch of 3 B: the products of each
// plane pair are exact in the fp32 accumulators, exactly as the in-register split of the convolution kernels (common.h).
// order 0: hi | lo | hi (the layer's input), 1: hi | hi | lo (dz). n = elements of the f32 tensor.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long n, bf16_t* __restrict__ out, int order) {
    const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;                                         // (n is a multiple of 8: channel-padded tensors)
    const float4 v = *reinterpret_cast<const float4*>(x + i4);
    const uint32_t h0 = f32x2_to_bf16x2(v.x, v.y), h1 = f32x2_to_bf16x2(v.z, v.w);
    const uint32_t l0 = f32x2_to_bf16x2(v.x - __uint_as_float(h0 << 16), v.y - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = f32x2_to_bf16x2(v.z - __uint_as_float(h1 << 16), v.w - __uint_as_float(h1 & 0xffff0000u));
    const uint2 hi = make_uint2(h0, h1), lo = make_uint2(l0, l1);
    *reinterpret_cast<uint2*>(out + i4) = hi;
    *reinterpret_cast<uint2*>(out + n + i4) = order == 0 ? lo : hi;
    *reinterpret_cast<uint2*>(out + 2 * n + i4) = order == 0 ? hi : lo;
}
// every input tensor of a backward pass in ONE launch (the activations are all final when the pass starts): 4096 elements per
// block, four 16-byte loads in flight per thread; the thirteen deep-level tensors of configs[1] no longer cost a launch each
__global__ __launch_bounds__(256) void split3_all_kernel(Split3Table t) {
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.job[j + 1].blk_begin) ++j;
    const float* __restrict__ x = t.job[j].src; bf16_t* __restrict__ out = t.job[j].dst;
    const long n = t.job[j].n; const int order = t.job[j].order;
    const long base = (long)((int)blockIdx.x - t.job[j].blk_begin) * 4096 + threadIdx.x * 4;
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                                // (clamped: n is a multiple of 8, so n - 4 is a valid, aligned index)
        const long i4 = base + u * 1024;
        v[u] = *reinterpret_cast<const float4*>(x + (i4 < n ? i4 : n - 4));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i4 = base + u * 1024;
        if (i4 >= n) continue;
        const uint32_t h0 = f32x2_to_bf16x2(v[u].x, v[u].y), h1 = f32x2_to_bf16x2(v[u].z, v[u].w);
        const uint32_t l0 = f32x2_to_bf16x2(v[u].x - __uint_as_float(h0 << 16), v[u].y - __uint_as_float(h0 & 0xffff0000u));
        const uint32_t l1 = f32x2_to_bf16x2(v[u].z - __uint_as_float(h1 << 16), v[u].w - __uint_as_float(h1 & 0xffff0000u));
        const uint2 hi = make_uint2(h0, h1), lo = make_uint2(l0, l1);
        *reinterpret_cast<uint2*>(out + i4) = hi;
        if (order == 2) { *reinterpret_cast<uint2*>(out + n + i4) = lo; continue; }      // two stored planes (wgrad_taps folds the batch)
        *reinterpret_cast<uint2*>(out + n + i4) = order == 0 ? lo : hi;
        *reinterpret_cast<uint2*>(out + 2 * n + i4) = order == 0 ? hi : lo;
    }
}
int launch_split3_all(Split3Table& t, hipStream_t st) {
    int blocks = 0;
    for (int i = 0; i < t.n; ++i) {
        if (t.job[i].n % 8 || t.job[i].n <= 0) return fail(MPU_EINVAL, "%s", "split3_all: element counts must be positive multiples of 8");
        t.job[i].blk_begin = blocks;
        blocks += (int)((t.job[i].n + 4095) / 4096);
    }
    if (t.n == 0) return MPU_OK;
    split3_all_kernel<<<blocks, 256, 0, st>>>(t);
    return launch_ok();
}
int launch_split3(const float* x, long n, void* out, int order, hipStream_t st) {
    if (n % 4) return fail(MPU_EINVAL, "%s", "split3: element count must be a multiple of 4");
    if (n == 0) return MPU_OK;
    split3_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, st>>>(x, n, (bf16_t*)out, order);
    return launch_ok();
}

// dtype "bf16x3" with the fixed-point accumulators on: ONE pass over the f32 dz of a conv writes its bf16 planes (hi | hi | lo) and adds
// its column sums -- the bias gradient -- to an accumulator of the bn_acc layout (row 0 of each XCD's pair); launch_db_from_acc
// turns the accumulators of all convs queued so far into the f32 bias gradients with one small launch
int launch_split3_colsum(const float* dz, long M, int C, void* planes, long long* acc, float scale, hipStream_t st, int two_planes) {
    if (C % 4 || !acc) return fail(MPU_EINVAL, "%s", "split3_colsum: channel count must be a multiple of 4, accumulator required");
    if (M == 0) return MPU_OK;
    int rpb; const int nblk = red_blocks(M, C, &rpb);
    colreduce_kernel<float, 2, true><<<nblk, CR_THREADS, 0, st>>>(dz, nullptr, M, C, nullptr, nullptr, rpb, nullptr, acc, scale, 0.f,
                                                                  (bf16_t*)planes, two_planes);
    return launch_ok();
}
__global__ __launch_bounds__(256) void db_from_acc_kernel(DbAccTable t) {
    const DbAccJob j = t.job[blockIdx.y];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < j.C) j.db[c] = (float)bn_acc_sum(j.acc, j.C, 0, c, t.inv_scale);
}
int launch_db_from_acc(const DbAccTable& t, hipStream_t st) {
    if (t.n <= 0) return MPU_OK;
    int cmax = 0;
    for (int i = 0; i < t.n; ++i) cmax = t.job[i].C > cmax ? t.job[i].C : cmax;
    db_from_acc_kernel<<<dim3(cdiv(cmax, 256), t.n), 256, 0, st>>>(t);
    return launch_ok();
}

int launch_colsum_finalize(const float* partial, int nblk, int C, float* out, hipStream_t st) {
    colsum_finalize_kernel<<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, out);
    return launch_ok();
}
int launch_colsum(int dtype, const void* dz, long M, int C, float* partial, float* out, hipStream_t st) {
    int nblk;
    int rc = launch_colreduce<2>(dtype, dz, nullptr, M, C, nullptr, nullptr, partial, &nblk, st);
    if (rc) return rc;
    colsum_finalize_kernel<<<cdiv(C, FIN_COLS), 256, 0, st>>>(partial, nblk, C, out);
    return launch_ok();
}

// ------------------------------------------------------------------------- //
// 1x1 head (unet.py:211): K <= 16 classes, direct (HBM-bound) kernels.
// G = C/N lanes cooperate on one pixel (each owns one 16-B chunk of channels).
// ------------------------------------------------------------------------- //
constexpr int HEAD_MAXC = 512;     // channels the head kernels stage in LDS

template <typename T, int K>
__global__ __launch_bounds__(256) void head_forward_kernel(const T* __restrict__ n, long M, int C, const float* __restrict__ Wh,
                                                           int ldw, const float* __restrict__ bh, int softmax,
                                                           float* __restrict__ out) {
    constexpr int N = Vec<T>::N;
    __shared__ float w[HEAD_MAXC * K];
    for (int i = threadIdx.x; i < C * K; i += 256) w[i] = Wh[(i / K) * ldw + (i % K)];
    __syncthreads();
    const int cpr = C / N;
    int G = 1; while (G < cpr && G < 64) G <<= 1;        // lanes per pixel (power of two)
    const int sub = threadIdx.x % G;
    const long ppb = 256 / G;
    const bool single = cpr <= G;                        // one 16-byte chunk per lane: its weights live in registers
    float wr[N][K];
    if (single && sub < cpr) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) wr[i][k] = w[(sub * N + i) * K + k];
    }
    auto finish = [&](long m, float (&z)[K]) {           // the G lanes of a group share m: reduce, bias, softmax, store
        for (int off = G >> 1; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] += __shfl_xor(z[k], off, 64);
        if (sub == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] += bh[k];
            if (softmax) {
                float mx = z[0];
#pragma unroll
                for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - mx); s += z[k]; }
#pragma unroll
                for (int k = 0; k < K; ++k) z[k] = z[k] / s;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) out[m * K + k] = z[k];
        }
    };
    const long stride = (long)gridDim.x * ppb;
    long m = (long)blockIdx.x * ppb + threadIdx.x / G;
    if (single) {
        const bool lane_on = sub < cpr;
        for (; m < M; m += 2 * stride) {                 // two pixels per pass: both loads in flight together
            const long m2 = m + stride;
            const bool on2 = m2 < M;                     // (uniform over the G lanes of a group)
            float va[N], vb[N];
#pragma unroll
            for (int i = 0; i < N; ++i) { va[i] = 0.f; vb[i] = 0.f; }
            if (lane_on) {
                Vec<T>::load(n + m * C + (long)sub * N, va);
                Vec<T>::load(n + (on2 ? m2 : m) * C + (long)sub * N, vb);
            }
            float za[K], zb[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { za[k] = 0.f; zb[k] = 0.f; }
            if (lane_on) {
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int k = 0; k < K; ++k) { za[k] += va[i] * wr[i][k]; zb[k] += vb[i] * wr[i][k]; }
            }
            finish(m, za);
            if (on2) finish(m2, zb);
        }
        return;
    }
    for (; m < M; m += stride) {
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = 0.f;
        for (int c = sub; c < cpr; c += G) {
            float v[N];
            Vec<T>::load(n + m * C + (long)c * N, v);
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int k = 0; k < K; ++k) z[k] += v[i] * w[(c * N + i) * K + k];
        }
        finish(m, z);
    }
}

#define MPU_HEAD_DISPATCH_K(K_, CALL)                                       \
    switch (K_) {                                                           \
        case 1: { constexpr int KK = 1; CALL; } break;                      \
        case 2: { constexpr int KK = 2; CALL; } break;                      \
        case 3: { constexpr int KK = 3; CALL; } break;                      \
        case 4: { constexpr int KK = 4; CALL; } break;                      \
        case 5: { constexpr int KK = 5; CALL; } break;                      \
        case 6: { constexpr int KK = 6; CALL; } break;                      \
        case 7: { constexpr int KK = 7; CALL; } break;                      \
        case 8: { constexpr int KK = 8; CALL; } break;                      \
        default: return fail(MPU_EUNSUPPORTED, "%s", "U-Net head supports 1..8 classes"); \
    }

// head_forward_rs_kernel: same result for C / N == GS lanes per pixel (GS = 8 or 16: a 64-channel last block in bf16 /
// f32) and GS * K <= 64. A group of GS lanes works on GS consecutive pixels per pass: every lane multiplies its 16-byte
// channel chunk of all GS pixels (GS independent loads in flight), the GS x K partial sums are reduce-scattered over the
// group (log2 GS exchange steps; lane `sub` ends up with the complete logits of pixel m0 + sub), and EVERY lane then
// finishes one pixel (bias, softmax, store). In head_forward_kernel the softmax of one pixel occupied all GS lanes
// and only two loads were in flight.
template <typename T, int K, int GS>
__global__ __launch_bounds__(256) void head_forward_rs_kernel(const T* __restrict__ n, long M, int C, const float* __restrict__ Wh,
                                                              int ldw, const float* __restrict__ bh, int softmax,
                                                              float* __restrict__ out) {
    constexpr int N = Vec<T>::N;
    const int sub = threadIdx.x % GS;
    float wr[N][K];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) wr[i][k] = Wh[(long)(sub * N + i) * ldw + k];
    float bias[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bias[k] = bh[k];
    constexpr int PPB = 256;                                     // pixels per block and pass (256 / GS groups x GS pixels)
    for (long m0 = ((long)blockIdx.x * (256 / GS) + threadIdx.x / GS) * GS; m0 < M; m0 += (long)gridDim.x * PPB) {
        const long left = M - m0;
        const int nv = left < GS ? (int)left : GS;
        float z[GS][K];
#pragma unroll
        for (int h = 0; h < GS; h += 4) {                        // four pixels per round: registers for 4 chunks at a time
            float v[4][N];
#pragma unroll
            for (int u = 0; u < 4; ++u)                          // (clamped: surplus pixels re-read the last one, never stored)
                Vec<T>::load(n + (m0 + (h + u < nv ? h + u : nv - 1)) * C + (long)sub * N, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) acc += v[u][i] * wr[i][k];
                    z[h + u][k] = acc;
                }
        }
        // reduce-scatter: after the step with distance s a lane holds the sums over its 2*... partners of the s pixels
        // whose index agrees with its own in the bits above s; at the end z[0] = logits of pixel m0 + sub
#pragma unroll
        for (int s = GS / 2; s >= 1; s >>= 1) {
            const bool upper = (sub & s) != 0;
#pragma unroll
            for (int q = 0; q < s; ++q)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float keep = upper ? z[s + q][k] : z[q][k];
                    const float send = upper ? z[q][k] : z[s + q][k];
                    z[q][k] = keep + __shfl_xor(send, s, 64);
                }
        }
        if (sub < nv) {
            float zz[K];
#pragma unroll
            for (int k = 0; k < K; ++k) zz[k] = z[0][k] + bias[k];
            if (softmax) {
                float mx = zz[0];
#pragma unroll
                for (int k = 1; k < K; ++k) mx = fmaxf(mx, zz[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) { zz[k] = expf(zz[k] - mx); ssum += zz[k]; }
#pragma unroll
                for (int k = 0; k < K; ++k) zz[k] = zz[k] / ssum;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) out[(m0 + sub) * K + k] = zz[k];
        }
    }
}

int launch_head_forward(int dtype, const void* n, long M, int C, int K, const float* Wh, int ldw, const float* bh,
                        int softmax, float* out, hipStream_t st) {
    if (C > HEAD_MAXC) return fail(MPU_EUNSUPPORTED, "%s", "head: more than 512 input channels");
    const int N = dtype == MPU_BF16 ? 8 : 4;
    int G = 1; while (G < C / N && G < 64) G <<= 1;
    const long ppb = 256 / G;
    long blocks = (M + ppb - 1) / ppb; if (blocks > 4096) blocks = 4096;
    const bool rs_on = env(ENV_HEAD_RS) != 0;
    if (rs_on && C == 64 && K * (C / N) <= 64) {                 // reduce-scatter variant: every lane finishes a pixel
        long rb = (M + 255) / 256; if (rb > 4096) rb = 4096;
        if (dtype == MPU_BF16) {
            MPU_HEAD_DISPATCH_K(K, (head_forward_rs_kernel<bf16_t, KK, 8><<<(unsigned)rb, 256, 0, st>>>((const bf16_t*)n, M, C, Wh, ldw, bh, softmax, out)))
        } else {
            MPU_HEAD_DISPATCH_K(K, (head_forward_rs_kernel<float, (KK <= 4 ? KK : 1), 16><<<(unsigned)rb, 256, 0, st>>>((const float*)n, M, C, Wh, ldw, bh, softmax, out)))
        }
        return launch_ok();
    }
    if (dtype == MPU_BF16) {
        MPU_HEAD_DISPATCH_K(K, (head_forward_kernel<bf16_t, KK><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)n, M, C, Wh, ldw, bh, softmax, out)))
    } else {
        MPU_HEAD_DISPATCH_K(K, (head_forward_kernel<float, KK><<<(unsigned)blocks, 256, 0, st>>>((const float*)n, M, C, Wh, ldw, bh, softmax, out)))
    }
    return launch_ok();
}

// Second stage of the head fused into the last conv's epilogue (conv_ws, ConvArgs.head_partial): the two channel halves of
// the logits, + bias, softmax (or linear). 12-36 bytes per pixel in, K floats out.
template <int K>
__global__ __launch_bounds__(256) void head_combine_kernel(const float* __restrict__ partial, long M, const float* __restrict__ bh,
                                                           int softmax, float* __restrict__ out) {
    float bias[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bias[k] = bh[k];
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = (partial[m * K + k] + partial[(M + m) * K + k]) + bias[k];
        if (softmax) {
            float mx = z[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
            float ssum = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - mx); ssum += z[k]; }
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = z[k] / ssum;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) out[m * K + k] = z[k];
    }
}
int launch_head_combine(const float* partial, long M, int K, const float* bh, int softmax, float* out, hipStream_t st) {
    long rb = (M + 255) / 256; if (rb > 8192) rb = 8192;
    MPU_HEAD_DISPATCH_K(K, (head_combine_kernel<KK><<<(unsigned)rb, 256, 0, st>>>(partial, M, bh, softmax, out)))
    return launch_ok();
}

// Gradient of the Keras sparse CE on clipped probabilities through the softmax
// (oracle/unet_ref.py keras_sparse_ce). Per pixel:
//   q = clip(p, eps, 1-eps); S = sum q; L = (-log q_y + log S) * w
//   g_k = 1[eps <= p_k <= 1-eps] * (-[k==y]/q_y + 1/S) * w ;  dz_j = p_j (g_j - sum_k g_k p_k)
// then dn = dz @ Wh^T, dWh += n^T dz, dbh += dz.
// partial layout: [nblk][C*K + K + 1]: head weight gradient, head bias gradient, and (round 6) the block's sum of the weighted
// per-pixel loss -- head_bwd_finalize_kernel turns the last column into the step's MEAN loss (a device scalar in the workspace:
// what `mp train` accumulates per step; a torch reduction of the 1-MB loss tensor inside the captured graph went stale for
// stretches of replays in the bf16x3 graph -- its cross-block semaphore logic, gpurun R6al -- and costs three more launches)
template <typename T, int K>
__global__ __launch_bounds__(256, (K <= 3 ? 4 : 2)) void head_backward_kernel(const T* __restrict__ n, const float* __restrict__ probs,
                                                            const uint8_t* __restrict__ y, const float* __restrict__ sw,
                                                            long M, long ppi, int C, const float* __restrict__ Wh, int ldw,
                                                            float* __restrict__ partial, T* __restrict__ dn,
                                                            float* __restrict__ loss) {
    constexpr int N = Vec<T>::N;
    constexpr float EPS = 1e-7f;
    __shared__ float w[HEAD_MAXC * K];
    __shared__ float red[HEAD_MAXC * K + K + 1];
    for (int i = threadIdx.x; i < C * K; i += 256) w[i] = Wh[(i / K) * ldw + (i % K)];
    for (int i = threadIdx.x; i < C * K + K + 1; i += 256) red[i] = 0.f;
    __syncthreads();
    const int cpr = C / N;           // host guarantees cpr <= 64
    int G = 1; while (G < cpr) G <<= 1;
    const int sub = threadIdx.x % G;
    const bool act = sub < cpr;
    const int ppb = 256 / G;                                     // pixel groups per block
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1);  // first lane of this lane's group
    // A group of G lanes works on G consecutive pixels per pass. The per-pixel part (clipped probabilities, CE
    // gradient, loss) is computed ONCE, by lane `sub` for pixel m0 + sub; every lane then handles its 16-byte channel
    // chunk of all G pixels, fetching each pixel's dz from the lane that owns it. (One pixel per group and pass made
    // all G lanes repeat the per-pixel part: the kernel was bound by that vector-ALU work, not by its 67 MB.)
    constexpr int JR = K <= 3 ? 2 : 4;                           // (K <= 3 is held to 128 registers: 4 waves per SIMD)
    float wr[N][K];                                              // this lane's rows of Wh
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) wr[i][k] = act ? w[(sub * N + i) * K + k] : 0.f;
    float aw[N][K], ab[K];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) aw[i][k] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) ab[k] = 0.f;
    float lsum = 0.f;                                            // this lane's sum of the weighted per-pixel loss
    for (long m0 = ((long)blockIdx.x * ppb + threadIdx.x / G) * G; m0 < M; m0 += (long)gridDim.x * ppb * G) {
        const long mm = m0 + sub;
        float dzv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dzv[k] = 0.f;
        if (mm < M) {
            float p[K], g[K];
            const int yy = y[mm];
            const float wt = sw[mm / ppi];
            float S = 0.f, qy = 1.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                p[k] = probs[mm * K + k];
                const float q = fminf(fmaxf(p[k], EPS), 1.f - EPS);
                S += q;
                if (k == yy) qy = q;
            }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const bool pass = p[k] >= EPS && p[k] <= 1.f - EPS;
                g[k] = pass ? ((k == yy ? -1.f / qy : 0.f) + 1.f / S) * wt : 0.f;
                dot += g[k] * p[k];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) { dzv[k] = p[k] * (g[k] - dot); ab[k] += dzv[k]; }
            const float lv = (-logf(qy) + logf(S)) * wt;
            lsum += lv;
            if (loss) loss[mm] = lv;
        }
        const long left = M - m0;
        const int nv = left < G ? (int)left : G;                 // pixels of this pass (uniform over the group)
        // (requesting the chunks one round ahead / before the per-pixel part measured the same: 17.8 us at configs[1],
        //  67 MB -- the kernel sits at 3.9 TB/s against 5.3 TB/s of the plain streaming kernels)
        for (int j0 = 0; j0 < nv; j0 += JR) {                    // JR pixels per round: their loads go out together
            float v[JR][N];
            if (act) {
#pragma unroll
                for (int u = 0; u < JR; ++u)                     // (clamped: the surplus ones re-read the last pixel)
                    Vec<T>::load(n + (m0 + (j0 + u < nv ? j0 + u : nv - 1)) * C + (long)sub * N, v[u]);
            }
#pragma unroll
            for (int u = 0; u < JR; ++u) {
                const bool valid = j0 + u < nv;
                float dj[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float t = __shfl(dzv[k], gbase + ((j0 + u) & (G - 1)), 64);
                    dj[k] = valid ? t : 0.f;
                }
                if (act) {
                    float d[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            acc += dj[k] * wr[i][k];
                            aw[i][k] += v[u][i] * dj[k];
                        }
                        d[i] = acc;
                    }
                    if (valid) Vec<T>::store(dn + (m0 + j0 + u) * C + (long)sub * N, d);
                }
            }
        }
    }
    // block reduction, fixed order: butterfly over the groups of a wave (lanes that share `sub` are G apart); the
    // bias gradient additionally over the lanes of a group (each lane summed its own pixels); then the four waves
    // one after the other through LDS
    for (int o = G; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) aw[i][k] += __shfl_xor(aw[i][k], o, 64);
    }
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) ab[k] += __shfl_xor(ab[k], o, 64);
        lsum += __shfl_xor(lsum, o, 64);
    }
    for (int wv = 0; wv < 4; ++wv) {                             // G <= 64: lanes 0..G-1 of each wave hold its totals
        const bool mine = (int)(threadIdx.x >> 6) == wv;
        if (mine && act && lane < G) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int k = 0; k < K; ++k) red[(sub * N + i) * K + k] += aw[i][k];
        }
        if (mine && lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[C * K + k] += ab[k];
            red[C * K + K] += lsum;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < C * K + K + 1; i += 256) partial[(long)blockIdx.x * (C * K + K + 1) + i] = red[i];
}

__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int K, int ldw,
                                         float* dWh, float* dbh, long long* step_incr, float* loss_mean, double inv_m) {
    __shared__ double red[256];
    // (mpu_unet_backward_adam: the optimizer's device step counter moves HERE, at the start of the backward pass -- nothing
    // reads it before the optimizer kernels at the pass's end, and none of those then has to be the last reader)
    if (step_incr && blockIdx.x == 0 && threadIdx.x == 0) *step_incr += 1;
    const int i = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    const int tot = C * K + K + 1;
    double s;
    partial_sums<1>(partial, nblk, tot, 0, i, i < tot, red, &s);
    if (i >= tot || threadIdx.x >= FIN_COLS) return;
    if (i < C * K) dWh[(i / K) * ldw + (i % K)] = (float)s;
    else if (i < C * K + K) dbh[i - C * K] = (float)s;
    else if (loss_mean) *loss_mean = (float)(s * inv_m);         // mean over the B*H*W pixels of the weighted per-pixel loss
}

int launch_head_backward(int dtype, const void* n, const float* probs, const uint8_t* y, const float* sw, long M,
                         long ppi, int C, int K, const float* Wh, int ldw, float* partial, void* dn, float* dWh,
                         float* dbh, float* loss, hipStream_t st, long long* step_incr, float* loss_mean) {
    const int N = dtype == MPU_BF16 ? 8 : 4;
    const int cpr = C / N;
    if (C > HEAD_MAXC || cpr > 64 || C % N != 0)
        return fail(MPU_EUNSUPPORTED, "%s", "head backward: at most 64 16-byte channel chunks");
    int G = 1; while (G < cpr) G <<= 1;
    const long ppb = 256;                        // pixels per block and pass (256 / G groups of G pixels)
    long blocks = (M + ppb - 1) / ppb; if (blocks > HEAD_BWD_MAX_BLOCKS) blocks = HEAD_BWD_MAX_BLOCKS;
    if (dtype == MPU_BF16) {
        MPU_HEAD_DISPATCH_K(K, (head_backward_kernel<bf16_t, KK><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)n, probs, y, sw, M, ppi, C, Wh, ldw, partial, (bf16_t*)dn, loss)))
    } else {
        MPU_HEAD_DISPATCH_K(K, (head_backward_kernel<float, KK><<<(unsigned)blocks, 256, 0, st>>>((const float*)n, probs, y, sw, M, ppi, C, Wh, ldw, partial, (float*)dn, loss)))
    }
    int rc = launch_ok();
    if (rc) return rc;
    head_bwd_finalize_kernel<<<cdiv(C * K + K + 1, FIN_COLS), 256, 0, st>>>(partial, (int)blocks, C, K, ldw, dWh, dbh, step_incr, loss_mean,
                                                                            1.0 / (double)M);
    return launch_ok();
}

// ------------------------------------------------------------------------------------------------------------------- //
// Round 6: the training step's head WITHOUT the post-BatchNorm tensor of the last block (bf16, 64 channels, accumulator mode).
// The unfused chain around the head is six HBM passes over 33.5-MB tensors at configs[1] -- BN apply (c3 -> n2), head forward
// (n2 -> probs), head backward (n2, probs -> dn2 + head gradients), column reduction (dn2, c3 -> BatchNorm-backward sums),
// BN backward (dn2, c3 -> dz3) -- although n2 and dn2 are functions of c3 and of K numbers per pixel:
//   n2[m][c] = bf16(scale_c * c3[m][c] + shift_c),     dn2[m][c] = sum_k dzh[m][k] * Wh[c][k]   (dzh: the CE gradient at the logits)
// so that  sum_m dn2[m][c]          = sum_k Wh[c][k] * dbh[k]
//          sum_m dn2[m][c] * xhat_c = sum_k Wh[c][k] * T[c][k],   T[c][k] = sum_m xhat[m][c] * dzh[m][k]
//          dWh[c][k] = sum_m n2[m][c] * dzh[m][k] = gamma_c * T[c][k] + beta_c * dbh[k]        (n2 before its bf16 rounding)
// Three passes over c3 remain: head_bn_forward (fold of the BatchNorm statistics + affine + 1x1 + softmax -> probs; the logits
// are the unfused path's bit for bit: the affine result takes the same bf16 rounding the stored n2 had), head_bn_backward (probs,
// labels, c3 -> T, dbh, loss: partial rows, summed by head_bwd_finalize_kernel) and head_bn_bwd_apply (recomputes dzh and dn2 per
// pixel, forms the BatchNorm-backward coefficients from T / dbh / Wh in its prologue, writes dz3; its first workgroup writes
// dgamma, dbeta and dWh). dn2 stays in fp32 registers (the unfused path rounds it to bf16 in HBM), n2 in dWh is unrounded:
// both closer to the fp64 oracle. Switch MPU_HEAD_TRAIN_FUSED=0; not taken while a launch tap is installed (the replay tests
// check the unfused kernels launch by launch).
// ------------------------------------------------------------------------------------------------------------------- //
// CE gradient at the logits of one pixel (head_backward_kernel's per-pixel part, expression for expression)
template <int K>
__device__ __forceinline__ void head_ce_grad(const float* __restrict__ probs, const uint8_t* __restrict__ y, const float* __restrict__ sw,
                                             long mm, long ppi, float (&dzv)[K], float* lv) {
    constexpr float EPS = 1e-7f;
    float p[K], g[K];
    const int yy = y[mm];
    const float wt = sw[mm / ppi];
    float S = 0.f, qy = 1.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        p[k] = probs[mm * K + k];
        const float q = fminf(fmaxf(p[k], EPS), 1.f - EPS);
        S += q;
        if (k == yy) qy = q;
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool pass = p[k] >= EPS && p[k] <= 1.f - EPS;
        g[k] = pass ? ((k == yy ? -1.f / qy : 0.f) + 1.f / S) * wt : 0.f;
        dot += g[k] * p[k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) dzv[k] = p[k] * (g[k] - dot);
    if (lv) *lv = (-logf(qy) + logf(S)) * wt;
}

template <typename T, int K>
__global__ __launch_bounds__(256) void head_bn_forward_kernel(const T* __restrict__ x, long M, const long long* __restrict__ acc,
                                                              float inv0, float inv1, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* mmean, float* mvar, float* mean,
                                                              float* invstd, float* scale, float* shift, float eps, float mom,
                                                              const float* __restrict__ Wh, int ldw, const float* __restrict__ bh,
                                                              int softmax, float* __restrict__ out) {
    constexpr int N = Vec<T>::N, GS = 64 / N, C = 64;            // 8 lanes x 8 channels (bf16) / 16 lanes x 4 channels (f32 storage)
    static_assert(GS * K <= 64, "reduce-scatter group");
    constexpr int PPB = 256;
    __shared__ double red[2][64];
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    const int sub = threadIdx.x % GS;
    float g_ = 0.f, b_ = 0.f, mm_ = 0.f, mv_ = 0.f;
    if (threadIdx.x < 64) { g_ = gamma[threadIdx.x]; b_ = beta[threadIdx.x]; mm_ = mmean[threadIdx.x]; mv_ = mvar[threadIdx.x]; }
    if (threadIdx.x < 128) {
        const int cl = threadIdx.x & 63, st = threadIdx.x >> 6;
        red[st][cl] = bn_acc_sum(acc, C, st, cl, st ? inv1 : inv0);
    }
    float wr[N][K];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) wr[i][k] = Wh[(long)(sub * N + i) * ldw + k];
    float bias[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bias[k] = bh[k];
    __syncthreads();
    if (threadIdx.x < 64) {                                      // bn_fold_apply_kernel's prologue, expression for expression
        const BnFwdCoef kf = bn_fwd_coeffs(red[0][threadIdx.x], red[1][threadIdx.x], M, g_, b_, mm_, mv_, eps, mom);
        coef[0][threadIdx.x] = kf.scale; coef[1][threadIdx.x] = kf.shift;
        if (blockIdx.x == 0) {
            const int c = threadIdx.x;
            mean[c] = kf.mean; invstd[c] = kf.invstd; scale[c] = kf.scale; shift[c] = kf.shift;
            mmean[c] = kf.mmean; mvar[c] = kf.mvar;
        }
    }
    __syncthreads();
    float sc[N], sh[N];
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        *reinterpret_cast<float4*>(sc + i) = *reinterpret_cast<const float4*>(&coef[0][sub * N + i]);
        *reinterpret_cast<float4*>(sh + i) = *reinterpret_cast<const float4*>(&coef[1][sub * N + i]);
    }
    // (requesting the first pass's chunks before the statistics prologue measured the same, 10.7 / 11.0 us at configs[1]: R6ax)
    for (long m0 = ((long)blockIdx.x * (256 / GS) + threadIdx.x / GS) * GS; m0 < M; m0 += (long)gridDim.x * PPB) {
        const long left = M - m0;
        const int nv = left < GS ? (int)left : GS;
        float z[GS][K];
#pragma unroll
        for (int h = 0; h < GS; h += 4) {
            float v[4][N];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                Vec<T>::load(x + (m0 + (h + u < nv ? h + u : nv - 1)) * C + (long)sub * N, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < N; ++i)                      // the stored n2: affine, then the rounding of the store
                    v[u][i] = to_f32<T>(from_f32<T>(bn_affine(v[u][i], sc[i], sh[i])));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) a += v[u][i] * wr[i][k];
                    z[h + u][k] = a;
                }
            }
        }
#pragma unroll
        for (int s = GS / 2; s >= 1; s >>= 1) {
            const bool upper = (sub & s) != 0;
#pragma unroll
            for (int q = 0; q < s; ++q)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float keep = upper ? z[s + q][k] : z[q][k];
                    const float send = upper ? z[q][k] : z[s + q][k];
                    z[q][k] = keep + __shfl_xor(send, s, 64);
                }
        }
        if (sub < nv) {
            float zz[K];
#pragma unroll
            for (int k = 0; k < K; ++k) zz[k] = z[0][k] + bias[k];
            if (softmax) {
                float mx = zz[0];
#pragma unroll
                for (int k = 1; k < K; ++k) mx = fmaxf(mx, zz[k]);
                float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) { zz[k] = expf(zz[k] - mx); ssum += zz[k]; }
#pragma unroll
                for (int k = 0; k < K; ++k) zz[k] = zz[k] / ssum;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) out[(m0 + sub) * K + k] = zz[k];
        }
    }
}

// partial layout as head_backward_kernel's: [nblk][64 * K + K + 1] = T, dbh, sum of the weighted per-pixel loss
template <typename T, int K>
__global__ __launch_bounds__(256, (K <= 4 ? 4 : 2)) void head_bn_backward_kernel(const T* __restrict__ x, const float* __restrict__ probs,
                                                                  const uint8_t* __restrict__ y, const float* __restrict__ sw, long M,
                                                                  long ppi, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, float* __restrict__ partial,
                                                                  float* __restrict__ loss) {
    constexpr int N = Vec<T>::N, G = 64 / N, C = 64, JR = 2;
    __shared__ float red[C * K + K + 1];
    for (int i = threadIdx.x; i < C * K + K + 1; i += 256) red[i] = 0.f;
    const int sub = threadIdx.x % G;
    const int ppb = 256 / G;
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1);
    float mu[N], is[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { mu[i] = mean[sub * N + i]; is[i] = invstd[sub * N + i]; }
    float at[N][K], ab[K];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int k = 0; k < K; ++k) at[i][k] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) ab[k] = 0.f;
    float lsum = 0.f;
    __syncthreads();
    for (long m0 = ((long)blockIdx.x * ppb + threadIdx.x / G) * G; m0 < M; m0 += (long)gridDim.x * ppb * G) {
        const long mm = m0 + sub;
        float dzv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dzv[k] = 0.f;
        if (mm < M) {
            float lv;
            head_ce_grad<K>(probs, y, sw, mm, ppi, dzv, &lv);
#pragma unroll
            for (int k = 0; k < K; ++k) ab[k] += dzv[k];
            lsum += lv;
            if (loss) loss[mm] = lv;
        }
        const long left = M - m0;
        const int nv = left < G ? (int)left : G;
        // (all chunks of a pass requested before the per-pixel part, fully unrolled: 5-67 spilled registers at K = 3-4; this form: 92)
        for (int j0 = 0; j0 < nv; j0 += JR) {
            float v[JR][N];
#pragma unroll
            for (int u = 0; u < JR; ++u)
                Vec<T>::load(x + (m0 + (j0 + u < nv ? j0 + u : nv - 1)) * C + (long)sub * N, v[u]);
#pragma unroll
            for (int u = 0; u < JR; ++u) {
                const bool valid = j0 + u < nv;
                float dj[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float t = __shfl(dzv[k], gbase + ((j0 + u) & (G - 1)), 64);
                    dj[k] = valid ? t : 0.f;
                }
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const float xh = (v[u][i] - mu[i]) * is[i];
#pragma unroll
                    for (int k = 0; k < K; ++k) at[i][k] += xh * dj[k];
                }
            }
        }
    }
    for (int o = G; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) at[i][k] += __shfl_xor(at[i][k], o, 64);
    }
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) ab[k] += __shfl_xor(ab[k], o, 64);
        lsum += __shfl_xor(lsum, o, 64);
    }
    for (int wv = 0; wv < 4; ++wv) {
        const bool mine = (int)(threadIdx.x >> 6) == wv;
        if (mine && lane < G) {
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
                for (int k = 0; k < K; ++k) red[(sub * N + i) * K + k] += at[i][k];
        }
        if (mine && lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[C * K + k] += ab[k];
            red[C * K + K] += lsum;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < C * K + K + 1; i += 256) partial[(long)blockIdx.x * (C * K + K + 1) + i] = red[i];
}

template <typename T, int K>
__global__ __launch_bounds__(256, (K <= 4 ? 4 : 2)) void head_bn_bwd_apply_kernel(const T* __restrict__ x, const float* __restrict__ probs,
                                                                   const uint8_t* __restrict__ y, const float* __restrict__ sw, long M,
                                                                   long ppi, const float* __restrict__ Wh, int ldw,
                                                                   const float* __restrict__ Tsum, const float* __restrict__ dbh,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                   float* dgamma, float* dbeta, float* dWh, float* coeffs,
                                                                   T* __restrict__ dz) {
    constexpr int N = Vec<T>::N, G = 64 / N, C = 64, JR = 2;
    __shared__ __attribute__((aligned(16))) float coef[3][64];
    __shared__ float w[C * K];
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        const float g_ = gamma[c], b_ = beta[c];
        double s0 = 0.0, s1 = 0.0;
        float tr[K], db[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float wk = Wh[(long)c * ldw + k];
            w[c * K + k] = wk; tr[k] = Tsum[c * K + k]; db[k] = dbh[k];
            s0 += (double)wk * (double)db[k]; s1 += (double)wk * (double)tr[k];
        }
        float k1f, k2f, k3f;
        bn_bwd_coeffs(s0, s1, M, g_, mean[c], invstd[c], k1f, k2f, k3f);
        coef[0][c] = k1f; coef[1][c] = k2f; coef[2][c] = k3f;
        if (blockIdx.x == 0) {
            dgamma[c] = (float)s1; dbeta[c] = (float)s0;
            coeffs[c] = k1f; coeffs[C + c] = k2f; coeffs[2 * C + c] = k3f;
#pragma unroll
            for (int k = 0; k < K; ++k) dWh[(long)c * ldw + k] = (float)((double)g_ * (double)tr[k] + (double)b_ * (double)db[k]);
        }
    }
    __syncthreads();
    const int sub = threadIdx.x % G;
    const int ppb = 256 / G;
    const int lane = threadIdx.x & 63, gbase = lane & ~(G - 1);
    float wr[N][K], k1[N], k2[N], k3[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < K; ++k) wr[i][k] = w[(sub * N + i) * K + k];
        k1[i] = coef[0][sub * N + i]; k2[i] = coef[1][sub * N + i]; k3[i] = coef[2][sub * N + i];
    }
    for (long m0 = ((long)blockIdx.x * ppb + threadIdx.x / G) * G; m0 < M; m0 += (long)gridDim.x * ppb * G) {
        const long mm = m0 + sub;
        float dzv[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dzv[k] = 0.f;
        if (mm < M) head_ce_grad<K>(probs, y, sw, mm, ppi, dzv, nullptr);
        const long left = M - m0;
        const int nv = left < G ? (int)left : G;
        for (int j0 = 0; j0 < nv; j0 += JR) {
            float v[JR][N];
#pragma unroll
            for (int u = 0; u < JR; ++u)
                Vec<T>::load(x + (m0 + (j0 + u < nv ? j0 + u : nv - 1)) * C + (long)sub * N, v[u]);
#pragma unroll
            for (int u = 0; u < JR; ++u) {
                const bool valid = j0 + u < nv;
                float dj[K];
#pragma unroll
                for (int k = 0; k < K; ++k) dj[k] = __shfl(dzv[k], gbase + ((j0 + u) & (G - 1)), 64);
                float o[N];
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    float dn = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) dn += dj[k] * wr[i][k];
                    o[i] = v[u][i] > 0.f ? bn_bwd_affine(dn, v[u][i], k1[i], k2[i], k3[i]) : 0.f;
                }
                if (valid) Vec<T>::store(dz + (m0 + j0 + u) * C + (long)sub * N, o);
            }
        }
    }
}

// bf16 storage: 8 lanes per pixel, 1..8 classes; f32 storage (dtype "bf16x3"): 16 lanes per pixel, the reduce-scatter needs 16 * K <= 64
bool head_train_fused_shape_ok(int dtype, int C, int K) {
    return C == 64 && K >= 1 && ((dtype == MPU_BF16 && K <= 8) || (dtype == MPU_F32 && K <= 4));
}
#define MPU_HEAD_DISPATCH_K4(K_, CALL)                                      \
    switch (K_) {                                                           \
        case 1: { constexpr int KK = 1; CALL; } break;                      \
        case 2: { constexpr int KK = 2; CALL; } break;                      \
        case 3: { constexpr int KK = 3; CALL; } break;                      \
        case 4: { constexpr int KK = 4; CALL; } break;                      \
        default: return fail(MPU_EUNSUPPORTED, "%s", "fused training head in f32 storage: 1..4 classes"); \
    }

int launch_head_bn_forward(int dtype, const void* x, long M, const long long* acc, const float* acc_scale, const float* gamma, const float* beta,
                           float* mmean, float* mvar, float* mean, float* invstd, float* scale, float* shift, float eps, float momentum,
                           int K, const float* Wh, int ldw, const float* bh, int softmax, float* out, hipStream_t st) {
    long rb = (M + 255) / 256; if (rb > 4096) rb = 4096;
    const float inv0 = 1.f / acc_scale[0], inv1 = 1.f / acc_scale[1];
    if (dtype == MPU_BF16) {
        MPU_HEAD_DISPATCH_K(K, (head_bn_forward_kernel<bf16_t, KK><<<(unsigned)rb, 256, 0, st>>>((const bf16_t*)x, M, acc, inv0, inv1, gamma, beta, mmean, mvar,
                                                                                                 mean, invstd, scale, shift, eps, momentum, Wh, ldw, bh, softmax, out)))
    } else {
        MPU_HEAD_DISPATCH_K4(K, (head_bn_forward_kernel<float, KK><<<(unsigned)rb, 256, 0, st>>>((const float*)x, M, acc, inv0, inv1, gamma, beta, mmean, mvar,
                                                                                                 mean, invstd, scale, shift, eps, momentum, Wh, ldw, bh, softmax, out)))
    }
    if (sched_log_on()) sched_note("bn_fold fwd C=64 rows=-1 pool=0 head=1 grid=%ld", rb);
    return launch_ok();
}

// T (-> tsum [64][K]), dbh, mean loss; the device step counter moves in the finalizer as in launch_head_backward
int launch_head_bn_backward(int dtype, const void* x, const float* probs, const uint8_t* y, const float* sw, long M, long ppi, int K,
                            const float* mean, const float* invstd, float* partial, float* tsum, float* dbh, float* loss,
                            hipStream_t st, long long* step_incr, float* loss_mean) {
    long blocks = (M + 255) / 256; if (blocks > HEAD_BWD_MAX_BLOCKS) blocks = HEAD_BWD_MAX_BLOCKS;
    if (dtype == MPU_BF16) {
        MPU_HEAD_DISPATCH_K(K, (head_bn_backward_kernel<bf16_t, KK><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)x, probs, y, sw, M, ppi, mean, invstd, partial, loss)))
    } else {
        MPU_HEAD_DISPATCH_K4(K, (head_bn_backward_kernel<float, KK><<<(unsigned)blocks, 256, 0, st>>>((const float*)x, probs, y, sw, M, ppi, mean, invstd, partial, loss)))
    }
    int rc = launch_ok();
    if (rc) return rc;
    head_bwd_finalize_kernel<<<cdiv(64 * K + K + 1, FIN_COLS), 256, 0, st>>>(partial, (int)blocks, 64, K, K, tsum, dbh, step_incr, loss_mean,
                                                                             1.0 / (double)M);
    return launch_ok();
}

int launch_head_bn_bwd_apply(int dtype, const void* x, const float* probs, const uint8_t* y, const float* sw, long M, long ppi, int K,
                             const float* Wh, int ldw, const float* tsum, const float* dbh, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dWh, float* coeffs, void* dz,
                             hipStream_t st) {
    long blocks = (M + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (dtype == MPU_BF16) {
        MPU_HEAD_DISPATCH_K(K, (head_bn_bwd_apply_kernel<bf16_t, KK><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)x, probs, y, sw, M, ppi, Wh, ldw, tsum, dbh, gamma,
                                                                                                       beta, mean, invstd, dgamma, dbeta, dWh, coeffs, (bf16_t*)dz)))
    } else {
        MPU_HEAD_DISPATCH_K4(K, (head_bn_bwd_apply_kernel<float, KK><<<(unsigned)blocks, 256, 0, st>>>((const float*)x, probs, y, sw, M, ppi, Wh, ldw, tsum, dbh, gamma,
                                                                                                       beta, mean, invstd, dgamma, dbeta, dWh, coeffs, (float*)dz)))
    }
    if (sched_log_on()) sched_note("bn_fold bwd C=64 rows=-1 head=1 grid=%ld", blocks);
    return launch_ok();
}

// TF ApplyAdam: m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps)
// (one definition for the plain and the fused kernels: the same instruction sequence, bit-identical results)
// No FMA contraction inside (pragma): the compiler contracted the multiply-adds in one kernel and not in the other, and
// the fused and the plain path differed by one ulp in 1.4 % of the first moments (HIP's __fadd_rn & co. are plain
// operators and contract just the same).
__device__ __forceinline__ void adam_update(float gg, float& m, float& v, float& p, float alpha, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    const float d1 = gg - m, o1 = 1.f - b1;
    const float mm = m + d1 * o1;
    const float g2 = gg * gg;
    const float d2 = g2 - v, o2 = 1.f - b2;
    const float vv = v + d2 * o2;
    m = mm; v = vv;
    const float num = mm * alpha, den = sqrtf(vv) + eps;
    p = p - num / den;
}
// step_bias: 1 = the counter holds t - 1 (the caller increments it after the update), 0 = it already holds t
__device__ __forceinline__ float adam_alpha_dev(const long long* step, double lr, double b1d, double b2d, int step_bias = 1) {
    const double t = (double)(*step + step_bias);
    return (float)(lr * sqrt(1.0 - pow(b2d, t)) / (1.0 - pow(b1d, t)));
}

// ---- Adam + weight packing in ONE pass (round 3) ---------------------------------------------------------------
// The separate chain read the gradients and wrote the parameters (adam_kernel), then read the parameters twice more
// to write the two bf16 operand copies (pack_all_kernel). Here a unit loads g, m, v, p of one kernel tile, updates
// them, and writes m, v, p AND both packed copies from the tile: 0.25 GB less traffic per step and one launch less.
//   CONV3 job  : unit = (tap, 64 ci, 64 co) tile; forward copy [tap][co][ci] transposed through LDS, data-gradient
//                copy [8 - tap][ci][co] from the same tile (16-byte stores).
//   UPCONV2 job: unit = (32 ci, 32 co) x the four taps (the data-gradient copy is the 3x3 stride-2 combination
//                W_eff[dy][dx] = sum of the taps S(dy) x S(dx), which needs all four updated taps of an element).
//   plain units: everything that is not a 3x3 / 2x2 kernel (biases, BatchNorm gamma / beta, the 1x1 head): Adam only,
//                1024 floats per unit, ranges in the table.
// dtype "bf16x3": the packed f32 operand words hold bf16 hi | bf16 lo << 16 (common.h: x3_word) -- written by the optimizer pass itself
template <int N> __device__ __forceinline__ void x3_words_of(float (&v)[N]) {
#pragma unroll
    for (int e = 0; e < N; ++e) v[e] = __uint_as_float(x3_word(v[e]));
}
struct AdamRange { long off, n; int unit_begin, _pad; };
constexpr int ADAM_MAX_RANGES = 48;
struct AdamPackTable { int njobs, nranges, plain_begin, _pad; PackJob job[PACK_MAX_JOBS]; AdamRange range[ADAM_MAX_RANGES]; };

template <typename T, bool X3 = false>
__device__ __forceinline__ void adam_pack_conv3_tile(const PackJob& j, int t, float* __restrict__ params,
                                                     const float* __restrict__ grads, float* __restrict__ am,
                                                     float* __restrict__ av, T* packed, float (*tile)[65], float alpha,
                                                     float b1, float b2, float eps) {
    constexpr int N = Vec<T>::N;
    const int Cin = j.Cin, Cout = j.Cout;
    const int tci = (Cin + 63) / 64, tco = (Cout + 63) / 64;
    const int tap = t / (tci * tco); const int r = t % (tci * tco);
    const int ci0 = (r / tco) * 64, co0 = (r % tco) * 64;
    const long base = j.w + (long)tap * Cin * Cout;
    {
        const int ty = threadIdx.x >> 4, tx4 = (threadIdx.x & 15) * 4;
        float4 g4[4], m4[4], v4[4], p4[4];
        long off[4]; bool in[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                            // all 16 loads of the thread in flight together
            const int ci = ci0 + ty + 16 * k, co = co0 + tx4;
            in[k] = ci < Cin && co < Cout;
            off[k] = base + (long)(ci < Cin ? ci : Cin - 1) * Cout + (co < Cout ? co : Cout - 4);
            g4[k] = *reinterpret_cast<const float4*>(grads + off[k]); m4[k] = *reinterpret_cast<const float4*>(am + off[k]);
            v4[k] = *reinterpret_cast<const float4*>(av + off[k]); p4[k] = *reinterpret_cast<const float4*>(params + off[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            adam_update(g4[k].x, m4[k].x, v4[k].x, p4[k].x, alpha, b1, b2, eps);
            adam_update(g4[k].y, m4[k].y, v4[k].y, p4[k].y, alpha, b1, b2, eps);
            adam_update(g4[k].z, m4[k].z, v4[k].z, p4[k].z, alpha, b1, b2, eps);
            adam_update(g4[k].w, m4[k].w, v4[k].w, p4[k].w, alpha, b1, b2, eps);
            if (in[k]) {                                         // (a clamped duplicate would be updated twice: in-range only)
                *reinterpret_cast<float4*>(am + off[k]) = m4[k]; *reinterpret_cast<float4*>(av + off[k]) = v4[k];
                *reinterpret_cast<float4*>(params + off[k]) = p4[k];
            }
            const int cil = ty + 16 * k;
            const float4 q = in[k] ? p4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            tile[cil][tx4] = q.x; tile[cil][tx4 + 1] = q.y; tile[cil][tx4 + 2] = q.z; tile[cil][tx4 + 3] = q.w;
        }
    }
    __syncthreads();
    T* dstf = packed + j.wf + (long)tap * Cin * Cout;
    constexpr int GPR = 64 / N, RPP = 256 / GPR;
#pragma unroll
    for (int pass = 0; pass < 64 / RPP; ++pass) {                // forward copy [co][ci]: columns of the tile
        const int col = threadIdx.x / GPR + pass * RPP, cil = (threadIdx.x % GPR) * N;
        const int co = co0 + col, ci = ci0 + cil;
        if (ci < Cin && co < Cout) {
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = tile[cil + e][col];
            if (X3) x3_words_of<N>(v);
            Vec<T>::store(dstf + (long)co * Cin + ci, v);
        }
    }
    T* dstd = packed + j.wd + (long)(8 - tap) * Cin * Cout;      // data-gradient copy: 180-degree rotated taps, rows of the tile
#pragma unroll
    for (int pass = 0; pass < 64 / RPP; ++pass) {
        const int row = threadIdx.x / GPR + pass * RPP, col = (threadIdx.x % GPR) * N;
        const int ci = ci0 + row, co = co0 + col;
        if (ci < Cin && co < Cout) {
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = tile[row][col + e];
            if (X3) x3_words_of<N>(v);
            Vec<T>::store(dstd + (long)ci * Cout + co, v);
        }
    }
}

template <typename T, bool X3 = false>
__device__ __forceinline__ void adam_pack_upconv_tile(const PackJob& j, int t, float* __restrict__ params,
                                                      const float* __restrict__ grads, float* __restrict__ am,
                                                      float* __restrict__ av, T* packed, float (*tile64)[65], float alpha,
                                                      float b1, float b2, float eps) {
    constexpr int N = Vec<T>::N;
    float (*tile)[32][33] = reinterpret_cast<float (*)[32][33]>(&tile64[0][0]);       // [4 taps][32 ci][33] (the kernel's LDS array is sized for it)
    const int Cin = j.Cin, Cout = j.Cout;
    const int tco = (Cout + 31) / 32;
    const int ci0 = (t / tco) * 32, co0 = (t % tco) * 32;
    const long per_tap = (long)Cin * Cout;
    {
        const int cil = threadIdx.x >> 3, tx4 = (threadIdx.x & 7) * 4;
        const int ci = ci0 + cil, co = co0 + tx4;
        const bool in = ci < Cin && co < Cout;
        const long o0 = j.w + (long)(ci < Cin ? ci : Cin - 1) * Cout + (co < Cout ? co : Cout - 4);
        float4 g4[4], m4[4], v4[4], p4[4];
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const long o = o0 + tp * per_tap;
            g4[tp] = *reinterpret_cast<const float4*>(grads + o); m4[tp] = *reinterpret_cast<const float4*>(am + o);
            v4[tp] = *reinterpret_cast<const float4*>(av + o); p4[tp] = *reinterpret_cast<const float4*>(params + o);
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            adam_update(g4[tp].x, m4[tp].x, v4[tp].x, p4[tp].x, alpha, b1, b2, eps);
            adam_update(g4[tp].y, m4[tp].y, v4[tp].y, p4[tp].y, alpha, b1, b2, eps);
            adam_update(g4[tp].z, m4[tp].z, v4[tp].z, p4[tp].z, alpha, b1, b2, eps);
            adam_update(g4[tp].w, m4[tp].w, v4[tp].w, p4[tp].w, alpha, b1, b2, eps);
            const long o = o0 + tp * per_tap;
            if (in) {
                *reinterpret_cast<float4*>(am + o) = m4[tp]; *reinterpret_cast<float4*>(av + o) = v4[tp];
                *reinterpret_cast<float4*>(params + o) = p4[tp];
            }
            const float4 q = in ? p4[tp] : make_float4(0.f, 0.f, 0.f, 0.f);
            tile[tp][cil][tx4] = q.x; tile[tp][cil][tx4 + 1] = q.y; tile[tp][cil][tx4 + 2] = q.z; tile[tp][cil][tx4 + 3] = q.w;
        }
    }
    __syncthreads();
    constexpr int GPR = 32 / N;                                  // 16-byte groups per 32-element row
    // forward copy [tap][co][ci]
    for (int idx = threadIdx.x; idx < 4 * 32 * GPR; idx += 256) {
        const int tp = idx / (32 * GPR), rem = idx % (32 * GPR);
        const int col = rem / GPR, cil = (rem % GPR) * N;
        const int co = co0 + col, ci = ci0 + cil;
        if (ci < Cin && co < Cout) {
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = tile[tp][cil + e][col];
            if (X3) x3_words_of<N>(v);
            Vec<T>::store(packed + j.wf + tp * per_tap + (long)co * Cin + ci, v);
        }
    }
    // data-gradient copy [tap'][ci][co], tap' = (dy+1)*3 + (dx+1): S(-1) = {1}, S(0) = {0, 1}, S(1) = {0} per axis
    // (same summation order as pack_dgrad_chunk: ky outer, kx inner)
    for (int idx = threadIdx.x; idx < 9 * 32 * GPR; idx += 256) {
        const int tp = idx / (32 * GPR), rem = idx % (32 * GPR);
        const int row = rem / GPR, col = (rem % GPR) * N;
        const int ci = ci0 + row, co = co0 + col;
        if (ci < Cin && co < Cout) {
            const int dy = tp / 3 - 1, dx = tp % 3 - 1;
            float v[N];
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = 0.f;
            for (int ky = 0; ky < 2; ++ky) {
                if ((dy == -1 && ky != 1) || (dy == 1 && ky != 0)) continue;
                for (int kx = 0; kx < 2; ++kx) {
                    if ((dx == -1 && kx != 1) || (dx == 1 && kx != 0)) continue;
#pragma unroll
                    for (int e = 0; e < N; ++e) v[e] += tile[ky * 2 + kx][row][col + e];
                }
            }
            if (X3) x3_words_of<N>(v);
            Vec<T>::store(packed + j.wd + tp * per_tap + (long)ci * Cout + co, v);
        }
    }
}

template <typename T, bool X3 = false>
__global__ __launch_bounds__(256) void adam_pack_all_kernel(AdamPackTable tab, float* __restrict__ params,
                                                            const float* __restrict__ grads, float* __restrict__ am,
                                                            float* __restrict__ av, T* packed, const long long* __restrict__ step,
                                                            double lr, double b1d, double b2d, float alpha_host, float eps, int step_bias) {
    __shared__ float tile_raw[4 * 32 * 33];                      // >= 64 x 65: both tile views live here
    float (*tile)[65] = reinterpret_cast<float (*)[65]>(tile_raw);
    const float alpha = step ? adam_alpha_dev(step, lr, b1d, b2d, step_bias) : alpha_host;
    const float b1 = (float)b1d, b2 = (float)b2d;
    const int u0 = (int)blockIdx.x;
    if (u0 >= tab.plain_begin) {                                 // biases, BatchNorm parameters, 1x1 head
        int ri = 0;
        while (ri + 1 < tab.nranges && u0 >= tab.range[ri + 1].unit_begin) ++ri;
        const AdamRange& r = tab.range[ri];
        const long e = (long)(u0 - r.unit_begin) * 1024 + threadIdx.x * 4;
        if (e >= r.n) return;
        const long o = r.off + e;
        if (e + 4 <= r.n && (o & 3) == 0) {
            float4 g4 = *reinterpret_cast<const float4*>(grads + o), m4 = *reinterpret_cast<float4*>(am + o),
                   v4 = *reinterpret_cast<float4*>(av + o), p4 = *reinterpret_cast<float4*>(params + o);
            adam_update(g4.x, m4.x, v4.x, p4.x, alpha, b1, b2, eps); adam_update(g4.y, m4.y, v4.y, p4.y, alpha, b1, b2, eps);
            adam_update(g4.z, m4.z, v4.z, p4.z, alpha, b1, b2, eps); adam_update(g4.w, m4.w, v4.w, p4.w, alpha, b1, b2, eps);
            *reinterpret_cast<float4*>(am + o) = m4; *reinterpret_cast<float4*>(av + o) = v4; *reinterpret_cast<float4*>(params + o) = p4;
        } else {
            for (int i = 0; i < 4 && e + i < r.n; ++i) {
                float mm = am[o + i], vv = av[o + i], pp = params[o + i];
                adam_update(grads[o + i], mm, vv, pp, alpha, b1, b2, eps);
                am[o + i] = mm; av[o + i] = vv; params[o + i] = pp;
            }
        }
        return;
    }
    int ji = 0;
    while (ji + 1 < tab.njobs && u0 >= tab.job[ji + 1].unit_begin) ++ji;
    const PackJob& j = tab.job[ji];
    const int u = u0 - j.unit_begin;
    if (j.mode == UPCONV2) adam_pack_upconv_tile<T, X3>(j, u, params, grads, am, av, packed, tile, alpha, b1, b2, eps);
    else adam_pack_conv3_tile<T, X3>(j, u, params, grads, am, av, packed, tile, alpha, b1, b2, eps);
}

__global__ void incr_step_kernel(long long* step);

// jobs: the 3x3 / 2x2 conv kernels (PackTable fields w, wf, wd, mode, Cin, Cout set); n_params = length of the flat buffers.
int launch_adam_pack_all(int dtype, PackTable& jobs, float* params, const float* grads, float* am, float* av, long n_params,
                         void* packed, long long* step, long long t_host, double lr, double b1, double b2, float eps,
                         hipStream_t st) {
    AdamPackTable tab; tab.njobs = jobs.njobs; tab.nranges = 0; tab._pad = 0;
    int units = 0;
    for (int i = 0; i < jobs.njobs; ++i) {
        PackJob& j = jobs.job[i];
        if (i > 0 && jobs.job[i - 1].w > j.w) return fail(MPU_EINVAL, "%s", "adam_pack: jobs must be ordered by parameter offset");
        j.unit_begin = units;
        j.fwd_units = j.mode == UPCONV2 ? cdiv(j.Cin, 32) * cdiv(j.Cout, 32) : 9 * cdiv(j.Cin, 64) * cdiv(j.Cout, 64);
        units += j.fwd_units;
        tab.job[i] = j;
    }
    tab.plain_begin = units;
    long cur = 0;                                                // the complement of the packed kernels inside [0, n_params)
    for (int i = 0; i <= jobs.njobs; ++i) {
        const long lo = i < jobs.njobs ? jobs.job[i].w : n_params;
        if (lo > cur) {
            if (tab.nranges >= ADAM_MAX_RANGES) return fail(MPU_EINVAL, "%s", "adam_pack: too many parameter ranges");
            AdamRange& r = tab.range[tab.nranges++];
            r.off = cur; r.n = lo - cur; r.unit_begin = units; r._pad = 0;
            units += (int)cdiv(r.n, 1024L);
        }
        if (i < jobs.njobs) {
            const PackJob& j = jobs.job[i];
            cur = j.w + (long)(j.mode == UPCONV2 ? 4 : 9) * j.Cin * j.Cout;
        }
    }
    if (units == 0) return MPU_OK;
    float alpha_host = 0.f;
    if (!step) alpha_host = (float)(lr * std::sqrt(1.0 - std::pow(b2, (double)t_host)) / (1.0 - std::pow(b1, (double)t_host)));
    if (dtype == MPU_BF16)
        adam_pack_all_kernel<bf16_t><<<units, 256, 0, st>>>(tab, params, grads, am, av, (bf16_t*)packed, step, lr, b1, b2, alpha_host, eps, 1);
    else if (dtype == MPU_F32X3)                                 // (f32 storage, operand words = bf16 hi | lo: no x3_words pass afterwards)
        adam_pack_all_kernel<float, true><<<units, 256, 0, st>>>(tab, params, grads, am, av, (float*)packed, step, lr, b1, b2, alpha_host, eps, 1);
    else
        adam_pack_all_kernel<float><<<units, 256, 0, st>>>(tab, params, grads, am, av, (float*)packed, step, lr, b1, b2, alpha_host, eps, 1);
    if (step) incr_step_kernel<<<1, 1, 0, st>>>(step);
    return launch_ok();
}

// ---- round 6: the optimizer beside the weight gradients -----------------------------------------------------------
// adam_pack_lean_kernel (bf16 operands): the same update and the same two packed copies as adam_pack_all_kernel, but small
// enough -- <= 64 registers, 8.5 KB of LDS -- to be CO-RESIDENT with a wgrad_taps workgroup (448 of a SIMD's 512 registers,
// 148 of a CU's 160 KB): the grouped weight-gradient launch is bound by MFMA issue and LDS reads, this kernel by HBM, so the
// optimizer of the parameters whose gradients are already final (the deep levels: 90 % of the bytes) runs on a second stream
// UNDER the weight gradients of the high-resolution levels instead of behind them (run_backward_adam, unet_model.hip).
//   CONV3 job  : unit = (tap, 64 ci, 64 co) tile as in adam_pack_all_kernel, loaded in two halves of 32 ci (8 instead of 16
//                16-byte loads per thread in flight), the tile held in LDS as bf16 -- the values both copies store.
//   UPCONV2 job: unit = (16 ci, 32 co) x the four taps, fp32 in LDS (the data-gradient copy sums taps in fp32 before rounding).
//   plain units: as adam_pack_all_kernel.
// Bit-identical to adam_pack_all_kernel (tests/test_gpu_unet.py).
constexpr int LEAN_TP = 68;                      // bf16 tile pitch (elements): rows 8-byte aligned
__global__ __launch_bounds__(256, 8) void adam_pack_lean_kernel(AdamPackTable tab, float* __restrict__ params,
                                                                const float* __restrict__ grads, float* __restrict__ am,
                                                                float* __restrict__ av, bf16_t* packed, const long long* __restrict__ step,
                                                                double lr, double b1d, double b2d, float alpha_host, float eps, int step_bias) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[64 * LEAN_TP * 2];          // 8704 B >= 4 x 16 x 33 floats (8448)
    const float alpha = step ? adam_alpha_dev(step, lr, b1d, b2d, step_bias) : alpha_host;
    const float b1 = (float)b1d, b2 = (float)b2d;
    const int u0 = (int)blockIdx.x;
    if (u0 >= tab.plain_begin) {                                 // biases, BatchNorm parameters, 1x1 head
        int ri = 0;
        while (ri + 1 < tab.nranges && u0 >= tab.range[ri + 1].unit_begin) ++ri;
        const AdamRange& r = tab.range[ri];
        const long e = (long)(u0 - r.unit_begin) * 1024 + threadIdx.x * 4;
        if (e >= r.n) return;
        const long o = r.off + e;
        if (e + 4 <= r.n && (o & 3) == 0) {
            float4 g4 = *reinterpret_cast<const float4*>(grads + o), m4 = *reinterpret_cast<float4*>(am + o),
                   v4 = *reinterpret_cast<float4*>(av + o), p4 = *reinterpret_cast<float4*>(params + o);
            adam_update(g4.x, m4.x, v4.x, p4.x, alpha, b1, b2, eps); adam_update(g4.y, m4.y, v4.y, p4.y, alpha, b1, b2, eps);
            adam_update(g4.z, m4.z, v4.z, p4.z, alpha, b1, b2, eps); adam_update(g4.w, m4.w, v4.w, p4.w, alpha, b1, b2, eps);
            *reinterpret_cast<float4*>(am + o) = m4; *reinterpret_cast<float4*>(av + o) = v4; *reinterpret_cast<float4*>(params + o) = p4;
        } else {
            for (int i = 0; i < 4 && e + i < r.n; ++i) {
                float mm = am[o + i], vv = av[o + i], pp = params[o + i];
                adam_update(grads[o + i], mm, vv, pp, alpha, b1, b2, eps);
                am[o + i] = mm; av[o + i] = vv; params[o + i] = pp;
            }
        }
        return;
    }
    int ji = 0;
    while (ji + 1 < tab.njobs && u0 >= tab.job[ji + 1].unit_begin) ++ji;
    const PackJob& j = tab.job[ji];
    const int t = u0 - j.unit_begin;
    const int Cin = j.Cin, Cout = j.Cout;
    if (j.mode != UPCONV2) {
        bf16_t (*tile)[LEAN_TP] = reinterpret_cast<bf16_t (*)[LEAN_TP]>(lds_raw);
        const int tci = (Cin + 63) / 64, tco = (Cout + 63) / 64;
        const int tap = t / (tci * tco); const int r = t % (tci * tco);
        const int ci0 = (r / tco) * 64, co0 = (r % tco) * 64;
        const long base = j.w + (long)tap * Cin * Cout;
        const int ty = threadIdx.x >> 4, tx4 = (threadIdx.x & 15) * 4;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            float4 g4[2], m4[2], v4[2], p4[2];
            long off[2]; bool in[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ci = ci0 + ty + 16 * (2 * half + k), co = co0 + tx4;
                in[k] = ci < Cin && co < Cout;
                off[k] = base + (long)(ci < Cin ? ci : Cin - 1) * Cout + (co < Cout ? co : Cout - 4);
                g4[k] = *reinterpret_cast<const float4*>(grads + off[k]); m4[k] = *reinterpret_cast<const float4*>(am + off[k]);
                v4[k] = *reinterpret_cast<const float4*>(av + off[k]); p4[k] = *reinterpret_cast<const float4*>(params + off[k]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                adam_update(g4[k].x, m4[k].x, v4[k].x, p4[k].x, alpha, b1, b2, eps);
                adam_update(g4[k].y, m4[k].y, v4[k].y, p4[k].y, alpha, b1, b2, eps);
                adam_update(g4[k].z, m4[k].z, v4[k].z, p4[k].z, alpha, b1, b2, eps);
                adam_update(g4[k].w, m4[k].w, v4[k].w, p4[k].w, alpha, b1, b2, eps);
                if (in[k]) {
                    *reinterpret_cast<float4*>(am + off[k]) = m4[k]; *reinterpret_cast<float4*>(av + off[k]) = v4[k];
                    *reinterpret_cast<float4*>(params + off[k]) = p4[k];
                }
                const int cil = ty + 16 * (2 * half + k);
                const float4 q = in[k] ? p4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(&tile[cil][tx4]) = make_uint2(f32x2_to_bf16x2(q.x, q.y), f32x2_to_bf16x2(q.z, q.w));
            }
        }
        __syncthreads();
        bf16_t* dstf = packed + j.wf + (long)tap * Cin * Cout;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {                   // forward copy [co][ci]: columns of the tile
            const int col = threadIdx.x / 8 + pass * 32, cil = (threadIdx.x % 8) * 8;
            const int co = co0 + col, ci = ci0 + cil;
            if (ci < Cin && co < Cout) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[cil + 2 * e][col] | ((uint32_t)tile[cil + 2 * e + 1][col] << 16);
                *reinterpret_cast<uint4*>(dstf + (long)co * Cin + ci) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        bf16_t* dstd = packed + j.wd + (long)(8 - tap) * Cin * Cout;      // data-gradient copy: rotated taps, rows of the tile
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = threadIdx.x / 8 + pass * 32, col = (threadIdx.x % 8) * 8;
            const int ci = ci0 + row, co = co0 + col;
            if (ci < Cin && co < Cout) {
                const uint2 a = *reinterpret_cast<const uint2*>(&tile[row][col]), b = *reinterpret_cast<const uint2*>(&tile[row][col + 4]);
                *reinterpret_cast<uint4*>(dstd + (long)ci * Cout + co) = make_uint4(a.x, a.y, b.x, b.y);
            }
        }
        return;
    }
    // UPCONV2: unit = (16 ci, 32 co) x four taps
    float (*tile)[16][33] = reinterpret_cast<float (*)[16][33]>(lds_raw);
    const int tco = (Cout + 31) / 32;
    const int ci0 = (t / tco) * 16, co0 = (t % tco) * 32;
    const long per_tap = (long)Cin * Cout;
    {
        const int th = threadIdx.x >> 7, cil = (threadIdx.x & 127) >> 3, tx4 = (threadIdx.x & 7) * 4;
        const int ci = ci0 + cil, co = co0 + tx4;
        const bool in = ci < Cin && co < Cout;
        const long o0 = j.w + (long)(ci < Cin ? ci : Cin - 1) * Cout + (co < Cout ? co : Cout - 4);
        float4 g4[2], m4[2], v4[2], p4[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long o = o0 + (2 * th + k) * per_tap;
            g4[k] = *reinterpret_cast<const float4*>(grads + o); m4[k] = *reinterpret_cast<const float4*>(am + o);
            v4[k] = *reinterpret_cast<const float4*>(av + o); p4[k] = *reinterpret_cast<const float4*>(params + o);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int tp = 2 * th + k;
            adam_update(g4[k].x, m4[k].x, v4[k].x, p4[k].x, alpha, b1, b2, eps);
            adam_update(g4[k].y, m4[k].y, v4[k].y, p4[k].y, alpha, b1, b2, eps);
            adam_update(g4[k].z, m4[k].z, v4[k].z, p4[k].z, alpha, b1, b2, eps);
            adam_update(g4[k].w, m4[k].w, v4[k].w, p4[k].w, alpha, b1, b2, eps);
            const long o = o0 + tp * per_tap;
            if (in) {
                *reinterpret_cast<float4*>(am + o) = m4[k]; *reinterpret_cast<float4*>(av + o) = v4[k];
                *reinterpret_cast<float4*>(params + o) = p4[k];
            }
            const float4 q = in ? p4[k] : make_float4(0.f, 0.f, 0.f, 0.f);
            tile[tp][cil][tx4] = q.x; tile[tp][cil][tx4 + 1] = q.y; tile[tp][cil][tx4 + 2] = q.z; tile[tp][cil][tx4 + 3] = q.w;
        }
    }
    __syncthreads();
    // forward copy [tap][co][ci]: 16 ci = two 16-byte groups per (tap, co)
    {
        const int tp = threadIdx.x >> 6, col = (threadIdx.x & 63) >> 1, cil = (threadIdx.x & 1) * 8;
        const int co = co0 + col, ci = ci0 + cil;
        if (ci < Cin && co < Cout) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[tp][cil + e][col];
            Vec<bf16_t>::store(packed + j.wf + tp * per_tap + (long)co * Cin + ci, v);
        }
    }
    // data-gradient copy [tap'][ci][co], tap' = (dy+1)*3 + (dx+1): S(-1) = {1}, S(0) = {0, 1}, S(1) = {0} per axis
    // (same summation order as adam_pack_upconv_tile / pack_dgrad_chunk: ky outer, kx inner)
    for (int idx = threadIdx.x; idx < 9 * 16 * 4; idx += 256) {
        const int tp = idx / 64, rem = idx % 64;
        const int row = rem / 4, col = (rem % 4) * 8;
        const int ci = ci0 + row, co = co0 + col;
        if (ci < Cin && co < Cout) {
            const int dy = tp / 3 - 1, dx = tp % 3 - 1;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            for (int ky = 0; ky < 2; ++ky) {
                if ((dy == -1 && ky != 1) || (dy == 1 && ky != 0)) continue;
                for (int kx = 0; kx < 2; ++kx) {
                    if ((dx == -1 && kx != 1) || (dx == 1 && kx != 0)) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += tile[ky * 2 + kx][row][col + e];
                }
            }
            Vec<bf16_t>::store(packed + j.wd + tp * per_tap + (long)ci * Cout + co, v);
        }
    }
}

// Adam + pack of the parameters in the nr (<= 2) ascending, disjoint ranges [p_lo[k], p_hi[k]) only (jobs: every packed kernel
// of the model, ordered by offset; a job is taken when its kernel lies inside a range, which must not cut one). lean: the
// co-resident kernel above (bf16 only). The device step counter (step != NULL) must already hold THIS step's number t
// (step_bias 0): mpu_unet_backward_adam advances it at the start of the backward pass, so that no launch of the tail has to
// wait for "every reader is done" before it moves.
int launch_adam_pack_ranges(int dtype, PackTable& jobs, float* params, const float* grads, float* am, float* av, const long* p_lo,
                            const long* p_hi, int nr, void* packed, long long* step, long long t_host, double lr, double b1,
                            double b2, float eps, bool lean, hipStream_t st) {
    AdamPackTable tab; tab.njobs = 0; tab.nranges = 0; tab._pad = 0;
    const bool use_lean = lean && dtype == MPU_BF16;
    int units = 0;
    long prev_w = -1;
    int job_range[PACK_MAX_JOBS];
    for (int i = 0; i < jobs.njobs; ++i) {
        PackJob j = jobs.job[i];
        if (j.w < prev_w) return fail(MPU_EINVAL, "%s", "adam_pack: jobs must be ordered by parameter offset");
        prev_w = j.w;
        const long end = j.w + (long)(j.mode == UPCONV2 ? 4 : 9) * j.Cin * j.Cout;
        int in = -1;
        for (int k = 0; k < nr; ++k) {
            if (end <= p_lo[k] || j.w >= p_hi[k]) continue;
            if (j.w < p_lo[k] || end > p_hi[k]) return fail(MPU_EINVAL, "%s", "adam_pack: a parameter range cuts a kernel");
            in = k;
        }
        if (in < 0) continue;
        j.unit_begin = units;
        j.fwd_units = j.mode == UPCONV2 ? cdiv(j.Cin, use_lean ? 16 : 32) * cdiv(j.Cout, 32) : 9 * cdiv(j.Cin, 64) * cdiv(j.Cout, 64);
        units += j.fwd_units;
        job_range[tab.njobs] = in;
        tab.job[tab.njobs++] = j;
    }
    tab.plain_begin = units;
    for (int k = 0; k < nr; ++k) {                               // the complement of the packed kernels inside each range
        if (k > 0 && p_lo[k] < p_hi[k - 1]) return fail(MPU_EINVAL, "%s", "adam_pack: ranges must ascend and not overlap");
        long cur = p_lo[k];
        for (int i = 0; i <= tab.njobs; ++i) {
            if (i < tab.njobs && job_range[i] != k) continue;
            const long lo = i < tab.njobs ? tab.job[i].w : p_hi[k];
            if (lo > cur) {
                if (tab.nranges >= ADAM_MAX_RANGES) return fail(MPU_EINVAL, "%s", "adam_pack: too many parameter ranges");
                AdamRange& r = tab.range[tab.nranges++];
                r.off = cur; r.n = lo - cur; r.unit_begin = units; r._pad = 0;
                units += (int)cdiv(r.n, 1024L);
            }
            if (i < tab.njobs) {
                const PackJob& j = tab.job[i];
                cur = j.w + (long)(j.mode == UPCONV2 ? 4 : 9) * j.Cin * j.Cout;
            }
        }
    }
    float alpha_host = 0.f;
    if (!step) alpha_host = (float)(lr * std::sqrt(1.0 - std::pow(b2, (double)t_host)) / (1.0 - std::pow(b1, (double)t_host)));
    if (units == 0) return MPU_OK;
    if (use_lean) {
        // ONE workgroup per compute unit, whatever arrives first: the launch claims 82 KB of LDS (8.5 KB used), so two of
        // them never share a CU, and 82 + 74 KB (wgrad_taps) do. Without the cap the 7 k short workgroups of this kernel
        // fill every CU eight deep and the weight-gradient workgroups (448 of 512 registers) wait for them to drain: the
        // two launches then run one after the other (measured, gpurun R6c: 135 + 311 us instead of side by side).
        constexpr int LEAN_CLAIM = 82 * 1024, LEAN_STATIC = 64 * LEAN_TP * 2;
        static unsigned long long attr_set = 0;
        if (first_use_on_device(attr_set)) {
            MPU_CHECK_HIP(hipFuncSetAttribute((const void*)adam_pack_lean_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LEAN_CLAIM - LEAN_STATIC));
            mark_used_on_device(attr_set);
        }
        adam_pack_lean_kernel<<<units, 256, LEAN_CLAIM - LEAN_STATIC, st>>>(tab, params, grads, am, av, (bf16_t*)packed, step, lr, b1, b2, alpha_host, eps, 0);
    } else if (dtype == MPU_BF16)
        adam_pack_all_kernel<bf16_t><<<units, 256, 0, st>>>(tab, params, grads, am, av, (bf16_t*)packed, step, lr, b1, b2, alpha_host, eps, 0);
    else
        adam_pack_all_kernel<float><<<units, 256, 0, st>>>(tab, params, grads, am, av, (float*)packed, step, lr, b1, b2, alpha_host, eps, 0);
    return launch_ok();
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float alpha, float b1, float b2, float eps) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float mm = m[e], vv = v[e], pp = p[e];
        adam_update(g[e], mm, vv, pp, alpha, b1, b2, eps);
        m[e] = mm; v[e] = vv; p[e] = pp;
    }
}
// graph-replayable variant: the 1-based step count lives in device memory (a captured launch cannot take a
// new host-computed step size on every replay)
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, const long long* __restrict__ step, double lr,
                                double b1d, double b2d, float eps) {
    const float alpha = adam_alpha_dev(step, lr, b1d, b2d);
    const float b1 = (float)b1d, b2 = (float)b2d;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        float mm = m[e], vv = v[e], pp = p[e];
        adam_update(g[e], mm, vv, pp, alpha, b1, b2, eps);
        m[e] = mm; v[e] = vv; p[e] = pp;
    }
}
__global__ void incr_step_kernel(long long* step) { *step += 1; }
int launch_adam_dev(float* p, const float* g, float* m, float* v, long n, long long* step, double lr, double b1,
                    double b2, float eps, hipStream_t st) {
    adam_dev_kernel<<<ew_grid(n), 256, 0, st>>>(p, g, m, v, n, step, lr, b1, b2, eps);
    incr_step_kernel<<<1, 1, 0, st>>>(step);
    return launch_ok();
}

// kernel_regularizer=l2(lambda) of the 3x3 / 2x2 conv kernels (reference unet.py:122-177,189): g += 2*lambda*W and,
// when wanted, lambda * sum W^2 (fixed summation order: L2_BLOCKS partial sums per kernel tensor, combined by one block)
constexpr int L2_BLOCKS = 64;
__global__ __launch_bounds__(256) void l2_grad_kernel(L2Table tab, const float* __restrict__ p, float* __restrict__ g,
                                                      float two_l2, double* __restrict__ partial) {
    const long off = tab.off[blockIdx.y], n = tab.n[blockIdx.y];
    double acc = 0.0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)L2_BLOCKS * 256) {
        const float w = p[off + e];
        g[off + e] = g[off + e] + two_l2 * w;
        acc += (double)w * (double)w;
    }
    if (!partial) return;
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long)blockIdx.y * L2_BLOCKS + blockIdx.x] = red[0];
}
__global__ void l2_loss_kernel(const double* __restrict__ partial, int n, float l2, float* __restrict__ out) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    *out = (float)(s * (double)l2);
}
int launch_l2_regularizer(const L2Table& tab, const float* params, float* grads, float l2, double* partial,
                          float* reg_loss, hipStream_t st) {
    if (tab.njobs == 0) return MPU_OK;
    l2_grad_kernel<<<dim3(L2_BLOCKS, tab.njobs), 256, 0, st>>>(tab, params, grads, 2.f * l2, reg_loss ? partial : nullptr);
    if (reg_loss) l2_loss_kernel<<<1, 1, 0, st>>>(partial, tab.njobs * L2_BLOCKS, l2, reg_loss);
    return launch_ok();
}

int launch_adam(float* p, const float* g, float* m, float* v, long n, float alpha, float b1, float b2, float eps,
                hipStream_t st) {
    adam_kernel<<<ew_grid(n), 256, 0, st>>>(p, g, m, v, n, alpha, b1, b2, eps);
    return launch_ok();
}

}  // namespace mpu
