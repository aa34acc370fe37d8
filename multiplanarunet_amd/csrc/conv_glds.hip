// conv_glds_kernel: the implicit-GEMM convolution of conv_igemm.hip with the operand tiles
// DMA'd straight from HBM/L2 into LDS (buffer_load_dwordx4 ... lds), no VGPR staging and no
// ds_write pass.
//
//  * one wave-instruction moves 64 lanes x 16 B = 8 K-rows of 128 B into 1 KiB of LDS
//    (destination = wave-uniform base + lane*16, so LDS rows are unpadded);
//  * bank conflicts are removed by an XOR swizzle applied on the SOURCE side: LDS slot p of row r
//    holds channel chunk p ^ ((r>>1)&7); a fragment read of chunk q goes to slot q ^ ((r>>1)&7).
//    For the non-contiguous 16-lane groups of ds_read_b128 on gfx950 this hits 16 distinct
//    16-byte slots of the 256-byte bank row (checked on paper and with SQ_LDS_BANK_CONFLICT);
//  * zero padding (image border, ragged tiles, channel tails, concat boundaries) costs nothing:
//    a buffer descriptor bounds the tensor and out-of-range lanes get an offset beyond
//    num_records, for which the DMA writes zeros (verified on hardware, tools/probe_glds3.hip).
//
// Pipeline: NSTAGE LDS stages; the loads of tile t+NSTAGE-1 are issued before the MFMAs of tile t.
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int MODE> struct GModeTraits;
template <> struct GModeTraits<CONV3>   { static constexpr int NTAPS = 9, KW = 3; };
template <> struct GModeTraits<UPCONV2> { static constexpr int NTAPS = 4, KW = 2; };
template <> struct GModeTraits<CONV3S2> { static constexpr int NTAPS = 9, KW = 3; };
template <> struct GModeTraits<CONV1>   { static constexpr int NTAPS = 1, KW = 1; };

template <int MODE>
__device__ __forceinline__ bool g_tap_src(int oy, int ox, int ky, int kx, int Ho, int Wo, int& iy, int& ix) {
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
template <int MODE> __device__ __forceinline__ int g_in_h(int Ho) {
    return MODE == UPCONV2 ? Ho / 2 : (MODE == CONV3S2 ? Ho * 2 : Ho);
}

template <typename T> struct GMma;
template <> struct GMma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
    }
};
template <> struct GMma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

typedef __attribute__((ext_vector_type(4))) int i32x4;

// raw buffer descriptor (stride 0, byte-granular range check) in four SGPRs
__device__ __forceinline__ i32x4 make_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa;
    r.y = (int)((unsigned)(pa >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = 0x00020000;
    return r;
}
// One LDS-DMA piece: 64 lanes x 16 B -> LDS[lds_addr + lane*16]. Issued as inline asm so that the
// compiler does not treat it as a pending LDS write (it would drain vmcnt(0) before every ds_read);
// completion is tracked by the caller's counted s_waitcnt vmcnt(N) + s_barrier.
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}

__device__ __forceinline__ int g_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T, int BN, int BM>
struct GldsCfg {
    static constexpr int STAGE = (BN + BM) * 128;
    static constexpr int OROW = BN * (int)sizeof(T) + 16;
    static constexpr int EPI = BM * OROW + 3 * BN * 4;
    // three stages (prefetch distance 2) only where three workgroups still fit one CU's 160 KiB:
    // measured, workgroups per CU matter more than prefetch depth (64x128: 2 stages/3 WGs 42.9 us
    // vs 3 stages/2 WGs 52.1 us on the 64-filter full-resolution layer)
    static constexpr int NSTAGE = (3 * STAGE <= 53 * 1024) ? 3 : 2;
    static constexpr int SMEM = (NSTAGE * STAGE > EPI) ? NSTAGE * STAGE : EPI;
};

template <typename T, int MODE, int BN, int BM, int WN, int WM>
__global__ __launch_bounds__(256, 2) void conv_glds_kernel(ConvArgs a) {
    using Cfg = GldsCfg<T, BN, BM>;
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BKE = 128 / sizeof(T);
    constexpr int TN = WN / 32, TM = WM / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NTAPS = GModeTraits<MODE>::NTAPS, KW = GModeTraits<MODE>::KW;
    constexpr int STAGE = Cfg::STAGE;
    constexpr int GW = BN / 32, GP = BM / 32;        // 8-row groups per wave (weights / pixels)
    static_assert((BN / WN) * (BM / WM) == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int logical = g_xcd_remap(blockIdx.x, gridDim.x);
    const int n0 = (logical % tiles_n) * BN;
    const long m0 = (long)(logical / tiles_n) * BM;
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    const int nit_all = NTAPS * nchunks;
    // split-K (deep layers with few output tiles): blockIdx.y owns K steps [it0, it0 + nit)
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    const int it0 = (int)((long)blockIdx.y * nit_all / ks);
    const int nit = (int)((long)(blockIdx.y + 1) * nit_all / ks) - it0;
    const int Hi = g_in_h<MODE>(a.Ho), Wi = g_in_h<MODE>(a.Wo);
    const long M = (long)a.B * a.Ho * a.Wo;
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)a.B * Hi * Wi;
    const i32x4 rs0 = make_rsrc(a.in0, npix * a.C0 * (long)sizeof(T));
    const i32x4 rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * (long)sizeof(T) : 0);
    const i32x4 rsw = make_rsrc(a.w, a.w_elems * (long)sizeof(T));
    const unsigned lds0 = (unsigned)(uintptr_t)smem;      // LDS byte address of the staging area

    // DMA roles: wave w moves weight rows [w*BN/4, (w+1)*BN/4) and pixel rows [w*BM/4, (w+1)*BM/4),
    // 8 rows per instruction; lane -> (row = lane>>3, LDS slot = lane&7), source chunk = slot ^ swz(row)
    const int lrow = lane >> 3, slot = lane & 7;
    unsigned wrow[GW]; int wchunk[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int rl = wave * (BN / 4) + g * 8 + lrow;           // tile-local weight row
        const int n = n0 + rl;
        wrow[g] = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * (long)sizeof(T)) : OOB;
        wchunk[g] = slot ^ ((rl >> 1) & 7);
    }
    int pb[GP], py[GP], px[GP], pchunk[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        const int rl = wave * (BM / 4) + g * 8 + lrow;           // tile-local pixel row
        const long m = m0 + rl;
        pchunk[g] = slot ^ ((rl >> 1) & 7);
        if (m < M) {
            const int ox = (int)(m % a.Wo); const long t = m / a.Wo;
            const int oy = (int)(t % a.Ho); const int b = (int)(t / a.Ho);
            pb[g] = b * Hi * Wi; py[g] = oy; px[g] = ox;
        } else { pb[g] = -1; py[g] = 0; px[g] = 0; }
    }

    // Requests: per-lane byte offsets computed once (weights) or once per tap (pixels: the tap's source pixel and
    // its validity), so a request is one multiply-add + the DMA instead of ~25 integer instructions.
    unsigned wlane[GW]; int wch[GW], pch[GP];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        wch[g] = wchunk[g] * EPC;
        wlane[g] = wrow[g] == OOB ? OOB : wrow[g] + (unsigned)(wch[g] * (int)sizeof(T));
    }
#pragma unroll
    for (int g = 0; g < GP; ++g) pch[g] = pchunk[g] * EPC;
    int ptap[GP], cur_tap = -1;                                  // source pixel index of the current tap (or -1)
    auto issue = [&](int tap, int cc, int stage) {
        const bool s1 = cc >= nch0;
        const int cbase = (s1 ? cc - nch0 : cc) * BKE;
        const int Cs = s1 ? a.C1 : a.C0;
        const int room = Cs - cbase;                             // only a tail chunk masks channels
        const unsigned soff = (unsigned)(((long)tap * a.w_tap_stride + (s1 ? a.C0 : 0) + cbase) * (long)sizeof(T));
        const unsigned sbase = lds0 + stage * STAGE;
#pragma unroll
        for (int g = 0; g < GW; ++g) {
            const unsigned off = (wch[g] < room && wlane[g] != OOB) ? wlane[g] + soff : OOB;
            dma16(rsw, off, sbase + (wave * (BN / 4) + g * 8) * 128);
        }
        if (tap != cur_tap) {
            cur_tap = tap;
            const int ky = tap / KW, kx = tap % KW;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                int iy, ix;
                const bool v = g_tap_src<MODE>(py[g], px[g], ky, kx, a.Ho, a.Wo, iy, ix) && pb[g] >= 0;
                ptap[g] = v ? pb[g] + iy * Wi + ix : -1;
            }
        }
#pragma unroll
        for (int g = 0; g < GP; ++g) {
            const unsigned off = (ptap[g] >= 0 && pch[g] < room)
                                     ? (unsigned)((ptap[g] * Cs + cbase + pch[g]) * (int)sizeof(T)) : OOB;
            const unsigned dst = sbase + BN * 128 + (wave * (BM / 4) + g * 8) * 128;
            if (s1) dma16(rs1, off, dst);
            else    dma16(rs0, off, dst);
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: row (lane&31) of a 32-row block, chunk q = 2s + (lane>>5) -> slot q ^ swz
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    auto compute = [&](int stage) {
        const unsigned char* Wb = smem + stage * STAGE + (wn * WN + (lane & 31)) * 128;
        const unsigned char* Pb = smem + stage * STAGE + BN * 128 + (wm * WM + (lane & 31)) * 128;
        constexpr int SG = (TN + TM > 4) ? 2 : 4;                // k-steps requested together (register budget)
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += SG) {
            uint4 af[SG][TN], bf[SG][TM];
#pragma unroll
            for (int s = 0; s < SG; ++s) {
                const int so = ((2 * (g0 + s) + fh) ^ fsw) << 4;
#pragma unroll
                for (int i = 0; i < TN; ++i) af[s][i] = *(const uint4*)(Wb + i * 32 * 128 + so);
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[s][j] = *(const uint4*)(Pb + j * 32 * 128 + so);
            }
            __builtin_amdgcn_sched_barrier(0);                   // keep the reads ahead of the MFMAs (see conv_halo.hip)
            if (sizeof(T) == 4 && a.x3) {                        // dtype "bf16x3": pairs of k-steps as split-bf16 products (common.h)
#pragma unroll
                for (int s = 0; s < SG; s += 2) {
                    s16x8 ah[TN], al[TN];
#pragma unroll
                    for (int i = 0; i < TN; ++i) x3_unpack(af[s][i], af[s + 1][i], ah[i], al[i]);   // weights: split when packed
#pragma unroll
                    for (int j = 0; j < TM; ++j) {               // (one pixel fragment split at a time: register budget)
                        s16x8 bh, bl;
                        x3_split(bf[s][j], bf[s + 1][j], bh, bl);
#pragma unroll
                        for (int i = 0; i < TN; ++i) x3_mma(ah[i], al[i], bh, bl, acc[i][j]);
                    }
                }
            } else {
#pragma unroll
            for (int s = 0; s < SG; ++s)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) GMma<T>::run(af[s][i], bf[s][j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int tapN = it0 / nchunks, ccN = it0 % nchunks;
    auto advance = [&]() { if (++ccN == nchunks) { ccN = 0; ++tapN; } };
    if constexpr (Cfg::NSTAGE == 2) {
        issue(tapN, ccN, 0); advance();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it < nit; ++it) {
            if (it + 1 < nit) { issue(tapN, ccN, (it + 1) & 1); advance(); }
            compute(it & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // Three stages, counted waits: tile t+2 is in flight across the barrier of iteration t.
        // A wave's DMA is ordered for the other waves by its own vmcnt wait followed by the barrier.
        constexpr int NLD = GW + GP;                   // DMA instructions per tile and wave
        issue(tapN, ccN, 0); advance();
        if (nit > 1) { issue(tapN, ccN, 1); advance(); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int st = 0;                                    // stage of tile `it`
        for (int it = 0; it < nit; ++it) {
            const int st2 = st >= 1 ? st - 1 : 2;      // (it + 2) % 3
            if (it + 2 < nit) { issue(tapN, ccN, st2); advance(); }
            compute(st);
            if (it + 2 < nit) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            st = st == 2 ? 0 : st + 1;
        }
    }

    if (ks > 1) {        // split-K: raw f32 partial sums; bias / ReLU / mask / convert happen in splitk_finish
        float* P = a.partial + (long)blockIdx.y * M * a.Cout;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const long m = m0 + wm * WM + j * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * WN + i * 32 + 8 * q + 4 * (lane >> 5);
                    if (n < a.Cout)
                        *(float4*)(P + m * a.Cout + n) = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1],
                                                                     acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
        }
        return;
    }
    // epilogue (same as conv_igemm_kernel): bias -> LDS, tile -> LDS, coalesced 16-byte row stores
    constexpr int OROW = Cfg::OROW;
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
        constexpr int CPRO = BN * (int)sizeof(T) / 16;
        T* out = (T*)a.out; const T* mask = (const T*)a.mask;
        constexpr int NIT = BM * CPRO / 256;
        static_assert(BM * CPRO % 256 == 0, "whole passes");
        // Round 6, accumulator mode only (a.stats_acc, set by the launcher when K is not split): the fused BatchNorm sums of the
        // stored values -- (sum y, sum y^2) of a forward launch, (sum dn, sum dn * xhat) of a data gradient whose output feeds a
        // BatchNorm backward (bn_x: that BatchNorm's input, read where a ReLU mask would be; sum dn * x stays raw per thread and
        // becomes invstd * (sum dn * x - mean * sum dn) once per workgroup). A thread keeps its channel chunk over all passes.
        const bool st_on = a.stats_acc != nullptr;
        const bool bnx = st_on && a.bn_x && !mask;
        const T* aux = mask ? mask : (const T*)a.bn_x;
        float ssum[EPC], ssq[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
        uint4 mkv[NIT];                         // ReLU masks of the data-gradient launches, requested up front (clamped
        if (mask || bnx) {                      // addresses: a load inside the pass is one exposed round trip per pass)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 256, row = idx / CPRO, c = idx % CPRO;
                const long m = m0 + row;
                const int n = n0 + c * EPC;
                mkv[it] = *(const uint4*)(aux + ((m < M && n < a.Cout) ? m * a.Cout + n : 0));
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            const int row = idx / CPRO, c = idx % CPRO;
            const long m = m0 + row;
            const int n = n0 + c * EPC;
            if (m >= M || n >= a.Cout) continue;
            uint4 val = *(const uint4*)(smem + row * OROW + c * 16);
            const long o = m * a.Cout + n;
            if (st_on) {
                const uint4 xk = bnx ? mkv[it] : val;
                const uint32_t vw[4] = {val.x, val.y, val.z, val.w}, xw[4] = {xk.x, xk.y, xk.z, xk.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (sizeof(T) == 2) {
                        const float lo = __uint_as_float(vw[e] << 16), hi = __uint_as_float(vw[e] & 0xffff0000u);
                        const float lo2 = __uint_as_float(xw[e] << 16), hi2 = __uint_as_float(xw[e] & 0xffff0000u);
                        ssum[(2 * e) % EPC] += lo; ssq[(2 * e) % EPC] += lo * lo2;
                        ssum[(2 * e + 1) % EPC] += hi; ssq[(2 * e + 1) % EPC] += hi * hi2;
                    } else {
                        const float v = __uint_as_float(vw[e]);
                        ssum[e % EPC] += v; ssq[e % EPC] += v * __uint_as_float(xw[e]);
                    }
                }
            }
            if (mask) {
                const uint4 mk = mkv[it];
                if (sizeof(T) == 2) {
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
        if (st_on) {                            // column sums over the 256 / CPRO row-lanes through the (now free) staging area
            constexpr int R = 256 / CPRO;
            __syncthreads();
            float* red = (float*)smem;          // [R][BN][2]
            const int rl = tid / CPRO, cc = tid % CPRO;
#pragma unroll
            for (int k = 0; k < EPC; ++k) {
                red[((rl * BN) + cc * EPC + k) * 2] = ssum[k];
                red[((rl * BN) + cc * EPC + k) * 2 + 1] = ssq[k];
            }
            __syncthreads();
            for (int col = tid; col < BN; col += 256) {
                float s0 = 0.f, s1 = 0.f;
                for (int r2 = 0; r2 < R; ++r2) { s0 += red[(r2 * BN + col) * 2]; s1 += red[(r2 * BN + col) * 2 + 1]; }
                const int ch = n0 + col;
                if (ch < a.Cout) {
                    if (bnx) s1 = a.bn_invstd[ch] * (s1 - a.bn_mean[ch] * s0);
                    stats_acc_add(a.stats_acc, a.Cout, 0, ch, s0, a.stats_scale[0]);
                    stats_acc_add(a.stats_acc, a.Cout, 1, ch, s1, a.stats_scale[1]);
                }
            }
        }
    }
}

// out = act(sum_z partial[z] + bias) (* ReLU mask), 16 bytes of output per thread (one 16-byte load of the mask, one
// 16-byte store). STATS: also the fused BatchNorm statistics of the stored values - the grid is sized so that a
// thread keeps its channel group over all its rows; per-block column sums go to stats[2][Cout][blocks].
// STATS 2: the BatchNorm-backward sums (sum out, sum out * xhat of bn_x) instead of (sum, sum of squares)
template <typename T, int STATS>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ partial, int ks, long M, int Cout,
                                                            const float* __restrict__ bias, const T* __restrict__ mask,
                                                            int relu, const float* __restrict__ ps,
                                                            const float* __restrict__ ph, T* __restrict__ out,
                                                            float* __restrict__ stats, const T* __restrict__ bn_x,
                                                            const float* __restrict__ bn_mean,
                                                            const float* __restrict__ bn_invstd, long long* stats_acc,
                                                            float acc_scale0, float acc_scale1) {
    constexpr int EPC = 16 / sizeof(T);
    const long total = M * Cout / EPC, stride = M * Cout;
    float ssum[EPC], ssq[EPC], bmu[EPC], bis[EPC];
#pragma unroll
    for (int k = 0; k < EPC; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; bmu[k] = 0.f; bis[k] = 0.f; }
    if (STATS == 2) {                                 // the thread's channel group is the same for all its rows
        const int n = (int)((((long)blockIdx.x * 256 + threadIdx.x) * EPC) % Cout);
#pragma unroll
        for (int k = 0; k < EPC; ++k) { bmu[k] = bn_mean[n + k]; bis[k] = bn_invstd[n + k]; }
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int n = (int)((e * EPC) % Cout);
        float v[EPC];
#pragma unroll
        for (int k = 0; k < EPC; k += 4) {
            const float4 b4 = bias ? *(const float4*)(bias + n + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[k] = b4.x; v[k + 1] = b4.y; v[k + 2] = b4.z; v[k + 3] = b4.w;
        }
        uint4 mk = make_uint4(0, 0, 0, 0);              // requested together with the first partials
        if (mask) mk = *(const uint4*)(mask + e * EPC);
        uint4 xq = make_uint4(0, 0, 0, 0);
        if (STATS == 2) xq = *(const uint4*)(bn_x + e * EPC);
        for (int z0 = 0; z0 < ks; z0 += 4) {            // four splits per round trip (clamped, unconditional loads;
            float4 p[4][EPC / 4];                       //  the adds keep the order z = 0, 1, 2, ...)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < EPC; k += 4)
                    p[u][k / 4] = *(const float4*)(partial + (z0 + u < ks ? z0 + u : ks - 1) * stride + e * EPC + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) {               // surplus splits add +0 through a select (a branch would pull
                const bool on = z0 + u < ks;            // their loads in and wait for each on the spot)
#pragma unroll
                for (int k = 0; k < EPC; k += 4) {
                    v[k] += on ? p[u][k / 4].x : 0.f; v[k + 1] += on ? p[u][k / 4].y : 0.f;
                    v[k + 2] += on ? p[u][k / 4].z : 0.f; v[k + 3] += on ? p[u][k / 4].w : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EPC; ++k) {
            if (relu) v[k] = fmaxf(v[k], 0.f);
            if (ps) v[k] = v[k] * ps[n + k] + ph[n + k];
        }
        uint4 o;
        if (sizeof(T) == 2) {
            o.x = f32x2_to_bf16x2(v[0], v[1]); o.y = f32x2_to_bf16x2(v[2], v[3]);
            o.z = f32x2_to_bf16x2(v[4 % EPC], v[5 % EPC]); o.w = f32x2_to_bf16x2(v[6 % EPC], v[7 % EPC]);
            if (mask) {
                auto keep = [](uint32_t mw, uint32_t vw) {
                    const uint32_t lo = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                    const uint32_t hi = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                    return vw & (lo | hi);
                };
                o.x = keep(mk.x, o.x); o.y = keep(mk.y, o.y); o.z = keep(mk.z, o.z); o.w = keep(mk.w, o.w);
            }
        } else {
            o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
            if (mask) {
                if (!(__uint_as_float(mk.x) > 0.f)) o.x = 0;
                if (!(__uint_as_float(mk.y) > 0.f)) o.y = 0;
                if (!(__uint_as_float(mk.z) > 0.f)) o.z = 0;
                if (!(__uint_as_float(mk.w) > 0.f)) o.w = 0;
            }
        }
        *(uint4*)(out + e * EPC) = o;
        if (STATS) {                                  // statistics of the STORED (rounded) values
            const uint32_t xw[4] = {xq.x, xq.y, xq.z, xq.w};
            if (sizeof(T) == 2) {
                const uint32_t wv[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = __uint_as_float(wv[q] << 16), hi = __uint_as_float(wv[q] & 0xffff0000u);
                    const int kl = (2 * q) % EPC, kh = (2 * q + 1) % EPC;
                    const float fl = STATS == 2 ? (__uint_as_float(xw[q] << 16) - bmu[kl]) * bis[kl] : lo;
                    const float fh = STATS == 2 ? (__uint_as_float(xw[q] & 0xffff0000u) - bmu[kh]) * bis[kh] : hi;
                    ssum[kl] += lo; ssq[kl] += lo * fl;
                    ssum[kh] += hi; ssq[kh] += hi * fh;
                }
            } else {
#pragma unroll
                for (int k = 0; k < EPC; ++k) {
                    const float f2 = STATS == 2 ? (__uint_as_float(xw[k % 4]) - bmu[k]) * bis[k] : v[k];
                    ssum[k] += v[k]; ssq[k] += v[k] * f2;
                }
            }
        }
    }
    if (STATS) {      // (gridDim.x * 256) % (Cout / EPC) == 0: the thread's channel group n is the same for all its rows
        __shared__ float red[256 * EPC * 2];
        const int cpr = Cout / EPC, cgi = threadIdx.x % cpr, nl = 256 / cpr;       // nl row-lanes per channel group
#pragma unroll
        for (int k = 0; k < EPC; ++k) { red[(threadIdx.x * EPC + k) * 2] = ssum[k]; red[(threadIdx.x * EPC + k) * 2 + 1] = ssq[k]; }
        __syncthreads();
        for (int vv = threadIdx.x; vv < Cout * 2; vv += 256) {
            const int col = vv >> 1, st2 = vv & 1, cg2 = col / EPC, k = col % EPC;
            double acc = 0.0;
            for (int rl = 0; rl < nl; ++rl) acc += (double)red[(((rl * cpr) + cg2) * EPC + k) * 2 + st2];
            if (stats_acc) stats_acc_add(stats_acc, Cout, st2, col, (float)acc, st2 ? acc_scale1 : acc_scale0);
            else stats[((long)st2 * Cout + col) * gridDim.x + blockIdx.x] = (float)acc;      // [2][Cout][blocks]
        }
        (void)cgi;
    }
}

// second pass of a split-K launch: sum the partials in fixed order, apply the epilogue (and the fused BN statistics)
template <typename T>
static int launch_splitk_finish(const ConvArgs& a, int ks, long M, hipStream_t st) {
    long work = M * a.Cout / (16 / (long)sizeof(T));
    long blocks = (work + 255) / 256; if (blocks > 4096) blocks = 4096;
    const int cpr = a.Cout / (16 / (int)sizeof(T));
    // fused BN statistics: <= 256 blocks, every thread keeps its channel group (256 % cpr == 0, cpr <= 256)
    const bool st_ok = a.stats && a.stats_rows && cpr <= 256 && 256 % cpr == 0 && a.Cout % (16 / (int)sizeof(T)) == 0;
    if (st_ok) {
        if (blocks > 256) blocks = 256;                          // (512 / 1024 blocks measured slower: +0.4 % / +1.2 % on the step, gpurun R5c)
        if (blocks * 2 * a.Cout > a.stats_cap) blocks = a.stats_cap / (2 * a.Cout);
        *a.stats_rows = (int)blocks;
        if (a.bn_x)
            launch_k(splitk_finish_kernel<T, 2>, (unsigned)blocks, 256, 0, st, a.partial, ks, M, a.Cout, a.bias, (const T*)a.mask,
                                                                         a.relu, a.post_scale, a.post_shift, (T*)a.out, a.stats,
                                                                         (const T*)a.bn_x, a.bn_mean, a.bn_invstd, a.stats_acc, a.stats_scale[0], a.stats_scale[1]);
        else
            launch_k(splitk_finish_kernel<T, 1>, (unsigned)blocks, 256, 0, st, a.partial, ks, M, a.Cout, a.bias, (const T*)a.mask,
                                                                         a.relu, a.post_scale, a.post_shift, (T*)a.out, a.stats,
                                                                         nullptr, nullptr, nullptr, a.stats_acc, a.stats_scale[0], a.stats_scale[1]);
    } else {
        launch_k(splitk_finish_kernel<T, 0>, (unsigned)blocks, 256, 0, st, a.partial, ks, M, a.Cout, a.bias, (const T*)a.mask,
                                                                     a.relu, a.post_scale, a.post_shift, (T*)a.out, nullptr,
                                                                     nullptr, nullptr, nullptr, nullptr, 0.f, 0.f);
    }
    return launch_ok();
}

template <typename T, int MODE, int BN, int BM, int WN, int WM>
static int launch_glds_cfg(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = GldsCfg<T, BN, BM>;
    auto kern = conv_glds_kernel<T, MODE, BN, BM, WN, WM>;
    ConvArgs a = a_in;
    constexpr int NT = GModeTraits<MODE>::NTAPS;
    if (a.w_elems <= 0) a.w_elems = (NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long tiles = (long)cdiv(a.Cout, BN) * cdiv(M, BM);
    {
        const long hi = MODE == UPCONV2 ? a.Ho / 2 : (MODE == CONV3S2 ? a.Ho * 2 : a.Ho);
        const long wi = MODE == UPCONV2 ? a.Wo / 2 : (MODE == CONV3S2 ? a.Wo * 2 : a.Wo);
        const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
        if ((long)a.B * hi * wi * cmax * (long)sizeof(T) >= (1L << 31) || a.w_elems * (long)sizeof(T) >= (1L << 31))
            return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    }
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * NT * (a.C0 + a.C1), st);
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    a.ksplit = ks;
    ConvArgs ak = a;                             // (the kernel's copy: its epilogue sums the BatchNorm terms only when K is not split
    if (ks == 1) {                               //  and only into an accumulator; the split-K finish pass has its own)
        const bool ok = a.stats_acc && a.stats && a.stats_rows && !(a.bn_x && (a.mask || !a.bn_mean || !a.bn_invstd)) &&
                        Cfg::SMEM >= (256 / (BN * (int)sizeof(T) / 16)) * BN * 2 * 4;
        if (ok) *a.stats_rows = 1; else ak.stats_acc = nullptr;
    } else ak.stats_acc = nullptr;
    launch_k(kern, dim3((unsigned)tiles, ks), dim3(256), Cfg::SMEM, st, ak);
    int rc = launch_ok();
    if (!rc && ks > 1) rc = launch_splitk_finish<T>(a, ks, M, st);
    if (prof_on()) prof_end(st);
    return rc;
}

// ------------------------------------------------------------------------- //
// conv_pipe_kernel (bf16): the same implicit GEMM for the DEEP U-Net levels (few pixels, long reductions, the
// weights dominate the traffic), scheduled for ONE workgroup per CU:
//   * 256-pixel x 128-channel tiles, 8 waves = 2 per SIMD (each a 64 x 64 sub-tile); L2->LDS fill per K step =
//     48 KB per 1024 MFMA cycles (47 B/clk against the ~62 B/clk a CU can pull);
//   * three 48-KB LDS stages, the DMA of K step t+2 issued during step t (prefetch distance two steps);
//   * a hand-placed instruction stream. In-kernel s_memtime stamps of the first version showed a wave spending
//     ~2100 cycles on a step whose 16 MFMAs need 512: a wave issues in order, so everything between two of its
//     MFMAs delays the second one, and ~200 address / select / branch instructions per step sat there. Now every
//     per-lane DMA offset is precomputed (masked lanes carry a poison offset beyond the buffer's num_records:
//     one v_add per piece), every fragment address lives in a register (none computed in the loop), the K loop is
//     unrolled over the three stages (LDS destinations are immediates), and the stream alternates ONE MFMA with
//     ONE ds_read and at most one DMA piece, so that the non-matrix work hides in the shadow of the wave's own MFMA
//     and of its SIMD partner's;
//   * MFMA operand fragments triple-buffered in registers: the reads of k-step s+1 (and s+2 at the stage end) are
//     issued under the MFMAs of k-step s, also across the stage boundary; one barrier per 16 MFMAs and wave, with no
//     LDS read outstanding at it;
//   * 1-D grid decoded m-tile fastest, then n-tile, then K slice, XCD-aware: the workgroups that share a slice of
//     the weights run on one XCD, so the 19 MB of bottom-level weights are pulled from HBM once;
//   * split-K partials leave through wave-private LDS staging as 256-byte row runs (the direct accumulator stores
//     were 32-byte pieces and took ~3 us of every launch).
// ksplit == 1 runs the same staged epilogue as conv_glds_kernel. RAGGED = channel counts that are not multiples of
// 64 (tail chunks mask lanes per step); DBG = s_memtime stamps (MPU_PIPE_DEBUG=32, dev aid).
// ------------------------------------------------------------------------- //
struct PipeCfg {
    static constexpr int BN = 128, BM = 256, NS = 3;
    static constexpr int STAGE = (BN + BM) * 128;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int EPI = BM * OROW + 3 * BN * 4;
    static constexpr int SROW = 64 * 4 + 16;                     // split-K staging: 64 f32 + pad per pixel row, per wave
    static constexpr int STG = 8 * 64 * SROW;
    static constexpr int S0 = NS * STAGE > EPI ? NS * STAGE : EPI;
    static constexpr int SMEM = S0 > STG ? S0 : STG;
};
constexpr unsigned PIPE_POISON = 0x80001000u;                    // + any in-range byte offset stays >= num_records (< 2^31 - 8192)

// LDS-DMA piece with the LDS destination = scalar base + immediate (one SALU op)
template <int IMM>
__device__ __forceinline__ void dma16_at(const i32x4& rsrc, unsigned voff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_base), "s"(rsrc), "n"(IMM) : "memory", "scc");
}

template <int MODE, bool RAGGED, int DBG>
__global__ __launch_bounds__(512, 2) void conv_pipe_kernel(ConvArgs a, int tiles_m, int tiles_n, unsigned magic_w, unsigned magic_h) {
    typedef bf16_t T;
    constexpr int BN = PipeCfg::BN, BM = PipeCfg::BM, STAGE = PipeCfg::STAGE;
    constexpr int EPC = 8, BKE = 64;
    constexpr int NTAPS = GModeTraits<MODE>::NTAPS, KW = GModeTraits<MODE>::KW;
    constexpr int GW = 2, GP = 4, NLD = GW + GP;                 // 8-row DMA pieces per wave: weights, pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;                     // 2 x 4 waves of 64 channels x 64 pixels
    const int logical = g_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = logical % tiles_m, r1 = logical / tiles_m;
    const int nt = r1 % tiles_n, kz = r1 / tiles_n;
    const int n0 = nt * BN, m0 = mt * BM;
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    const int nit_all = NTAPS * nchunks;
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    const int it0 = (int)((long)kz * nit_all / ks);
    const int nit = (int)((long)(kz + 1) * nit_all / ks) - it0;
    const int Hi = g_in_h<MODE>(a.Ho), Wi = g_in_h<MODE>(a.Wo);
    const int M = a.B * a.Ho * a.Wo;                             // (all operands < 2 GiB: checked by the launcher)
    const long npix = (long)a.B * Hi * Wi;
    const i32x4 rs0 = make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rs1 = make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * 2L : 0);
    const i32x4 rsw = make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const int wts = (int)a.w_tap_stride;

    // s_memtime stamps (dev aid): workgroups 0 and 131, waves 0 and 5
    const bool stamping = DBG && (a.dbg & 32) && a.dbg_buf && (blockIdx.x == 0 || blockIdx.x == 131) && (wave == 0 || wave == 5) && lane == 0;
    __shared__ unsigned long long s_stamps[DBG ? 2 * 64 : 1];
    unsigned long long* sl = s_stamps + (wave == 0 ? 0 : (DBG ? 64 : 0));
    int nstamp = 0;
    auto stamp = [&]() { if (DBG && stamping && nstamp < 63) sl[nstamp++] = __builtin_amdgcn_s_memtime(); };
    auto flush_stamps = [&]() {
        if (DBG && stamping) {
            unsigned long long* sbuf = a.dbg_buf + ((blockIdx.x == 0 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 64;
            for (int k = 0; k < nstamp; ++k) sbuf[k] = sl[k];
            sbuf[63] = (unsigned long long)nstamp;
        }
    };
    stamp();

    // ---- per-lane DMA roles: byte offsets computed once; invalid lanes carry the poison offset -----------------
    const int lrow = lane >> 3, slot = lane & 7;
    unsigned wl[GW]; int wch[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int rl = wave * (BN / 8) + g * 8 + lrow;           // tile-local weight row
        const int n = n0 + rl;
        wch[g] = (slot ^ ((rl >> 1) & 7)) * EPC;
        wl[g] = n < a.Cout ? (unsigned)(n * a.w_row_stride * 2 + wch[g] * 2) : PIPE_POISON;
    }
    int pb[GP], py[GP], px[GP], pch[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        const int rl = wave * (BM / 8) + g * 8 + lrow;           // tile-local pixel row
        const int m = m0 + rl;
        pch[g] = (slot ^ ((rl >> 1) & 7)) * EPC;
        if (m < M) {                                             // exact multiply-high division (host-checked range)
            // (a magic of 0 = divisor 1: ceil(2^32 / 1) does not fit 32 bits -- 1 x 1 and N x 1 maps)
            const int t = magic_w ? (int)__umulhi((unsigned)m, magic_w) : m, ox = m - t * a.Wo;
            const int b = magic_h ? (int)__umulhi((unsigned)t, magic_h) : t, oy = t - b * a.Ho;
            pb[g] = b * Hi * Wi; py[g] = oy; px[g] = ox;
        } else { pb[g] = -1; py[g] = 0; px[g] = 0; }
    }
    const unsigned wdst = lds0 + wave * (BN / 8) * 128;          // + g * 1024 + stage * STAGE (immediates)
    const unsigned pdst = lds0 + BN * 128 + wave * (BM / 8) * 128;

    // ---- request state: (r_tap, r_cc) = K step to request next; scalars of that step -----------------------------
    int r_tap = it0 / nchunks, r_cc = it0 % nchunks;
    int cur_tap = -1, cur_src = -1;
    unsigned pbase[GP];                                          // per lane: (source pixel * Cs + chunk channel) * 2, or poison
    unsigned q_wsoff = 0, q_pcoff = 0; int q_room = 0; i32x4 q_rs = rs0;
    // valid = false (past the last K step of this workgroup): the step's scalar offsets become the poison, so the
    // six DMA pieces of the stream stay unconditional (they fetch nothing useful into a stage nobody reads any
    // more) and every step leaves exactly NLD requests in flight for the counted vmcnt wait
    auto begin_step = [&](bool valid) {
        const int tap = r_tap, cc = r_cc;
        if (++r_cc == nchunks) { r_cc = 0; ++r_tap; }
        const int s1 = cc >= nch0 ? 1 : 0;
        const int cb = (s1 ? cc - nch0 : cc) * BKE;
        const int Cs = s1 ? a.C1 : a.C0;
        q_wsoff = valid ? (unsigned)((tap * wts + (s1 ? a.C0 : 0) + cb) * 2) : PIPE_POISON;
        q_pcoff = valid ? (unsigned)(cb * 2) : PIPE_POISON;
        q_room = Cs - cb;
        q_rs.x = s1 ? rs1.x : rs0.x; q_rs.y = s1 ? rs1.y : rs0.y; q_rs.z = s1 ? rs1.z : rs0.z; q_rs.w = rs0.w;
        if (valid && (tap != cur_tap || s1 != cur_src)) {        // rare: a new tap or the second concat source
            cur_tap = tap; cur_src = s1;
            const int ky = tap / KW, kx = tap % KW;
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                int iy, ix;
                const bool v = g_tap_src<MODE>(py[g], px[g], ky, kx, a.Ho, a.Wo, iy, ix) && pb[g] >= 0;
                pbase[g] = v ? (unsigned)(((pb[g] + iy * Wi + ix) * Cs + pch[g]) * 2) : PIPE_POISON;
            }
        }
    };
    auto w_off = [&](int g) -> unsigned {
        const unsigned o = wl[g] + q_wsoff;
        return (RAGGED && wch[g] >= q_room) ? PIPE_POISON : o;
    };
    auto p_off = [&](int g) -> unsigned {
        const unsigned o = pbase[g] + q_pcoff;
        return (RAGGED && pch[g] >= q_room) ? PIPE_POISON : o;
    };
#define PIPE_DMA_W(G, ST_) dma16_at<(ST_) * STAGE + (G) * 1024>(rsw, w_off(G), wdst)
#define PIPE_DMA_P(G, ST_) dma16_at<(ST_) * STAGE + (G) * 1024>(q_rs, p_off(G), pdst)

    // the first two K steps are requested before anything else is set up (their round trip covers the rest)
    stamp();
    begin_step(true);
    PIPE_DMA_W(0, 0); PIPE_DMA_W(1, 0); PIPE_DMA_P(0, 0); PIPE_DMA_P(1, 0); PIPE_DMA_P(2, 0); PIPE_DMA_P(3, 0);
    begin_step(nit > 1);
    PIPE_DMA_W(0, 1); PIPE_DMA_W(1, 1); PIPE_DMA_P(0, 1); PIPE_DMA_P(1, 1); PIPE_DMA_P(2, 1); PIPE_DMA_P(3, 1);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addresses: one register per (operand, k-step) for stages 0/1 (stage 1 = +STAGE as the
    // instruction's immediate) and one for stage 2 (2 * STAGE does not fit the 16-bit immediate) -------------------
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    unsigned LA[4], LB[4], LA2[4], LB2[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const unsigned so = (unsigned)(((2 * s_ + fh) ^ fsw) << 4);
        LA[s_] = lds0 + (wn * 64 + (lane & 31)) * 128 + so;
        LB[s_] = lds0 + BN * 128 + (wm * 64 + (lane & 31)) * 128 + so;
        LA2[s_] = LA[s_] + 2 * STAGE; LB2[s_] = LB[s_] + 2 * STAGE;
    }
    typedef unsigned int pipe_u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const pipe_u32x4* lds_u4;
#define PIPE_LD(DST, ARR, ARR2, ST_, S_, HALF)                                                         \
    DST = *(lds_u4)(uintptr_t)(((ST_) == 2 ? ARR2[S_] : ARR[S_]) + ((ST_) == 1 ? STAGE : 0) + (HALF) * 4096)
#define PIPE_SB() __builtin_amdgcn_sched_barrier(0)
#define PIPE_MM(FA, FB, I, J)                                                                          \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, FA[I]), __builtin_bit_cast(s16x8, FB[J]), acc[I][J], 0, 0, 0)

    pipe_u32x4 fa0[2], fb0[2], fa1[2], fb1[2], fa2[2], fb2[2];   // three fragment sets

    // one K step on stage ST: 16 MFMAs, each followed by one fragment read and (12 of them) a share of the next DMA
    auto kstep = [&](auto stc, int it) {
        constexpr int ST = decltype(stc)::value, STN = (ST + 1) % 3, ST2 = (ST + 2) % 3;
        begin_step(it + 2 < nit);                                // request K step it+2 into stage ST2 (free since step it-1)
        PIPE_SB();
        // k-step 0 on set 0; set 1 <- (ST, 1)
        PIPE_MM(fa0, fb0, 0, 0); PIPE_LD(fa1[0], LA, LA2, ST, 1, 0); PIPE_DMA_W(0, ST2); PIPE_SB();
        PIPE_MM(fa0, fb0, 0, 1); PIPE_LD(fb1[0], LB, LB2, ST, 1, 0); PIPE_DMA_W(1, ST2); PIPE_SB();
        PIPE_MM(fa0, fb0, 1, 0); PIPE_LD(fa1[1], LA, LA2, ST, 1, 1); PIPE_SB();
        PIPE_MM(fa0, fb0, 1, 1); PIPE_LD(fb1[1], LB, LB2, ST, 1, 1); PIPE_SB();
        // k-step 1 on set 1; set 0 <- (ST, 2), set 2 <- (ST, 3)
        PIPE_MM(fa1, fb1, 0, 0); PIPE_LD(fa0[0], LA, LA2, ST, 2, 0); PIPE_LD(fa2[0], LA, LA2, ST, 3, 0); PIPE_DMA_P(0, ST2); PIPE_SB();
        PIPE_MM(fa1, fb1, 0, 1); PIPE_LD(fb0[0], LB, LB2, ST, 2, 0); PIPE_LD(fb2[0], LB, LB2, ST, 3, 0); PIPE_DMA_P(1, ST2); PIPE_SB();
        PIPE_MM(fa1, fb1, 1, 0); PIPE_LD(fa0[1], LA, LA2, ST, 2, 1); PIPE_LD(fa2[1], LA, LA2, ST, 3, 1); PIPE_SB();
        PIPE_MM(fa1, fb1, 1, 1); PIPE_LD(fb0[1], LB, LB2, ST, 2, 1); PIPE_LD(fb2[1], LB, LB2, ST, 3, 1); PIPE_SB();
        // k-step 2 on set 0: no reads (every read of stage ST has been issued 4+ MFMAs before the barrier)
        PIPE_MM(fa0, fb0, 0, 0); PIPE_DMA_P(2, ST2); PIPE_SB();
        PIPE_MM(fa0, fb0, 0, 1); PIPE_DMA_P(3, ST2); PIPE_SB();
        PIPE_MM(fa0, fb0, 1, 0); PIPE_SB();
        PIPE_MM(fa0, fb0, 1, 1); PIPE_SB();
        if (it + 1 < nit) {
            // every read of stage ST has returned; stage it+1 has landed (own pieces: vmcnt, all waves': barrier)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (DBG && it < 8) stamp();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            if (DBG && it < 8) stamp();
            __builtin_amdgcn_s_barrier();
            if (DBG && it < 8) stamp();
        }
        PIPE_SB();
        // k-step 3 on set 2; set 0 <- (STN, 0) (after the last step: a harmless read of a stale stage)
        PIPE_MM(fa2, fb2, 0, 0); PIPE_LD(fa0[0], LA, LA2, STN, 0, 0); PIPE_SB();
        PIPE_MM(fa2, fb2, 0, 1); PIPE_LD(fb0[0], LB, LB2, STN, 0, 0); PIPE_SB();
        PIPE_MM(fa2, fb2, 1, 0); PIPE_LD(fa0[1], LA, LA2, STN, 0, 1); PIPE_SB();
        PIPE_MM(fa2, fb2, 1, 1); PIPE_LD(fb0[1], LB, LB2, STN, 0, 1); PIPE_SB();
    };

    // waves 4-7 (the younger half: the arbitration loser on every segment) get static priority
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
    stamp();
    __builtin_amdgcn_s_barrier();
    stamp();
    PIPE_LD(fa0[0], LA, LA2, 0, 0, 0); PIPE_LD(fb0[0], LB, LB2, 0, 0, 0);
    PIPE_LD(fa0[1], LA, LA2, 0, 0, 1); PIPE_LD(fb0[1], LB, LB2, 0, 0, 1);
    for (int it = 0; it < nit; it += 3) {
        kstep(std::integral_constant<int, 0>(), it);
        if (it + 1 < nit) kstep(std::integral_constant<int, 1>(), it + 1);
        if (it + 2 < nit) kstep(std::integral_constant<int, 2>(), it + 2);
    }
#undef PIPE_LD
#undef PIPE_SB
#undef PIPE_MM
#undef PIPE_DMA_W
#undef PIPE_DMA_P
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the trailing (poisoned) requests still write zeros into LDS
    stamp();

    if (ks > 1) {
        // split-K: raw f32 partial sums [kz][M][Cout], staged through wave-private LDS rows so that each store
        // instruction writes four 256-byte row runs
        constexpr int SROW = PipeCfg::SROW;
        float* P = a.partial + (long)kz * M * a.Cout;
        __syncthreads();                                         // lagging waves still read the last stage
        unsigned char* sw = smem + wave * (64 * SROW);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(float4*)(sw + (j * 32 + (lane & 31)) * SROW + (i * 32 + 8 * q + 4 * fh) * 4) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int nl = n0 + wn * 64 + (lane & 15) * 4;
        float4 v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) v[it] = *(const float4*)(sw + (it * 4 + (lane >> 4)) * SROW + (lane & 15) * 16);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + wm * 64 + it * 4 + (lane >> 4);
            if (m < M && nl < a.Cout) *(float4*)(P + (long)m * a.Cout + nl) = v[it];
        }
        if (DBG && stamping) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); flush_stamps(); }
        return;
    }
    // ksplit == 1: bias / ReLU / folded-BN affine, tile staged through LDS, coalesced 16-byte row stores
    constexpr int OROW = PipeCfg::OROW;
    __syncthreads();                                             // lagging waves still read the last stage
    float* sbias = (float*)(smem + BM * OROW);
    if (tid < BN) {
        const bool nv = n0 + tid < a.Cout;
        sbias[tid] = (a.bias && nv) ? a.bias[n0 + tid] : 0.f;
        sbias[BN + tid] = (a.post_scale && nv) ? a.post_scale[n0 + tid] : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && nv) ? a.post_shift[n0 + tid] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ml = wm * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * 64 + i * 32 + 8 * q + 4 * (lane >> 5);
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
                uint2 pk;
                pk.x = f32x2_to_bf16x2(v[0], v[1]);
                pk.y = f32x2_to_bf16x2(v[2], v[3]);
                *(uint2*)(smem + ml * OROW + nl * 2) = pk;
            }
        }
    }
    __syncthreads();
    {
        constexpr int CPRO = BN * 2 / 16;                        // 16-byte pieces per output row
        T* out = (T*)a.out; const T* mask = (const T*)a.mask;
        constexpr int NIT = BM * CPRO / 512;
        uint4 mkv[NIT];
        if (mask) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + it * 512, row = idx / CPRO, c = idx % CPRO;
                const int m = m0 + row;
                const int n = n0 + c * EPC;
                mkv[it] = *(const uint4*)(mask + ((m < M && n < a.Cout) ? (long)m * a.Cout + n : 0));
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512;
            const int row = idx / CPRO, c = idx % CPRO;
            const int m = m0 + row;
            const int n = n0 + c * EPC;
            if (m >= M || n >= a.Cout) continue;
            uint4 val = *(const uint4*)(smem + row * OROW + c * 16);
            if (mask) {
                const uint4 mk = mkv[it];
                auto keep = [](uint32_t mw, uint32_t vw) {
                    const uint32_t lo = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                    const uint32_t hi = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                    return vw & (lo | hi);
                };
                val.x = keep(mk.x, val.x); val.y = keep(mk.y, val.y);
                val.z = keep(mk.z, val.z); val.w = keep(mk.w, val.w);
            }
            *(uint4*)(out + (long)m * a.Cout + n) = val;
        }
    }
}

template <int MODE, bool RAGGED, int DBG>
static int launch_pipe(const ConvArgs& a_in, int ks, hipStream_t st) {
    auto kern = conv_pipe_kernel<MODE, RAGGED, DBG>;
    ConvArgs a = a_in;
    constexpr int NT = GModeTraits<MODE>::NTAPS;
    if (a.w_elems <= 0) a.w_elems = (NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PipeCfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const int tiles_m = cdiv(M, PipeCfg::BM), tiles_n = cdiv(a.Cout, PipeCfg::BN);
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * NT * (a.C0 + a.C1), st);
    a.ksplit = ks > 1 ? ks : 1;
    const unsigned mw = (unsigned)(((1UL << 32) + a.Wo - 1) / a.Wo), mh = (unsigned)(((1UL << 32) + a.Ho - 1) / a.Ho);
    launch_k(kern, dim3((unsigned)((long)tiles_m * tiles_n * a.ksplit)), dim3(512), PipeCfg::SMEM, st, a, tiles_m, tiles_n, mw, mh);
    int rc = launch_ok();
    if (!rc && a.ksplit > 1) rc = launch_splitk_finish<bf16_t>(a, a.ksplit, M, st);
    if (prof_on()) prof_end(st);
    return rc;
}

// 1 = launched by the one-workgroup-per-CU schedule, 0 = shape not suited, < 0 = error
template <typename T, int MODE>
static int try_pipe(const ConvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) != 2 || MODE == CONV1) return 0;
    else {
        const int on = (int)env(ENV_CONV_PIPE), dbg = (int)env(ENV_PIPE_DEBUG);
        constexpr int min_steps = 12, wgs = 256;                 // >= 12 K steps per workgroup; about one workgroup per CU
        if (!on || a.Cout < 128) return 0;
        const long M = (long)a.B * a.Ho * a.Wo;
        const long tiles = (long)cdiv(M, PipeCfg::BM) * cdiv(a.Cout, PipeCfg::BN);
        const int nit = GModeTraits<MODE>::NTAPS * (cdiv(a.C0, 64) + cdiv(a.C1, 64));
        if (tiles > 2L * wgs || nit < min_steps) return 0;      // large grids: the two-workgroup schedules fill the chip
        {   // 32-bit offsets with a poison margin: every operand (and the f32 output rows) below 2 GiB - 8 KiB
            const long hi = MODE == UPCONV2 ? a.Ho / 2 : (MODE == CONV3S2 ? a.Ho * 2 : a.Ho);
            const long wi = MODE == UPCONV2 ? a.Wo / 2 : (MODE == CONV3S2 ? a.Wo * 2 : a.Wo);
            const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
            const long wel = a.w_elems > 0 ? a.w_elems : (GModeTraits<MODE>::NTAPS - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
            const long lim = (1L << 31) - 8192;
            if ((long)a.B * hi * wi * cmax * 2L >= lim || wel * 2L >= lim || M * a.Cout * 2L >= lim || M >= (1L << 30)) return 0;
            if ((M + 256) * (a.Wo > a.Ho ? a.Wo : a.Ho) >= (1L << 32)) return 0;      // multiply-high division is exact
        }
        long ks = 1;
        if (a.partial && tiles < wgs) {
            ks = (wgs + tiles / 2) / tiles;                      // ~one workgroup per CU
            if (ks > nit / 8) ks = nit / 8;
            while (ks > 1 && ks * M * a.Cout > a.partial_cap) --ks;
            if (ks < 1) ks = 1;
        }
        // Round 6 (gpurun R6at): a stride-2 data gradient (up-convolution) whose reduction would be split in TWO takes the
        // unsplit 64 x 128 tiles of conv_glds instead when those fill the chip: at configs[1] the 256 -> 128 up-conv's data
        // gradient (16 K output pixels, K = 1152) ran 21.3 us + a 17.2-us finish pass (33 MB of fp32 partials, BatchNorm-backward
        // sums) against 29.3 us as ONE launch of 512 workgroups with the sums in its epilogue. With 4 or 8 splits the pair wins
        // (30.9 against 37.4 us, 30.0 against 37.3 us on the two deeper up-convs).
        if (MODE == CONV3S2 && ks == 2 && (long)cdiv(a.Cout, 64) * cdiv(M, 128) >= 384) return 0;
        ConvArgs b = a; b.dbg = dbg;
        const bool ragged = (a.C0 % 64) != 0 || (a.C1 % 64) != 0;
        int rc;
        if (dbg & 32) {
            static unsigned long long* dbuf = nullptr;
            if (!dbuf) { MPU_CHECK_HIP(hipMalloc(&dbuf, 4 * 64 * 8)); }
            b.dbg_buf = dbuf;
            MPU_CHECK_HIP(hipMemsetAsync(dbuf, 0, 4 * 64 * 8, st));
            rc = ragged ? launch_pipe<MODE, true, 1>(b, (int)ks, st) : launch_pipe<MODE, false, 1>(b, (int)ks, st);
            if (!rc) {                                           // dev aid: print the stamps of this launch
                unsigned long long h[4 * 64];
                MPU_CHECK_HIP(hipStreamSynchronize(st));
                MPU_CHECK_HIP(hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost));
                for (int w = 0; w < 4; ++w) {
                    const int n = (int)h[w * 64 + 63];
                    fprintf(stderr, "pipe stamps M=%ld N=%d ks=%ld wg%d wave%d:", M, a.Cout, ks, w < 2 ? 0 : 131, (w & 1) ? 5 : 0);
                    for (int k = 1; k < n && k < 63; ++k) fprintf(stderr, " %lld", (long long)(h[w * 64 + k] - h[w * 64 + k - 1]));
                    fprintf(stderr, "\n");
                }
            }
        } else {
            rc = ragged ? launch_pipe<MODE, true, 0>(b, (int)ks, st) : launch_pipe<MODE, false, 0>(b, (int)ks, st);
        }
        return rc ? rc : 1;
    }
}

static thread_local const char* g_glds_sched = "glds";       // which schedule the last launch_conv_glds took (schedule log)
const char* last_glds_schedule() { return g_glds_sched; }

template <typename T, int MODE>
static int launch_glds_mode(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    a.ksplit = 1;
    g_glds_sched = "glds";
    {
        const int p = try_pipe<T, MODE>(a, st);
        if (p != 0) { g_glds_sched = "pipe"; return p < 0 ? p : MPU_OK; }
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long t128 = (long)cdiv(a.Cout, 128) * cdiv(M, 128);
    const long t64x128 = (long)cdiv(a.Cout, 64) * cdiv(M, 128);
    if (a.Cout > 64 && t128 >= 384) return launch_glds_cfg<T, MODE, 128, 128, 64, 64>(a, st);
    // few output tiles but a long reduction (deep U-Net levels): 128x128 tiles, K split over workgroups
    if (a.Cout >= 128 && a.partial && t128 < 256) {
        constexpr int BKE = 128 / sizeof(T);
        const int nit = GModeTraits<MODE>::NTAPS * (cdiv(a.C0, BKE) + cdiv(a.C1, BKE));
        constexpr long sk_target = 512, sk_max = 8;              // ~2 workgroups per CU, at most 8 partial copies
        long ks = sk_target / (t128 > 0 ? t128 : 1);
        if (ks > sk_max) ks = sk_max;
        if (ks > nit / 8) ks = nit / 8;
        while (ks > 1 && ks * M * a.Cout > a.partial_cap) --ks;
        if (ks > 1) { a.ksplit = (int)ks; return launch_glds_cfg<T, MODE, 128, 128, 64, 64>(a, st); }
    }
    if (t64x128 >= 384 || a.Cout <= 64) {
        if (M >= 128 * 64) return launch_glds_cfg<T, MODE, 64, 128, 64, 32>(a, st);
    }
    return launch_glds_cfg<T, MODE, 64, 64, 32, 32>(a, st);
}

int launch_conv_glds(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
#define MPU_GCASE(TT)                                                              \
    switch (mode) {                                                                \
        case CONV3: return launch_glds_mode<TT, CONV3>(a, st);                     \
        case UPCONV2: return launch_glds_mode<TT, UPCONV2>(a, st);                 \
        case CONV3S2: return launch_glds_mode<TT, CONV3S2>(a, st);                 \
        case CONV1: return launch_glds_mode<TT, CONV1>(a, st);                     \
        default: return fail(MPU_EINVAL, "%s", "conv: bad mode");                  \
    }
    if (dtype == MPU_BF16) { MPU_GCASE(bf16_t) }
    if (dtype == MPU_F32) { MPU_GCASE(float) }
#undef MPU_GCASE
    return fail(MPU_EINVAL, "%s", "conv: bad dtype");
}


// ------------------------------------------------------------------------- //
// wgrad_glds_kernel: dW[tap][ci][co] = sum_m X[m@tap][ci] * dZ[m][co] with LDS-DMA staging.
// Both operands are pixel-major (K-major): per K step of 32 pixels the stage holds
// X[32][BCI] and dZ[32][BCO] as unpadded rows; fragments are read with the LDS transpose read
// ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32). The four k-rows a transpose read touches
// are spread over four 64-byte bank groups by XOR-ing the 64-byte granule index of the row
// (256-byte rows: granule ^= row&3; 128-byte rows: granule ^= (row>>1)&1) on the DMA source side.
// ------------------------------------------------------------------------- //
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int ROWB> __device__ __forceinline__ int wg_swz16(int row) {   // XOR mask on the 16-byte slot index
    return ROWB == 256 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2);
}

template <typename T, int BCI, int BCO> struct WgradGldsCfg {
    static constexpr int KP = 32, RX = BCI * sizeof(T), RZ = BCO * sizeof(T), STAGE = KP * (RX + RZ), NSTAGE = 3;
    static constexpr int SMEM = NSTAGE * STAGE;
};

// one workgroup: bid = its index inside the job's 1-D grid, smem = the kernel's LDS array (WgradGldsCfg::SMEM bytes)
template <typename T, int MODE, int BCI, int BCO>
__device__ __forceinline__ void wgrad_glds_body(const WgradArgs& a, const unsigned bid, unsigned char* smem) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int KP = 32;
    constexpr int RX = BCI * sizeof(T), RZ = BCO * sizeof(T);            // row bytes: 128 or 256
    static_assert((RX == 128 || RX == 256) && (RZ == 128 || RZ == 256), "row bytes");
    constexpr int STAGE = KP * (RX + RZ);
    constexpr int NSTAGE = 3;
    constexpr int PX = KP * RX / 1024, PZ = KP * RZ / 1024;              // 1-KiB pieces per image
    constexpr int NPX = PX / 4, NPZ = PZ / 4;                            // pieces per wave
    static_assert(NPX >= 1 && NPZ >= 1, "tile too small");
    constexpr int WCI = BCI / 2, WCO = BCO / 2, TI = WCI / 32, TJ = WCO / 32;
    constexpr int KW = GModeTraits<MODE>::KW;
    static_assert(NSTAGE * STAGE == WgradGldsCfg<T, BCI, BCO>::SMEM, "LDS size");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    const int Cin = a.C0 + a.C1;
    const int tiles_co = (a.Cout + BCO - 1) / BCO;
    // XCD-aware decode of the 1-D grid: all taps (and tiles) of one pixel chunk z run on ONE XCD
    // (workgroup g is dispatched to XCD g % 8), back to back, so x and dZ of the chunk are fetched
    // from HBM once and re-read from that XCD's L2 by the other taps (PMC: 9x less FETCH_SIZE).
    constexpr int NTAPS_ = GModeTraits<MODE>::NTAPS;
    const int ntile = tiles_co * ((Cin + BCI - 1) / BCI);
    // locality unit = (pixel chunk z, output tile): its taps sit on one XCD, units are dealt round-robin
    const int xcd = bid & 7, slot = bid >> 3;
    const int tap = slot % NTAPS_, unit = (slot / NTAPS_) * 8 + xcd;
    if (unit >= a.ksplit * ntile) return;
    const int tile = unit % ntile, zsplit = unit / ntile;
    const int ci0 = (tile / tiles_co) * BCI, co0 = (tile % tiles_co) * BCO;
    const int ky = tap / KW, kx = tap % KW;
    const long M = (long)a.B * a.Ho * a.Wo;
    const long mbeg = (long)zsplit * a.mchunk;
    const long mend = (mbeg + a.mchunk < M) ? mbeg + a.mchunk : M;
    const int Hi = g_in_h<MODE>(a.Ho), Wi = g_in_h<MODE>(a.Wo);
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)a.B * Hi * Wi;
    // the ci tile lies entirely in one concat source (host guarantees C0 % BCI == 0 when C1 > 0)
    const bool s1 = ci0 >= a.C0 && a.C1 > 0;
    const int Cs = s1 ? a.C1 : a.C0, cs0 = s1 ? ci0 - a.C0 : ci0;
    const i32x4 rsx = make_rsrc(s1 ? a.x1 : a.x0, npix * Cs * (long)sizeof(T));
    const i32x4 rsz = make_rsrc(a.dz, M * a.Cout * (long)sizeof(T));
    const unsigned lds0 = (unsigned)(uintptr_t)smem;

    // per-lane DMA roles
    constexpr int SPRX = RX / 16, SPRZ = RZ / 16;                        // 16-byte slots per row
    int xrow[NPX], xch[NPX], zrow[NPZ], zch[NPZ];
#pragma unroll
    for (int g = 0; g < NPX; ++g) {
        const int piece = wave * NPX + g;
        xrow[g] = piece * (64 / SPRX) + lane / SPRX;
        xch[g] = ((lane % SPRX) ^ wg_swz16<RX>(xrow[g])) * EPC;
    }
#pragma unroll
    for (int g = 0; g < NPZ; ++g) {
        const int piece = wave * NPZ + g;
        zrow[g] = piece * (64 / SPRZ) + lane / SPRZ;
        zch[g] = ((lane % SPRZ) ^ wg_swz16<RZ>(zrow[g])) * EPC;
    }
    // Requests advance by KP pixels per step. Every lane keeps the (image, oy, ox) of its next X pixel; round 3: no
    // loops and no divergent branches in the advance (the former `while (ox >= Wo)` carry ran 2-4 masked iterations per
    // piece at the deep levels' 8- and 16-pixel rows and the step carried ~350 instructions beside its 8 MFMAs):
    //   * Wo divides KP (deep levels): ox is a lane constant, oy advances by KP / Wo with at most one wrap;
    //   * otherwise the coordinates are decoded from the pixel index with multiply-high divisions (exact for
    //     M * max(Wo, Ho) < 2^32: wgrad_glds_supported).
    // dZ rows beyond the chunk's end are zero (dZ descriptor bounded to mend), so the X side needs no `m < mend` test:
    // whatever it fetches for those rows is multiplied by zero. ASSUMPTION: the X rows of the neighbouring chunk
    // are FINITE -- 0 * Inf / 0 * NaN would put a NaN into this chunk's partials although the offending activation belongs
    // to the next chunk. Activations only stop being finite after an overflow upstream, which bench.py's guard and the
    // loss already report; when localising such a NaN remember that it may sit one K chunk (mchunk pixels) further on.
    int xm[NPX], xox[NPX], xoy[NPX], xb[NPX];
    const unsigned magic_w = 0xffffffffu / (unsigned)a.Wo + 1u, magic_h = 0xffffffffu / (unsigned)a.Ho + 1u;   // ceil(2^32 / d)
    auto decode = [&](int g) {
        const unsigned m = (unsigned)xm[g];
        const unsigned t = a.Wo == 1 ? m : __umulhi(m, magic_w);         // (ceil(2^32 / 1) does not fit 32 bits: 1x1 / Nx1 maps)
        xox[g] = (int)(m - t * (unsigned)a.Wo);
        const unsigned bb = a.Ho == 1 ? t : __umulhi(t, magic_h);
        xoy[g] = (int)(t - bb * (unsigned)a.Ho); xb[g] = (int)bb;
    };
#pragma unroll
    for (int g = 0; g < NPX; ++g) { xm[g] = (int)(mbeg + xrow[g]); decode(g); }
    const int dY = KP / a.Wo;
    const bool fastw = dY * a.Wo == KP && dY <= a.Ho;            // (workgroup-uniform)
    const i32x4 rszb = make_rsrc(a.dz, mend * a.Cout * (long)sizeof(T));          // dZ rows >= mend read as zeros
    unsigned zoffl[NPZ];                                         // running dZ offsets (poison: channel beyond Cout)
#pragma unroll
    for (int g = 0; g < NPZ; ++g)
        zoffl[g] = co0 + zch[g] < a.Cout ? (unsigned)(((mbeg + zrow[g]) * a.Cout + co0 + zch[g]) * (long)sizeof(T)) : PIPE_POISON;
    const unsigned zstep = (unsigned)(KP * a.Cout * (int)sizeof(T));
    auto issue = [&](long /*mb: sequential, KP apart*/, int stage) {
        const unsigned sb = lds0 + stage * STAGE;
#pragma unroll
        for (int g = 0; g < NPX; ++g) {
            int iy, ix;
            const bool v = g_tap_src<MODE>(xoy[g], xox[g], ky, kx, a.Ho, a.Wo, iy, ix) && cs0 + xch[g] < Cs;
            const unsigned off = v ? (unsigned)((((xb[g] * Hi + iy) * Wi + ix) * Cs + cs0 + xch[g]) * (int)sizeof(T)) : OOB;
            dma16(rsx, off, sb + (wave * NPX + g) * 1024);
            xm[g] += KP;
            if (fastw) {
                xoy[g] += dY;
                const bool wrap = xoy[g] >= a.Ho;
                xoy[g] = wrap ? xoy[g] - a.Ho : xoy[g];
                xb[g] += wrap ? 1 : 0;
            } else decode(g);
        }
#pragma unroll
        for (int g = 0; g < NPZ; ++g) {
            dma16(rszb, zoffl[g], sb + KP * RX + (wave * NPZ + g) * 1024);
            zoffl[g] += zstep;                                   // (poison + steps stays out of range: M * Cout bytes < 2 GiB - 8 KiB)
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int stage) {
        const unsigned char* xb = smem + stage * STAGE;
        const unsigned char* zb = xb + KP * RX;
        if constexpr (sizeof(T) == 2) {
            const int krow = 8 * (lane >> 5) + ((lane & 15) >> 2);        // + s*16 (+4 for the upper half)
            const int ccol = 16 * ((lane >> 4) & 1) + (lane & 3) * 4;     // element column inside a 32-block
            // byte offset of (block col + ccol) in a row, swizzled; row&3 and (row>>1)&1 are lane constants
            auto xoff = [&](int blk) { const int byte = (blk + ccol) * 2;
                return (((byte >> 4) ^ wg_swz16<RX>(krow)) << 4) + (byte & 15); };
            auto zoff = [&](int blk) { const int byte = (blk + ccol) * 2;
                return (((byte >> 4) ^ wg_swz16<RZ>(krow)) << 4) + (byte & 15); };
            s16x8 af[KP / 16][TI], bf[KP / 16][TJ];               // all fragments of the step first, then the MFMAs
#pragma unroll
            for (int s = 0; s < KP / 16; ++s) {
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const unsigned char* p = xb + (s * 16 + krow) * RX + xoff(wi * WCI + i * 32);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * RX));
                    af[s][i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const unsigned char* p = zb + (s * 16 + krow) * RZ + zoff(wj * WCO + j * 32);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * RZ));
                    bf[s][j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KP / 16; ++s)
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bf[s][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else if (a.x3) {
            // dtype "bf16x3": eight pixels (K) per lane and product -- lanes 0-31 rows k .. k+3 and k+8 .. k+11, lanes 32-63 the
            // rows 4 further on; both operands in the same order (common.h x3_split)
#pragma unroll 2
            for (int k = 0; k < KP; k += 16) {
                s16x8 ah[TI], al[TI], bh[TJ], bl[TJ];
                const int r0 = k + 4 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int byte = (wi * WCI + i * 32 + (lane & 31)) * 4;
                    uint32_t v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = r0 + (e & 3) + 8 * (e >> 2);
                        v[e] = *(const uint32_t*)(xb + row * RX + (((byte >> 4) ^ wg_swz16<RX>(row)) << 4) + (byte & 15));
                    }
                    x3_split(make_uint4(v[0], v[1], v[2], v[3]), make_uint4(v[4], v[5], v[6], v[7]), ah[i], al[i]);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int byte = (wj * WCO + j * 32 + (lane & 31)) * 4;
                    uint32_t v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int row = r0 + (e & 3) + 8 * (e >> 2);
                        v[e] = *(const uint32_t*)(zb + row * RZ + (((byte >> 4) ^ wg_swz16<RZ>(row)) << 4) + (byte & 15));
                    }
                    x3_split(make_uint4(v[0], v[1], v[2], v[3]), make_uint4(v[4], v[5], v[6], v[7]), bh[j], bl[j]);
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) x3_mma(ah[i], al[i], bh[j], bl[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < KP; k += 2) {
                float af[TI], bf[TJ];
                const int row = k + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const int byte = (wi * WCI + i * 32 + (lane & 31)) * 4;
                    af[i] = *(const float*)(xb + row * RX + (((byte >> 4) ^ wg_swz16<RX>(row)) << 4) + (byte & 15));
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int byte = (wj * WCO + j * 32 + (lane & 31)) * 4;
                    bf[j] = *(const float*)(zb + row * RZ + (((byte >> 4) ^ wg_swz16<RZ>(row)) << 4) + (byte & 15));
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // bias gradient (a.fuse_db): every workgroup sums the dZ tile of its K steps it % nshare == share
    // (share = tap * tiles_ci + ci tile), so the extra LDS reads are spread evenly over all workgroups.
    // thread -> 16-byte channel chunk (tid % CPZ) of rows (tid / CPZ) + k * (256 / CPZ)
    constexpr int CPZ = RZ / 16;
    const int tiles_ci = (Cin + BCI - 1) / BCI;
    const int nshare = NTAPS_ * tiles_ci, share = tap * tiles_ci + (tile / tiles_co);
    float dbacc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) dbacc[e] = 0.f;
    auto colsum_stage = [&](int stage) {
        const unsigned char* zb = smem + stage * STAGE + KP * RX;
        const int c = tid % CPZ;
        for (int r = tid / CPZ; r < KP; r += 256 / CPZ) {
            const uint4 v = *(const uint4*)(zb + r * RZ + ((c ^ wg_swz16<RZ>(r)) << 4));
            if (sizeof(T) == 2) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dbacc[2 * e] += __uint_as_float(w[e] << 16);
                    dbacc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            } else {
                dbacc[0] += __uint_as_float(v.x); dbacc[1] += __uint_as_float(v.y);
                dbacc[2 % EPC] += __uint_as_float(v.z); dbacc[3 % EPC] += __uint_as_float(v.w);
            }
        }
    };

    const int nit = (int)((mend - mbeg + KP - 1) / KP);
    constexpr int NLD = NPX + NPZ;
    if (nit > 0) {
        issue(mbeg, 0);
        if (nit > 1) { issue(mbeg + KP, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int st = 0;
        int dbc = a.fuse_db ? share : -1;                        // steps until this workgroup's next bias-gradient share (it % nshare == share)
        for (int it = 0; it < nit; ++it) {
            const int st2 = st >= 1 ? st - 1 : 2;
            if (it + 2 < nit) issue(mbeg + (long)(it + 2) * KP, st2);
            compute(st);
            if (dbc == 0) { colsum_stage(st); dbc = nshare; }
            if (dbc > 0) --dbc;
            if (it + 2 < nit) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            st = st == 2 ? 0 : st + 1;
        }
    }
    float* P = a.partial + ((long)zsplit * NTAPS_ + tap) * (long)Cin * a.Cout;
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
    if (a.fuse_db) {        // reduce the 256/CPZ row-groups per chunk through LDS (staging area is free now)
        float* red = (float*)smem;                 // [256][EPC]
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[tid * EPC + e] = dbacc[e];
        __syncthreads();
        if (tid < BCO) {
            const int c = tid / EPC, e = tid % EPC;
            float s = 0.f;
            for (int k = 0; k < 256 / CPZ; ++k) s += red[(c + k * CPZ) * EPC + e];
            if (co0 + tid < a.Cout)
                a.db_partial[((long)zsplit * nshare + share) * a.Cout + co0 + tid] = s;
        }
    }
}

template <typename T, int MODE, int BCI, int BCO>
__global__ __launch_bounds__(256, 2) void wgrad_glds_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WgradGldsCfg<T, BCI, BCO>::SMEM];
    wgrad_glds_body<T, MODE, BCI, BCO>(a, blockIdx.x, smem);
}

// Every deferred wgrad_glds job (bf16, 128 x 128 channel tiles: the deep levels) of a backward pass in ONE launch
// (kernels.h: WgradGroup): workgroup -> job by the table's block ranges (multiples of 8: the XCD decode stays aligned).
struct GldsGroupTable { int njobs, _pad; GldsGroupJob job[GLDS_GROUP_MAX]; };
__global__ __launch_bounds__(256, 2) void wgrad_glds_group_kernel(GldsGroupTable t) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WgradGldsCfg<bf16_t, 128, 128>::SMEM];
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < t.njobs; ++k) if ((int)blockIdx.x >= t.job[k].blk_begin) j = k;
    const WgradArgs a = t.job[j].a;
    const unsigned bid = blockIdx.x - (unsigned)t.job[j].blk_begin;
    if (t.job[j].mode == UPCONV2) wgrad_glds_body<bf16_t, UPCONV2, 128, 128>(a, bid, smem);
    else wgrad_glds_body<bf16_t, CONV3, 128, 128>(a, bid, smem);
}

bool wgrad_glds_supported(int dtype, int mode, const WgradArgs& a) {
    const int esz = dtype == MPU_BF16 ? 2 : 4;
    if (mode != CONV3 && mode != UPCONV2 && mode != CONV1) return false;
    const long M = (long)a.B * a.Ho * a.Wo;
    const long hi = mode == UPCONV2 ? a.Ho / 2 : a.Ho, wi = mode == UPCONV2 ? a.Wo / 2 : a.Wo;
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    if ((long)a.B * hi * wi * cmax * esz >= (1L << 31) - 8192 || M * a.Cout * esz >= (1L << 31) - 8192) return false;
    if ((M + 64) * (a.Wo > a.Ho ? a.Wo : a.Ho) >= (1L << 32)) return false;     // the multiply-high pixel decode is exact below this
    const int Cin = a.C0 + a.C1;
    const bool big = esz == 2 && Cin >= 128 && a.Cout >= 128 && (a.C1 == 0 || a.C0 % 128 == 0);
    if (!big && a.C1 > 0 && a.C0 % 64 != 0) return false;
    return true;
}

// returns 1 if launched, 0 if this shape must use the register-staged kernel, <0 on error
template <typename T, int MODE>
static int try_wgrad_glds_mode(const WgradArgs& a, hipStream_t st) {
    const int Cin = a.C0 + a.C1;
    constexpr int ntaps = GModeTraits<MODE>::NTAPS;
    if (!wgrad_glds_supported(sizeof(T) == 2 ? MPU_BF16 : MPU_F32, MODE, a)) return 0;
    bool big = false;
    if constexpr (sizeof(T) == 2) big = Cin >= 128 && a.Cout >= 128 && (a.C1 == 0 || a.C0 % 128 == 0);
    if (big) {
        if constexpr (sizeof(T) == 2) {
            const long units = (long)a.ksplit * cdiv(Cin, 128) * cdiv(a.Cout, 128);
            const long g = 8 * ((units + 7) / 8) * ntaps;
            launch_k(wgrad_glds_kernel<T, MODE, 128, 128>, dim3((unsigned)g), dim3(256), 0, st, a);
        }
    } else {
        const long units = (long)a.ksplit * cdiv(Cin, 64) * cdiv(a.Cout, 64);
        const long g = 8 * ((units + 7) / 8) * ntaps;
        launch_k(wgrad_glds_kernel<T, MODE, 64, 64>, dim3((unsigned)g), dim3(256), 0, st, a);
    }
    int rc = launch_ok();
    return rc ? rc : 1;
}

// workgroups of a job on the groupable variant (bf16, 3x3 or 2x2, 128 x 128 channel tiles); 0 = some other variant
long wgrad_glds_grid(int mode, const WgradArgs& a) {
    if ((mode != CONV3 && mode != UPCONV2) || !wgrad_glds_supported(MPU_BF16, mode, a)) return 0;
    const int Cin = a.C0 + a.C1;
    if (!(Cin >= 128 && a.Cout >= 128 && (a.C1 == 0 || a.C0 % 128 == 0))) return 0;
    const long units = (long)a.ksplit * cdiv(Cin, 128) * cdiv(a.Cout, 128);
    return 8 * ((units + 7) / 8) * (mode == UPCONV2 ? 4 : 9);
}

int launch_wgrad_glds_group(int dtype, const GldsGroupJob* jobs, int n, hipStream_t st) {
    if (n <= 0) return MPU_OK;
    if (dtype != MPU_BF16 || n > GLDS_GROUP_MAX) return fail(MPU_EINVAL, "%s", "wgrad_glds group: bad job list");
    GldsGroupTable t; t.njobs = n; t._pad = 0;
    long grid = 0;
    for (int k = 0; k < n; ++k) {
        t.job[k] = jobs[k];
        t.job[k].blk_begin = (int)grid;
        const long g = wgrad_glds_grid(jobs[k].mode, jobs[k].a);
        if (g <= 0) return fail(MPU_EINVAL, "%s", "wgrad_glds group: job is not on the grouped variant");
        grid += g;
    }
    launch_k(wgrad_glds_group_kernel, dim3((unsigned)grid), dim3(256), 0, st, t);
    return launch_ok();
}

int try_wgrad_glds(int dtype, int mode, const WgradArgs& a, hipStream_t st) {
#define MPU_WGG(TT)                                                               \
    switch (mode) {                                                               \
        case CONV3: return try_wgrad_glds_mode<TT, CONV3>(a, st);                 \
        case UPCONV2: return try_wgrad_glds_mode<TT, UPCONV2>(a, st);             \
        case CONV1: return try_wgrad_glds_mode<TT, CONV1>(a, st);                 \
        default: return 0;                                                        \
    }
    if (dtype == MPU_BF16) { MPU_WGG(bf16_t) }
    if (dtype == MPU_F32) { MPU_WGG(float) }
#undef MPU_WGG
    return 0;
}

}  // namespace mpu
