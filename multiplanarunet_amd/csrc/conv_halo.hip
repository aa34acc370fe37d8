// conv_halo_kernel: 3x3 SAME convolution (forward, and data gradient with rotated weights) whose
// input patch stays resident in LDS across the nine taps.
//
// The L2 -> LDS fill rate (~57 GB/s per CU measured) bounds the plain implicit GEMM of
// conv_glds.hip at ~35-40 % of the MFMA peak for 128x128 tiles: every K step re-fetches a
// 128-pixel operand tile although consecutive taps read the same pixels shifted by one. Here a
// workgroup owns TH x 32 output pixels x BN channels and, per 64-channel chunk, DMAs the
// (TH+2) x 34 pixel halo patch ONCE; the nine taps then read their B fragments from the patch at
// shifted rows (row = (py+ky)*34 + px+kx). Only the [BN][64ch] weight tile changes per tap
// (3-stage LDS ring, counted vmcnt). Fill bytes per chunk drop from 9*(BN+BM)*128 to
// PATCH*128 + 9*BN*128 (BN=BM=128: 288 KB -> 170 KB; BN=64: 216 KB -> 98 KB).
// The XOR swizzle (slot ^= (row>>1)&7) is applied on patch-row indices; 32 consecutive patch rows
// starting at ANY row are conflict-free for ds_read_b128 (checked for even and odd starts).
#include <stdlib.h>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

__device__ __forceinline__ i32x4 h_make_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa; r.y = (int)((unsigned)(pa >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void h_dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
constexpr unsigned HALO_POISON = 0x80001000u;                    // + any in-range byte offset (< 2 GiB - 8 KiB) stays >= num_records
template <typename T> struct HMma;
template <> struct HMma<bf16_t> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
    }
};
template <> struct HMma<float> {
    static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

// MODE: CONV3 (3x3 SAME: halo of one pixel) or UPCONV2 (nearest-upsample x2 + 2x2 SAME with TF's 0/1 padding: the
// patch lives at the LOW resolution, output pixel (oy, ox) and tap (ky, kx) read low-res pixel ((oy+ky)>>1, (ox+kx)>>1))
template <typename T, int BN, int TH, int NWS_, int MODE = CONV3>
struct HaloCfg {
    static constexpr int NT = MODE == UPCONV2 ? 4 : 9;
    static constexpr int KW = MODE == UPCONV2 ? 2 : 3;
    static constexpr int TW = 32, PW = MODE == UPCONV2 ? TW / 2 + 2 : TW + 2;
    static constexpr int PH = MODE == UPCONV2 ? TH / 2 + 1 : TH + 2;
    static constexpr int PROWS = (PH * PW + 7) / 8 * 8;          // patch rows, padded to whole DMA pieces
    static constexpr int PATCH = PROWS * 128;
    static constexpr int WSTAGE = BN * 128, NWS = NWS_;
    static constexpr int BM = TH * TW;
    static constexpr int OROW = BN * (int)sizeof(T) + 16;
    static constexpr int EPI = BM * OROW + 3 * BN * 4;
    static constexpr int MAIN = PATCH + NWS * WSTAGE;
    static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
};

// DBG (dev aid; builds with -DMPU_HALO_KNOCKOUT_BUILD only, selected by MPU_HALO_KNOCKOUT=bits): knock-out timing of the
// kernel's components, the mask a COMPILE-TIME constant (a run-time mask spilled 1.1 KB of registers and ran 30x slower:
// gpurun R4a) -- bit 0: no global stores, 1: no MFMAs, 2: no weight requests after the prologue, 3: no fragment reads,
// 4: no patch reloads, 5: no epilogue at all, 6: no per-tap barrier. Results are garbage by design.
template <typename T, int BN, int TH, int NWS, int MODE = CONV3, int DBG = 0>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(ConvArgs a) {
    using Cfg = HaloCfg<T, BN, TH, NWS, MODE>;
    constexpr int ko = DBG;
    // bit 8 of the compile-time mask (no knock-out uses it): the split-bf16 instantiation of the f32 kernel (dtype "bf16x3",
    // round 6). A run-time switch inside the unrolled tap loop cost the exact-f32 path 37 registers and 244 bytes of scratch.
    constexpr bool X3 = (DBG & 256) != 0;
    constexpr int NT = Cfg::NT, KW = Cfg::KW;
    constexpr int EPC = 16 / sizeof(T), BKE = 128 / sizeof(T);
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int WAVES_N = BN / 64, WAVES_M = 4 / WAVES_N;
    constexpr int TN = 2, TM = TH / WAVES_M;
    static_assert(TM >= 1 && TM * WAVES_M == TH, "tile split");
    constexpr int NPP = PROWS / 8;                               // patch DMA pieces
    constexpr int NPW = (NPP + 3) / 4;                           // ... per wave
    constexpr int GW = BN / 32;                                  // weight DMA pieces per wave
    constexpr int BM = Cfg::BM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    // n-tile fastest: the n-tiles of one pixel tile are neighbours in time (shared patch in L2)
    const int logical = a.xcd ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;      // (kernels.h)
    int t = logical;
    const int n0 = (t % tiles_n) * BN; t /= tiles_n;
    const int x0 = (t % tiles_x) * TW; t /= tiles_x;
    const int y0 = (t % tiles_y) * TH; const int b = t / tiles_y;
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    constexpr unsigned OOB = 0xfffffff0u;
    const int Hi = MODE == UPCONV2 ? H / 2 : H, Wi = MODE == UPCONV2 ? W / 2 : W;     // input resolution
    const long npix = (long)a.B * Hi * Wi;
    const i32x4 rs0 = h_make_rsrc(a.in0, npix * a.C0 * (long)sizeof(T));
    const i32x4 rs1 = h_make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * (long)sizeof(T) : 0);
    const i32x4 rsw = h_make_rsrc(a.w, a.w_elems * (long)sizeof(T));
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + Cfg::PATCH;

    // Epilogue constants of this thread's output channel, requested NOW so that their latency hides under the main
    // loop (round 3: the epilogue used to start with an exposed global load). Unconditional loads from a pointer that is
    // always valid (the weights stand in for a null table; the value is discarded by a select in the epilogue).
    const bool early_ok = tid < BN && n0 + tid < a.Cout;
    const int eidx = early_ok ? n0 + tid : 0;
    const float early_bias_raw = (a.bias ? a.bias : (const float*)a.w)[eidx];
    const float early_scale = (a.post_scale ? a.post_scale : (const float*)a.w)[eidx];
    const float early_shift_raw = (a.post_scale ? a.post_shift : (const float*)a.w)[eidx];

    // --- per-lane DMA roles -------------------------------------------------------------
    const int lrow = lane >> 3, slot = lane & 7;
    int ppix[NPW], pchunk[NPW];                                 // patch: input pixel index (or -1), source chunk
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int piece = wave + 4 * k;
        const int pr = piece * 8 + lrow;                         // patch row
        const int py = pr / PW, px = pr % PW;
        const int iy = MODE == UPCONV2 ? y0 / 2 + py : y0 + py - 1;
        const int ix = MODE == UPCONV2 ? x0 / 2 + px : x0 + px - 1;
        const bool v = piece < NPP && pr < Cfg::PH * PW && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        ppix[k] = v ? (b * Hi + iy) * Wi + ix : (int)npix;      // padding: the first pixel BEYOND the tensor (out of range for either source)
        pchunk[k] = slot ^ ((pr >> 1) & 7);
    }
    unsigned wrow[GW]; int wchunk[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int rl = wave * (BN / 4) + g * 8 + lrow;
        const int n = n0 + rl;
        wrow[g] = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * (long)sizeof(T)) : OOB;
        wchunk[g] = slot ^ ((rl >> 1) & 7);
    }
    auto chunk_src = [&](int cc, bool& s1, int& cbase, int& Cs) {
        s1 = cc >= nch0; cbase = (s1 ? cc - nch0 : cc) * BKE; Cs = s1 ? a.C1 : a.C0;
    };
    auto issue_patch = [&](int cc) {
        bool s1; int cbase, Cs; chunk_src(cc, s1, cbase, Cs);
        const bool tail = Cs - cbase < BKE;                      // only a tail chunk masks channels
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int piece = wave + 4 * k;
            if (piece < NPP) {                                   // wave-uniform
                const int ch = cbase + pchunk[k] * EPC;
                unsigned off = (unsigned)((ppix[k] * Cs + ch) * (int)sizeof(T));
                if (tail) off = ch < Cs ? off : OOB;
                if (s1) h_dma16(rs1, off, lds0 + piece * 1024);
                else    h_dma16(rs0, off, lds0 + piece * 1024);
            }
        }
    };
    // Weight tile of the NEXT not-yet-requested step: running (tap, chunk) counters and a per-lane byte offset
    // computed once, so a request is one add + the DMA (the former per-step index arithmetic cost ~400 cycles of
    // every ~1800-cycle step).
    unsigned wlane[GW]; int wch[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        wch[g] = wchunk[g] * EPC;
        wlane[g] = wrow[g] == OOB ? OOB : wrow[g] + (unsigned)(wch[g] * (int)sizeof(T));
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    // kv = k-steps of 16 channels that hold any channel of the chunk (a 96-channel source: 4 + 2); the rest of a tail
    // chunk is zero fill and its MFMAs are skipped in pairs
    auto compute = [&](int ky, int kx, int stage, int kv) {
        // (with the taps unrolled the fragment addresses of all nine taps are loop invariants; the 8 x 2 accumulator
        // tile has no registers for them -- 140 spilled -- so there the lane term is made opaque per tap and the
        // ~40 address instructions are recomputed, as the rolled loop did)
        int l31 = lane & 31;
        if constexpr (TN * TM >= 8) asm volatile("" : "+v"(l31));
        // byte offsets of k-step 0; k-step s flips bits 5-6 of the swizzled 16-byte slot: ((2s + fh) ^ sw) << 4 ==
        // ((fh ^ sw) << 4) ^ (s << 5) (rows are 128 bytes, so those bits belong to the slot alone): one XOR per read
        const unsigned wo = (unsigned)(Cfg::PATCH + stage * Cfg::WSTAGE + (wn * 64 + l31) * 128) + (unsigned)((fh ^ fsw) << 4);
        unsigned po[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = MODE == UPCONV2 ? ((wm * TM + j + ky) >> 1) * PW + ((kx + l31) >> 1)
                                             : (wm * TM + j + ky) * PW + kx + l31;   // patch row of this lane's pixel
            po[j] = (unsigned)(prow * 128) + (unsigned)((fh ^ ((prow >> 1) & 7)) << 4);
        }
        // All fragments of the tap are requested before the first MFMA (sched_barrier keeps the compiler from sinking
        // the reads back next to their uses, which serialises LDS latency -> wait -> MFMA and ran the loop at ~1/3
        // of the matrix pipe's rate); the MFMAs then wait on lgkmcnt progressively.
        constexpr int SG = (TN + TM > 4) ? 2 : 4;                // k-steps requested together (register budget)
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += SG) {
            if (g0 >= kv) break;                                 // (workgroup-uniform)
            uint4 af[SG][TN], bf[SG][TM];
#pragma unroll
            for (int s = 0; s < SG; ++s) {
                const unsigned ks = (unsigned)((g0 + s) << 5);
                if (DBG && (ko & 8)) {
#pragma unroll
                    for (int i = 0; i < TN; ++i) af[s][i] = make_uint4(ks, wo, 0, 0);
#pragma unroll
                    for (int j = 0; j < TM; ++j) bf[s][j] = make_uint4(ks, po[j], 0, 0);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) af[s][i] = *(const uint4*)(smem + (wo ^ ks) + i * 32 * 128);
#pragma unroll
                for (int j = 0; j < TM; ++j) bf[s][j] = *(const uint4*)(smem + (po[j] ^ ks));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DBG && (ko & 2)) {                               // (keep the fragments alive without the matrix pipe)
#pragma unroll
                for (int s = 0; s < SG; ++s)
#pragma unroll
                    for (int i = 0; i < TN; ++i)
#pragma unroll
                        for (int j = 0; j < TM; ++j) acc[i][j][s] += __uint_as_float(af[s][i].x ^ bf[s][j].y);
            } else if constexpr (sizeof(T) == 4 && X3) {         // dtype "bf16x3": pairs of k-steps as split-bf16 products (common.h)
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
            } else
#pragma unroll
            for (int s = 0; s < SG; ++s) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) HMma<T>::run(af[s][i], bf[s][j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // --- pipeline: patch once per chunk (single buffer), weights through an NWS-stage ring ---
    // The taps of a chunk are unrolled (round 3, after conv_halo8: every scalar / address instruction of the loop
    // sits in front of an MFMA or a fragment read of an in-order wave): tap position, request target and the counted
    // waits are compile-time; the chunk's scalars and the weight stage stay in registers.
    constexpr int AHEAD = NWS - 1;                               // prefetch distance (taps)
    static_assert(AHEAD >= 1 && AHEAD < NT, "request distance");
    auto chunk_woff = [&](int c_) { const bool s1 = c_ >= nch0; return (unsigned)(((s1 ? a.C0 : 0) + (s1 ? c_ - nch0 : c_) * BKE) * (int)sizeof(T)); };
    auto chunk_room = [&](int c_) { const bool s1 = c_ >= nch0; return (s1 ? a.C1 : a.C0) - (s1 ? c_ - nch0 : c_) * BKE; };
    // rows beyond Cout carry a poison offset that stays out of range after the add (operands < 2 GiB - 8 KiB: checked
    // by the launcher), so a request is one add per piece; only a tail chunk masks channels
    unsigned wpo[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) wpo[g] = wlane[g] == OOB ? HALO_POISON : wlane[g];
    auto request_w = [&](unsigned soff, int room, int stage) {
        const unsigned dst = ldsW + stage * Cfg::WSTAGE + wave * (BN / 4) * 128;
        if (room >= BKE) {
#pragma unroll
            for (int g = 0; g < GW; ++g) h_dma16(rsw, wpo[g] + soff, dst + g * 8 * 128);
        } else {
#pragma unroll
            for (int g = 0; g < GW; ++g) h_dma16(rsw, wch[g] < room ? wpo[g] + soff : HALO_POISON, dst + g * 8 * 128);
        }
    };
    const unsigned w_tap_b = (unsigned)(a.w_tap_stride * (long)sizeof(T));
    unsigned woffA = chunk_woff(0); int roomA = chunk_room(0);
    issue_patch(0);
    request_w(woffA, roomA, 0);
    if (AHEAD == 2) { request_w(woffA + w_tap_b, roomA, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const bool hasnext = cc + 1 < nchunks;
        const int left = roomA;
        const int kv = left >= BKE ? 4 : (left * (int)sizeof(T) + 31) / 32;
        const unsigned woffB = hasnext ? chunk_woff(cc + 1) : 0u;
        const int roomB = hasnext ? chunk_room(cc + 1) : 0;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int wt = tap + AHEAD;                          // weights requested now (compile-time position)
            const bool more = wt < NT || hasnext;
            int stn = st + AHEAD; if (stn >= NWS) stn -= NWS;
            if (DBG && (ko & 4)) {}
            else if (wt < NT) request_w(woffA + (unsigned)wt * w_tap_b, roomA, stn);
            else if (hasnext) request_w(woffB + (unsigned)(wt - NT) * w_tap_b, roomB, stn);
            compute(tap / KW, tap % KW, st, kv);
            if (tap == NT - 1 && hasnext) {
                // every wave has finished reading the patch before it is overwritten
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (!(DBG && (ko & 16))) issue_patch(cc + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (AHEAD == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (!(DBG && (ko & 64))) __builtin_amdgcn_s_barrier();
            if (++st == NWS) st = 0;
        }
        woffA = woffB; roomA = roomB;
    }

    // --- epilogue: bias -> LDS, tile -> LDS, coalesced 16-byte row stores ------------------
    constexpr int OROW = Cfg::OROW;
    if (DBG && (ko & 32)) {                                      // no epilogue: one never-taken store keeps the accumulators live
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 1.2345e-30f) *(float*)a.out = sacc;
        return;
    }
    float* sbias = (float*)(smem + BM * OROW);
    if (tid < BN) {                                              // (requested at kernel entry: no load latency here)
        sbias[tid] = (a.bias && early_ok) ? early_bias_raw : 0.f;
        sbias[BN + tid] = (a.post_scale && early_ok) ? early_scale : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && early_ok) ? early_shift_raw : 0.f;
    }
    __syncthreads();
    {   // accumulators -> staging tile: bias, ReLU as a clamp (no branch, no canonicalisation), optional affine, convert
        const float lo = a.relu ? 0.f : -__builtin_inff();
        const int nbase = wn * 64 + 4 * (lane >> 5);
        float4 bq[TN][4];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[i][q] = *(const float4*)(sbias + nbase + i * 32 + 8 * q);
        unsigned char* drow = smem + ((wm * TM) * TW + (lane & 31)) * OROW + nbase * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[i][j][4 * q] + bq[i][q].x, acc[i][j][4 * q + 1] + bq[i][q].y,
                                  acc[i][j][4 * q + 2] + bq[i][q].z, acc[i][j][4 * q + 3] + bq[i][q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
                    if (a.post_scale) {
                        const int nl = nbase + i * 32 + 8 * q;
                        const float4 sq = *(const float4*)(sbias + BN + nl), hq = *(const float4*)(sbias + 2 * BN + nl);
                        v[0] = v[0] * sq.x + hq.x; v[1] = v[1] * sq.y + hq.y;
                        v[2] = v[2] * sq.z + hq.z; v[3] = v[3] * sq.w + hq.w;
                    }
                    unsigned char* dst = drow + j * TW * OROW + (i * 32 + 8 * q) * (int)sizeof(T);
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
    }
    __syncthreads();
    {   // staging tile -> global: 16-byte pieces, buffer stores with hardware bounds checks; per thread the channel
        // piece is fixed and the pixel advances by RPI rows of the tile per iteration (all compile-time strides)
        constexpr int CPRO = BN * (int)sizeof(T) / 16, RPI = 256 / CPRO, XPI = TW / RPI > 0 ? TW / RPI : 1;
        static_assert(RPI <= TW && TW % RPI == 0, "a pass covers a fraction of one tile row");
        const long npo = (long)a.B * H * W;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npo * a.Cout * (long)sizeof(T)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0,
                                                                              (int)(npo * a.Cout * (long)sizeof(T)), 0x00020000);
        const int c = tid % CPRO, r0 = tid / CPRO;
        const int n = n0 + c * EPC;
        const int pixB = a.Cout * (int)sizeof(T);
        const int obase = ((b * H + y0) * W + x0) * pixB;            // scalar part
        const int lane_off = n * (int)sizeof(T) + r0 * pixB;
        const unsigned char* srow = smem + r0 * OROW + c * 16;
        const bool n_ok = n < a.Cout;
        float ssum[EPC], ssq[EPC];                                                // fused BN statistics (a.stats)
        float bmu[EPC], bis[EPC];                                                 // ... of the backward pass (a.bn_x)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            ssum[e] = 0.f; ssq[e] = 0.f;
            const bool on = a.bn_x && n + e < a.Cout;
            bmu[e] = on ? a.bn_mean[n + e] : 0.f; bis[e] = on ? a.bn_invstd[n + e] : 0.f;
        }
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? a.bn_x : a.out), 0,
                                                                              (int)(npo * a.Cout * (long)sizeof(T)), 0x00020000);
        // ReLU masks of the data-gradient launches: every pass's 16 bytes requested up front (a load inside the
        // pass would be waited for on the spot, together with the previous pass's store: one round trip per pass).
        // Launches without a mask request nothing (out-of-range marker).
        constexpr int NIT = BM / RPI;
        u32x4 mkv[NIT], bxv[NIT];                                                 // bxv: the BatchNorm input at the same positions
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int yy = it / XPI, xx = (it % XPI) * RPI;
            const bool in = n_ok && (y0 + yy < H) && (x0 + xx + r0 < W);
            const unsigned o = (unsigned)(obase + lane_off + (yy * W + xx) * pixB);
            mkv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsm, (a.mask && in) ? o : OOB, 0, 0);
            bxv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (a.bn_x && in) ? o : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int yy = it / XPI, xx = (it % XPI) * RPI;                       // tile row / column offset of this pass
            const bool ok = n_ok && (y0 + yy < H) && (x0 + xx + r0 < W);
            u32x4 val = *(const u32x4*)(srow + it * RPI * OROW);
            if (a.stats && ok) {
                // second factor: the value itself (forward: sum of squares) or xhat of the BatchNorm input (backward)
                if (sizeof(T) == 2) {
                    const uint32_t wv[4] = {val.x, val.y, val.z, val.w};
                    const uint32_t xw[4] = {bxv[it].x, bxv[it].y, bxv[it].z, bxv[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(wv[e] << 16), hi = __uint_as_float(wv[e] & 0xffff0000u);
                        const float xl = __uint_as_float(xw[e] << 16), xh = __uint_as_float(xw[e] & 0xffff0000u);
                        const float fl = a.bn_x ? (xl - bmu[2 * e]) * bis[2 * e] : lo;
                        const float fh = a.bn_x ? (xh - bmu[2 * e + 1]) * bis[2 * e + 1] : hi;
                        ssum[2 * e] += lo; ssq[2 * e] += lo * fl; ssum[2 * e + 1] += hi; ssq[2 * e + 1] += hi * fh;
                    }
                } else {
                    const float fv[4] = {__uint_as_float(val.x), __uint_as_float(val.y), __uint_as_float(val.z), __uint_as_float(val.w)};
                    const float xf[4] = {__uint_as_float(bxv[it].x), __uint_as_float(bxv[it].y), __uint_as_float(bxv[it].z),
                                         __uint_as_float(bxv[it].w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float f2 = a.bn_x ? (xf[e] - bmu[e % EPC]) * bis[e % EPC] : fv[e];
                        ssum[e % EPC] += fv[e]; ssq[e % EPC] += fv[e] * f2;
                    }
                }
            }
            // (no scalar offset operand: it is added after the range check of the vector offset, which would wrap
            // the out-of-range marker of masked lanes back into the buffer)
            const unsigned off = ok ? (unsigned)(obase + lane_off + (yy * W + xx) * pixB) : OOB;
            if (a.mask) {
                const u32x4 mk = mkv[it];
                if (sizeof(T) == 2) {
                    auto keep = [](uint32_t mw, uint32_t vw) {
                        const uint32_t lo16 = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                        const uint32_t hi16 = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                        return vw & (lo16 | hi16);
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
            if (!(DBG && (ko & 1))) __builtin_amdgcn_raw_buffer_store_b128(val, rso, off, 0, 0);
        }
        if (a.pooled) {  // second output: 2x2 max pooling of the tile (TH and TW even, tile origin even), from the staging tile
            constexpr int PW2 = TW / 2, PPIX = BM / 4;
            const int Hp = H >> 1, Wp = W >> 1;
            const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(a.pooled, 0, (int)((npo >> 2) * a.Cout * (long)sizeof(T)), 0x00020000);
            for (int v = tid; v < PPIX * CPRO; v += 256) {
                const int pc = v % CPRO, pp = v / CPRO;
                const int py = pp / PW2, px = pp % PW2;
                const unsigned char* s0 = smem + ((2 * py) * TW + 2 * px) * OROW + pc * 16;
                const u32x4 q0 = *(const u32x4*)s0, q1 = *(const u32x4*)(s0 + OROW), q2 = *(const u32x4*)(s0 + TW * OROW),
                            q3 = *(const u32x4*)(s0 + TW * OROW + OROW);
                u32x4 m;
                m.x = piece_max<T>(piece_max<T>(q0.x, q1.x), piece_max<T>(q2.x, q3.x));
                m.y = piece_max<T>(piece_max<T>(q0.y, q1.y), piece_max<T>(q2.y, q3.y));
                m.z = piece_max<T>(piece_max<T>(q0.z, q1.z), piece_max<T>(q2.z, q3.z));
                m.w = piece_max<T>(piece_max<T>(q0.w, q1.w), piece_max<T>(q2.w, q3.w));
                const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px, nn = n0 + pc * EPC;
                const bool okp = gy < Hp && gx < Wp && nn < a.Cout;
                const unsigned offp = okp ? (unsigned)((((b * Hp + gy) * Wp + gx) * a.Cout + nn) * (int)sizeof(T)) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(m, rsp, offp, 0, 0);
            }
        }
        if (a.stats) {   // block reduction over the RPI row-lanes of each column (fixed order), one partial row per pixel tile
            __syncthreads();                                                      // the staging tile has been read
            float* red = (float*)smem;                                            // [RPI][BN][2]
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                red[(r0 * BN + c * EPC + e) * 2] = ssum[e];
                red[(r0 * BN + c * EPC + e) * 2 + 1] = ssq[e];
            }
            __syncthreads();
            const int ptile = logical / tiles_n;                               // pixel-tile index (n-tile fastest)
            for (int v = tid; v < BN * 2; v += 256) {
                const int col = v >> 1, st2 = v & 1;
                double acc = 0.0;
                for (int rl = 0; rl < RPI; ++rl) acc += (double)red[(rl * BN + col) * 2 + st2];
                // layout [2][Cout][pixel tiles]: the finalize reads each column's partials as one contiguous run
                if (n0 + col < a.Cout) stats_emit(a, st2, n0 + col, gridDim.x / tiles_n, ptile, (float)acc);
            }
        }
    }
}


// ------------------------------------------------------------------------- //
// conv_halo8_kernel (bf16, round 3): the same tile as conv_halo_kernel<BN, 8> but ONE 8-wave workgroup per CU with the
// halo patch DOUBLE-BUFFERED. At configs[1] sizes every layer is one wave of workgroups that all start together: the
// single-buffer kernel above runs [patch burst][9 taps][patch burst][9 taps]...[store burst] with the whole chip
// in the same phase, so the patch bytes (the entire input tensor, once per 64-channel chunk) are read while no MFMA
// runs. Here chunk cc+1's patch is requested at the first tap of chunk cc (behind that step's weight request, so the
// counted vmcnt waits leave it in flight for three steps) and lands in the other buffer under the nine taps of chunk
// cc; the eight waves share one weight ring (half the weight DMA bytes per pixel of two 4-wave workgroups), and the
// fragment reads of k-step s+1 are issued under the MFMAs of k-step s (the waves of one workgroup run in lockstep
// behind the per-tap barrier, so nothing else hides LDS latency).
// ------------------------------------------------------------------------- //
template <int BN, int TH, int MODE, int NWS_ = 3>
struct Halo8Cfg {
    static constexpr int NT = MODE == UPCONV2 ? 4 : 9, KW = MODE == UPCONV2 ? 2 : 3;
    static constexpr int TW = 32, PW = MODE == UPCONV2 ? TW / 2 + 2 : TW + 2, PH = MODE == UPCONV2 ? TH / 2 + 1 : TH + 2;
    static constexpr int PROWS = (PH * PW + 7) / 8 * 8;
    static constexpr int PATCH = PROWS * 128;
    static constexpr int WSTAGE = BN * 128, NWS = NWS_;
    static constexpr int BM = TH * TW;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int EPI = BM * OROW + 3 * BN * 4;
    static constexpr int MAIN = 2 * PATCH + NWS * WSTAGE;
    static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
};

template <int BN, int TH, int MODE, int NWS_, int SCHED>
__global__ __launch_bounds__(512, 2) void conv_halo8_kernel(ConvArgs a) {
    typedef bf16_t T;
    using Cfg = Halo8Cfg<BN, TH, MODE, NWS_>;
    constexpr int NT = Cfg::NT, KW = Cfg::KW, NWS = NWS_, AHEAD = NWS_ - 1;   // weight stages, request distance (taps)
    constexpr int EPC = 8, BKE = 64, NW = 8, NTHR = 512;
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int WAVES_N = BN / 64, WAVES_M = NW / WAVES_N;
    constexpr int TN = 2, TM = TH / WAVES_M;
    static_assert(TM >= 1 && TM * WAVES_M == TH, "tile split");
    constexpr int NPP = PROWS / 8;                               // patch DMA pieces
    constexpr int NPW = (NPP + NW - 1) / NW;                     // ... per wave (every wave issues exactly NPW: see below)
    static_assert(NPW >= 2 || NPP % NW == 0, "duplicate-piece padding needs a previous piece");
    constexpr int GW = BN / (8 * NW);                            // weight DMA pieces per wave and stage
    static_assert(GW >= 1, "weights");
    constexpr int BM = Cfg::BM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SCHED: 0 = lockstep halves, 1 = halves one phase apart, 2 = 1 + the dev aids (s_memtime stamps, MPU_STAMPS=1; the
    // s_setprio placement switchable at run time, MPU_HALO8_PRIO): their ~30 scalar instructions per tap stay out of 1
    constexpr bool DBG = SCHED != 1;
    unsigned long long* stamps = (DBG && a.dbg_buf && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 32 && tid == 0)
                                     ? a.dbg_buf + (blockIdx.x >> 3) * 16 : nullptr;
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int logical = a.xcd ? xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;      // (kernels.h)
    int t = logical;
    const int n0 = (t % tiles_n) * BN; t /= tiles_n;
    const int x0 = (t % tiles_x) * TW; t /= tiles_x;
    const int y0 = (t % tiles_y) * TH; const int b = t / tiles_y;
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    const int nsteps = nchunks * NT;
    constexpr unsigned OOB = 0xfffffff0u;
    const int Hi = MODE == UPCONV2 ? H / 2 : H, Wi = MODE == UPCONV2 ? W / 2 : W;
    const long npix = (long)a.B * Hi * Wi;
    const i32x4 rs0 = h_make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rs1 = h_make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * 2L : 0);
    const i32x4 rsw = h_make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + 2 * Cfg::PATCH;

    // Epilogue constants of this thread's output channel, requested NOW so that their latency hides under the main
    // loop (round 3: the epilogue used to start with an exposed global load). Unconditional loads from a pointer that is
    // always valid (the weights stand in for a null table; the value is discarded by a select in the epilogue).
    const bool early_ok = tid < BN && n0 + tid < a.Cout;
    const int eidx = early_ok ? n0 + tid : 0;
    const float early_bias_raw = (a.bias ? a.bias : (const float*)a.w)[eidx];
    const float early_scale = (a.post_scale ? a.post_scale : (const float*)a.w)[eidx];
    const float early_shift_raw = (a.post_scale ? a.post_shift : (const float*)a.w)[eidx];

    // --- per-lane DMA roles -------------------------------------------------------------
    const int lrow = lane >> 3, slot = lane & 7;
    int ppix[NPW], pchunk[NPW], ppiece[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        int piece = wave + NW * k;
        if (piece >= NPP) piece -= NW;                           // padding: the wave's previous piece again (same bytes, same place)
        ppiece[k] = piece;
        const int pr = piece * 8 + lrow;
        const int py = pr / PW, px = pr % PW;
        const int iy = MODE == UPCONV2 ? y0 / 2 + py : y0 + py - 1, ix = MODE == UPCONV2 ? x0 / 2 + px : x0 + px - 1;
        const bool v = pr < Cfg::PH * PW && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        ppix[k] = v ? (b * Hi + iy) * Wi + ix : (int)npix;      // padding: the first pixel beyond the tensor
        pchunk[k] = slot ^ ((pr >> 1) & 7);
    }
    auto issue_patch = [&](int cc, int buf) {
        const bool s1 = cc >= nch0;
        const int cbase = (s1 ? cc - nch0 : cc) * BKE, Cs = s1 ? a.C1 : a.C0;
        i32x4 qrs;                                               // source descriptor by scalar selects (no branch per piece)
        qrs.x = s1 ? rs1.x : rs0.x; qrs.y = s1 ? rs1.y : rs0.y; qrs.z = s1 ? rs1.z : rs0.z; qrs.w = rs0.w;
        const bool tail = Cs - cbase < BKE;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int ch = cbase + pchunk[k] * EPC;
            unsigned off = (unsigned)((ppix[k] * Cs + ch) * 2);
            if (tail) off = ch < Cs ? off : OOB;
            const unsigned dst = lds0 + buf * Cfg::PATCH + __builtin_amdgcn_readfirstlane(ppiece[k]) * 1024;
            h_dma16(qrs, off, dst);
        }
    };
    unsigned wlane[GW]; int wch[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int rl = wave * (BN / NW) + g * 8 + lrow;
        const int n = n0 + rl;
        wch[g] = (slot ^ ((rl >> 1) & 7)) * EPC;
        wlane[g] = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * 2L) + (unsigned)(wch[g] * 2) : OOB;
    }
    int w_tap = 0, w_cc = 0;
    const unsigned w_tap_bytes = (unsigned)(a.w_tap_stride * 2L);   // (the launcher checks w_elems * 2 < 2 GiB)
    unsigned w_tapoff = 0;                                       // w_tap * w_tap_bytes, kept running
    auto issue_w = [&](int stage) {
        const bool s1 = w_cc >= nch0;
        const int cbase = (s1 ? w_cc - nch0 : w_cc) * BKE, Cs = s1 ? a.C1 : a.C0;
        const unsigned soff = w_tapoff + (unsigned)(((s1 ? a.C0 : 0) + cbase) * 2);
        const unsigned dst = ldsW + stage * Cfg::WSTAGE + wave * (BN / NW) * 128;
        const int room = Cs - cbase;
#pragma unroll
        for (int g = 0; g < GW; ++g) {
            const unsigned off = (wch[g] < room && wlane[g] != OOB) ? wlane[g] + soff : OOB;
            h_dma16(rsw, off, dst + g * 8 * 128);
        }
        w_tapoff += w_tap_bytes;
        if (++w_tap == NT) { w_tap = 0; w_tapoff = 0; ++w_cc; }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    // Fragment addressing of one tap: weight rows of the stage, patch rows shifted by the tap.
    struct TapAddr { const unsigned char* Wb; const unsigned char* Pr[TM]; int psw[TM]; };
    auto tap_addr_yx = [&](int ky, int kx, int stage, int buf, TapAddr& A) {
        A.Wb = smem + 2 * Cfg::PATCH + stage * Cfg::WSTAGE + (wn * 64 + (lane & 31)) * 128;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow = MODE == UPCONV2 ? ((wm * TM + j + ky) >> 1) * PW + ((kx + (lane & 31)) >> 1)
                                             : (wm * TM + j + ky) * PW + kx + (lane & 31);
            A.Pr[j] = smem + buf * Cfg::PATCH + prow * 128;
            A.psw[j] = (prow >> 1) & 7;
        }
    };
    auto tap_addr = [&](int tap, int stage, int buf, TapAddr& A) {
        const int ky = tap / KW;
        tap_addr_yx(ky, tap - ky * KW, stage, buf, A);
    };
    uint4 af[2][TN], bf[2][TM];
    auto load = [&](const TapAddr& A, int s_, int set) {
        const int q = 2 * s_ + fh;
#pragma unroll
        for (int i = 0; i < TN; ++i) af[set][i] = *(const uint4*)(A.Wb + i * 32 * 128 + ((q ^ fsw) << 4));
#pragma unroll
        for (int j = 0; j < TM; ++j) bf[set][j] = *(const uint4*)(A.Pr[j] + ((q ^ A.psw[j]) << 4));
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) HMma<T>::run(af[set][i], bf[set][j], acc[i][j]);
    };

    // --- pipeline ---------------------------------------------------------------------------
    // A tap = four k-steps of 16 channels on fragment sets 0,1,0,1; the reads of k-step s+1 are issued before the
    // MFMAs of k-step s. The per-tap barrier sits between k-steps 2 and 3: by then every read of this tap's weight
    // stage has returned (k-step 3's fragments are in registers), and behind it the NEXT tap's first fragments are
    // requested under k-step 3's MFMAs -- no tap starts with an exposed LDS round trip.
    // Request order inside a tap: weights of tap+2 first, then (first tap of a chunk) the next chunk's patch. The
    // queue is in order, so "at most GW (+ NPW) requests outstanding" means the weights of tap+1 have landed while
    // the newest weight stage -- and a patch requested in this or the previous tap -- stay in flight; the patch is
    // forced by the wait of the third tap, six taps before its first reader.
    if constexpr (SCHED == 0) {
        issue_patch(0, 0);
        issue_w(0);
        if (nsteps >= AHEAD) {                                       // (layers have >= 4 taps: always)
    #pragma unroll
            for (int k = 1; k < AHEAD; ++k) issue_w(k);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * GW) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);                // the younger half loses every arbitration otherwise
        __builtin_amdgcn_s_barrier();
        if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();
        int st = 0, tap = 0, cc = 0;
        TapAddr cur;
        tap_addr(0, 0, 0, cur);
        load(cur, 0, 0);
        for (int step = 0; step < nsteps; ++step) {
            const int stn = (st + AHEAD) % NWS;
            const bool more = step + AHEAD < nsteps;
            if (more) issue_w(stn);
            const bool pre = tap == 0 && cc + 1 < nchunks;
            if (pre) issue_patch(cc + 1, (cc + 1) & 1);
            const bool patch_young = tap <= AHEAD - 1 && cc + 1 < nchunks;  // a patch requested within the last AHEAD taps
            int ntap = tap + 1, ncc = cc;
            if (ntap == NT) { ntap = 0; ++ncc; }
            const bool last = step + 1 >= nsteps;
            TapAddr nxt;
            tap_addr(last ? tap : ntap, last ? st : (st + 1) % NWS, (last ? cc : ncc) & 1, nxt);
            load(cur, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            load(cur, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
            load(cur, 3, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(0);
            __builtin_amdgcn_sched_barrier(0);
            if (stamps && (step == 4 || step == 5)) stamps[8 + 4 * (step - 4)] = __builtin_amdgcn_s_memtime();
            if (more) {
                if (patch_young) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * GW + NPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * GW) : "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (stamps && (step == 4 || step == 5)) stamps[9 + 4 * (step - 4)] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stamps && (step == 4 || step == 5)) stamps[10 + 4 * (step - 4)] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (stamps && (step == 4 || step == 5)) stamps[11 + 4 * (step - 4)] = __builtin_amdgcn_s_memtime();
            load(nxt, 0, 0);                                         // (after the last tap: a harmless re-read)
            __builtin_amdgcn_sched_barrier(0);
            mma(1);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            st = (st + 1) % NWS;
            tap = ntap; cc = ncc;
        }
    } else {
        // SCHED 1 (round 3): the two halves of the workgroup run ONE PHASE APART. A tap is a load phase L (every
        // fragment of the tap's four k-steps read into registers, the tap's address arithmetic, counted waits) and a
        // compute phase C (the tap's MFMAs with the DMA requests in their shadow: weights of tap+3, a share of the
        // next chunk's patch); a workgroup barrier closes each phase, and waves 4-7 start one barrier late: while one
        // half's MFMAs own the matrix pipes, the other half's LDS reads run beside them (waves w and w+4 share a SIMD)
        // instead of in front of its own MFMAs.
        //   interval:   I0    I1    I2    I3    I4 ...
        //   waves 0-3:  L(0)  C(0)  L(1)  C(1)  L(2)
        //   waves 4-7:   -    L(0)  C(0)  L(1)  C(1)
        // Hazards (B(k) = the barrier that ends interval k): the weights of tap u (stage u % 4) are read in I(2u) and
        // I(2u+1); tap u+4 is requested into that stage in C(u+1) = I(2u+3) / I(2u+4), behind B(2u+1) and each reader's
        // lgkmcnt(0). Every wave ends L(v) with a counted vmcnt that forces its pieces of tap v+1 (requested in
        // C(v-2); the queue is in order: w(v+1) p(v-2) w(v+2) p(v-1), p = patch pieces of that tap if any), so they
        // have landed when it arrives at B(2v) (first half) / B(2v+1) (second half), and the first reader of tap v+1
        // starts behind B(2v+1). The next chunk's patch goes to the other buffer, PPT pieces per wave and tap in the
        // first taps of a chunk; that buffer's last reader finished before B(2u-1), u the chunk's first tap, and the
        // pieces are forced three taps after their request, at least three taps before their first reader.
        static_assert(NWS == 4, "four weight stages: requests three taps ahead");
        constexpr int PPT = 2;                                   // patch pieces per wave and tap
        static_assert(NPW % PPT == 0 && NPW / PPT <= NT - 3, "patch pieces of tap k are forced at the end of L(k+3): before the chunk ends");
        constexpr int PTAPS = NPW / PPT;
        constexpr bool WREQ_IN_L = TM >= 2;                      // where the weight requests sit: see the load phase
        const bool second = wave >= 4;
        const bool prio_c = DBG ? (a.dbg & 1) != 0 : false, prio_l = DBG ? (a.dbg & 2) != 0 : true;   // (DBG: MPU_HALO8_PRIO)
        uint4 fa[4][TN], fb[4][TM];
        // The nine (four) taps of a chunk are unrolled: tap position, patch-piece schedule, counted waits and the tap's
        // patch-row shift are compile-time, only the weight stage (period 4) and the chunk's scalars stay in registers --
        // the wave issues in order, and every scalar / address instruction of the loop sits in front of an MFMA or a
        // fragment read (the lean instantiation runs the 64-channel tiles 8 % faster than the one with ~30 more scalar
        // instructions per tap).
        // per chunk: byte offset of its channels inside a weight row, channels left in its source, k-steps that hold data
        auto chunk_woff = [&](int c_) { const bool s1 = c_ >= nch0; return (unsigned)(((s1 ? a.C0 : 0) + (s1 ? c_ - nch0 : c_) * BKE) * 2); };
        auto chunk_room = [&](int c_) { const bool s1 = c_ >= nch0; return (s1 ? a.C1 : a.C0) - (s1 ? c_ - nch0 : c_) * BKE; };
        auto ksteps_of = [&](int room) { return room >= BKE ? 4 : (room + 15) / 16; };
        unsigned wpo[GW];                                         // (poison offset: see conv_halo_kernel)
#pragma unroll
        for (int g = 0; g < GW; ++g) wpo[g] = wlane[g] == OOB ? HALO_POISON : wlane[g];
        auto request_w = [&](unsigned soff, int room, int stage) {         // weights of one tap: GW pieces per wave
            const unsigned dst = ldsW + stage * Cfg::WSTAGE + wave * (BN / NW) * 128;
            if (room >= BKE) {
#pragma unroll
                for (int g = 0; g < GW; ++g) h_dma16(rsw, wpo[g] + soff, dst + g * 8 * 128);
            } else {
#pragma unroll
                for (int g = 0; g < GW; ++g) h_dma16(rsw, wch[g] < room ? wpo[g] + soff : HALO_POISON, dst + g * 8 * 128);
            }
        };
        const unsigned w_tap_b = (unsigned)(a.w_tap_stride * 2L);
        unsigned woffA = chunk_woff(0); int roomA = chunk_room(0);
        issue_patch(0, 0);
        request_w(woffA, roomA, 0);
        request_w(woffA + w_tap_b, roomA, 1);
        request_w(woffA + 2 * w_tap_b, roomA, 2);                 // (NT >= 4: the first three taps are of chunk 0)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GW) : "memory");
        __builtin_amdgcn_s_barrier();
        if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();
        if (second) __builtin_amdgcn_s_barrier();                // one phase behind
        int st = 0;
        for (int cc = 0; cc < nchunks; ++cc) {
            const bool hasnext = cc + 1 < nchunks;
            const int kv = ksteps_of(roomA);
            const unsigned woffB = hasnext ? chunk_woff(cc + 1) : 0u;
            const int roomB = hasnext ? chunk_room(cc + 1) : 0;
            const unsigned pbuf = (unsigned)(cc & 1) * Cfg::PATCH, pnext = (unsigned)((cc + 1) & 1) * Cfg::PATCH;
            // next chunk's patch source (requested in the first PTAPS compute phases)
            const int ncc = cc + 1;
            const bool ns1 = ncc >= nch0;
            const int ncbase = (ns1 ? ncc - nch0 : ncc) * BKE, nCs = ns1 ? a.C1 : a.C0;
            i32x4 qrs;
            qrs.x = ns1 ? rs1.x : rs0.x; qrs.y = ns1 ? rs1.y : rs0.y; qrs.z = ns1 ? rs1.z : rs0.z; qrs.w = rs0.w;
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                constexpr int dummy = 0; (void)dummy;
                const int ky = tap / KW, kx = tap % KW;           // (compile-time)
                // patch pieces in flight behind w(tap+1) at the end of this load phase: requested in C(tap-1), C(tap-2)
                const int cnt = ((tap >= 1 && tap - 1 < PTAPS) ? 1 : 0) + ((tap >= 2 && tap - 2 < PTAPS) ? 1 : 0);
                const bool stamp_here = DBG && stamps && cc == 0 && (tap == 4 % NT || tap == 5 % NT) && NT > 5;
                const int sidx = tap == 4 ? 0 : 1;
                // ---- L: fragments of the tap
                if (stamp_here) stamps[8 + 4 * sidx] = __builtin_amdgcn_s_memtime();
                if (prio_l) __builtin_amdgcn_s_setprio(1);
                TapAddr A;
                tap_addr_yx(ky, kx, st, 0, A);
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    if (s_ == 2 && kv <= 2) break;
                    const int q = 2 * s_ + fh;
#pragma unroll
                    for (int i = 0; i < TN; ++i) fa[s_][i] = *(const uint4*)(A.Wb + i * 32 * 128 + ((q ^ fsw) << 4));
#pragma unroll
                    for (int j = 0; j < TM; ++j) fb[s_][j] = *(const uint4*)(A.Pr[j] + pbuf + ((q ^ A.psw[j]) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WREQ_IN_L) {
                    // weights three taps ahead, requested HERE (128-channel tiles: the load phase otherwise sits idle in
                    // its lgkmcnt wait while the compute phase carries the DMA issue): the queue behind w(tap+1) is then
                    // p(tap-2) w(tap+2) p(tap-1) w(tap+3)
                    const int wt = tap + 3;
                    const bool req = wt < NT || hasnext;
                    if (wt < NT) request_w(woffA + (unsigned)wt * w_tap_b, roomA, (st + 3) & 3);
                    else if (hasnext) request_w(woffB + (unsigned)(wt - NT) * w_tap_b, roomB, (st + 3) & 3);
                    __builtin_amdgcn_sched_barrier(0);
                    const bool w2 = tap + 2 < NT || hasnext;      // w(tap+2) was requested (in L(tap-1))
                    if (!w2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (!req) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");             // (last chunk: no patch pieces)
                    else if (cnt == 2 && hasnext) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GW + 2 * PPT) : "memory");
                    else if (cnt == 1 && hasnext) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GW + PPT) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GW) : "memory");
                } else {
                if (tap >= NT - 2 && !hasnext) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing requested behind w(tap+1)
                else if (cnt == 2 && hasnext) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW + 2 * PPT) : "memory");
                else if (cnt == 1 && hasnext) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW + PPT) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");
                }
                if (stamp_here) stamps[9 + 4 * sidx] = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (stamp_here) stamps[10 + 4 * sidx] = __builtin_amdgcn_s_memtime();
                if (prio_l) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                // ---- C: the tap's MFMAs; DMA requests between them
                if (stamp_here) stamps[11 + 4 * sidx] = __builtin_amdgcn_s_memtime();
                if (prio_c) __builtin_amdgcn_s_setprio(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) HMma<T>::run(fa[0][i], fb[0][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!WREQ_IN_L) {   // weights three taps ahead: of this chunk, or the first taps of the next one
                    const int wt = tap + 3;
                    if (wt < NT) request_w(woffA + (unsigned)wt * w_tap_b, roomA, (st + 3) & 3);
                    else if (hasnext) request_w(woffB + (unsigned)(wt - NT) * w_tap_b, roomB, (st + 3) & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) HMma<T>::run(fa[1][i], fb[1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                if (tap < PTAPS && hasnext) {
#pragma unroll
                    for (int k = PPT * tap; k < PPT * tap + PPT; ++k) {
                        if (k < NPW) {
                            const int ch = ncbase + pchunk[k] * EPC;
                            unsigned off = (unsigned)((ppix[k] * nCs + ch) * 2);
                            if (nCs - ncbase < BKE) off = ch < nCs ? off : OOB;
                            h_dma16(qrs, off, lds0 + pnext + __builtin_amdgcn_readfirstlane(ppiece[k]) * 1024);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kv > 2) {
#pragma unroll
                    for (int s_ = 2; s_ < 4; ++s_)
#pragma unroll
                        for (int i = 0; i < TN; ++i)
#pragma unroll
                            for (int j = 0; j < TM; ++j) HMma<T>::run(fa[s_][i], fb[s_][j], acc[i][j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (prio_c) __builtin_amdgcn_s_setprio(0);
                if (stamp_here && tap == 4) stamps[7] = __builtin_amdgcn_s_memtime();
                __builtin_amdgcn_s_barrier();
                st = (st + 1) & 3;
            }
            woffA = woffB; roomA = roomB;
        }
        if (!second) __builtin_amdgcn_s_barrier();               // the second half's last compute phase
    }
    __builtin_amdgcn_s_setprio(0);
    if (stamps) stamps[2] = __builtin_amdgcn_s_memtime();

    // --- epilogue: as conv_halo_kernel, for 512 threads --------------------------------------
    constexpr int OROW = Cfg::OROW;
    float* sbias = (float*)(smem + BM * OROW);
    if (tid < BN) {                                              // (requested at kernel entry: no load latency here)
        sbias[tid] = (a.bias && early_ok) ? early_bias_raw : 0.f;
        sbias[BN + tid] = (a.post_scale && early_ok) ? early_scale : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && early_ok) ? early_shift_raw : 0.f;
    }
    __syncthreads();
    {
        const float lo = a.relu ? 0.f : -__builtin_inff();
        const int nbase = wn * 64 + 4 * (lane >> 5);
        float4 bq[TN][4];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[i][q] = *(const float4*)(sbias + nbase + i * 32 + 8 * q);
        unsigned char* drow = smem + ((wm * TM) * TW + (lane & 31)) * OROW + nbase * 2;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[i][j][4 * q] + bq[i][q].x, acc[i][j][4 * q + 1] + bq[i][q].y,
                                  acc[i][j][4 * q + 2] + bq[i][q].z, acc[i][j][4 * q + 3] + bq[i][q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
                    if (a.post_scale) {
                        const int nl = nbase + i * 32 + 8 * q;
                        const float4 sq = *(const float4*)(sbias + BN + nl), hq = *(const float4*)(sbias + 2 * BN + nl);
                        v[0] = v[0] * sq.x + hq.x; v[1] = v[1] * sq.y + hq.y;
                        v[2] = v[2] * sq.z + hq.z; v[3] = v[3] * sq.w + hq.w;
                    }
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(v[0], v[1]);
                    pk.y = f32x2_to_bf16x2(v[2], v[3]);
                    *(uint2*)(drow + j * TW * OROW + (i * 32 + 8 * q) * 2) = pk;
                }
            }
        }
    }
    __syncthreads();
    if (stamps) stamps[3] = __builtin_amdgcn_s_memtime();
    {
        // thread = (16-byte channel piece c, pixel lane r0); a pass covers RPI consecutive pixels of the tile
        constexpr int CPRO = BN * 2 / 16, RPI = NTHR / CPRO, NIT = BM / RPI;
        static_assert(BM % RPI == 0 && (RPI % TW == 0 || TW % RPI == 0), "pass shape");
        const long npo = (long)a.B * H * W;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npo * a.Cout * 2L), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0,
                                                                              (int)(npo * a.Cout * 2L), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? a.bn_x : a.out), 0,
                                                                              (int)(npo * a.Cout * 2L), 0x00020000);
        const int c = tid % CPRO, r0 = tid / CPRO;
        const int r0y = r0 / TW, r0x = r0 % TW;
        const int n = n0 + c * EPC;
        const int pixB = a.Cout * 2;
        const int obase = ((b * H + y0) * W + x0) * pixB;
        const int lane_off = n * 2 + (r0y * W + r0x) * pixB;
        const unsigned char* srow = smem + r0 * OROW + c * 16;
        const bool n_ok = n < a.Cout;
        float ssum[EPC], ssq[EPC], bmu[EPC], bis[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            ssum[e] = 0.f; ssq[e] = 0.f;
            const bool on = a.bn_x && n + e < a.Cout;
            bmu[e] = on ? a.bn_mean[n + e] : 0.f; bis[e] = on ? a.bn_invstd[n + e] : 0.f;
        }
        auto pass_yx = [&](int it, int& yy, int& xx) {          // tile-local origin of pass `it` (compile-time per pass)
            if (RPI >= TW) { yy = it * (RPI / TW); xx = 0; }
            else { constexpr int XPI = TW / (RPI < TW ? RPI : TW); yy = it / XPI; xx = (it % XPI) * RPI; }
        };
        u32x4 mkv[NIT], bxv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int yy, xx; pass_yx(it, yy, xx);
            const bool in = n_ok && (y0 + yy + r0y < H) && (x0 + xx + r0x < W);
            const unsigned o = (unsigned)(obase + lane_off + (yy * W + xx) * pixB);
            mkv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsm, (a.mask && in) ? o : OOB, 0, 0);
            bxv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (a.bn_x && in) ? o : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int yy, xx; pass_yx(it, yy, xx);
            const bool ok = n_ok && (y0 + yy + r0y < H) && (x0 + xx + r0x < W);
            u32x4 val = *(const u32x4*)(srow + it * RPI * OROW);
            if (a.stats && ok) {
                const uint32_t wv[4] = {val.x, val.y, val.z, val.w};
                const uint32_t xw[4] = {bxv[it].x, bxv[it].y, bxv[it].z, bxv[it].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __uint_as_float(wv[e] << 16), hi = __uint_as_float(wv[e] & 0xffff0000u);
                    const float xl = __uint_as_float(xw[e] << 16), xh = __uint_as_float(xw[e] & 0xffff0000u);
                    const float fl = a.bn_x ? (xl - bmu[2 * e]) * bis[2 * e] : lo;
                    const float fh2 = a.bn_x ? (xh - bmu[2 * e + 1]) * bis[2 * e + 1] : hi;
                    ssum[2 * e] += lo; ssq[2 * e] += lo * fl; ssum[2 * e + 1] += hi; ssq[2 * e + 1] += hi * fh2;
                }
            }
            const unsigned off = ok ? (unsigned)(obase + lane_off + (yy * W + xx) * pixB) : OOB;
            if (a.mask) {
                const u32x4 mk = mkv[it];
                auto keep = [](uint32_t mw, uint32_t vw) {
                    const uint32_t lo16 = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                    const uint32_t hi16 = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                    return vw & (lo16 | hi16);
                };
                val.x = keep(mk.x, val.x); val.y = keep(mk.y, val.y);
                val.z = keep(mk.z, val.z); val.w = keep(mk.w, val.w);
            }
            __builtin_amdgcn_raw_buffer_store_b128(val, rso, off, 0, 0);
        }
        if (a.pooled) {
            constexpr int PW2 = TW / 2, PPIX = BM / 4;
            const int Hp = H >> 1, Wp = W >> 1;
            const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(a.pooled, 0, (int)((npo >> 2) * a.Cout * 2L), 0x00020000);
            for (int v = tid; v < PPIX * CPRO; v += NTHR) {
                const int pc = v % CPRO, pp = v / CPRO;
                const int py = pp / PW2, px = pp % PW2;
                const unsigned char* s0 = smem + ((2 * py) * TW + 2 * px) * OROW + pc * 16;
                const u32x4 q0 = *(const u32x4*)s0, q1 = *(const u32x4*)(s0 + OROW), q2 = *(const u32x4*)(s0 + TW * OROW),
                            q3 = *(const u32x4*)(s0 + TW * OROW + OROW);
                u32x4 m;
                m.x = piece_max<T>(piece_max<T>(q0.x, q1.x), piece_max<T>(q2.x, q3.x));
                m.y = piece_max<T>(piece_max<T>(q0.y, q1.y), piece_max<T>(q2.y, q3.y));
                m.z = piece_max<T>(piece_max<T>(q0.z, q1.z), piece_max<T>(q2.z, q3.z));
                m.w = piece_max<T>(piece_max<T>(q0.w, q1.w), piece_max<T>(q2.w, q3.w));
                const int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px, nn = n0 + pc * EPC;
                const bool okp = gy < Hp && gx < Wp && nn < a.Cout;
                const unsigned offp = okp ? (unsigned)((((b * Hp + gy) * Wp + gx) * a.Cout + nn) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(m, rsp, offp, 0, 0);
            }
        }
        if (a.stats) {
            __syncthreads();
            float* red = (float*)smem;                                            // [RPI][BN][2]
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                red[(r0 * BN + c * EPC + e) * 2] = ssum[e];
                red[(r0 * BN + c * EPC + e) * 2 + 1] = ssq[e];
            }
            __syncthreads();
            const int ptile = logical / tiles_n;
            for (int v = tid; v < BN * 2; v += NTHR) {
                const int col = v >> 1, st2 = v & 1;
                double acc2 = 0.0;
                for (int rl = 0; rl < RPI; ++rl) acc2 += (double)red[(rl * BN + col) * 2 + st2];
                if (n0 + col < a.Cout) stats_emit(a, st2, n0 + col, gridDim.x / tiles_n, ptile, (float)acc2);
            }
        }
    }
    if (stamps) {
        stamps[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[5] = __builtin_amdgcn_s_memtime();
        stamps[6] = (unsigned long long)nsteps;
    }
}

template <int BN, int TH, int MODE, int NWS_, int SCHED>
int launch_halo8_cfg_n(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = Halo8Cfg<BN, TH, MODE, NWS_>;
    static_assert(Cfg::SMEM <= 160 * 1024, "LDS");
    auto kern = conv_halo8_kernel<BN, TH, MODE, NWS_, SCHED>;
    ConvArgs a = a_in;
    if (a.w_elems <= 0) a.w_elems = (Cfg::NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    const long Min = MODE == UPCONV2 ? M / 4 : M;
    if (Min * cmax * 2L >= (1L << 31) - 8192 || a.w_elems * 2L >= (1L << 31) - 8192 || M * a.Cout * 2L >= (1L << 31) - 8192)
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    const long tiles = (long)a.B * cdiv(a.Ho, TH) * cdiv(a.Wo, Cfg::TW) * cdiv(a.Cout, BN);
    const long ptiles = tiles / cdiv(a.Cout, BN);
    if (a.stats && a.stats_rows) {
        if (ptiles * 2 * a.Cout <= a.stats_cap) *a.stats_rows = (int)ptiles;
        else { a.stats = nullptr; *a.stats_rows = 0; }
    } else a.stats = nullptr;
    if (a.pooled && a.pooled_done && MODE == CONV3 && !a.mask && !(a.Ho & 1) && !(a.Wo & 1) && TH % 2 == 0) *a.pooled_done = 1;
    else a.pooled = nullptr;
    a.dbg_buf = stamp_buffer();
    a.dbg = 2;                                                   // (SCHED 2: s_setprio 1 in the load phase, as SCHED 1 has it compiled in)
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * Cfg::NT * (a.C0 + a.C1), st);
    a.xcd = env(ENV_XCD_TILES) != 0;
    launch_k(kern, dim3((unsigned)tiles), dim3(512), Cfg::SMEM, st, a);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

template <typename T, int BN, int TH, int NWS, int MODE = CONV3, int DBG = 0>
int launch_halo_cfg(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = HaloCfg<T, BN, TH, NWS, MODE>;
    auto kern = conv_halo_kernel<T, BN, TH, NWS, MODE, DBG>;
    ConvArgs a = a_in;
    if (a.w_elems <= 0) a.w_elems = (Cfg::NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    const long Min = MODE == UPCONV2 ? M / 4 : M;   // input pixels (32-bit DMA / buffer-store offsets)
    if (Min * cmax * (long)sizeof(T) >= (1L << 31) - 8192 || a.w_elems * (long)sizeof(T) >= (1L << 31) - 8192 ||
        M * a.Cout * (long)sizeof(T) >= (1L << 31) - 8192)
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    const long tiles = (long)a.B * cdiv(a.Ho, TH) * cdiv(a.Wo, Cfg::TW) * cdiv(a.Cout, BN);
    const long ptiles = tiles / cdiv(a.Cout, BN);
    if (a.stats && a.stats_rows) {
        if (ptiles * 2 * a.Cout <= a.stats_cap) *a.stats_rows = (int)ptiles;
        else { a.stats = nullptr; *a.stats_rows = 0; }
    } else a.stats = nullptr;
    if (a.pooled && a.pooled_done && MODE == CONV3 && !a.mask && !(a.Ho & 1) && !(a.Wo & 1) && TH % 2 == 0) *a.pooled_done = 1;
    else a.pooled = nullptr;
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * Cfg::NT * (a.C0 + a.C1), st);
    a.xcd = env(ENV_XCD_TILES) != 0;
    launch_k(kern, dim3((unsigned)tiles), dim3(256), Cfg::SMEM, st, a);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// (three weight stages = requests two taps ahead; a fourth stage measured the same: the in-loop stamps show < 100
// cycles in the counted vmcnt wait, the tap is bound by MFMA issue + fragment reads)
template <int BN, int TH, int MODE>
int launch_halo8_cfg(const ConvArgs& a, hipStream_t st) {
    const long sched = env(ENV_HALO8_SCHED);                     // 0 = lockstep halves, 1 = one phase apart (default)
    if (sched != 1) return launch_halo8_cfg_n<BN, TH, MODE, 3, 0>(a, st);
    static int dev = -1;                                         // stamps asked for (MPU_STAMPS=1): the instrumented build
    if (dev < 0) dev = stamp_buffer() != nullptr ? 1 : 0;
    return dev ? launch_halo8_cfg_n<BN, TH, MODE, 4, 2>(a, st) : launch_halo8_cfg_n<BN, TH, MODE, 4, 1>(a, st);
}

// 1 = launched (2: the 8-wave double-buffered variant), 0 = shape not suited (caller falls back to the plain
// implicit GEMM), < 0 = error
int try_conv_halo(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    if ((mode != CONV3 && mode != UPCONV2) || a.Wo < 32 || a.Ho < 4) return 0;
    int rc;
    {   // round 3: one 8-wave workgroup per CU with a double-buffered patch, for grids of about one workgroup per CU
        // (configs[1] levels 1-2: the single-buffer kernel's patch bursts are exposed there). MPU_HALO8=0 disables.
        // Grids of 192..400 workgroups (sweeps R3ag / R3aa: below, half the CUs idle; above, two rounds of one
        // workgroup per CU lose to the 4-wave kernel's two workgroups per CU).
        const bool h8 = env(ENV_HALO8) != 0; constexpr long h8_max = 400, h8_min = 192;
        if (h8 && dtype == MPU_BF16 && a.Ho % 8 == 0 && !a.head_w && (mode == CONV3 || !(a.Wo & 1))) {
            const long pt = (long)a.B * (a.Ho / 8) * cdiv(a.Wo, 32);
            const long g128 = pt * cdiv(a.Cout, 128), g64 = pt * cdiv(a.Cout, 64);
            const bool wide = a.Cout > 64 && g128 >= h8_min;
            const long g = wide ? g128 : g64;
            if (g >= h8_min && g <= h8_max) {
                if (mode == CONV3) rc = wide ? launch_halo8_cfg<128, 8, CONV3>(a, st) : launch_halo8_cfg<64, 8, CONV3>(a, st);
                else rc = wide ? launch_halo8_cfg<128, 8, UPCONV2>(a, st) : launch_halo8_cfg<64, 8, UPCONV2>(a, st);
                return rc ? rc : 2;
            }
        }
    }
    if (mode == UPCONV2) {                       // low-resolution patch variant of the up-convolution
        const bool up_on = env(ENV_HALO_UPCONV) != 0;
        if (!up_on || dtype != MPU_BF16 || (a.Ho & 3) || (a.Wo & 1)) return 0;
        // 8-row tiles on large grids (predict batches): twice the work per workgroup for the same patch / weight prologue
        const long up8_min = env(ENV_HALO_UP8_MIN);
        const long t8 = (long)a.B * cdiv(a.Ho, 8) * cdiv(a.Wo, 32) * cdiv(a.Cout, a.Cout > 64 ? 128 : 64);
        if (!(a.Ho & 7) && t8 >= up8_min)
            rc = a.Cout > 64 ? launch_halo_cfg<bf16_t, 128, 8, 3, UPCONV2>(a, st) : launch_halo_cfg<bf16_t, 64, 8, 3, UPCONV2>(a, st);
        else
            rc = a.Cout > 64 ? launch_halo_cfg<bf16_t, 128, 4, 3, UPCONV2>(a, st) : launch_halo_cfg<bf16_t, 64, 4, 3, UPCONV2>(a, st);
        return rc ? rc : 1;
    }
    const long tiles8 = (long)a.B * cdiv(a.Ho, 8) * cdiv(a.Wo, 32);
    const bool tall = a.Ho % 8 == 0 && tiles8 * cdiv(a.Cout, 64) >= 512;
    // 128-channel tiles: 8-row pixel tiles halve the weight re-streaming per pixel (the L2->LDS fill bounds this
    // kernel) but need >= ~4 workgroups per CU to keep the chip busy: large batches / images only (predict)
    constexpr long th8_min = 768;
    const bool tall128 = a.Ho % 8 == 0 && tiles8 * cdiv(a.Cout, 128) >= th8_min;
    // few 128-channel tiles (deep levels at small batch): 64-channel tiles double the workgroup count at nearly the
    // same L2->LDS bytes per flop (patch + 9 x 8 KB vs patch + 9 x 16 KB per chunk, half the flops)
    constexpr long bn64_below = 768;
    const long tiles4 = (long)a.B * cdiv(a.Ho, 4) * cdiv(a.Wo, 32);
    const bool narrow = a.Cout > 64 && tiles4 * cdiv(a.Cout, 128) < bn64_below;
    if (dtype == MPU_BF16) {
#ifdef MPU_HALO_KNOCKOUT_BUILD
        const int knock = (int)env(ENV_HALO_KNOCKOUT);           // dev aid: knock-out instantiations of the predict kernel
        if (a.Cout > 64 && !narrow && tall128 && knock) {
            switch (knock) {
#define MPU_KO(V) case V: rc = launch_halo_cfg<bf16_t, 128, 8, 2, CONV3, V>(a, st); break;
                MPU_KO(1) MPU_KO(2) MPU_KO(4) MPU_KO(8) MPU_KO(10) MPU_KO(16) MPU_KO(20) MPU_KO(32) MPU_KO(33) MPU_KO(64) MPU_KO(30) MPU_KO(62)
#undef MPU_KO
                default: return fail(MPU_EINVAL, "%s", "MPU_HALO_KNOCKOUT: mask not instantiated");
            }
            return rc ? rc : 1;
        }
#endif
        if (a.Cout > 64 && !narrow) rc = tall128 ? launch_halo_cfg<bf16_t, 128, 8, 2>(a, st)
                                                 : launch_halo_cfg<bf16_t, 128, 4, 3>(a, st);
        else rc = tall ? launch_halo_cfg<bf16_t, 64, 8, 3>(a, st) : launch_halo_cfg<bf16_t, 64, 4, 3>(a, st);
    } else if (dtype == MPU_F32) {
        if (a.x3) rc = a.Cout > 64 ? launch_halo_cfg<float, 128, 4, 3, CONV3, 256>(a, st) : launch_halo_cfg<float, 64, 4, 3, CONV3, 256>(a, st);
        else if (a.Cout > 64) rc = launch_halo_cfg<float, 128, 4, 3>(a, st);
        else rc = launch_halo_cfg<float, 64, 4, 3>(a, st);
    } else return 0;
    return rc ? rc : 1;
}

}  // namespace mpu
