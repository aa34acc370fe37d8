// Two kernels: conv_halo16_kernel (one tile per workgroup, every epilogue; opt-in) and, further down, conv_halo16p_kernel (its
// PERSISTENT form with the inference epilogue; the default for large predict grids). The schedule is described once, here.
//
// conv_halo16_kernel (bf16, round 4): the LDS-resident-patch 3x3 convolution for LARGE grids (predict batches, the
// configs[3] train step): one 8-wave workgroup per CU on a 16-row x 32-pixel x 128-channel tile, the two halves of the
// workgroup ONE PHASE APART as in conv_halo8 -- but with a 64-channel x 128-pixel accumulator tile per wave.
//
// Why (knock-out timing of conv_halo<128,8,2> on a 138-plane predict layer, gpurun R4b, profiles/r04_knockout_*.txt): the
// 4-wave kernel with two independent workgroups per CU takes 762 us where its data path alone (no MFMAs) takes 463 us and
// its MFMAs alone would take ~300: the two are NOT overlapped. Both co-resident workgroups run the same loop with the same
// period and nothing keeps one workgroup's fragment reads beside the other's MFMAs. conv_halo8 forces that pairing with a
// workgroup barrier per phase, but with its 64 x 64 wave tile a phase is 16 MFMAs against 16 fragment reads and the LOAD
// phase (reads + address arithmetic + DMA issue + waits, ~740 cycles) is as long as the COMPUTE phase (~780): 67 % matrix
// pipe at best. Here a wave owns 64 channels x 4 rows x 32 pixels (TN = 2, TM = 4): a phase is 32 MFMAs (1024 cycles of
// matrix pipe per wave) against 20 fragment reads in the load phase + 4 in the compute phase, so the compute phase is the
// longer one and the load phase of the other half hides beside it.
//
// Schedule. The reduction runs over ITEMS (32-channel chunk, tap): 16 MFMAs per wave each; an INTERVAL is two consecutive
// items (32 MFMAs per wave), a BLOCK two chunk32s A | B = 18 items = 9 intervals (iv 0..3: A taps (0,1) (2,3) (4,5) (6,7);
// iv 4: (A, 8) (B, 0); iv 5..8: B taps (1,2) .. (7,8)) -- unrolled, so every tap position, patch buffer and counted wait is
// a compile-time constant. Waves 4-7 start one barrier late:
//   L(iv): read the pixel fragments of both items and the weight fragments of item 0 (20 reads); request the weights of
//          interval iv+3 (stage + 3 of five) and this interval's share of the patch prefetch; counted vmcnt; barrier
//   C(iv): 8 MFMAs | read item 1's weight fragments (k-step 0) | 8 MFMAs | read (k-step 1) | 16 MFMAs; barrier
// Patch: TWO 32-channel half patches (18 x 34 pixels x 64 bytes = 39 KB each): buffer 0 holds chunk A, buffer 1 chunk B. All
// patch readers sit in load phases. B of a block is requested in L(0..2) (2 + 2 + 1 pieces per wave) and forced by the
// strict wait of L(3) (first reader: iv 4); A of the NEXT block is requested in L(5..7) -- buffer 0's last reader is iv 4 --
// and forced by L(8). So the activation stream is spread over the whole block instead of arriving as one exposed burst per
// chunk: version 1 of this kernel (64-channel single-buffered patch) stalled 13-15 k cycles at every chunk boundary, the
// whole chip waiting on HBM in lockstep (stamps, gpurun R4e: profiles/r04b_halo16_stamps.txt).
// In-order DMA queue of a wave behind the weights W(iv+1) that L(iv) must see landed: P(iv-2) W(iv+2) P(iv-1) W(iv+3) P(iv)
// with |W| = 2, |P| = {2,2,1,0,0,2,2,1,0}: allowed in flight = {6,8,9,2*,4,6,8,9,2*} (* strict: also forces the patch pieces).
// Epilogue: as conv_halo8 (bias / ReLU / folded-BN affine, staged tile, coalesced stores, ReLU mask, BN statistics of
// both passes, fused 2x2 max pooling) on the 512-pixel tile.
#include <stdlib.h>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

__device__ __forceinline__ i32x4 x_make_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa; r.y = (int)((unsigned)(pa >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void x_dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
constexpr unsigned X_POISON = 0x80001000u;                       // + any in-range byte offset (< 2 GiB - 8 KiB) stays >= num_records
__device__ __forceinline__ void x_mma(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
}

struct Halo16Cfg {
    static constexpr int NT = 9, KW = 3, BN = 128, TH = 16, TW = 32, PW = TW + 2, PH = TH + 2;
    static constexpr int PROWS = (PH * PW + 15) / 16 * 16;       // 624 patch rows of 64 bytes (612 used): 39 DMA pieces
    static constexpr int PBUF = PROWS * 64;                      // one 32-channel half patch
    static constexpr int WSTAGE = 2 * BN * 64, NWS = 5;          // a stage = the two items of an interval; requests three intervals ahead
    static constexpr int BM = TH * TW;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int EPI = BM * OROW;
    static constexpr int MAIN = 2 * PBUF + NWS * WSTAGE;
    static constexpr int CONSTS = MAIN > EPI ? MAIN : EPI;       // bias / folded-BN scale / shift of the tile's channels: NOT aliased
    static constexpr int SMEM = CONSTS + 3 * BN * 4;
};
static_assert(Halo16Cfg::SMEM <= 160 * 1024, "LDS");

// STAMP (dev aid, MPU_STAMPS=1): s_memtime stamps of wave 0 of every 8th workgroup at the phase boundaries
template <bool STAMP>
__global__ __launch_bounds__(512, 2) void conv_halo16_kernel(ConvArgs a) {
    typedef bf16_t T;
    using Cfg = Halo16Cfg;
    constexpr int NT = Cfg::NT, KW = Cfg::KW, BN = Cfg::BN, TH = Cfg::TH;
    constexpr int EPC = 8, BKE = 32, NW = 8, NTHR = 512;
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int TN = 2, TM = 4;                                // wave tile: 2 x 32 channels, 4 rows of 32 pixels
    constexpr int NPP = PROWS / 16;                              // 39 patch DMA pieces per chunk32
    constexpr int NPW = (NPP + NW - 1) / NW;                     // 5 per wave (wave 7 repeats its fourth)
    constexpr int BM = Cfg::BM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;                     // 2 x 4 waves; waves w and w + 4 share a SIMD
    // (a.dbg = first workgroup of the sampled window of 256: MPU_STAMPS_FIRST, default 0 = the first round)
    const unsigned sblk = blockIdx.x - (unsigned)a.dbg;
    unsigned long long* stamps = (STAMP && a.dbg_buf && (sblk & 7) == 0 && (sblk >> 3) < 32 && tid == 0)
                                     ? a.dbg_buf + (sblk >> 3) * 16 : nullptr;
    if (STAMP && stamps) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[14] = __builtin_amdgcn_s_memrealtime(); }
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    int t = blockIdx.x;
    const int n0 = (t % tiles_n) * BN; t /= tiles_n;
    const int x0 = (t % tiles_x) * TW; t /= tiles_x;
    const int y0 = (t % tiles_y) * TH; const int b = t / tiles_y;
    const int nc0 = a.C0 / BKE, nc1 = a.C1 / BKE;                // chunk32s per source (the launcher requires multiples of 32 and an even total)
    const int nblocks = (nc0 + nc1) / 2;
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)a.B * H * W;
    const i32x4 rs0 = x_make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rs1 = x_make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * 2L : 0);
    const i32x4 rsw = x_make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + 2 * Cfg::PBUF;

    // epilogue constants of the tile's channels -> their own LDS rows, now (no register held over the main loop, no
    // exposed global load in front of the epilogue)
    float* sbias = (float*)(smem + Cfg::CONSTS);
    if (tid < BN) {
        const bool nv = n0 + tid < a.Cout;
        const int e = nv ? n0 + tid : 0;
        sbias[tid] = (a.bias && nv) ? a.bias[e] : 0.f;
        sbias[BN + tid] = (a.post_scale && nv) ? a.post_scale[e] : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && nv) ? a.post_shift[e] : 0.f;
    }

    // --- per-lane DMA roles (64-byte rows: a 1-KB piece = 16 rows x 4 slots of 16 bytes) ----------------------
    const int drow = lane >> 2, dslot = lane & 3;
    // chunk32 c: source, first channel inside the source, byte offset of its channels inside a packed weight row
    auto chunk_src = [&](int c, bool& s1, int& cbase, int& Cs) { s1 = c >= nc0; cbase = (s1 ? c - nc0 : c) * BKE; Cs = s1 ? a.C1 : a.C0; };
    auto chunk_woff = [&](int c) { const bool s1 = c >= nc0; return (unsigned)(((s1 ? a.C0 : 0) + (s1 ? c - nc0 : c) * BKE) * 2); };
    // patch piece k (0..4) of this wave for chunk32 c into buffer pb: rows 16 (wave + 8 k) .. + 15, this lane row drow
    auto issue_patch_piece = [&](int c, int pb, int k) {
        bool s1; int cbase, Cs; chunk_src(c, s1, cbase, Cs);
        i32x4 qrs;
        qrs.x = s1 ? rs1.x : rs0.x; qrs.y = s1 ? rs1.y : rs0.y; qrs.z = s1 ? rs1.z : rs0.z; qrs.w = rs0.w;
        int q = wave + NW * k;
        if (q >= NPP) q -= NW;                                   // (wave 7, k = 4) the wave's previous piece again: same bytes, same place
        const int pr = q * 16 + drow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = y0 + py - 1, ix = x0 + px - 1;
        const bool v = pr < Cfg::PH * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int pix = v ? (b * H + iy) * W + ix : (int)npix;   // padding: the first pixel beyond the tensor
        const int ch = cbase + ((dslot ^ ((pr >> 2) & 3)) * EPC);
        const unsigned off = (unsigned)((pix * Cs + ch) * 2);
        x_dma16(qrs, off, lds0 + pb * Cfg::PBUF + __builtin_amdgcn_readfirstlane(q) * 1024);
    };
    // weights of one interval = two items (chunk32, tap): 128 rows x 64 bytes each; this wave's piece of an item = rows
    // 16 wave .. + 15
    unsigned wpo;
    {
        const int rl = wave * 16 + drow;
        const int n = n0 + rl;
        wpo = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * 2L) + (unsigned)(((dslot ^ ((rl >> 2) & 3)) * EPC) * 2) : X_POISON;
    }
    const unsigned w_tap_b = (unsigned)(a.w_tap_stride * 2L);
    auto request_item = [&](int c, int tap, int stage, int slot2) {
        const unsigned soff = (unsigned)tap * w_tap_b + chunk_woff(c);
        x_dma16(rsw, wpo + soff, ldsW + stage * Cfg::WSTAGE + slot2 * (Cfg::WSTAGE / 2) + wave * 1024);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fh = lane >> 5, l31v = lane & 31;
    const bool second = wave >= 4;
    // block = two chunk32s A (even, patch buffer 0) and B (odd, buffer 1) = 18 items (chunk, tap) = 9 intervals of two items:
    //   iv 0..3: (A, 2 iv), (A, 2 iv + 1)    iv 4: (A, 8), (B, 0)    iv 5..8: (B, 2 iv - 9), (B, 2 iv - 8)
    auto item_is_b = [](int iv, int which) { return iv > 4 || (iv == 4 && which == 1); };
    auto item_tap = [](int iv, int which) { return iv < 4 ? 2 * iv + which : (iv == 4 ? (which ? 0 : 8) : 2 * iv - 9 + which); };
    // request the weights of interval (blk, iv) -- iv may run past 8 into the next block -- into stage stg
    auto request_interval = [&](int blk, int iv, int stg) {
        if (iv >= 9) { iv -= 9; ++blk; }
        request_item(2 * blk + (item_is_b(iv, 0) ? 1 : 0), item_tap(iv, 0), stg, 0);
        request_item(2 * blk + (item_is_b(iv, 1) ? 1 : 0), item_tap(iv, 1), stg, 1);
    };
    // prologue: chunk A of block 0 (5 pieces per wave), the weights of intervals 0, 1, 2
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_patch_piece(0, 0, k);
    request_interval(0, 0, 0);
    request_interval(0, 1, 1);
    request_interval(0, 2, 2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // patch A + interval 0 landed (intervals 1, 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    if (STAMP && stamps) stamps[1] = __builtin_amdgcn_s_memtime();
    if (second) __builtin_amdgcn_s_barrier();                    // one phase behind
    // fragment offsets of this lane at k-step 0 (k-step 1: ^ 32): weights inside an item's 8-KB half stage, pixels by patch row
    const unsigned wlane = (unsigned)(2 * Cfg::PBUF + (wn * 64 + l31v) * 64) + (unsigned)((fh ^ ((l31v >> 2) & 3)) << 4);
    const int prow0 = (wm * TM) * PW + l31v;                     // patch row of the lane's pixel in tile row wm * 4, tap (0, 0)
    unsigned stb = 0;                                            // byte offset of the current weight stage
    for (int blk = 0; blk < nblocks; ++blk) {
        const bool lastb = blk + 1 >= nblocks;
#pragma unroll
        for (int iv = 0; iv < 9; ++iv) {
            // ---- L: the pixel fragments of both items and the weight fragments of item 0
            const bool sh = STAMP && stamps && blk == 0 && iv == 1;
            if (STAMP && stamps && blk == 0 && iv == 2) stamps[12] = __builtin_amdgcn_s_memtime();
            if (sh) stamps[8] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_setprio(1);
            int l31 = prow0;
            asm volatile("" : "+v"(l31));                        // (keeps the nine intervals' addresses from being hoisted into registers)
            const unsigned wst = wlane + stb;
            uint4 fa[2][2][TN], fb[2][2][TM];                    // [item][k-step][block]
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int tap = item_tap(iv, it), ky = tap / KW, kx = tap % KW;
                const unsigned pbo = item_is_b(iv, it) ? (unsigned)Cfg::PBUF : 0u;
                unsigned po[TM];
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int prow = l31 + (j + ky) * PW + kx;
                    po[j] = pbo + (unsigned)(prow * 64) + (unsigned)((fh ^ ((prow >> 2) & 3)) << 4);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (it == 0) {
#pragma unroll
                        for (int i = 0; i < TN; ++i) fa[0][ks][i] = *(const uint4*)(smem + (wst ^ (unsigned)(ks << 5)) + i * 32 * 64);
                    }
#pragma unroll
                    for (int j = 0; j < TM; ++j) fb[it][ks][j] = *(const uint4*)(smem + (po[j] ^ (unsigned)(ks << 5)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sh) stamps[13] = __builtin_amdgcn_s_memtime();
            {   // DMA requests: the weights of the interval three ahead (stage + 3), then this interval's share of the patch
                // prefetch: iv 0..2 -> chunk B of this block into buffer 1, iv 5..7 -> chunk A of the next block into buffer 0
                const bool wreq = iv + 3 < 9 || !lastb;
                int stn = (int)(stb >> 14) + 3; if (stn >= Cfg::NWS) stn -= Cfg::NWS;
                if (wreq) request_interval(blk, iv + 3, stn);
                constexpr int pk0[9] = {0, 2, 4, 5, 5, 0, 2, 4, 5}, pk1[9] = {2, 4, 5, 5, 5, 2, 4, 5, 5};
                if (iv <= 2) {
#pragma unroll
                    for (int k = pk0[iv]; k < pk1[iv]; ++k) issue_patch_piece(2 * blk + 1, 1, k);
                } else if (iv >= 5 && iv <= 7 && !lastb) {
#pragma unroll
                    for (int k = pk0[iv]; k < pk1[iv]; ++k) issue_patch_piece(2 * blk + 2, 0, k);
                }
                __builtin_amdgcn_sched_barrier(0);
                // The weights of interval iv + 1 have landed (in-order queue: everything requested before them too); allowed
                // in flight behind them: see the table in the header comment. iv 3 / iv 8 also force the patch pieces
                // (buffer 1 is read from iv 4 on, buffer 0 from the next block's iv 0 on).
                if (!lastb || iv <= 4) {
                    constexpr int allow[9] = {6, 8, 9, 2, 4, 6, 8, 9, 2};
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow[iv]) : "memory");
                } else {
                    constexpr int allow_last[9] = {0, 0, 0, 0, 0, 4, 2, 0, 0};
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow_last[iv]) : "memory");
                }
            }
            if (sh) stamps[9] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (sh) stamps[10] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            if (sh) stamps[11] = __builtin_amdgcn_s_memtime();
            // ---- C: the 32 MFMAs of the two items; item 1's weight fragments are read under item 0's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) x_mma(fa[0][0][i], fb[0][0][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i) fa[1][0][i] = *(const uint4*)(smem + wst + (Cfg::WSTAGE / 2) + i * 32 * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) x_mma(fa[0][1][i], fb[0][1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i) fa[1][1][i] = *(const uint4*)(smem + (wst ^ 32u) + (Cfg::WSTAGE / 2) + i * 32 * 64);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN) : "memory");     // item 1 k-step 0 (k-step 1 may be in flight)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) x_mma(fa[1][0][i], fb[1][0][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) x_mma(fa[1][1][i], fb[1][1][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (sh) stamps[7] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            stb += Cfg::WSTAGE; if (stb == Cfg::NWS * Cfg::WSTAGE) stb = 0;
        }
    }
    if (!second) __builtin_amdgcn_s_barrier();                   // the second half's last compute phase
    __builtin_amdgcn_s_setprio(0);
    if (STAMP && stamps) stamps[2] = __builtin_amdgcn_s_memtime();

    // --- epilogue: as conv_halo8, on the 512-pixel tile -------------------------------------------
    constexpr int OROW = Cfg::OROW;
    {
        const float lo = a.relu ? 0.f : -__builtin_inff();
        const int nbase = wn * 64 + 4 * (lane >> 5);
        unsigned char* drow = smem + ((wm * TM) * TW + (lane & 31)) * OROW + nbase * 2;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            float4 bq[4], sq[4], hq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = nbase + i * 32 + 8 * q;
                bq[q] = *(const float4*)(sbias + nl);
                sq[q] = *(const float4*)(sbias + BN + nl);
                hq[q] = *(const float4*)(sbias + 2 * BN + nl);
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[i][j][4 * q] + bq[q].x, acc[i][j][4 * q + 1] + bq[q].y,
                                  acc[i][j][4 * q + 2] + bq[q].z, acc[i][j][4 * q + 3] + bq[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
                    if (a.post_scale) {
                        v[0] = v[0] * sq[q].x + hq[q].x; v[1] = v[1] * sq[q].y + hq[q].y;
                        v[2] = v[2] * sq[q].z + hq[q].z; v[3] = v[3] * sq[q].w + hq[q].w;
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
    if (STAMP && stamps) stamps[3] = __builtin_amdgcn_s_memtime();
    {
        // thread = (16-byte channel piece c, pixel r0 of a tile row); a pass covers one 32-pixel row of the tile
        constexpr int CPRO = BN * 2 / 16, RPI = NTHR / CPRO, NIT = BM / RPI;
        static_assert(RPI == TW && NIT == TH, "a pass is one tile row");
        const long npo = npix;
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npo * a.Cout * 2L), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0,
                                                                              (int)(npo * a.Cout * 2L), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? a.bn_x : a.out), 0,
                                                                              (int)(npo * a.Cout * 2L), 0x00020000);
        const int c = tid % CPRO, r0 = tid / CPRO;
        const int n = n0 + c * EPC;
        const int pixB = a.Cout * 2;
        const int obase = ((b * H + y0) * W + x0) * pixB;
        const int lane_off = n * 2 + r0 * pixB;
        const unsigned char* srow = smem + r0 * OROW + c * 16;
        const bool n_ok = n < a.Cout && x0 + r0 < W;
        float ssum[EPC], ssq[EPC], bmu[EPC], bis[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            ssum[e] = 0.f; ssq[e] = 0.f;
            const bool on = a.bn_x && n + e < a.Cout;
            bmu[e] = on ? a.bn_mean[n + e] : 0.f; bis[e] = on ? a.bn_invstd[n + e] : 0.f;
        }
        // two half-tiles of eight passes: the ReLU masks / BatchNorm inputs of a half are requested up front (a load inside
        // a pass would be waited for on the spot); eight passes' worth of them fit the registers the accumulators freed
        constexpr int HP = NIT / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 mkv[HP], bxv[HP];
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                const int yy = half * HP + k;
                const bool in = n_ok && (y0 + yy < H);
                const unsigned o = (unsigned)(obase + lane_off + yy * W * pixB);
                mkv[k] = __builtin_amdgcn_raw_buffer_load_b128(rsm, (a.mask && in) ? o : OOB, 0, 0);
                bxv[k] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (a.bn_x && in) ? o : OOB, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < HP; ++k) {
                const int yy = half * HP + k;
                const bool ok = n_ok && (y0 + yy < H);
                u32x4 val = *(const u32x4*)(srow + yy * RPI * OROW);
                if (a.stats && ok) {
                    const uint32_t wv[4] = {val.x, val.y, val.z, val.w};
                    const uint32_t xw[4] = {bxv[k].x, bxv[k].y, bxv[k].z, bxv[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(wv[e] << 16), hi = __uint_as_float(wv[e] & 0xffff0000u);
                        const float xl = __uint_as_float(xw[e] << 16), xh = __uint_as_float(xw[e] & 0xffff0000u);
                        const float fl = a.bn_x ? (xl - bmu[2 * e]) * bis[2 * e] : lo;
                        const float fh2 = a.bn_x ? (xh - bmu[2 * e + 1]) * bis[2 * e + 1] : hi;
                        ssum[2 * e] += lo; ssq[2 * e] += lo * fl; ssum[2 * e + 1] += hi; ssq[2 * e + 1] += hi * fh2;
                    }
                }
                const unsigned off = ok ? (unsigned)(obase + lane_off + yy * W * pixB) : OOB;
                if (a.mask) {
                    const u32x4 mk = mkv[k];
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
            const int ptile = blockIdx.x / tiles_n;
            for (int v = tid; v < BN * 2; v += NTHR) {
                const int col = v >> 1, st2 = v & 1;
                double acc2 = 0.0;
                for (int rl = 0; rl < RPI; ++rl) acc2 += (double)red[(rl * BN + col) * 2 + st2];
                if (n0 + col < a.Cout) a.stats[((long)st2 * a.Cout + n0 + col) * (gridDim.x / tiles_n) + ptile] = (float)acc2;
            }
        }
    }
    if (STAMP && stamps) {
        stamps[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[5] = __builtin_amdgcn_s_memtime();
        stamps[15] = __builtin_amdgcn_s_memrealtime();
        stamps[6] = (unsigned long long)(nblocks * 9);
    }
}

// ---- persistent form (round 4, second half): ONE workgroup per CU walks a strided list of pixel tiles of its n-tile ----------
// The item stream of consecutive tiles is one stream: the weight ring wraps (same n-tile, same weights), the next tile's
// first half patch is prefetched in L(5..7) of the current tile's last block exactly as the next block's is, and the
// accumulators leave between C(8) of a tile and L(0) of the next, wave by wave, through 2 KB of wave-private LDS (see the
// epilogue below) -- no staged workgroup tile, no barrier, no drain: prologue (8.6 k cycles) and epilogue (12.2 k of a
// 107-k-cycle workgroup on a 4-chunk layer, stamps in profiles/r04b_halo16_stamps.txt) shrink to the conversion + store
// issue of one wave (both halves in turn, the other half's compute phase beside it).
// vmcnt: stores count on the same in-order counter as the DMA requests (gfx9 family), so the two load phases after an
// epilogue allow its NST stores in flight on top of their usual allowance; by L(2) they are older than the weights that
// phase needs anyway. Inference epilogue only (bias, ReLU, folded-BN affine, optional fused 2x2 max pooling): no ReLU
// mask, no BatchNorm statistics -- those launches take conv_halo16_kernel / conv_halo.
template <bool STAMP, bool POOL>
__global__ __launch_bounds__(512, 2) void conv_halo16p_kernel(ConvArgs a, int ptiles, int gp) {
    typedef bf16_t T;
    using Cfg = Halo16Cfg;
    constexpr int KW = Cfg::KW, BN = Cfg::BN, TH = Cfg::TH;
    constexpr int EPC = 8, BKE = 32, NW = 8;
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int TN = 2, TM = 4;
    constexpr int NPP = PROWS / 16, NPW = (NPP + NW - 1) / NW;
    constexpr int NST = TN * TM * 2 + (POOL ? TN * (TM / 2) * 2 : 0);        // stores per wave and tile
    static_assert(9 + NST + 1 <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    unsigned long long* stamps = (STAMP && a.dbg_buf && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 32 && tid == 0)
                                     ? a.dbg_buf + (blockIdx.x >> 3) * 16 : nullptr;
    if (STAMP && stamps) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[14] = __builtin_amdgcn_s_memrealtime(); }
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = H / TH;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int n0 = ((int)blockIdx.x % tiles_n) * BN;             // the workgroup's n-tile: fixed, so weights and constants are too
    const int g = (int)blockIdx.x / tiles_n;                      // its pixel tiles: g, g + gp, ...
    const int ntl = (ptiles - g + gp - 1) / gp;
    auto decode = [&](int p, int& tb, int& ty0, int& tx0) {
        tx0 = (p % tiles_x) * TW; p /= tiles_x;
        ty0 = (p % tiles_y) * TH; tb = p / tiles_y;
    };
    const int nc0 = a.C0 / BKE, nc1 = a.C1 / BKE;
    const int nblocks = (nc0 + nc1) / 2;
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)a.B * H * W;
    const i32x4 rs0 = x_make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rs1 = x_make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * 2L : 0);
    const i32x4 rsw = x_make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + 2 * Cfg::PBUF;

    float* sbias = (float*)(smem + Cfg::CONSTS);
    if (tid < BN) {
        const bool nv = n0 + tid < a.Cout;
        const int e = nv ? n0 + tid : 0;
        sbias[tid] = (a.bias && nv) ? a.bias[e] : 0.f;
        sbias[BN + tid] = (a.post_scale && nv) ? a.post_scale[e] : 1.f;
        sbias[2 * BN + tid] = (a.post_scale && nv) ? a.post_shift[e] : 0.f;
    }

    const int drow = lane >> 2, dslot = lane & 3;
    auto chunk_src = [&](int c, bool& s1, int& cbase, int& Cs) { s1 = c >= nc0; cbase = (s1 ? c - nc0 : c) * BKE; Cs = s1 ? a.C1 : a.C0; };
    auto chunk_woff = [&](int c) { const bool s1 = c >= nc0; return (unsigned)(((s1 ? a.C0 : 0) + (s1 ? c - nc0 : c) * BKE) * 2); };
    auto issue_patch_piece = [&](int c, int pb, int k, int tb, int ty0, int tx0) {
        bool s1; int cbase, Cs; chunk_src(c, s1, cbase, Cs);
        i32x4 qrs;
        qrs.x = s1 ? rs1.x : rs0.x; qrs.y = s1 ? rs1.y : rs0.y; qrs.z = s1 ? rs1.z : rs0.z; qrs.w = rs0.w;
        int q = wave + NW * k;
        if (q >= NPP) q -= NW;
        const int pr = q * 16 + drow;
        const int py = pr / PW, px = pr - py * PW;
        const int iy = ty0 + py - 1, ix = tx0 + px - 1;
        const bool v = pr < Cfg::PH * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const int pix = v ? (tb * H + iy) * W + ix : (int)npix;
        const int ch = cbase + ((dslot ^ ((pr >> 2) & 3)) * EPC);
        const unsigned off = (unsigned)((pix * Cs + ch) * 2);
        x_dma16(qrs, off, lds0 + pb * Cfg::PBUF + __builtin_amdgcn_readfirstlane(q) * 1024);
    };
    unsigned wpo;
    {
        const int rl = wave * 16 + drow;
        const int n = n0 + rl;
        wpo = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * 2L) + (unsigned)(((dslot ^ ((rl >> 2) & 3)) * EPC) * 2) : X_POISON;
    }
    const unsigned w_tap_b = (unsigned)(a.w_tap_stride * 2L);
    auto request_item = [&](int c, int tap, int stage, int slot2) {
        const unsigned soff = (unsigned)tap * w_tap_b + chunk_woff(c);
        x_dma16(rsw, wpo + soff, ldsW + stage * Cfg::WSTAGE + slot2 * (Cfg::WSTAGE / 2) + wave * 1024);
    };

    f32x16 acc[TN][TM];
    const int fh = lane >> 5, l31v = lane & 31;
    const bool second = wave >= 4;
    auto item_is_b = [](int iv, int which) { return iv > 4 || (iv == 4 && which == 1); };
    auto item_tap = [](int iv, int which) { return iv < 4 ? 2 * iv + which : (iv == 4 ? (which ? 0 : 8) : 2 * iv - 9 + which); };
    // weights of interval iv (0..8) of block blk into stage stg
    auto request_interval = [&](int blk, int iv, int stg) {
        request_item(2 * blk + (item_is_b(iv, 0) ? 1 : 0), item_tap(iv, 0), stg, 0);
        request_item(2 * blk + (item_is_b(iv, 1) ? 1 : 0), item_tap(iv, 1), stg, 1);
    };
    int tb, ty0, tx0;
    decode(g, tb, ty0, tx0);
#pragma unroll
    for (int k = 0; k < NPW; ++k) issue_patch_piece(0, 0, k, tb, ty0, tx0);
    request_interval(0, 0, 0);
    request_interval(0, 1, 1);
    request_interval(0, 2, 2);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: the constants' LDS writes, read right after the barrier)
    __builtin_amdgcn_s_barrier();
    if (STAMP && stamps) stamps[1] = __builtin_amdgcn_s_memtime();
    // the accumulators START at the bias of their channels (one LDS read per four of them, here and after every tile):
    // the epilogue then has no add and no zeroing
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *(const float4*)(sbias + wn * 64 + 4 * fh + i * 32 + 8 * q);
                acc[i][j][4 * q] = bv.x; acc[i][j][4 * q + 1] = bv.y; acc[i][j][4 * q + 2] = bv.z; acc[i][j][4 * q + 3] = bv.w;
            }
    if (second) __builtin_amdgcn_s_barrier();
    const unsigned wlane = (unsigned)(2 * Cfg::PBUF + (wn * 64 + l31v) * 64) + (unsigned)((fh ^ ((l31v >> 2) & 3)) << 4);
    const int prow0 = (wm * TM) * PW + l31v;
    unsigned stb = 0;
    const int pixB = a.Cout * 2;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npix * a.Cout * 2L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(POOL ? a.pooled : (void*)a.out, 0,
                                                                          (int)((POOL ? (npix >> 2) : npix) * a.Cout * 2L), 0x00020000);
    for (int ti = 0; ti < ntl; ++ti) {
        const bool last_tile = ti + 1 >= ntl;
        int nb_ = tb, ny0 = ty0, nx0 = tx0;
        if (!last_tile) decode(g + (ti + 1) * gp, nb_, ny0, nx0);
        for (int blk = 0; blk < nblocks; ++blk) {
            const bool endb = blk + 1 >= nblocks;
            const bool lastb = endb && last_tile;
            const bool after_epi = blk == 0 && ti > 0;            // the stores of the previous tile are still in the queue
            const int nblk = endb ? 0 : blk + 1;                  // the block the look-ahead runs into (next tile's first at a tile's end)
            const int pb_b = endb ? nb_ : tb, pb_y = endb ? ny0 : ty0, pb_x = endb ? nx0 : tx0;
#pragma unroll
            for (int iv = 0; iv < 9; ++iv) {
                __builtin_amdgcn_s_setprio(1);
                int l31 = prow0;
                asm volatile("" : "+v"(l31));
                const unsigned wst = wlane + stb;
                uint4 fa[2][2][TN], fb[2][2][TM];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int tap = item_tap(iv, it), ky = tap / KW, kx = tap % KW;
                    const unsigned pbo = item_is_b(iv, it) ? (unsigned)Cfg::PBUF : 0u;
                    unsigned po[TM];
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const int prow = l31 + (j + ky) * PW + kx;
                        po[j] = pbo + (unsigned)(prow * 64) + (unsigned)((fh ^ ((prow >> 2) & 3)) << 4);
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        if (it == 0) {
#pragma unroll
                            for (int i = 0; i < TN; ++i) fa[0][ks][i] = *(const uint4*)(smem + (wst ^ (unsigned)(ks << 5)) + i * 32 * 64);
                        }
#pragma unroll
                        for (int j = 0; j < TM; ++j) fb[it][ks][j] = *(const uint4*)(smem + (po[j] ^ (unsigned)(ks << 5)));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const bool wreq = iv + 3 < 9 || !lastb;
                    int stn = (int)(stb >> 14) + 3; if (stn >= Cfg::NWS) stn -= Cfg::NWS;
                    if (wreq) request_interval(iv + 3 < 9 ? blk : nblk, iv + 3 < 9 ? iv + 3 : iv + 3 - 9, stn);
                    constexpr int pk0[9] = {0, 2, 4, 5, 5, 0, 2, 4, 5}, pk1[9] = {2, 4, 5, 5, 5, 2, 4, 5, 5};
                    if (iv <= 2) {
#pragma unroll
                        for (int k = pk0[iv]; k < pk1[iv]; ++k) issue_patch_piece(2 * blk + 1, 1, k, tb, ty0, tx0);
                    } else if (iv >= 5 && iv <= 7 && !lastb) {
#pragma unroll
                        for (int k = pk0[iv]; k < pk1[iv]; ++k) issue_patch_piece(2 * nblk, 0, k, pb_b, pb_y, pb_x);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (after_epi && iv < 2) {                    // + the previous tile's stores (issued between W(2) and W(3))
                        constexpr int allow_epi[2] = {6 + NST, 8 + NST};
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow_epi[iv < 2 ? iv : 0]) : "memory");
                    } else if (!lastb || iv <= 4) {
                        constexpr int allow[9] = {6, 8, 9, 2, 4, 6, 8, 9, 2};
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow[iv]) : "memory");
                    } else {
                        constexpr int allow_last[9] = {0, 0, 0, 0, 0, 4, 2, 0, 0};
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow_last[iv]) : "memory");
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[0][0][i], fb[0][0][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[1][0][i] = *(const uint4*)(smem + wst + (Cfg::WSTAGE / 2) + i * 32 * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[0][1][i], fb[0][1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[1][1][i] = *(const uint4*)(smem + (wst ^ 32u) + (Cfg::WSTAGE / 2) + i * 32 * 64);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN) : "memory");
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[1][0][i], fb[1][0][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[1][1][i], fb[1][1][j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                stb += Cfg::WSTAGE; if (stb == Cfg::NWS * Cfg::WSTAGE) stb = 0;
            }
        }
        if (STAMP && stamps && ti < 6) stamps[2 + 2 * ti] = __builtin_amdgcn_s_memtime();      // (wave 0: C(8) of the tile's last block done)
        // ---- the tile's accumulators out. An MFMA lane owns 4 consecutive channels (8 (r / 4) + 4 (lane / 32) + r % 4 of
        // block i) of ONE pixel: stored as they lie that is 8 bytes into each of 32 cache lines per instruction (measured:
        // 909 vs 775 us on the 2-chunk predict layer -- the store pipe, not the matrix pipe, sets the pace). So every
        // 32-channel x 32-pixel piece (i, j) takes a turn through 2 KB of WAVE-PRIVATE LDS -- the wave's own two 1-KB slices
        // of the weight stage that its own L(0) request refills next (interval 7's weights: both halves are done with them,
        // nobody else ever writes these slices) -- and leaves as 16-byte pieces, 64 contiguous bytes per pixel. No
        // workgroup barrier: one wave's LDS instructions execute in order.
        {
            const float lo = a.relu ? 0.f : -__builtin_inff();
            int lx = l31v, ln = lane, fq = fh;
            asm volatile("" : "+v"(lx), "+v"(ln), "+v"(fq));      // (per-tile address arithmetic stays here: hoisted out of the tile loop it spills)
            int stn = (int)(stb >> 14) + 3; if (stn >= Cfg::NWS) stn -= Cfg::NWS;
            unsigned char* sl0 = smem + 2 * Cfg::PBUF + stn * Cfg::WSTAGE + wave * 1024;     // pixels 0..15 of a row: 16 x 64 bytes
            unsigned char* sl1 = sl0 + Cfg::WSTAGE / 2;                                          // pixels 16..31
            // write side: lane = pixel lx; its 8 bytes of slot q land at slot q ^ ((lx >> 1) & 3) of the pixel's 64-byte row
            unsigned char* wbase = ((lx & 16) ? sl1 : sl0) + (lx & 15) * 64 + 8 * fq;
            const int wsw = (lx >> 1) & 3;
            // read side: lane = (pixel rp of a 16-pixel pass, 16-byte slot rs)
            const int rp = ln >> 2, rs = ln & 3;
            const unsigned roff = (unsigned)(rp * 64 + ((rs ^ ((rp >> 1) & 3)) << 4));
            const int nbase = wn * 64 + 4 * fq;
            const int nch = n0 + wn * 64 + 8 * rs;
            const int gbase = ((tb * H + ty0 + wm * TM) * W + tx0 + rp) * pixB + nch * 2;
            const bool ok0 = tx0 + rp < W, ok1 = tx0 + 16 + rp < W;
            const int Hp = H >> 1, Wp = W >> 1;
            const int pbase = POOL ? (((tb * Hp + ((ty0 + wm * TM) >> 1)) * Wp + ((tx0 + rp) >> 1)) * pixB + nch * 2) : 0;
            // software pipeline over the eight pieces p = (i, j): convert piece p while the read-back of piece p - 1 is in
            // flight, store piece p - 1, then stage piece p (LDS executes one wave's instructions in order: the write of p
            // cannot overtake the read of p - 1)
            auto pool = [&](const u32x4& x, const u32x4& y) {     // 2 x 2 max: the row pair is in this lane, the pixel pair four lanes apart
                u32x4 m;
                m.x = piece_max<T>(x.x, y.x); m.y = piece_max<T>(x.y, y.y);
                m.z = piece_max<T>(x.z, y.z); m.w = piece_max<T>(x.w, y.w);
                u32x4 n;                                          // row_shl:4 -- lane r reads lane r + 4
                n.x = (uint32_t)__builtin_amdgcn_mov_dpp((int)m.x, 0x104, 0xf, 0xf, true);
                n.y = (uint32_t)__builtin_amdgcn_mov_dpp((int)m.y, 0x104, 0xf, 0xf, true);
                n.z = (uint32_t)__builtin_amdgcn_mov_dpp((int)m.z, 0x104, 0xf, 0xf, true);
                n.w = (uint32_t)__builtin_amdgcn_mov_dpp((int)m.w, 0x104, 0xf, 0xf, true);
                m.x = piece_max<T>(m.x, n.x); m.y = piece_max<T>(m.y, n.y);
                m.z = piece_max<T>(m.z, n.z); m.w = piece_max<T>(m.w, n.w);
                return m;
            };
            u32x4 rb0, rb1, prev0, prev1;                          // read-back of the piece in flight; the even row of a pooled pair
#ifndef MPU_H16P_KO
#define MPU_H16P_KO 0                                              // dev builds: 1 no global stores, 2 no staging, 4 no bias reload, 8 no conversion
#endif
            auto flush = [&](int i, int j) {
                const bool n_ok = nch + i * 32 < a.Cout;
                const unsigned o = (unsigned)(gbase + j * W * pixB + i * 64);
                if (MPU_H16P_KO & 1) { asm volatile("" :: "v"(rb0), "v"(rb1)); return; }
                __builtin_amdgcn_raw_buffer_store_b128(rb0, rso, (ok0 && n_ok) ? o : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(rb1, rso, (ok1 && n_ok) ? o + 16u * (unsigned)pixB : OOB, 0, 0);
                if (POOL) {
                    if (j & 1) {
                        const u32x4 m0 = pool(prev0, rb0), m1 = pool(prev1, rb1);
                        const bool pe = !(rp & 1) && n_ok;
                        const unsigned po = (unsigned)(pbase + (j >> 1) * Wp * pixB + i * 64);
                        __builtin_amdgcn_raw_buffer_store_b128(m0, rsp, (pe && ok0) ? po : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(m1, rsp, (pe && ok1) ? po + 8u * (unsigned)pixB : OOB, 0, 0);
                    } else { prev0 = rb0; prev1 = rb1; }
                }
            };
            float4 sq[4], hq[4];                                   // folded-BN scale / shift of the 32-channel block (1, 0 without)
#pragma unroll
            for (int pc = 0; pc < TN * TM; ++pc) {
                const int i = pc / TM, j = pc % TM;
                if (j == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nl = nbase + i * 32 + 8 * q;
                        sq[q] = *(const float4*)(sbias + BN + nl);
                        hq[q] = *(const float4*)(sbias + 2 * BN + nl);
                    }
                }
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
                    if (MPU_H16P_KO & 8) {
                        pk[q].x = __float_as_uint(acc[i][j][4 * q]) ^ __float_as_uint(acc[i][j][4 * q + 1]);
                        pk[q].y = __float_as_uint(acc[i][j][4 * q + 2]) ^ __float_as_uint(acc[i][j][4 * q + 3]);
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(acc[i][j][4 * q + e]), "v"(lo));   // (fmaxf / fmed3 compile to canonicalise + max)
                    if (a.post_scale) {                            // (a real branch: the asm keeps it from becoming 128 selects)
                        asm volatile("");
                        v[0] = v[0] * sq[q].x + hq[q].x; v[1] = v[1] * sq[q].y + hq[q].y;
                        v[2] = v[2] * sq[q].z + hq[q].z; v[3] = v[3] * sq[q].w + hq[q].w;
                    }
                    pk[q].x = f32x2_to_bf16x2(v[0], v[1]);
                    pk[q].y = f32x2_to_bf16x2(v[2], v[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pc > 0) flush((pc - 1) / TM, (pc - 1) % TM);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!(MPU_H16P_KO & 2)) *(uint2*)(wbase + ((q ^ wsw) << 4)) = pk[q];
                    if (MPU_H16P_KO & 4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] = 0.f;
                        continue;
                    }
                    const float4 bv = *(const float4*)(sbias + nbase + i * 32 + 8 * q);      // the next tile starts at the bias again
                    acc[i][j][4 * q] = bv.x; acc[i][j][4 * q + 1] = bv.y; acc[i][j][4 * q + 2] = bv.z; acc[i][j][4 * q + 3] = bv.w;
                }
                if (MPU_H16P_KO & 2) { rb0.x = pk[0].x; rb0.y = pk[0].y; rb0.z = pk[1].x; rb0.w = pk[1].y; rb1.x = pk[2].x; rb1.y = pk[2].y; rb1.z = pk[3].x; rb1.w = pk[3].y; }
                else { rb0 = *(const u32x4*)(sl0 + roff); rb1 = *(const u32x4*)(sl1 + roff); }
                __builtin_amdgcn_sched_barrier(0);
            }
            flush(TN - 1, TM - 1);
        }
        // The epilogue is a phase of its own: [h0 epilogue | h1 C(8)], [h0 L(0) | h1 epilogue], [h0 C(0) | h1 L(0)] costs
        // 2 x 3.6 k + 1.45 k cycles at a tile boundary where [h0 epilogue + L(0) | h1 C(8)], [h0 C(0) | h1 epilogue + L(0)]
        // cost 2 x 5.05 k (measured: tile period 92.8 k -> 90.9 k cycles on the 4-chunk layer; a second barrier in the middle of
        // the epilogue -- [E1 | C(8)] [E2 | E1] [L(0) | E2] -- measured 90.1 k there and 50.9 k vs 49.7 k on the 2-chunk layer: not kept).
        __builtin_amdgcn_s_barrier();
        if (STAMP && stamps && ti < 6) stamps[3 + 2 * ti] = __builtin_amdgcn_s_memtime();      // (wave 0: epilogue issued)
        tb = nb_; ty0 = ny0; tx0 = nx0;
    }
    if (!second) __builtin_amdgcn_s_barrier();
    if (STAMP && stamps) { stamps[15] = __builtin_amdgcn_s_memrealtime(); }
}

int launch_halo16(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = Halo16Cfg;
    unsigned long long* sbuf = stamp_buffer();                  // MPU_STAMPS=1: the instrumented instantiation
    auto kern = sbuf ? conv_halo16_kernel<true> : conv_halo16_kernel<false>;
    ConvArgs a = a_in;
    a.dbg_buf = sbuf;
    a.dbg = (int)env(ENV_STAMPS_FIRST);
    if (a.w_elems <= 0) a.w_elems = (Cfg::NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    if (M * cmax * 2L >= (1L << 31) - 8192 || a.w_elems * 2L >= (1L << 31) - 8192 || M * a.Cout * 2L >= (1L << 31) - 8192)
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    const long tiles = (long)a.B * cdiv(a.Ho, Cfg::TH) * cdiv(a.Wo, Cfg::TW) * cdiv(a.Cout, Cfg::BN);
    const long ptiles = tiles / cdiv(a.Cout, Cfg::BN);
    if (a.stats && a.stats_rows) {
        if (ptiles * 2 * a.Cout <= a.stats_cap) *a.stats_rows = (int)ptiles;
        else { a.stats = nullptr; *a.stats_rows = 0; }
    } else a.stats = nullptr;
    if (a.pooled && a.pooled_done && !a.mask && !(a.Ho & 1) && !(a.Wo & 1)) *a.pooled_done = 1;
    else a.pooled = nullptr;
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * Cfg::NT * (a.C0 + a.C1), st);
    launch_k(kern, dim3((unsigned)tiles), dim3(512), Cfg::SMEM, st, a);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

// persistent form: one workgroup per CU (the LDS footprint allows no second one), gp workgroups per n-tile
int launch_halo16p(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = Halo16Cfg;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        MPU_CHECK_HIP(hipGetDevice(&dev));
        MPU_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    unsigned long long* sbuf = stamp_buffer();
    ConvArgs a = a_in;
    a.dbg_buf = sbuf;
    if (a.w_elems <= 0) a.w_elems = (Cfg::NT - 1) * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16p_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16p_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_halo16p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
    if (M * cmax * 2L >= (1L << 31) - 8192 || a.w_elems * 2L >= (1L << 31) - 8192 || M * a.Cout * 2L >= (1L << 31) - 8192)
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    const int tiles_n = cdiv(a.Cout, Cfg::BN);
    const long ptiles = (long)a.B * (a.Ho / Cfg::TH) * cdiv(a.Wo, Cfg::TW);
    const long wgs = env(ENV_HALO16P_WGS);                       // fewer workgroups (tests: many tiles each on small shapes)
    long gp = (wgs > 0 ? wgs : ncu) / tiles_n; if (gp < 1) gp = 1; if (gp > ptiles) gp = ptiles;
    if (a.stats_rows) *a.stats_rows = 0;
    a.stats = nullptr;
    if (a.pooled && a.pooled_done && !(a.Ho & 1) && !(a.Wo & 1)) *a.pooled_done = 1;
    else a.pooled = nullptr;
    auto kern = a.pooled ? (sbuf ? conv_halo16p_kernel<true, true> : conv_halo16p_kernel<false, true>)
                         : (sbuf ? conv_halo16p_kernel<true, false> : conv_halo16p_kernel<false, false>);
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * Cfg::NT * (a.C0 + a.C1), st);
    launch_k(kern, dim3((unsigned)(gp * tiles_n)), dim3(512), Cfg::SMEM, st, a, (int)ptiles, (int)gp);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// 3 = launched, 0 = shape not suited (the caller falls back to conv_halo), < 0 = error.
// Large grids of 128-channel tiles on 16-row x 32-pixel pixel tiles: predict batches, the configs[3] train step.
int try_conv_halo16(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    // Two kernels. conv_halo16p (persistent, inference epilogue) is ON by default for the launches it covers (MPU_HALO16P=0:
    // off): 3.41 vs 3.55-3.59 ms over four predict-size layers against conv_halo<128,8,2>, a predict 160.8 vs 164.3 ms
    // (gpurun R4p). conv_halo16 (one tile per workgroup, every epilogue) stays opt-in (MPU_HALO16=1): it ties with the
    // 4-wave kernel (gpurun R4e/R4f: 128 -> 128 @ 138 x 128^2: 799 vs 770 us; 256 -> 256 @ 64^2: 587 vs 606; 256 -> 128:
    // 1164 vs 1150), prologue + epilogue (8 % + 11 % of a workgroup's life) exposed with one workgroup per CU. DESIGN section 5.
    const bool on = env(ENV_HALO16) != 0, pers = env(ENV_HALO16P) != 0;
    const long mt = env(ENV_HALO16_MIN);
    const long min_tiles = mt >= 0 ? mt : 768, min_tiles_p = mt >= 0 ? mt : 512;       // (persistent: two tiles per CU and up)
    if ((!on && !pers) || dtype != MPU_BF16 || mode != CONV3 || a.Cout <= 64 || a.Wo < 32 || (a.Ho & 15) || a.head_w) return 0;
    // 32-channel chunks, two per block: sources that are multiples of 32 channels, an even number of chunks in total
    if ((a.C0 & 31) || (a.C1 & 31) || (((a.C0 + a.C1) >> 5) & 1)) return 0;
    const long tiles = (long)a.B * (a.Ho / 16) * cdiv(a.Wo, 32) * cdiv(a.Cout, 128);
    if (pers && tiles >= min_tiles_p && !a.mask && !a.stats && !a.bn_x && !(a.Cout & 7)) {
        const int rc = launch_halo16p(a, st);
        return rc ? rc : 5;
    }
    if (!on || tiles < min_tiles) return 0;
    const int rc = launch_halo16(a, st);
    return rc ? rc : 3;
}

}  // namespace mpu
