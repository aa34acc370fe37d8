// conv_halo16p_kernel (bf16, round 4): the PERSISTENT LDS-resident-patch 3x3 convolution with the inference epilogue, the
// default for large predict grids. (Round 4 also carried its one-tile-per-workgroup ancestor conv_halo16_kernel with the
// training epilogues; it tied with conv_halo<128,8,2> -- prologue + epilogue exposed with one workgroup per CU, DESIGN
// section 5 round 4 -- and was removed in round 5.) One 8-wave workgroup per CU on 16-row x 32-pixel x 128-channel tiles,
// the two halves of the workgroup ONE PHASE APART as in conv_halo8 -- but with a 64-channel x 128-pixel accumulator tile
// per wave.
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
    static constexpr int MAIN = 2 * PBUF + NWS * WSTAGE;
    static constexpr int CONSTS = MAIN;                          // bias / folded-BN scale / shift of the tile's channels: NOT aliased
    static constexpr int SMEM = CONSTS + 3 * BN * 4;
};
static_assert(Halo16Cfg::SMEM <= 160 * 1024, "LDS");

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
// mask, no BatchNorm statistics -- those launches take conv_halo.
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

// persistent form: one workgroup per CU (the LDS footprint allows no second one), gp workgroups per n-tile
int launch_halo16p(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = Halo16Cfg;
    const int ncu = device_cu_count();
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
        mark_used_on_device(attr_set);
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

// 5 = launched, 0 = shape not suited (the caller falls back to conv_halo), < 0 = error.
// Large INFERENCE grids of 128-channel tiles on 16-row x 32-pixel pixel tiles (predict batches). On by default for the
// launches it covers (MPU_HALO16P=0: off): 3.41 vs 3.55-3.59 ms over four predict-size layers against conv_halo<128,8,2>, a
// predict 160.8 vs 164.3 ms (gpurun R4p).
int try_conv_halo16(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    const long mt = env(ENV_HALO16_MIN);
    const long min_tiles_p = mt >= 0 ? mt : 512;                 // two tiles per CU and up
    if (env(ENV_HALO16P) == 0 || dtype != MPU_BF16 || mode != CONV3 || a.Cout <= 64 || a.Wo < 32 || (a.Ho & 15) || a.head_w) return 0;
    // 32-channel chunks, two per block: sources that are multiples of 32 channels, an even number of chunks in total
    if ((a.C0 & 31) || (a.C1 & 31) || (((a.C0 + a.C1) >> 5) & 1)) return 0;
    const long tiles = (long)a.B * (a.Ho / 16) * cdiv(a.Wo, 32) * cdiv(a.Cout, 128);
    if (tiles < min_tiles_p || a.mask || a.stats || a.bn_x || (a.Cout & 7)) return 0;
    const int rc = launch_halo16p(a, st);
    return rc ? rc : 5;
}

}  // namespace mpu
