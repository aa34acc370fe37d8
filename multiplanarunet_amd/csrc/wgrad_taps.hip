// wgrad_taps_kernel: weight gradient of a 3x3 SAME convolution with ALL NINE TAPS accumulated by
// one workgroup (bf16, the high-resolution layers: few channels, very many pixels).
//
//   dW[tap][ci][co] = sum_m X[m @ tap][ci] * dZ[m][co]
//
// The per-tap kernel (wgrad_glds_kernel) streams an X tile and a dZ tile per 32-pixel K step for
// ONE tap: at 64x64 channels that is 8 KB of L2->LDS traffic per 0.26 MFLOP (32 flop/B) and the
// nine taps re-read the same pixels nine times. Here a workgroup walks a 32-pixel-wide column
// strip of one image, one output row per K step: it keeps a rolling window of three X rows (34
// pixels each, halo included) plus the dZ row in LDS, so a step fetches ONE new X row and ONE dZ
// row (8.25 KB) for all nine taps: 2.36 MFLOP, 286 flop/B. The L2->LDS fill and the HBM re-reads
// drop 9x; the kernel is MFMA-bound.
//
// MFMA shape: v_mfma_f32_16x16x32_bf16 (K = 32 pixels = one step). Wave w owns input channels
// [16w, 16w+16) of the 64-channel tile, all 64 output channels, all nine taps: 36 accumulators of
// 4 VGPRs. Per step a wave reads the four dZ fragments once (shared by the nine taps) and one X
// fragment per tap (the tap's shifted pixels): 26 LDS transpose reads for 36 MFMAs.
// Both operands are pixel-major in memory (K-major): fragments come from ds_read_b64_tr_b16.
// Bias gradient: one extra MFMA per wave and step against an all-ones A fragment sums dZ over
// the pixels (wave w: output channels [16w, 16w+16)).
// A workgroup is TWO such 4-wave groups working on neighbouring strips with their own LDS rings; at
// the end they add their accumulators through LDS (taps 0-4 end up in group 0, taps 5-8 in group 1)
// so only one fp32 partial copy per PAIR of strips goes to HBM: the partial write + re-read is
// this kernel's dominant HBM traffic.
// Output: fp32 partials [pair][tap][ci/4][co][4] (+ bias partials [pair][co]) reduced in fixed order
// by wgrad_reduce*_kernel: deterministic, no atomics.
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace {

constexpr int XPX = 40;                  // pixels per staged X row (34 used: 32 + halo; 5 DMA pieces of 8)
constexpr int XROWB = XPX * 128;         // bytes per staged X row (64 channels bf16 per pixel)
constexpr int NXR = 5;                   // X row ring: rows t..t+2 in use, t+3 landed, t+4 in flight
constexpr int ZROWB = 32 * 128;          // bytes per staged dZ row
constexpr int NZR = 3;

__device__ __forceinline__ i32x4 t_make_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa; r.y = (int)((unsigned)(pa >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void t_dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}
__device__ __forceinline__ s16x8 t_frag(const unsigned char* p_lo, const unsigned char* p_hi) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p_hi));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

constexpr int GROUP_LDS = NXR * XROWB + NZR * ZROWB;            // rings of one 4-wave group
constexpr int TAPS_SMEM = 2 * GROUP_LDS;                        // 75776 B: the accumulator exchange at the end runs in rounds that fit the rings (round 6)

// One round of the end-of-workgroup accumulator exchange between the two 4-wave groups: group 1 hands taps [A0, A1) (+ the
// bias sums) to group 0, group 0 hands taps [B0, B1) to group 1, through [wave][floats][lane] arrays at the start of the LDS.
template <int NT, int A0, int A1, bool BIAS, int B0, int B1>
__device__ __forceinline__ void taps_xround(f32x4 (&acc)[NT][4], f32x4& accdb, unsigned char* smem_all, const int grp,
                                            const int wave, const int lane) {
    constexpr int NAF = (A1 - A0) * 16 + (BIAS ? 4 : 0), NBF = (B1 - B0) * 16;               // floats per lane, each way
    static_assert(4 * (NAF + NBF) * 256 <= TAPS_SMEM, "exchange round larger than the rings");
    float* xa = reinterpret_cast<float*>(smem_all) + (wave * NAF) * 64 + lane;               // [wave][NAF][lane]
    float* xb = reinterpret_cast<float*>(smem_all + 4 * NAF * 256) + (wave * NBF) * 64 + lane;
    if (grp == 1) {
#pragma unroll
        for (int tp = A0; tp < A1; ++tp)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) xa[(((tp - A0) * 4 + cb) * 4 + r) * 64] = acc[tp][cb][r];
        if constexpr (BIAS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xa[((A1 - A0) * 16 + r) * 64] = accdb[r];
        }
        asm volatile("; xround: group 1 wrote" ::: "memory");
    } else {
#pragma unroll
        for (int tp = B0; tp < B1; ++tp)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) xb[(((tp - B0) * 4 + cb) * 4 + r) * 64] = acc[tp][cb][r];
        asm volatile("; xround: group 0 wrote" ::: "memory");
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int tp = A0; tp < A1; ++tp)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tp][cb][r] += xa[(((tp - A0) * 4 + cb) * 4 + r) * 64];
        if constexpr (BIAS) accdb[0] += xa[(A1 - A0) * 16 * 64];
        asm volatile("; xround: group 0 added" ::: "memory");
    } else {
#pragma unroll
        for (int tp = B0; tp < B1; ++tp)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tp][cb][r] += xb[(((tp - B0) * 4 + cb) * 4 + r) * 64];
        asm volatile("; xround: group 1 added" ::: "memory");
    }
}

// MODE CONV3: nine taps, X rows y-1..y+1 at full resolution. MODE UPCONV2 (nearest-upsample x2 + 2x2 conv): four
// taps; staged "X row r" is the low-resolution row (y0 + r) >> 1 (each low-res row is staged for both upsampled rows
// it feeds), 17 low-res pixels wide, and tap (ky, kx) reads staged row t + ky at pixel (px + kx) >> 1.
// The body handles the taps [T0, T1) of one (strip pair, channel tile) -- all of them in every instantiation in use.
// (Round 3, s_memtime stamps inside the loop: a K step takes ~2300 cycles against ~1260 of MFMA issue for the two waves
// of a SIMD; the counted vmcnt wait costs < 100 of them and a ring one step deeper changes nothing -- the loop is
// bound by MFMA issue + the LDS transpose reads in front of them, not by the L2 -> LDS latency.)
// STAG (round 3, after conv_halo8): the two 4-wave groups run ONE PHASE APART. A step is a load phase L (the 26 transposing
// fragment reads of the step's dZ row and nine shifted X fragments, into registers) and a compute phase C (the step's 36
// MFMAs, the DMA requests of step t+3 between them); a workgroup barrier closes each phase and group 1 starts one barrier
// late, so one group's MFMAs run beside the other group's LDS reads on the SIMDs they share. The rings allow the longer
// request distance because the request follows the reads of the slot it overwrites (X row t+5 -> slot of row t, dZ row
// t+3 -> slot of row t, both read in L(t)); a wave ends L(t) with only the group it requested in C(t-1) in flight. No
// stamps / run-time ring arithmetic in this variant: the ring slots are running scalars.
template <int MODE, int T0, int T1, bool STAG>
__device__ __forceinline__ void wgrad_taps_body(const WgradArgs& a, const TapsPlan& p, unsigned char* smem_all,
                                                const int tile, const int pair, const unsigned bid) {
    constexpr int NT = T1 - T0, KW = MODE == UPCONV2 ? 2 : 3;
    constexpr int NTALL = MODE == UPCONV2 ? 4 : 9;
    constexpr int NXRT = NXR, NZRT = NZR, DIST = 2;              // request distance (steps)
    constexpr int NXP = MODE == UPCONV2 ? 3 : 5;                 // DMA pieces (8 pixels) per staged X row
    constexpr unsigned OOB = 0xfffffff0u;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    // dev aid (MPU_STAMPS=1): phase boundaries of every 8th workgroup, wave 0
    unsigned long long* stamps = (!STAG && a.dbg_buf && (bid & 7) == 0 && (bid >> 3) < 32 && tid == 0)
                                     ? a.dbg_buf + (bid >> 3) * 16 : nullptr;
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();
    unsigned char* smem = smem_all + grp * GROUP_LDS;
    const int H = a.Ho, W = a.Wo;
    const int Cin = a.C0 + a.C1;
    const int tiles_co = (a.Cout + 63) / 64;
    const int strip = pair * 2 + grp;
    const bool valid = strip < p.nstrips;
    const int xs = strip % p.sx; int t_ = strip / p.sx;
    const int ys = t_ % p.sy; const int b = t_ / p.sy;
    const int x0 = xs * 32, y0 = ys * p.RH;
    // dtype "bf16x3" (a.x3 = -images per plane): the batch is three plane pairs (x: hi | lo | hi, dz: hi | hi | lo) of TWO stored
    // planes each (hi | lo) -- the image index of a strip is folded back onto the plane that holds its values
    const int pl = a.x3 < 0 ? -a.x3 : 0;
    const int bX = (pl && b >= 2 * pl) ? b - 2 * pl : b;
    const int bZ = (pl && b >= pl) ? b - pl : b;
    const int nsteps = valid ? (y0 + p.RH < H ? y0 + p.RH : H) - y0 : 0;
    int nsteps_wg;                                               // both groups run the same number of barriers
    {
        const int s0 = pair * 2, s1 = pair * 2 + 1;
        const int ya = ((s0 / p.sx) % p.sy) * p.RH, yb = ((s1 / p.sx) % p.sy) * p.RH;
        const int n0 = (ya + p.RH < H ? ya + p.RH : H) - ya;
        const int n1 = s1 < p.nstrips ? (yb + p.RH < H ? yb + p.RH : H) - yb : 0;
        nsteps_wg = n0 > n1 ? n0 : n1;
    }
    const int ci0 = (tile / tiles_co) * 64, co0 = (tile % tiles_co) * 64;
    const bool s1 = ci0 >= a.C0 && a.C1 > 0;             // the ci tile lies in one concat source (C0 % 64 == 0 then)
    const int Cs = s1 ? a.C1 : a.C0, cs0 = s1 ? ci0 - a.C0 : ci0;
    const long npix = (long)a.B * H * W;
    const int Hi = MODE == UPCONV2 ? H / 2 : H, Wi = MODE == UPCONV2 ? W / 2 : W;
    const i32x4 rsx = t_make_rsrc(s1 ? a.x1 : a.x0, (long)a.B * Hi * Wi * Cs * 2L);
    const i32x4 rsz = t_make_rsrc(a.dz, npix * a.Cout * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsZ = lds0 + NXR * XROWB;

    // ---- DMA: piece q of a step's group: q < 5: 8 pixels of the new X row; q >= 5: 8 pixels of the dZ row.
    // 128-byte pixel rows, 16-byte slots permuted per pixel (swz below) so that a transposing fragment read is free of
    // bank conflicts. One ds_read_b64_tr_b16 is serviced in two groups of 32 lanes; a group reads 32 contiguous bytes (this
    // wave's 16 channels) of EIGHT pixels: c0 .. c0+3 (lane bits 2-3) and c0+8 .. c0+11 (lane bit 4; up-conv: c, c+1(,c+2)
    // and c+4 ..). The 64 banks span 256 bytes = 8 such 32-byte blocks, and the block of pixel c is
    // 4 (c & 1) + (wave ^ swizzle bits). Round 3 flipped only the 64-byte granule by bit 1 of c: pixels c and c+8 (up-conv:
    // c and c+4) shared their banks -- a 2-way conflict on every fragment read, SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS = 1.9
    // (profiles/r03b_conv_pmc_cfg1_shapes.txt). Now the 32-byte granule is flipped as well, by bit 3 of c
    // (bit 2 for the low-resolution up-conv rows): the eight pixels of a group take the eight blocks once.
    auto swz = [](int c, bool lowres) { return (((c >> 1) & 1) << 2) | (((c >> (lowres ? 2 : 3)) & 1) << 1); };
    const int dpx = lane >> 3, dsl = lane & 7;
    // per-lane parts of the request offsets are fixed for the whole strip (column, channel chunk, their validity);
    // a request then costs a scalar row base + one add instead of ~20 integer instructions per piece and step
    unsigned xlane[2]; unsigned zlane;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = wave + 4 * k;
        const int c = q * 8 + dpx;
        const int ix = MODE == UPCONV2 ? x0 / 2 + c : x0 - 1 + c;
        const int ch = (dsl ^ swz(c, MODE == UPCONV2)) * 8;
        const bool v = q < NXP && c < (MODE == UPCONV2 ? 17 : 34) && (unsigned)ix < (unsigned)Wi && cs0 + ch < Cs;
        xlane[k] = v ? (unsigned)((ix * Cs + cs0 + ch) * 2) : OOB;
    }
    {
        const int c = wave * 8 + dpx, x = x0 + c;
        const int ch = (dsl ^ swz(c, false)) * 8;
        zlane = (x < W && co0 + ch < a.Cout) ? (unsigned)((x * a.Cout + co0 + ch) * 2) : OOB;
    }
    auto issue_x = [&](int r) {                                  // staged X row r of this strip
        int iy; bool rowok;
        if (MODE == UPCONV2) { const int uy = y0 + r; rowok = uy < H; iy = uy >> 1; }
        else { iy = y0 - 1 + r; rowok = (unsigned)iy < (unsigned)H; }
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (r % NXRT) * XROWB);
        const unsigned rowoff = (unsigned)(((bX * Hi + iy) * Wi) * Cs * 2);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int q = wave + 4 * k;
            if (q < NXP) {                                       // wave-uniform
                const unsigned off = (rowok && xlane[k] != OOB) ? rowoff + xlane[k] : OOB;
                t_dma16(rsx, off, base + q * 1024);
            }
        }
    };
    auto issue_z = [&](int t) {                                  // dZ row of step t = image row y0 + t
        const int y = y0 + t;
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsZ + (t % NZRT) * ZROWB);
        const unsigned rowoff = (unsigned)(((bZ * H + y) * W) * a.Cout * 2);
        const unsigned off = (y < H && zlane != OOB) ? rowoff + zlane : OOB;
        t_dma16(rsz, off, base + wave * 1024);
    };
    // DMAs per wave in one step group (X row pieces wave, wave+4, ... < NXP, plus one dZ piece): wait until only the
    // group issued last is outstanding
    // a step group = this wave's X row pieces (wave, wave+4, ... < NXP) plus one dZ piece; wait until only the `keep`
    // groups issued last are outstanding
    auto wait_keep_groups = [&](int keep) {
        const int per = (NXP - wave + 3) / 4 + 1;                // CONV3: 3,2,2,2   UPCONV2: 2,2,2,1
        const int n = per * keep;
        if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ---- fragment addressing (16x16x32: lane group g = lane>>4 holds k = 8g..8g+7; i = lane&15 the row/col) ----
    const int g = lane >> 4, i = lane & 15;
    // a transpose read covers 4 k-rows x 16 columns; lane i addresses k-row (i>>2), columns (i&3)*4
    int offA[KW][2];                                             // X: per kx shift and low/high half, inside a row slot
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cu = kx + 8 * g + (i >> 2) + 4 * h;        // (upsampled) pixel column of this k-row
            const int c = MODE == UPCONV2 ? cu >> 1 : cu;        // staged pixel column
            const int slot16 = (wave * 2 + ((i & 3) >> 1)) ^ swz(c, MODE == UPCONV2);
            offA[kx][h] = c * 128 + (slot16 << 4) + (i & 1) * 8;
        }
    int offB[4][2];                                              // dZ: per 16-channel block and half
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 8 * g + (i >> 2) + 4 * h;
            const int slot16 = (cb * 2 + ((i & 3) >> 1)) ^ swz(c, false);
            offB[cb][h] = c * 128 + (slot16 << 4) + (i & 1) * 8;
        }

    f32x4 acc[NT][4];
#pragma unroll
    for (int tp = 0; tp < NT; ++tp)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[tp][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accdb = {0.f, 0.f, 0.f, 0.f};
    const s16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

    if constexpr (!STAG) {
        // ---- pipeline: step t uses X rows t, t+1, t+2 and dZ row t; issues X row t+4 and dZ row t+2 -----------
        if (valid) {
            issue_x(0); issue_x(1); issue_x(2); issue_z(0);
            issue_x(3); issue_z(1);
        }
        // wait for the first group (rows 0..2 + dZ 0); the later ones may still be in flight
        wait_keep_groups(DIST - 1);
        __builtin_amdgcn_s_barrier();
        if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < nsteps_wg; ++t) {
            if (t >= nsteps) {                                       // the other group still has rows to do
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                continue;
            }
            if (t + DIST < nsteps) { issue_x(t + DIST + 2); issue_z(t + DIST); }
            const unsigned char* zb = smem + NXR * XROWB + (t % NZRT) * ZROWB;
            s16x8 bz[4];
    #pragma unroll
            for (int cb = 0; cb < 4; ++cb) bz[cb] = t_frag(zb + offB[cb][0], zb + offB[cb][1]);
            const unsigned char* xr[KW];
    #pragma unroll
            for (int ky = 0; ky < KW; ++ky) xr[ky] = smem + ((t + ky) % NXRT) * XROWB;
            s16x8 af = t_frag(xr[T0 / KW] + offA[T0 % KW][0], xr[T0 / KW] + offA[T0 % KW][1]);
    #pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                s16x8 an = af;
                if (tp + 1 < NT) {
                    const int ky = (T0 + tp + 1) / KW, kx = (T0 + tp + 1) % KW;
                    an = t_frag(xr[ky] + offA[kx][0], xr[ky] + offA[kx][1]);
                }
    #pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[tp][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bz[cb], acc[tp][cb], 0, 0, 0);
                af = an;
            }
            if (a.fuse_db && T0 == 0) {                              // (the half that holds tap 0 also sums dZ)
                s16x8 bw = bz[0];
                if (wave == 1) bw = bz[1]; else if (wave == 2) bw = bz[2]; else if (wave == 3) bw = bz[3];
                accdb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bw, accdb, 0, 0, 0);
            }
            if (stamps && (t == 4 || t == 5)) stamps[8 + 4 * (t - 4)] = __builtin_amdgcn_s_memtime();
            // group t+1 has landed (the DIST-1 groups behind it may be in flight); all waves are done with step t's rows
            {
                int younger = nsteps - 2 - t;                        // groups t+2 .. min(t+DIST, nsteps-1) are outstanding behind group t+1
                if (younger > DIST - 1) younger = DIST - 1;
                wait_keep_groups(younger > 0 ? younger : 0);
            }
            if (stamps && (t == 4 || t == 5)) stamps[9 + 4 * (t - 4)] = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stamps && (t == 4 || t == 5)) stamps[10 + 4 * (t - 4)] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            if (stamps && (t == 4 || t == 5)) stamps[11 + 4 * (t - 4)] = __builtin_amdgcn_s_memtime();
        }

    } else {
        // ---- staggered pipeline (see the comment above the template) ----------------------------------------
        const int per = (NXP - wave + 3) / 4 + 1;                // DMA pieces of this wave per step group (CONV3: 3,2,2,2)
        // running request state: next X row / dZ row to request, their ring slots and global row offsets
        const int xrow_b = Wi * Cs * 2, zrow_b = W * a.Cout * 2;     // bytes per input / dZ image row
        int rx = 0;                                              // next staged X row index
        unsigned xslot = 0, zslot = 0;                           // ring slot byte offsets of the next requests
        int iy_next = MODE == UPCONV2 ? 0 : y0 - 1;              // (CONV3) image row of staged row rx
        auto req_x = [&]() {
            int iy; bool rowok;
            if (MODE == UPCONV2) { const int uy = y0 + rx; rowok = uy < H; iy = uy >> 1; }
            else { iy = iy_next; rowok = (unsigned)iy < (unsigned)H; }
            const unsigned rowoff = (unsigned)((bX * Hi + iy) * xrow_b);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = wave + 4 * k;
                if (q < NXP) {
                    const unsigned off = (rowok && xlane[k] != OOB) ? rowoff + xlane[k] : OOB;
                    t_dma16(rsx, off, lds0 + xslot + q * 1024);
                }
            }
            ++rx; ++iy_next;
            xslot += XROWB; if (xslot == NXR * XROWB) xslot = 0;
        };
        int rz = 0;
        auto req_z = [&]() {
            const int y = y0 + rz;
            const unsigned rowoff = (unsigned)((bZ * H + y) * zrow_b);
            const unsigned off = (y < H && zlane != OOB) ? rowoff + zlane : OOB;
            t_dma16(rsz, off, ldsZ + zslot + wave * 1024);
            ++rz;
            zslot += ZROWB; if (zslot == NZR * ZROWB) zslot = 0;
        };
        auto wait_keep = [&](int groups) {                       // this wave's pieces of the `groups` youngest step groups stay in flight
            if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (groups == 1) { if (per == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (per == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
            else { if (per == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (per == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        };
        if (valid) {
            req_x(); req_x(); req_x(); req_z();                  // step 0: rows 0..2, dZ 0
            req_x(); req_z();                                    // step 1
            req_x(); req_z();                                    // step 2
        }
        wait_keep(2);
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();              // one phase behind
        unsigned x0s = 0, x1s = XROWB, x2s = 2 * XROWB, zs = 0;  // ring slots of rows t, t+1, t+2 and dZ row t
        const bool dbw = a.fuse_db && T0 == 0;
        for (int t = 0; t < nsteps_wg; ++t) {
            if (t >= nsteps) {                                   // the other group still has rows to do
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_barrier();
                continue;
            }
            // ---- L(t)
            __builtin_amdgcn_s_setprio(1);
            const unsigned char* zb = smem + NXR * XROWB + zs;
            s16x8 bz[4], af[NT];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) bz[cb] = t_frag(zb + offB[cb][0], zb + offB[cb][1]);
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int ky = (T0 + tp) / KW, kx = (T0 + tp) % KW;
                const unsigned char* xr = smem + (ky == 0 ? x0s : (ky == 1 ? x1s : x2s));
                af[tp] = t_frag(xr + offA[kx][0], xr + offA[kx][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // step t+1's group (requested in C(t-2)) has landed; the one requested in C(t-1) may be in flight
            wait_keep(t + 2 < nsteps ? 1 : 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- C(t)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[tp][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tp], bz[cb], acc[tp][cb], 0, 0, 0);
                if (tp == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 3 < nsteps) { req_x(); req_z(); }    // step t+3: X row t+5 -> slot of row t, dZ row t+3 -> slot of dZ row t
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (dbw) {
                s16x8 bw = bz[0];
                if (wave == 1) bw = bz[1]; else if (wave == 2) bw = bz[2]; else if (wave == 3) bw = bz[3];
                accdb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bw, accdb, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            x0s = x1s; x1s = x2s; x2s += XROWB; if (x2s == NXR * XROWB) x2s = 0;
            zs += ZROWB; if (zs == NZR * ZROWB) zs = 0;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();              // group 1's last compute phase
    }
    if (stamps) stamps[2] = __builtin_amdgcn_s_memtime();
    // ---- combine the two groups through LDS (the rings are free now): group 0 ends up with taps 0-4 and the
    // bias sums, group 1 with taps 5-8; each stores its share of the pair's partial copy.
    // Round 6: in ROUNDS that fit the rings' own 74 KB (one pass needed 148 KB, which left a compute unit's LDS to this
    // workgroup alone): the kernel now leaves 84 KB of LDS and 64 registers per lane free, exactly the room of one
    // workgroup of the optimizer kernel that runs beside it (adam_pack_lean_kernel, unet_ops.hip).
    {
        constexpr int NA = (NT + 1) / 2;                                                     // taps kept by group 0
        if constexpr (NT == 9) {
            taps_xround<NT, 0, 2, true, 5, 7>(acc, accdb, smem_all, grp, wave, lane);
            __syncthreads();
            taps_xround<NT, 2, 4, false, 7, 9>(acc, accdb, smem_all, grp, wave, lane);
            __syncthreads();
            taps_xround<NT, 4, 5, false, 9, 9>(acc, accdb, smem_all, grp, wave, lane);
        } else {
            static_assert(NT == 4, "exchange rounds are written for nine or four taps");
            taps_xround<NT, 0, NA, true, NA, NT>(acc, accdb, smem_all, grp, wave, lane);
        }
    }
    if (stamps) stamps[3] = __builtin_amdgcn_s_memtime();
    // ---- partial sums: [pair][tap][ci / 4][co][4 ci] -- a lane's accumulator (four consecutive input channels of one
    // output channel) is ONE 16-byte store, 16 lanes cover 256 contiguous bytes; the [ci][co] layout needed four
    // 4-byte stores per accumulator and the tail of the kernel was store-issue bound. The reduction un-interleaves
    // (store_dw_sum, conv_igemm.hip).
    float* P = a.partial + (long)pair * NTALL * Cin * a.Cout;
    const int cin4 = Cin >> 2;
    const int cib = (ci0 + wave * 16 + 4 * g) >> 2;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
        if ((tp < (NT + 1) / 2) != (grp == 0)) continue;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int co = co0 + cb * 16 + i;
            if (cib < cin4 && co < a.Cout)
                *(f32x4*)(P + (((long)(T0 + tp) * cin4 + cib) * a.Cout + co) * 4) = acc[tp][cb];
        }
    }
    if (a.fuse_db && T0 == 0 && grp == 0 && ci0 == 0 && g == 0) {          // every row of accdb holds the column sums: take row 0
        const int co = co0 + wave * 16 + i;
        if (co < a.Cout) a.db_partial[(long)pair * a.Cout + co] = accdb[0];
    }
    if (stamps) {
        stamps[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[5] = __builtin_amdgcn_s_memtime();
        stamps[6] = (unsigned long long)nsteps_wg;
    }
}

// one workgroup of a job: bid = its index inside the job's grid (XCD-aware decode: the tiles of one strip pair run on one
// XCD -- shared X / dZ in its L2; a job's first workgroup sits at a multiple of 8 of the launch grid)
template <int MODE, bool STAG>
__device__ __forceinline__ void wgrad_taps_entry(const WgradArgs& a, const TapsPlan& p, unsigned bid, unsigned char* smem_all) {
    constexpr int NTALL = MODE == UPCONV2 ? 4 : 9;
    const int Cin = a.C0 + a.C1;
    const int ntile = ((a.Cout + 63) / 64) * ((Cin + 63) / 64);
    const int xcd = bid & 7, slot = bid >> 3;
    const int npairs = (p.nstrips + 1) / 2;
    const int sub = slot % ntile, pair = (slot / ntile) * 8 + xcd;
    if (pair >= npairs) return;
    wgrad_taps_body<MODE, 0, NTALL, STAG>(a, p, smem_all, sub, pair, bid);
}

template <int MODE, bool STAG>
__global__ __launch_bounds__(512, 1) void wgrad_taps_kernel(WgradArgs a, TapsPlan p) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem_all[];
    wgrad_taps_entry<MODE, STAG>(a, p, blockIdx.x, smem_all);
}

// Every deferred wgrad_taps job of a backward pass in ONE launch (kernels.h: WgradGroup): workgroup -> job by the table's
// block ranges (each a multiple of 8), then exactly the single-job kernel.
struct TapsGroupTable { int njobs, _pad; TapsGroupJob job[TAPS_GROUP_MAX]; };
static_assert(sizeof(TapsGroupTable) <= 4096, "kernel-argument size");
static_assert(sizeof(TapsGroupTable) <= 4096, "kernel-argument size");
template <bool STAG>
__global__ __launch_bounds__(512, 1) void wgrad_taps_group_kernel(TapsGroupTable t) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem_all[];
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < t.njobs; ++k) if ((int)blockIdx.x >= t.job[k].blk_begin) j = k;
    const WgradArgs a = t.job[j].a;
    const TapsPlan p = t.job[j].p;
    const unsigned bid = blockIdx.x - (unsigned)t.job[j].blk_begin;
    if (t.job[j].mode == UPCONV2) wgrad_taps_entry<UPCONV2, STAG>(a, p, bid, smem_all);
    else wgrad_taps_entry<CONV3, STAG>(a, p, bid, smem_all);
}

}  // namespace

// Strip decomposition: 32-pixel-wide column strips of RH rows; aims at ~2 workgroups per CU while keeping
// the number of fp32 partial copies (one per strip) small: their write + re-read is the kernel's HBM traffic.
TapsPlan wgrad_taps_plan(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, bool grouped) {
    TapsPlan p; p.use = 0; p.RH = 0; p.sx = 0; p.sy = 0; p.nstrips = 0; p.split = 0;
    const bool on = env(ENV_WGRAD_TAPS) != 0; constexpr int target = 512; constexpr long max_cico = TAPS_MAX_CICO;
    const int Cin = C0 + C1;
    if (!on || dtype != MPU_BF16 || (mode != CONV3 && mode != UPCONV2) || W < 32 || H < 8) return p;
    if (mode == UPCONV2 && ((H | W) & 1)) return p;
    if (C1 > 0 && C0 % 64 != 0) return p;
    if ((long)Cin * Cout > max_cico) return p;
    const long M = (long)B * H * W;
    if (M * (C0 > C1 ? C0 : C1) * 2L >= (1L << 31) || M * Cout * 2L >= (1L << 31)) return p;
    // (splitting the nine taps of a strip pair over two workgroups halves the partial traffic but doubles the K steps of
    // every workgroup: measured 8-25 % slower per layer on configs[1], round 2; removed)
    p.split = 0;
    const int ntile = cdiv(Cin, 64) * cdiv(Cout, 64) * (p.split ? 2 : 1);       // workgroups per strip pair
    const int sx = cdiv(W, 32);
    int best = 0; long bestd = 1L << 60;
    for (int rh = 8; rh <= 256; rh *= 2) {
        const long ns = (long)B * sx * cdiv(H, rh);
        if (ns * ntile > TAPS_MAX_WGS) continue;
        const long wgs = ns * ntile;
        // a job of a grouped launch need not fill the chip on its own: a quarter of the strips (64 workgroups of four
        // times the rows per layer instead of 256): a quarter of the fp32 partial copies, the per-workgroup prologue /
        // epilogue amortised over four times the K steps (round 3 sweep on configs[1]: 3.03 -> 2.94 -> 2.91 ms per step
        // for 512 / 256 / 128 strips x tiles; 64: the layers fall below the parallelism bound)
        const long tg = grouped ? (target + 3) / 4 : target;
        const long d = wgs > tg ? wgs - tg : tg - wgs;
        if (d < bestd) { bestd = d; best = rh; }
        if (rh >= H) break;
    }
    if (!best) return p;
    p.RH = best; p.sx = sx; p.sy = cdiv(H, best); p.nstrips = B * sx * p.sy;
    if ((long)p.nstrips * ntile < (grouped ? 64 : 128)) return p;   // too little parallelism: the per-tap kernel splits finer
    if (p.nstrips < 3) return p;                                 // (one strip pair would write straight into dW, which is not in the
                                                                 //  ci-interleaved partial layout: always go through the reduction)
    p.use = 1;
    return p;
}

int wgrad_taps_grid(int /*mode*/, const WgradArgs& a, const TapsPlan& p) {
    const int Cin = a.C0 + a.C1;
    const int ntile = cdiv(Cin, 64) * cdiv(a.Cout, 64);
    const int npairs = (p.nstrips + 1) / 2;
    return cdiv(npairs, 8) * 8 * ntile;
}

static bool taps_stag() {                                        // MPU_WGRAD_TAPS_STAG=0: the lockstep groups
    return env(ENV_WGRAD_TAPS_STAG) != 0;
}

static int taps_attrs() {
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_kernel<CONV3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_kernel<UPCONV2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_kernel<CONV3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_kernel<UPCONV2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_group_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_taps_group_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TAPS_SMEM));
        mark_used_on_device(attr_set);
    }
    return MPU_OK;
}

int launch_wgrad_taps(int mode, const WgradArgs& a_in, const TapsPlan& p, hipStream_t st) {
    WgradArgs a = a_in;
    a.dbg_buf = stamp_buffer();
    const int grid = wgrad_taps_grid(mode, a, p);
    { const int rc = taps_attrs(); if (rc) return rc; }
    if (taps_stag() && !a.dbg_buf) {                             // (the stamps live in the lockstep variant)
        if (mode == UPCONV2) launch_k(wgrad_taps_kernel<UPCONV2, true>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, a, p);
        else launch_k(wgrad_taps_kernel<CONV3, true>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, a, p);
    } else {
        if (mode == UPCONV2) launch_k(wgrad_taps_kernel<UPCONV2, false>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, a, p);
        else launch_k(wgrad_taps_kernel<CONV3, false>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, a, p);
    }
    return launch_ok();
}

int launch_wgrad_taps_group(const TapsGroupJob* jobs, int n, hipStream_t st) {
    if (n <= 0) return MPU_OK;
    if (n > TAPS_GROUP_MAX) return fail(MPU_EINVAL, "%s", "wgrad_taps group: too many jobs");
    { const int rc = taps_attrs(); if (rc) return rc; }
    TapsGroupTable t; t.njobs = n; t._pad = 0;
    int grid = 0;
    // the longest jobs first: the tail of the launch is then made of short workgroups
    int order[TAPS_GROUP_MAX];
    for (int k = 0; k < n; ++k) order[k] = k;
    for (int i = 1; i < n; ++i)
        for (int k = i; k > 0 && jobs[order[k]].p.RH > jobs[order[k - 1]].p.RH; --k) { const int tmp = order[k]; order[k] = order[k - 1]; order[k - 1] = tmp; }
    for (int k = 0; k < n; ++k) {
        t.job[k] = jobs[order[k]];
        t.job[k].a.dbg_buf = nullptr;
        t.job[k].blk_begin = grid;
        grid += wgrad_taps_grid(t.job[k].mode, t.job[k].a, t.job[k].p);      // a multiple of 8: the XCD decode of every job stays aligned
    }
    if (taps_stag()) launch_k(wgrad_taps_group_kernel<true>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, t);
    else launch_k(wgrad_taps_group_kernel<false>, dim3((unsigned)grid), dim3(512), TAPS_SMEM, st, t);
    return launch_ok();
}

}  // namespace mpu
