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
// Schedule (g = running tap index, stage = g & 3; waves 4-7 start one barrier late):
//   L(g): read the tap's pixel fragments of all four k-steps and the weight fragments of k-steps 0-1 into registers;
//         request the weights of tap g+2 into stage (g+2) & 3; counted vmcnt (tap g+1 has landed); lgkmcnt(0); barrier
//   C(g): 8 MFMAs of k-step 0 | read weight fragments of k-step 2 | 8 MFMAs k-step 1 | read k-step 3 | 16 MFMAs; barrier
//   interval:   I0    I1    I2    I3 ...        stage g is read in I(2g) .. I(2g+2); its next occupant, tap g+4, is
//   waves 0-3:  L(0)  C(0)  L(1)  C(1)          requested in L(g+2) = I(2g+4) / I(2g+5): two barriers later.
//   waves 4-7:   -    L(0)  C(0)  L(1)
// The halo patch (18 x 34 pixels x 64 channels = 77 KB) is SINGLE-buffered (two of them do not fit beside four weight
// stages): all its readers sit in load phases, so at a chunk boundary the second half's L(8) is the last reader, both halves
// request their pieces of the next chunk's patch right behind that barrier (the first half before idling one interval, the
// second half in front of its C(8)), and the schedule slips by ONE interval per 64-channel chunk (~9 % of a chunk).
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
    static constexpr int PROWS = (PH * PW + 7) / 8 * 8;          // 616 patch rows (612 used)
    static constexpr int PATCH = PROWS * 128;
    static constexpr int WSTAGE = BN * 128, NWS = 4;
    static constexpr int BM = TH * TW;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int EPI = BM * OROW;
    static constexpr int MAIN = PATCH + NWS * WSTAGE;
    static constexpr int CONSTS = MAIN > EPI ? MAIN : EPI;       // bias / folded-BN scale / shift of the tile's channels: NOT aliased
    static constexpr int SMEM = CONSTS + 3 * BN * 4;
};
static_assert(Halo16Cfg::SMEM <= 160 * 1024, "LDS");

// FULL: every chunk holds 64 channels (C0, C1 multiples of 64): no k-step guards in the stream
template <bool FULL>
__global__ __launch_bounds__(512, 2) void conv_halo16_kernel(ConvArgs a) {
    typedef bf16_t T;
    using Cfg = Halo16Cfg;
    constexpr int NT = Cfg::NT, KW = Cfg::KW, BN = Cfg::BN, TH = Cfg::TH;
    constexpr int EPC = 8, BKE = 64, NW = 8, NTHR = 512;
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int TN = 2, TM = 4;                                // wave tile: 2 x 32 channels, 4 rows of 32 pixels
    constexpr int NPP = PROWS / 8;                               // 77 patch DMA pieces
    constexpr int NPW = (NPP + NW - 1) / NW;                     // 10 per wave (the last one of waves 5-7 repeats a piece)
    constexpr int GW = BN / (8 * NW);                            // 2 weight DMA pieces per wave and stage
    constexpr int BM = Cfg::BM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;                     // 2 x 4 waves; waves w and w + 4 share a SIMD
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    int t = blockIdx.x;
    const int n0 = (t % tiles_n) * BN; t /= tiles_n;
    const int x0 = (t % tiles_x) * TW; t /= tiles_x;
    const int y0 = (t % tiles_y) * TH; const int b = t / tiles_y;
    const int nch0 = (a.C0 + BKE - 1) / BKE, nch1 = (a.C1 + BKE - 1) / BKE;
    const int nchunks = nch0 + nch1;
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)a.B * H * W;
    const i32x4 rs0 = x_make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rs1 = x_make_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? npix * a.C1 * 2L : 0);
    const i32x4 rsw = x_make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + Cfg::PATCH;

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

    // --- per-lane DMA roles -------------------------------------------------------------
    const int lrow = lane >> 3, slot = lane & 7;
    // Patch pieces of this wave: wave, wave + 8, ...; piece q covers patch rows 8 q .. 8 q + 7, this lane row lrow of it. The
    // pixel of piece k + 1 lies 64 patch rows behind that of piece k: (py, px) advance by (1, 30) with one carry --
    // recomputed at every chunk (ten pieces per ~21 k cycles) instead of held in 20 registers beside 128 accumulators.
    auto issue_patch = [&](int cc) {
        const bool s1 = cc >= nch0;
        const int cbase = (s1 ? cc - nch0 : cc) * BKE, Cs = s1 ? a.C1 : a.C0;
        i32x4 qrs;
        qrs.x = s1 ? rs1.x : rs0.x; qrs.y = s1 ? rs1.y : rs0.y; qrs.z = s1 ? rs1.z : rs0.z; qrs.w = rs0.w;
        const bool tail = Cs - cbase < BKE;
        int pr = wave * 8 + lrow;
        int py = pr / PW, px = pr - py * PW;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const bool dup = wave + NW * k >= NPP;               // (wave-uniform, only k = NPW - 1): the wave's previous piece again
            if (dup) { pr -= 64; px -= 30; py -= 1; if (px < 0) { px += PW; py -= 1; } }
            const int iy = y0 + py - 1, ix = x0 + px - 1;
            const bool v = pr < Cfg::PH * PW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int pix = v ? (b * H + iy) * W + ix : (int)npix;   // padding: the first pixel beyond the tensor
            const int ch = cbase + ((slot ^ ((pr >> 1) & 7)) * EPC);
            unsigned off = (unsigned)((pix * Cs + ch) * 2);
            if (tail) off = ch < Cs ? off : OOB;
            const int piece = __builtin_amdgcn_readfirstlane(dup ? wave + NW * (k - 1) : wave + NW * k);
            x_dma16(qrs, off, lds0 + piece * 1024);
            pr += 64; px += 30; py += 1; if (px >= PW) { px -= PW; py += 1; }
        }
    };
    unsigned wpo[GW]; int wch[GW];
#pragma unroll
    for (int g = 0; g < GW; ++g) {
        const int rl = wave * (BN / NW) + g * 8 + lrow;
        const int n = n0 + rl;
        wch[g] = (slot ^ ((rl >> 1) & 7)) * EPC;
        wpo[g] = n < a.Cout ? (unsigned)((long)n * a.w_row_stride * 2L) + (unsigned)(wch[g] * 2) : X_POISON;
    }
    auto chunk_woff = [&](int c_) { const bool s1 = c_ >= nch0; return (unsigned)(((s1 ? a.C0 : 0) + (s1 ? c_ - nch0 : c_) * BKE) * 2); };
    auto chunk_room = [&](int c_) { const bool s1 = c_ >= nch0; return (s1 ? a.C1 : a.C0) - (s1 ? c_ - nch0 : c_) * BKE; };
    auto request_w = [&](unsigned soff, int room, int stage) {   // weights of one tap: GW pieces per wave
        const unsigned dst = ldsW + stage * Cfg::WSTAGE + wave * (BN / NW) * 128;
        if (FULL || room >= BKE) {
#pragma unroll
            for (int g = 0; g < GW; ++g) x_dma16(rsw, wpo[g] + soff, dst + g * 8 * 128);
        } else {
#pragma unroll
            for (int g = 0; g < GW; ++g) x_dma16(rsw, wch[g] < room ? wpo[g] + soff : X_POISON, dst + g * 8 * 128);
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    const bool second = wave >= 4;
    const unsigned w_tap_b = (unsigned)(a.w_tap_stride * 2L);
    unsigned woffA = chunk_woff(0); int roomA = chunk_room(0);
    issue_patch(0);
    request_w(woffA, roomA, 0);
    request_w(woffA + w_tap_b, roomA, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");    // patch + tap 0 landed (tap 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    if (second) __builtin_amdgcn_s_barrier();                    // one phase behind
    // weight-fragment offset of this lane inside a stage at k-step 0 (k-step s flips bits 5-6 of the swizzled slot: ^ s << 5)
    const unsigned wlane = (unsigned)(Cfg::PATCH + (wn * 64 + (lane & 31)) * 128) + (unsigned)((fh ^ fsw) << 4);
    const int prow0 = (wm * TM) * PW + (lane & 31);              // patch row of the lane's pixel in tile row wm * 4, tap (0, 0)
    unsigned stb = 0;                                            // byte offset of the current weight stage
    for (int cc = 0; cc < nchunks; ++cc) {
        const bool hasnext = cc + 1 < nchunks;
        const int kv = (FULL || roomA >= BKE) ? 4 : (roomA + 15) / 16;   // k-steps of 16 channels that hold data (a tail chunk: fewer)
        const unsigned woffB = hasnext ? chunk_woff(cc + 1) : 0u;
        const int roomB = hasnext ? chunk_room(cc + 1) : 0;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ky = tap / KW, kx = tap % KW;              // (compile-time)
            // ---- L: the tap's pixel fragments (all k-steps) and the weight fragments of k-steps 0, 1
            __builtin_amdgcn_s_setprio(1);
            int l31 = prow0;
            asm volatile("" : "+v"(l31));                        // (keeps the nine taps' addresses from being hoisted into registers)
            unsigned po[TM];
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int prow = l31 + (j + ky) * PW + kx;
                po[j] = (unsigned)(prow * 128) + (unsigned)((fh ^ ((prow >> 1) & 7)) << 4);
            }
            const unsigned wst = wlane + stb;
            uint4 fa[4][TN], fb[4][TM];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                if (s_ > 0 && s_ >= kv) break;                   // (workgroup-uniform)
                const unsigned ks = (unsigned)(s_ << 5);
                if (s_ < 2) {
#pragma unroll
                    for (int i = 0; i < TN; ++i) fa[s_][i] = *(const uint4*)(smem + (wst ^ ks) + i * 32 * 128);
                }
#pragma unroll
                for (int j = 0; j < TM; ++j) fb[s_][j] = *(const uint4*)(smem + (po[j] ^ ks));
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // weights two taps ahead (of this chunk, or the first taps of the next one) into stage + 2
                const int wt = tap + 2;
                const bool req = wt < NT || hasnext;
                const int stn = (int)(((stb >> 14) + 2) & 3);
                if (wt < NT) request_w(woffA + (unsigned)wt * w_tap_b, roomA, stn);
                else if (hasnext) request_w(woffB + (unsigned)(wt - NT) * w_tap_b, roomB, stn);
                __builtin_amdgcn_sched_barrier(0);
                // tap g + 1 has landed; only the stage requested just now may stay in flight
                if (req) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---- C: the tap's MFMAs; the weight fragments of k-steps 2, 3 are read under them
            const bool reload = tap == NT - 1 && hasnext;        // chunk boundary (see the header): the patch is dead behind the
            if (reload && second) issue_patch(cc + 1);           // barrier that closed the second half's L(8)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) x_mma(fa[0][i], fb[0][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (kv > 2) {
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[2][i] = *(const uint4*)(smem + (wst ^ (2u << 5)) + i * 32 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kv > 1) {
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[1][i], fb[1][j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kv > 3) {
#pragma unroll
                for (int i = 0; i < TN; ++i) fa[3][i] = *(const uint4*)(smem + (wst ^ (3u << 5)) + i * 32 * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kv > 2) {
                if (kv > 3) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TN) : "memory");   // k-step 2's fragments (k-step 3's may be in flight)
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[2][i], fb[2][j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kv > 3) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) x_mma(fa[3][i], fb[3][j], acc[i][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (reload) {
                // second half: its pieces were requested in front of this C(8); first half: one interval of its own for
                // them (behind the barrier that closes [first: C(8) | second: L(8)]). The barrier that closes
                // [first: reload | second: C(8)] makes the whole patch visible; the second half then idles one interval
                // so that the halves stay one phase apart.
                if (second) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (!second) { issue_patch(cc + 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
            } else {
                __builtin_amdgcn_s_barrier();
            }
            stb = (stb + Cfg::WSTAGE) & (4 * Cfg::WSTAGE - 1);
        }
        woffA = woffB; roomA = roomB;
    }
    if (!second) __builtin_amdgcn_s_barrier();                   // the second half's last compute phase
    __builtin_amdgcn_s_setprio(0);

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
}

int launch_halo16(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = Halo16Cfg;
    const bool full = a_in.C0 % 64 == 0 && a_in.C1 % 64 == 0;
    auto kern = full ? conv_halo16_kernel<true> : conv_halo16_kernel<false>;
    ConvArgs a = a_in;
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
    kern<<<dim3((unsigned)tiles), dim3(512), Cfg::SMEM, st>>>(a);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// 3 = launched, 0 = shape not suited (the caller falls back to conv_halo), < 0 = error.
// Large grids of 128-channel tiles on 16-row x 32-pixel pixel tiles: predict batches, the configs[3] train step.
int try_conv_halo16(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    static int on = -1; static long min_tiles = 768;
    if (on < 0) {
        const char* e = getenv("MPU_HALO16"); on = (e && e[0] == '0') ? 0 : 1;
        const char* m = getenv("MPU_HALO16_MIN"); if (m) min_tiles = atol(m);
    }
    if (!on || dtype != MPU_BF16 || mode != CONV3 || a.Cout <= 64 || a.Wo < 32 || (a.Ho & 15) || a.head_w) return 0;
    const long tiles = (long)a.B * (a.Ho / 16) * cdiv(a.Wo, 32) * cdiv(a.Cout, 128);
    if (tiles < min_tiles) return 0;
    const int rc = launch_halo16(a, st);
    return rc ? rc : 3;
}

}  // namespace mpu
