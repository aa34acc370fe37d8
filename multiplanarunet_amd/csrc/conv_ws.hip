// conv_ws_kernel: 3x3 SAME convolution for the full-resolution layers (Cin <= 64, Cout <= 64, bf16)
// with the WHOLE weight tensor stationary in registers and persistent workgroups.
//
// At the first U-Net level the GEMM is skinny (N = 64, K = 576) and HBM-bound (33 MB in, 33 MB out
// at B=16 128x128): the tiled kernels of conv_halo.hip / conv_glds.hip re-fetch the 74 KB weight
// tensor into LDS for every 256-pixel tile (75 MB of L2 -> LDS traffic per launch, more than the
// activations) and are bound by that fill plus the LDS read rate (weights AND pixels come from LDS:
// 128 B/clk at full MFMA rate). Here
//   * two workgroups per CU loop over pixel tiles (persistent, round-robin);
//   * each wave owns 32 output channels and keeps its A operands - 9 taps x 4 k-steps = 36
//     fragments = 144 VGPRs - in registers for the whole kernel (2 waves per SIMD x 256 registers);
//   * LDS only holds pixel patches ((TH+2) x 34 halo rows of 128 B, double buffered, filled by
//     LDS-DMA one tile ahead) and the output staging tile: the L2 -> LDS traffic is the activations
//     only and half of the MFMA operand traffic never touches LDS.
// Same operand conventions and epilogue (bias, ReLU, folded BN affine, ReLU-mask for data
// gradients, coalesced 16-byte stores) as conv_halo.hip.
#include <stdlib.h>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

__device__ __forceinline__ i32x4 ws_make_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa; r.y = (int)((unsigned)(pa >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void ws_dma16(const i32x4& rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_addr), "s"(rsrc) : "memory");
}

template <int TH>
struct WsCfg {
    static constexpr int TW = 32, PW = TW + 2, PH = TH + 2;
    static constexpr int PROWS = (PH * PW + 7) / 8 * 8;
    static constexpr int PATCH = PROWS * 128;
    static constexpr int BM = TH * TW;
    static constexpr int WPX = BM / 2;                           // pixels per wave (2 x 2 waves: channels x rows)
    static constexpr int OROW = 32 * 2 + 16;                     // per-wave staging row: 32 channels + pad
    static constexpr int WBYTES = 9 * 64 * 128;
    static constexpr int HEADW = 64 * 8 * 4;                     // head weights [64 channels][8 classes] f32 (fused head)
    static constexpr int STAGE_MIN = 4 * WPX * OROW + 3 * 64 * 4 + HEADW;
    // LDS: [per-wave staging tiles + bias tables][patch 1][patch 0]; the whole weight tensor is staged
    // through the same bytes once, before the first patch is requested
    static constexpr int STAGE0 = (STAGE_MIN + 1023) / 1024 * 1024;
    static constexpr int STAGE = STAGE0 + 2 * PATCH >= WBYTES ? STAGE0 : WBYTES - 2 * PATCH;
    static constexpr int SMEM = STAGE + 2 * PATCH;
    static_assert(STAGE % 1024 == 0 && SMEM >= WBYTES, "LDS plan");
};

template <int TH, bool HEAD>
__global__ __launch_bounds__(256, 2) void conv_ws_kernel(ConvArgs a, int ntiles, unsigned magic_x, unsigned magic_y,
                                                         unsigned magic_g, int xcd_chunk) {
    using Cfg = WsCfg<TH>;
    constexpr int TW = Cfg::TW, PW = Cfg::PW, PROWS = Cfg::PROWS;
    constexpr int TM = TH / 2;                                   // pixel rows per wave (2 x 2 waves: channels x rows)
    constexpr int NPP = PROWS / 8, NPW = (NPP + 3) / 4;
    constexpr int OROW = Cfg::OROW;
    constexpr unsigned OOB = 0xfffffff0u;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int H = a.Ho, W = a.Wo;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long npix = (long)a.B * H * W;
    const i32x4 rs0 = ws_make_rsrc(a.in0, npix * a.C0 * 2L);
    const i32x4 rsw = ws_make_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const int lrow = lane >> 3, slot = lane & 7;

    // The loop below walks VIRTUAL tile indices v = blockIdx.x + round * gridDim.x. With xcd_chunk (= ntiles / 8; the launcher
    // sets it when 8 divides the grid and the grid divides the tiles) the tile behind v is
    //   (blockIdx.x & 7) * xcd_chunk + round * (gridDim.x / 8) + (blockIdx.x >> 3):
    // XCD k (workgroups with id = k mod 8) owns tiles [k * xcd_chunk, (k + 1) * xcd_chunk) and the workgroups of an XCD take
    // neighbouring tiles in the same round, so the halo rows two tiles share are fetched into that XCD's L2 once (in launch
    // order vertical neighbours sit on different XCDs: 1.59 x the input bytes from the fabric at 4-row tiles).
    auto tile_coords = [&](int t, int& b, int& y0, int& x0) {      // exact multiply-high division (host-checked range)
        if (xcd_chunk) {
            const int rnd = (int)__umulhi((unsigned)t, magic_g), bid = t - rnd * (int)gridDim.x;
            t = (bid & 7) * xcd_chunk + rnd * ((int)gridDim.x >> 3) + (bid >> 3);
        }
        const int t1 = magic_x ? (int)__umulhi((unsigned)t, magic_x) : t;   // t / tiles_x (magic 0 = one tile column: 2^32 / 1 does not fit)
        x0 = (t - t1 * tiles_x) * TW;
        b = magic_y ? (int)__umulhi((unsigned)t1, magic_y) : t1;    // t1 / tiles_y
        y0 = (t1 - b * tiles_y) * TH;
    };
    // Patch DMA: source offset = (tile base, scalar) + (lane part, tile-invariant); 24-bit multiplies only
    auto issue_patch = [&](int t, int buf) {
        int b, y0, x0; tile_coords(t, b, y0, x0);
        const unsigned base = lds0 + Cfg::STAGE + (buf ? 0 : Cfg::PATCH);
        const int tbase = (((b * H + y0 - 1) * W + x0 - 1) * a.C0) * 2;     // may be negative: masked by the bounds below
        const int rowB = W * a.C0 * 2, pixB = a.C0 * 2;
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            const int piece = wave + 4 * k;
            if (piece < NPP) {                     // wave-uniform
                const int pr = piece * 8 + lrow;
                const int py = (pr * 1928) >> 16, px = pr - py * PW;          // pr / 34 for pr < 2^11
                const int ch = (slot ^ ((pr >> 1) & 7)) * 8;
                const bool v = pr < Cfg::PH * PW && (unsigned)(y0 - 1 + py) < (unsigned)H &&
                               (unsigned)(x0 - 1 + px) < (unsigned)W && ch < a.C0;
                const int off = tbase + __mul24(py, rowB) + __mul24(px, pixB) + ch * 2;
                ws_dma16(rs0, v ? (unsigned)off : OOB, base + piece * 1024);
            }
        }
    };

    // ---- prologue: weights -> LDS (9 taps x 64 rows x 128 B, swizzled) -> registers (own 32-channel half) ----
    for (int p = wave; p < 72; p += 4) {
        const int tap = p >> 3, rl = (p & 7) * 8 + lrow;
        const int ch = (slot ^ ((rl >> 1) & 7)) * 8;
        const unsigned off = (rl < a.Cout && ch < a.C0)
            ? (unsigned)(((long)rl * a.w_row_stride + (long)tap * a.w_tap_stride + ch) * 2) : OOB;
        ws_dma16(rsw, off, lds0 + p * 1024);
    }
    int tile = blockIdx.x;
    const int fsw = ((lane & 31) >> 1) & 7, fh = lane >> 5;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    uint4 wreg[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            wreg[tap][s] = *(const uint4*)(smem + tap * 8192 + (wn * 32 + (lane & 31)) * 128 + (((2 * s + fh) ^ fsw) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* stab = (float*)(smem + 4 * Cfg::WPX * OROW);         // bias; folded-BN affine (inference): scale, shift
    // fused head: the head weights of classes 0..7 as two bf16 parts (w = hi + lo, |error| <= 2^-18 |w|), rows of 64
    // channels: hwt[part][class][channel] -- the A operand of a 32x32x16 MFMA whose rows 8..31 are zero
    bf16_t* hwt = (bf16_t*)(stab + 3 * 64);
    if (HEAD) {
        for (int i = tid; i < 8 * 64; i += 256) {
            const int k = i >> 6, c = i & 63;
            const float w = (c < a.Cout && k < a.head_k) ? a.head_w[(long)c * a.head_ldw + k] : 0.f;
            const bf16_t hi = f32_to_bf16(w);
            hwt[i] = hi;
            hwt[8 * 64 + i] = f32_to_bf16(w - bf16_to_f32(hi));
        }
    }
    if (tid < 64) {
        const bool nv = tid < a.Cout;
        stab[128 + tid] = (a.bias && nv) ? a.bias[tid] : 0.f;
        if (a.post_scale) {
            stab[tid] = nv ? a.post_scale[tid] : 1.f;
            stab[64 + tid] = nv ? a.post_shift[tid] : 0.f;
        }
    }
    issue_patch(tile, 0);
    if (tile + (int)gridDim.x < ntiles) issue_patch(tile + gridDim.x, 1);
    // patch 0 has landed (patch 1 may still be in flight: NPW or NPW-1 DMAs per wave, wait for all but those)
    if (tile + (int)gridDim.x < ntiles) {
        if (wave + 4 * (NPW - 1) < NPP) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW - 1) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- persistent loop over pixel tiles ------------------------------------------------------------
    unsigned char* wstage = smem + wave * (Cfg::WPX * OROW);
    unsigned char* stage_w = wstage + (lane & 31) * OROW + fh * 8;            // acc fragment -> staging row of its pixel
    const unsigned char* stage_r = wstage + (lane >> 2) * OROW + (lane & 3) * 16;
    const float* bias_l = stab + 128 + wn * 32 + 4 * fh;
    const bool n_ok = wn * 32 + (lane & 3) * 8 < a.Cout;
    const int out_l = ((wm * TM * W + (lane >> 2)) * a.Cout + wn * 32 + (lane & 3) * 8) * 2;
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npix * a.Cout * 2L), 0x00020000);
    // second per-pixel input of the epilogue, same layout as the output: the ReLU mask of a data-gradient launch, or (round 6) the
    // pre-BatchNorm activation x of a launch that also sums BatchNorm-backward terms (bn_x; the two never come together)
    const bool bnx = a.stats && a.bn_x && !a.mask;
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : (bnx ? a.bn_x : a.out)), 0,
                                                                          (int)(npix * a.Cout * 2L), 0x00020000);
    float stat_acc = 0.f;     // fused BN statistics: lane L of a wave owns value L = (channel group, channel, sum | sum^2)
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int poff = Cfg::STAGE + (buf ? 0 : Cfg::PATCH);
        f32x16 acc[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // Hand-pipelined: the B fragments of step k+1 (step = tap*4 + k-step) are requested before the MFMAs
        // of step k; sched_barrier keeps the compiler from serialising load -> wait -> MFMA or hoisting the
        // 36 address computations (which spills). Per (tap, pixel row) one address A; the k-step only flips
        // bits 5-6 of the swizzled slot: addr = A ^ (s << 5).
        int rowbase = wm * TM * PW + (lane & 31);
        asm volatile("" : "+v"(rowbase));                        // recompute the 18 addresses per tile, do not hoist (spills)
        auto row_addr = [&](int tap, int j) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int prow = rowbase + (j + ky) * PW + kx;
            return poff + prow * 128 + ((fh ^ ((prow >> 1) & 7)) << 4);
        };
        constexpr int NB = 3, D = NB - 1;                        // B-fragment ring: requested D steps ahead of use
        int A[TM];
        uint4 bf[NB][TM];
        auto request = [&](int st) {                             // st = tap*4 + k-step (compile-time after unrolling)
            const int tp = st >> 2, s1 = st & 3;
            if (s1 == 0) {
#pragma unroll
                for (int j = 0; j < TM; ++j) A[j] = row_addr(tp, j);
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[st % NB][j] = *(const uint4*)(smem + (A[j] ^ (s1 << 5)));
        };
#pragma unroll
        for (int st = 0; st < D; ++st) request(st);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int tap = step >> 2, s = step & 3;
            if (step + D < 36) request(step + D);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, wreg[tap][s]),
                                                                 __builtin_bit_cast(s16x8, bf[step % NB][j]), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The next tile's patch (requested one tile ago) has landed and the previous tile's stores are long
        // done; after the barrier every wave has also finished reading this tile's patch buffer.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + 2 * (int)gridDim.x < ntiles) issue_patch(tile + 2 * gridDim.x, buf);
        // ---- epilogue, wave-private (no workgroup barrier): registers -> own staging rows -> 64-byte runs ----
        {
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *(const float4*)(bias_l + q * 8);
            const float lo = a.relu ? 0.f : -__builtin_inff();          // ReLU as a clamp: no branch, no canonicalisation
#pragma unroll
            for (int j = 0; j < TM; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4] = {acc[j][4 * q] + bq[q].x, acc[j][4 * q + 1] + bq[q].y, acc[j][4 * q + 2] + bq[q].z,
                                  acc[j][4 * q + 3] + bq[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
                    if (a.post_scale) {
                        const float4 sq = *(const float4*)(bias_l - 128 + q * 8), hq = *(const float4*)(bias_l - 64 + q * 8);
                        v[0] = v[0] * sq.x + hq.x; v[1] = v[1] * sq.y + hq.y;
                        v[2] = v[2] * sq.z + hq.z; v[3] = v[3] * sq.w + hq.w;
                    }
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(v[0], v[1]);
                    pk.y = f32x2_to_bf16x2(v[2], v[3]);
                    *(uint2*)(stage_w + j * 32 * OROW + q * 16) = pk;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (HEAD) {
            // fused 1x1 head as 4 * TM MFMAs: logits[class][pixel] = sum_c w[class][c] * y[pixel][c] over this wave's 32
            // channels (two 16-channel k-steps, two weight parts), B fragments = the staged (rounded) pixels. The
            // accumulator of lane l holds classes 4 * (l >> 5) + 0..3 of pixel l & 31; the output tensor is not stored.
            int b, y0, x0; tile_coords(tile, b, y0, x0);
            const bool arow = (lane & 31) < 8;
            uint4 ah[2], al[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16_t* src = hwt + (lane & 7) * 64 + wn * 32 + s2 * 16 + 8 * fh;
                ah[s2] = *(const uint4*)src; al[s2] = *(const uint4*)(src + 8 * 64);
                if (!arow) { ah[s2] = make_uint4(0, 0, 0, 0); al[s2] = make_uint4(0, 0, 0, 0); }
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                f32x16 hacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const uint4 bfr = *(const uint4*)(wstage + (j * 32 + (lane & 31)) * OROW + (2 * s2 + fh) * 16);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, ah[s2]), __builtin_bit_cast(s16x8, bfr), hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, al[s2]), __builtin_bit_cast(s16x8, bfr), hacc, 0, 0, 0);
                }
                const int py = y0 + wm * TM + j, px = x0 + (lane & 31);
                if (py < H && px < W) {
                    float* hp = a.head_partial + ((long)wn * npix + ((long)b * H + py) * W + px) * a.head_k + 4 * fh;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (4 * fh + e < a.head_k) hp[e] = hacc[e];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // before the next tile's staging writes
        } else {
            int b, y0, x0; tile_coords(tile, b, y0, x0);
            const int obase = ((b * H + y0) * W + x0) * a.Cout * 2;          // scalar part of the output offset
            const bool xok0 = x0 + (lane >> 2) < W, xok1 = x0 + 16 + (lane >> 2) < W;
            float tsum[8], tsq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { tsum[e] = 0.f; tsq[e] = 0.f; }
            u32x4 mkv[Cfg::WPX / 16];                                // ReLU masks (data-gradient launches): all requested
#pragma unroll                                                       // up front, not one round trip per pass
            for (int it = 0; it < Cfg::WPX / 16; ++it) {
                const int j = it >> 1;
                const bool ok = (a.mask || bnx) && n_ok && ((it & 1) ? xok1 : xok0) && (y0 + wm * TM + j < H);
                mkv[it] = __builtin_amdgcn_raw_buffer_load_b128(
                    rsm, ok ? (unsigned)(obase + out_l + (j * W + (it & 1) * 16) * a.Cout * 2) : OOB, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < Cfg::WPX / 16; ++it) {             // 4 lanes per pixel: 64 contiguous bytes
                const int j = it >> 1;                               // pixel row of this wave, column (it&1)*16 + lane>>2
                const bool ok = n_ok && ((it & 1) ? xok1 : xok0) && (y0 + wm * TM + j < H);
                u32x4 val = *(const u32x4*)(stage_r + it * 16 * OROW);
                // (no scalar offset operand: it would be added after the range check and wrap the out-of-range marker)
                const unsigned off = ok ? (unsigned)(obase + out_l + (j * W + (it & 1) * 16) * a.Cout * 2) : OOB;
                if (a.mask) {
                    const u32x4 mk = mkv[it];
                    auto keep = [](uint32_t mw, uint32_t vw) {
                        const uint32_t lo = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                        const uint32_t hi = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                        return vw & (lo | hi);
                    };
                    val.x = keep(mk.x, val.x); val.y = keep(mk.y, val.y);
                    val.z = keep(mk.z, val.z); val.w = keep(mk.w, val.w);
                }
                __builtin_amdgcn_raw_buffer_store_b128(val, rso, off, 0, 0);
                if (a.stats && ok) {
                    const uint32_t wv[4] = {val.x, val.y, val.z, val.w};
                    const u32x4 xk = mkv[it];
                    const uint32_t xv[4] = {xk.x, xk.y, xk.z, xk.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __uint_as_float(wv[e] << 16), hi = __uint_as_float(wv[e] & 0xffff0000u);
                        // forward: (sum y, sum y^2); BatchNorm backward: (sum dn, sum dn * x) -- x still raw: sum dn * xhat =
                        // invstd * (sum dn * x - mean * sum dn) is formed once per workgroup at the end
                        const float lo2 = bnx ? __uint_as_float(xv[e] << 16) : lo, hi2 = bnx ? __uint_as_float(xv[e] & 0xffff0000u) : hi;
                        tsum[2 * e] += lo; tsq[2 * e] += lo * lo2; tsum[2 * e + 1] += hi; tsq[2 * e + 1] += hi * hi2;
                    }
                }
            }
            if (a.pooled) {    // second output: 2x2 max pooling of this wave's TM x 32 pixels x 32 channels (staging rows)
                static_assert(TM % 2 == 0, "pooling windows stay inside a wave's rows");
                const int Hp = H >> 1, Wp = W >> 1;
                const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(a.pooled, 0, (int)((npix >> 2) * a.Cout * 2L), 0x00020000);
#pragma unroll
                for (int jp = 0; jp < TM / 2; ++jp) {                // lane -> pooled pixel (lane >> 2) of 16, 16-byte piece (lane & 3)
                    const unsigned char* s0 = wstage + ((2 * jp) * 32 + 2 * (lane >> 2)) * OROW + (lane & 3) * 16;
                    const u32x4 q0 = *(const u32x4*)s0, q1 = *(const u32x4*)(s0 + OROW), q2 = *(const u32x4*)(s0 + 32 * OROW),
                                q3 = *(const u32x4*)(s0 + 32 * OROW + OROW);
                    u32x4 m;
                    m.x = bf16x2_max(bf16x2_max(q0.x, q1.x), bf16x2_max(q2.x, q3.x));
                    m.y = bf16x2_max(bf16x2_max(q0.y, q1.y), bf16x2_max(q2.y, q3.y));
                    m.z = bf16x2_max(bf16x2_max(q0.z, q1.z), bf16x2_max(q2.z, q3.z));
                    m.w = bf16x2_max(bf16x2_max(q0.w, q1.w), bf16x2_max(q2.w, q3.w));
                    const int gy = ((y0 + wm * TM) >> 1) + jp, gx = (x0 >> 1) + (lane >> 2);
                    const bool okp = n_ok && gy < Hp && gx < Wp;
                    const unsigned offp = okp ? (unsigned)((((b * Hp + gy) * Wp + gx) * a.Cout + wn * 32 + (lane & 3) * 8) * 2) : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(m, rsp, offp, 0, 0);
                }
                if (!a.stats) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tile's staging writes
            }
            if (a.stats) {     // transpose-reduce over the 16 pixel-lanes through the wave's own staging rows (now free)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float* tb = (float*)wstage;                          // [16 pixel-lanes][64 values]
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    tb[(lane >> 2) * 64 + (lane & 3) * 16 + 2 * e] = tsum[e];
                    tb[(lane >> 2) * 64 + (lane & 3) * 16 + 2 * e + 1] = tsq[e];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float s = 0.f;
#pragma unroll
                for (int pl = 0; pl < 16; ++pl) s += tb[pl * 64 + lane];
                stat_acc += s;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tile's staging writes
            }
        }
    }
    if (a.stats) {             // one partial row per workgroup: the two pixel-row waves of a channel half are combined
        __syncthreads();
        float* cb = (float*)smem;                                    // [4 waves][64]
        cb[wave * 64 + lane] = stat_acc;
        __syncthreads();
        if (wm == 0) {
            float v = cb[wave * 64 + lane] + cb[(wave + 2) * 64 + lane];
            const int ch = wn * 32 + (lane >> 4) * 8 + ((lane & 15) >> 1), st2 = lane & 1;
            if (bnx) {                                               // odd lanes: sum dn * x -> sum dn * xhat (even lane = sum dn)
                const float sdn = __shfl_xor(v, 1, 64);
                const int chc = ch < a.Cout ? ch : 0;
                if (st2) v = a.bn_invstd[chc] * (v - a.bn_mean[chc] * sdn);
            }
            if (ch < a.Cout) stats_emit(a, st2, ch, gridDim.x, blockIdx.x, v);      // [2][Cout][workgroups], or the accumulator
        }
    }
}

template <int TH, bool HEAD>
int launch_ws(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = WsCfg<TH>;
    auto kern = conv_ws_kernel<TH, HEAD>;
    ConvArgs a = a_in;
    if (a.w_elems <= 0) a.w_elems = 8 * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;                      // (per instantiation: this function is a template)
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const int ncu = device_cu_count();
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M * a.C0 * 2L >= (1L << 31) || M * a.Cout * 2L >= (1L << 31) || a.w_elems * 2L >= (1L << 31))
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    const long tiles = (long)a.B * cdiv(a.Ho, TH) * cdiv(a.Wo, Cfg::TW);
    const int grid = (int)(tiles < 2L * ncu ? tiles : 2L * ncu);
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * 9 * a.C0, st);
    const long tx = cdiv(a.Wo, Cfg::TW), ty = cdiv(a.Ho, TH);
    if (tiles * (tx > ty ? tx : ty) >= (1L << 32)) return fail(MPU_EUNSUPPORTED, "%s", "conv: too many tiles");
    const unsigned mx = (unsigned)(((1UL << 32) + tx - 1) / tx), my = (unsigned)(((1UL << 32) + ty - 1) / ty);
    if (a.stats && a.stats_rows && !(a.bn_x && (a.mask || HEAD || !a.bn_mean || !a.bn_invstd)) && (long)grid * 2 * a.Cout <= a.stats_cap)
        *a.stats_rows = grid;                                    // (forward statistics, or the BatchNorm-backward sums of a data gradient)
    else a.stats = nullptr;
    if (a.pooled && a.pooled_done && !a.mask && !(a.Ho & 1) && !(a.Wo & 1)) *a.pooled_done = 1;
    else a.pooled = nullptr;
    if (HEAD) *a.head_done = 1;
    else a.head_partial = nullptr;
    // XCD-contiguous tile ranges (MPU_XCD_TILES): only the regular case -- every workgroup runs the same number of rounds
    // (and tiles * grid < 2^32: the multiply-high division by the grid is exact)
    const bool xcd = env(ENV_XCD_TILES) != 0 && !(grid & 7) && grid >= 8 && tiles % grid == 0 && tiles * (long)grid < (1L << 32);
    const unsigned mg = xcd ? (unsigned)(((1UL << 32) + grid - 1) / grid) : 0u;
    launch_k(kern, dim3((unsigned)grid), dim3(256), Cfg::SMEM, st, a, (int)tiles, mx, my, mg, xcd ? (int)(tiles / 8) : 0);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// 1 = launched, 0 = shape not suited (the caller falls back to the tiled kernels), < 0 = error
int try_conv_ws(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    const bool on = env(ENV_CONV_WS) != 0;
    if (!on || dtype != MPU_BF16 || mode != CONV3 || a.C1 != 0 || a.in1 || a.C0 > 64 || a.Cout > 64) return 0;
    if (a.ksplit > 1) return 0;
    const long tiles4 = (long)a.B * cdiv(a.Ho, 4) * cdiv(a.Wo, 32);
    if (a.Wo < 32 || a.Ho < 4 || tiles4 < 1024) return 0;    // needs >= 2 tiles per workgroup to amortise the weight load
    const bool head = a.head_partial && a.head_done && a.head_w && a.head_k >= 1 && a.head_k <= 8 && !a.mask && !a.stats && !a.pooled;
    const int rc = head ? launch_ws<4, true>(a, st) : launch_ws<4, false>(a, st);
    return rc ? rc : 1;
}

}  // namespace mpu
