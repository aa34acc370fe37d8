// conv_deepk_kernel (bf16, CONV3, round 5): the deep levels' 3x3 layers WITHOUT a split of K over workgroups.
//
// Why. On 16 x 16 maps at configs[1] sizes (4096 pixels x 512 filters x K = 2304..9216) conv_pipe / conv_deep cut 64 tiles
// of 256 pixels x 128 filters and split K four ways to fill 256 CUs: 33.5 MB of fp32 partial sums per layer are written by
// the convolution (at the ~3.4 TB/s this part sustains for bulk writes: 9 us when every workgroup stores at the same time,
// gpurun R5f / R5g: conv_deep 28.1 us with its stores, 19.7 without) and read back by splitk_finish (7.5 us) -- more than
// the 9-13 us the matrix work takes. Here the tile is 128 pixels x 64 filters (256 tiles: one per CU), every workgroup runs
// the WHOLE reduction, and K is split over the eight WAVES of the workgroup instead: every wave owns the full tile (2 x 4
// MFMA blocks, 128 accumulator registers) and takes one of the eight (item, k-step) units of an interval. The eight partial
// tiles are summed through LDS at the end in a fixed order (deterministic), each wave finishing one 32 x 32 block, and the
// epilogue (bias, ReLU, ReLU mask, bf16) runs in the kernel: no partial sums in HBM, no second launch.
//
// Price: a 64-filter tile re-reads the weights for 32 pixel tiles instead of 16 (2x the L2 -> LDS weight stream per MFMA of
// conv_deep: 16 KB per 8 MFMAs and wave), so the request rings are as deep as LDS allows: six 16-KB weight stages
// (requests six intervals ahead) and FOUR 16-KB half-patch buffers (a half patch is requested six intervals before its
// first read).
//
// Item stream (conv_deep's): items = (32-channel half chunk, tap), chunk-major; an INTERVAL is four items (8 MFMAs per
// wave), nine intervals = 36 items = one PAIR of 64-channel chunks = four half patches H0..H3 in buffers 0..3. H_h is read
// in intervals floor(9h / 4) .. floor((9h + 8) / 4) = {0-2, 2-4, 4-6, 6-8}; its buffer is re-requested for the next pair
// after the barrier of interval 2h + 2.
// One interval k of a wave:  wait lgkmcnt(0) [its operands of interval k sit in registers] - counted vmcnt - barrier B_k -
// requests R_k - 8 MFMAs of interval k with the 6 fragment reads of interval k + 1 between them.
// B_k guarantees (i) every wave has read interval k's operands: its weight stage is free; (ii) what interval k + 1 reads
// has landed. R_k = W(k + 6) [2 pieces per wave] + P_h(next pair) [2 pieces] if k % 9 == 2h + 2. In-order DMA queue: B_k
// needs R_{k-5} and older, so |R_{k-4}| + .. + |R_{k-1}| requests may stay in flight (dk_allow, compile-time).
// LDS rows are 64 bytes with conv_deep's swizzle and pixel permutation (bank rule in conv_glds.hip): checked for this
// tile geometry by the same exhaustive model (tools/round5/model_conv_deepk.py).
#include <stdlib.h>
#include <type_traits>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;

namespace {

__device__ __forceinline__ i32x4 dk_rsrc(const void* p, long bytes) {
    const unsigned long long pa = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)pa; r.y = (int)((unsigned)(pa >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
template <int IMM>
__device__ __forceinline__ void dk_dma(const i32x4& rsrc, unsigned voff, unsigned lds_base) {
    asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds"
                 :: "v"(voff), "s"(lds_base), "s"(rsrc), "n"(IMM) : "memory", "scc");
}
__device__ __forceinline__ int dk_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
constexpr unsigned DK_POISON = 0x80001000u;                      // + any in-range byte offset (< 2 GiB - 8 KiB) stays >= num_records

template <int W_>
struct DkCfg {
    static constexpr int W = W_, H = W_, HW = W * W, BN = 64, BM = 128;
    static constexpr int PW = W == 16 ? 20 : 12;                  // patch pitch = 0 (mod 4): the bank rule
    static constexpr int TROWS = W == 16 ? 8 : 8;                 // image rows of a tile per image (16: half an image; 8: two images)
    static constexpr int IPT = W == 16 ? 1 : 2, IMG = (TROWS + 2) * PW;          // 200 / 120 patch rows per image
    static constexpr int PROWS = 256, PBUF = PROWS * 64, NB = 4;  // 16 DMA pieces per half patch = 2 per wave
    static_assert(IPT * IMG <= PROWS, "patch rows");
    static constexpr int WITEM = BN * 64, WSTAGE = 4 * WITEM, NWS = 6;
    static constexpr int MAIN = NB * PBUF + NWS * WSTAGE;
    static constexpr int RED = 8 * 4 * 4096;                      // reduction scratch: 8 waves x 4 blocks x 4 KB
    static constexpr int SMEM = MAIN > RED ? MAIN : RED;
};
static_assert(DkCfg<16>::SMEM <= 160 * 1024, "LDS");

template <int W_>
__device__ __forceinline__ int dk_pix_in_block(int l31) {        // = deep_pix_in_block (conv_glds.hip)
    const int q = l31 >> 2, t = l31 & 3;
    const int g = (0x96 >> q) & 1;
    const int k = (q >> 1) * 4 + t;
    if (W_ == 16) return (k >> 3) * 16 + ((k >> 2) & 1) * 8 + (k & 3) + 4 * g;
    return (k >> 2) * 8 + (k & 3) + 4 * g;
}
constexpr int dk_rsize(int j) { const int m = ((j % 9) + 9) % 9; return 2 + ((m == 2 || m == 4 || m == 6 || m == 8) ? 2 : 0); }
constexpr int dk_allow(int k) { return dk_rsize(k - 4) + dk_rsize(k - 3) + dk_rsize(k - 2) + dk_rsize(k - 1); }

// EPI: 0 = bias / ReLU / ReLU mask, bf16 store
template <int W_, bool STAMP>
__global__ __launch_bounds__(512, 2) void conv_deepk_kernel(ConvArgs a, int tiles_m, int tiles_n) {
    using Cfg = DkCfg<W_>;
    constexpr int BN = Cfg::BN, BM = Cfg::BM, HW = Cfg::HW, PW = Cfg::PW, IMG = Cfg::IMG, PBUF = Cfg::PBUF;
    constexpr int WITEM = Cfg::WITEM, WSTAGE = Cfg::WSTAGE, NWS = Cfg::NWS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int up = wave >> 1, uk = wave & 1;                     // the wave's unit of an interval: item slot 0..3, k-step 0..1
    const int logical = dk_xcd_remap(blockIdx.x, gridDim.x);     // the pixel tiles of one filter tile share an XCD (its L2 holds the weights)
    const int mt = logical % tiles_m, nt = logical / tiles_m;
    const int n0 = nt * BN, m0 = mt * BM;
    const int nch0 = a.C0 >> 6, nchunks = nch0 + (a.C1 >> 6), npairs = nchunks >> 1;
    const int M = a.B * HW;
    const i32x4 rs0 = dk_rsrc(a.in0, (long)M * a.C0 * 2L);
    const i32x4 rs1 = dk_rsrc(a.in1 ? a.in1 : a.in0, a.in1 ? (long)M * a.C1 * 2L : 0);
    const i32x4 rsw = dk_rsrc(a.w, a.w_elems * 2L);
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    const unsigned ldsW = lds0 + Cfg::NB * PBUF;
    const unsigned w_tap_b = (unsigned)(a.w_tap_stride * 2L);
    unsigned long long* stamps = (STAMP && a.dbg_buf && (blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 16 && (tid == 0 || tid == 448))
                                     ? a.dbg_buf + ((blockIdx.x >> 3) + (tid ? 16 : 0)) * 16 : nullptr;
    if (STAMP && stamps) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[6] = (unsigned long long)(9 * npairs); }

    // ---- per-lane DMA roles (a piece = 16 rows of 64 bytes: row = lane / 4, 16-byte slot = lane % 4) --------------------
    const int drow = lane >> 2, dslot = lane & 3;
    unsigned wpo[2];                                             // weights: rows (uk * 2 + g) * 16 .. + 15 of the wave's OWN item
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int rl = (uk * 2 + g) * 16 + drow;
        wpo[g] = (unsigned)((n0 + rl) * a.w_row_stride * 2) + (unsigned)((dslot ^ ((rl >> 2) & 3)) * 16);
    }
    int ppix[2]; unsigned pch[2];                                // patch: pieces wave and wave + 8 of a half patch
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pr = (wave + 8 * k) * 16 + drow;
        const int img = pr / IMG, rem = pr - img * IMG;
        const int py = rem / PW, px = rem - py * PW;
        int pix; bool v;
        if (W_ == 16) {                                          // tile = image m0 / 256, rows (m0 / 128 & 1) * 8 .. + 7
            const int iy = ((m0 >> 7) & 1) * 8 + py - 1;
            v = img == 0 && px >= 1 && px <= 16 && (unsigned)iy < 16u;
            pix = (m0 >> 8) * 256 + iy * 16 + (px - 1);
        } else {                                                 // tile = images m0 / 64 and m0 / 64 + 1
            v = img < 2 && py >= 1 && py <= 8 && px >= 1 && px <= 8;
            pix = m0 + img * 64 + (py - 1) * 8 + (px - 1);
        }
        ppix[k] = v ? pix : M;                                   // padding: the first pixel BEYOND the tensor
        pch[k] = (unsigned)((dslot ^ ((pr >> 2) & 3)) * 16);
    }
    const unsigned sbase_p = lds0 + wave * 1024;
    const unsigned sbase_w = ldsW + up * WITEM + uk * 2048;      // + stage offset + g * 1024

    // scalars of a 64-channel chunk (POISON in the byte offsets of a chunk past the end: SCALAR selects, see conv_deep)
    struct Chunk { i32x4 rs; int pitch2; unsigned cb2, wcol2; };
    auto chunk_of = [&](int c) {
        Chunk q;
        const bool valid = c < nchunks;
        const bool s1 = c >= nch0;
        const int cb = ((s1 ? c - nch0 : c) << 6);
        q.rs.x = s1 ? rs1.x : rs0.x; q.rs.y = s1 ? rs1.y : rs0.y; q.rs.z = s1 ? rs1.z : rs0.z; q.rs.w = rs0.w;
        q.pitch2 = (s1 ? a.C1 : a.C0) * 2;
        q.cb2 = valid ? (unsigned)(cb * 2) : DK_POISON;
        q.wcol2 = valid ? (unsigned)(((s1 ? a.C0 : 0) + cb) * 2) : DK_POISON;
        return q;
    };
    // half patch H4 (0..3) of the pair whose chunks are (Q0, Q1): the wave's two pieces into buffer H4
#define DK_PATCH(Q0, Q1, H4)                                                                                         \
    do {                                                                                                            \
        const Chunk& q_ = ((H4) >> 1) ? (Q1) : (Q0);                                                                \
        const unsigned o0_ = (unsigned)(ppix[0] * q_.pitch2) + q_.cb2 + (unsigned)(((H4) & 1) * 64) + pch[0];       \
        const unsigned o1_ = (unsigned)(ppix[1] * q_.pitch2) + q_.cb2 + (unsigned)(((H4) & 1) * 64) + pch[1];       \
        dk_dma<(H4) * PBUF>(q_.rs, o0_, sbase_p); dk_dma<(H4) * PBUF + 8192>(q_.rs, o1_, sbase_p);                   \
    } while (0)
    // the wave's two weight pieces of ITS item of interval IV_ (of the pair (Q0, Q1)) into the stage at ring offset STB_
#define DK_WEIGHTS(Q0, Q1, IV_, STB_)                                                                                \
    do {                                                                                                            \
        const int idx_ = 4 * (IV_) + up, h4_ = (idx_ * 57) >> 9, tap_ = idx_ - 9 * h4_;                             \
        const unsigned so_ = (unsigned)tap_ * w_tap_b + ((h4_ >> 1) ? (Q1).wcol2 : (Q0).wcol2) + (unsigned)((h4_ & 1) * 64); \
        dk_dma<0>(rsw, wpo[0] + so_, sbase_w + (STB_)); dk_dma<1024>(rsw, wpo[1] + so_, sbase_w + (STB_));            \
    } while (0)

    // ---- prologue = slots -7 .. -2 of the request rule (R_{-1} follows the first barrier) -------------------------------
    Chunk c0 = chunk_of(0), c1 = chunk_of(1), d0 = chunk_of(2), d1 = chunk_of(3);
    DK_PATCH(c0, c1, 0);
    DK_WEIGHTS(c0, c1, 0, 0 * WSTAGE);
    DK_WEIGHTS(c0, c1, 1, 1 * WSTAGE); DK_PATCH(c0, c1, 1);
    DK_WEIGHTS(c0, c1, 2, 2 * WSTAGE);
    DK_WEIGHTS(c0, c1, 3, 3 * WSTAGE); DK_PATCH(c0, c1, 2);
    DK_WEIGHTS(c0, c1, 4, 4 * WSTAGE);

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment addresses ---------------------------------------------------------------------------------------
    const int fh = lane >> 5, l31 = lane & 31;
    const int kslot = 2 * uk + fh;                               // 16-byte slot (before the swizzle) of the wave's k-step
    const unsigned wlane = ldsW + (unsigned)(up * WITEM) + (unsigned)(l31 * 64) + (unsigned)((kslot ^ ((l31 >> 2) & 3)) << 4);
    const int pib = dk_pix_in_block<W_>(l31);
    int brow[4];                                                 // patch row of the lane's pixel of block j at tap (0, 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ml = j * 32 + pib;
        if (W_ == 16) brow[j] = (ml >> 4) * PW + (ml & 15);
        else brow[j] = (ml >> 6) * IMG + ((ml >> 3) & 7) * PW + (ml & 7);
    }
    typedef unsigned int dk_u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const dk_u32x4* lds_u4;
#define DK_LD(DST, ADDR, IMM) DST = *(lds_u4)(uintptr_t)((ADDR) + (IMM))
#define DK_SB() __builtin_amdgcn_sched_barrier(0)
#define DK_MM(FA, FB, I, J)                                                                                        \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, FA[I]), __builtin_bit_cast(s16x8, FB[J]), acc[I][J], 0, 0, 0)
    dk_u32x4 fa0[2], fb0[4], fa1[2], fb1[4];                     // two operand sets (intervals alternate)
    // address of the lane's pixel fragment of block J_ for the item of interval IV_
    auto pix_addr = [&](int iv, int j) -> unsigned {
        const int idx = 4 * iv + up, h4 = (idx * 57) >> 9, tap = idx - 9 * h4;
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        const int prow = brow[j] + ky * PW + kx;
        return lds0 + (unsigned)(h4 * PBUF) + (unsigned)(prow * 64) + (unsigned)((kslot ^ ((prow >> 2) & 3)) << 4);
    };

    unsigned stb = 0;                                            // ring offset of the CURRENT interval's stage
    bool first_pair = true;
    // One interval (see the header). Operands of interval IV sit in (FA, FB); those of IV + 1 are read into (GA, GB).
    auto interval = [&](auto ivc, dk_u32x4 (&FA)[2], dk_u32x4 (&FB)[4], dk_u32x4 (&GA)[2], dk_u32x4 (&GB)[4]) {
        constexpr int IV = decltype(ivc)::value;
        if (STAMP && stamps && IV == 4 && first_pair) stamps[8] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (STAMP && stamps && IV == 4 && first_pair) stamps[9] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dk_allow(IV)) : "memory");
        if (STAMP && stamps && IV == 4 && first_pair) stamps[10] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        if (STAMP && stamps && IV == 4 && first_pair) stamps[11] = __builtin_amdgcn_s_memtime();
        if (STAMP && stamps && IV == 5 && first_pair) stamps[12] = __builtin_amdgcn_s_memtime();
        if (STAMP && stamps && IV == 3 && first_pair) stamps[7] = __builtin_amdgcn_s_memtime();
        DK_SB();
        const unsigned freed = stb;                              // this interval's stage: every wave has read it
        stb += WSTAGE; if (stb == (unsigned)(NWS * WSTAGE)) stb = 0;
        constexpr int RIV = (IV + 6) % 9, NIV = (IV + 1) % 9;
        const unsigned wa = wlane + stb;                         // the next interval's weight fragments
        // (the six reads of the next interval first, two per MFMA: the last one is five MFMAs old at the next lgkmcnt(0))
        DK_MM(FA, FB, 0, 0); DK_SB(); DK_LD(GA[0], wa, 0); DK_LD(GB[0], pix_addr(NIV, 0), 0); DK_SB();
        DK_MM(FA, FB, 0, 1); DK_SB(); DK_LD(GB[1], pix_addr(NIV, 1), 0); DK_LD(GB[2], pix_addr(NIV, 2), 0); DK_SB();
        DK_MM(FA, FB, 0, 2); DK_SB(); DK_LD(GB[3], pix_addr(NIV, 3), 0); DK_LD(GA[1], wa, 2048); DK_SB();
        DK_MM(FA, FB, 0, 3); DK_SB();
        if (IV + 6 < 9) DK_WEIGHTS(c0, c1, RIV, freed); else DK_WEIGHTS(d0, d1, RIV, freed);
        DK_SB();
        DK_MM(FA, FB, 1, 0); DK_SB();
        DK_MM(FA, FB, 1, 1); DK_SB();
        if (IV == 2) { DK_PATCH(d0, d1, 0); DK_SB(); }           // buffer h: last read in interval 2h + 2
        if (IV == 4) { DK_PATCH(d0, d1, 1); DK_SB(); }
        if (IV == 6) { DK_PATCH(d0, d1, 2); DK_SB(); }
        if (IV == 8) { DK_PATCH(d0, d1, 3); DK_SB(); }
        DK_MM(FA, FB, 1, 2); DK_SB();
        DK_MM(FA, FB, 1, 3); DK_SB();
    };

    // barrier of "interval -1": W(0) and H0 of the first pair
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dk_allow(-1)) : "memory");
    __builtin_amdgcn_s_barrier();
    if (STAMP && stamps) stamps[1] = __builtin_amdgcn_s_memtime();
    DK_WEIGHTS(c0, c1, 5, 5 * WSTAGE); DK_PATCH(c0, c1, 3);      // R_{-1}
    DK_LD(fa0[0], wlane, 0); DK_LD(fa0[1], wlane, 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) DK_LD(fb0[j], pix_addr(0, j), 0);
    for (int P = 0; P < npairs; ++P) {
        interval(std::integral_constant<int, 0>(), fa0, fb0, fa1, fb1); interval(std::integral_constant<int, 1>(), fa1, fb1, fa0, fb0);
        interval(std::integral_constant<int, 2>(), fa0, fb0, fa1, fb1); interval(std::integral_constant<int, 3>(), fa1, fb1, fa0, fb0);
        interval(std::integral_constant<int, 4>(), fa0, fb0, fa1, fb1); interval(std::integral_constant<int, 5>(), fa1, fb1, fa0, fb0);
        interval(std::integral_constant<int, 6>(), fa0, fb0, fa1, fb1); interval(std::integral_constant<int, 7>(), fa1, fb1, fa0, fb0);
        interval(std::integral_constant<int, 8>(), fa0, fb0, fa1, fb1);
        // nine intervals are an odd number: the sets have swapped roles; copy back (6 x 4 moves per pair of chunks)
        fa0[0] = fa1[0]; fa0[1] = fa1[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb0[j] = fb1[j];
        c0 = d0; c1 = d1; d0 = chunk_of(2 * P + 4); d1 = chunk_of(2 * P + 5);
        first_pair = false;
    }
#undef DK_LD
#undef DK_SB
#undef DK_MM
#undef DK_PATCH
#undef DK_WEIGHTS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // the trailing (poisoned) requests still write zeros into LDS
    if (STAMP && stamps) stamps[2] = __builtin_amdgcn_s_memtime();
    if (a.dbg & 2) return;                                       // dev aid (MPU_PIPE_DEBUG): main loop only

    // ---- sum of the eight waves' partial tiles, fixed order: wave w finishes block (i = w / 4, j = w % 4) --------------
    // scratch [source wave][block j][q][lane] float4: every ds_write / ds_read_b128 is lane-contiguous (conflict-free)
    const int iw = wave >> 2, jw = wave & 3;
    // the epilogue's operands (bias, ReLU mask of the wave's block) are requested NOW, unconditionally (a null tensor gets an
    // empty descriptor: zeros) -- a load under `if (mask)` is waited for on the spot ("serialised loads")
    const int m_out = m0 + jw * 32 + pib;
    typedef unsigned int dk_u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? (const void*)a.bias : a.out), 0,
                                                                          a.bias ? a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0,
                                                                          a.mask ? (int)((long)M * a.Cout * 2L) : 0, 0x00020000);
    // BatchNorm-backward sums (a.bn_x, SURVEY 8a row a7 / DESIGN 4.5): the BatchNorm's input at the output positions, its
    // mean and 1 / std per channel
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? a.bn_x : a.out), 0,
                                                                          a.bn_x ? (int)((long)M * a.Cout * 2L) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsmu = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? (const void*)a.bn_mean : a.out), 0,
                                                                           a.bn_x ? a.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsis = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bn_x ? (const void*)a.bn_invstd : a.out), 0,
                                                                           a.bn_x ? a.Cout * 4 : 0, 0x00020000);
    dk_u32x4 bq4[4], mu4[4], is4[4]; dk_u32x2 mk2[4], bx2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = n0 + iw * 32 + 8 * q + 4 * fh;
        bq4[q] = __builtin_amdgcn_raw_buffer_load_b128(rsb, c * 4, 0, 0);
        mk2[q] = __builtin_amdgcn_raw_buffer_load_b64(rsm, (m_out * a.Cout + c) * 2, 0, 0);
        bx2[q] = __builtin_amdgcn_raw_buffer_load_b64(rsx, (m_out * a.Cout + c) * 2, 0, 0);
        mu4[q] = __builtin_amdgcn_raw_buffer_load_b128(rsmu, c * 4, 0, 0);
        is4[q] = __builtin_amdgcn_raw_buffer_load_b128(rsis, c * 4, 0, 0);
    }
    float res[16], sq[16];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();                                         // pass 0: lagging waves still read the last stage; 1: pass 0's reads
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(smem + ((wave * 4 + j) * 4 + q) * 1024 + lane * 16) =
                    make_float4(acc[pass][j][4 * q], acc[pass][j][4 * q + 1], acc[pass][j][4 * q + 2], acc[pass][j][4 * q + 3]);
        __syncthreads();
        if (iw == pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *(const float4*)(smem + ((s * 4 + jw) * 4 + q) * 1024 + lane * 16);
                    res[4 * q] += v.x; res[4 * q + 1] += v.y; res[4 * q + 2] += v.z; res[4 * q + 3] += v.w;
                }
        }
    }
    if (STAMP && stamps) stamps[3] = __builtin_amdgcn_s_memtime();
    // ---- epilogue on the wave's block: lane = (pixel l31 [permuted], half fh): channels n0 + iw*32 + 8q + 4fh + 0..3 -------
    {
        bf16_t* out = (bf16_t*)a.out;
        const bool has_mask = a.mask != nullptr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = n0 + iw * 32 + 8 * q + 4 * fh;
            float v[4] = {res[4 * q] + __uint_as_float(bq4[q].x), res[4 * q + 1] + __uint_as_float(bq4[q].y),
                          res[4 * q + 2] + __uint_as_float(bq4[q].z), res[4 * q + 3] + __uint_as_float(bq4[q].w)};
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            uint2 o;
            o.x = f32x2_to_bf16x2(v[0], v[1]); o.y = f32x2_to_bf16x2(v[2], v[3]);
            auto keep = [](uint32_t mw, uint32_t vw) {
                const uint32_t lo = ((mw & 0x8000u) == 0 && (mw & 0x7fffu) != 0) ? 0x0000ffffu : 0u;
                const uint32_t hi = ((mw & 0x80000000u) == 0 && (mw & 0x7fff0000u) != 0) ? 0xffff0000u : 0u;
                return vw & (lo | hi);
            };
            const uint32_t kx = keep(mk2[q].x, o.x), ky = keep(mk2[q].y, o.y);
            o.x = has_mask ? kx : o.x; o.y = has_mask ? ky : o.y;
            if (!(a.dbg & 1)) *(uint2*)(out + (long)m_out * a.Cout + c) = o;
            // statistics of the STORED (rounded, masked) values: (sum x, sum x^2), or with bn_x (sum dn, sum dn * xhat)
            const float s0 = __uint_as_float(o.x << 16), s1 = __uint_as_float(o.x & 0xffff0000u);
            const float s2 = __uint_as_float(o.y << 16), s3 = __uint_as_float(o.y & 0xffff0000u);
            const bool bw = a.bn_x != nullptr;
            const float f0 = bw ? (__uint_as_float(bx2[q].x << 16) - __uint_as_float(mu4[q].x)) * __uint_as_float(is4[q].x) : s0;
            const float f1 = bw ? (__uint_as_float(bx2[q].x & 0xffff0000u) - __uint_as_float(mu4[q].y)) * __uint_as_float(is4[q].y) : s1;
            const float f2 = bw ? (__uint_as_float(bx2[q].y << 16) - __uint_as_float(mu4[q].z)) * __uint_as_float(is4[q].z) : s2;
            const float f3 = bw ? (__uint_as_float(bx2[q].y & 0xffff0000u) - __uint_as_float(mu4[q].w)) * __uint_as_float(is4[q].w) : s3;
            res[4 * q] = s0; res[4 * q + 1] = s1; res[4 * q + 2] = s2; res[4 * q + 3] = s3;
            sq[4 * q] = s0 * f0; sq[4 * q + 1] = s1 * f1; sq[4 * q + 2] = s2 * f2; sq[4 * q + 3] = s3 * f3;
        }
    }
    if (a.stats) {
        // per channel over the tile's 128 pixels: the 32 pixels of the wave's block by a lane butterfly (fixed order), the four
        // pixel blocks of a channel half through LDS; one partial row per pixel tile: stats[2][Cout][tiles_m]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float u = res[r], w2 = sq[r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { u += __shfl_xor(u, o, 64); w2 += __shfl_xor(w2, o, 64); }
            res[r] = u; sq[r] = w2;
        }
        __syncthreads();                                         // the reduction scratch is free again
        float* sc = (float*)smem;                                // [wave][2 stats][32 channels of the block]
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = 8 * (r >> 2) + 4 * fh + (r & 3);
                sc[(wave * 2 + 0) * 32 + cl] = res[r]; sc[(wave * 2 + 1) * 32 + cl] = sq[r];
            }
        }
        __syncthreads();
        if (tid < 128) {                                         // (channel of the tile, statistic)
            const int ch = tid & 63, st2 = tid >> 6;
            const int ib = ch >> 5, cl = ch & 31;
            float t = 0.f;
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) t += sc[((ib * 4 + jb) * 2 + st2) * 32 + cl];
            stats_emit(a, st2, n0 + ch, tiles_m, mt, t);
        }
    }
    if (STAMP && stamps) {
        stamps[4] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[5] = __builtin_amdgcn_s_memtime();
    }
}

template <int W_>
int launch_deepk(const ConvArgs& a_in, hipStream_t st) {
    using Cfg = DkCfg<W_>;
    unsigned long long* sbuf = stamp_buffer();                  // MPU_STAMPS=1: the instrumented instantiation
    auto kern = sbuf ? conv_deepk_kernel<W_, true> : conv_deepk_kernel<W_, false>;
    ConvArgs a = a_in;
    a.dbg_buf = sbuf;
    a.dbg = (int)env(ENV_PIPE_DEBUG);
    if (a.w_elems <= 0) a.w_elems = 8 * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    static unsigned long long attr_set = 0;
    if (first_use_on_device(attr_set)) {
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_deepk_kernel<W_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        MPU_CHECK_HIP(hipFuncSetAttribute((const void*)conv_deepk_kernel<W_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM));
        mark_used_on_device(attr_set);
    }
    const long M = (long)a.B * a.Ho * a.Wo;
    const int tiles_m = (int)(M / Cfg::BM), tiles_n = a.Cout / Cfg::BN;
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * 9 * (a.C0 + a.C1), st);
    launch_k(kern, dim3((unsigned)((long)tiles_m * tiles_n)), dim3(512), Cfg::SMEM, st, a, tiles_m, tiles_n);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// 6 = launched, 0 = shape / epilogue not suited (conv_deep / conv_pipe take it), < 0 = error. bf16 3 x 3 layers on square
// 16-pixel maps, sources in multiples of 64 channels with an EVEN number of 64-channel chunks, filters in multiples of 64,
// about one 128-pixel x 64-filter tile per CU (192..512); epilogue: bias, ReLU, ReLU mask, fused BatchNorm statistics of the
// stored values (forward: sum x, sum x^2; with bn_x: the BatchNorm-backward sums), no folded-BN affine / pooling / head.
int try_conv_deepk(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    if (!env(ENV_CONV_DEEPK) || dtype != MPU_BF16 || mode != CONV3 || a.Ho != 16 || a.Wo != 16) return 0;
    if (a.pooled || a.head_w || a.post_scale) return 0;
    if (a.bn_x && !(a.stats && a.bn_mean && a.bn_invstd)) return 0;
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M % 128 || a.Cout % 64 || a.C0 % 64 || a.C1 % 64 || a.C0 <= 0 || (((a.C0 + a.C1) >> 6) & 1)) return 0;
    const long tiles = (M / 128) * (a.Cout / 64);
    if (tiles < 192 || tiles > 512) return 0;
    // Reductions over more than 512 input channels stay on conv_pipe: inside the train step (weights cold in L2) the
    // 1024 -> 512 concat layer took 52.0 us here against 40.5 + 7.2 for conv_pipe + finish (gpurun R5j); up to 512 channels
    // this kernel is 3-7 us ahead per layer.
    if (a.C0 + a.C1 > 512) return 0;
    // (8 x 8 maps -- tile = two images, 128 tiles at configs[1] sizes, K split two ways over workgroups with the finish pass of
    // conv_pipe -- were built and parity-green in this round: 25.1 / 32.5 us against 29.6 / 34.1 for conv_pipe + finish back to
    // back, but slower inside the step, where the 9-19 MB of bottom-level weights arrive cold and this kernel streams them for
    // twice as many pixel tiles: step 2.598 ms with them against 2.584 without, gpurun R5m vs R5i; removed)
    {   // 32-bit offsets with a poison margin
        const long cmax = a.C0 > a.C1 ? a.C0 : a.C1;
        const long wel = a.w_elems > 0 ? a.w_elems : 8 * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
        const long lim = (1L << 31) - 8192;
        if ((M + 1) * cmax * 2L >= lim || wel * 2L >= lim || M * a.Cout * 2L >= lim) return 0;
    }
    ConvArgs b = a;
    if (b.stats && b.stats_rows) {                               // one partial row per pixel tile
        const long rows = M / 128;
        if (rows * 2 * b.Cout <= b.stats_cap) *b.stats_rows = (int)rows;
        else { b.stats = nullptr; *b.stats_rows = 0; if (b.bn_x) return 0; }
    } else { if (b.bn_x) return 0; b.stats = nullptr; }
    const int rc = launch_deepk<16>(b, st);
    return rc ? rc : 6;
}

}  // namespace mpu
