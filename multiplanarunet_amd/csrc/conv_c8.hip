// conv_c8_kernel: 3x3 SAME convolution of a layer with at most 8 (padded) input channels -- the first U-Net
// layer (Cin = n_channels, stored as one 16-byte record per pixel) -- as a K = 9 taps x 8 channels = 72 (-> 80)
// GEMM. The tiled kernels run this layer as if it had 64 input channels (36 MFMA k-steps, 32 of them on zero
// padding, and a 128-byte patch row per pixel); here
//   * one 32x32x16 MFMA k-step covers TWO taps: lanes 0-31 supply the 8 channels of tap 2j, lanes 32-63 those
//     of tap 2j+1, so the B fragment of a lane IS the 16-byte record of its pixel shifted by that tap (read
//     straight from global memory / L1; the input is <= 1/8 of the output) and 5 k-steps do the whole conv;
//   * the weights (Cout x 80) live in registers for the whole workgroup (NB x 5 fragments);
//   * the kernel is bound by its output stream: each wave turns a 32-pixel row segment into NB x 32 channels,
//     stages it through its own LDS rows and writes full 16-byte, pixel-contiguous runs.
// Same operand conventions as the other forward kernels (packed weights [co][tap][ci], fp32 bias, ReLU).
#include <stdlib.h>
#include "kernels.h"

namespace mpu {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

constexpr int C8_ROWS_DEFAULT = 16;                             // output rows per workgroup (4 waves, interleaved rows)

template <int NB>
__global__ __launch_bounds__(256) void conv_c8_kernel(ConvArgs a, int tiles_x, int tiles_y, int nseg) {
    constexpr int ROWB = NB * 64 + 16;                           // staging row of one pixel: NB*32 channels + pad
    constexpr int CPP = NB * 4;                                  // 16-byte chunks per staged pixel
    constexpr unsigned OOB = 0xfffffff0u;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][32 * ROWB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 31, fh = lane >> 5;
    const int H = a.Ho, W = a.Wo;
    int t = blockIdx.x;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int b = t / tiles_y;
    const int x0 = tx * 32, y0 = ty * nseg * 4;                  // nseg row segments per wave
    const long npix = (long)a.B * H * W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.in0, 0, (int)(npix * 16L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)(a.w_elems * 2L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)(npix * a.Cout * 2L), 0x00020000);

    // weights: fragment (nb, j) of a lane = row co = 32 nb + (lane & 31), k = 8 channels of tap 2j + (lane >> 5)
    u32x4 wreg[NB][5];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int co = nb * 32 + px, tap = 2 * j + fh;
            const unsigned off = (co < a.Cout && tap < 9)
                ? (unsigned)(((long)co * a.w_row_stride + (long)tap * a.w_tap_stride) * 2) : OOB;
            wreg[nb][j] = __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0);
        }
    // bias table in LDS (a lane's accumulators hold channels 32 nb + 8 q + 4 fh + (0..3))
    __shared__ __attribute__((aligned(16))) float bias_s[NB * 32];
    if (tid < NB * 32) bias_s[tid] = (a.bias && tid < a.Cout) ? a.bias[tid] : 0.f;
    __syncthreads();
    const float* const bias_l = bias_s + fh * 4;
    // per-lane tap geometry of the 5 k-steps
    int tdy[5], tdx[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap = 2 * j + fh;
        tdy[j] = tap / 3 - 1; tdx[j] = tap - (tap / 3) * 3 - 1;
    }
    const float lo = a.relu ? 0.f : -__builtin_inff();
    unsigned char* const st_w = stage[wave] + px * ROWB + fh * 8;
    const int x = x0 + px;

    auto load_row = [&](int y, u32x4 (&bf)[5]) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int yy = y + tdy[j], xx = x + tdx[j];
            const bool ok = (2 * j + fh < 9) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned off = ok ? (unsigned)((((long)b * H + yy) * W + xx) * 16) : OOB;
            bf[j] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
        }
    };

    u32x4 cur[5], nxt[5];
    int y = y0 + wave;
    if (y < H) load_row(y, cur);
    for (int s = 0; s < nseg; ++s, y += 4) {
        if (y >= H) break;                                       // wave-uniform
        if (s + 1 < nseg && y + 4 < H) load_row(y + 4, nxt);     // next segment in flight during this one
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, wreg[nb][j]),
                                                                 __builtin_bit_cast(s16x8, cur[j]), acc[nb], 0, 0, 0);
        // registers -> the wave's staging rows (8 bytes = 4 channels per write)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *(const float4*)(bias_l + nb * 32 + q * 8);
                float v[4] = {acc[nb][4 * q] + bq.x, acc[nb][4 * q + 1] + bq.y, acc[nb][4 * q + 2] + bq.z,
                              acc[nb][4 * q + 3] + bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], lo, __builtin_inff());
                uint2 pk;
                pk.x = f32x2_to_bf16x2(v[0], v[1]);
                pk.y = f32x2_to_bf16x2(v[2], v[3]);
                *(uint2*)(st_w + nb * 64 + q * 16) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // staging rows -> global: lane c of round i moves 16-byte chunk (pixel c / CPP, chunk c % CPP)
        const long obase = (((long)b * H + y) * W + x0) * a.Cout * 2L;
#pragma unroll
        for (int i = 0; i < 32 * CPP / 64; ++i) {
            const int c = lane + 64 * i;
            const int p = c / CPP, ch = c - p * CPP;
            const u32x4 val = *(const u32x4*)(stage[wave] + p * ROWB + ch * 16);
            const bool ok = x0 + p < W && ch * 8 < a.Cout;
            const unsigned off = ok ? (unsigned)(obase + ((long)p * a.Cout + ch * 8) * 2L) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(val, rso, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staging reads done before the next segment's writes
#pragma unroll
        for (int j = 0; j < 5; ++j) cur[j] = nxt[j];
    }
}

template <int NB>
int launch_c8(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (a.w_elems <= 0) a.w_elems = 8 * a.w_tap_stride + (long)a.Cout * a.w_row_stride;
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M * 16L >= (1L << 31) || M * a.Cout * 2L >= (1L << 31) || a.w_elems * 2L >= (1L << 31))
        return fail(MPU_EUNSUPPORTED, "%s", "conv: operand larger than 2 GiB (split the batch)");
    // measured (B=16 128x128 / B=138 256x256, 64 channels): 16 rows 12.7 / 357 us, 32 rows 15.0 / 342 us, 8 rows 13.4 / 377 us
    int rows = C8_ROWS_DEFAULT;
    if ((long)a.B * cdiv(a.Wo, 32) * cdiv(a.Ho, C8_ROWS_DEFAULT) >= 4096) rows = 2 * C8_ROWS_DEFAULT;
    const int tx = cdiv(a.Wo, 32), ty = cdiv(a.Ho, rows);
    const long tiles = (long)a.B * tx * ty;
    if (tiles >= (1L << 31)) return fail(MPU_EUNSUPPORTED, "%s", "conv: too many tiles");
    if (prof_on()) prof_begin(PROF_CONV, a.flops > 0 ? a.flops : 2.0 * M * a.Cout * 9 * a.C0, st);
    launch_k(conv_c8_kernel<NB>, dim3((unsigned)tiles), dim3(256), 0, st, a, tx, ty, rows / 4);
    if (prof_on()) prof_end(st);
    return launch_ok();
}

}  // namespace

// 1 = launched, 0 = shape not suited (the caller falls back to the tiled kernels), < 0 = error
int try_conv_c8(int dtype, int mode, const ConvArgs& a, hipStream_t st) {
    const bool on = env(ENV_CONV_C8) != 0;
    if (!on || dtype != MPU_BF16 || mode != CONV3 || a.C1 != 0 || a.in1 || a.C0 != 8) return 0;
    if (a.Cout % 8 || a.Cout > 128 || a.mask || a.post_scale || a.ksplit > 1) return 0;
    int rc;
    switch (cdiv(a.Cout, 32)) {
        case 1: rc = launch_c8<1>(a, st); break;
        case 2: rc = launch_c8<2>(a, st); break;
        case 3: rc = launch_c8<3>(a, st); break;
        default: rc = launch_c8<4>(a, st); break;
    }
    return rc ? rc : 1;
}

}  // namespace mpu
