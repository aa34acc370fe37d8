// wgrad_c8_kernel: weight and bias gradient of the first U-Net layer (3x3 conv, CI = 1 or 2 image channels stored in
// 8-channel pixel records, Cout <= 128):  dW[tap][ci][co] = sum_px x[px + tap][ci] * dz[px][co],  db[co] = sum_px dz.
// The MFMA weight-gradient kernels treat the layer as a 64-input-channel one (63/64 of their work is zero padding);
// here the arithmetic is 9 * CI * Cout FMAs per pixel on the vector ALU and the kernel is bound by streaming dz once
// (algorithmic bytes: M * Cout * 2 for dz + M * 16 for x; HBM roofline).
//   thread = (pixel lane, group of 8 output channels): one 16-byte dz load per pixel, nine 4-byte x loads (clamped
//   addresses + select, so that all ten are in flight together), 72 * CI accumulators in registers;
//   workgroup = a strip of image rows; wave butterfly over the pixel lanes, then the four waves through LDS in a
//   fixed order -> one compact fp32 partial row [9][CI][Cout] (+ [Cout]) per workgroup;
//   wgrad_c8_finalize_kernel sums the rows (double, fixed order), writes dW in the padded [9][8][Cout] master layout
//   (zeros for the padding channels) and db.
#include <stdlib.h>
#include "kernels.h"
#include "reduce.h"

namespace mpu {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

namespace {

template <int CI>
__device__ __forceinline__ void wgrad_c8_body(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, int H, int W,
                                              int Cout, int R, int strips, float* __restrict__ partial) {
    constexpr int NA = 9 * CI * 8;                               // accumulators per thread
    const int groups = Cout >> 3;                                // power of two <= 16 (host-checked)
    const int tid = threadIdx.x;
    const int g = tid & (groups - 1), pl = tid / groups, npl = 256 / groups;
    const int b = blockIdx.x / strips, y0 = (blockIdx.x - b * strips) * R;
    const int y1 = y0 + R < H ? y0 + R : H;
    float acc[NA], db[8];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) db[k] = 0.f;
    // buffer loads: out-of-image taps / lanes past the row end use the out-of-range marker and read 0 -- one
    // unconditional instruction each (a select around a plain load becomes a branch that is waited for on the spot)
    constexpr unsigned OOB = 0xfffffff0u;
    const long npix = (long)(gridDim.x / strips) * H * W;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(npix * 16L), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsz = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (int)(npix * Cout * 2L), 0x00020000);
    // two pixels per pass: their 2 x (one 16-byte dz load + nine 4-byte x loads) are requested before the first FMA
    // (one pixel per pass left each wave with a single exposed round trip per pixel: 21 us for 33 MB)
    // 32-bit offsets (host-checked ranges): pixel base once, per-tap deltas are scalars, validity = an OR-mask that
    // turns the offset into the out-of-range marker: column masks per lane (3 per pixel), row masks per wave (uniform).
    // The masks go through an empty asm so that the compiler cannot turn them back into branches around the loads
    // (it duplicates the load on both sides and waits in between). The address arithmetic of the first version
    // (64-bit products per tap) cost more vector instructions than the FMAs.
    auto vmask = [](bool ok) { int m = ok ? 0 : (int)OOB; asm volatile("" : "+v"(m)); return (unsigned)m; };
    auto smask = [](bool ok) { int m = ok ? 0 : (int)OOB; asm volatile("" : "+s"(m)); return (unsigned)m; };
    const unsigned zpix = (unsigned)Cout * 2u;
    auto issue = [&](int y, int xx, u32x4& zq, uint32_t (&xr)[9]) {
        const unsigned pix = (unsigned)((b * H + y) * W + xx);
        const bool live = xx < W;
        const unsigned cm[3] = {vmask(live && xx >= 1), vmask(live), vmask(live && xx + 1 < W)};
        const unsigned rm[3] = {smask(y >= 1), 0u, smask(y + 1 < H)};
        zq = __builtin_amdgcn_raw_buffer_load_b128(rsz, (pix * zpix + (unsigned)g * 16u) | cm[1], 0, 0);
        const unsigned pbase = pix * 16u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t % 3;
            const unsigned delta = (unsigned)(((dy - 1) * W + (dx - 1)) * 16);          // uniform
            xr[t] = __builtin_amdgcn_raw_buffer_load_b32(rsx, (pbase + delta) | cm[dx] | rm[dy], 0, 0);
        }
    };
    auto consume = [&](const u32x4& zq, const uint32_t (&xr)[9]) {       // a dead lane read zeros everywhere
        const uint32_t zw[4] = {zq.x, zq.y, zq.z, zq.w};
        float zf[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            zf[2 * q] = __uint_as_float(zw[q] << 16);
            zf[2 * q + 1] = __uint_as_float(zw[q] & 0xffff0000u);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) db[k] += zf[k];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float x0f = __uint_as_float(xr[t] << 16), x1f = __uint_as_float(xr[t] & 0xffff0000u);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[(t * CI) * 8 + k] += x0f * zf[k];
                if (CI > 1) acc[(t * CI + 1) * 8 + k] += x1f * zf[k];
            }
        }
    };
    for (int y = y0; y < y1; ++y) {
        for (int xb = 0; xb < W; xb += 2 * npl) {
            u32x4 za, zb; uint32_t xa[9], xb9[9];
            issue(y, xb + pl, za, xa);
            issue(y, xb + npl + pl, zb, xb9);
            consume(za, xa);
            consume(zb, xb9);
        }
    }
    // wave butterfly over the pixel lanes (lanes that share g are `groups` apart), fixed order
    for (int o = groups; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
#pragma unroll
        for (int k = 0; k < 8; ++k) db[k] += __shfl_xor(db[k], o, 64);
    }
    __shared__ float red[4][16 * (NA + 8)];
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < groups) {
#pragma unroll
        for (int i = 0; i < NA; ++i) red[wave][lane * (NA + 8) + i] = acc[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave][lane * (NA + 8) + NA + k] = db[k];
    }
    __syncthreads();
    // compact partial row: [9 * CI][Cout] then [Cout]
    const int ncol = 9 * CI * Cout + Cout;
    float* prow = partial + (long)blockIdx.x * ncol;
    for (int c = tid; c < ncol; c += 256) {
        int gi, idx;
        if (c < 9 * CI * Cout) { const int r = c / Cout, co = c - r * Cout; gi = co >> 3; idx = r * 8 + (co & 7); }
        else { const int co = c - 9 * CI * Cout; gi = co >> 3; idx = NA + (co & 7); }
        const int o = gi * (NA + 8) + idx;
        prow[c] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    }
}

// one image channel: 4 waves per SIMD (<= 128 registers) so that the 1024 strips of configs[1] are resident at once
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void wgrad_c8_kernel_1(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, int H, int W, int Cout, int R, int strips,
                       float* __restrict__ partial) {
    wgrad_c8_body<1>(x, dz, H, W, Cout, R, strips, partial);
}
__global__ __launch_bounds__(256)
void wgrad_c8_kernel_2(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dz, int H, int W, int Cout, int R, int strips,
                       float* __restrict__ partial) {
    wgrad_c8_body<2>(x, dz, H, W, Cout, R, strips, partial);
}

// dW[(tap * 8 + ci) * Cout + co] = sum_rows partial[row][(tap * CI + ci) * Cout + co] (0 for ci >= CI); db likewise
__global__ __launch_bounds__(256) void wgrad_c8_finalize_kernel(const float* __restrict__ partial, int rows, int CI, int Cout,
                                                                int sum_blocks, float* __restrict__ dW, float* __restrict__ db) {
    if ((int)blockIdx.x >= sum_blocks) {                         // tail blocks: zeros for the padding channels [9][8 - CI][Cout]
        const int e = ((int)blockIdx.x - sum_blocks) * 256 + threadIdx.x, per = (8 - CI) * Cout;
        if (e < 9 * per) { const int tap = e / per; dW[((long)tap * 8 + CI) * Cout + (e - tap * per)] = 0.f; }
        return;
    }
    __shared__ double red[256];
    const int ncol = 9 * CI * Cout + Cout;
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x % FIN_COLS);
    double s;
    partial_sums<1>(partial, rows, ncol, 0, c, c < ncol, red, &s);
    if (c >= ncol || threadIdx.x >= FIN_COLS) return;
    if (c < 9 * CI * Cout) {
        const int r = c / Cout, co = c - r * Cout, tap = r / CI, ci = r - tap * CI;
        dW[((long)tap * 8 + ci) * Cout + co] = (float)s;
    } else if (db) {
        db[c - 9 * CI * Cout] = (float)s;
    }
}

}  // namespace

// Shape test of the schedule, shared by the launcher and by the workspace plan (wgrad_scratch_need): returns the
// number of strips (= partial rows, one per workgroup) or 0 when the shape is not suited.
static long c8_strips(int dtype, int mode, int B, int H, int W, int C0, int C1, int c0_logical, int Cout, int* R_out,
                      int* strips_out) {
    const bool on = env(ENV_WGRAD_C8) != 0;
    if (!on || dtype != MPU_BF16 || mode != CONV3 || C1 != 0 || C0 != 8) return 0;
    if (c0_logical < 1 || c0_logical > 2) return 0;
    const int groups = Cout / 8;
    if (Cout % 8 || groups < 1 || groups > 16 || (groups & (groups - 1))) return 0;
    const long rows_all = (long)B * H;
    const int R = (int)((rows_all + 1023) / 1024);
    const int strips = cdiv(H, R);
    const long wgs = (long)B * strips;
    if (wgs > 2048 || W < 1) return 0;
    if (R_out) *R_out = R;
    if (strips_out) *strips_out = strips;
    return wgs;
}

long wgrad_c8_scratch_floats(int dtype, int mode, int B, int H, int W, int C0, int C1, int c0_logical, int Cout) {
    const long wgs = c8_strips(dtype, mode, B, H, W, C0, C1, c0_logical, Cout, nullptr, nullptr);
    return wgs * (9L * c0_logical * Cout + Cout);
}

// 1 = handled (dW and, if wanted, db written), 0 = shape not suited, < 0 = error
int try_wgrad_c8(int dtype, int mode, const WgradArgs& a, float* dW, hipStream_t st) {
    if (a.x1) return 0;
    int R = 0, strips = 0;
    const long wgs = c8_strips(dtype, mode, a.B, a.Ho, a.Wo, a.C0, a.C1, a.c0_logical, a.Cout, &R, &strips);
    if (wgs == 0) return 0;
    // one compact partial row per workgroup: the caller's region must hold them (the plan sizes it with
    // wgrad_c8_scratch_floats; an op-level caller states its capacity)
    if (a.partial_cap < wgs * (9L * a.c0_logical * a.Cout + a.Cout)) return 0;
    const long M = (long)a.B * a.Ho * a.Wo;
    if (M * 16L >= (1L << 31) || M * a.Cout * 2L >= (1L << 31)) return 0;   // 32-bit buffer offsets
    if (prof_on()) prof_begin(PROF_WGRAD, a.flops > 0 ? a.flops : 2.0 * M * 9 * a.C0 * a.Cout, st);
    if (a.c0_logical == 1)
        launch_k(wgrad_c8_kernel_1, dim3((unsigned)wgs), dim3(256), 0, st, (const bf16_t*)a.x0, (const bf16_t*)a.dz, a.Ho, a.Wo,
                                                                      a.Cout, R, strips, a.partial);
    else
        launch_k(wgrad_c8_kernel_2, dim3((unsigned)wgs), dim3(256), 0, st, (const bf16_t*)a.x0, (const bf16_t*)a.dz, a.Ho, a.Wo,
                                                                      a.Cout, R, strips, a.partial);
    if (prof_on()) prof_end(st);
    int rc = launch_ok();
    if (rc) return rc;
    const int ncol = 9 * a.c0_logical * a.Cout + a.Cout;
    const int sum_blocks = cdiv(ncol, FIN_COLS), npad = 9 * (8 - a.c0_logical) * a.Cout;
    launch_k(wgrad_c8_finalize_kernel, sum_blocks + cdiv(npad, 256), 256, 0, st, a.partial, (int)wgs, a.c0_logical, a.Cout,
                                                                          sum_blocks, dW, a.db);
    rc = launch_ok();
    return rc ? rc : 1;
}

}  // namespace mpu
