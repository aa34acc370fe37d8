// Internal C++ launch API shared by the translation units of libmpunet_hip.so.
#pragma once
#include "common.h"
#include "env.h"

namespace mpu {

enum { CONV3 = 0, UPCONV2 = 1, CONV3S2 = 2, CONV1 = 3 };

struct ConvArgs {
    const void* in0; const void* in1; int C0, C1;
    const void* w; long w_tap_stride; int w_row_stride;
    const float* bias; const void* mask; void* out;
    int B, Ho, Wo, Cout, relu;
    double flops;                // algorithmic FLOPs of this launch (profiling only; 0 = derive)
    long w_elems;                // elements addressable from `w` (0 = derive from strides)
    const float* post_scale;     // optional per-channel affine applied AFTER bias/ReLU (inference-mode
    const float* post_shift;     //   BatchNormalization folded into the producing conv); NULL = none
    float* partial;              // split-K scratch [ksplit][M][Cout] f32 (NULL = never split)
    long partial_cap;            // floats available at `partial`
    int ksplit;                  // set by the launcher
    // Optional fused BatchNorm statistics of the STORED output (training forward): the kernel writes per-workgroup
    // column sums [2][Cout][rows] (sum x, sum x^2; a column's partials contiguous) to `stats` and the launcher sets
    // *stats_rows to the number of rows
    // (0: the schedule chosen for this shape does not produce them; the caller runs the column reduction instead).
    float* stats; int* stats_rows; long stats_cap;              // capacity of `stats` in floats
    // Round 6: with stats_acc set the column sums are NOT written as partial rows but ADDED, as fixed-point integers, to
    // stats_acc[8 XCDs][2][Cout] (int64, zeroed by the caller; stats_emit below) and the launcher reports *stats_rows = -1: the
    // consumer then needs no finalize launch at all -- it sums eight integers per (statistic, channel). Integer sums are exact,
    // so the result does not depend on the order in which workgroups arrive (deterministic, as the fixed-order rows were).
    long long* stats_acc = nullptr; float stats_scale[2] = {0.f, 0.f};   // units per 1.0 of statistic 0 / 1 (powers of two)
    // With bn_x set the two sums are those of the BatchNorm BACKWARD pass of the layer that consumes this output as
    // its dn: sum out, sum out * (bn_x - mean) * invstd (bn_x: the BatchNorm's input, same shape as out). Schedules that
    // cannot produce them leave *stats_rows = 0.
    const void* bn_x = nullptr; const float* bn_mean = nullptr; const float* bn_invstd = nullptr;
    // Optional second output (inference forward of an encoder block): the 2x2 / stride-2 max pooling of the stored
    // output [B][Ho/2][Wo/2][Cout] (Ho, Wo even), taken from the epilogue's staging tile. The launcher sets
    // *pooled_done = 1 when the schedule it chose wrote it (else the caller runs launch_maxpool).
    void* pooled = nullptr; int* pooled_done = nullptr;
    // Optional fused 1x1 head (inference; the last conv of the up path on the conv_ws schedule): the output tensor is NOT
    // stored; every wave multiplies its 32-channel half of the staged (rounded) pixels with the head weights
    // head_w[c * head_ldw + k] and stores the partial logits head_partial[half][M][head_k] (f32). The launcher sets
    // *head_done = 1 when the schedule did so; launch_head_combine then adds the halves and the bias and applies the
    // softmax. Otherwise the caller runs launch_head_forward on the stored output.
    const float* head_w = nullptr; int head_k = 0, head_ldw = 0; float* head_partial = nullptr; int* head_done = nullptr;
    int x3 = 0;                  // f32 storage only: split-bf16 products (three bf16 MFMAs, common.h x3_mma) instead of exact-f32 MFMAs
    int xcd = 0;                 // set by the launchers of conv_halo / conv_halo8 / conv_ws (MPU_XCD_TILES): workgroup -> tile through
                                 //   xcd_contiguous() below, so that the tiles one XCD works on are neighbours (shared halo rows and,
                                 //   with several filter tiles, the shared patch hit in that XCD's L2 instead of being fetched per XCD)
    int dbg = 0;                 // profiling switches of conv_pipe_kernel (MPU_PIPE_DEBUG: 1 no stores, 2 no MFMAs, 4 no DMA,
                                 //   8 no fragment reads, 16 no barrier, 32 s_memtime stamps into dbg_buf)
    unsigned long long* dbg_buf = nullptr;
};
// Workgroups are dealt to the eight XCDs round-robin (XCD = workgroup id & 7). xcd_contiguous maps workgroup `bid` of `nwg` to
// a logical index such that XCD k owns the contiguous range [k * nwg / 8, (k + 1) * nwg / 8) (ragged counts: the first nwg & 7
// XCDs take one more), in launch order inside the range. A bijection on [0, nwg).
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// One column sum of a producer's tile: a partial row (the fixed-order scheme), or -- stats_acc mode -- a fixed-point atomic add
// into the row of THIS workgroup's XCD (hardware register XCC_ID, so the eight per-XCD L2 caches never share an address; inside an
// XCD the L2 is the point of coherence of every compute unit, which is where a scope-less global atomic executes).
__device__ __forceinline__ void stats_acc_add(long long* acc, int C, int st2, int ch, float v, float scale) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;       // HW_REG_XCC_ID[3:0]
    double d = (double)v * (double)scale;
    d = d > 9.0e18 ? 9.0e18 : (d < -9.0e18 ? -9.0e18 : d);       // (saturate: an overflowing sum must not wrap)
    const long long q = __double2ll_rn(d);
    __hip_atomic_fetch_add(acc + ((long)xcc * 2 + st2) * C + ch, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void stats_emit(const ConvArgs& a, int st2, int ch, long nrows, long row, float v) {
    if (a.stats_acc) stats_acc_add(a.stats_acc, a.Cout, st2, ch, v, a.stats_scale[st2]);
    else a.stats[((long)st2 * a.Cout + ch) * nrows + row] = v;
}
// bias gradients of dtype "bf16x3" out of accumulators of the same layout (unet_ops.hip: launch_split3_colsum / launch_db_from_acc)
struct DbAccJob { const long long* acc; float* db; int C; };
constexpr int DB_ACC_MAX_JOBS = 32;
struct DbAccTable { DbAccJob job[DB_ACC_MAX_JOBS]; int n = 0; float inv_scale = 0.f; };
constexpr int BN_ACC_ROWS = 8;                   // XCDs
inline long bn_acc_elems(int C) { return (long)BN_ACC_ROWS * 2 * C; }      // int64 elements of one accumulator

struct WgradArgs {
    const void* x0; const void* x1; int C0, C1;
    const void* dz; int Cout;
    int x3 = 0;                  // f32 storage only: split-bf16 products instead of exact-f32 MFMAs (sits in the padding behind Cout:
                                 //   the grouped launches carry 24 of these structs in 4 KB of kernel arguments). In a BF16 job of
                                 //   wgrad_taps: -P = the batch is three plane pairs of P images over TWO stored planes (hi | lo)
    float* partial;              // [ksplit][ntaps*Cin*Cout]
    float* db_partial;           // [ksplit][Cout] (fused bias-gradient partials; set by the launcher)
    int B, Ho, Wo, ksplit, mchunk;
    double flops;
    float* db;                   // bias gradient sum_m dz[m][co] (NULL = not wanted)
    float* colsum_scratch;       // [RED_MAX_BLOCKS][Cout] scratch for the unfused bias-gradient path
    int fuse_db;                 // set by the launcher: the wgrad kernel also produces the db partials
    int c0_logical;              // image channels actually present in an 8-channel x0 (first layer); 0 = unknown
    long partial_cap = 0;        // floats available at `partial` (0 = unknown: schedules with their own layout refuse)
    unsigned long long* dbg_buf = nullptr;   // dev aid (MPU_STAMPS=1): s_memtime stamps of a few workgroups
    // One SOURCE of a concat layer run as its own job (round 3): x0 / C0 is that source alone (C1 = 0), the result belongs
    // to rows [ci_base, ci_base + C0) of every tap of a kernel with cin_total input channels. 0 = the whole layer.
    int cin_total = 0, ci_base = 0;
};

// element-wise maximum of two 16-byte pieces of T (8 bf16 or 4 f32)
__device__ __forceinline__ uint32_t bf16x2_max(uint32_t a, uint32_t b) {
    const float ml = fmaxf(__uint_as_float(a << 16), __uint_as_float(b << 16));
    const float mh = fmaxf(__uint_as_float(a & 0xffff0000u), __uint_as_float(b & 0xffff0000u));
    return (__float_as_uint(mh) & 0xffff0000u) | (__float_as_uint(ml) >> 16);
}
template <typename T> __device__ __forceinline__ uint32_t piece_max(uint32_t a, uint32_t b) {
    if (sizeof(T) == 2) return bf16x2_max(a, b);
    return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b)));
}

// dev aid: device buffer for in-kernel s_memtime stamps (nullptr unless MPU_STAMPS=1); 64 slots of 8 stamps
unsigned long long* stamp_buffer();

// ---- profile.hip: optional per-launch HIP-event timing of the MFMA kernels ----
enum { PROF_CONV = 0, PROF_WGRAD = 1, PROF_KINDS = 2 };
bool prof_on();
void prof_begin(int kind, double flops, hipStream_t st);   // opens a timing scope (one "launch": one or more kernels)
void prof_end(hipStream_t st);
hipEvent_t prof_start_event();                  // the open scope's start event, ONCE (its first kernel); nullptr otherwise
hipEvent_t prof_stop_event();                   // the open scope's stop event (every kernel re-binds it); nullptr when no scope is open
// Kernel launch of the timed families. Inside an open scope the dispatch itself carries the scope's events
// (hipExtLaunchKernel: start / stop time stamps of the dispatch packet, which is what rocprofv3 reports), instead of two
// hipEventRecord marker packets around it (those add ~2 us of dispatch latency to every ~30-us kernel).
template <typename... KA, typename... A>
inline void launch_k(void (*kern)(KA...), dim3 grid, dim3 block, unsigned shmem, hipStream_t st, A... args) {
    if (hipEvent_t stop = prof_stop_event())
        hipExtLaunchKernelGGL(kern, grid, block, shmem, st, prof_start_event(), stop, 0u, static_cast<KA>(args)...);
    else
        kern<<<grid, block, shmem, st>>>(static_cast<KA>(args)...);
}
bool sched_log_on();                            // schedule log (mpu_schedule_log_*): one line per conv / wgrad launch
void sched_note(const char* fmt, ...);

int  launch_conv(int dtype, int mode, const ConvArgs& a, hipStream_t st);       // dispatches on MPU_CONV_IMPL
int  launch_conv_glds(int dtype, int mode, const ConvArgs& a, hipStream_t st);  // LDS-DMA variant (conv_glds.hip)
const char* last_glds_schedule();               // "glds" or "pipe": what the last launch_conv_glds of this thread ran
int  try_conv_deepk(int dtype, int mode, const ConvArgs& a, hipStream_t st);   // 3x3 on 16-pixel maps, K split over the waves of a workgroup (conv_deepk.hip)
int  try_conv_c8(int dtype, int mode, const ConvArgs& a, hipStream_t st);       // <= 8 input channels: first layer (conv_c8.hip)
int  try_conv_ws(int dtype, int mode, const ConvArgs& a, hipStream_t st);       // register-stationary weights, persistent (conv_ws.hip)
int  try_conv_halo(int dtype, int mode, const ConvArgs& a, hipStream_t st);     // LDS-resident patch variant (conv_halo.hip)
int  try_conv_halo16(int dtype, int mode, const ConvArgs& a, hipStream_t st);   // ... 16-row tiles, persistent, staggered halves: large inference grids (conv_halo16.hip)
// grouped: the job will run inside a grouped launch (WgradGroup): it need not fill the chip on its own, so it takes
// about half the workgroups (K splits / pixel strips) of a stand-alone launch -- half the fp32 partial copies
long wgrad_partial_elems(int mode, int Cin, int Cout, long M, int* ksplit_out, int* mchunk_out, bool grouped = false);
// all-taps weight gradient for the high-resolution 3x3 layers (wgrad_taps.hip)
struct TapsPlan { int use, RH, sx, sy, nstrips, split; };
constexpr long TAPS_MAX_CICO = 512L * 512;     // eligible layers: Cin * Cout up to this
constexpr int TAPS_MAX_WGS = 1024;             // strips * 64x64 tiles; bounds the fp32 partial workspace (one copy of
                                               // dW per strip): <= 1024 * 9 * 4096 floats = 151 MB for any layer
TapsPlan wgrad_taps_plan(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, bool grouped = false);
int  launch_wgrad_taps(int mode, const WgradArgs& a, const TapsPlan& p, hipStream_t st);
// Deferred second stage of the weight gradients: launch_wgrad(.., q) records its reduction (fixed-order sum of the
// K-split partials into dW, bias-gradient finalize) in q instead of launching it; flush_wgrad_reduces() runs every
// recorded job in ONE launch. The partial buffers of the recorded layers must stay untouched until the flush.
struct ReduceJob {
    const float* partial; float* dW; long n; int ksplit, kl4;    // main part (n = 0: none)
    const float* db_partial; float* db; int nshare, C;           // bias gradient (db = nullptr: none)
    int blk_begin, main_blocks, db_blocks, il4_cout;             // il4_cout: see store_dw_sum (conv_igemm.hip)
    int cin_job = 0, cin_total = 0, ci_base = 0, cout = 0;       // cout: output channels; cin_job > 0: the partials hold cin_job input channels per tap that
                                                                 // belong to rows [ci_base, ci_base + cin_job) of a cin_total-row tap of dW
};
constexpr int REDUCE_MAX_JOBS = 32;
struct ReduceQueue { int njobs = 0, nblocks = 0; ReduceJob job[REDUCE_MAX_JOBS]; };
enum { WG_TAPS = 1, WG_GLDS = 2, WG_ALL = 3 };  // which: jobs of the wgrad_taps schedule / all others
int  flush_wgrad_reduces(ReduceQueue& q, hipStream_t st, int which = WG_ALL);
// Deferred weight-gradient KERNELS (round 3): the weight gradients of a backward pass do not depend on each other, so
// launch_wgrad(.., q, grp) only prepares them (schedule, split, scratch) and records them here; flush_wgrad_group() runs
// all recorded wgrad_taps jobs as ONE launch and the wgrad_glds jobs as one launch per tile variant. A grid of many
// waves of workgroups amortises what a one-wave launch pays in full -- dispatch ramp, prologue, the tail behind the
// slowest workgroup and the L2 write-back of the fp32 partials at the kernel boundary (measured on configs[1] shapes:
// four times the workgroups of one layer in one launch take 0.77-0.94 of four launches). Inputs (x, dz) must stay
// untouched until the flush: every conv owns its dz buffer (Plan::dz).
struct TapsGroupJob { WgradArgs a; TapsPlan p; int mode, blk_begin; };
struct GldsGroupJob { WgradArgs a; int mode, blk_begin; };
constexpr int TAPS_GROUP_MAX = 24, GLDS_GROUP_MAX = 12;   // (24: every non-first conv of a depth-4 network; the job table stays under 4 KB of kernel arguments)
struct WgradGroup {
    int ntaps = 0, nglds = 0; double taps_flops = 0, glds_flops = 0;
    TapsGroupJob taps[TAPS_GROUP_MAX]; GldsGroupJob glds[GLDS_GROUP_MAX];
};
int  flush_wgrad_group(int dtype, WgradGroup& g, hipStream_t st, int which = WG_ALL);
int  launch_wgrad_taps_group(const TapsGroupJob* jobs, int n, hipStream_t st);          // wgrad_taps.hip
int  launch_wgrad_glds_group(int dtype, const GldsGroupJob* jobs, int n, hipStream_t st);   // conv_glds.hip (bf16, 128 x 128 tiles)
int  wgrad_taps_grid(int mode, const WgradArgs& a, const TapsPlan& p);                  // workgroups of one job
long wgrad_glds_grid(int mode, const WgradArgs& a);                                      // (0: not the groupable variant)
// exact scratch need (floats) of one layer's weight gradient: K-split partials + bias-gradient partials
// c0_logical: image channels of an 8-channel first layer (selects the wgrad_c8 schedule, which has its own layout)
long wgrad_job_floats(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, bool grouped);   // one plan's exact need
long wgrad_scratch_need(int dtype, int mode, int B, int H, int W, int C0, int C1, int Cout, int c0_logical = 0);
long wgrad_c8_scratch_floats(int dtype, int mode, int B, int H, int W, int C0, int C1, int c0_logical, int Cout);   // 0 = not eligible
int  launch_wgrad(int dtype, int mode, const WgradArgs& a, float* dW, hipStream_t st, ReduceQueue* q = nullptr,
                  WgradGroup* grp = nullptr);
int  try_wgrad_c8(int dtype, int mode, const WgradArgs& a, float* dW, hipStream_t st);   // first layer (wgrad_c8.hip)
int  try_wgrad_glds(int dtype, int mode, const WgradArgs& a, hipStream_t st);   // 1 launched, 0 unsupported shape
bool wgrad_glds_supported(int dtype, int mode, const WgradArgs& a);

// ---- unet_ops.hip ---------------------------------------------------------
constexpr int RED_MAX_BLOCKS = 256;
constexpr int HEAD_BWD_MAX_BLOCKS = 1024;      // partial rows of the head weight and bias gradient (4 workgroups per CU; 2048: slower)

// fp32 master [taps][Cin][Cout] -> packed operands in T
// one launch for every 3x3 / up-conv layer of a model (offsets in ELEMENTS of params / the packed buffer)
struct PackJob { int mode, Cin, Cout, unit_begin, fwd_units, _pad; long w, wf, wd; };
constexpr int PACK_MAX_JOBS = 40;
struct PackTable { int njobs, _pad; PackJob job[PACK_MAX_JOBS]; };
int launch_pack_all(int dtype, PackTable& tab, const float* params, void* packed, hipStream_t st);
// Adam on the whole flat buffer + both packed operand copies of the listed kernels in ONE launch (unet_ops.hip);
// step != NULL: device-resident step counter (graph replay; incremented afterwards), else t_host (1-based)
int launch_adam_pack_all(int dtype, PackTable& jobs, float* params, const float* grads, float* am, float* av, long n_params,
                         void* packed, long long* step, long long t_host, double lr, double b1, double b2, float eps,
                         hipStream_t st);
// ... of the parameters in nr (<= 2) ascending ranges [p_lo[k], p_hi[k]) only; lean: the register- and LDS-lean kernel that is
// co-resident with wgrad_taps (bf16); a device step counter must already hold this step's number (see launch_head_backward)
int launch_adam_pack_ranges(int dtype, PackTable& jobs, float* params, const float* grads, float* am, float* av, const long* p_lo,
                            const long* p_hi, int nr, void* packed, long long* step, long long t_host, double lr, double b1,
                            double b2, float eps, bool lean, hipStream_t st);
int launch_pack_weights(int dtype, int mode, const float* W, int Cin, int Cout,
                        void* w_fwd, void* w_dgrad, hipStream_t st);
// dtype "bf16x3": n packed f32 words -> (bf16 hi | bf16 lo << 16) in place (after every refresh of the packed operand copies)
int launch_x3_words(void* buf, long n, hipStream_t st);
int launch_cast_pad(int dtype, const float* x, long M, int Cin, int Cpad, void* out, hipStream_t st, long long* zero_p = nullptr,
                    long zero_n = 0);       // zero_p: int64 words zeroed by the same launch (the BatchNorm accumulators of a training step)

// BatchNormalization, training: batch statistics of x [M][C]
//   partial: [RED_MAX_BLOCKS][2][C] floats scratch
//   writes mean, invstd, scale = gamma*invstd, shift = beta - mean*scale and
//   updates moving_mean / moving_var in place (momentum .99, Bessel-corrected var)
int launch_bn_stats(int dtype, const void* x, long M, int C, float* partial,
                    const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                    float* mean, float* invstd, float* scale, float* shift, float eps, float momentum,
                    int ready_rows /* > 0: partial already holds [2][C][ready_rows] sums from the conv epilogue */,
                    hipStream_t st);
// inference: scale/shift from the moving statistics
int launch_bn_infer_coeffs(const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, int C, float eps, float* scale, float* shift, hipStream_t st);
// y = x*scale + shift (and optionally 2x2 max-pool of y)
// 2x2/stride-2 max pooling (inference path; training pools inside bn_apply)
int launch_maxpool(int dtype, const void* x, int B, int H, int W, int C, void* pooled, hipStream_t st);
// finalize + apply (+ pool) in ONE launch for producers that left <= 64 column-major partial rows (1 = launched, 0 = not suited)
int launch_bn_fold_fwd(int dtype, const void* x, int B, int H, int W, int C, const float* partial, int nblk,
                       const float* gamma, const float* beta, float* mmean, float* mvar, float* mean, float* invstd,
                       float* scale, float* shift, float eps, float momentum, void* y, void* pooled, hipStream_t st,
                       const long long* acc = nullptr, const float* acc_scale = nullptr);   // acc: the accumulator mode of ConvArgs.stats_acc
bool bn_fold_shape_ok(int C, int H, int W, bool pooled);          // shapes the folded kernels take (accumulator mode is offered only there)
int launch_bn_apply(int dtype, const void* x, int B, int H, int W, int C, const float* scale,
                    const float* shift, void* y, void* pooled, hipStream_t st);
// BN backward: dgamma, dbeta and dz = (x>0) * d(x) where x is the post-ReLU BN input
int launch_bn_backward(int dtype, const void* dn, const void* x, long M, int C, float* partial,
                       const float* gamma, const float* mean, const float* invstd,
                       float* dgamma, float* dbeta, float* coeffs /*[3][C]*/, void* dz,
                       int ready_rows /* > 0: partial already holds that many partial rows; -1: the producer used `acc` */,
                       int ready_colmajor /* their layout: 0 = [rows][2][C], 1 = [2][C][rows] (conv epilogues) */,
                       hipStream_t st, long long* acc = nullptr /* zeroed accumulator of this BatchNorm (ConvArgs.stats_acc), or NULL */,
                       const float* acc_scale = nullptr);
// dn = dskip + unpool(dp) (gradient of MaxPooling2D routed to the first max of each window)
int launch_maxpool_bwd_add(int dtype, const void* n, const void* dskip, const void* dp,
                           int B, int H, int W, int C, void* dn, hipStream_t st);
// ... and the BN-backward partial sums (sum dn, sum dn * xhat) of the level's BatchNorm in the same pass
int launch_maxpool_bwd_add_stats(int dtype, const void* n, const void* dskip, const void* dp, int B, int H, int W, int C,
                                 void* dn, const void* x, const float* mean, const float* invstd, float* partial,
                                 long partial_cap, int* rows, hipStream_t st, long long* acc = nullptr, const float* acc_scale = nullptr);
// f32 tensor -> three bf16 planes stacked along the batch axis (order 0: hi | lo | hi, 1: hi | hi | lo): dtype "bf16x3" weight gradients
struct Split3Job { const float* src; uint16_t* dst; long n; int blk_begin; int order; };   // order 0: hi | lo | hi, 1: hi | hi | lo, 2: hi | lo
constexpr int SPLIT3_MAX_JOBS = 64;
struct Split3Table { Split3Job job[SPLIT3_MAX_JOBS]; int n = 0; };
int launch_split3_all(Split3Table& t, hipStream_t st);           // all jobs in one launch (blk_begin is filled in)
int launch_split3(const float* x, long n, void* out, int order, hipStream_t st);
int launch_split3_colsum(const float* dz, long M, int C, void* planes, long long* acc, float scale, hipStream_t st, int two_planes = 0);
int launch_db_from_acc(const DbAccTable& t, hipStream_t st);
// db[c] = sum_m dz[m][c]
int launch_colsum(int dtype, const void* dz, long M, int C, float* partial, float* out, hipStream_t st);
// out[c] = sum_k partial[k*C + c], k < nblk (second stage only)
int launch_colsum_finalize(const float* partial, int nblk, int C, float* out, hipStream_t st);

// 1x1 head: logits = n @ Wh + bh ; probs = softmax (or linear)
int launch_head_forward(int dtype, const void* n, long M, int C, int K, const float* Wh, int ldw,
                        const float* bh, int softmax, float* out, hipStream_t st);
// second stage of the head fused into the last conv (ConvArgs.head_partial [2][M][K]): logits = p0 + p1 + bh, softmax / linear
int launch_head_combine(const float* partial, long M, int K, const float* bh, int softmax, float* out, hipStream_t st);
// Keras sparse CE (clipped probabilities, see oracle/unet_ref.py keras_sparse_ce), sum-gradient:
// dn = dlogits @ Wh^T, dWh, dbh, per-pixel loss
int launch_head_backward(int dtype, const void* n, const float* probs, const uint8_t* y,
                         const float* sample_w, long M, long pix_per_image, int C, int K,
                         const float* Wh, int ldw, float* partial, void* dn, float* dWh, float* dbh,
                         float* loss, hipStream_t st, long long* step_incr = nullptr /* device counter to advance by one */,
                         float* loss_mean = nullptr /* device scalar: mean of the weighted per-pixel loss */);
// Round 6: max-pool backward + skip add + BatchNorm backward of an encoder level WITHOUT the dn tensor between them (two passes that
// both recompute it; accumulator mode). 1 = launched, 0 = not suited, < 0 = error
int launch_maxpool_bwd_bn(int dtype, const void* dskip, const void* dp, int B, int H, int W, int C, const void* x,
                          const float* mean, const float* invstd, const float* scale, const float* shift, const float* gamma,
                          float* dgamma, float* dbeta, float* coeffs, void* dz, long long* acc, const float* acc_scale, hipStream_t st);
// Round 6: the training step's head without the post-BatchNorm tensor of the last block (unet_ops.hip, "head_bn_*")
bool head_train_fused_shape_ok(int dtype, int C, int K);
int launch_head_bn_forward(int dtype, const void* x, long M, const long long* acc, const float* acc_scale, const float* gamma, const float* beta,
                           float* mmean, float* mvar, float* mean, float* invstd, float* scale, float* shift, float eps, float momentum,
                           int K, const float* Wh, int ldw, const float* bh, int softmax, float* out, hipStream_t st);
int launch_head_bn_backward(int dtype, const void* x, const float* probs, const uint8_t* y, const float* sw, long M, long ppi, int K,
                            const float* mean, const float* invstd, float* partial, float* tsum, float* dbh, float* loss,
                            hipStream_t st, long long* step_incr, float* loss_mean);
int launch_head_bn_bwd_apply(int dtype, const void* x, const float* probs, const uint8_t* y, const float* sw, long M, long ppi, int K,
                             const float* Wh, int ldw, const float* tsum, const float* dbh, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dWh, float* coeffs, void* dz,
                             hipStream_t st);

// l2 kernel regulariser: grads += 2*l2*W over the listed tensors; reg_loss (optional) = l2 * sum W^2
struct L2Table { int njobs, _pad; long off[PACK_MAX_JOBS]; long n[PACK_MAX_JOBS]; };
constexpr int L2_PARTIAL_DOUBLES = PACK_MAX_JOBS * 64;
int launch_l2_regularizer(const L2Table& tab, const float* params, float* grads, float l2, double* partial,
                          float* reg_loss, hipStream_t st);
int launch_adam_dev(float* p, const float* g, float* m, float* v, long n, long long* step, double lr, double b1,
                    double b2, float eps, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, long n, float alpha, float b1, float b2,
                float eps, hipStream_t st);

}  // namespace mpu
