// Host-side orchestration of the 2-D U-Net (mpunet/models/unet.py:114-216) on
// top of the HIP kernels, plus the U-Net part of the C ABI. The model object
// is a host-only description (layer table, parameter / workspace layout); every
// device buffer belongs to the caller.
#include <string>
#include <vector>
#include <cmath>
#include <stdlib.h>
#include "kernels.h"

using namespace mpu;

namespace {

constexpr float BN_EPS = 1e-3f, BN_MOM = 0.99f;      // Keras BatchNormalization defaults
// fixed-point units of the accumulator mode (ConvArgs.stats_acc; powers of two): forward sum x in 2^-24, sum x^2 in 2^-16 (a
// mean-square activation up to 6e4 per XCD share of a 128^2 x 16 batch before the int64 saturates; their errors, <= 2^-17 / 256
// per pixel, are far below epsilon = 1e-3); the backward sums (gradients: small numbers) in 2^-40
constexpr float BN_ACC_F[2] = {16777216.f, 65536.f}, BN_ACC_B[2] = {1099511627776.f, 1099511627776.f};

inline int pad8(int c) { return (c + 7) / 8 * 8; }
inline long align64(long e) { return (e + 63) / 64 * 64; }

struct Tensor {                 // one entry of the flat parameter / BN-state tables
    std::string name;           // "<keras layer>/<var>"
    int kind;                   // 0 = trainable (params/grads/m/v), 1 = BN moving statistic
    long offset;                // floats, into d_params (kind 0) or d_bn_state (kind 1)
    int pshape[4];              // stored (channel-padded) shape, 0-terminated
    int lshape[4];              // logical Keras shape
};

struct Conv { int mode, Cin, Cout; long w, b; long wf, wd; int lCin, lCout; };        // offsets
struct BN { int C; long g, b, mm, mv; long st; };                    // st: mean,invstd,scale,shift (4*C)

}  // namespace

struct mpu_unet {
    mpu_unet_config cfg;
    int cin_pad = 0;
    std::vector<int> F, Fl;                 // padded / logical filters per level (depth+1)
    std::vector<Conv> conv;                 // creation order
    std::vector<BN> bn;
    std::vector<Tensor> tensors;
    long n_params = 0, n_state = 0, n_packed = 0, n_stats = 0, n_logical = 0;
    long infer_off = 0;                      // byte offset of the inference BN coefficients inside the packed buffer
    int head_C = 0; long head_w = 0, head_b = 0;
    int cmax = 0;
    int x3 = 0;                              // MPU_F32X3: cfg.dtype holds MPU_F32 (storage, every elementwise kernel), the MFMA kernels split
    mpu_launch_tap_fn tap = nullptr; void* tap_user = nullptr;      // test aid: mpu_unet_set_launch_tap
    // Round 6: the last training forward ran the fused head (head_bn_forward: no post-BatchNorm tensor of the last block exists);
    // its backward pass must take head_bn_backward / head_bn_bwd_apply. A backward pass belongs to exactly one training forward.
    mutable int head_fused_fwd = 0;

    // indices into conv / bn
    int enc_c1(int i) const { return 2 * i; }
    int enc_c2(int i) const { return 2 * i + 1; }
    int bot_c1() const { return 2 * cfg.depth; }
    int bot_c2() const { return 2 * cfg.depth + 1; }
    int up_c(int j, int k) const { return 2 * cfg.depth + 2 + 3 * j + k; }       // k = 0,1,2
    int enc_bn(int i) const { return i; }
    int bot_bn() const { return cfg.depth; }
    int up_bn(int j, int k) const { return cfg.depth + 1 + 2 * j + k; }          // k = 0,1
};

namespace {

void add_tensor(mpu_unet* m, const std::string& name, int kind, long off, std::initializer_list<int> p,
                std::initializer_list<int> l) {
    Tensor t; t.name = name; t.kind = kind; t.offset = off;
    int i = 0; for (int v : p) t.pshape[i++] = v; for (; i < 4; ++i) t.pshape[i] = 0;
    i = 0; for (int v : l) t.lshape[i++] = v; for (; i < 4; ++i) t.lshape[i] = 0;
    m->tensors.push_back(t);
}

void add_conv(mpu_unet* m, const std::string& name, int mode, int Cin, int Cout, int lCin, int lCout) {
    const int esz = 1;
    const int k = mode == UPCONV2 ? 2 : (mode == CONV1 ? 1 : 3);
    Conv c; c.mode = mode; c.Cin = Cin; c.Cout = Cout; c.lCin = lCin; c.lCout = lCout;
    c.w = m->n_params; m->n_params += (long)k * k * Cin * Cout;
    c.b = m->n_params; m->n_params += Cout;
    add_tensor(m, name + "/kernel", 0, c.w, {k, k, Cin, Cout}, {k, k, lCin, lCout});
    add_tensor(m, name + "/bias", 0, c.b, {Cout}, {lCout});
    m->n_logical += (long)k * k * lCin * lCout + lCout;
    c.wf = c.wd = -1;
    if (mode != CONV1) {
        c.wf = m->n_packed; m->n_packed = align64(m->n_packed + (long)k * k * Cin * Cout * esz);
        c.wd = m->n_packed; m->n_packed = align64(m->n_packed + 9L * Cin * Cout * esz);
    }
    m->conv.push_back(c);
}

void add_bn(mpu_unet* m, const std::string& name, int C, int lC) {
    BN b; b.C = C;
    b.g = m->n_params; m->n_params += C;
    b.b = m->n_params; m->n_params += C;
    b.mm = m->n_state; m->n_state += C;
    b.mv = m->n_state; m->n_state += C;
    b.st = m->n_stats; m->n_stats += 4L * C;
    add_tensor(m, name + "/gamma", 0, b.g, {C}, {lC});
    add_tensor(m, name + "/beta", 0, b.b, {C}, {lC});
    add_tensor(m, name + "/moving_mean", 1, b.mm, {C}, {lC});
    add_tensor(m, name + "/moving_variance", 1, b.mv, {C}, {lC});
    m->n_logical += 2L * lC;
    m->bn.push_back(b);
    if (C > m->cmax) m->cmax = C;
}

// ---- workspace plan --------------------------------------------------------
struct Plan {
    long xin;
    std::vector<long> c1, c2, n, p, dskip;          // encoder levels
    long c1b, c2b, nb;
    std::vector<long> u1, n1, c2u, c3u, n2;         // up levels
    long probs, loss_mean, gA, gB, partial, partial_floats, partial2, wpartial, wpartial_floats, cpartial, cpartial_floats, coeffs, stats, total;
    std::vector<long> wscratch;                     // per conv: float offset of its weight-gradient scratch inside wpartial
    std::vector<long> dz;                           // per conv: its own dz (gradient at the conv's pre-activation output): the weight
                                                    // gradients of a whole backward pass run as grouped launches at its end
    std::vector<long> bnacc_f, bnacc_b;             // per BatchNorm: its fixed-point accumulators of the forward statistics / backward sums
    long bnacc, bnacc_elems;                        //   (ConvArgs.stats_acc: int64 [8 XCDs][2][C]); one region: forward half first
    std::vector<long> x3x0, x3x1, x3dz;             // dtype "bf16x3": per conv the bf16 plane triples of its input source(s) and its dz
    std::vector<long> dbacc;                        //   and, in the accumulator region, the fixed-point sums of its bias gradient
                                                    // (launch_split3), read by the grouped bf16 weight-gradient launches at the pass's end
};

Plan make_plan(const mpu_unet* m, int B) {
    const int D = m->cfg.depth, esz = m->cfg.dtype == MPU_BF16 ? 2 : 4;
    Plan P; long off = 0;
    auto take = [&](long bytes) { long o = off; off += (bytes + 255) / 256 * 256; return o; };
    auto act = [&](int lvl, int C) { return take((long)B * (m->cfg.H >> lvl) * (m->cfg.W >> lvl) * C * esz); };
    P.xin = act(0, m->cin_pad);
    for (int i = 0; i < D; ++i) {
        P.c1.push_back(act(i, m->F[i])); P.c2.push_back(act(i, m->F[i])); P.n.push_back(act(i, m->F[i]));
        P.p.push_back(act(i + 1, m->F[i])); P.dskip.push_back(act(i, m->F[i]));
    }
    P.c1b = act(D, m->F[D]); P.c2b = act(D, m->F[D]); P.nb = act(D, m->F[D]);
    for (int j = 0; j < D; ++j) {
        const int lvl = D - 1 - j;
        P.u1.push_back(act(lvl, m->F[lvl])); P.n1.push_back(act(lvl, m->F[lvl]));
        P.c2u.push_back(act(lvl, m->F[lvl])); P.c3u.push_back(act(lvl, m->F[lvl]));
        P.n2.push_back(act(lvl, m->F[lvl]));
    }
    const long M0 = (long)B * m->cfg.H * m->cfg.W;
    P.probs = take(M0 * m->cfg.n_classes * 4);
    P.loss_mean = take(16);                          // the last backward pass's mean weighted per-pixel loss (one float)
    long gmax = 0;
    for (int l = 0; l <= D; ++l) {
        const long e = (long)B * (m->cfg.H >> l) * (m->cfg.W >> l) * m->F[l];
        if (e > gmax) gmax = e;
    }
    P.gA = take(gmax * esz); P.gB = take(gmax * esz);   // (every conv has its own dz buffer below: Plan::dz)
    P.dz.assign(m->conv.size(), -1);
    for (size_t i = 0; i < m->conv.size(); ++i) {
        if (m->conv[i].mode == CONV1) continue;
        const int nenc = 2 * D;
        const int l = (int)i < nenc ? (int)i / 2 : ((int)i < nenc + 2 ? D : D - 1 - ((int)i - nenc - 2) / 3);
        P.dz[i] = act(l, m->conv[i].Cout);
    }
    long pe = (long)RED_MAX_BLOCKS * 2 * m->cmax;
    const long he = (long)HEAD_BWD_MAX_BLOCKS * (m->head_C * m->cfg.n_classes + m->cfg.n_classes + 1);
    if (he > pe) pe = he;
    if (pe < (4L << 20)) pe = 4L << 20;         // room for the per-tile BN statistics rows of the fused conv epilogues
    P.partial = take((pe + 1024) * 4);          // (+ 1024 floats the producers are never told about: T of the fused training head)
    P.partial_floats = pe;
    P.partial2 = take((long)RED_MAX_BLOCKS * m->cmax * 4);
    // Weight-gradient scratch: every layer has its OWN region (K-split partials + bias-gradient partials), because the
    // second-stage reductions are deferred and run batched (flush_wgrad_reduces); sized exactly per layer.
    long we = 0;
    {
        auto level_of = [&](size_t i) -> int {          // output resolution level of conv i (creation order)
            const int nenc = 2 * D;
            if ((int)i < nenc) return (int)i / 2;
            if ((int)i < nenc + 2) return D;
            const int j = ((int)i - nenc - 2) / 3;
            return j < D ? D - 1 - j : 0;
        };
        P.wscratch.assign(m->conv.size(), 0);
        for (size_t i = 0; i < m->conv.size(); ++i) {
            const Conv& c = m->conv[i];
            if (c.mode == CONV1) continue;
            const int l = level_of(i);
            const bool concat = c.mode == CONV3 && (int)i >= 2 * D + 2 && ((int)i - 2 * D - 2) % 3 == 1;
            const int C0 = concat ? c.Cin / 2 : c.Cin, C1 = concat ? c.Cin / 2 : 0;
            P.wscratch[i] = we;
            // (dtype "bf16x3": the weight gradient runs on the bf16 kernels over three plane pairs = a batch of 3 B)
            we += wgrad_scratch_need(m->x3 ? MPU_BF16 : m->cfg.dtype, c.mode, m->x3 ? 3 * B : B, m->cfg.H >> l, m->cfg.W >> l, C0, C1,
                                     c.Cout, i == 0 ? c.lCin : 0);
        }
        if (m->x3) {
            P.x3x0.assign(m->conv.size(), -1); P.x3x1.assign(m->conv.size(), -1); P.x3dz.assign(m->conv.size(), -1);
            for (size_t i = 0; i < m->conv.size(); ++i) {
                const Conv& c = m->conv[i];
                if (c.mode == CONV1) continue;
                const int l = level_of(i), li = c.mode == UPCONV2 ? l + 1 : l;      // (the up-conv reads the level below)
                const bool concat = c.mode == CONV3 && (int)i >= 2 * D + 2 && ((int)i - 2 * D - 2) % 3 == 1;
                const long pin = (long)B * (m->cfg.H >> li) * (m->cfg.W >> li), pout = (long)B * (m->cfg.H >> l) * (m->cfg.W >> l);
                const int C0 = concat ? c.Cin / 2 : c.Cin;
                P.x3x0[i] = take(3 * pin * C0 * 2);
                if (concat) P.x3x1[i] = take(3 * pin * (c.Cin - C0) * 2);
                P.x3dz[i] = take(3 * pout * c.Cout * 2);
            }
        }
    }
    P.wpartial = take(we * 4);
    P.wpartial_floats = we;
    long ce = 0;        // split-K scratch of the forward / data-gradient convs (its own region: the weight gradient of the
    {                   // same layer may run next to the data gradient on the side stream): [ks <= 16][M][Cout] at the deep levels
        for (int l = 0; l <= D; ++l) {
            const long M = (long)B * (m->cfg.H >> l) * (m->cfg.W >> l);
            const int fmax = l > 0 && m->F[l - 1] > m->F[l] ? m->F[l - 1] : m->F[l];
            if ((long)cdiv(M, 128) * cdiv(m->F[l], 128) < 256) { const long e = 16 * M * (long)fmax; if (e > ce) ce = e; }
        }
        if (ce > (96L << 20)) ce = 96L << 20;          // 384 MB is plenty: larger layers fill the chip without a K split
    }
    P.cpartial = take(ce * 4);
    P.cpartial_floats = ce;
    {   // accumulators of the fused BatchNorm sums: forward ones first, then the backward ones (each half zeroed by ONE launch)
        long e = 0;
        P.bnacc_f.assign(m->bn.size(), 0); P.bnacc_b.assign(m->bn.size(), 0);
        for (size_t i = 0; i < m->bn.size(); ++i) { P.bnacc_f[i] = e; e += bn_acc_elems(m->bn[i].C); }
        for (size_t i = 0; i < m->bn.size(); ++i) { P.bnacc_b[i] = e; e += bn_acc_elems(m->bn[i].C); }
        if (m->x3) {                                // (dtype "bf16x3": the bias-gradient sums of every conv, zeroed with the rest)
            P.dbacc.assign(m->conv.size(), -1);
            for (size_t i = 0; i < m->conv.size(); ++i)
                if (m->conv[i].mode != CONV1) { P.dbacc[i] = e; e += bn_acc_elems(m->conv[i].Cout); }
        }
        P.bnacc_elems = e;
        P.bnacc = take(e * 8);
    }
    P.coeffs = take(3L * m->cmax * 4);
    P.stats = take(m->n_stats * 4);
    P.total = off;
    return P;
}

struct Run {
    const mpu_unet* m; int B; hipStream_t st; unsigned char* ws; Plan P;
    const float* params; const unsigned char* packed; float* state; float* grads;
    void* const* ready_events = nullptr; int n_ready = 0;        // gradient-ready points (mpu_unet_backward_events)
    mutable ReduceQueue rq;                                      // deferred weight-gradient reductions (one launch per flush)
    mutable WgradGroup grp;                                      // deferred weight-gradient kernels (grouped launches at the end)
    mutable std::vector<char> late;                              // per conv: its weight gradient sits in the wgrad_taps group (final only after that launch)
    bool group = false;
    int esz;
    void* at(long off) const { return ws + off; }
    const void* wf(const Conv& c) const { return packed + c.wf * esz; }
    const void* wd(const Conv& c) const { return packed + c.wd * esz; }
    float* stat(const BN& b, int k) const { return (float*)(ws + P.stats) + b.st + (long)k * b.C; }
    long long* acc_f(const BN& b) const { return (long long*)(ws + P.bnacc) + P.bnacc_f[&b - &m->bn[0]]; }
    long long* acc_b(const BN& b) const { return (long long*)(ws + P.bnacc) + P.bnacc_b[&b - &m->bn[0]]; }
    long long* acc_db(size_t ci) const { return (long long*)(ws + P.bnacc) + P.dbacc[ci]; }
    bool acc_mode = false;                                       // fused BatchNorm sums into fixed-point accumulators (MPU_BN_ATOMIC)
    mutable DbAccTable dbq;                                      // dtype "bf16x3": bias gradients waiting in their accumulators
    mutable std::vector<char> x3_presplit;                       //   per conv: its input planes were written by the pass's first launch
    mutable std::vector<char> x3_two;                            //   per conv: two stored planes (hi | lo): its weight gradient is a wgrad_taps job
};

#define RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// algorithmic FLOPs of one pass over conv `c` at its OUTPUT resolution level `lvl`
// (2*M*N*K with the logical channel counts; the 2x2 up-conv counted at output resolution)
double conv_flops(const Run& r, const Conv& c, int lvl, int n_cnt_logical = -1) {
    const double M = (double)r.B * (r.m->cfg.H >> lvl) * (r.m->cfg.W >> lvl);
    const int taps = c.mode == UPCONV2 ? 4 : (c.mode == CONV1 ? 1 : 9);
    return 2.0 * M * taps * c.lCin * c.lCout;
}

// fused head request of the last conv (inference): weights, class count, scratch for the partial logits [2][M][k]
struct HeadFuse { const float* w; int k, ldw; float* partial; int* done; };

int conv_fwd(const Run& r, const Conv& c, const void* in0, int C0, const void* in1, int C1, void* out, int lvl,
             const float* post_scale = nullptr, const float* post_shift = nullptr, int* stats_rows = nullptr,
             void* pooled = nullptr, int* pooled_done = nullptr, const HeadFuse* head = nullptr, long long* stats_acc = nullptr) {
    ConvArgs a;
    const bool fused_head = env(ENV_FUSED_HEAD) != 0;   // inference: 1x1 head out of the last conv's epilogue (partial logits)
    if (head && head->done) *head->done = 0;
    if (head && fused_head && head->done) {
        a.head_w = head->w; a.head_k = head->k; a.head_ldw = head->ldw; a.head_partial = head->partial; a.head_done = head->done;
    }
    const bool fused_pool = env(ENV_FUSED_POOL) != 0;   // inference: 2x2 max pooling as a second output of the conv epilogue
    if (pooled_done) *pooled_done = 0;
    a.pooled = (fused_pool && pooled_done) ? pooled : nullptr; a.pooled_done = a.pooled ? pooled_done : nullptr;
    // training: the conv in front of a BatchNormalization also produces the per-tile column sums of its output
    const bool fused_stats = env(ENV_FUSED_BN_STATS) != 0;
    if (!fused_stats) stats_rows = nullptr;
    a.stats = stats_rows ? (float*)r.at(r.P.partial) : nullptr; a.stats_rows = stats_rows; a.stats_cap = r.P.partial_floats;
    if (stats_rows && stats_acc) { a.stats_acc = stats_acc; a.stats_scale[0] = BN_ACC_F[0]; a.stats_scale[1] = BN_ACC_F[1]; }
    a.bn_x = nullptr; a.bn_mean = nullptr; a.bn_invstd = nullptr;
    a.flops = conv_flops(r, c, lvl); a.w_elems = 0;
    a.partial = r.P.cpartial_floats ? (float*)r.at(r.P.cpartial) : nullptr; a.partial_cap = r.P.cpartial_floats; a.ksplit = 1;
    a.post_scale = post_scale; a.post_shift = post_shift;
    a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1;
    a.w = r.wf(c); a.w_tap_stride = (long)c.Cin * c.Cout; a.w_row_stride = c.Cin;
    a.bias = r.params + c.b; a.mask = nullptr; a.out = out;
    a.B = r.B; a.Ho = r.m->cfg.H >> lvl; a.Wo = r.m->cfg.W >> lvl; a.Cout = c.Cout; a.relu = 1;
    a.x3 = r.m->x3;
    const int rc = launch_conv(r.m->cfg.dtype, c.mode, a, r.st);
    if (!rc && r.m->tap && !post_scale) {
        mpu_launch_info li{};
        li.kind = 0; li.conv_index = (int)(&c - &r.m->conv[0]); li.mode = c.mode; li.dtype = r.m->cfg.dtype;
        li.B = a.B; li.H = a.Ho; li.W = a.Wo; li.C0 = C0; li.C1 = C1; li.Cout = c.Cout; li.n_off = 0; li.n_cnt = c.Cin; li.relu = 1;
        li.in0 = in0; li.in1 = in1; li.dz = nullptr; li.mask = nullptr; li.out = out; li.w_off = c.w; li.b_off = c.b;
        r.m->tap(r.m->tap_user, &li);
    }
    return rc;
}

// data gradient of conv `c` w.r.t. input channels [n_off, n_off + n_cnt); out_lvl = resolution of the result
// bn (optional): the output is the dn of that BatchNorm (input bn_x); the epilogue then also produces the partial sums
// of its backward pass and *bn_rows (> 0) tells bn_bwd to skip the column reduction (column-major layout).
int conv_dgrad(const Run& r, const Conv& c, const void* dz, const void* mask, void* out, int out_lvl,
               int n_off, int n_cnt, const BN* bn = nullptr, const void* bn_x = nullptr, int* bn_rows = nullptr) {
    ConvArgs a;
    const bool fused_bwd = env(ENV_FUSED_BN_BWD_CONV) != 0;
    if (bn_rows) *bn_rows = 0;
    const bool want = fused_bwd && bn && bn_x && bn_rows && !mask;
    a.stats = want ? (float*)r.at(r.P.partial) : nullptr; a.stats_rows = want ? bn_rows : nullptr;
    a.stats_cap = want ? r.P.partial_floats : 0;
    a.bn_x = want ? bn_x : nullptr; a.bn_mean = want ? r.stat(*bn, 0) : nullptr; a.bn_invstd = want ? r.stat(*bn, 1) : nullptr;
    if (want && r.acc_mode && !(bn->C & 63)) { a.stats_acc = r.acc_b(*bn); a.stats_scale[0] = BN_ACC_B[0]; a.stats_scale[1] = BN_ACC_B[1]; }
    a.in0 = dz; a.in1 = nullptr; a.C0 = c.Cout; a.C1 = 0; a.w_elems = 0;
    a.partial = r.P.cpartial_floats ? (float*)r.at(r.P.cpartial) : nullptr; a.partial_cap = r.P.cpartial_floats; a.ksplit = 1;
    a.post_scale = nullptr; a.post_shift = nullptr;
    a.w = (const unsigned char*)r.wd(c) + (long)n_off * c.Cout * r.esz;
    a.w_tap_stride = (long)c.Cin * c.Cout; a.w_row_stride = c.Cout;
    a.bias = nullptr; a.mask = mask; a.out = out;
    a.B = r.B; a.Ho = r.m->cfg.H >> out_lvl; a.Wo = r.m->cfg.W >> out_lvl; a.Cout = n_cnt; a.relu = 0;
    // the data gradient costs the forward's FLOPs (at the conv's own output level), pro rata of the slice
    a.flops = conv_flops(r, c, c.mode == UPCONV2 ? out_lvl - 1 : out_lvl) * ((double)n_cnt / c.Cin);
    a.x3 = r.m->x3;
    const int rc = launch_conv(r.m->cfg.dtype, c.mode == UPCONV2 ? CONV3S2 : CONV3, a, r.st);
    if (!rc && r.m->tap) {
        mpu_launch_info li{};
        li.kind = 1; li.conv_index = (int)(&c - &r.m->conv[0]); li.mode = c.mode; li.dtype = r.m->cfg.dtype;
        li.B = a.B; li.H = a.Ho; li.W = a.Wo; li.C0 = c.Cout; li.C1 = 0; li.Cout = c.Cout; li.n_off = n_off; li.n_cnt = n_cnt;
        li.relu = 0; li.in0 = nullptr; li.in1 = nullptr; li.dz = dz; li.mask = mask; li.out = out; li.w_off = c.w; li.b_off = c.b;
        r.m->tap(r.m->tap_user, &li);
    }
    return rc;
}

// The tensors the weight gradient of conv ci reads as its input (what run_backward passes to conv_wgrad; checked there)
struct WgradIn { const void* x0; int C0; const void* x1; int C1; long pin; int lvl_out; };
WgradIn wgrad_inputs(const Run& r, int ci) {
    const mpu_unet* m = r.m; const int D = m->cfg.depth; const Plan& P = r.P;
    WgradIn w{nullptr, 0, nullptr, 0, 0, 0};
    int lvl_in;
    if (ci < 2 * D) {
        const int i = ci / 2; lvl_in = i;
        if (ci % 2 == 0) { w.x0 = i > 0 ? r.at(P.p[i - 1]) : r.at(P.xin); w.C0 = i > 0 ? m->F[i - 1] : m->cin_pad; }
        else { w.x0 = r.at(P.c1[i]); w.C0 = m->F[i]; }
    } else if (ci == 2 * D) { w.x0 = D > 0 ? r.at(P.p[D - 1]) : r.at(P.xin); w.C0 = D > 0 ? m->F[D - 1] : m->cin_pad; lvl_in = D; }
    else if (ci == 2 * D + 1) { w.x0 = r.at(P.c1b); w.C0 = m->F[D]; lvl_in = D; }
    else {
        const int j = (ci - 2 * D - 2) / 3, k = (ci - 2 * D - 2) % 3, lvl = D - 1 - j, f = m->F[lvl];
        if (k == 0) { w.x0 = j > 0 ? r.at(P.n2[j - 1]) : r.at(P.nb); w.C0 = j > 0 ? m->F[lvl + 1] : m->F[D]; lvl_in = lvl + 1; }
        else if (k == 1) { w.x0 = r.at(P.n[lvl]); w.C0 = f; w.x1 = r.at(P.n1[j]); w.C1 = f; lvl_in = lvl; }
        else { w.x0 = r.at(P.c2u[j]); w.C0 = f; lvl_in = lvl; }
    }
    w.pin = (long)r.B * (m->cfg.H >> lvl_in) * (m->cfg.W >> lvl_in);
    w.lvl_out = (ci >= 2 * D + 2 && (ci - 2 * D - 2) % 3 == 0) ? lvl_in - 1 : lvl_in;      // (the up-conv writes the level above its input)
    return w;
}
// dtype "bf16x3": the bf16 planes of every conv's input in ONE launch at the start of the backward pass
int x3_presplit_inputs(const Run& r) {
    const mpu_unet* m = r.m;
    Split3Table t;
    r.x3_presplit.assign(m->conv.size(), 0);
    // Which layers will be wgrad_taps jobs of the grouped launch (the only kernel that folds the batch onto two stored planes):
    // launch_wgrad_mode's decision, from the shapes -- the first layer goes to wgrad_c8, a layer joins the taps group while it has
    // room. conv_wgrad CHECKS the outcome and fails if a predicted job went elsewhere.
    r.x3_two.assign(m->conv.size(), 0);
    if (r.group && r.acc_mode && env(ENV_WGRAD_BATCHED_REDUCE) != 0) {     // (acc_mode: the dz pass that knows two planes)
        int ntaps = 0, njobs = 0;
        for (size_t ci = 1; ci < m->conv.size(); ++ci) {
            const Conv& c = m->conv[ci];
            if (c.mode == CONV1) continue;
            const WgradIn w = wgrad_inputs(r, (int)ci);
            const int Ho = m->cfg.H >> w.lvl_out, Wo = m->cfg.W >> w.lvl_out;
            ++njobs;
            if (wgrad_taps_plan(MPU_BF16, c.mode, 3 * r.B, Ho, Wo, w.C0, w.C1, c.Cout, true).use) { r.x3_two[ci] = 1; ++ntaps; }
        }
        if (ntaps > TAPS_GROUP_MAX || njobs + 1 >= REDUCE_MAX_JOBS) r.x3_two.assign(m->conv.size(), 0);
    }
    for (size_t ci = 0; ci < m->conv.size(); ++ci) {
        if (m->conv[ci].mode == CONV1 || r.P.x3x0[ci] < 0) continue;
        const WgradIn w = wgrad_inputs(r, (int)ci);
        if (t.n + 2 > SPLIT3_MAX_JOBS) break;                    // (the rest is split layer by layer in conv_wgrad)
        const int order = r.x3_two[ci] ? 2 : 0;
        t.job[t.n++] = Split3Job{(const float*)w.x0, (uint16_t*)r.at(r.P.x3x0[ci]), w.pin * w.C0, 0, order};
        if (w.x1) t.job[t.n++] = Split3Job{(const float*)w.x1, (uint16_t*)r.at(r.P.x3x1[ci]), w.pin * w.C1, 0, order};
        r.x3_presplit[ci] = 1;
    }
    return launch_split3_all(t, r.st);
}

// dtype "bf16x3": the bias gradients queued by conv_wgrad, out of their accumulators (one launch for all of them)
constexpr float DB_ACC_SCALE = 17592186044416.f;                 // 2^44: |sum dz| < 2^19, quantum 6e-14 per workgroup sum
int flush_db(const Run& r) {
    if (r.dbq.n == 0) return MPU_OK;
    r.dbq.inv_scale = 1.f / DB_ACC_SCALE;
    const int rc = launch_db_from_acc(r.dbq, r.st);
    r.dbq.n = 0;
    return rc;
}

int conv_wgrad(const Run& r, const Conv& c, const void* x0, int C0, const void* x1, int C1, const void* dz, int lvl) {
    WgradArgs a;
    a.x0 = x0; a.x1 = x1; a.C0 = C0; a.C1 = C1; a.dz = dz; a.Cout = c.Cout;
    const size_t ci_ = &c - &r.m->conv[0];
    a.partial = (float*)r.at(r.P.wpartial) + r.P.wscratch[ci_];                     // this layer's own scratch
    a.partial_cap = (ci_ + 1 < r.P.wscratch.size() ? r.P.wscratch[ci_ + 1] : r.P.wpartial_floats) - r.P.wscratch[ci_];
    a.B = r.B; a.Ho = r.m->cfg.H >> lvl; a.Wo = r.m->cfg.W >> lvl;
    a.flops = conv_flops(r, c, lvl);
    const long M = (long)a.B * a.Ho * a.Wo;
    wgrad_partial_elems(c.mode, c.Cin, c.Cout, M, &a.ksplit, &a.mchunk, r.group);
    a.db = r.grads + c.b; a.db_partial = nullptr; a.colsum_scratch = (float*)r.at(r.P.partial2); a.fuse_db = 0;
    a.c0_logical = (C1 == 0 && C0 == c.Cin) ? c.lCin : 0;
    a.x3 = r.m->x3;
    const bool defer = env(ENV_WGRAD_BATCHED_REDUCE) != 0;   // 0: reduce right behind every weight-gradient kernel (A/B)
    ReduceQueue* q = defer ? &r.rq : nullptr;
    if (r.m->tap) {
        mpu_launch_info li{};
        li.kind = 2; li.conv_index = (int)ci_; li.mode = c.mode; li.dtype = r.m->cfg.dtype;
        li.B = a.B; li.H = a.Ho; li.W = a.Wo; li.C0 = C0; li.C1 = C1; li.Cout = c.Cout; li.n_off = 0; li.n_cnt = c.Cin; li.relu = 0;
        li.in0 = x0; li.in1 = x1; li.dz = dz; li.mask = nullptr; li.out = nullptr; li.w_off = c.w; li.b_off = c.b;
        r.m->tap(r.m->tap_user, &li);                  // (before the launch: x and dz are final, dW is read after the pass)
    }
    const int ntaps_before = r.grp.ntaps;
    if (r.m->x3) {
        // dtype "bf16x3": split x and dz into bf16 plane triples (batch 3 B) and run the bf16 weight-gradient schedules on them
        // (unet_ops.hip: launch_split3); the bias gradient, which is no product, is the plain column sum of the f32 dz
        const int Hi = c.mode == UPCONV2 ? a.Ho / 2 : a.Ho, Wi = c.mode == UPCONV2 ? a.Wo / 2 : a.Wo;
        const long pin = (long)r.B * Hi * Wi;
        if (ci_ < r.x3_presplit.size() && r.x3_presplit[ci_]) {  // (planes written by x3_presplit_inputs: same tensors?)
            const WgradIn w = wgrad_inputs(r, (int)ci_);
            if (w.x0 != x0 || w.x1 != x1 || w.C0 != C0 || w.C1 != C1 || w.pin != pin)
                return fail(MPU_EINVAL, "%s", "conv_wgrad: the pre-split input planes belong to other tensors (wgrad_inputs out of date)");
        } else {
            RC(launch_split3((const float*)x0, pin * C0, r.at(r.P.x3x0[ci_]), 0, r.st));
            if (x1) RC(launch_split3((const float*)x1, pin * C1, r.at(r.P.x3x1[ci_]), 0, r.st));
        }
        const bool two = r.acc_mode && ci_ < r.x3_presplit.size() && r.x3_presplit[ci_] && r.x3_two[ci_];   // (hi | lo planes: a wgrad_taps job)
        if (r.acc_mode) {                              // one pass over dz: its planes + its column sums (finalized by flush_db)
            RC(launch_split3_colsum((const float*)dz, M, c.Cout, r.at(r.P.x3dz[ci_]), r.acc_db(ci_), DB_ACC_SCALE, r.st, two ? 1 : 0));
            if (r.dbq.n == DB_ACC_MAX_JOBS) RC(flush_db(r));
            r.dbq.job[r.dbq.n++] = DbAccJob{r.acc_db(ci_), r.grads + c.b, c.Cout};
        } else {
            RC(launch_split3((const float*)dz, M * c.Cout, r.at(r.P.x3dz[ci_]), 1, r.st));
            RC(launch_colsum(MPU_F32, dz, M, c.Cout, a.colsum_scratch, a.db, r.st));
        }
        a.x0 = r.at(r.P.x3x0[ci_]); a.x1 = x1 ? r.at(r.P.x3x1[ci_]) : nullptr; a.dz = r.at(r.P.x3dz[ci_]);
        a.B = 3 * r.B; a.db = nullptr; a.x3 = two ? -r.B : 0;
        wgrad_partial_elems(c.mode, c.Cin, c.Cout, 3 * M, &a.ksplit, &a.mchunk, r.group);
        RC(launch_wgrad(MPU_BF16, c.mode, a, r.grads + c.w, r.st, q, (r.group && q) ? &r.grp : nullptr));
        if (two && r.grp.ntaps != ntaps_before + 1)
            return fail(MPU_EINVAL, "%s", "conv_wgrad (bf16x3): a layer split into two planes did not become a wgrad_taps job");
        return MPU_OK;
    }
    const int rc = launch_wgrad(r.m->cfg.dtype, c.mode, a, r.grads + c.w, r.st, q, (r.group && q) ? &r.grp : nullptr);
    if (!rc && r.grp.ntaps > ntaps_before) { if (r.late.size() != r.m->conv.size()) r.late.assign(r.m->conv.size(), 0); r.late[ci_] = 1; }
    return rc;
}

// test aid: report a non-convolution launch to the tap (kinds 3-7 of mpu_launch_info)
void tap_aux(const Run& r, int kind, int index, int lvl, int C0, int Cout, const void* in0, const void* in1, const void* dz,
             const void* mask, const void* out, long w_off, long b_off, const void* aux0 = nullptr, const void* aux1 = nullptr) {
    if (!r.m->tap) return;
    mpu_launch_info li{};
    li.kind = kind; li.conv_index = index; li.mode = 0; li.dtype = r.m->cfg.dtype;
    li.B = r.B; li.H = r.m->cfg.H >> lvl; li.W = r.m->cfg.W >> lvl; li.C0 = C0; li.C1 = 0; li.Cout = Cout;
    li.in0 = in0; li.in1 = in1; li.dz = dz; li.mask = mask; li.out = out; li.w_off = w_off; li.b_off = b_off;
    li.aux0 = aux0; li.aux1 = aux1;
    r.m->tap(r.m->tap_user, &li);
}

// Round 6: the training step's head as the three head_bn_* passes (unet_ops.hip): bf16 or bf16x3 (accumulator mode), a 64-channel last block
// feeding a softmax head, no launch tap (the replay tests check the unfused kernels launch by launch)
bool head_train_fused_wanted(const Run& r) {
    const mpu_unet* m = r.m;
    return env(ENV_HEAD_TRAIN_FUSED) != 0 && r.acc_mode && !m->tap && m->cfg.depth > 0 && m->cfg.softmax &&
           m->F[0] == 64 && head_train_fused_shape_ok(m->cfg.dtype, m->head_C, m->cfg.n_classes);
}

int bn_fwd(const Run& r, const BN& b, const void* x, int lvl, int training, void* y, void* pooled, int stats_rows = 0) {
    const int H = r.m->cfg.H >> lvl, W = r.m->cfg.W >> lvl;
    const long M = (long)r.B * H * W;
    int rc;
    if (training && stats_rows == -1) { // the producer added its sums to this BatchNorm's accumulator: no finalize at all
        rc = launch_bn_fold_fwd(r.m->cfg.dtype, x, r.B, H, W, b.C, nullptr, -1, r.params + b.g, r.params + b.b, r.state + b.mm,
                                r.state + b.mv, r.stat(b, 0), r.stat(b, 1), r.stat(b, 2), r.stat(b, 3), BN_EPS, BN_MOM, y, pooled, r.st,
                                r.acc_f(b), BN_ACC_F);
        if (rc < 0) return rc;
        if (rc != 1) return fail(MPU_EINVAL, "%s", "bn_fwd: the folded kernel refused an accumulator-mode BatchNorm");
        tap_aux(r, 3, (int)(&b - &r.m->bn[0]), lvl, b.C, b.C, x, nullptr, nullptr, pooled, y, b.g, b.b, r.stat(b, 0), r.stat(b, 1));
        return MPU_OK;
    }
    if (training && stats_rows > 0) {   // few enough rows: finalize folded into the apply pass (one launch)
        rc = launch_bn_fold_fwd(r.m->cfg.dtype, x, r.B, H, W, b.C, (const float*)r.at(r.P.partial), stats_rows, r.params + b.g,
                                r.params + b.b, r.state + b.mm, r.state + b.mv, r.stat(b, 0), r.stat(b, 1), r.stat(b, 2), r.stat(b, 3),
                                BN_EPS, BN_MOM, y, pooled, r.st);
        if (rc < 0) return rc;
        if (rc == 1) {
            tap_aux(r, 3, (int)(&b - &r.m->bn[0]), lvl, b.C, b.C, x, nullptr, nullptr, pooled, y, b.g, b.b, r.stat(b, 0), r.stat(b, 1));
            return MPU_OK;
        }
    }
    if (training)   // stats_rows > 0: the producing conv already wrote that many partial rows (sum, sum of squares)
        rc = launch_bn_stats(r.m->cfg.dtype, x, M, b.C, (float*)r.at(r.P.partial), r.params + b.g, r.params + b.b,
                             r.state + b.mm, r.state + b.mv, r.stat(b, 0), r.stat(b, 1), r.stat(b, 2), r.stat(b, 3),
                             BN_EPS, BN_MOM, stats_rows, r.st);
    else
        rc = launch_bn_infer_coeffs(r.params + b.g, r.params + b.b, r.state + b.mm, r.state + b.mv, b.C, BN_EPS,
                                    r.stat(b, 2), r.stat(b, 3), r.st);
    if (rc) return rc;
    rc = launch_bn_apply(r.m->cfg.dtype, x, r.B, H, W, b.C, r.stat(b, 2), r.stat(b, 3), y, pooled, r.st);
    if (!rc && training) tap_aux(r, 3, (int)(&b - &r.m->bn[0]), lvl, b.C, b.C, x, nullptr, nullptr, pooled, y, b.g, b.b, r.stat(b, 0), r.stat(b, 1));
    return rc;
}

int bn_bwd(const Run& r, const BN& b, const void* dn, const void* x, int lvl, void* dz, int ready_rows = 0,
           int ready_colmajor = 0) {
    const long M = (long)r.B * (r.m->cfg.H >> lvl) * (r.m->cfg.W >> lvl);
    const int rc = launch_bn_backward(r.m->cfg.dtype, dn, x, M, b.C, (float*)r.at(r.P.partial), r.params + b.g, r.stat(b, 0),
                                      r.stat(b, 1), r.grads + b.g, r.grads + b.b, (float*)r.at(r.P.coeffs), dz, ready_rows,
                                      ready_colmajor, r.st, (r.acc_mode && !(b.C & 63)) ? r.acc_b(b) : nullptr, BN_ACC_B);
    if (!rc) tap_aux(r, 4, (int)(&b - &r.m->bn[0]), lvl, b.C, b.C, dn, x, nullptr, nullptr, dz, b.g, b.b, r.stat(b, 0), r.stat(b, 1));
    return rc;
}


// inference: BatchNormalization (moving statistics) is a per-channel affine after the ReLU; it is applied in
// the epilogue of the producing conv (coefficients from mpu_unet_prepare_inference), so no BN kernel runs.
int run_forward_infer(const Run& r, const float* d_x, float* d_out) {
    const mpu_unet* m = r.m; const int D = m->cfg.depth; const Plan& P = r.P;
    const int dt = m->cfg.dtype;
    const long M0 = (long)r.B * m->cfg.H * m->cfg.W;
    const float* co = (const float*)(r.packed + m->infer_off);
    auto sc = [&](const BN& b) { return co + b.st + 2L * b.C; };
    auto sh = [&](const BN& b) { return co + b.st + 3L * b.C; };
    RC(launch_cast_pad(dt, d_x, M0, m->cfg.n_channels, m->cin_pad, r.at(P.xin), r.st));
    const void* cur = r.at(P.xin); int Ccur = m->cin_pad;
    for (int i = 0; i < D; ++i) {
        const BN& b = m->bn[m->enc_bn(i)];
        RC(conv_fwd(r, m->conv[m->enc_c1(i)], cur, Ccur, nullptr, 0, r.at(P.c1[i]), i));
        int pooled_done = 0;
        RC(conv_fwd(r, m->conv[m->enc_c2(i)], r.at(P.c1[i]), m->F[i], nullptr, 0, r.at(P.n[i]), i, sc(b), sh(b), nullptr,
                    r.at(P.p[i]), &pooled_done));
        if (!pooled_done) RC(launch_maxpool(dt, r.at(P.n[i]), r.B, m->cfg.H >> i, m->cfg.W >> i, m->F[i], r.at(P.p[i]), r.st));
        cur = r.at(P.p[i]); Ccur = m->F[i];
    }
    {
        const BN& b = m->bn[m->bot_bn()];
        RC(conv_fwd(r, m->conv[m->bot_c1()], cur, Ccur, nullptr, 0, r.at(P.c1b), D));
        RC(conv_fwd(r, m->conv[m->bot_c2()], r.at(P.c1b), m->F[D], nullptr, 0, r.at(P.nb), D, sc(b), sh(b)));
    }
    const void* prev = r.at(P.nb); int Cprev = m->F[D];
    for (int j = 0; j < D; ++j) {
        const int lvl = D - 1 - j, f = m->F[lvl];
        const BN& b1 = m->bn[m->up_bn(j, 0)]; const BN& b2 = m->bn[m->up_bn(j, 1)];
        RC(conv_fwd(r, m->conv[m->up_c(j, 0)], prev, Cprev, nullptr, 0, r.at(P.n1[j]), lvl, sc(b1), sh(b1)));
        RC(conv_fwd(r, m->conv[m->up_c(j, 1)], r.at(P.n[lvl]), f, r.at(P.n1[j]), f, r.at(P.c2u[j]), lvl));
        // the last conv of the up path: the 1x1 head rides in its epilogue when the schedule can (conv_ws); the partial
        // logits [2][M][K] f32 go to the level-0 conv1 buffer, which is dead by now (128 B per pixel >= 8 K bytes)
        int head_done = 0;
        HeadFuse hf{r.params + m->head_w, m->cfg.n_classes, m->cfg.n_classes, (float*)r.at(P.c1[0]), &head_done};
        const bool last = j == D - 1 && D > 0 && m->head_C == f && (long)m->cfg.n_classes * 8 <= (long)m->F[0] * r.esz;
        RC(conv_fwd(r, m->conv[m->up_c(j, 2)], r.at(P.c2u[j]), f, nullptr, 0, r.at(P.n2[j]), lvl, sc(b2), sh(b2), nullptr,
                    nullptr, nullptr, last ? &hf : nullptr));
        prev = r.at(P.n2[j]); Cprev = f;
        if (head_done) {
            float* out = d_out ? d_out : (float*)r.at(P.probs);
            return launch_head_combine(hf.partial, M0, m->cfg.n_classes, r.params + m->head_b, m->cfg.softmax, out, r.st);
        }
    }
    float* out = d_out ? d_out : (float*)r.at(P.probs);
    return launch_head_forward(dt, prev, M0, m->head_C, m->cfg.n_classes, r.params + m->head_w, m->cfg.n_classes,
                               r.params + m->head_b, m->cfg.softmax, out, r.st);
}

int run_forward(const Run& r, const float* d_x, int training, float* d_out) {
    if (!training) return run_forward_infer(r, d_x, d_out);
    const mpu_unet* m = r.m; const int D = m->cfg.depth; const Plan& P = r.P;
    const long M0 = (long)r.B * m->cfg.H * m->cfg.W;
    m->head_fused_fwd = 0;
    // accumulator mode of the fused BatchNorm statistics: offered where the folded kernel takes the shape; the accumulators of BOTH
    // passes are zeroed by the step's first launch (a backward pass belongs to exactly one training forward)
    RC(launch_cast_pad(m->cfg.dtype, d_x, M0, m->cfg.n_channels, m->cin_pad, r.at(P.xin), r.st,
                       r.acc_mode ? (long long*)r.at(P.bnacc) : nullptr, P.bnacc_elems));
    auto accf = [&](const BN& b, int lvl, bool pooled) -> long long* {
        return (r.acc_mode && bn_fold_shape_ok(b.C, m->cfg.H >> lvl, m->cfg.W >> lvl, pooled)) ? r.acc_f(b) : nullptr;
    };
    const void* cur = r.at(P.xin); int Ccur = m->cin_pad;
    for (int i = 0; i < D; ++i) {
        RC(conv_fwd(r, m->conv[m->enc_c1(i)], cur, Ccur, nullptr, 0, r.at(P.c1[i]), i));
        int rows = 0;
        RC(conv_fwd(r, m->conv[m->enc_c2(i)], r.at(P.c1[i]), m->F[i], nullptr, 0, r.at(P.c2[i]), i, nullptr, nullptr, &rows,
                    nullptr, nullptr, nullptr, accf(m->bn[m->enc_bn(i)], i, true)));
        RC(bn_fwd(r, m->bn[m->enc_bn(i)], r.at(P.c2[i]), i, training, r.at(P.n[i]), r.at(P.p[i]), rows));
        cur = r.at(P.p[i]); Ccur = m->F[i];
    }
    RC(conv_fwd(r, m->conv[m->bot_c1()], cur, Ccur, nullptr, 0, r.at(P.c1b), D));
    {
        int rows = 0;
        RC(conv_fwd(r, m->conv[m->bot_c2()], r.at(P.c1b), m->F[D], nullptr, 0, r.at(P.c2b), D, nullptr, nullptr, &rows,
                    nullptr, nullptr, nullptr, accf(m->bn[m->bot_bn()], D, false)));
        RC(bn_fwd(r, m->bn[m->bot_bn()], r.at(P.c2b), D, training, r.at(P.nb), nullptr, rows));
    }
    const void* prev = r.at(P.nb); int Cprev = m->F[D];
    for (int j = 0; j < D; ++j) {
        const int lvl = D - 1 - j, f = m->F[lvl];
        int rows = 0;
        RC(conv_fwd(r, m->conv[m->up_c(j, 0)], prev, Cprev, nullptr, 0, r.at(P.u1[j]), lvl, nullptr, nullptr, &rows,
                    nullptr, nullptr, nullptr, accf(m->bn[m->up_bn(j, 0)], lvl, false)));
        RC(bn_fwd(r, m->bn[m->up_bn(j, 0)], r.at(P.u1[j]), lvl, training, r.at(P.n1[j]), nullptr, rows));
        RC(conv_fwd(r, m->conv[m->up_c(j, 1)], r.at(P.n[lvl]), f, r.at(P.n1[j]), f, r.at(P.c2u[j]), lvl));
        rows = 0;
        RC(conv_fwd(r, m->conv[m->up_c(j, 2)], r.at(P.c2u[j]), f, nullptr, 0, r.at(P.c3u[j]), lvl, nullptr, nullptr, &rows,
                    nullptr, nullptr, nullptr, accf(m->bn[m->up_bn(j, 1)], lvl, false)));
        if (j == D - 1 && rows == -1 && head_train_fused_wanted(r)) { m->head_fused_fwd = 1; break; }     // (no n2 of the last block)
        RC(bn_fwd(r, m->bn[m->up_bn(j, 1)], r.at(P.c3u[j]), lvl, training, r.at(P.n2[j]), nullptr, rows));
        prev = r.at(P.n2[j]); Cprev = f;
    }
    float* out = d_out ? d_out : (float*)r.at(P.probs);
    if (m->head_fused_fwd) {
        const BN& b = m->bn[m->up_bn(D - 1, 1)];
        RC(launch_head_bn_forward(m->cfg.dtype, r.at(P.c3u[D - 1]), M0, r.acc_f(b), BN_ACC_F, r.params + b.g, r.params + b.b, r.state + b.mm,
                                  r.state + b.mv, r.stat(b, 0), r.stat(b, 1), r.stat(b, 2), r.stat(b, 3), BN_EPS, BN_MOM,
                                  m->cfg.n_classes, r.params + m->head_w, m->cfg.n_classes, r.params + m->head_b, m->cfg.softmax, out, r.st));
    } else {
        RC(launch_head_forward(m->cfg.dtype, prev, M0, m->head_C, m->cfg.n_classes, r.params + m->head_w,
                               m->cfg.n_classes, r.params + m->head_b, m->cfg.softmax, out, r.st));
        tap_aux(r, 6, -1, 0, m->head_C, m->cfg.n_classes, prev, nullptr, nullptr, nullptr, out, m->head_w, m->head_b);
    }
    if (training && d_out)     // keep a copy for the backward pass
        MPU_CHECK_HIP(hipMemcpyAsync(r.at(P.probs), d_out, M0 * m->cfg.n_classes * 4, hipMemcpyDeviceToDevice, r.st));
    return MPU_OK;
}

// gradient-ready point k (see mpu_unet_grad_ready_points): everything the backward pass will write at or
// above that offset of the flat gradient buffer has been enqueued; record the caller's event there
int mark_ready(const Run& r, int k) {
    if (!r.ready_events || k >= r.n_ready || !r.ready_events[k]) return MPU_OK;
    int rc = flush_wgrad_group(r.m->x3 ? MPU_BF16 : r.m->cfg.dtype, r.grp, r.st);     // the gradients above this point must be final before the event
    if (rc) return rc;
    rc = flush_wgrad_reduces(r.rq, r.st);
    if (rc) return rc;
    rc = flush_db(r);
    if (rc) return rc;
    MPU_CHECK_HIP(hipEventRecord((hipEvent_t)r.ready_events[k], r.st));
    return MPU_OK;
}

// The optimizer of mpu_unet_backward_adam (Keras Adam on the flat buffers + refresh of the packed operands)
struct AdamOpt { float* params; void* packed; float* am; float* av; long long* step; long long t; double lr, b1, b2; float eps; };
int finish_backward(const Run& r, const AdamOpt* opt);
// the overlapped tail is decided BEFORE the pass (the device step counter then moves at its start): the switch, the dtype
// and the grouped weight gradients it rides beside; whether a layer actually takes wgrad_taps is known only at the end --
// finish_backward handles "none did" with the same pre-advanced counter
bool tail_overlap_wanted(const Run& r) { return env(ENV_TAIL_OVERLAP) != 0 && r.m->cfg.dtype == MPU_BF16 && r.group; }

int run_backward(const Run& r, const uint8_t* d_y, const float* d_sw, float* d_loss, const AdamOpt* opt = nullptr) {
    const mpu_unet* m = r.m; const int D = m->cfg.depth; const Plan& P = r.P;
    const int dt = m->cfg.dtype;
    const long M0 = (long)r.B * m->cfg.H * m->cfg.W;
    void* gA = r.at(P.gA); void* gB = r.at(P.gB);
    const void* last = D > 0 ? r.at(P.n2[D - 1]) : r.at(P.nb);
    const bool head_fused = m->head_fused_fwd != 0;
    if (head_fused) {           // (the forward pass left no n2 of the last block: head_bn_* passes over c3 instead)
        const BN& b = m->bn[m->up_bn(D - 1, 1)];
        const void* c3 = r.at(P.c3u[D - 1]);
        float* tsum = (float*)r.at(P.partial) + P.partial_floats;
        const long ppi = (long)m->cfg.H * m->cfg.W;
        RC(launch_head_bn_backward(dt, c3, (const float*)r.at(P.probs), d_y, d_sw, M0, ppi, m->cfg.n_classes, r.stat(b, 0), r.stat(b, 1),
                                   (float*)r.at(P.partial), tsum, r.grads + m->head_b, d_loss, r.st,
                                   (opt && opt->step && tail_overlap_wanted(r)) ? opt->step : nullptr, (float*)r.at(P.loss_mean)));
        RC(launch_head_bn_bwd_apply(dt, c3, (const float*)r.at(P.probs), d_y, d_sw, M0, ppi, m->cfg.n_classes, r.params + m->head_w,
                                    m->cfg.n_classes, tsum, r.grads + m->head_b, r.params + b.g, r.params + b.b, r.stat(b, 0), r.stat(b, 1),
                                    r.grads + b.g, r.grads + b.b, r.grads + m->head_w, (float*)r.at(P.coeffs),
                                    r.at(P.dz[m->up_c(D - 1, 2)]), r.st));
    } else {
        RC(launch_head_backward(dt, last, (const float*)r.at(P.probs), d_y, d_sw, M0, (long)m->cfg.H * m->cfg.W,
                                m->head_C, m->cfg.n_classes, r.params + m->head_w, m->cfg.n_classes,
                                (float*)r.at(P.partial), gA, r.grads + m->head_w, r.grads + m->head_b, d_loss, r.st,
                                (opt && opt->step && tail_overlap_wanted(r)) ? opt->step : nullptr, (float*)r.at(P.loss_mean)));
        tap_aux(r, 7, -1, 0, m->head_C, m->cfg.n_classes, last, r.at(P.probs), d_y, d_sw, gA, m->head_w, m->head_b);
    }
    if (m->x3) RC(x3_presplit_inputs(r));
    int point = 0;
    RC(mark_ready(r, point++));                                                            // head
    // (weight gradients on a side stream next to the data gradients were measured twice -- 3.08 vs 3.03 ms in round 2 --
    // and removed in round 3: both kernels need a whole CU's LDS, so they never share one)
    int rowsA = 0;       // partial rows of BN-backward sums already produced for the dn in gA (0: head_backward wrote it)
    auto DZ = [&](int ci) { return r.at(P.dz[ci]); };            // every conv's dz lives in its own buffer until the pass ends
    for (int j = D - 1; j >= 0; --j) {
        const int lvl = D - 1 - j, f = m->F[lvl];
        const int iu = m->up_c(j, 0), i2 = m->up_c(j, 1), i3 = m->up_c(j, 2);
        const Conv& cu = m->conv[iu]; const Conv& c2 = m->conv[i2]; const Conv& c3 = m->conv[i3];
        const void* prev = j > 0 ? r.at(P.n2[j - 1]) : r.at(P.nb);
        const int Cprev = j > 0 ? m->F[lvl + 1] : m->F[D];
        if (!(head_fused && j == D - 1))                                                   // (fused head: dz3 of the last block is written)
            RC(bn_bwd(r, m->bn[m->up_bn(j, 1)], gA, r.at(P.c3u[j]), lvl, DZ(i3), rowsA, 1));   // dz3
        RC(conv_wgrad(r, c3, r.at(P.c2u[j]), f, nullptr, 0, DZ(i3), lvl));
        RC(conv_dgrad(r, c3, DZ(i3), r.at(P.c2u[j]), DZ(i2), lvl, 0, f));                  // dz2
        RC(conv_wgrad(r, c2, r.at(P.n[lvl]), f, r.at(P.n1[j]), f, DZ(i2), lvl));
        RC(conv_dgrad(r, c2, DZ(i2), nullptr, r.at(P.dskip[lvl]), lvl, 0, f));             // d skip
        int rowsB = 0;                                                                     // (BN-backward sums from the epilogue)
        RC(conv_dgrad(r, c2, DZ(i2), nullptr, gB, lvl, f, f, &m->bn[m->up_bn(j, 0)], r.at(P.u1[j]), &rowsB));   // d n1 -> gB
        RC(bn_bwd(r, m->bn[m->up_bn(j, 0)], gB, r.at(P.u1[j]), lvl, DZ(iu), rowsB, 1));    // dz of the up-conv
        RC(conv_wgrad(r, cu, prev, Cprev, nullptr, 0, DZ(iu), lvl));
        // d prev -> gA: the dn of the previous block's second BatchNorm (or of the bottom one)
        const BN& pbn = j > 0 ? m->bn[m->up_bn(j - 1, 1)] : m->bn[m->bot_bn()];
        const void* pbx = j > 0 ? r.at(P.c3u[j - 1]) : r.at(P.c2b);
        RC(conv_dgrad(r, cu, DZ(iu), nullptr, gA, lvl + 1, 0, Cprev, &pbn, pbx, &rowsA));
        RC(mark_ready(r, point++));                                                        // up block j
    }
    {   // bottom
        const int i1 = m->bot_c1(), i2 = m->bot_c2();
        const Conv& c1 = m->conv[i1]; const Conv& c2 = m->conv[i2];
        const void* xin = D > 0 ? r.at(P.p[D - 1]) : r.at(P.xin);
        const int Cx = D > 0 ? m->F[D - 1] : m->cin_pad;
        RC(bn_bwd(r, m->bn[m->bot_bn()], gA, r.at(P.c2b), D, DZ(i2), D > 0 ? rowsA : 0, 1));
        RC(conv_wgrad(r, c2, r.at(P.c1b), m->F[D], nullptr, 0, DZ(i2), D));
        RC(conv_dgrad(r, c2, DZ(i2), r.at(P.c1b), DZ(i1), D, 0, m->F[D]));
        RC(conv_wgrad(r, c1, xin, Cx, nullptr, 0, DZ(i1), D));
        if (D > 0) RC(conv_dgrad(r, c1, DZ(i1), nullptr, gB, D, 0, Cx));                   // d pooled -> gB
        RC(mark_ready(r, point++));                                                        // bottom
    }
    for (int i = D - 1; i >= 0; --i) {
        const int i1 = m->enc_c1(i), i2 = m->enc_c2(i);
        const Conv& c1 = m->conv[i1]; const Conv& c2 = m->conv[i2];
        const int H = m->cfg.H >> i, W = m->cfg.W >> i;
        // skip gradient + un-pooled gradient, and in the same pass the BN-backward sums of the result
        const BN& eb = m->bn[m->enc_bn(i)];
        int bwd_rows = 0;
        // Round 6 (MPU_POOL_BWD_RECOMPUTE): neither the post-BatchNorm tensor n is read nor the summed gradient written -- both
        // passes recompute them from the BatchNorm input, the skip gradient and the pooled gradient (same bits; no launch tap:
        // the replay tests check the two-tensor kernels)
        int fused_pool = 0;
        // (from 4 M elements: measured R6av at configs[1] -- levels 0 / 1 / 2 50.0 -> 41.6, 28.7 -> 26.4, 17.2 -> 16.2 us; the
        //  2 M elements x 512 channels of level 3 lose 1.6 us to the all-channel coefficient prologue of the second pass)
        if (env(ENV_POOL_BWD_RECOMPUTE) != 0 && r.acc_mode && !(eb.C & 63) && !m->tap &&
            (long)r.B * H * W * m->F[i] >= (env(ENV_POOL_BWD_RECOMPUTE) > 1 ? 0L : 4L << 20)) {
            fused_pool = launch_maxpool_bwd_bn(dt, r.at(P.dskip[i]), gB, r.B, H, W, m->F[i], r.at(P.c2[i]), r.stat(eb, 0), r.stat(eb, 1),
                                               r.stat(eb, 2), r.stat(eb, 3), r.params + eb.g, r.grads + eb.g, r.grads + eb.b,
                                               (float*)r.at(P.coeffs), DZ(i2), r.acc_b(eb), BN_ACC_B, r.st);
            if (fused_pool < 0) return fused_pool;
        }
        if (!fused_pool) {
            RC(launch_maxpool_bwd_add_stats(dt, r.at(P.n[i]), r.at(P.dskip[i]), gB, r.B, H, W, m->F[i], gA, r.at(P.c2[i]),
                                            r.stat(eb, 0), r.stat(eb, 1), (float*)r.at(P.partial), P.partial_floats, &bwd_rows,
                                            r.st, (r.acc_mode && !(eb.C & 63)) ? r.acc_b(eb) : nullptr, BN_ACC_B));
            tap_aux(r, 5, m->enc_bn(i), i, m->F[i], m->F[i], r.at(P.n[i]), r.at(P.dskip[i]), gB, nullptr, gA, 0, 0);
            RC(bn_bwd(r, eb, gA, r.at(P.c2[i]), i, DZ(i2), bwd_rows));
        }
        RC(conv_wgrad(r, c2, r.at(P.c1[i]), m->F[i], nullptr, 0, DZ(i2), i));
        RC(conv_dgrad(r, c2, DZ(i2), r.at(P.c1[i]), DZ(i1), i, 0, m->F[i]));
        const void* xin = i > 0 ? r.at(P.p[i - 1]) : r.at(P.xin);
        const int Cx = i > 0 ? m->F[i - 1] : m->cin_pad;
        RC(conv_wgrad(r, c1, xin, Cx, nullptr, 0, DZ(i1), i));
        if (i > 0) RC(conv_dgrad(r, c1, DZ(i1), nullptr, gB, i, 0, Cx));
        RC(mark_ready(r, point++));                                                        // encoder level i
    }
    // the deferred weight-gradient kernels of the whole pass as grouped launches, then their reductions in one more
    // (with an optimizer: the part of it whose gradients are final early runs BESIDE the last grouped launch)
    return finish_backward(r, opt);
}

// per-device side stream + fork / join events of the tail overlap (created at the first use: run one eager step before
// capturing a graph, as UNet.make_graphed_train_step does)
struct SideStream { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
int side_stream(SideStream** out) {
    static SideStream side[64];
    int dev = 0;
    MPU_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev > 63) return fail(MPU_EINVAL, "%s", "side_stream: device index out of range");
    SideStream& d = side[dev];
    if (!d.s) {
        // NORMAL priority. A high-priority side stream was measured (gpurun R6n): eager launches gain nothing (2.534 ms per step
        // either way), and once the library has used a high-priority stream in a process, every REPLAY of a captured step takes
        // 6.5 instead of 2.55 ms -- whichever stream the capture itself forked onto.
        hipStream_t s; hipEvent_t f, j;
        MPU_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        MPU_CHECK_HIP(hipEventCreateWithFlags(&f, hipEventDisableTiming));
        MPU_CHECK_HIP(hipEventCreateWithFlags(&j, hipEventDisableTiming));
        d.fork = f; d.join = j; d.s = s;
    }
    *out = &d;
    return MPU_OK;
}

// dev aid (mpu_debug_tail_events): timing events at the branch points of the tail, recorded on BOTH streams -- the one way to
// see the two branches' timeline without rocprofv3, whose per-dispatch signals change how the queues interleave
constexpr int TAIL_EVENTS = 8;
bool g_tail_events_on = false;
hipEvent_t g_tail_ev[TAIL_EVENTS] = {nullptr};
inline void tail_stamp(int k, hipStream_t st) { if (g_tail_events_on && g_tail_ev[k]) (void)hipEventRecord(g_tail_ev[k], st); }

void pack_jobs_of(const mpu_unet* m, PackTable& tab) {
    tab.njobs = 0; tab._pad = 0;
    for (const Conv& c : m->conv) {
        if (c.mode == CONV1 || tab.njobs >= PACK_MAX_JOBS) continue;
        PackJob& j = tab.job[tab.njobs++];
        j.mode = c.mode; j.Cin = c.Cin; j.Cout = c.Cout; j.unit_begin = j.fwd_units = j._pad = 0;
        j.w = c.w; j.wf = c.wf; j.wd = c.wd;
    }
}

// Round 6. The weight gradients of the high-resolution levels (wgrad_taps_group, ~13 % of a configs[1] step) are bound by
// MFMA issue and LDS reads; the optimizer (adam_pack_all, ~7 %) by HBM. The gradients of the deep levels -- the layers that do
// NOT take the wgrad_taps schedule: 86 % of the parameters at configs[1] -- are final once wgrad_glds_group and its
// reductions have run, so: glds group -> its reductions -> fork: [side stream: Adam + pack of that contiguous parameter range
// with the co-resident lean kernel] beside [main: wgrad_taps_group -> its reductions -> Adam + pack of the remaining
// parameters] -> join -> step counter. Same kernels' arithmetic, disjoint parameter ranges: bit-identical to the serial
// order (tests/test_gpu_unet.py); MPU_TAIL_OVERLAP=0 runs the serial order. Under stream capture the fork / join become
// parallel branches of the graph.
int finish_backward(const Run& r, const AdamOpt* opt) {
    const mpu_unet* m = r.m; const int dt = m->cfg.dtype;
    const int wdt = m->x3 ? MPU_BF16 : dt;                       // (dtype "bf16x3": the grouped weight gradients are bf16 jobs)
    RC(flush_db(r));
    if (!opt) {
        RC(flush_wgrad_group(wdt, r.grp, r.st));
        return flush_wgrad_reduces(r.rq, r.st);
    }
    PackTable jobs; pack_jobs_of(m, jobs);
    if (!tail_overlap_wanted(r)) {                               // the serial order: the step counter moves behind the update
        RC(flush_wgrad_group(wdt, r.grp, r.st));
        RC(flush_wgrad_reduces(r.rq, r.st));
        return launch_adam_pack_all(m->x3 ? MPU_F32X3 : dt, jobs, opt->params, r.grads, opt->am, opt->av, m->n_params, opt->packed,
                                    opt->step, opt->t, opt->lr, opt->b1, opt->b2, opt->eps, r.st);   // (x3: operand words written here)
    }
    // (from here on a device step counter already holds this step's number: launch_head_backward advanced it)
    long lo = 0, hi = 0;                                         // the early range: longest run of convs outside the taps group
    if (r.grp.ntaps > 0 && r.late.size() == m->conv.size()) {
        const int nc = (int)m->conv.size();
        for (int i = 0; i < nc; ) {
            if (r.late[i] || m->conv[i].mode == CONV1) { ++i; continue; }
            int j = i;
            while (j + 1 < nc && !r.late[j + 1] && m->conv[j + 1].mode != CONV1) ++j;
            const long a = m->conv[i].w, b = j + 1 < nc ? m->conv[j + 1].w : m->n_params;
            if (b - a > hi - lo) { lo = a; hi = b; }
            i = j + 1;
        }
    }
    const long all_lo[1] = {0}, all_hi[1] = {m->n_params};
    if (hi - lo < (1L << 20)) {                                  // nothing worth a second stream
        RC(flush_wgrad_group(dt, r.grp, r.st));
        RC(flush_wgrad_reduces(r.rq, r.st));
        return launch_adam_pack_ranges(dt, jobs, opt->params, r.grads, opt->am, opt->av, all_lo, all_hi, 1, opt->packed, opt->step,
                                       opt->t, opt->lr, opt->b1, opt->b2, opt->eps, false, r.st);
    }
    SideStream* sd = nullptr;
    RC(side_stream(&sd));
    tail_stamp(0, r.st);
    RC(flush_wgrad_group(dt, r.grp, r.st, WG_GLDS));
    RC(flush_wgrad_reduces(r.rq, r.st, WG_GLDS));
    tail_stamp(1, r.st);
    MPU_CHECK_HIP(hipEventRecord(sd->fork, r.st));
    MPU_CHECK_HIP(hipStreamWaitEvent(sd->s, sd->fork, 0));
    tail_stamp(2, sd->s);
    RC(launch_adam_pack_ranges(dt, jobs, opt->params, r.grads, opt->am, opt->av, &lo, &hi, 1, opt->packed, opt->step, opt->t, opt->lr,
                               opt->b1, opt->b2, opt->eps, true, sd->s));
    tail_stamp(3, sd->s);
    MPU_CHECK_HIP(hipEventRecord(sd->join, sd->s));
    if (sched_log_on()) sched_note("tail-overlap adam range=[%ld,%ld) of %ld", lo, hi, m->n_params);
    RC(flush_wgrad_group(dt, r.grp, r.st, WG_TAPS));
    tail_stamp(4, r.st);
    RC(flush_wgrad_reduces(r.rq, r.st, WG_ALL));
    tail_stamp(5, r.st);
    const long rest_lo[2] = {0, hi}, rest_hi[2] = {lo, m->n_params};       // everything else in ONE launch
    RC(launch_adam_pack_ranges(dt, jobs, opt->params, r.grads, opt->am, opt->av, rest_lo, rest_hi, 2, opt->packed, opt->step, opt->t,
                               opt->lr, opt->b1, opt->b2, opt->eps, false, r.st));
    tail_stamp(6, r.st);
    MPU_CHECK_HIP(hipStreamWaitEvent(r.st, sd->join, 0));           // the side branch joins at the very end
    tail_stamp(7, r.st);
    return MPU_OK;
}

int make_run(Run& r, const mpu_unet* m, int batch, const float* params, const void* packed, float* state,
             float* grads, void* ws, void* stream) {
    MPU_REQUIRE(m && params && packed && state && ws, "unet: null argument");
    MPU_REQUIRE(batch >= 1, "unet: batch must be >= 1");
    r.m = m; r.B = batch; r.st = (hipStream_t)stream; r.ws = (unsigned char*)ws; r.P = make_plan(m, batch);
    r.params = params; r.packed = (const unsigned char*)packed; r.state = state; r.grads = grads;
    r.esz = m->cfg.dtype == MPU_BF16 ? 2 : 4;
    r.group = env(ENV_WGRAD_GROUP) != 0;      // 0: every weight-gradient kernel as its own launch, in place (A/B)
    // bf16 storage only: a fixed-point unit of 2^-16 on a tile's sum of squares is far below the rounding of the stored bf16 values,
    // but it is visible at the f32 parity mode's level (train-mode logits 1.3e-4 against 4e-5 with the rows: gpurun R6v)
    r.acc_mode = env(ENV_BN_ATOMIC) != 0 && env(ENV_BN_FOLD) != 0 && env(ENV_FUSED_BN_STATS) != 0 && (m->cfg.dtype == MPU_BF16 || m->x3);
    return MPU_OK;
}

}  // namespace

extern "C" {

mpu_unet* mpu_unet_create(const mpu_unet_config* cfg) {
    if (!cfg) { fail(MPU_EINVAL, "%s", "mpu_unet_create: null config"); return nullptr; }
    const int D = cfg->depth;
    if (cfg->n_classes < 1 || cfg->n_classes > 8 || cfg->n_channels < 1 || D < 1 || D > 6 || cfg->H < 1 || cfg->W < 1 ||
        (cfg->H % (1 << D)) || (cfg->W % (1 << D)) || (cfg->dtype != MPU_F32 && cfg->dtype != MPU_BF16 && cfg->dtype != MPU_F32X3)) {
        fail(MPU_EINVAL, "%s", "mpu_unet_create: unsupported configuration (need 1<=n_classes<=8, 1<=depth<=6, "
                               "H and W multiples of 2^depth, dtype f32|bf16|f32x3)");
        return nullptr;
    }
    mpu_unet* m = new mpu_unet();
    m->cfg = *cfg;
    if (cfg->dtype == MPU_F32X3) {
        if (env(ENV_CONV_IMPL) == 0) {      // (the packed operands of this mode are hi | lo words: only the LDS-DMA kernels read them)
            delete m; fail(MPU_EUNSUPPORTED, "%s", "mpu_unet_create: dtype f32x3 is not available under MPU_CONV_IMPL=regs"); return nullptr;
        }
        m->cfg.dtype = MPU_F32; m->x3 = 1;
    }
    m->cin_pad = pad8(cfg->n_channels);
    for (int l = 0; l <= D; ++l) {
        const int fl = cfg->filters[l];
        if (fl < 1) { delete m; fail(MPU_EINVAL, "%s", "mpu_unet_create: filters[] must be positive"); return nullptr; }
        m->Fl.push_back(fl); m->F.push_back(pad8(fl));
    }
    // Flat-buffer layout = Keras layer creation order (unet.py:114-216): each block's BatchNormalization
    // gamma/beta directly follow that block's convs, so that a block's whole gradient (convs AND BN) is final
    // at the block's gradient-ready point (mpu_unet_grad_ready_points; the backward pass completes the buffer
    // from its end towards its start).
    char buf[64];
    int cin = m->cin_pad, lcin = cfg->n_channels;
    for (int i = 0; i < D; ++i) {
        snprintf(buf, sizeof(buf), "encoder_L%d", i);
        add_conv(m, std::string(buf) + "_conv1", CONV3, cin, m->F[i], lcin, m->Fl[i]);
        add_conv(m, std::string(buf) + "_conv2", CONV3, m->F[i], m->F[i], m->Fl[i], m->Fl[i]);
        add_bn(m, std::string(buf) + "_BN", m->F[i], m->Fl[i]);
        cin = m->F[i]; lcin = m->Fl[i];
    }
    add_conv(m, "bottom_conv1", CONV3, cin, m->F[D], lcin, m->Fl[D]);
    add_conv(m, "bottom_conv2", CONV3, m->F[D], m->F[D], m->Fl[D], m->Fl[D]);
    add_bn(m, "bottom_BN", m->F[D], m->Fl[D]);
    cin = m->F[D]; lcin = m->Fl[D];
    for (int j = 0; j < D; ++j) {
        const int lvl = D - 1 - j;
        snprintf(buf, sizeof(buf), "upsample_L%d", j);
        add_conv(m, std::string(buf) + "_conv1", UPCONV2, cin, m->F[lvl], lcin, m->Fl[lvl]);
        add_bn(m, std::string(buf) + "_BN1", m->F[lvl], m->Fl[lvl]);
        add_conv(m, std::string(buf) + "_conv2", CONV3, 2 * m->F[lvl], m->F[lvl], 2 * m->Fl[lvl], m->Fl[lvl]);
        add_conv(m, std::string(buf) + "_conv3", CONV3, m->F[lvl], m->F[lvl], m->Fl[lvl], m->Fl[lvl]);
        add_bn(m, std::string(buf) + "_BN2", m->F[lvl], m->Fl[lvl]);
        cin = m->F[lvl]; lcin = m->Fl[lvl];
    }
    add_conv(m, "conv2d", CONV1, cin, cfg->n_classes, lcin, cfg->n_classes);
    m->head_C = cin; m->head_w = m->conv.back().w; m->head_b = m->conv.back().b;
    m->infer_off = (m->n_packed * (cfg->dtype == MPU_BF16 ? 2 : 4) + 255) / 256 * 256;
    return m;
}

void mpu_unet_destroy(mpu_unet* m) {
    if (!m) return;
    delete m;
}

int mpu_unet_set_launch_tap(mpu_unet* m, mpu_launch_tap_fn fn, void* user) {
    MPU_REQUIRE(m, "mpu_unet_set_launch_tap: null model");
    m->tap = fn; m->tap_user = user;
    return MPU_OK;
}

int64_t mpu_unet_param_floats(const mpu_unet* m) { return m ? m->n_params : 0; }
int64_t mpu_unet_bn_state_floats(const mpu_unet* m) { return m ? m->n_state : 0; }
int64_t mpu_unet_packed_bytes(const mpu_unet* m) {
    if (!m) return 0;
    return m->infer_off + m->n_stats * 4;   // MFMA operands, scale/shift of every BN for inference
}
int64_t mpu_unet_logical_param_count(const mpu_unet* m) { return m ? m->n_logical : 0; }
int32_t mpu_unet_num_tensors(const mpu_unet* m) { return m ? (int32_t)m->tensors.size() : 0; }

int mpu_unet_tensor_info(const mpu_unet* m, int32_t idx, char* name, int32_t name_cap, int32_t* kind,
                         int64_t* offset, int32_t stored_shape[4], int32_t logical_shape[4]) {
    MPU_REQUIRE(m && name && kind && offset && stored_shape && logical_shape, "mpu_unet_tensor_info: null argument");
    MPU_REQUIRE(idx >= 0 && idx < (int)m->tensors.size(), "mpu_unet_tensor_info: index out of range");
    const Tensor& t = m->tensors[idx];
    snprintf(name, name_cap, "%s", t.name.c_str());
    *kind = t.kind; *offset = t.offset;
    for (int i = 0; i < 4; ++i) { stored_shape[i] = t.pshape[i]; logical_shape[i] = t.lshape[i]; }
    return MPU_OK;
}

int64_t mpu_unet_workspace_bytes(const mpu_unet* m, int32_t batch) {
    if (!m || batch < 1) return 0;
    return make_plan(m, batch).total;
}
int64_t mpu_unet_workspace_probs_offset(const mpu_unet* m, int32_t batch) {
    if (!m || batch < 1) return -1;
    return make_plan(m, batch).probs;
}
int64_t mpu_unet_workspace_loss_mean_offset(const mpu_unet* m, int32_t batch) {
    if (!m || batch < 1) return -1;
    return make_plan(m, batch).loss_mean;
}

int mpu_unet_pack_weights(const mpu_unet* m, const float* d_params, void* d_packed, void* stream) {
    MPU_REQUIRE(m && d_params && d_packed, "mpu_unet_pack_weights: null argument");
    PackTable tab; tab.njobs = 0; tab._pad = 0;
    for (const Conv& c : m->conv) {
        if (c.mode == CONV1) continue;
        MPU_REQUIRE(tab.njobs < PACK_MAX_JOBS, "mpu_unet_pack_weights: too many layers");
        PackJob& j = tab.job[tab.njobs++];
        j.mode = c.mode; j.Cin = c.Cin; j.Cout = c.Cout; j.unit_begin = j.fwd_units = j._pad = 0;
        j.w = c.w; j.wf = c.wf; j.wd = c.wd;
    }
    RC(launch_pack_all(m->cfg.dtype, tab, d_params, d_packed, (hipStream_t)stream));
    return m->x3 ? launch_x3_words(d_packed, m->n_packed, (hipStream_t)stream) : MPU_OK;
}

int mpu_unet_adam_pack(const mpu_unet* m, float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t t,
                       int64_t* d_step, double lr, double beta1, double beta2, double eps, void* d_packed, void* stream) {
    MPU_REQUIRE(m && d_params && d_grads && d_m && d_v && d_packed, "mpu_unet_adam_pack: null argument");
    MPU_REQUIRE(d_step || t >= 1, "mpu_unet_adam_pack: need a device step counter or a 1-based step number");
    PackTable tab; tab.njobs = 0; tab._pad = 0;
    for (const Conv& c : m->conv) {
        if (c.mode == CONV1) continue;
        MPU_REQUIRE(tab.njobs < PACK_MAX_JOBS, "mpu_unet_adam_pack: too many layers");
        PackJob& j = tab.job[tab.njobs++];
        j.mode = c.mode; j.Cin = c.Cin; j.Cout = c.Cout; j.unit_begin = j.fwd_units = j._pad = 0;
        j.w = c.w; j.wf = c.wf; j.wd = c.wd;
    }
    return launch_adam_pack_all(m->x3 ? MPU_F32X3 : m->cfg.dtype, tab, d_params, d_grads, d_m, d_v, m->n_params, d_packed,
                                (long long*)d_step, (long long)t, lr, beta1, beta2, (float)eps, (hipStream_t)stream);
}

int mpu_unet_prepare_inference(const mpu_unet* m, const float* d_params, const float* d_bn_state, void* d_packed,
                                void* stream) {
    MPU_REQUIRE(m && d_params && d_bn_state && d_packed, "mpu_unet_prepare_inference: null argument");
    float* co = (float*)((unsigned char*)d_packed + m->infer_off);
    for (const BN& b : m->bn)
        RC(launch_bn_infer_coeffs(d_params + b.g, d_params + b.b, d_bn_state + b.mm, d_bn_state + b.mv, b.C, BN_EPS,
                                  co + b.st + 2L * b.C, co + b.st + 3L * b.C, (hipStream_t)stream));
    return MPU_OK;
}

int mpu_unet_forward(const mpu_unet* m, int32_t batch, const float* d_x, const float* d_params, const void* d_packed,
                     float* d_bn_state, void* d_workspace, int32_t training, float* d_out, void* stream) {
    MPU_REQUIRE(d_x, "mpu_unet_forward: null input");
    Run r;
    RC(make_run(r, m, batch, d_params, d_packed, d_bn_state, nullptr, d_workspace, stream));
    return run_forward(r, d_x, training, d_out);
}

int mpu_unet_backward(const mpu_unet* m, int32_t batch, const uint8_t* d_y, const float* d_sample_weight,
                      const float* d_params, const void* d_packed, float* d_bn_state, void* d_workspace,
                      float* d_grads, float* d_loss, void* stream) {
    MPU_REQUIRE(d_y && d_sample_weight && d_grads, "mpu_unet_backward: null argument");
    MPU_REQUIRE(m && m->cfg.softmax, "mpu_unet_backward: training needs out_activation='softmax'");
    Run r;
    RC(make_run(r, m, batch, d_params, d_packed, d_bn_state, d_grads, d_workspace, stream));
    return run_backward(r, d_y, d_sample_weight, d_loss);
}

int mpu_unet_backward_adam(const mpu_unet* m, int32_t batch, const uint8_t* d_y, const float* d_sample_weight,
                           float* d_params, void* d_packed, float* d_bn_state, void* d_workspace, float* d_grads,
                           float* d_loss, float* d_m, float* d_v, int64_t t, int64_t* d_step, double lr, double beta1,
                           double beta2, double eps, void* stream) {
    MPU_REQUIRE(d_y && d_sample_weight && d_grads && d_m && d_v, "mpu_unet_backward_adam: null argument");
    MPU_REQUIRE(m && m->cfg.softmax, "mpu_unet_backward_adam: training needs out_activation='softmax'");
    MPU_REQUIRE(d_step || t >= 1, "mpu_unet_backward_adam: need a device step counter or a 1-based step number");
    Run r;
    RC(make_run(r, m, batch, d_params, d_packed, d_bn_state, d_grads, d_workspace, stream));
    AdamOpt o{d_params, d_packed, d_m, d_v, (long long*)d_step, (long long)t, lr, beta1, beta2, (float)eps};
    return run_backward(r, d_y, d_sample_weight, d_loss, &o);
}

int mpu_debug_tail_events(int32_t on, float* ms_out) {
    if (on) {
        for (int k = 0; k < TAIL_EVENTS; ++k) if (!g_tail_ev[k]) MPU_CHECK_HIP(hipEventCreate(&g_tail_ev[k]));
        g_tail_events_on = true;
        return MPU_OK;
    }
    g_tail_events_on = false;
    if (ms_out) {
        MPU_CHECK_HIP(hipDeviceSynchronize());
        for (int k = 0; k < TAIL_EVENTS; ++k) {
            float t = -1.f;
            if (g_tail_ev[0] && g_tail_ev[k] && hipEventElapsedTime(&t, g_tail_ev[0], g_tail_ev[k]) != hipSuccess) { t = -1.f; (void)hipGetLastError(); }
            ms_out[k] = t;
        }
    }
    return MPU_OK;
}

int32_t mpu_unet_grad_ready_points(const mpu_unet* m, int64_t* offsets, int32_t cap) {
    if (!m) return 0;
    const int D = m->cfg.depth;
    std::vector<long> pts;
    pts.push_back(m->head_w < m->head_b ? m->head_w : m->head_b);
    for (int j = D - 1; j >= 0; --j) pts.push_back(m->conv[m->up_c(j, 0)].w);
    pts.push_back(m->conv[m->bot_c1()].w);
    for (int i = D - 1; i >= 0; --i) pts.push_back(m->conv[m->enc_c1(i)].w);
    for (size_t k = 0; k < pts.size() && (int)k < cap; ++k) offsets[k] = pts[k];
    return (int32_t)pts.size();
}

int mpu_unet_backward_events(const mpu_unet* m, int32_t batch, const uint8_t* d_y, const float* d_sample_weight,
                             const float* d_params, const void* d_packed, float* d_bn_state, void* d_workspace,
                             float* d_grads, float* d_loss, void* const* ready_events, int32_t n_events,
                             void* stream) {
    MPU_REQUIRE(d_y && d_sample_weight && d_grads, "mpu_unet_backward_events: null argument");
    MPU_REQUIRE(m && m->cfg.softmax, "mpu_unet_backward_events: training needs out_activation='softmax'");
    Run r;
    RC(make_run(r, m, batch, d_params, d_packed, d_bn_state, d_grads, d_workspace, stream));
    r.ready_events = ready_events; r.n_ready = ready_events ? n_events : 0;
    return run_backward(r, d_y, d_sample_weight, d_loss);
}

int mpu_unet_l2_regularizer(const mpu_unet* m, const float* d_params, float* d_grads, double l2, double* d_partial,
                            float* d_reg_loss, void* stream) {
    MPU_REQUIRE(m && d_params && d_grads, "mpu_unet_l2_regularizer: null argument");
    MPU_REQUIRE(l2 >= 0.0 && (!d_reg_loss || d_partial), "mpu_unet_l2_regularizer: bad argument");
    L2Table tab; tab.njobs = 0; tab._pad = 0;
    for (const Conv& c : m->conv) {
        if (c.mode == CONV1) continue;                      // the 1x1 output conv carries no regulariser (unet.py:211)
        MPU_REQUIRE(tab.njobs < PACK_MAX_JOBS, "mpu_unet_l2_regularizer: too many layers");
        const int k = c.mode == UPCONV2 ? 2 : 3;
        tab.off[tab.njobs] = c.w; tab.n[tab.njobs] = (long)k * k * c.Cin * c.Cout; ++tab.njobs;
    }
    return launch_l2_regularizer(tab, d_params, d_grads, (float)l2, d_partial, d_reg_loss, (hipStream_t)stream);
}
int64_t mpu_unet_l2_workspace_doubles(void) { return L2_PARTIAL_DOUBLES; }

int mpu_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, int64_t t,
                  double lr, double beta1, double beta2, double eps, void* stream) {
    MPU_REQUIRE(d_params && d_grads && d_m && d_v && n >= 0 && t >= 1, "mpu_adam_step: bad argument");
    const double alpha = lr * std::sqrt(1.0 - std::pow(beta2, (double)t)) / (1.0 - std::pow(beta1, (double)t));
    return launch_adam(d_params, d_grads, d_m, d_v, n, (float)alpha, (float)beta1, (float)beta2, (float)eps,
                       (hipStream_t)stream);
}

int mpu_adam_step_device_counter(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n,
                                 int64_t* d_step, double lr, double beta1, double beta2, double eps, void* stream) {
    MPU_REQUIRE(d_params && d_grads && d_m && d_v && d_step && n >= 0, "mpu_adam_step_device_counter: bad argument");
    return launch_adam_dev(d_params, d_grads, d_m, d_v, n, (long long*)d_step, lr, beta1, beta2, (float)eps, (hipStream_t)stream);
}

// ---- op-level entry points (unit tests, integration of single layers) ------
int mpu_conv2d_pack_weights(int32_t dtype, int32_t mode, const float* d_w, int32_t Cin, int32_t Cout,
                            void* d_w_fwd, void* d_w_dgrad, void* stream) {
    MPU_REQUIRE(d_w && d_w_fwd, "mpu_conv2d_pack_weights: null argument");
    MPU_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "mpu_conv2d_pack_weights: channels must be multiples of 8");
    MPU_REQUIRE(mode >= CONV3 && mode <= CONV1, "mpu_conv2d_pack_weights: unknown mode");
    if (dtype == MPU_F32X3) {                 // f32 operands, then the hi | lo words of the split-bf16 kernels
        const long k = mode == UPCONV2 ? 4 : (mode == CONV1 ? 1 : 9);
        RC(launch_pack_weights(MPU_F32, mode, d_w, Cin, Cout, d_w_fwd, d_w_dgrad, (hipStream_t)stream));
        RC(launch_x3_words(d_w_fwd, k * Cin * Cout, (hipStream_t)stream));
        return d_w_dgrad ? launch_x3_words(d_w_dgrad, (mode == CONV1 ? 1L : 9L) * Cin * Cout, (hipStream_t)stream) : MPU_OK;
    }
    return launch_pack_weights(dtype, mode, d_w, Cin, Cout, d_w_fwd, d_w_dgrad, (hipStream_t)stream);
}

static int conv2d_igemm_impl(int32_t dtype, int32_t mode, const void* d_in0, int32_t C0, const void* d_in1, int32_t C1,
                             const void* d_w_packed, int64_t w_tap_stride, int32_t w_row_stride, const float* d_bias,
                             const void* d_mask, void* d_out, int32_t B, int32_t Ho, int32_t Wo, int32_t Cout, int32_t relu,
                             float* d_workspace, int64_t workspace_floats, void* stream) {
    MPU_REQUIRE(d_in0 && d_w_packed && d_out, "mpu_conv2d_igemm: null argument");
    MPU_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && Cout % 8 == 0 && C0 > 0, "mpu_conv2d_igemm: channels must be multiples of 8");
    MPU_REQUIRE((C1 == 0) == (d_in1 == nullptr), "mpu_conv2d_igemm: in1 / C1 mismatch");
    MPU_REQUIRE(mode >= CONV3 && mode <= CONV1, "mpu_conv2d_igemm: unknown mode");
    MPU_REQUIRE(mode != UPCONV2 || (Ho % 2 == 0 && Wo % 2 == 0), "mpu_conv2d_igemm: UPCONV2 needs even output size");
    ConvArgs a;
    a.in0 = d_in0; a.in1 = d_in1; a.C0 = C0; a.C1 = C1; a.w = d_w_packed; a.w_tap_stride = w_tap_stride;
    a.w_row_stride = w_row_stride; a.bias = d_bias; a.mask = d_mask; a.out = d_out;
    a.B = B; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.relu = relu; a.flops = 0; a.w_elems = 0;
    a.partial = d_workspace; a.partial_cap = d_workspace ? workspace_floats : 0; a.ksplit = 1;
    a.stats = nullptr; a.stats_rows = nullptr; a.stats_cap = 0;
    a.bn_x = nullptr; a.bn_mean = nullptr; a.bn_invstd = nullptr;
    a.post_scale = nullptr; a.post_shift = nullptr;
    if (dtype == MPU_F32X3) { a.x3 = 1; dtype = MPU_F32; }
    return launch_conv(dtype, mode, a, (hipStream_t)stream);
}

int mpu_conv2d_igemm(int32_t dtype, int32_t mode, const void* d_in0, int32_t C0, const void* d_in1, int32_t C1,
                     const void* d_w_packed, int64_t w_tap_stride, int32_t w_row_stride, const float* d_bias,
                     const void* d_mask, void* d_out, int32_t B, int32_t Ho, int32_t Wo, int32_t Cout, int32_t relu,
                     void* stream) {
    return conv2d_igemm_impl(dtype, mode, d_in0, C0, d_in1, C1, d_w_packed, w_tap_stride, w_row_stride, d_bias, d_mask,
                             d_out, B, Ho, Wo, Cout, relu, nullptr, 0, stream);
}

int mpu_conv2d_igemm_ws(int32_t dtype, int32_t mode, const void* d_in0, int32_t C0, const void* d_in1, int32_t C1,
                        const void* d_w_packed, int64_t w_tap_stride, int32_t w_row_stride, const float* d_bias,
                        const void* d_mask, void* d_out, int32_t B, int32_t Ho, int32_t Wo, int32_t Cout, int32_t relu,
                        float* d_workspace, int64_t workspace_floats, void* stream) {
    MPU_REQUIRE(d_workspace && workspace_floats > 0, "mpu_conv2d_igemm_ws: null workspace");
    return conv2d_igemm_impl(dtype, mode, d_in0, C0, d_in1, C1, d_w_packed, w_tap_stride, w_row_stride, d_bias, d_mask,
                             d_out, B, Ho, Wo, Cout, relu, d_workspace, workspace_floats, stream);
}

int64_t mpu_conv2d_wgrad_workspace_floats(int32_t mode, int32_t Cin, int32_t Cout, int64_t M) {
    // the whole layer as one job, plus -- a concat layer whose first source is not a multiple of 64 channels runs as one job
    // per source -- room for two half-layer jobs (an upper bound that needs no image shape)
    const int half = ((Cin + 1) / 2 + 7) / 8 * 8;
    return wgrad_partial_elems(mode, Cin, Cout, M, nullptr, nullptr) + 2 * (wgrad_partial_elems(mode, half, Cout, M, nullptr, nullptr) + 128) + 256;
}

int64_t mpu_conv2d_wgrad_job_floats(int32_t dtype, int32_t mode, int32_t B, int32_t Ho, int32_t Wo, int32_t C0, int32_t C1,
                                    int32_t Cout, int32_t grouped) {
    return wgrad_job_floats(dtype, mode, B, Ho, Wo, C0, C1, Cout, grouped != 0);
}
int64_t mpu_conv2d_wgrad_scratch_floats(int32_t dtype, int32_t mode, int32_t B, int32_t Ho, int32_t Wo, int32_t C0, int32_t C1,
                                        int32_t Cout) {
    return wgrad_scratch_need(dtype, mode, B, Ho, Wo, C0, C1, Cout);
}

int mpu_conv2d_wgrad(int32_t dtype, int32_t mode, const void* d_x0, int32_t C0, const void* d_x1, int32_t C1,
                     const void* d_dz, int32_t Cout, int32_t B, int32_t Ho, int32_t Wo, float* d_workspace,
                     float* d_dW, void* stream) {
    MPU_REQUIRE(d_x0 && d_dz && d_workspace && d_dW, "mpu_conv2d_wgrad: null argument");
    MPU_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && Cout % 8 == 0, "mpu_conv2d_wgrad: channels must be multiples of 8");
    WgradArgs a;
    a.x0 = d_x0; a.x1 = d_x1; a.C0 = C0; a.C1 = C1; a.dz = d_dz; a.Cout = Cout; a.partial = d_workspace;
    a.B = B; a.Ho = Ho; a.Wo = Wo; a.flops = 0; a.db = nullptr; a.db_partial = nullptr; a.colsum_scratch = nullptr; a.fuse_db = 0;
    a.c0_logical = 0;
    a.partial_cap = C1 == C0 ? mpu_conv2d_wgrad_workspace_floats(mode, C0 + C1, Cout, (long)B * Ho * Wo) : 0;   // (the query's contract)
    wgrad_partial_elems(mode, C0 + C1, Cout, (long)B * Ho * Wo, &a.ksplit, &a.mchunk);
    if (dtype == MPU_F32X3) { a.x3 = 1; dtype = MPU_F32; }
    ReduceQueue q;                               // second stages recorded, then run as one launch (as the U-Net does)
    int rc = launch_wgrad(dtype, mode, a, d_dW, (hipStream_t)stream, &q);
    if (rc) return rc;
    return flush_wgrad_reduces(q, (hipStream_t)stream);
}

/* first layer: x0 holds n_image_channels (<= 8) real channels in 8-channel pixel records */
int mpu_conv2d_wgrad_first_layer(int32_t dtype, const void* d_x, int32_t n_image_channels, const void* d_dz, int32_t Cout,
                                 int32_t B, int32_t H, int32_t W, float* d_workspace, float* d_dW, float* d_db,
                                 void* stream) {
    MPU_REQUIRE(d_x && d_dz && d_workspace && d_dW, "mpu_conv2d_wgrad_first_layer: null argument");
    MPU_REQUIRE(n_image_channels >= 1 && n_image_channels <= 8 && Cout >= 8 && Cout % 8 == 0 && B >= 1 && H >= 1 && W >= 1,
                "mpu_conv2d_wgrad_first_layer: bad shape");
    WgradArgs a;
    a.x0 = d_x; a.x1 = nullptr; a.C0 = 8; a.C1 = 0; a.dz = d_dz; a.Cout = Cout; a.partial = d_workspace;
    a.B = B; a.Ho = H; a.Wo = W; a.flops = 0; a.db = d_db; a.db_partial = nullptr; a.fuse_db = 0;
    a.c0_logical = n_image_channels;
    const long M = (long)B * H * W;
    long we = wgrad_partial_elems(CONV3, 8, Cout, M, &a.ksplit, &a.mchunk);
    if (we < 2048L * 19 * Cout) we = 2048L * 19 * Cout;
    a.partial_cap = we;                              // what mpu_conv2d_wgrad_first_layer_workspace_floats promised
    a.colsum_scratch = d_workspace + we;             // (the workspace query below includes this tail)
    return launch_wgrad(dtype, CONV3, a, d_dW, (hipStream_t)stream);
}
int64_t mpu_conv2d_wgrad_first_layer_workspace_floats(int32_t Cout, int64_t M) {
    long we = wgrad_partial_elems(CONV3, 8, Cout, M, nullptr, nullptr);
    if (we < 2048L * 19 * Cout) we = 2048L * 19 * Cout;       // wgrad_c8: <= 2048 strips x (9 * 2 * Cout + Cout) floats
    return we + (int64_t)RED_MAX_BLOCKS * Cout;
}

}  // extern "C"
