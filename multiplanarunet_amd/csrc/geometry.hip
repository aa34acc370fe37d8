// Predict-time geometry kernels: plane resampling (trilinear image / nearest
// labels), nearest back-mapping and multi-view fusion. All HBM-bound gathers;
// coordinate math is fp64 with the reference's exact operation order so that
// nearest-neighbour decisions agree with NumPy bit for bit:
//   * 3x3 @ point products follow the BLAS k-ordered FMA chain
//     fma(m2,z, fma(m1,y, m0*x)) (verified against numpy.dot in the survey box);
//   * everything else is plain IEEE mul/add/div (this file is compiled with
//     -ffp-contract=off).
// Reference: mpunet/interpolation/regular_grid_interpolator.py:152-270,
// view_interpolator.py:54-133, sample_grid.py:101-130,192-244,
// utils/fusion/fuse_and_predict.py:92-137, models/fusion_model.py:38-39.
#include <stdlib.h>
#include "common.h"

namespace mpu {

thread_local char g_err[512] = "";

struct Mat3 { double m[9]; };

__device__ __forceinline__ void mat3_apply(const Mat3& M, double x, double y, double z,
                                           double& ox, double& oy, double& oz) {
    ox = fma(M.m[2], z, fma(M.m[1], y, M.m[0] * x));
    oy = fma(M.m[5], z, fma(M.m[4], y, M.m[3] * x));
    oz = fma(M.m[8], z, fma(M.m[7], y, M.m[6] * x));
}

// A coordinate axis: device array + (optionally) a closed form the host has verified to
// reproduce that array bit for bit (mpu_axis), so values can be computed in registers.
struct AxisDev { const double* g; int n, kind; double start, step, last, inv_h, g0; };

__device__ __forceinline__ double axis_at(const AxisDev& a, int i) {
    if (a.kind == 1) return (i == a.n - 1) ? a.last : (double)i * a.step + a.start;   // np.linspace
    if (a.kind == 2) return ((double)i - a.start) * a.step;                            // voxel axis
    return a.g[i];
}

// RegularGridInterpolator._find_indices for one axis: i = searchsorted_left(g,x)-1
// clipped to [0,n-2]; oob = x<g[0] || x>g[n-1]. The cell is found from a reciprocal-spacing
// guess and then fixed up against the ACTUAL axis values, so it is exactly NumPy's cell.
__device__ __forceinline__ int find_cell_index(const AxisDev& a, double x, bool& oob, double& gc, double& gc1) {
    const int n = a.n;
    const double g0 = axis_at(a, 0), gl = axis_at(a, n - 1);
    oob = (x < g0) || (x > gl);
    int c;
    if (!(x > g0)) c = 0;
    else if (x > gl) c = n - 2;
    else {
        c = (int)ceil((x - g0) * a.inv_h) - 1;
        c = c < 0 ? 0 : (c > n - 2 ? n - 2 : c);
    }
    gc = axis_at(a, c); gc1 = axis_at(a, c + 1);
    if (!oob) {
        while (c < n - 2 && gc1 < x) { ++c; gc = gc1; gc1 = axis_at(a, c + 1); }     // need x <= g[c+1]
        while (c > 0 && gc >= x) { --c; gc1 = gc; gc = axis_at(a, c); }              // need g[c] <  x
    }
    return c;
}
// ---- fast paths on uniform axes (closed-form kinds 1, 2) ---------------------------------------------
// u = (x - g[0]) / h in index units carries an error < 1e-11 (|x|, |g| < 1e4: a few ulps of the operands), and the
// actual axis nodes the exact search compares against lie within 1e-11 index units of the integers. So whenever
// u is farther than TAU = 1e-6 from every value at which the exact procedure changes its answer -- an integer
// (searchsorted / out-of-bounds limits g[0], g[n-1]) for the cell search, a half-integer (the y <= 0.5 tie) and
// the two axis ends for the nearest search -- the closed form below IS the exact answer. The remaining samples
// (a fraction ~6e-6) take the exact search. g_fast_geometry = 0 (MPU_GEOM_FAST=0) forces the exact search for
// every sample (A/B and the equality test).
__constant__ int g_fast_geometry_dev = 1;
constexpr double GEOM_TAU = 1e-6;

__device__ __forceinline__ bool cell_fast(const AxisDev& a, double x, int& c, bool& oob) {
    if (a.kind == 0 || !g_fast_geometry_dev) return false;
    const double u = (x - a.g0) * a.inv_h;
    const double f = floor(u), nm1 = (double)(a.n - 1);
    if (!(fabs(u) < 1e9) || (u - f) < GEOM_TAU || (f + 1.0 - u) < GEOM_TAU) return false;   // near a node (or NaN / huge)
    oob = (u < 0.0) || (u > nm1);
    const int ci = (int)f;
    c = ci < 0 ? 0 : (ci > a.n - 2 ? a.n - 2 : ci);
    return true;
}
__device__ __forceinline__ bool nearest_fast(const AxisDev& a, double x, int& n, bool& oob) {
    if (a.kind == 0 || !g_fast_geometry_dev) return false;
    const double u = (x - a.g0) * a.inv_h;
    const double r = rint(u), nm1 = (double)(a.n - 1);
    if (!(fabs(u) < 1e9) || fabs(fabs(u - r) - 0.5) < GEOM_TAU || fabs(u) < GEOM_TAU || fabs(u - nm1) < GEOM_TAU) return false;
    oob = (u < 0.0) || (u > nm1);
    n = (int)r;
    return true;
}

// linear: also the normalised distance y = (x-g[i])/(g[i+1]-g[i]) (IEEE division as NumPy)
__device__ __forceinline__ void find_cell(const AxisDev& a, double x, int& i, double& y, bool& oob) {
    double gc, gc1;
    int c;
    if (cell_fast(a, x, c, oob)) { gc = axis_at(a, c); gc1 = axis_at(a, c + 1); }
    else c = find_cell_index(a, x, oob, gc, gc1);
    i = c;
    y = (x - gc) / (gc1 - gc);
}
// nearest: index of np.where(y <= .5, i, i+1). With num = fl(x-g[i]), den = fl(g[i+1]-g[i]) > 0 and a
// correctly rounded quotient, fl(num/den) <= 0.5  <=>  2*num <= den (0.5 is a double, the next
// double above den is den+ulp > den*(1+2^-53)), so no division is needed.
__device__ __forceinline__ int find_nearest_exact(const AxisDev& a, double x, bool& oob);
__device__ __forceinline__ int find_nearest(const AxisDev& a, double x, bool& oob) {
    int nf;
    if (nearest_fast(a, x, nf, oob)) return nf;
    return find_nearest_exact(a, x, oob);
}
__device__ __forceinline__ int find_nearest_exact(const AxisDev& a, double x, bool& oob) {
    double gc, gc1;
    const int c = find_cell_index(a, x, oob, gc, gc1);
    const double num = x - gc, den = gc1 - gc;
    return (2.0 * num <= den) ? c : c + 1;
}

struct SampleArgs {
    const float* vol; const uint8_t* labels;
    int X, Y, Z, C;
    AxisDev ax, ay, az; const double* offsets;
    Mat3 basis, rot; int has_rot;
    int dim, P; double g_start, g_step;
    const float* bg; uint8_t bg_class;
    const double *center, *scale;
    float* out; uint8_t* out_lab;
};

// one workgroup = one 16x16 patch of one plane (compact footprint in the volume for any view)
__global__ __launch_bounds__(256) void sample_view_planes_kernel(SampleArgs a) {
    const int tpd = (a.dim + 15) / 16;
    const long nblk = (long)a.P * tpd * tpd;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int p = (int)(blk / (tpd * tpd));
        const int ti = (int)((blk / tpd) % tpd), tj = (int)(blk % tpd);
        const int i = ti * 16 + (threadIdx.x >> 4), j = tj * 16 + (threadIdx.x & 15);
        if (i >= a.dim || j >= a.dim) continue;
        const long t = ((long)p * a.dim + i) * a.dim + j;
        const double gx = (double)i * a.g_step + a.g_start;
        const double gy = (double)j * a.g_step + a.g_start;
        const double off = a.offsets[p];
        double rx, ry, rz;
        mat3_apply(a.basis, gx, gy, off, rx, ry, rz);
        if (a.has_rot) {
            double qx, qy, qz;
            mat3_apply(a.rot, rx, ry, rz, qx, qy, qz);
            rx = qx; ry = qy; rz = qz;
        }
        int i0, i1, i2; double y0, y1, y2; bool o0, o1, o2;
        find_cell(a.ax, rx, i0, y0, o0);
        find_cell(a.ay, ry, i1, y1, o1);
        find_cell(a.az, rz, i2, y2, o2);
        const bool oob = o0 || o1 || o2;
        // itertools.product order of the 8 corners, weight = ((1*wx)*wy)*wz
        const double wx[2] = {1.0 - y0, y0}, wy[2] = {1.0 - y1, y1}, wz[2] = {1.0 - y2, y2};
        float* o = a.out + t * a.C;
        // the two z-neighbours of a corner pair are contiguous in memory: one 8-byte (C == 1) or 16-byte (C == 2)
        // load per (x, y) corner instead of two / four scalar gathers; all four issued before the first use
        float cv[2][8];                                    // [channel][corner e]
        const bool paired = a.C <= 2 && !oob;
        if (paired) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long base = (((long)(i0 + (q >> 1)) * a.Y + (i1 + (q & 1))) * a.Z + i2) * a.C;
                if (a.C == 1) {
                    float2 v; __builtin_memcpy(&v, a.vol + base, 8);
                    cv[0][2 * q] = v.x; cv[0][2 * q + 1] = v.y;
                } else {
                    float4 v; __builtin_memcpy(&v, a.vol + base, 16);
                    cv[0][2 * q] = v.x; cv[1][2 * q] = v.y; cv[0][2 * q + 1] = v.z; cv[1][2 * q + 1] = v.w;
                }
            }
        }
        for (int c = 0; c < a.C; ++c) {
            float v32;
            if (oob) {
                v32 = a.bg[c];
            } else {
                double acc = 0.0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ex = e >> 2, ey = (e >> 1) & 1, ez = e & 1;
                    const double w = ((1.0 * wx[ex]) * wy[ey]) * wz[ez];
                    const long idx = (((long)(i0 + ex) * a.Y + (i1 + ey)) * a.Z + (i2 + ez)) * a.C + c;
                    const float val = paired ? cv[c & 1][e] : a.vol[idx];
                    acc = acc + (double)val * w;
                }
                v32 = (float)acc;
            }
            if (a.center) {    // sklearn: X -= center_; X /= scale_ (f64 op, f32 store)
                v32 = (float)((double)v32 - a.center[c]);
                v32 = (float)((double)v32 / a.scale[c]);
            }
            o[c] = v32;
        }
        if (a.out_lab) {
            uint8_t l = a.bg_class;
            if (!oob) {
                const int n0 = (y0 <= .5) ? i0 : i0 + 1;
                const int n1 = (y1 <= .5) ? i1 : i1 + 1;
                const int n2 = (y2 <= .5) ? i2 : i2 + 1;
                l = a.labels[((long)n0 * a.Y + n1) * a.Z + n2];
            }
            a.out_lab[t] = l;
        }
    }
}

// ------------------------------------------------------------------------- //
struct ViewDev {
    Mat3 invb; const float* pred; AxisDev g, offs; int dim, P;
};
struct GridDev { Mat3 A; double c[3]; int X, Y, Z; };

// nearest lookup of one view for voxel (x,y,z): returns element offset of the
// K-vector in pred[P,dim,dim,K], or -1 when out of the view's box; plane index in pl.
__device__ __forceinline__ long view_lookup(const ViewDev& v, double rx, double ry, double rz,
                                            int K, int& pl) {
    double qx, qy, qz;
    mat3_apply(v.invb, rx, ry, rz, qx, qy, qz);
    bool o0, o1, o2;
    int n0, n1, n2;
    const bool f0 = nearest_fast(v.g, qx, n0, o0), f1 = nearest_fast(v.g, qy, n1, o1), f2 = nearest_fast(v.offs, qz, n2, o2);
    if (__builtin_expect(!(f0 && f1 && f2), 0)) {            // ~1e-5 of the lookups: near a tie / an axis end
        n0 = find_nearest_exact(v.g, qx, o0);
        n1 = find_nearest_exact(v.g, qy, o1);
        n2 = find_nearest_exact(v.offs, qz, o2);
    }
    if (o0 || o1 || o2) { pl = -1; return -1; }
    pl = n2;
    return (((long)n2 * v.dim + n0) * v.dim + n1) * K;
}

__device__ __forceinline__ void voxel_real(const GridDev& g, int x, int y, int z, double& rx, double& ry, double& rz) {
    mat3_apply(g.A, (double)x, (double)y, (double)z, rx, ry, rz);
    rx = rx - g.c[0]; ry = ry - g.c[1]; rz = rz - g.c[2];
}

// one workgroup = one 4x4x16 (x,y,z) brick of voxels: compact footprint in every view's
// prediction volume, 16 consecutive z per row of the brick for the stores.
constexpr int BRX = 4, BRY = 4, BRZ = 16;
__device__ __forceinline__ long brick_count(const GridDev& g) {
    return (long)((g.X + BRX - 1) / BRX) * ((g.Y + BRY - 1) / BRY) * ((g.Z + BRZ - 1) / BRZ);
}
__device__ __forceinline__ bool brick_voxel(const GridDev& g, long blk, int& x, int& y, int& z, long& t) {
    const int nz = (g.Z + BRZ - 1) / BRZ, ny = (g.Y + BRY - 1) / BRY;
    const int bz = (int)(blk % nz), by = (int)((blk / nz) % ny), bx = (int)(blk / ((long)nz * ny));
    z = bz * BRZ + (threadIdx.x & 15);
    y = by * BRY + ((threadIdx.x >> 4) & 3);
    x = bx * BRX + (threadIdx.x >> 6);
    t = ((long)x * g.Y + y) * g.Z + z;
    return x < g.X && y < g.Y && z < g.Z;
}

template <int K>
__device__ __forceinline__ void softmax_argmax_store(float (&z)[K], bool do_softmax, long t,
                                                     float* probs, uint8_t* labels) {
    if (do_softmax) {
        float m = z[0];
#pragma unroll
        for (int k = 1; k < K; ++k) m = fmaxf(m, z[k]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - m); s += z[k]; }
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = z[k] / s;
    }
    if (probs) {
#pragma unroll
        for (int k = 0; k < K; ++k) probs[t * K + k] = z[k];
    }
    if (labels) {
        int best = 0; float bv = z[0];
#pragma unroll
        for (int k = 1; k < K; ++k) if (z[k] > bv) { bv = z[k]; best = k; }
        labels[t] = (uint8_t)best;
    }
}

constexpr int MAX_VIEWS = 16;
struct FuseArgs {
    GridDev grid; ViewDev views[MAX_VIEWS]; int V;
    const float* W; const float* b; int sum_fusion;
    float* probs; uint8_t* labels;
};

// Fused kernel: one workgroup = one 4x4x64 (x,y,z) brick; a thread owns 4 voxels (z, z+16, z+32,
// z+48) so that every view's parameters are fetched once per 4 voxels and the 4 gathers of a
// view are independent (ILP). Stores are rows of 16 consecutive z.
constexpr int FZ = 4;
template <int K>
__global__ __launch_bounds__(256) void map_fuse_kernel(FuseArgs a) {
    const GridDev& g = a.grid;
    const int nz = (g.Z + BRZ * FZ - 1) / (BRZ * FZ), ny = (g.Y + BRY - 1) / BRY;
    const long blk = blockIdx.x;
    const int bz = (int)(blk % nz), by = (int)((blk / nz) % ny), bx = (int)(blk / ((long)nz * ny));
    const int vx = bx * BRX + (threadIdx.x >> 6);
    const int vy = by * BRY + ((threadIdx.x >> 4) & 3);
    const int vz0 = bz * BRZ * FZ + (threadIdx.x & 15);
    if (vx >= g.X || vy >= g.Y) return;
    double rx[FZ], ry[FZ], rz[FZ];
    float z[FZ][K];
#pragma unroll
    for (int u = 0; u < FZ; ++u) {
        voxel_real(g, vx, vy, vz0 + 16 * u, rx[u], ry[u], rz[u]);
#pragma unroll
        for (int k = 0; k < K; ++k) z[u][k] = 0.f;
    }
    for (int v = 0; v < a.V; ++v) {
        const ViewDev& vw = a.views[v];
        // closed-form nearest lookups of the four voxels (branch-free); the rare lookups near a tie or an axis end are
        // redone by ONE copy of the exact search (a rolled loop: the exact search is large, and four inlined copies
        // per view made the kernel's code larger than the instruction cache)
        long off[FZ];
        unsigned risky = 0;
#pragma unroll
        for (int u = 0; u < FZ; ++u) {
            double qx, qy, qz;
            mat3_apply(vw.invb, rx[u], ry[u], rz[u], qx, qy, qz);
            int n0, n1, n2; bool o0, o1, o2;
            const bool f0 = nearest_fast(vw.g, qx, n0, o0), f1 = nearest_fast(vw.g, qy, n1, o1), f2 = nearest_fast(vw.offs, qz, n2, o2);
            if (!(f0 && f1 && f2)) risky |= 1u << u;
            off[u] = (o0 || o1 || o2) ? -1 : (((long)n2 * vw.dim + n0) * vw.dim + n1) * K;
        }
        if (__builtin_expect(risky != 0, 0)) {
#pragma unroll 1
            for (int u = 0; u < FZ; ++u) {
                if (!((risky >> u) & 1u)) continue;
                int pl;
                const long o = view_lookup(vw, rx[u], ry[u], rz[u], K, pl);
#pragma unroll
                for (int t = 0; t < FZ; ++t) if (t == u) off[t] = o;
            }
        }
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = a.sum_fusion ? 1.f : a.W[v * K + k];
#pragma unroll
        for (int u = 0; u < FZ; ++u) {
            float x[K];                                   // one K-wide gather per voxel and view: unconditional (clamped
            __builtin_memcpy(x, vw.pred + (off[u] >= 0 ? off[u] : 0), K * sizeof(float));   // offset), the four of a view in
            if (off[u] < 0) {                                                              // flight together
#pragma unroll
                for (int k = 0; k < K; ++k) x[k] = k == 0 ? 1.f : 0.f;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) z[u][k] = a.sum_fusion ? (z[u][k] + x[k]) : (z[u][k] + w[k] * x[k]);
        }
    }
#pragma unroll
    for (int u = 0; u < FZ; ++u) {
        const int vz = vz0 + 16 * u;
        if (vz >= g.Z) continue;
        if (!a.sum_fusion) {
#pragma unroll
            for (int k = 0; k < K; ++k) z[u][k] = z[u][k] + a.b[k];
        }
        softmax_argmax_store<K>(z[u], !a.sum_fusion, ((long)vx * g.Y + vy) * g.Z + vz, a.probs, a.labels);
    }
}

struct MapArgs {
    GridDev grid; ViewDev view; const float* Wv; int p_lo, p_hi, owns_oob; float* out;
};

// ACCUM=false: mapped[t,:] = nearest (map_real_space_pred). ACCUM=true: z[t,:] += Wv*nearest
// restricted to planes [p_lo,p_hi) (pred points at plane p_lo).
template <int K, bool ACCUM>
__global__ __launch_bounds__(256) void map_view_kernel(MapArgs a) {
    const long nblk = brick_count(a.grid);
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int vx, vy, vz; long t;
        if (!brick_voxel(a.grid, blk, vx, vy, vz, t)) continue;
        double rx, ry, rz;
        voxel_real(a.grid, vx, vy, vz, rx, ry, rz);
        int pl;
        const long off = view_lookup(a.view, rx, ry, rz, K, pl);
        if (!ACCUM) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                a.out[t * K + k] = (off >= 0) ? a.view.pred[off + k] : (k == 0 ? 1.f : 0.f);
        } else {
            if (off >= 0) {
                if (pl >= a.p_lo && pl < a.p_hi) {
                    const long o2 = off - (long)a.p_lo * a.view.dim * a.view.dim * K;
#pragma unroll
                    for (int k = 0; k < K; ++k) a.out[t * K + k] += a.Wv[k] * a.view.pred[o2 + k];
                }
            } else if (a.owns_oob) {
                a.out[t * K] += a.Wv[0];
            }
        }
    }
}

template <int K>
__global__ __launch_bounds__(256) void fusion_forward_kernel(const float* __restrict__ x, long n, int V,
                                                             const float* W, const float* b,
                                                             float* probs, uint8_t* labels) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = 0.f;
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = z[k] + W[v * K + k] * x[(t * V + v) * K + k];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = z[k] + b[k];
        softmax_argmax_store<K>(z, true, t, probs, labels);
    }
}

template <int K>
__global__ __launch_bounds__(256) void fusion_finalize_kernel(const float* __restrict__ zin, long n,
                                                              const float* b, int sum_fusion,
                                                              float* probs, uint8_t* labels) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = zin[t * K + k] + (sum_fusion ? 0.f : b[k]);
        softmax_argmax_store<K>(z, !sum_fusion, t, probs, labels);
    }
}

static void to_mat3(const double* s, Mat3& m) { memcpy(m.m, s, sizeof(m.m)); }
static AxisDev to_axis(const double* d_arr, int n, const mpu_axis& m) {
    AxisDev a;
    a.g = d_arr; a.n = n;
    a.kind = (m.kind == 1 || m.kind == 2) && m.n == n ? m.kind : 0;
    a.start = m.start; a.step = m.step; a.last = m.last;
    a.inv_h = m.step > 0 ? 1.0 / m.step : 1.0;
    a.g0 = a.kind == 1 ? m.start : (a.kind == 2 ? (0.0 - m.start) * m.step : 0.0);     // == axis_at(a, 0)
    if (!(m.step > 0)) a.kind = a.kind ? 0 : 0;
    return a;
}
static void to_view(const mpu_view_pred& v, ViewDev& d) {
    to_mat3(v.inv_basis, d.invb);
    d.pred = v.d_pred; d.dim = v.dim; d.P = v.n_planes;
    d.g = to_axis(v.d_g, v.dim, v.g_axis);
    d.offs = to_axis(v.d_offsets, v.n_planes, v.o_axis);
}
static void to_grid(const mpu_voxel_grid& g, GridDev& d) {
    to_mat3(g.A, d.A);
    for (int i = 0; i < 3; ++i) d.c[i] = g.center[i];
    d.X = g.shape[0]; d.Y = g.shape[1]; d.Z = g.shape[2];
}
static int brick_grid(const GridDev& g) {
    long b = (long)((g.X + 3) / 4) * ((g.Y + 3) / 4) * ((g.Z + 15) / 16);
    return (int)(b < 1 ? 1 : (b > (1L << 20) ? (1L << 20) : b));
}
static unsigned fuse_grid(const GridDev& g) {
    return (unsigned)((long)((g.X + 3) / 4) * ((g.Y + 3) / 4) * ((g.Z + 63) / 64));
}
static int grid_for(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256L * 16 ? 256L * 16 : b));
}

#define MPU_DISPATCH_K(K_, CALL)                                            \
    switch (K_) {                                                           \
        case 1: { constexpr int KK = 1; CALL; } break;                      \
        case 2: { constexpr int KK = 2; CALL; } break;                      \
        case 3: { constexpr int KK = 3; CALL; } break;                      \
        case 4: { constexpr int KK = 4; CALL; } break;                      \
        case 5: { constexpr int KK = 5; CALL; } break;                      \
        case 6: { constexpr int KK = 6; CALL; } break;                      \
        case 7: { constexpr int KK = 7; CALL; } break;                      \
        case 8: { constexpr int KK = 8; CALL; } break;                      \
        case 9: { constexpr int KK = 9; CALL; } break;                      \
        case 10: { constexpr int KK = 10; CALL; } break;                    \
        case 11: { constexpr int KK = 11; CALL; } break;                    \
        case 12: { constexpr int KK = 12; CALL; } break;                    \
        case 13: { constexpr int KK = 13; CALL; } break;                    \
        case 14: { constexpr int KK = 14; CALL; } break;                    \
        case 15: { constexpr int KK = 15; CALL; } break;                    \
        case 16: { constexpr int KK = 16; CALL; } break;                    \
        default: return mpu::fail(MPU_EUNSUPPORTED, "%s", "n_classes must be in 1..16"); \
    }

}  // namespace mpu

using namespace mpu;

static int g_fast_host = 1;
static int fast_path_host() { return g_fast_host; }
static int sync_fast_switch() {           // MPU_GEOM_FAST=0: exact search for every sample (read once per process)
    static int done = 0;
    if (done) return MPU_OK;
    const char* e = getenv("MPU_GEOM_FAST");
    const int v = (e && e[0] == '0') ? 0 : 1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(mpu::g_fast_geometry_dev), &v, sizeof(int)) != hipSuccess)
        return mpu::fail(MPU_EHIP, "%s", "geometry: cannot set the fast-path switch");
    g_fast_host = v;
    done = 1;
    return MPU_OK;
}

extern "C" {

int mpu_abi_version(void) { return 1; }

/* test aid: 1 = closed-form fast paths with exact fall-back (default), 0 = exact search everywhere */
int mpu_geometry_set_fast_path(int32_t on) {
    const int v = on ? 1 : 0;
    { const int rc_ = sync_fast_switch(); if (rc_) return rc_; }
    MPU_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mpu::g_fast_geometry_dev), &v, sizeof(int)));
    g_fast_host = v;
    return MPU_OK;
}
const char* mpu_last_error(void) { return mpu::g_err; }

int mpu_sample_view_planes(const float* d_vol, const uint8_t* d_labels, const int32_t vol_shape[4],
                           const double* d_ax, const double* d_ay, const double* d_az,
                           const mpu_view_geom* geom, const double* d_offsets,
                           const float* d_bg, uint8_t bg_class,
                           const double* d_center, const double* d_scale,
                           float* d_out, uint8_t* d_out_lab, void* stream) {
    { const int rc_ = sync_fast_switch(); if (rc_) return rc_; }
    MPU_REQUIRE(d_vol && vol_shape && d_ax && d_ay && d_az && geom && d_offsets && d_bg && d_out,
                "mpu_sample_view_planes: null argument");
    MPU_REQUIRE(vol_shape[0] >= 2 && vol_shape[1] >= 2 && vol_shape[2] >= 2 && vol_shape[3] >= 1,
                "mpu_sample_view_planes: volume must be at least 2x2x2x1");
    MPU_REQUIRE(geom->dim >= 2 && geom->n_planes >= 1, "mpu_sample_view_planes: bad dim / n_planes");
    MPU_REQUIRE((d_center == nullptr) == (d_scale == nullptr),
                "mpu_sample_view_planes: center and scale must both be given or both NULL");
    MPU_REQUIRE(!d_out_lab || d_labels, "mpu_sample_view_planes: label output requested without labels");
    SampleArgs a;
    a.vol = d_vol; a.labels = d_labels;
    a.X = vol_shape[0]; a.Y = vol_shape[1]; a.Z = vol_shape[2]; a.C = vol_shape[3];
    a.ax = to_axis(d_ax, a.X, geom->vol_axis[0]);
    a.ay = to_axis(d_ay, a.Y, geom->vol_axis[1]);
    a.az = to_axis(d_az, a.Z, geom->vol_axis[2]);
    a.offsets = d_offsets;
    to_mat3(geom->basis, a.basis); to_mat3(geom->rot, a.rot); a.has_rot = geom->has_rot;
    a.dim = geom->dim; a.P = geom->n_planes; a.g_start = geom->g_start; a.g_step = geom->g_step;
    a.bg = d_bg; a.bg_class = bg_class; a.center = d_center; a.scale = d_scale;
    a.out = d_out; a.out_lab = d_out_lab;
    const long tpd = (a.dim + 15) / 16;
    long nblk = (long)a.P * tpd * tpd;
    if (nblk > (1L << 20)) nblk = 1L << 20;
    sample_view_planes_kernel<<<dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream>>>(a);
    return launch_ok();
}

int mpu_map_view_nearest(const mpu_voxel_grid* grid, const mpu_view_pred* view, int32_t n_classes,
                         float* d_mapped, void* stream) {
    { const int rc_ = sync_fast_switch(); if (rc_) return rc_; }
    MPU_REQUIRE(grid && view && d_mapped && view->d_pred && view->d_g && view->d_offsets,
                "mpu_map_view_nearest: null argument");
    MPU_REQUIRE(view->dim >= 2 && view->n_planes >= 2, "mpu_map_view_nearest: view needs dim>=2, planes>=2");
    MapArgs a; to_grid(*grid, a.grid); to_view(*view, a.view);
    a.Wv = nullptr; a.p_lo = 0; a.p_hi = view->n_planes; a.owns_oob = 1; a.out = d_mapped;
    MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, false><<<dim3(brick_grid(a.grid)), dim3(256), 0, (hipStream_t)stream>>>(a)));
    return launch_ok();
}

int mpu_map_accumulate_view(const mpu_voxel_grid* grid, const mpu_view_pred* view, int32_t n_classes,
                            const float* d_Wv, int32_t p_lo, int32_t p_hi, int32_t owns_oob,
                            float* d_z, void* stream) {
    { const int rc_ = sync_fast_switch(); if (rc_) return rc_; }
    MPU_REQUIRE(grid && view && d_z && d_Wv && view->d_pred && view->d_g && view->d_offsets,
                "mpu_map_accumulate_view: null argument");
    MPU_REQUIRE(0 <= p_lo && p_lo < p_hi && p_hi <= view->n_planes, "mpu_map_accumulate_view: bad plane range");
    MapArgs a; to_grid(*grid, a.grid); to_view(*view, a.view);
    a.Wv = d_Wv; a.p_lo = p_lo; a.p_hi = p_hi; a.owns_oob = owns_oob; a.out = d_z;
    MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, true><<<dim3(brick_grid(a.grid)), dim3(256), 0, (hipStream_t)stream>>>(a)));
    return launch_ok();
}

int mpu_map_fuse_views(const mpu_voxel_grid* grid, const mpu_view_pred* views, int32_t n_views,
                       int32_t n_classes, const float* d_W, const float* d_b, int32_t sum_fusion,
                       float* d_probs, uint8_t* d_labels, void* stream) {
    { const int rc_ = sync_fast_switch(); if (rc_) return rc_; }
    MPU_REQUIRE(grid && views, "mpu_map_fuse_views: null argument");
    MPU_REQUIRE(n_views >= 1 && n_views <= MAX_VIEWS, "mpu_map_fuse_views: n_views must be in 1..16");
    MPU_REQUIRE(sum_fusion || (d_W && d_b), "mpu_map_fuse_views: W and b required unless sum_fusion");
    MPU_REQUIRE(d_probs || d_labels, "mpu_map_fuse_views: no output requested");
    FuseArgs a; to_grid(*grid, a.grid);
    for (int v = 0; v < n_views; ++v) {
        MPU_REQUIRE(views[v].d_pred && views[v].d_g && views[v].d_offsets, "mpu_map_fuse_views: null view field");
        MPU_REQUIRE(views[v].dim >= 2 && views[v].n_planes >= 2, "mpu_map_fuse_views: view needs dim>=2, planes>=2");
        to_view(views[v], a.views[v]);
    }
    a.V = n_views; a.W = d_W; a.b = d_b; a.sum_fusion = sum_fusion; a.probs = d_probs; a.labels = d_labels;
    MPU_DISPATCH_K(n_classes, (map_fuse_kernel<KK><<<dim3(fuse_grid(a.grid)), dim3(256), 0, (hipStream_t)stream>>>(a)));
    return launch_ok();
}

int mpu_fusion_forward(const float* d_x, int64_t n, int32_t n_views, int32_t n_classes,
                       const float* d_W, const float* d_b, float* d_probs, uint8_t* d_labels, void* stream) {
    MPU_REQUIRE(d_x && d_W && d_b && (d_probs || d_labels), "mpu_fusion_forward: null argument");
    MPU_REQUIRE(n >= 0 && n_views >= 1, "mpu_fusion_forward: bad sizes");
    if (n == 0) return MPU_OK;
    MPU_DISPATCH_K(n_classes, (fusion_forward_kernel<KK><<<dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream>>>(
                                   d_x, (long)n, n_views, d_W, d_b, d_probs, d_labels)));
    return launch_ok();
}

int mpu_fusion_finalize(const float* d_z, int64_t n, int32_t n_classes, const float* d_b, int32_t sum_fusion,
                        float* d_probs, uint8_t* d_labels, void* stream) {
    MPU_REQUIRE(d_z && (sum_fusion || d_b) && (d_probs || d_labels), "mpu_fusion_finalize: null argument");
    if (n <= 0) return MPU_OK;
    MPU_DISPATCH_K(n_classes, (fusion_finalize_kernel<KK><<<dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream>>>(
                                   d_z, (long)n, d_b, sum_fusion, d_probs, d_labels)));
    return launch_ok();
}

}  // extern "C"
