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
#include <math.h>
#include <mutex>
#include <vector>
#include "common.h"
#include "kernels.h"

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
// u = (x - g[0]) / h in index units carries an error of a few ulps of (|g[0]| / h + n), and the actual axis nodes the
// exact search compares against lie within the same distance of the integers; to_axis() keeps an axis closed-form
// only while 64 * 2^-53 * (|g[0]| / h + n) < TAU / 4 (typical: 2e-12). So whenever u is farther than TAU = 1e-8 from
// every value at which the exact procedure changes its answer -- an integer (searchsorted / out-of-bounds limits
// g[0], g[n-1]) for the cell search, a half-integer (the y <= 0.5 tie) and the two axis ends for the nearest search
// -- the closed form below IS the exact answer. The remaining samples (a fraction ~1e-7; integer spans put more of
// them exactly ON nodes and ties) take the exact search. g_fast_geometry = 0 (MPU_GEOM_FAST=0) forces the exact
// search for every sample (A/B and the equality test).
__constant__ int g_fast_geometry_dev = 1;
constexpr double GEOM_TAU = 1e-8;

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

// One sample of get_view_from: the exact NumPy-order evaluation (any axis kind, any channel count).
__device__ __forceinline__ void sample_one(const SampleArgs& a, int p, int i, int j) {
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
                acc = acc + (double)a.vol[idx] * w;
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
// Generic kernel (any axis kind / channel count): one workgroup = one 16x16 patch of one plane.
__global__ __launch_bounds__(256) void sample_view_planes_kernel(SampleArgs a) {
    const int tpd = (a.dim + 15) / 16;
    const long nblk = (long)a.P * tpd * tpd;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int p = (int)(blk / (tpd * tpd));
        const int ti = (int)((blk / tpd) % tpd), tj = (int)(blk % tpd);
        const int i = ti * 16 + (threadIdx.x >> 4), j = tj * 16 + (threadIdx.x & 15);
        if (i >= a.dim || j >= a.dim) continue;
        sample_one(a, p, i, j);
    }
}

// exact recomputation of the samples on the straight-line kernel's work list (all samples if the list overflowed)
__global__ __launch_bounds__(256) void sample_fixup_kernel(SampleArgs a, const unsigned* list, const unsigned* count, unsigned cap,
                                                           unsigned* next_count) {
    const unsigned n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    const long total = (long)a.P * a.dim * a.dim;
    const bool all = n > cap;
    const long m = all ? total : (long)n;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < m; k += (long)gridDim.x * 256) {
        const long t = all ? k : (long)list[k];
        sample_one(a, (int)(t / ((long)a.dim * a.dim)), (int)((t / a.dim) % a.dim), (int)(t % a.dim));
    }
}

// ---- straight-line sampler for three voxel axes of one closed-form kind (1 or 2), C = 1 or 2 ------------
// The point coordinates follow the exact chain (they feed the interpolation weights). Per axis the cell is
// c = floor(u), u = (x - g[0]) / h: exact whenever u is farther than GEOM_TAU from an integer (see cell_fast); the
// other samples (~1e-7 of them) go to a work list and are redone by sample_fixup_kernel. Everything else is the reference's arithmetic
// op for op, written without branches: out-of-bounds samples load from a clamped cell and select the fill value.
// num / den for a cell width den = g[c+1] - g[c] of a uniform axis, bit for bit the IEEE quotient, without the
// generic division's scaling / fix-up instructions: rh = fl(1/h) is within ~n * 2^-52 (relative) of 1/den, one
// Newton step makes it a reciprocal good to an ulp, and the quotient is then refined through the exact remainder --
// the same final steps as the compiler's own f64 division expansion (v_rcp + 2 Newton steps, q = num * r,
// rem = fma(-den, q, num), q' = fma(rem, r, q)), whose operands here are far from the overflow / denormal ranges that
// the skipped v_div_scale / v_div_fixup handle. tests/test_gpu_geometry.py::test_cell_division_is_ieee compares 2^28
// quotients per axis step with the `/` operator; MPU_GEOM_FAST=0 keeps `/` everywhere.
__device__ __forceinline__ double cell_div(double num, double den, double rh) {
    const double e = fma(-den, rh, 1.0);
    const double r = fma(rh, e, rh);
    const double q = num * r;
    const double rem = fma(-den, q, num);
    return fma(rem, r, q);
}
struct CellF { int c; double num, den; };
template <int KIND>
__device__ __forceinline__ CellF cell_uniform(const AxisDev& a, double x, bool& oob, bool& risky) {
    const double u = (x - a.g0) * a.inv_h;
    const double f = floor(u);
    const double nm1 = (double)(a.n - 1);
    risky |= !(fabs((u - f) - 0.5) < 0.5 - GEOM_TAU);                // within TAU of a node, or NaN
    oob |= (u < 0.0) | (u > nm1);
    const double fc = fmin(fmax(f, 0.0), nm1 - 1.0);                 // in-bounds samples: fc == f
    double gc, gc1;                                                  // axis_at(c), axis_at(c + 1)
    if (KIND == 2) { gc = (fc - a.start) * a.step; gc1 = ((fc + 1.0) - a.start) * a.step; }
    else { gc = fc * a.step + a.start; gc1 = (fc + 1.0 == nm1) ? a.last : (fc + 1.0) * a.step + a.start; }
    CellF r;
    r.c = (int)fc;
    r.num = x - gc; r.den = gc1 - gc;
    return r;
}

// test aid: count the lanes whose cell_div differs from `/` on pseudo-random (num, den) with den = a cell width of
// the axis (start, step, n), num = x - g[c] for x inside or just outside the cell
__global__ __launch_bounds__(256) void cell_div_check_kernel(double start, double step, int n, int kind, long count,
                                                             unsigned long long seed, unsigned long long* bad) {
    const double rh = 1.0 / step;
    unsigned long long local = 0;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (long)gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(t + 1);      // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        const int c = (int)((z >> 40) % (unsigned long long)(n - 1));
        const double frac = (double)(z & 0xFFFFFFFFFFull) * (1.0 / 1099511627776.0) * 1.25 - 0.125;   // [-0.125, 1.125)
        const double fc = (double)c;
        double gc, gc1;
        if (kind == 2) { gc = (fc - start) * step; gc1 = ((fc + 1.0) - start) * step; }
        else { gc = fc * step + start; gc1 = (fc + 1.0) * step + start; }
        const double x = gc + frac * step;
        const double num = x - gc, den = gc1 - gc;
        if (cell_div(num, den, rh) != num / den) ++local;
    }
    if (local) atomicAdd(bad, local);
}

template <int KIND, int C, bool LAB>
__global__ __launch_bounds__(256) void sample_fast_kernel(SampleArgs a, unsigned* list, unsigned* count, unsigned cap) {
    // one workgroup = 8 rows x 32 columns of one plane (128-byte rows of output, compact footprint in the volume)
    const int p = blockIdx.z;
    const int i = blockIdx.y * 8 + (threadIdx.x >> 5), j = blockIdx.x * 32 + (threadIdx.x & 31);
    if (i >= a.dim || j >= a.dim) return;
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
    bool oob = false, risky = false;
    const CellF c0 = cell_uniform<KIND>(a.ax, rx, oob, risky);
    const CellF c1 = cell_uniform<KIND>(a.ay, ry, oob, risky);
    const CellF c2 = cell_uniform<KIND>(a.az, rz, oob, risky);
    const long t = ((long)p * a.dim + i) * a.dim + j;
    if (__builtin_expect(risky, 0)) {                  // redone (and stored) by sample_fixup_kernel
        const unsigned idx = atomicAdd(count, 1u);
        if (idx < cap) list[idx] = (unsigned)t;
        return;
    }
    // the two z-neighbours of an (x, y) corner are contiguous: one 8-byte (C == 1) / 16-byte (C == 2) load per corner,
    // all four issued before the divisions below
    const unsigned sy = (unsigned)a.Z * C, sx = (unsigned)a.Y * sy;
    const unsigned base = (unsigned)c0.c * sx + (unsigned)c1.c * sy + (unsigned)c2.c * C;
    float cv[C][8];
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.vol, 0, (int)((long)a.X * sx * 4L), 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* src = a.vol + (base + (q >> 1) * sx + (q & 1) * sy);
        if (C == 1) {
            float2 v; __builtin_memcpy(&v, src, 8);
            cv[0][2 * q] = v.x; cv[0][2 * q + 1] = v.y;
        } else {                                                 // two channels: one 16-byte buffer load per corner pair (a 16-byte
            typedef unsigned int geo_u32x4 __attribute__((ext_vector_type(4)));      // memcpy of unknown alignment became 4 dword loads)
            const geo_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(vrs, (base + (q >> 1) * sx + (q & 1) * sy) * 4u, 0, 0);
            cv[0][2 * q] = __uint_as_float(v.x); cv[C - 1][2 * q] = __uint_as_float(v.y);
            cv[0][2 * q + 1] = __uint_as_float(v.z); cv[C - 1][2 * q + 1] = __uint_as_float(v.w);
        }
    }
    const double y0 = cell_div(c0.num, c0.den, a.ax.inv_h), y1 = cell_div(c1.num, c1.den, a.ay.inv_h),
                 y2 = cell_div(c2.num, c2.den, a.az.inv_h);
    uint8_t lab = a.bg_class;
    if (LAB) {
        const unsigned n0 = (unsigned)c0.c + (y0 <= .5 ? 0u : 1u), n1 = (unsigned)c1.c + (y1 <= .5 ? 0u : 1u),
                       n2 = (unsigned)c2.c + (y2 <= .5 ? 0u : 1u);
        const uint8_t l = a.labels[(n0 * (unsigned)a.Y + n1) * (unsigned)a.Z + n2];
        lab = oob ? a.bg_class : l;
    }
    // itertools.product order of the 8 corners, weight = ((1*wx)*wy)*wz (1*wx is exact)
    const double wx[2] = {1.0 - y0, y0}, wy[2] = {1.0 - y1, y1}, wz[2] = {1.0 - y2, y2};
    double w[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = (wx[e >> 2] * wy[(e >> 1) & 1]) * wz[e & 1];
    float res[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = acc + (double)cv[c][e] * w[e];
        float v32 = oob ? a.bg[c] : (float)acc;
        if (a.center) {    // sklearn: X -= center_; X /= scale_ (f64 op, f32 store)
            v32 = (float)((double)v32 - a.center[c]);
            v32 = (float)((double)v32 / a.scale[c]);
        }
        res[c] = v32;
    }
    if (C == 1) a.out[t] = res[0];
    else { float2 v; v.x = res[0]; v.y = res[C - 1]; __builtin_memcpy(a.out + t * 2, &v, 8); }
    if (LAB) a.out_lab[t] = lab;
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

// ---- straight-line fused kernel on composed affine maps -------------------------------------------------
// For a view whose in-plane axis and offsets are uniform (kinds 1, 2) the three nearest searches of a voxel depend
// only on the index-space coordinates u = diag(1/h) (invb (A v - c) - g0) = M v + t of the voxel v = (x, y, z). The
// host composes M and t once per view (compose_view); the kernel evaluates u with one FMA chain per axis, rounds,
// and flags a lookup as RISKY when u is within GEOM_TAU of a value where the exact procedure changes its answer (a
// half-integer tie, an axis end). |u - u_exact| is bounded by compose_view's err (it must be < GEOM_TAU / 4, else
// the view is not eligible), so a lookup that is not risky equals the exact one. Risky voxels (~1e-7 of them) are
// appended to a work list and recomputed from scratch by map_fuse_fixup_kernel with the exact search.
struct AffView { double M[9], t[3]; const float* pred; int dim, P; int S[3], _pad; };   // S: fixed-point z step (FX kernels)
struct FuseFastArgs {
    AffView v[MAX_VIEWS]; int V, X, Y, Z;
    const float* W; const float* b; int sum_fusion;
    float* probs; uint8_t* labels;
    unsigned* list; unsigned* count; unsigned cap, nblk8;
    int morton, px2, py2;        // brick columns (x, y) in Morton order (log2 of the padded column grid), z fastest
};

// FX (round 4, CFG 2 only): the index coordinates of a lane's four consecutive z voxels in 32-bit FIXED POINT. Only a rounded
// index and three range / tie decisions are needed per axis, not a value: voxel 0's coordinate u0 comes from the fp64 chain
// as before, is truncated to Q11.20 (FP0 = trunc(u0 * 2^20), clamped to +-1.5 * 2^30), and voxels 1..3 add the host-rounded
// step S = rint(M_z * 2^20). Error against the fp64 coordinate: < 2^-20 (truncation) + 3 * 2^-21 (steps) = 2.4e-6 index
// units. A decision of the exact procedure changes only at multiples of 0.5 (ties at half-integers; the range limits 0 and
// n-1 are integers), so a voxel whose fixed-point coordinate is farther than FX_T * 2^-20 = 1.5e-5 from every multiple of
// 0.5 on all three axes has the same rounded index and the same in / out answer: per axis and voxel one add (position), one
// unsigned compare (range), add + shift (rounded index) and one shift-add (distance to the 0.5 lattice; min3 + one compare
// per voxel) -- ~20 integer instructions per voxel and view where the fp64 form below spends ~33 fp64 ones (half rate).
// Lanes inside the band (1.8e-4 of the lookups) recompute THAT view's lookup with the fp64 form in a divergent branch;
// what that flags (|u - lattice| < GEOM_TAU, ~1e-7) goes to the exact fix-up as before. Host-side eligibility: every
// axis <= 1024 nodes and |M_z| <= 8 (compose_view fills S).
constexpr int FX_T = 16, FX_SH = 20;
template <int K, int CFG, bool FX = false>
__global__ __launch_bounds__(256) void map_fuse_fast_kernel(FuseFastArgs a) {
    static_assert(!FX || CFG == 2, "fixed-point screening: the 4-consecutive-z lane layout");
    // lanes: 16 along z, LY along y, the rest along x; a thread owns FZ voxels strided along z (CFG 0: brick
    // 4x4x64) or along x (CFG 1: brick 8x8x16)
    // CFG 2: a WAVE covers a 4x4x16 block, lane (tx, ty, tz) owns the 4 consecutive voxels z = 4 tz .. 4 tz + 3; the 4
    //        waves tile a 8x8x16 brick. Each of a view's 4 gather instructions then has a compact 4x4x(4 strided) footprint and
    //        the four together touch the lines of a 4x4x16 block once (fewest L2 requests per voxel for an arbitrary view).
    // (Round 5, gpurun R5t: a lane owning z = tz, tz + 4, tz + 8, tz + 12 instead -- every gather instruction then covers a
    //        COMPACT 4x4x4 cube, labels exchanged inside the quad by DPP -- measured 0.416 vs 0.424 ms: the number of distinct lines
    //        per gather instruction is not what bounds this kernel; not kept.)
    constexpr int LY = CFG == 1 ? 8 : 4, BX = CFG ? 8 : 4, BY = CFG ? 8 : 4, BZ = CFG ? 16 : 64, OWN = CFG == 1 ? 0 : 2,
                  STRIDE = CFG == 1 ? 2 : (CFG == 2 ? 1 : 16);
    const int nz = (a.Z + BZ - 1) / BZ, ny = (a.Y + BY - 1) / BY, nx = (a.X + BX - 1) / BX;
    // workgroups with equal blockIdx % 8 share an XCD (one L2): each XCD walks its own contiguous run of bricks
    const unsigned L = (blockIdx.x & 7u) * a.nblk8 + (blockIdx.x >> 3);
    int bx, by, bz;
    if (a.morton) {
        // a 128-byte line of a view's predictions (10.7 voxels) straddles neighbouring bricks: walk the brick columns in
        // Morton order so that x- and y-neighbours follow within a few columns (while the line is still in this XCD's L2)
        auto compact = [](unsigned v) { v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu;
                                        v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0x0000ffffu; return v; };
        unsigned col;
        if (a.morton == 1) {                                     // z fastest: one whole column of bricks after the other
            col = L / (unsigned)nz;
            bz = (int)(L % (unsigned)nz);
        } else {
            // order inside the XCD's run (its cpx consecutive Morton columns): a.morton == 2 (default): z SLOWEST -- the
            // run's columns are swept slab by slab, so the x- and y-neighbours that share a view's 128-byte lines follow
            // each other within a few bricks (z fastest put 16 bricks = ~3 MB of lines between them): FETCH_SIZE
            // 2.04 -> 1.33 GB per 256^3 launch (algorithmic 1.21), kernel 354 -> 336 us (round 3, gpurun R3i);
            // a.morton == 3: 2x2 columns x 4 z-bricks interleaved on two levels (1.50 GB, 333 us)
            const unsigned cpx = a.nblk8 / (unsigned)nz, Lr = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
            unsigned colr, z;
            if (a.morton == 2 || (nz & 15) || (cpx & 15)) { z = Lr / cpx; colr = Lr % cpx; }
            else {
                const unsigned grp = Lr / (16u * (unsigned)nz), w = Lr % (16u * (unsigned)nz);     // 16 columns x nz bricks
                const unsigned c_lo = w & 3u, z_lo = (w >> 2) & 3u, c_mid = (w >> 4) & 3u, z_hi = w >> 6;
                colr = grp * 16u + c_mid * 4u + c_lo; z = z_hi * 4u + z_lo;
            }
            col = xcd * cpx + colr; bz = (int)z;
        }
        const int mb = a.px2 < a.py2 ? a.px2 : a.py2;
        const unsigned lo = col & ((1u << (2 * mb)) - 1u), hi = col >> (2 * mb);
        unsigned cx = compact(lo), cy = compact(lo >> 1);
        if (a.px2 > a.py2) cx |= hi << mb; else cy |= hi << mb;
        if (cx >= (unsigned)nx || cy >= (unsigned)ny || bz >= nz) return;
        bx = (int)cx; by = (int)cy;
    } else {
        if (L >= (unsigned)nz * (unsigned)ny * (unsigned)nx) return;
        bz = (int)(L % (unsigned)nz); by = (int)((L / (unsigned)nz) % (unsigned)ny); bx = (int)(L / ((unsigned)nz * (unsigned)ny));
    }
    int vx0, vy, vz0;
    if (CFG == 2) {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        vx0 = bx * BX + (wv & 1) * 4 + (ln >> 4);
        vy = by * BY + (wv >> 1) * 4 + ((ln >> 2) & 3);
        vz0 = bz * BZ + (ln & 3) * 4;
    } else {
        vx0 = bx * BX + (int)(threadIdx.x / (16 * LY));
        vy = by * BY + (int)((threadIdx.x >> 4) % LY);
        vz0 = bz * BZ + (int)(threadIdx.x & 15);
    }
    if (vx0 >= a.X || vy >= a.Y || vz0 >= a.Z) return;
    const double cf0 = OWN == 0 ? (double)vz0 : (double)vx0, cy = (double)vy;      // the two coordinates all owned voxels share
    const int own0 = OWN == 0 ? vx0 : vz0;
    double co[FZ];
    float z[FZ][K];
#pragma unroll
    for (int u = 0; u < FZ; ++u) {
        co[u] = (double)(own0 + STRIDE * u);
#pragma unroll
        for (int k = 0; k < K; ++k) z[u][k] = 0.f;
    }
    unsigned risky = 0;
    // two views per iteration (eight gathers in flight): 0.439 -> 0.433 ms at 256^3 x 6 views, same box (gpurun R5l). The same
    // call measured NON-TEMPORAL gathers + label stores (the float4 nt copy probe reaches 6.4 TB/s on this part, VERDICT r4
    // item 7a): 0.726 ms -- a view's 128-byte line serves ~10 neighbouring voxels of other lanes and bricks, nt throws that
    // L2 reuse away; not kept.
#pragma unroll 2
    for (int v = 0; v < a.V; ++v) {
        const AffView& w = a.v[v];
        const double hg = 0.5 * (double)(w.dim - 1), ho = 0.5 * (double)(w.P - 1);
        double base[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) base[r] = fma(w.M[3 * r + 1], cy, fma(w.M[3 * r + (OWN == 0 ? 2 : 0)], cf0, w.t[r]));
        float wk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = a.sum_fusion ? 1.f : a.W[v * K + k];
        unsigned off[FZ]; bool out[FZ];
        // one lookup in fp64 (the round-2 form): rounded indices, sure-out and "within GEOM_TAU of a decision" flags
        auto lookup64 = [&](int u, unsigned& offu, bool& outu) {
            int n[3]; bool o = false, rk = false;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double uu = fma(w.M[3 * r + OWN], co[u], base[r]);
                const double h = r == 2 ? ho : hg;
                const double rr = rint(uu);
                rk |= !(fabs(fabs(uu - rr) - 0.5) > GEOM_TAU);                 // tie (or NaN)
                const double e = fabs(uu - h);
                const bool in = e < h - GEOM_TAU, sure_out = e > h + GEOM_TAU;
                o |= sure_out; rk |= !(in | sure_out);
                n[r] = (int)rr;
            }
            outu = o;
            risky |= rk ? (1u << u) : 0u;
            offu = (o | rk) ? 0u : (((unsigned)n[2] * (unsigned)w.dim + (unsigned)n[0]) * (unsigned)w.dim + (unsigned)n[1]) * K;
        };
        if constexpr (FX) {
            constexpr int LIMC = 3 << 29;                            // clamp: 1.5 * 2^30 (+ 3 steps of <= 2^23 stays below 2^31)
            int fp0[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double u0 = fma(w.M[3 * r + OWN], co[0], base[r]);
                const int q = (int)(u0 * (double)(1 << FX_SH));      // v_cvt_i32_f64: truncates, saturates
                fp0[r] = q < -LIMC ? -LIMC : (q > LIMC ? LIMC : q);
            }
            const unsigned lim[3] = {(unsigned)(w.dim - 1) << FX_SH, (unsigned)(w.dim - 1) << FX_SH, (unsigned)(w.P - 1) << FX_SH};
            unsigned band = 0;
#pragma unroll
            for (int u = 0; u < FZ; ++u) {
                int n[3]; bool in = true; unsigned qmin = 0xffffffffu;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int fp = fp0[r] + u * w.S[r];
                    in &= (unsigned)fp < lim[r];
                    n[r] = (fp + (1 << (FX_SH - 1))) >> FX_SH;
                    const unsigned q = ((unsigned)fp << (32 - (FX_SH - 1))) + ((unsigned)FX_T << (32 - (FX_SH - 1)));
                    qmin = q < qmin ? q : qmin;
                }
                const bool near = qmin < ((unsigned)(2 * FX_T) << (32 - (FX_SH - 1)));
                band |= near ? (1u << u) : 0u;
                out[u] = !in;
                off[u] = (!in | near) ? 0u : (((unsigned)n[2] * (unsigned)w.dim + (unsigned)n[0]) * (unsigned)w.dim + (unsigned)n[1]) * K;
            }
            if (band) {                                              // ~2e-4 of the lookups: this view's fp64 form for those voxels
#pragma unroll
                for (int u = 0; u < FZ; ++u)
                    if ((band >> u) & 1u) lookup64(u, off[u], out[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < FZ; ++u) lookup64(u, off[u], out[u]);
        }
#pragma unroll
        for (int u = 0; u < FZ; ++u) {
            float x[K];
            __builtin_memcpy(x, w.pred + off[u], K * sizeof(float));
            if (out[u]) {
#pragma unroll
                for (int k = 0; k < K; ++k) x[k] = k == 0 ? 1.f : 0.f;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) z[u][k] = a.sum_fusion ? (z[u][k] + x[k]) : (z[u][k] + wk[k] * x[k]);
        }
    }
    // labels of the 4 consecutive voxels of a CFG-2 lane as one 32-bit store (risky ones are overwritten by the fix-up,
    // which runs after this kernel)
    const bool packed = CFG == 2 && a.labels && !a.probs && (a.Z & 3) == 0;
    unsigned pack = 0;
#pragma unroll
    for (int u = 0; u < FZ; ++u) {
        const int vx = OWN == 0 ? own0 + STRIDE * u : vx0, vz = OWN == 0 ? vz0 : own0 + STRIDE * u;
        if (vx >= a.X || vz >= a.Z) continue;
        const long t = ((long)vx * a.Y + vy) * a.Z + vz;
        if ((risky >> u) & 1u) {                       // redone by the fix-up kernel (which also stores it)
            const unsigned idx = atomicAdd(a.count, 1u);
            if (idx < a.cap) a.list[idx] = (unsigned)t;
            continue;
        }
        if (!a.sum_fusion) {
#pragma unroll
            for (int k = 0; k < K; ++k) z[u][k] = z[u][k] + a.b[k];
        }
        if (packed) {
            int best = 0; float bv = z[u][0];           // argmax is invariant under the softmax
#pragma unroll
            for (int k = 1; k < K; ++k) if (z[u][k] > bv) { bv = z[u][k]; best = k; }
            pack |= (unsigned)best << (8 * u);
        } else {
            softmax_argmax_store<K>(z[u], !a.sum_fusion, t, a.probs, a.labels);
        }
    }
    if (packed) *(unsigned*)(a.labels + (((long)vx0 * a.Y + vy) * a.Z + vz0)) = pack;
}

// exact recomputation of the voxels on the work list (all voxels if the list overflowed). The list is short and the exact
// search long, so the kernel's time is the latency of ONE voxel: a workgroup takes 64 voxels (lane = voxel), its eight
// waves look up the views w, w + 8, ... (a wave-uniform view: its parameters stay scalar loads), and wave 0 then
// accumulates the views in order, as the fused kernels do.
template <int K>
__global__ __launch_bounds__(512) void map_fuse_fixup_kernel(FuseArgs a, const unsigned* list, const unsigned* count, unsigned cap,
                                                             unsigned* next_count) {
    __shared__ float xs[MAX_VIEWS][K][64];
    const GridDev& g = a.grid;
    const unsigned n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    const long total = (long)g.X * g.Y * g.Z;
    const bool all = n > cap;
    const long m = all ? total : (long)n;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (long i0 = (long)blockIdx.x * 64; i0 < m; i0 += (long)gridDim.x * 64) {       // uniform trip count per workgroup
        const long i = i0 + lane;
        const bool live = i < m;
        const long t = live ? (all ? i : (long)list[i]) : 0;
        const int vz = (int)(t % g.Z), vy = (int)((t / g.Z) % g.Y), vx = (int)(t / ((long)g.Z * g.Y));
        double rx, ry, rz;
        voxel_real(g, vx, vy, vz, rx, ry, rz);
        for (int v = wave; v < a.V; v += 8) {                                         // (eight waves: six views = one round;
            const ViewDev& vw = a.views[v];                                           //  round 3 ran two rounds on four waves)
            int pl;
            const long o = live ? view_lookup(vw, rx, ry, rz, K, pl) : -1;
#pragma unroll
            for (int k = 0; k < K; ++k) xs[v][k][lane] = o >= 0 ? vw.pred[o + k] : (k == 0 ? 1.f : 0.f);
        }
        __syncthreads();
        if (wave == 0 && live) {
            float z[K];
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = 0.f;
            for (int v = 0; v < a.V; ++v) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float x = xs[v][k][lane];
                    const float wv = a.sum_fusion ? 1.f : a.W[v * K + k];
                    z[k] = a.sum_fusion ? (z[k] + x) : (z[k] + wv * x);
                }
            }
            if (!a.sum_fusion) {
#pragma unroll
                for (int k = 0; k < K; ++k) z[k] = z[k] + a.b[k];
            }
            softmax_argmax_store<K>(z, !a.sum_fusion, t, a.probs, a.labels);
        }
        __syncthreads();
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

// exact map / accumulate of the voxels on the work list of map_view_fast_kernel (all voxels if the list overflowed)
template <int K, bool ACCUM>
__global__ __launch_bounds__(256) void map_view_fixup_kernel(MapArgs a, const unsigned* list, const unsigned* count, unsigned cap,
                                                             unsigned* next_count) {
    const GridDev& g = a.grid;
    const unsigned n = *count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *next_count = 0;
    const long total = (long)g.X * g.Y * g.Z;
    const bool all = n > cap;
    const long m = all ? total : (long)n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < m; i += (long)gridDim.x * 256) {
        const long t = all ? i : (long)list[i];
        const int vz = (int)(t % g.Z), vy = (int)((t / g.Z) % g.Y), vx = (int)(t / ((long)g.Z * g.Y));
        double rx, ry, rz;
        voxel_real(g, vx, vy, vz, rx, ry, rz);
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

// ---- straight-line single-view kernels (map_real_space_pred / the sharded accumulate) ---------------------
// Same scheme as map_fuse_fast_kernel for ONE view: composed affine index map, a wave = a 4x4x16 block of voxels with 4
// consecutive z per lane, 8x8x16 bricks walked in Morton order per XCD; voxels within GEOM_TAU of a decision boundary are
// skipped here and handled by map_view_fixup_kernel with the exact search (for the accumulate: exactly once).
struct MapFastArgs {
    AffView v; int X, Y, Z;
    const float* Wv; int p_lo, p_hi, owns_oob; float* out;
    unsigned* list; unsigned* count; unsigned cap, nblk8; int px2, py2;
};
template <int K, bool ACCUM>
__global__ __launch_bounds__(256) void map_view_fast_kernel(MapFastArgs a) {
    const int nz = (a.Z + 15) / 16, ny = (a.Y + 7) / 8, nx = (a.X + 7) / 8;
    const unsigned L = (blockIdx.x & 7u) * a.nblk8 + (blockIdx.x >> 3);
    auto compact = [](unsigned v) { v &= 0x55555555u; v = (v | (v >> 1)) & 0x33333333u; v = (v | (v >> 2)) & 0x0f0f0f0fu;
                                    v = (v | (v >> 4)) & 0x00ff00ffu; v = (v | (v >> 8)) & 0x0000ffffu; return v; };
    const unsigned col = L / (unsigned)nz;
    const int bz = (int)(L % (unsigned)nz);
    const int mb = a.px2 < a.py2 ? a.px2 : a.py2;
    const unsigned lo = col & ((1u << (2 * mb)) - 1u), hi = col >> (2 * mb);
    unsigned cx = compact(lo), cy = compact(lo >> 1);
    if (a.px2 > a.py2) cx |= hi << mb; else cy |= hi << mb;
    if (cx >= (unsigned)nx || cy >= (unsigned)ny) return;
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const int vx = (int)cx * 8 + (wv & 1) * 4 + (ln >> 4);
    const int vy = (int)cy * 8 + (wv >> 1) * 4 + ((ln >> 2) & 3);
    const int vz0 = bz * 16 + (ln & 3) * 4;
    if (vx >= a.X || vy >= a.Y || vz0 >= a.Z) return;
    const AffView& w = a.v;
    const double hg = 0.5 * (double)(w.dim - 1), ho = 0.5 * (double)(w.P - 1);
    double base[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) base[r] = fma(w.M[3 * r + 1], (double)vy, fma(w.M[3 * r], (double)vx, w.t[r]));
    // the lane's 4 voxels are 4 K contiguous floats of the output: one 16-byte-vector read-modify-write per lane (four
    // separate 12-byte accesses per lane cost more than the exact kernel's whole run)
    const long t0 = ((long)vx * a.Y + vy) * a.Z + vz0;
    const bool vec = (a.Z & 3) == 0;                              // then vz0 + 3 < Z and t0 * K floats are 16-byte aligned
    float val[FZ][K];
    float4* dst4 = (float4*)(a.out + t0 * K);
    if (ACCUM && vec) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const float4 v4 = dst4[q];
            (&val[0][0])[4 * q] = v4.x; (&val[0][0])[4 * q + 1] = v4.y; (&val[0][0])[4 * q + 2] = v4.z; (&val[0][0])[4 * q + 3] = v4.w;
        }
    }
#pragma unroll
    for (int u = 0; u < FZ; ++u) {
        const int vz = vz0 + u;
        if (vz >= a.Z) continue;
        int n[3]; bool o = false, rk = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double uu = fma(w.M[3 * r + 2], (double)vz, base[r]);
            const double h = r == 2 ? ho : hg;
            const double rr = rint(uu);
            rk |= !(fabs(fabs(uu - rr) - 0.5) > GEOM_TAU);                     // tie (or NaN)
            const double e = fabs(uu - h);
            const bool in = e < h - GEOM_TAU, sure_out = e > h + GEOM_TAU;
            o |= sure_out; rk |= !(in | sure_out);
            n[r] = (int)rr;
        }
        const long t = t0 + u;
        if (rk) {                                                // left as it is here; map_view_fixup_kernel handles it
            const unsigned idx = atomicAdd(a.count, 1u);
            if (idx < a.cap) a.list[idx] = (unsigned)t;
            if (!ACCUM && vec) {
#pragma unroll
                for (int k = 0; k < K; ++k) val[u][k] = 0.f;
            }
            continue;
        }
        float* dst = a.out + t * K;
        if (!ACCUM) {
            float x[K];
            const unsigned off = o ? 0u : (((unsigned)n[2] * (unsigned)w.dim + (unsigned)n[0]) * (unsigned)w.dim + (unsigned)n[1]) * K;
            __builtin_memcpy(x, w.pred + off, K * sizeof(float));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float r_ = o ? (k == 0 ? 1.f : 0.f) : x[k];
                if (vec) val[u][k] = r_; else dst[k] = r_;
            }
        } else if (!o) {
            if (n[2] >= a.p_lo && n[2] < a.p_hi) {            // (w.pred points at plane p_lo of the view)
                const unsigned off = (((unsigned)(n[2] - a.p_lo) * (unsigned)w.dim + (unsigned)n[0]) * (unsigned)w.dim + (unsigned)n[1]) * K;
                float x[K];
                __builtin_memcpy(x, w.pred + off, K * sizeof(float));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (vec) val[u][k] = val[u][k] + a.Wv[k] * x[k]; else dst[k] += a.Wv[k] * x[k];
                }
            }
        } else if (a.owns_oob) {
            if (vec) val[u][0] = val[u][0] + a.Wv[0]; else dst[0] += a.Wv[0];
        }
    }
    if (vec) {
#pragma unroll
        for (int q = 0; q < K; ++q)
            dst4[q] = make_float4((&val[0][0])[4 * q], (&val[0][0])[4 * q + 1], (&val[0][0])[4 * q + 2], (&val[0][0])[4 * q + 3]);
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
    if (!(m.step > 0)) a.kind = 0;
    // error bound of the closed-form index coordinate against the fast paths' margin (see GEOM_TAU)
    if (a.kind && !(64.0 * 1.1102230246251565e-16 * (fabs(a.g0) * a.inv_h + (double)n) < 1e-8 / 4)) a.kind = 0;
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
    const int v = (int)env(ENV_GEOM_FAST);
    if (hipMemcpyToSymbol(HIP_SYMBOL(mpu::g_fast_geometry_dev), &v, sizeof(int)) != hipSuccess)
        return mpu::fail(MPU_EHIP, "%s", "geometry: cannot set the fast-path switch");
    g_fast_host = v;
    done = 1;
    return MPU_OK;
}

// Compose u = M v + t (index units of the view's three axes) for the straight-line fused kernel; false when the view
// is not eligible: a non-uniform axis, offsets beyond 32 bits, or an error bound that is not far below GEOM_TAU.
static bool compose_view(const GridDev& g, const ViewDev& v, int K, AffView& o) {
    if (v.g.kind == 0 || v.offs.kind == 0) return false;
    if ((long)v.P * v.dim * v.dim * K >= (1L << 31)) return false;
    const AxisDev* ax[3] = {&v.g, &v.g, &v.offs};
    const double vmax[3] = {(double)(g.X - 1), (double)(g.Y - 1), (double)(g.Z - 1)};
    for (int r = 0; r < 3; ++r) {
        const long double ih = 1.0L / (long double)ax[r]->step;
        long double tr = 0.0L, mag = 0.0L;
        for (int j = 0; j < 3; ++j) {
            long double m = 0.0L;
            for (int k = 0; k < 3; ++k) m += (long double)v.invb.m[3 * r + k] * (long double)g.A.m[3 * k + j];
            o.M[3 * r + j] = (double)(m * ih);
        }
        for (int k = 0; k < 3; ++k) {
            tr += (long double)v.invb.m[3 * r + k] * (long double)g.c[k];
            long double rk = fabsl((long double)g.c[k]);
            for (int j = 0; j < 3; ++j) rk += fabsl((long double)g.A.m[3 * k + j]) * vmax[j];
            mag += fabsl((long double)v.invb.m[3 * r + k]) * rk;
        }
        o.t[r] = (double)(-(tr + (long double)ax[r]->g0) * ih);
        // magnitude (index units) of every intermediate of both evaluations; each of their <= ~20 roundings
        // contributes at most 2^-53 of it
        const long double span = (long double)(ax[r]->n - 1) / ih;
        mag = (mag + fabsl((long double)ax[r]->g0) + fabsl((long double)ax[r]->g0 + span)) * ih;
        const long double err = 64.0L * 1.1102230246251565e-16L * mag;
        if (!(err < GEOM_TAU / 4) || !isfinite((double)mag)) return false;
        if (!isfinite(o.t[r]) || !isfinite(o.M[3 * r]) || !isfinite(o.M[3 * r + 1]) || !isfinite(o.M[3 * r + 2])) return false;
    }
    o.pred = v.pred; o.dim = v.dim; o.P = v.P;
    for (int r = 0; r < 3; ++r) o.S[r] = (int)llrint(o.M[3 * r + 2] * 1048576.0 < -2147483000.0 ? -2147483000.0
                                                     : (o.M[3 * r + 2] * 1048576.0 > 2147483000.0 ? 2147483000.0 : o.M[3 * r + 2] * 1048576.0));
    o._pad = 0;
    return true;
}
// fixed-point screening (map_fuse_fast_kernel<K, 2, true>): every axis <= 1024 nodes, z step of every axis at most 8 index units
static bool fx_eligible(const AffView& o) {
    if (o.dim > 1024 || o.P > 1024) return false;
    for (int r = 0; r < 3; ++r) if (!(fabs(o.M[3 * r + 2]) <= 8.0)) return false;
    return true;
}

// Work list of the straight-line kernels: [0], [1] = two counters, [16 ..] = sample / voxel indices. One buffer per
// (device, stream), allocated on first use and kept: calls on one stream are ordered, calls on different streams use
// different buffers. Call k of a stream counts in counter k & 1, and its fix-up kernel -- the last thing the call
// launches -- zeroes the other counter for call k + 1 (no memset launch per call).
constexpr unsigned FUSE_LIST_CAP = 1u << 20;
struct FuseScratch { int dev; hipStream_t st; unsigned* buf; unsigned seq; };
static int fuse_scratch(hipStream_t st, unsigned** count, unsigned** next_count, unsigned** list) {
    static std::mutex mu;
    static std::vector<FuseScratch> slots;
    int dev = 0;
    MPU_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    FuseScratch* hit = nullptr;
    for (FuseScratch& s : slots) if (s.dev == dev && s.st == st) hit = &s;
    if (!hit) {
        unsigned* p = nullptr;
        MPU_CHECK_HIP(hipMalloc((void**)&p, (size_t)(FUSE_LIST_CAP + 16) * sizeof(unsigned)));
        MPU_CHECK_HIP(hipMemset(p, 0, 16 * sizeof(unsigned)));
        slots.push_back({dev, st, p, 0u});
        hit = &slots.back();
    }
    *count = hit->buf + (hit->seq & 1u);
    *next_count = hit->buf + ((hit->seq + 1u) & 1u);
    *list = hit->buf + 16;
    ++hit->seq;
    return MPU_OK;
}

// The two counters only stay consistent when BOTH launches of a call (fast kernel, fix-up) were issued: an early
// return in between (launch error, unsupported class count) would leave the next call a stale non-zero count and
// replay list entries of an older, possibly larger volume. The guard zeroes both counters on such a path.
struct FuseCounterGuard {
    hipStream_t st; unsigned* list; bool done = false;
    ~FuseCounterGuard() { if (!done && list) (void)hipMemsetAsync(list - 16, 0, 16 * sizeof(unsigned), st); }
};

// map_real_space_pred / sharded accumulate of one view: straight-line kernel + exact fix-up when the view is eligible
static int launch_map_view(const MapArgs& a, int n_classes, bool accum, hipStream_t st) {
    MapFastArgs f;
    const bool fast = fast_path_host() && n_classes >= 1 && n_classes <= 16 && (long)a.grid.X * a.grid.Y * a.grid.Z < (1L << 32) &&
                      compose_view(a.grid, a.view, n_classes, f.v);
    if (fast) {
        unsigned* nxt = nullptr;
        { const int rc_ = fuse_scratch(st, &f.count, &nxt, &f.list); if (rc_) return rc_; }
        FuseCounterGuard guard{st, f.list};
        f.cap = FUSE_LIST_CAP;
        f.X = a.grid.X; f.Y = a.grid.Y; f.Z = a.grid.Z;
        f.Wv = a.Wv; f.p_lo = a.p_lo; f.p_hi = a.p_hi; f.owns_oob = a.owns_oob; f.out = a.out;
        const int nxb = cdiv(f.X, 8), nyb = cdiv(f.Y, 8), nzb = cdiv(f.Z, 16);
        f.px2 = f.py2 = 0;
        while ((1 << f.px2) < nxb) ++f.px2;
        while ((1 << f.py2) < nyb) ++f.py2;
        const long padded = (1L << f.px2) * (1L << f.py2) * nzb;
        if (padded < (1L << 31)) {
            f.nblk8 = (unsigned)((padded + 7) / 8);
            const dim3 g(f.nblk8 * 8u), b(256);
            if (accum) { MPU_DISPATCH_K(n_classes, (map_view_fast_kernel<KK, true><<<g, b, 0, st>>>(f))); }
            else       { MPU_DISPATCH_K(n_classes, (map_view_fast_kernel<KK, false><<<g, b, 0, st>>>(f))); }
            { const int rc_ = launch_ok(); if (rc_) return rc_; }
            if (accum) { MPU_DISPATCH_K(n_classes, (map_view_fixup_kernel<KK, true><<<dim3(64), dim3(256), 0, st>>>(a, f.list, f.count, f.cap, nxt))); }
            else       { MPU_DISPATCH_K(n_classes, (map_view_fixup_kernel<KK, false><<<dim3(64), dim3(256), 0, st>>>(a, f.list, f.count, f.cap, nxt))); }
            if (sched_log_on()) sched_note("map_view fast accum=%d K=%d", accum ? 1 : 0, n_classes);
            const int rc_ = launch_ok();
            guard.done = rc_ == MPU_OK;
            return rc_;
        }
        // (unreachable in practice: the scratch sequence number advanced without a launch; the guard zeroes the counters)
    }
    if (sched_log_on()) sched_note("map_view generic accum=%d K=%d", accum ? 1 : 0, n_classes);
    if (accum) { MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, true><<<dim3(brick_grid(a.grid)), dim3(256), 0, st>>>(a))); }
    else       { MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, false><<<dim3(brick_grid(a.grid)), dim3(256), 0, st>>>(a))); }
    return launch_ok();
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

/* test aid: number of pseudo-random (num, den) pairs, den a cell width of the axis, for which the sampler's division
 * differs from the IEEE quotient (must be 0) */
int mpu_geometry_check_cell_division(const mpu_axis* axis, int64_t count, uint64_t seed, uint64_t* n_bad) {
    MPU_REQUIRE(axis && n_bad && count >= 0, "mpu_geometry_check_cell_division: bad argument");
    MPU_REQUIRE((axis->kind == 1 || axis->kind == 2) && axis->n >= 2 && axis->step > 0, "mpu_geometry_check_cell_division: need a closed-form axis");
    unsigned long long* d = nullptr;
    MPU_CHECK_HIP(hipMalloc((void**)&d, sizeof(unsigned long long)));
    MPU_CHECK_HIP(hipMemset(d, 0, sizeof(unsigned long long)));
    cell_div_check_kernel<<<dim3(2048), dim3(256)>>>(axis->start, axis->step, axis->n, axis->kind, (long)count,
                                                     (unsigned long long)seed, d);
    unsigned long long h = 0;
    const hipError_t e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    MPU_CHECK_HIP(e);
    *n_bad = (uint64_t)h;
    return MPU_OK;
}

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
    // straight-line kernel: ImagePair voxel axes (kind 2), 1 or 2 channels, 32-bit element offsets
    // (single planes -- the train-time sampler cuts one candidate plane per call -- stay on the one-launch kernel: the
    // straight-line kernel needs a second launch for its work list)
    constexpr long fast_min = 262144;
    const bool fast = fast_path_host() && (long)a.P * a.dim * a.dim >= fast_min &&
                      a.ax.kind != 0 && a.ay.kind == a.ax.kind && a.az.kind == a.ax.kind && (a.C == 1 || a.C == 2) &&
                      (long)a.X * a.Y * a.Z * a.C < (1L << 29) && a.P < 65536 && (long)a.P * a.dim * a.dim < (1L << 32);
    if (fast) {
        const dim3 g((unsigned)((a.dim + 31) / 32), (unsigned)((a.dim + 7) / 8), (unsigned)a.P), b(256);
        hipStream_t st = (hipStream_t)stream;
        unsigned *cnt = nullptr, *nxt = nullptr, *lst = nullptr;
        { const int rc_ = fuse_scratch(st, &cnt, &nxt, &lst); if (rc_) return rc_; }
        FuseCounterGuard guard{st, lst};
#define MPU_SAMPLE_FAST(KIND_) \
        if (a.C == 1) { if (a.out_lab) sample_fast_kernel<KIND_, 1, true><<<g, b, 0, st>>>(a, lst, cnt, FUSE_LIST_CAP); else sample_fast_kernel<KIND_, 1, false><<<g, b, 0, st>>>(a, lst, cnt, FUSE_LIST_CAP); } \
        else          { if (a.out_lab) sample_fast_kernel<KIND_, 2, true><<<g, b, 0, st>>>(a, lst, cnt, FUSE_LIST_CAP); else sample_fast_kernel<KIND_, 2, false><<<g, b, 0, st>>>(a, lst, cnt, FUSE_LIST_CAP); }
        if (a.ax.kind == 1) { MPU_SAMPLE_FAST(1) } else { MPU_SAMPLE_FAST(2) }
#undef MPU_SAMPLE_FAST
        { const int rc_ = launch_ok(); if (rc_) return rc_; }
        sample_fixup_kernel<<<dim3(64), dim3(256), 0, st>>>(a, lst, cnt, FUSE_LIST_CAP, nxt);
        if (sched_log_on()) sched_note("sample fast kind=%d C=%d labels=%d", a.ax.kind, a.C, a.out_lab ? 1 : 0);
        const int rc_ = launch_ok();
        guard.done = rc_ == MPU_OK;
        return rc_;
    }
    if (sched_log_on()) sched_note("sample generic kinds=%d%d%d C=%d labels=%d", a.ax.kind, a.ay.kind, a.az.kind, a.C, a.out_lab ? 1 : 0);
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
    return launch_map_view(a, n_classes, false, (hipStream_t)stream);
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
    return launch_map_view(a, n_classes, true, (hipStream_t)stream);
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
    // straight-line kernel on composed affine maps + exact fix-up of the flagged voxels, when every view is eligible
    bool fast = fast_path_host() && n_classes >= 1 && n_classes <= 16 && (long)a.grid.X * a.grid.Y * a.grid.Z < (1L << 32);
    FuseFastArgs f;
    for (int v = 0; fast && v < n_views; ++v) fast = compose_view(a.grid, a.views[v], n_classes, f.v[v]);
    if (fast) {
        hipStream_t st = (hipStream_t)stream;
        unsigned* nxt = nullptr;
        { const int rc_ = fuse_scratch(st, &f.count, &nxt, &f.list); if (rc_) return rc_; }
        FuseCounterGuard guard{st, f.list};
        constexpr int cfg = 2;                                   // brick / lane layout (0 and 1 measured slower: DESIGN section 4.6)
        f.V = n_views; f.X = a.grid.X; f.Y = a.grid.Y; f.Z = a.grid.Z;
        f.W = d_W; f.b = d_b; f.sum_fusion = sum_fusion; f.probs = d_probs; f.labels = d_labels;
        f.cap = FUSE_LIST_CAP;
        long nblk = cfg ? (long)cdiv(f.X, 8) * cdiv(f.Y, 8) * cdiv(f.Z, 16) : (long)cdiv(f.X, 4) * cdiv(f.Y, 4) * cdiv(f.Z, 64);    // bricks
        constexpr int morton = 2;                                // brick order: z slowest inside an XCD's run (1 = z fastest, 3 = interleaved: section 4.6)
        f.morton = 0; f.px2 = f.py2 = 0;
        if (morton) {
            const int nxb = cdiv(f.X, cfg ? 8 : 4), nyb = cdiv(f.Y, cfg ? 8 : 4), nzb = cdiv(f.Z, cfg ? 16 : 64);
            while ((1 << f.px2) < nxb) ++f.px2;
            while ((1 << f.py2) < nyb) ++f.py2;
            const long padded = (1L << f.px2) * (1L << f.py2) * nzb;
            if (padded < (1L << 31)) { f.morton = (morton >= 2 && padded % (8L * nzb) == 0) ? morton : 1; nblk = padded; }
        }
        f.nblk8 = (unsigned)((nblk + 7) / 8);
        const dim3 g(f.nblk8 * 8u), b(256);
        const bool fx_on = env(ENV_FUSE_FX) != 0;                // 0: the fp64 index arithmetic of round 2 (A/B, equality test)
        bool fx = fx_on && cfg == 2;
        for (int v = 0; fx && v < n_views; ++v) fx = fx_eligible(f.v[v]);
        if (fx)            { MPU_DISPATCH_K(n_classes, (map_fuse_fast_kernel<KK, 2, true><<<g, b, 0, st>>>(f))); }
        else if (cfg == 2) { MPU_DISPATCH_K(n_classes, (map_fuse_fast_kernel<KK, 2><<<g, b, 0, st>>>(f))); }
        else if (cfg == 1) { MPU_DISPATCH_K(n_classes, (map_fuse_fast_kernel<KK, 1><<<g, b, 0, st>>>(f))); }
        else               { MPU_DISPATCH_K(n_classes, (map_fuse_fast_kernel<KK, 0><<<g, b, 0, st>>>(f))); }
        { const int rc_ = launch_ok(); if (rc_) return rc_; }
        MPU_DISPATCH_K(n_classes, (map_fuse_fixup_kernel<KK><<<dim3(256), dim3(512), 0, st>>>(a, f.list, f.count, f.cap, nxt)));
        if (sched_log_on()) sched_note("map_fuse fast views=%d K=%d brick=%d fx=%d", n_views, n_classes, cfg, fx ? 1 : 0);
        const int rc_ = launch_ok();
        guard.done = rc_ == MPU_OK;
        return rc_;
    }
    if (sched_log_on()) sched_note("map_fuse generic views=%d K=%d", n_views, n_classes);
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
