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

// RegularGridInterpolator._find_indices for one axis: i = searchsorted_left(g,x)-1
// clipped to [0,n-2]; y = (x-g[i])/(g[i+1]-g[i]); oob = x<g[0] || x>g[n-1].
__device__ __forceinline__ void find_cell(const double* __restrict__ g, int n, double x,
                                          int& i, double& y, bool& oob) {
    const double g0 = g[0], gl = g[n - 1];
    oob = (x < g0) || (x > gl);
    int c;
    if (!(x > g0)) c = 0;
    else if (x > gl) c = n - 2;
    else {
        c = (int)ceil((x - g0) / (g[1] - g0)) - 1;
        c = c < 0 ? 0 : (c > n - 2 ? n - 2 : c);
        while (c < n - 2 && g[c + 1] < x) ++c;     // need x <= g[c+1]
        while (c > 0 && g[c] >= x) --c;            // need g[c] <  x
    }
    i = c;
    y = (x - g[c]) / (g[c + 1] - g[c]);
}

struct SampleArgs {
    const float* vol; const uint8_t* labels;
    int X, Y, Z, C;
    const double *ax, *ay, *az, *offsets;
    Mat3 basis, rot; int has_rot;
    int dim, P; double g_start, g_step;
    const float* bg; uint8_t bg_class;
    const double *center, *scale;
    float* out; uint8_t* out_lab;
};

__global__ __launch_bounds__(256) void sample_view_planes_kernel(SampleArgs a) {
    const long total = (long)a.P * a.dim * a.dim;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long)gridDim.x * blockDim.x) {
        const int j = (int)(t % a.dim);
        const int i = (int)((t / a.dim) % a.dim);
        const int p = (int)(t / ((long)a.dim * a.dim));
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
        find_cell(a.ax, a.X, rx, i0, y0, o0);
        find_cell(a.ay, a.Y, ry, i1, y1, o1);
        find_cell(a.az, a.Z, rz, i2, y2, o2);
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
}

// ------------------------------------------------------------------------- //
struct ViewDev {
    Mat3 invb; const float* pred; const double* g; const double* offs; int dim, P;
};
struct GridDev { Mat3 A; double c[3]; int X, Y, Z; };

// nearest lookup of one view for voxel (x,y,z): returns element offset of the
// K-vector in pred[P,dim,dim,K], or -1 when out of the view's box; plane index in pl.
__device__ __forceinline__ long view_lookup(const ViewDev& v, double rx, double ry, double rz,
                                            int K, int& pl) {
    double qx, qy, qz;
    mat3_apply(v.invb, rx, ry, rz, qx, qy, qz);
    int i0, i1, i2; double y0, y1, y2; bool o0, o1, o2;
    find_cell(v.g, v.dim, qx, i0, y0, o0);
    find_cell(v.g, v.dim, qy, i1, y1, o1);
    find_cell(v.offs, v.P, qz, i2, y2, o2);
    if (o0 || o1 || o2) { pl = -1; return -1; }
    const int n0 = (y0 <= .5) ? i0 : i0 + 1;
    const int n1 = (y1 <= .5) ? i1 : i1 + 1;
    const int n2 = (y2 <= .5) ? i2 : i2 + 1;
    pl = n2;
    return (((long)n2 * v.dim + n0) * v.dim + n1) * K;
}

__device__ __forceinline__ void voxel_real(const GridDev& g, long t, double& rx, double& ry, double& rz) {
    const int z = (int)(t % g.Z);
    const int y = (int)((t / g.Z) % g.Y);
    const int x = (int)(t / ((long)g.Z * g.Y));
    mat3_apply(g.A, (double)x, (double)y, (double)z, rx, ry, rz);
    rx = rx - g.c[0]; ry = ry - g.c[1]; rz = rz - g.c[2];
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

template <int K>
__global__ __launch_bounds__(256) void map_fuse_kernel(FuseArgs a) {
    const long total = (long)a.grid.X * a.grid.Y * a.grid.Z;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long)gridDim.x * blockDim.x) {
        double rx, ry, rz;
        voxel_real(a.grid, t, rx, ry, rz);
        float z[K];
#pragma unroll
        for (int k = 0; k < K; ++k) z[k] = 0.f;
        for (int v = 0; v < a.V; ++v) {
            int pl;
            const long off = view_lookup(a.views[v], rx, ry, rz, K, pl);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float x = (off >= 0) ? a.views[v].pred[off + k] : (k == 0 ? 1.f : 0.f);
                z[k] = a.sum_fusion ? (z[k] + x) : (z[k] + a.W[v * K + k] * x);
            }
        }
        if (!a.sum_fusion) {
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = z[k] + a.b[k];
        }
        softmax_argmax_store<K>(z, !a.sum_fusion, t, a.probs, a.labels);
    }
}

struct MapArgs {
    GridDev grid; ViewDev view; const float* Wv; int p_lo, p_hi, owns_oob; float* out;
};

// ACCUM=false: mapped[t,:] = nearest (map_real_space_pred). ACCUM=true: z[t,:] += Wv*nearest
// restricted to planes [p_lo,p_hi) (pred points at plane p_lo).
template <int K, bool ACCUM>
__global__ __launch_bounds__(256) void map_view_kernel(MapArgs a) {
    const long total = (long)a.grid.X * a.grid.Y * a.grid.Z;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long)gridDim.x * blockDim.x) {
        double rx, ry, rz;
        voxel_real(a.grid, t, rx, ry, rz);
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
static void to_view(const mpu_view_pred& v, ViewDev& d) {
    to_mat3(v.inv_basis, d.invb);
    d.pred = v.d_pred; d.g = v.d_g; d.offs = v.d_offsets; d.dim = v.dim; d.P = v.n_planes;
}
static void to_grid(const mpu_voxel_grid& g, GridDev& d) {
    to_mat3(g.A, d.A);
    for (int i = 0; i < 3; ++i) d.c[i] = g.center[i];
    d.X = g.shape[0]; d.Y = g.shape[1]; d.Z = g.shape[2];
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

extern "C" {

int mpu_abi_version(void) { return 1; }
const char* mpu_last_error(void) { return mpu::g_err; }

int mpu_sample_view_planes(const float* d_vol, const uint8_t* d_labels, const int32_t vol_shape[4],
                           const double* d_ax, const double* d_ay, const double* d_az,
                           const mpu_view_geom* geom, const double* d_offsets,
                           const float* d_bg, uint8_t bg_class,
                           const double* d_center, const double* d_scale,
                           float* d_out, uint8_t* d_out_lab, void* stream) {
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
    a.ax = d_ax; a.ay = d_ay; a.az = d_az; a.offsets = d_offsets;
    to_mat3(geom->basis, a.basis); to_mat3(geom->rot, a.rot); a.has_rot = geom->has_rot;
    a.dim = geom->dim; a.P = geom->n_planes; a.g_start = geom->g_start; a.g_step = geom->g_step;
    a.bg = d_bg; a.bg_class = bg_class; a.center = d_center; a.scale = d_scale;
    a.out = d_out; a.out_lab = d_out_lab;
    const long total = (long)a.P * a.dim * a.dim;
    sample_view_planes_kernel<<<dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream>>>(a);
    return launch_ok();
}

int mpu_map_view_nearest(const mpu_voxel_grid* grid, const mpu_view_pred* view, int32_t n_classes,
                         float* d_mapped, void* stream) {
    MPU_REQUIRE(grid && view && d_mapped && view->d_pred && view->d_g && view->d_offsets,
                "mpu_map_view_nearest: null argument");
    MPU_REQUIRE(view->dim >= 2 && view->n_planes >= 2, "mpu_map_view_nearest: view needs dim>=2, planes>=2");
    MapArgs a; to_grid(*grid, a.grid); to_view(*view, a.view);
    a.Wv = nullptr; a.p_lo = 0; a.p_hi = view->n_planes; a.owns_oob = 1; a.out = d_mapped;
    const long total = (long)a.grid.X * a.grid.Y * a.grid.Z;
    MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, false><<<dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream>>>(a)));
    return launch_ok();
}

int mpu_map_accumulate_view(const mpu_voxel_grid* grid, const mpu_view_pred* view, int32_t n_classes,
                            const float* d_Wv, int32_t p_lo, int32_t p_hi, int32_t owns_oob,
                            float* d_z, void* stream) {
    MPU_REQUIRE(grid && view && d_z && d_Wv && view->d_pred && view->d_g && view->d_offsets,
                "mpu_map_accumulate_view: null argument");
    MPU_REQUIRE(0 <= p_lo && p_lo < p_hi && p_hi <= view->n_planes, "mpu_map_accumulate_view: bad plane range");
    MapArgs a; to_grid(*grid, a.grid); to_view(*view, a.view);
    a.Wv = d_Wv; a.p_lo = p_lo; a.p_hi = p_hi; a.owns_oob = owns_oob; a.out = d_z;
    const long total = (long)a.grid.X * a.grid.Y * a.grid.Z;
    MPU_DISPATCH_K(n_classes, (map_view_kernel<KK, true><<<dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream>>>(a)));
    return launch_ok();
}

int mpu_map_fuse_views(const mpu_voxel_grid* grid, const mpu_view_pred* views, int32_t n_views,
                       int32_t n_classes, const float* d_W, const float* d_b, int32_t sum_fusion,
                       float* d_probs, uint8_t* d_labels, void* stream) {
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
    const long total = (long)a.grid.X * a.grid.Y * a.grid.Z;
    MPU_DISPATCH_K(n_classes, (map_fuse_kernel<KK><<<dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream>>>(a)));
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
