// Elastic2D on-the-fly augmentation of training slices (reference:
// mpunet/augmentation/elastic_deformation.py:6-69). Compiled with -ffp-contract=off: the arithmetic restates the
// reference's fp64 NumPy / SciPy operations one by one and is bit-exact against its outputs (tests/golden/
// elastic_golden.npz):
//   displacement = gaussian_filter(2*noise - 1, sigma, mode="constant", cval=0) * alpha   (two fields)
//       scipy correlate1d, symmetric branch: out[l] = in[l]*w0 + sum_{j=-r..-1} (in[l+j] + in[l-j]) * w[j],
//       axis 0 first, then axis 1; weights from the host (exp(-x^2/2sigma^2), normalised, radius int(4 sigma + .5))
//   image  = RegularGridInterpolator(linear, fill=bg)(x + dx, y + dy)   per channel, f32 values x f64 weights
//   labels = RegularGridInterpolator(nearest, fill=0)(x + dx, y + dy)
#include "kernels.h"
#include "../../include/mpunet_hip.h"

namespace mpu {
namespace {

// one 1-D pass over both noise fields; FIRST: input = 2*noise - 1 along axis 0 (rows), else along axis 1
template <bool FIRST>
__global__ __launch_bounds__(256) void elastic_blur_kernel(const double* __restrict__ in, int H, int W,
                                                           const double* __restrict__ w, int radius,
                                                           double* __restrict__ out) {
    const long n = 2L * H * W;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
        const int f = (int)(t / ((long)H * W)); const int r = (int)(t % ((long)H * W));
        const int i = r / W, j = r % W;
        const double* src = in + (long)f * H * W;
        auto at = [&](int a, int b) -> double {
            if ((unsigned)a >= (unsigned)H || (unsigned)b >= (unsigned)W) return 0.0;
            const double v = src[(long)a * W + b];
            return FIRST ? v * 2 - 1 : v;
        };
        double tmp = at(i, j) * w[radius];
        for (int jj = -radius; jj < 0; ++jj) {
            const double lo = FIRST ? at(i + jj, j) : at(i, j + jj);
            const double hi = FIRST ? at(i - jj, j) : at(i, j - jj);
            tmp = tmp + (lo + hi) * w[radius + jj];
        }
        out[t] = tmp;
    }
}

struct Cell { int idx; double nd; bool oob; };
__device__ __forceinline__ Cell find_cell_arange(double x, int n) {
    // np.searchsorted(arange(n), x) - 1, clipped to [0, n-2]; norm distance (x - g[i]) / (g[i+1] - g[i])
    Cell c;
    int k;
    if (!(x > 0.0)) k = 0;                               // first grid value >= x (NaN compares false: index n)
    else if (x > (double)(n - 1)) k = n;
    else k = (int)ceil(x);
    if (x != x) k = n;
    int i = k - 1;
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    c.idx = i;
    c.nd = (x - (double)i) / 1.0;
    c.oob = (x < 0.0) || (x > (double)(n - 1));
    return c;
}

__global__ __launch_bounds__(256) void elastic_warp_kernel(const float* __restrict__ image, const uint8_t* __restrict__ labels,
                                                           int H, int W, int C, const double* __restrict__ disp,
                                                           double alpha, const float* __restrict__ bg,
                                                           float* __restrict__ out_image, uint8_t* __restrict__ out_labels) {
    const long n = (long)H * W;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < n; t += (long)gridDim.x * 256) {
        const int i = (int)(t / W), j = (int)(t % W);
        const double x = (double)i + disp[t] * alpha;
        const double y = (double)j + disp[n + t] * alpha;
        const Cell cx = find_cell_arange(x, H), cy = find_cell_arange(y, W);
        const bool oob = cx.oob || cy.oob;
        if (out_image) {
            // edges in itertools.product order; weight = (1 * wx) * wy; out = ((0 + v00 w00) + v01 w01) + ...
            const double wx0 = 1 - cx.nd, wx1 = cx.nd, wy0 = 1 - cy.nd, wy1 = cy.nd;
            const double w00 = (1. * wx0) * wy0, w01 = (1. * wx0) * wy1, w10 = (1. * wx1) * wy0, w11 = (1. * wx1) * wy1;
            const long o00 = ((long)cx.idx * W + cy.idx) * C, o01 = o00 + C, o10 = o00 + (long)W * C, o11 = o10 + C;
            for (int c = 0; c < C; ++c) {
                double acc = 0.;
                acc = acc + (double)image[o00 + c] * w00;
                acc = acc + (double)image[o01 + c] * w01;
                acc = acc + (double)image[o10 + c] * w10;
                acc = acc + (double)image[o11 + c] * w11;
                out_image[t * C + c] = oob ? bg[c] : (float)acc;
            }
        }
        if (out_labels) {
            const int si = cx.nd <= .5 ? cx.idx : cx.idx + 1, sj = cy.nd <= .5 ? cy.idx : cy.idx + 1;
            out_labels[t] = oob ? (uint8_t)0 : labels[(long)si * W + sj];
        }
    }
}

// class-presence mask and "not all background" flag of one sampled slice (the accept / reject test of the
// train-time sampler: validate_lab / validate_lab_vec / is_valid_im, isotrophic_live_view_sequence.py:91-128)
__global__ __launch_bounds__(256) void plane_stats_kernel(const uint8_t* __restrict__ y, const float* __restrict__ x, long npix,
                                                          int C, const float* __restrict__ bg, unsigned* __restrict__ out) {
    unsigned mask = 0, nonbg = 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < npix; t += (long)gridDim.x * 256) {
        if (y) { const unsigned l = y[t]; if (l < 32) mask |= 1u << l; }
        if (x)
            for (int c = 0; c < C; ++c) {
                const float b = bg[c];
                const double lhs = (double)fabsf(x[t * C + c] - b), rhs = 1e-8 + 1e-5 * fabs((double)b);   // ~np.isclose
                if (!(lhs <= rhs)) nonbg = 1;
            }
    }
    for (int o = 32; o > 0; o >>= 1) { mask |= __shfl_xor(mask, o, 64); nonbg |= __shfl_xor(nonbg, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (mask) atomicOr(&out[0], mask);
        if (nonbg) atomicOr(&out[1], nonbg);
    }
}

}  // namespace
}  // namespace mpu

using namespace mpu;

extern "C" {

int mpu_plane_stats(const uint8_t* d_labels, const float* d_image, int64_t n_pixels, int32_t n_channels,
                    const float* d_bg, uint32_t* d_out2, void* stream) {
    MPU_REQUIRE((d_labels || d_image) && d_out2 && n_pixels >= 1, "mpu_plane_stats: null argument");
    MPU_REQUIRE(!d_image || (d_bg && n_channels >= 1), "mpu_plane_stats: image needs its background values");
    hipStream_t st = (hipStream_t)stream;
    MPU_CHECK_HIP(hipMemsetAsync(d_out2, 0, 2 * sizeof(uint32_t), st));
    long blocks = (n_pixels + 255) / 256; if (blocks > 256) blocks = 256;
    plane_stats_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_labels, d_image, (long)n_pixels, n_channels, d_bg, d_out2);
    return launch_ok();
}

/* One candidate slice of the train-time sampler in ONE call (round 6: the sampler's host loop had become what bounds `mp train`):
 * mpu_sample_view_planes for a single plane followed by mpu_plane_stats of that plane. */
int mpu_sample_plane_stats(const float* d_vol, const uint8_t* d_labels, const int32_t vol_shape[4],
                           const double* d_ax, const double* d_ay, const double* d_az,
                           const mpu_view_geom* geom, const double* d_offset,
                           const float* d_bg, uint8_t bg_class, const double* d_center, const double* d_scale,
                           float* d_out, uint8_t* d_out_lab, const float* d_bg_scaled, uint32_t* d_stats2, void* stream) {
    MPU_REQUIRE(geom && geom->n_planes == 1 && d_stats2 && vol_shape, "mpu_sample_plane_stats: one plane and a statistics buffer");
    const int rc = mpu_sample_view_planes(d_vol, d_labels, vol_shape, d_ax, d_ay, d_az, geom, d_offset, d_bg, bg_class, d_center,
                                          d_scale, d_out, d_out_lab, stream);
    if (rc) return rc;
    return mpu_plane_stats(d_out_lab, d_out, (int64_t)geom->dim * geom->dim, vol_shape[3], d_bg_scaled, d_stats2, stream);
}

int64_t mpu_elastic_workspace_doubles(int32_t H, int32_t W) { return 4L * H * W; }

int mpu_elastic_transform_2d(const float* d_image, const uint8_t* d_labels, int32_t H, int32_t W, int32_t C,
                             const double* d_noise, const double* d_gauss_w, int32_t radius, double alpha,
                             const float* d_bg, double* d_workspace, float* d_out_image, uint8_t* d_out_labels,
                             void* stream) {
    MPU_REQUIRE(d_noise && d_gauss_w && d_workspace && (d_out_image || d_out_labels),
                "mpu_elastic_transform_2d: null argument");
    MPU_REQUIRE((!d_out_image || (d_image && d_bg)) && (!d_out_labels || d_labels),
                "mpu_elastic_transform_2d: an output needs its input");
    MPU_REQUIRE(H >= 2 && W >= 2 && C >= 1 && radius >= 0, "mpu_elastic_transform_2d: need H, W >= 2, C >= 1");
    hipStream_t st = (hipStream_t)stream;
    const long n2 = 2L * H * W;
    long blocks = (n2 + 255) / 256; if (blocks > 4096) blocks = 4096;
    double* t0 = d_workspace; double* t1 = d_workspace + n2;
    elastic_blur_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(d_noise, H, W, d_gauss_w, radius, t0);
    elastic_blur_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(t0, H, W, d_gauss_w, radius, t1);
    elastic_warp_kernel<<<dim3((unsigned)((blocks + 1) / 2)), dim3(256), 0, st>>>(d_image, d_labels, H, W, C, t1, alpha, d_bg,
                                                                                 d_out_image, d_out_labels);
    return launch_ok();
}

}  // extern "C"
