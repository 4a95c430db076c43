// Resampling between voxel grids on the device (round 4): step 4 of trainer.preprocess_patient (reference
// run_brats2021_inference_singlethread.py:89 -> nnU-Net v1 GenericPreprocessor.resample_and_normalize -> resample_patient) and
// the resampling inside save_segmentation_nifti_from_softmax(..., order=1, force_separate_z=None, interpolation_order_z=0)
// (driver :131-138, :144-156).  Both are un-vendored nnU-Net v1 code (PARITY UNPINNED, see oracle/tiler_ref.py): they resize with
// skimage.transform.resize(order, mode='edge', anti_aliasing=False), i.e. spline interpolation on the half-pixel-centred grid
// x_in = (x_out + 0.5) * n_in / n_out - 0.5 with edge replication, clipped to the input's range - what
// scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) computes - and, along a low-resolution axis treated separately,
// with scipy.ndimage.map_coordinates(order_z, mode='nearest') on the same grid.
//
// A tensor-product spline on a grid is separable: prefilter and evaluation along different axes commute, so an N-D resize is one
// 1-D pass per axis (each pass: optional B-spline prefilter of the line, then evaluation at the new positions), in any order,
// with per-axis interpolation orders - which also covers nnU-Net's "separate z" mode (order 3 in plane, order 0 along z).
//   order 0: nearest sample, floor(x + 0.5);  order 1: linear;  order 3: cubic B-spline with scipy's treatment of mode 'nearest':
//   the line is padded by 12 edge samples on either side, filtered with mirror boundary conditions (pole sqrt(3) - 2, gain 6, exact
//   initialisation sums, fp64) and evaluated with indices clamped to the padded line.
// HBM-bound streaming work: one thread per output element (evaluation) / per line (prefilter recursion).
#include "kernels.h"

namespace mi355 {

constexpr int RS_PAD = 12;  // scipy.ndimage._interpolation._prepad_for_spline_filter: npad = 12 for mode 'nearest'

// coef [outer][n + 2 * RS_PAD][inner] <- cubic B-spline coefficients of in [outer][n][inner] along the middle axis
__global__ __launch_bounds__(256) void spline3_prefilter_kernel(const float *__restrict__ in, float *__restrict__ coef, long outer, int n, long inner) {
    const long line = (long)blockIdx.x * 256 + threadIdx.x;
    if (line >= outer * inner) return;
    const long o = line / inner, i = line - o * inner;
    const float *src = in + o * n * inner + i;
    const int m = n + 2 * RS_PAD;
    float *dst = coef + o * m * inner + i;
    const double z = -0.26794919243112270647;  // sqrt(3) - 2
    auto at = [&](int k) { const int s = k - RS_PAD; return (double)src[(long)(s < 0 ? 0 : (s >= n ? n - 1 : s)) * inner]; };
    // gain, then causal initialisation with mirror boundaries: c0 = sum_k z^k c[k] (+ the mirrored tail), exactly as
    // scipy's _init_causal_mirror (full sum over the line)
    const double gain = 6.0;  // (1 - z) (1 - 1 / z)
    const double z_n_1 = pow(z, (double)(m - 1));
    double c0 = gain * at(0) + z_n_1 * gain * at(m - 1);
    double z_i = z;
    for (int k = 1; k < m - 1; ++k) { c0 += z_i * (gain * at(k) + z_n_1 * gain * at(m - 1 - k)); z_i *= z; }
    c0 /= 1.0 - z_n_1 * z_n_1;
    // causal pass (coefficients kept in the output line as fp32 between the passes would lose the fp64 recursion: recompute the
    // causal values in a second sweep is not possible without storage, so the causal result goes to the output as fp32 and the
    // anticausal pass re-reads it - 2^-24 relative per element, far inside the comparison tolerance)
    double prev = c0;
    dst[0] = (float)c0;
    for (int k = 1; k < m; ++k) { prev = gain * at(k) + z * prev; dst[(long)k * inner] = (float)prev; }
    // anticausal initialisation (mirror) and pass
    double cn = (z * (double)dst[(long)(m - 2) * inner] + (double)dst[(long)(m - 1) * inner]) * z / (z * z - 1.0);
    dst[(long)(m - 1) * inner] = (float)cn;
    double next = cn;
    for (int k = m - 2; k >= 0; --k) { next = z * (next - (double)dst[(long)k * inner]); dst[(long)k * inner] = (float)next; }
}

// out [outer][n_out][inner] <- in resampled along the middle axis.  ORDER 3 reads the prefiltered, padded line (n_in + 2 RS_PAD).
template <int ORDER>
__global__ __launch_bounds__(256) void resize_axis_kernel(const float *__restrict__ in, float *__restrict__ out, long outer, int n_in, int n_out,
                                                          long inner, double scale) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = outer * n_out * inner;
    if (idx >= total) return;
    const long i = idx % inner;
    const long t = idx / inner;
    const int k = (int)(t % n_out);
    const long o = t / n_out;
    const double x = ((double)k + 0.5) * scale - 0.5;  // half-pixel-centred grid (skimage.transform.resize / zoom(grid_mode=True))
    if (ORDER == 0) {
        int j = (int)floor(x + 0.5);
        j = j < 0 ? 0 : (j >= n_in ? n_in - 1 : j);
        out[idx] = in[(o * n_in + j) * inner + i];
    } else if (ORDER == 1) {
        // scipy clamps the COORDINATE to [0, n - 1] for mode 'nearest' first (map_coordinate), then interpolates
        const double xc = x < 0.0 ? 0.0 : (x > (double)(n_in - 1) ? (double)(n_in - 1) : x);
        int j = (int)floor(xc);
        if (j > n_in - 2) j = n_in - 2 < 0 ? 0 : n_in - 2;
        const double f = xc - (double)j;
        const float a = in[(o * n_in + j) * inner + i];
        const float b = in[(o * n_in + (j + 1 < n_in ? j + 1 : j)) * inner + i];
        out[idx] = (float)((1.0 - f) * (double)a + f * (double)b);
    } else {
        const int m = n_in + 2 * RS_PAD;
        double xp = x + (double)RS_PAD;
        xp = xp < 0.0 ? 0.0 : (xp > (double)(m - 1) ? (double)(m - 1) : xp);
        const int j0 = (int)floor(xp) - 1;
        const double f = xp - floor(xp);
        // cubic B-spline weights of the four taps j0 .. j0 + 3 (distance 1 + f, f, 1 - f, 2 - f)
        const double w0 = (1.0 - f) * (1.0 - f) * (1.0 - f) / 6.0;
        const double w1 = (4.0 - 6.0 * f * f + 3.0 * f * f * f) / 6.0;
        const double w2 = (1.0 + 3.0 * f + 3.0 * f * f - 3.0 * f * f * f) / 6.0;
        const double w3 = f * f * f / 6.0;
        const float *line = in + o * m * inner + i;
        auto c = [&](int j) { j = j < 0 ? 0 : (j >= m ? m - 1 : j); return (double)line[(long)j * inner]; };
        out[idx] = (float)(w0 * c(j0) + w1 * c(j0 + 1) + w2 * c(j0 + 2) + w3 * c(j0 + 3));
    }
}

// per group g: (min, max) of ref[g][0 .. n); two-stage: per-block partials through atomics on an order-preserving integer image
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
__global__ void minmax_init_kernel(unsigned *mm, long groups) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g < groups) { mm[2 * g] = 0xffffffffu; mm[2 * g + 1] = 0u; }
}
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ ref, long n, unsigned *mm) {
    const long g = blockIdx.y;
    const float *p = ref + g * n;
    unsigned lo = 0xffffffffu, hi = 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned v = f2ord(p[i]);
        lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    }
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned l2 = __shfl_xor(lo, s), h2 = __shfl_xor(hi, s);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(mm + 2 * g, lo); atomicMax(mm + 2 * g + 1, hi); }
}
__global__ __launch_bounds__(256) void clip_kernel(float *__restrict__ x, long n, const unsigned *__restrict__ mm) {
    const long g = blockIdx.y;
    const float lo = ord2f(mm[2 * g]), hi = ord2f(mm[2 * g + 1]);
    float *p = x + g * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = p[i];
        p[i] = v < lo ? lo : (v > hi ? hi : v);
    }
}

__global__ __launch_bounds__(256) void threshold_ge_kernel(const float *__restrict__ x, float thr, uint8_t *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] >= thr ? 1 : 0;
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t *__restrict__ x, float *__restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = x[i] ? 1.0f : 0.0f;
}

}  // namespace mi355

using namespace mi355;

extern "C" int mi355_mask_to_float(const uint8_t *mask_dev, float *out_dev, int64_t n, void *stream) {
    MI355_REQUIRE(mask_dev && out_dev && n >= 1, "mask_to_float: bad argument");
    MI355_TRY(bind_device());
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask_dev, out_dev, (long)n);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_threshold_ge(const float *x_dev, float thr, uint8_t *out_dev, int64_t n, void *stream) {
    MI355_REQUIRE(x_dev && out_dev && n >= 1, "threshold_ge: bad argument");
    MI355_TRY(bind_device());
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(threshold_ge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_dev, thr, out_dev, (long)n);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_resize_axis(const float *in_dev, float *out_dev, int64_t outer, int n_in, int n_out, int64_t inner, int order,
                                 void *stream) {
    MI355_REQUIRE(in_dev && out_dev && outer >= 1 && inner >= 1 && n_in >= 1 && n_out >= 1, "resize_axis: bad argument");
    MI355_REQUIRE(order == 0 || order == 1 || order == 3, "resize_axis: interpolation order %d (0, 1 and 3 are what the path uses)", order);
    MI355_REQUIRE(outer * (int64_t)(n_in > n_out ? n_in : n_out) * inner < ((int64_t)1 << 40), "resize_axis: tensor too large");
    MI355_TRY(bind_device());
    hipStream_t s = (hipStream_t)stream;
    const double scale = (double)n_in / (double)n_out;
    const int64_t total = outer * n_out * inner;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (order == 0) hipLaunchKernelGGL(resize_axis_kernel<0>, dim3(blocks), dim3(256), 0, s, in_dev, out_dev, (long)outer, n_in, n_out, (long)inner, scale);
    else if (order == 1) hipLaunchKernelGGL(resize_axis_kernel<1>, dim3(blocks), dim3(256), 0, s, in_dev, out_dev, (long)outer, n_in, n_out, (long)inner, scale);
    else {
        float *coef = nullptr;
        MI355_TRY(device_scratch(SCR_RESAMPLE, s, (size_t)outer * (n_in + 2 * RS_PAD) * inner * sizeof(float), (void **)&coef));
        const int64_t lines = outer * inner;
        hipLaunchKernelGGL(spline3_prefilter_kernel, dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, s, in_dev, coef, (long)outer, n_in, (long)inner);
        hipLaunchKernelGGL(resize_axis_kernel<3>, dim3(blocks), dim3(256), 0, s, (const float *)coef, out_dev, (long)outer, n_in, n_out, (long)inner, scale);
    }
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_clip_to_range_of(float *x_dev, int64_t groups, int64_t n_per_group, const float *ref_dev, int64_t ref_per_group,
                                      void *stream) {
    MI355_REQUIRE(x_dev && ref_dev && groups >= 1 && groups < 65536 && n_per_group >= 1 && ref_per_group >= 1, "clip_to_range_of: bad argument");
    MI355_TRY(bind_device());
    hipStream_t s = (hipStream_t)stream;
    unsigned *mm = nullptr;
    MI355_TRY(device_scratch(SCR_RESAMPLE_MM, s, (size_t)65536 * 2 * sizeof(unsigned), (void **)&mm));
    hipLaunchKernelGGL(minmax_init_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, mm, (long)groups);
    unsigned bx = (unsigned)((ref_per_group + 256 * 8 - 1) / (256 * 8));
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(minmax_kernel, dim3(bx, (unsigned)groups), dim3(256), 0, s, ref_dev, (long)ref_per_group, mm);
    unsigned cx = (unsigned)((n_per_group + 256 * 8 - 1) / (256 * 8));
    if (cx > 1024) cx = 1024;
    hipLaunchKernelGGL(clip_kernel, dim3(cx, (unsigned)groups), dim3(256), 0, s, x_dev, (long)n_per_group, (const unsigned *)mm);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}
