// HBM-bound kernels around the convolutions: norm finalize/apply, tile gather with mirror
// flips, the 1x1x1 segmentation head fused with sigmoid/softmax + mirror flip-back +
// Gaussian-weighted scatter-add, probability finish, region thresholding, label ensemble and
// the masked z-score.  All accesses are 16 B per lane where the layout allows it.
#include "kernels.h"

#include <algorithm>
#include <vector>

namespace mi355 {

// 4 consecutive elements of T (float | _Float16) <-> f32x4
template <typename T> __device__ __forceinline__ f32x4 load4(const T *p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float *p) { return *(const f32x4 *)p; }
template <> __device__ __forceinline__ f32x4 load4<_Float16>(const _Float16 *p) {
    const f16x4 h = *(const f16x4 *)p;
    f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
}
template <typename T> __device__ __forceinline__ void store4(T *p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float *p, f32x4 v) { *(f32x4 *)p = v; }
template <> __device__ __forceinline__ void store4<_Float16>(_Float16 *p, f32x4 v) {
    f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *(f16x4 *)p = h;
}

// ------------------------------------------------------------------ Instance / Group norm
// Finalises the (sum, sum^2) the conv epilogue accumulated into per-(n, channel) affine
// coefficients: y = x*scale + shift.  Replaces the statistics half of
// nn.InstanceNorm3d / nn.GroupNorm (reference generic_UNet.py:62-65,72); biased variance, eps
// inside the sqrt, as torch does.
__global__ void norm_finalize_kernel(const double *stats, int N, int C, double count, int kind, int groups,
                                     float eps, const float *gamma, const float *beta, float *scale,
                                     float *shift) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C)
        return;
    const int n = idx / C, c = idx - n * C;
    double s1, s2, cnt;
    if (kind == MI355_NORM_GROUP) {
        const int cpg = C / groups, g0 = (c / cpg) * cpg;
        s1 = 0.0; s2 = 0.0;
        for (int k = 0; k < cpg; ++k) {
            s1 += stats[((size_t)n * C + g0 + k) * 2 + 0];
            s2 += stats[((size_t)n * C + g0 + k) * 2 + 1];
        }
        cnt = count * cpg;
    } else {
        s1 = stats[(size_t)idx * 2 + 0];
        s2 = stats[(size_t)idx * 2 + 1];
        cnt = count;
    }
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double g = gamma ? (double)gamma[c] : 1.0, b = beta ? (double)beta[c] : 0.0;
    scale[idx] = (float)(g * rstd);
    shift[idx] = (float)(b - mean * rstd * g);
}

int norm_finalize(const double *stats, int N, int C, int64_t count, int kind, int groups, float eps,
                  const float *gamma, const float *beta, float *scale, float *shift, hipStream_t s) {
    MI355_REQUIRE(kind != MI355_NORM_GROUP || (groups > 0 && C % groups == 0), "GroupNorm: %d channels, %d groups", C, groups);
    const int total = N * C;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, stats, N, C,
                       (double)count, kind, groups, eps, gamma, beta, scale, shift);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

template <typename T>
__global__ void norm_apply_kernel(T *x, int64_t total4, int64_t VC4, int C4, const f32x4 *scale, const f32x4 *shift,
                                  int act, float slope) {
    // four quads per thread and trip, loads first (in place: the compiler cannot hoist a load above the previous store itself)
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < total4) v[u] = load4<T>(x + i * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= total4) continue;
            const int64_t n = i / VC4;
            const int c4 = (int)(i % C4);
            const f32x4 sc = scale[n * C4 + c4], sh = shift[n * C4 + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float y = v[u][k] * sc[k] + sh[k];
                if (act == ACT_LRELU)
                    y = y > 0.f ? y : y * slope;
                v[u][k] = y;
            }
            store4<T>(x + i * 4, v[u]);
        }
    }
}

// fp16 tensors are channel-blocked ([N][C / 8][V][8], common.h): one thread = the 8 channels of a voxel in one block,
// consecutive threads = consecutive voxels; blockIdx.y = (sample, block): scale / shift are wave-uniform.
__global__ void norm_apply_b8_kernel(_Float16 *x, int64_t V, int C, const float *scale, const float *shift, int act, float slope) {
    const int64_t nb = blockIdx.y;  // n * (C / 8) + block
    const int CB = C >> 3;
    const int64_t n = nb / CB;
    const int cb = (int)(nb - n * CB);
    const float *sc = scale + n * C + cb * 8, *sh = shift + n * C + cb * 8;
    float s8[8], h8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s8[k] = sc[k]; h8[k] = sh[k]; }
    _Float16 *base = x + nb * V * 8;
    // Four pieces per thread and trip, all four loads issued before the first store (round 3): the pass works in place, so the
    // compiler cannot move a load above the previous trip's store by itself, and one 16-B load in flight per thread left the
    // memory pipe waiting on occupancy alone.
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < V; v0 += U * stride) {
        f16x8 h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            if (v < V) h[u] = *(const f16x8 *)(base + v * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t v = v0 + u * stride;
            if (v >= V) continue;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float y = (float)h[u][k] * s8[k] + h8[k];
                if (act == ACT_LRELU) y = y > 0.f ? y : y * slope;
                h[u][k] = (_Float16)y;
            }
            *(f16x8 *)(base + v * 8) = h[u];
        }
    }
}

int norm_apply(void *x, int dtype, int N, int64_t V, int C, const float *scale, const float *shift, int act,
               float slope, hipStream_t s) {
    MI355_REQUIRE(C % 4 == 0, "norm_apply: C=%d", C);
    const int64_t total4 = (int64_t)N * V * C / 4;
    int64_t blocks = (total4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (dtype == MI355_F16) {
        MI355_REQUIRE(C % 8 == 0 && (int64_t)N * (C / 8) <= 65535, "norm_apply: fp16 tensors are blocked by 8 channels (N = %d, C = %d)", N, C);
        int64_t bx = (V + 255) / 256;
        const int64_t cap = std::max<int64_t>(1, 4096 / ((int64_t)N * (C / 8)));
        if (bx > cap) bx = cap;
        hipLaunchKernelGGL(norm_apply_b8_kernel, dim3((unsigned)bx, (unsigned)(N * (C / 8))), dim3(256), 0, s, (_Float16 *)x, V, C, scale, shift, act, slope);
    } else
        hipLaunchKernelGGL(norm_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (float *)x, total4,
                           V * C / 4, C / 4, (const f32x4 *)scale, (const f32x4 *)shift, act, slope);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// ------------------------------------------------------------------ tile gather (+ mirror flips)
// Replaces the patch slicing of nnU-Net v1 `_internal_predict_3D_3Dconv_tiled` and the
// torch.flip of `_internal_maybe_mirror_and_pred_3D`: sample b is the
// tile at tiles[b].{z0,y0,x0} of the zero-padded volume, flipped along the axes in its mirror mask.
constexpr int MAX_SAMPLES = 64;
struct TileList {
    TileDesc t[MAX_SAMPLES];
};

template <typename T>
__global__ void extract_tiles_kernel(const float *vol, int C, int Z, int Y, int X, int padz, int pady,
                                     int padx, TileList tl, int P0, int P1, int P2, int Cpad, T *x) {
    const int b = blockIdx.y;
    const TileDesc td = tl.t[b];
    const int64_t PV = (int64_t)P0 * P1 * P2;
    const int64_t ZYX = (int64_t)Z * Y * X;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < PV; v += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(v % P2);
        const int py = (int)((v / P2) % P1);
        const int pz = (int)(v / ((int64_t)P2 * P1));
        const int sz = (td.mirror & 1) ? P0 - 1 - pz : pz;
        const int sy = (td.mirror & 2) ? P1 - 1 - py : py;
        const int sx = (td.mirror & 4) ? P2 - 1 - px : px;
        const int gz = td.z0 + sz - padz, gy = td.y0 + sy - pady, gx = td.x0 + sx - padx;
        const bool ok = (unsigned)gz < (unsigned)Z && (unsigned)gy < (unsigned)Y && (unsigned)gx < (unsigned)X;
        const bool blocked = std::is_same<T, _Float16>::value && (Cpad % 8 == 0);  // (not the stem's NDHW4 input)
        T *dst = x + ((int64_t)b * PV + v) * Cpad;
        const int64_t g = ((int64_t)gz * Y + gy) * X + gx;
        for (int c = 0; c < Cpad; ++c) {
            const T val = (T)((ok && c < C) ? vol[c * ZYX + g] : 0.f);
            if (blocked) x[b8_index(b, c, v, Cpad, PV)] = val;
            else dst[c] = val;
        }
    }
}

int extract_tiles(const float *vol, int C, int Z, int Y, int X, int padz, int pady, int padx,
                  const TileDesc *tiles_host, int n_samples, int P0, int P1, int P2, int Cpad, void *x, int dtype,
                  hipStream_t s) {
    MI355_REQUIRE(n_samples > 0 && n_samples <= MAX_SAMPLES, "extract_tiles: %d samples (max %d)", n_samples, MAX_SAMPLES);
    TileList tl;
    for (int i = 0; i < n_samples; ++i) tl.t[i] = tiles_host[i];
    const int64_t PV = (int64_t)P0 * P1 * P2;
    int64_t bx = (PV + 255) / 256;
    if (bx > 4096) bx = 4096;
    if (dtype == MI355_F16)
        hipLaunchKernelGGL(extract_tiles_kernel<_Float16>, dim3((unsigned)bx, n_samples), dim3(256), 0, s, vol, C, Z, Y,
                           X, padz, pady, padx, tl, P0, P1, P2, Cpad, (_Float16 *)x);
    else
        hipLaunchKernelGGL(extract_tiles_kernel<float>, dim3((unsigned)bx, n_samples), dim3(256), 0, s, vol, C, Z, Y, X,
                           padz, pady, padx, tl, P0, P1, P2, Cpad, (float *)x);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// (fp16 with Cpad % 8 == 0, i.e. a network without the stem kernel: channel-blocked output, common.h; the stem's NDHW4 input
// is a plain 4-channel tensor in both dtypes)
template <typename T>
__global__ void nchw_to_ndhwc_kernel(const float *x, int C, int64_t V, int Cpad, T *y, int64_t total) {
    const bool blocked = std::is_same<T, _Float16>::value && (Cpad % 8 == 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / V, v = i - n * V;
        for (int c = 0; c < Cpad; ++c)
            y[blocked ? b8_index(n, c, v, Cpad, V) : (size_t)(i * Cpad + c)] = (T)(c < C ? x[(n * C + c) * V + v] : 0.f);
    }
}

int nchw_to_ndhwc(const float *x, int N, int C, int64_t V, int Cpad, void *y, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)N * V;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == MI355_F16)
        hipLaunchKernelGGL(nchw_to_ndhwc_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, x, C, V, Cpad, (_Float16 *)y, total);
    else
        hipLaunchKernelGGL(nchw_to_ndhwc_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, C, V, Cpad, (float *)y, total);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// plain NDHWC <-> channel-blocked fp16 (common.h): the single-op entry points of the C ABI take and return plain NDHWC
// tensors (include/mi355_nnunet.h), the kernels work on blocked ones
__global__ void ndhwc_to_b8_kernel(const _Float16 *x, int C, int64_t V, int64_t total8, _Float16 *y) {
    const int CB = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        // i = (n * CB + cb) * V + v : one 16-B piece of the blocked tensor
        const int64_t v = i % V, nb = i / V;
        const int64_t n = nb / CB;
        const int cb = (int)(nb - n * CB);
        *(f16x8 *)(y + i * 8) = *(const f16x8 *)(x + (n * V + v) * C + cb * 8);
    }
}
__global__ void b8_to_ndhwc_kernel(const _Float16 *x, int C, int64_t V, int64_t total8, _Float16 *y) {
    const int CB = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = i % V, nb = i / V;
        const int64_t n = nb / CB;
        const int cb = (int)(nb - n * CB);
        *(f16x8 *)(y + (n * V + v) * C + cb * 8) = *(const f16x8 *)(x + i * 8);
    }
}
int ndhwc_to_b8(const _Float16 *x, int N, int C, int64_t V, _Float16 *y, hipStream_t s) {
    MI355_REQUIRE(C % 8 == 0, "blocked fp16 layout needs C %% 8 == 0 (got %d)", C);
    const int64_t total8 = (int64_t)N * V * (C / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(ndhwc_to_b8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, C, V, total8, y);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}
int b8_to_ndhwc(const _Float16 *x, int N, int C, int64_t V, _Float16 *y, hipStream_t s) {
    MI355_REQUIRE(C % 8 == 0, "blocked fp16 layout needs C %% 8 == 0 (got %d)", C);
    const int64_t total8 = (int64_t)N * V * (C / 8);
    int64_t blocks = (total8 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(b8_to_ndhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, C, V, total8, y);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// ------------------------------------------------------------------ segmentation head
int head_weights_upload(const float *w_host, const float *b_host, int cin, int ncls, HeadWeights *out) {
    MI355_REQUIRE(ncls >= 1 && ncls <= 8, "head: %d classes unsupported (max 8)", ncls);
    MI355_REQUIRE(cin % 8 == 0, "head: cin %d not a multiple of 8", cin);
    HeadWeights h;
    h.cin = cin; h.ncls = ncls;
    MI355_HIP(hipMalloc(&h.w_dev, (size_t)ncls * cin * sizeof(float)));
    MI355_HIP(hipMemcpy(h.w_dev, w_host, (size_t)ncls * cin * sizeof(float), hipMemcpyHostToDevice));
    MI355_HIP(hipMalloc(&h.b_dev, 8 * sizeof(float)));
    MI355_HIP(hipMemset(h.b_dev, 0, 8 * sizeof(float)));
    if (b_host)
        MI355_HIP(hipMemcpy(h.b_dev, b_host, ncls * sizeof(float), hipMemcpyHostToDevice));
    *out = h;
    return MI355_OK;
}

void head_weights_free(HeadWeights *w) {
    if (w->w_dev) (void)hipFree(w->w_dev);
    if (w->b_dev) (void)hipFree(w->b_dev);
    *w = HeadWeights();
}

constexpr int HEAD_MAX_CLS = 8;

// One thread per voxel: the whole channel vector of the voxel is read with 16-B loads (every byte of every
// line is used, consecutive lanes = consecutive voxels), the ncls x C head weights are wave-uniform (scalar loads),
// no cross-lane traffic.  The aggregate / normaliser read-modify-writes are then fully coalesced along x.
// NORM: the feature map is the RAW output of the last decoder conv and its Instance/GroupNorm (+ LeakyReLU) is applied here,
// per channel with this sample's scale / shift (wave-uniform: scalar loads) - generic_UNet.py:62-72 is one expression,
// lrelu(instnorm(conv(x))), and this kernel reads every feature exactly once anyway (round 3: removes the norm_apply pass
// over the widest-resolution tensor of the decoder).
// `feat` = this sample's feature map, `v` = the voxel, V = voxels per sample.  fp32 tensors are plain NDHWC ([V][C]), fp16
// tensors channel-blocked ([C / 8][V][8], common.h): either way consecutive lanes (voxels) read consecutive 16-B pieces.
// CC > 0: the channel count is a compile-time constant (32: every head of the two BraTS networks) - the piece loop is unrolled and ALL
// 16-B pieces of the voxel are loaded before the first is used (round 3: as a run-time loop each piece was load -> wait -> use,
// one load in flight per thread, and head_aggregate ran at 2.0 TB/s with its lanes idle on memory latency).  Same order of
// additions either way: bit-identical logits.
template <typename T, bool NORM, int CC>
__device__ __forceinline__ void head_dot_impl(const T *feat, int64_t v, int64_t V, const float *__restrict__ w, const float *__restrict__ b,
                                              int Crt, int ncls, float *logit, const float *__restrict__ sc, const float *__restrict__ sh, float nslope) {
    const int C = CC > 0 ? CC : Crt;
#pragma unroll
    for (int k = 0; k < HEAD_MAX_CLS; ++k) logit[k] = (k < ncls) ? b[k] : 0.f;
    constexpr bool F16 = std::is_same<T, _Float16>::value;
    constexpr int CW = F16 ? 8 : 4;  // channels per 16-B piece
    constexpr int NP = CC > 0 ? CC / CW : 1;
    typedef typename std::conditional<F16, f16x8, f32x4>::type piece_t;
    piece_t pre[NP];
    if constexpr (CC > 0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if constexpr (F16) pre[i] = *(const f16x8 *)(feat + ((int64_t)i * V + v) * 8);
            else pre[i] = *(const f32x4 *)(feat + v * C + i * 4);
        }
    }
#pragma unroll
    for (int c = 0; c < C; c += CW) {
        float f[CW];
        piece_t h;
        if constexpr (CC > 0) h = pre[c / CW];
        else if constexpr (F16) h = *(const f16x8 *)(feat + ((int64_t)(c >> 3) * V + v) * 8);
        else h = *(const f32x4 *)(feat + v * C + c);
#pragma unroll
        for (int j = 0; j < CW; ++j) f[j] = (float)h[j];
        if constexpr (NORM) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
                const float y = fmaf(f[j], sc[c + j], sh[c + j]);
                f[j] = fmaxf(y, y * nslope);  // nslope = 1: no activation (max(y, y) = y)
            }
        }
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k)
            if (k < ncls) {
                const float *wk = w + k * C + c;
                // (same association as round 2: channels c+3, c+2, c+1, c innermost to outermost, 4 at a time)
#pragma unroll
                for (int q4 = 0; q4 < CW; q4 += 4)
                    logit[k] = fmaf(f[q4], wk[q4], fmaf(f[q4 + 1], wk[q4 + 1], fmaf(f[q4 + 2], wk[q4 + 2], fmaf(f[q4 + 3], wk[q4 + 3], logit[k]))));
            }
    }
}
template <typename T, bool NORM>
__device__ __forceinline__ void head_dot(const T *feat, int64_t v, int64_t V, const float *__restrict__ w, const float *__restrict__ b,
                                         int C, int ncls, float *logit, const float *__restrict__ sc = nullptr,
                                         const float *__restrict__ sh = nullptr, float nslope = 1.0f) {
    if (C == 32) head_dot_impl<T, NORM, 32>(feat, v, V, w, b, C, ncls, logit, sc, sh, nslope);  // (wave-uniform)
    else head_dot_impl<T, NORM, 0>(feat, v, V, w, b, C, ncls, logit, sc, sh, nslope);
}

template <typename T, bool NORM>
__global__ void head_logits_kernel(const T *feat, const float *w, const float *b, int C, int ncls,
                                   int64_t V, float *logits, FeatNorm fn) {
    // (blockIdx.y = sample: the per-sample scale / shift rows stay wave-uniform)
    const int64_t n = blockIdx.y;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
        float lg[HEAD_MAX_CLS];
        head_dot<T, NORM>(feat + n * V * C, v, V, w, b, C, ncls, lg, NORM ? fn.scale + n * C : nullptr, NORM ? fn.shift + n * C : nullptr, fn.slope);
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k)
            if (k < ncls) logits[(n * ncls + k) * V + v] = lg[k];
    }
}

template <typename T>
static void launch_head_logits(const HeadWeights &w, const T *feat, int N, int64_t V, float *logits, const FeatNorm &fn, hipStream_t s) {
    int64_t blocks = (V + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (fn.scale)
        hipLaunchKernelGGL((head_logits_kernel<T, true>), dim3((unsigned)blocks, N), dim3(256), 0, s, feat, w.w_dev, w.b_dev, w.cin, w.ncls, V, logits, fn);
    else
        hipLaunchKernelGGL((head_logits_kernel<T, false>), dim3((unsigned)blocks, N), dim3(256), 0, s, feat, w.w_dev, w.b_dev, w.cin, w.ncls, V, logits, fn);
}

int head_logits(const HeadWeights &w, const void *feat, int dtype, int N, int64_t V, float *logits, hipStream_t s, const FeatNorm &fn) {
    MI355_REQUIRE(N >= 1 && N <= 65535, "head_logits: %d samples", N);
    if (dtype == MI355_F16) launch_head_logits<_Float16>(w, (const _Float16 *)feat, N, V, logits, fn, s);
    else launch_head_logits<float>(w, (const float *)feat, N, V, logits, fn, s);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

struct MirrorList {
    int n;
    int m[8];
};

// Replaces for ONE tile:
//   result = sum_m  mult * flip_back(nonlin(net(flip_m(x))))      mult = 1/n_mirrors, m in list order
//   result *= gaussian ; aggregated[:, tile] += result ; normaliser[tile] += gaussian
// feat holds the last decoder feature map of the n_mirrors forwards of this tile.
// NM: compile-time mirror count (8 = the reference's full TTA; 0 = run time).  With the loop unrolled the feature loads of all
// mirrors are independent of the sigmoid / softmax arithmetic between them and go out together: this kernel is a stream
// of 16-B loads (32 per voxel for 32 channels x 8 mirrors), and as a run-time loop it had one mirror's four in flight.
template <typename T, bool NORM, int NM = 0>
__global__ void head_aggregate_kernel(const T *feat, const float *w, const float *b, int C, int ncls,
                                      MirrorList ml, int P0, int P1, int P2, int nonlin, const float *gauss,
                                      float *agg, float *cnt, int Zp, int Yp, int Xp, int z0, int y0, int x0, FeatNorm fn) {
    const int64_t PV = (int64_t)P0 * P1 * P2;
    const int64_t ZYXp = (int64_t)Zp * Yp * Xp;
    const float mult = 1.0f / (float)ml.n;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < PV; v += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(v % P2);
        const int py = (int)((v / P2) % P1);
        const int pz = (int)(v / ((int64_t)P2 * P1));
        float res[HEAD_MAX_CLS];
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k) res[k] = 0.f;
        const int nm = NM ? NM : ml.n;
#pragma unroll
        for (int mi = 0; mi < (NM ? NM : 8); ++mi) {
            if (!NM && mi >= nm) break;
            const int m = ml.m[mi];
            const int sz = (m & 1) ? P0 - 1 - pz : pz;
            const int sy = (m & 2) ? P1 - 1 - py : py;
            const int sx = (m & 4) ? P2 - 1 - px : px;
            const int64_t sv = ((int64_t)sz * P1 + sy) * P2 + sx;
            float lg[HEAD_MAX_CLS];
            // (fn.scale / fn.shift already point at this tile's first sample)
            head_dot<T, NORM>(feat + (int64_t)mi * PV * C, sv, PV, w, b, C, ncls, lg, NORM ? fn.scale + mi * C : nullptr,
                              NORM ? fn.shift + mi * C : nullptr, fn.slope);
            if (nonlin == MI355_NONLIN_SIGMOID) {
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) lg[k] = 1.0f / (1.0f + expf(-lg[k]));
            } else if (nonlin == MI355_NONLIN_SOFTMAX) {
                float mx = lg[0];
#pragma unroll
                for (int k = 1; k < HEAD_MAX_CLS; ++k) if (k < ncls) mx = fmaxf(mx, lg[k]);
                float den = 0.f;
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) { lg[k] = expf(lg[k] - mx); den += lg[k]; }
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) lg[k] = lg[k] / den;
            }
#pragma unroll
            for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) res[k] += mult * lg[k];
        }
        const float g = gauss ? gauss[v] : 1.0f;
        const int64_t gi = ((int64_t)(z0 + pz) * Yp + (y0 + py)) * Xp + (x0 + px);
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k)
            if (k < ncls) agg[k * ZYXp + gi] += res[k] * g;
        if (cnt) cnt[gi] += g;
    }
}

template <typename T>
static void launch_head_aggregate(const HeadWeights &w, const T *feat, const MirrorList &ml, int P0, int P1, int P2, int nonlin,
                                  const float *gauss, float *agg, float *cnt, int Zp, int Yp, int Xp, int z0, int y0, int x0,
                                  const FeatNorm &fn, hipStream_t s) {
    const int64_t PV = (int64_t)P0 * P1 * P2;
    int64_t blocks = (PV + 255) / 256;
    if (blocks > 16384) blocks = 16384;
#define MI355_HA_LAUNCH(NORM_, NM_) hipLaunchKernelGGL((head_aggregate_kernel<T, NORM_, NM_>), dim3((unsigned)blocks), dim3(256), 0, s, feat, w.w_dev, w.b_dev, \
                                                      w.cin, w.ncls, ml, P0, P1, P2, nonlin, gauss, agg, cnt, Zp, Yp, Xp, z0, y0, x0, fn)
    if (fn.scale) { if (ml.n == 8) MI355_HA_LAUNCH(true, 8); else MI355_HA_LAUNCH(true, 0); }
    else { if (ml.n == 8) MI355_HA_LAUNCH(false, 8); else MI355_HA_LAUNCH(false, 0); }
#undef MI355_HA_LAUNCH
}

int head_aggregate(const HeadWeights &w, const void *feat, int dtype, int first_sample, const int *mirrors_host,
                   int n_mirrors, int P0, int P1, int P2, int nonlin, const float *gauss, float *agg,
                   float *cnt, int Zp, int Yp, int Xp, int z0, int y0, int x0, hipStream_t s, const FeatNorm &fn_all) {
    MI355_REQUIRE(n_mirrors >= 1 && n_mirrors <= 8, "head_aggregate: %d mirrors", n_mirrors);
    MirrorList ml;
    ml.n = n_mirrors;
    for (int i = 0; i < 8; ++i) ml.m[i] = i < n_mirrors ? mirrors_host[i] : 0;
    const int64_t PV = (int64_t)P0 * P1 * P2;
    FeatNorm fn = fn_all;  // rows of this tile's first sample
    if (fn.scale) { fn.scale += (size_t)first_sample * w.cin; fn.shift += (size_t)first_sample * w.cin; }
    if (dtype == MI355_F16)
        launch_head_aggregate<_Float16>(w, (const _Float16 *)feat + (size_t)first_sample * PV * w.cin, ml, P0, P1, P2, nonlin, gauss, agg,
                                        cnt, Zp, Yp, Xp, z0, y0, x0, fn, s);
    else
        launch_head_aggregate<float>(w, (const float *)feat + (size_t)first_sample * PV * w.cin, ml, P0, P1, P2, nonlin, gauss, agg, cnt,
                                     Zp, Yp, Xp, z0, y0, x0, fn, s);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// Same as head_aggregate_kernel when the last conv already applied the head (logits [n_mirrors][ncls][PV] of this tile).
__global__ void logits_aggregate_kernel(const float *logits, int ncls, MirrorList ml, int P0, int P1, int P2, int nonlin,
                                        const float *gauss, float *agg, float *cnt, int Zp, int Yp, int Xp, int z0, int y0,
                                        int x0) {
    const int64_t PV = (int64_t)P0 * P1 * P2;
    const int64_t ZYXp = (int64_t)Zp * Yp * Xp;
    const float mult = 1.0f / (float)ml.n;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < PV; v += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(v % P2);
        const int py = (int)((v / P2) % P1);
        const int pz = (int)(v / ((int64_t)P2 * P1));
        float res[HEAD_MAX_CLS];
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k) res[k] = 0.f;
        for (int mi = 0; mi < ml.n; ++mi) {
            const int m = ml.m[mi];
            const int sz = (m & 1) ? P0 - 1 - pz : pz;
            const int sy = (m & 2) ? P1 - 1 - py : py;
            const int sx = (m & 4) ? P2 - 1 - px : px;
            const int64_t sv = ((int64_t)sz * P1 + sy) * P2 + sx;
            float lg[HEAD_MAX_CLS];
#pragma unroll
            for (int k = 0; k < HEAD_MAX_CLS; ++k) lg[k] = (k < ncls) ? logits[((int64_t)mi * ncls + k) * PV + sv] : 0.f;
            if (nonlin == MI355_NONLIN_SIGMOID) {
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) lg[k] = 1.0f / (1.0f + expf(-lg[k]));
            } else if (nonlin == MI355_NONLIN_SOFTMAX) {
                float mx = lg[0];
#pragma unroll
                for (int k = 1; k < HEAD_MAX_CLS; ++k) if (k < ncls) mx = fmaxf(mx, lg[k]);
                float den = 0.f;
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) { lg[k] = expf(lg[k] - mx); den += lg[k]; }
#pragma unroll
                for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) lg[k] = lg[k] / den;
            }
#pragma unroll
            for (int k = 0; k < HEAD_MAX_CLS; ++k) if (k < ncls) res[k] += mult * lg[k];
        }
        const float g = gauss ? gauss[v] : 1.0f;
        const int64_t gi = ((int64_t)(z0 + pz) * Yp + (y0 + py)) * Xp + (x0 + px);
#pragma unroll
        for (int k = 0; k < HEAD_MAX_CLS; ++k)
            if (k < ncls) agg[k * ZYXp + gi] += res[k] * g;
        if (cnt) cnt[gi] += g;
    }
}

int logits_aggregate(const float *logits, int ncls, int first_sample, const int *mirrors_host, int n_mirrors, int P0, int P1,
                     int P2, int nonlin, const float *gauss, float *agg, float *cnt, int Zp, int Yp, int Xp, int z0, int y0,
                     int x0, hipStream_t s) {
    MI355_REQUIRE(n_mirrors >= 1 && n_mirrors <= 8 && ncls >= 1 && ncls <= HEAD_MAX_CLS, "logits_aggregate: bad arguments");
    MirrorList ml;
    ml.n = n_mirrors;
    for (int i = 0; i < 8; ++i) ml.m[i] = i < n_mirrors ? mirrors_host[i] : 0;
    const int64_t PV = (int64_t)P0 * P1 * P2;
    int64_t blocks = (PV + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(logits_aggregate_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                       logits + (size_t)first_sample * ncls * PV, ncls, ml, P0, P1, P2, nonlin, gauss, agg, cnt, Zp, Yp, Xp, z0,
                       y0, x0);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// class_probabilities = aggregated / normaliser, cropped back from the padded grid.
__global__ void finish_probs_kernel(const float *agg, const float *cnt, int C, int Z, int Y, int X, int Zp,
                                    int Yp, int Xp, int pz, int py, int px, float *probs, int accumulate) {
    const int64_t ZYX = (int64_t)Z * Y * X, ZYXp = (int64_t)Zp * Yp * Xp;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < ZYX; v += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(v % X);
        const int y = (int)((v / X) % Y);
        const int z = (int)(v / ((int64_t)X * Y));
        const int64_t gi = ((int64_t)(z + pz) * Yp + (y + py)) * Xp + (x + px);
        const float c = cnt[gi];
        for (int k = 0; k < C; ++k) {
            const float p = agg[k * ZYXp + gi] / c;
            if (accumulate)
                probs[k * ZYX + v] += p;
            else
                probs[k * ZYX + v] = p;
        }
    }
}

int finish_probs(const float *agg, const float *cnt, int C, int Z, int Y, int X, int Zp, int Yp, int Xp,
                 int pz, int py, int px, float *probs, int accumulate, hipStream_t s) {
    const int64_t ZYX = (int64_t)Z * Y * X;
    int64_t blocks = (ZYX + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(finish_probs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, agg, cnt, C, Z, Y, X, Zp, Yp,
                       Xp, pz, py, px, probs, accumulate);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

__global__ void cnt_add_tile_kernel(const float *gauss, int P0, int P1, int P2, float *cnt, int Yp, int Xp, int z0,
                                    int y0, int x0) {
    const int64_t PV = (int64_t)P0 * P1 * P2;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < PV; v += (int64_t)gridDim.x * blockDim.x) {
        const int px = (int)(v % P2);
        const int py = (int)((v / P2) % P1);
        const int pz = (int)(v / ((int64_t)P2 * P1));
        cnt[((int64_t)(z0 + pz) * Yp + (y0 + py)) * Xp + (x0 + px)] += gauss ? gauss[v] : 1.0f;
    }
}

int cnt_add_tile(const float *gauss, int P0, int P1, int P2, float *cnt, int Yp, int Xp, int z0, int y0, int x0,
                 hipStream_t s) {
    const int64_t PV = (int64_t)P0 * P1 * P2;
    int64_t blocks = (PV + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cnt_add_tile_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gauss, P0, P1, P2, cnt, Yp, Xp,
                       z0, y0, x0);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

__global__ void scale_kernel(float *x, int64_t n, float divisor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = x[i] / divisor;
}

int scale_inplace(float *x, int64_t n, float divisor, hipStream_t s) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, divisor);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// ------------------------------------------------------------------ export / ensemble / preprocess
struct OrderList {
    int v[8];
    int argmax;
};

__global__ void regions_to_labels_kernel(const float *probs, int C, int Z, int Y, int X, OrderList order,
                                         int bz, int by, int bx, int FY, int FX, uint8_t *labels) {
    const int64_t ZYX = (int64_t)Z * Y * X;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < ZYX; v += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(v % X);
        const int y = (int)((v / X) % Y);
        const int z = (int)(v / ((int64_t)X * Y));
        int lab = 0;
        if (order.argmax) {  // non-region trainers: seg = probs.argmax(0), first maximum wins like numpy
            float best = probs[v];
            for (int k = 1; k < C; ++k) {
                const float p = probs[k * ZYX + v];
                if (p > best) { best = p; lab = k; }
            }
        } else {
            for (int k = 0; k < C; ++k)
                if (probs[k * ZYX + v] > 0.5f)
                    lab = order.v[k];
        }
        labels[((int64_t)(z + bz) * FY + (y + by)) * FX + (x + bx)] = (uint8_t)lab;
    }
}

__global__ void label_ensemble_kernel(const uint8_t *a, const uint8_t *b, uint8_t *out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // np.round((a+b)/2.0): halves round to the even neighbour
        const int s = (int)a[i] + (int)b[i];
        int r = s >> 1;
        if ((s & 1) && (r & 1))
            r += 1;
        out[i] = (uint8_t)r;
    }
}

__global__ void prob_mean_kernel(const float *a, const float *b, float *out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (a[i] + b[i]) / 2.0f;
}

// masked z-score: three launches per call (per-block partial sums, fixed-order finish, apply) over C channels.
// No atomics: every block writes its (sum, sum of squares, count) to its own slot and the finish kernel adds the slots
// in a fixed order, so the normalised volume is bit-reproducible run to run and identical on every rank.
__global__ void masked_sums_kernel(const float *vol, const uint8_t *mask, int64_t V, double *partial) {
    const int c = blockIdx.y;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x)
        if (mask[i]) {
            const double x = (double)vol[c * V + i];
            s1 += x; s2 += x * x; cnt += 1.0;
        }
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
        cnt += __shfl_down(cnt, off);
    }
    __shared__ double w[4][3];
    if ((threadIdx.x & 63) == 0) { w[threadIdx.x >> 6][0] = s1; w[threadIdx.x >> 6][1] = s2; w[threadIdx.x >> 6][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double *p = partial + ((size_t)c * gridDim.x + blockIdx.x) * 3;
        p[threadIdx.x] = ((w[0][threadIdx.x] + w[1][threadIdx.x]) + w[2][threadIdx.x]) + w[3][threadIdx.x];
    }
}

// sums[c][k] = partial[c][0][k] + partial[c][1][k] + ... : one 64-lane wave per (c, k), lane-strided then a butterfly
__global__ void masked_sums_finish_kernel(const double *partial, int nblocks, double *sums) {
    const int c = blockIdx.x, k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= 3) return;
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += partial[((size_t)c * nblocks + b) * 3 + k];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) sums[c * 3 + k] = a;
}

__global__ void masked_zscore_kernel(float *vol, const uint8_t *mask, int64_t V, const double *sums) {
    const int c = blockIdx.y;
    const double cnt = sums[c * 3 + 2];
    const double mean = cnt > 0 ? sums[c * 3 + 0] / cnt : 0.0;
    double var = cnt > 0 ? sums[c * 3 + 1] / cnt - mean * mean : 0.0;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean;
    const float denom = (float)sqrt(var) + 1e-8f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x)
        vol[c * V + i] = mask[i] ? (vol[c * V + i] - meanf) / denom : 0.f;
}

}  // namespace mi355

using namespace mi355;

extern "C" int mi355_regions_to_labels(const float *probs_dev, int C, int Z, int Y, int X, const int32_t *order,
                                       const int32_t bbox_lo[3], const int32_t full[3], uint8_t *labels_dev,
                                       void *stream) {
    MI355_REQUIRE(C >= 1 && C <= 8, "regions_to_labels: %d channels", C);
    MI355_TRY(bind_device());
    MI355_REQUIRE(bbox_lo[0] >= 0 && bbox_lo[1] >= 0 && bbox_lo[2] >= 0 && bbox_lo[0] + Z <= full[0] &&
                      bbox_lo[1] + Y <= full[1] && bbox_lo[2] + X <= full[2],
                  "regions_to_labels: crop box does not fit the full volume");
    hipStream_t s = (hipStream_t)stream;
    OrderList ol;
    for (int i = 0; i < 8; ++i) ol.v[i] = (order && i < C) ? order[i] : 0;
    ol.argmax = order == nullptr;
    MI355_HIP(hipMemsetAsync(labels_dev, 0, (size_t)full[0] * full[1] * full[2], s));
    const int64_t ZYX = (int64_t)Z * Y * X;
    int64_t blocks = (ZYX + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(regions_to_labels_kernel, dim3((unsigned)blocks), dim3(256), 0, s, probs_dev, C, Z, Y, X, ol,
                       bbox_lo[0], bbox_lo[1], bbox_lo[2], full[1], full[2], labels_dev);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_label_ensemble(const uint8_t *a_dev, const uint8_t *b_dev, uint8_t *out_dev, int64_t n,
                                    void *stream) {
    MI355_TRY(bind_device());
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(label_ensemble_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a_dev,
                       b_dev, out_dev, n);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_prob_mean(const float *a_dev, const float *b_dev, float *out_dev, int64_t n, void *stream) {
    MI355_TRY(bind_device());
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(prob_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a_dev, b_dev,
                       out_dev, n);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_zscore_masked(float *vol_dev, const uint8_t *mask_dev, int C, int64_t voxels, void *stream) {
    MI355_REQUIRE(C >= 1 && C <= 64, "zscore: %d channels", C);
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = (voxels + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    // persistent scratch, ONE slot of its own: [sums 64 x 3 | partial C x blocks x 3] doubles; asynchronous on `stream`.  The sums
    // used to sit in SCR_SMALL, which mi355_label_confusion / mi355_label_stats also use: a caller following INTEGRATION.md's
    // two-stream exception (preprocess the next case beside a prediction / evaluation) would have corrupted mean and std.
    double *sums = nullptr;
    MI355_TRY(device_scratch(SCR_ZSCORE, s, (size_t)(64 + 64 * 2048) * 3 * sizeof(double), (void **)&sums));
    double *partial = sums + 64 * 3;
    hipLaunchKernelGGL(masked_sums_kernel, dim3((unsigned)blocks, C), dim3(256), 0, s, vol_dev, mask_dev, voxels, partial);
    hipLaunchKernelGGL(masked_sums_finish_kernel, dim3(C), dim3(192), 0, s, partial, (int)blocks, sums);
    hipLaunchKernelGGL(masked_zscore_kernel, dim3((unsigned)blocks, C), dim3(256), 0, s, vol_dev, mask_dev, voxels, sums);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}
