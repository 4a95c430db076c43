// "Next" rows of SURVEY.md section 8f that are cheap once the label map is on the device:
//   * label convention remap          convert_labels_to_brats.py:34-55
//   * confusion counts for Dice/IoU   evaluate_segmentation.py:12-49,129-195
//   * cosine top-k retrieval          RAG_Assistant/rag_assistant.py:197-211 (DummyVectorStore.retrieve)
// All HBM-bound streaming kernels.
#include "kernels.h"

namespace mi355 {

struct ByteMap { unsigned char m[256]; };

__global__ void label_remap_kernel(const uint8_t *in, uint8_t *out, int64_t n, ByteMap map) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = map.m[in[i]];
}

// counts[p*K + g] += 1 for every voxel with pred label p, truth label g; bin K-1 is the "other" bin: it collects every
// label >= K-1, so callers that care about labels 0..L pass K = L + 2 and out-of-range labels never alias a real one
__global__ void confusion_kernel(const uint8_t *pred, const uint8_t *gt, int64_t n, int K, unsigned long long *counts) {
    __shared__ unsigned int local[64];
    if (threadIdx.x < 64) local[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int p = pred[i], g = gt[i];
        p = p < K ? p : K - 1;
        g = g < K ? g : K - 1;
        atomicAdd(&local[p * K + g], 1u);
    }
    __syncthreads();
    if (threadIdx.x < K * K && local[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)local[threadIdx.x]);
}

// scores[r] = dot(V[r, :], q): one wave per row, 16-B loads, wave reduction
__global__ void row_dot_kernel(const float *V, const float *q, int N, int D, float *scores) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < N; r += nwaves) {
        const float *row = V + (size_t)r * D;
        float acc = 0.f;
        if ((D & 3) == 0) {
            for (int c = lane * 4; c < D; c += 256) {
                const f32x4 a = *(const f32x4 *)(row + c), b = *(const f32x4 *)(q + c);
                acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            }
        } else {
            for (int c = lane; c < D; c += 64) acc += row[c] * q[c];
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) scores[r] = acc;
    }
}

// one pass of selection: best (score, index) not yet taken; ties -> larger index (np.argsort(scores)[::-1])
__global__ void argmax_pass_kernel(const float *scores, int N, const int *taken, int n_taken, unsigned long long *best) {
    unsigned long long loc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        bool skip = false;
        for (int t = 0; t < n_taken; ++t) skip |= (taken[t] == i);
        if (skip) continue;
        // order-preserving map of the float bits, index in the low word
        unsigned int u = __float_as_uint(scores[i]);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        const unsigned long long key = ((unsigned long long)u << 32) | (unsigned int)i;
        loc = key > loc ? key : loc;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(loc, off);
        loc = o > loc ? o : loc;
    }
    if ((threadIdx.x & 63) == 0 && loc) atomicMax(best, loc);
}

__global__ void take_best_kernel(const unsigned long long *best, const float *scores, int *taken, int slot, int *out_idx,
                                 float *out_score) {
    const int idx = (int)(*best & 0xffffffffu);
    taken[slot] = idx;
    out_idx[slot] = idx;
    out_score[slot] = scores[idx];
}

}  // namespace mi355

using namespace mi355;

extern "C" int mi355_label_remap(const uint8_t *in_dev, uint8_t *out_dev, int64_t n, const uint8_t *map256_host, void *stream) {
    MI355_REQUIRE(in_dev && out_dev && map256_host && n >= 0, "bad argument");
    MI355_TRY(bind_device());
    ByteMap bm;
    for (int i = 0; i < 256; ++i) bm.m[i] = map256_host[i];
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(label_remap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, n, bm);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

extern "C" int mi355_label_confusion(const uint8_t *pred_dev, const uint8_t *gt_dev, int64_t n, int K, uint64_t *counts_host,
                                     void *stream) {
    MI355_REQUIRE(pred_dev && gt_dev && counts_host && K >= 2 && K <= 8, "bad argument (2 <= K <= 8)");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *c = nullptr;
    MI355_TRY(device_scratch(SCR_SMALL, s, 1 << 20, (void **)&c));
    MI355_HIP(hipMemsetAsync(c, 0, 64 * sizeof(unsigned long long), s));
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred_dev, gt_dev, n, K, c);
    unsigned long long h[64];
    hipError_t e = hipMemcpyAsync(h, c, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    MI355_HIP(e);
    for (int i = 0; i < K * K; ++i) counts_host[i] = h[i];
    return MI355_OK;
}

extern "C" int mi355_cosine_topk(const float *vectors_dev, const float *query_dev, int N, int D, int k, int32_t *idx_host,
                                 float *scores_host, void *stream) {
    MI355_REQUIRE(vectors_dev && query_dev && idx_host && scores_host && N >= 1 && D >= 1 && k >= 1 && k <= 64, "bad argument");
    if (k > N) k = N;
    hipStream_t s = (hipStream_t)stream;
    // scratch: [best u64 | out_s 64 f32 | taken 64 i32 | out_i 64 i32 | scores N f32], persistent across calls
    char *scr = nullptr;
    MI355_TRY(device_scratch(SCR_TOPK, s, 1024 + (size_t)N * sizeof(float), (void **)&scr));
    unsigned long long *best = (unsigned long long *)scr;
    float *out_s = (float *)(scr + 64);
    int *taken = (int *)(scr + 64 + 256), *out_i = (int *)(scr + 64 + 512);
    float *scores = (float *)(scr + 1024);
    int blocks = (int)(((int64_t)N * 64 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(row_dot_kernel, dim3(blocks), dim3(256), 0, s, vectors_dev, query_dev, N, D, scores);
    int b2 = (N + 255) / 256;
    if (b2 > 1024) b2 = 1024;
    for (int t = 0; t < k; ++t) {
        (void)hipMemsetAsync(best, 0, sizeof(unsigned long long), s);
        hipLaunchKernelGGL(argmax_pass_kernel, dim3(b2), dim3(256), 0, s, scores, N, taken, t, best);
        hipLaunchKernelGGL(take_best_kernel, dim3(1), dim3(1), 0, s, best, scores, taken, t, out_i, out_s);
    }
    hipError_t e = hipMemcpyAsync(idx_host, out_i, k * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(scores_host, out_s, k * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    MI355_HIP(e);
    return k;
}

// ---------------------------------------------------------------------------------------------------------------
// crop_to_nonzero on the device (nnU-Net v1 cropping.crop_to_nonzero as trainer.preprocess_patient
// runs it, run_brats2021_inference_singlethread.py:89): mask = OR_c(vol[c] != 0), scipy.ndimage.binary_fill_holes
// (6-connectivity), bounding box.  Hole filling = flood fill of the background from the volume border: `state` is
// 1 = tissue, 0 = background not yet reached, 2 = background connected to the border.  One sweep kernel per axis walks
// every line of that axis forwards and backwards (a thread per line); sweeps repeat until none of them changes a
// voxel - brain masks are nearly convex, so two or three rounds.  Integer work: bit-exact against scipy.
__global__ void nonzero_state_kernel(const float *vol, int C, int64_t V, uint8_t *state) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        bool nz = false;
        for (int c = 0; c < C; ++c) nz |= vol[(int64_t)c * V + i] != 0.f;
        state[i] = nz ? 1 : 0;
    }
}

// axis 0 = z, 1 = y, 2 = x.  A line's end points are border voxels: background there is "outside" by definition.
__global__ void fill_sweep_kernel(uint8_t *state, int Z, int Y, int X, int axis, int *changed) {
    const int n1 = axis == 0 ? Y : Z, n2 = axis == 2 ? Y : X;          // the two other extents
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= n1 * n2) return;
    const int a = line / n2, b = line - a * n2;
    const int len = axis == 0 ? Z : (axis == 1 ? Y : X);
    int64_t base, stride;
    if (axis == 0) { base = (int64_t)a * X + b; stride = (int64_t)Y * X; }                 // a = y, b = x
    else if (axis == 1) { base = (int64_t)a * Y * X + b; stride = X; }                     // a = z, b = x
    else { base = ((int64_t)a * Y + b) * X; stride = 1; }                                  // a = z, b = y
    bool ch = false;
    bool reach = true;  // beyond the border everything is outside
    for (int i = 0; i < len; ++i) {
        uint8_t &v = state[base + i * stride];
        if (v == 1) reach = false;
        else if (v == 2) reach = true;
        else if (reach) { v = 2; ch = true; }
    }
    reach = true;
    for (int i = len - 1; i >= 0; --i) {
        uint8_t &v = state[base + i * stride];
        if (v == 1) reach = false;
        else if (v == 2) reach = true;
        else if (reach) { v = 2; ch = true; }
    }
    if (ch) *changed = 1;
}

// mask = state != 2 (tissue or enclosed background); bbox[0..2] = min z,y,x, bbox[3..5] = max z,y,x over the mask
__global__ void fill_finish_kernel(const uint8_t *state, uint8_t *mask, int Z, int Y, int X, int *bbox) {
    const int64_t V = (int64_t)Z * Y * X;
    int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        const bool m = state[i] != 2;
        mask[i] = m ? 1 : 0;
        if (m) {
            const int x = (int)(i % X), y = (int)((i / X) % Y), z = (int)(i / ((int64_t)X * Y));
            lo[0] = min(lo[0], z); lo[1] = min(lo[1], y); lo[2] = min(lo[2], x);
            hi[0] = max(hi[0], z); hi[1] = max(hi[1], y); hi[2] = max(hi[2], x);
        }
    }
    // wave butterfly -> LDS across the four waves -> ONE atomic per bound and workgroup (round 2: every wave of 8192 workgroups
    // added to the same six words - 196k serialised atomics, 2.2 ms for an 8.9-M-voxel volume)
    __shared__ int red[4][6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int l = lo[k], h = hi[k];
        for (int m = 1; m < 64; m <<= 1) { l = min(l, __shfl_xor(l, m)); h = max(h, __shfl_xor(h, m)); }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = l; red[threadIdx.x >> 6][3 + k] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        int v = red[0][k];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = k < 3 ? min(v, red[w][k]) : max(v, red[w][k]);
        if (k < 3) { if (v < (1 << 30)) atomicMin(&bbox[k], v); }
        else if (v >= 0) atomicMax(&bbox[k], v);
    }
}

extern "C" int mi355_crop_mask(const float *vol_dev, int C, int Z, int Y, int X, uint8_t *mask_dev, int32_t *bbox_host, void *stream) {
    MI355_REQUIRE(vol_dev && mask_dev && bbox_host && C >= 1 && Z >= 1 && Y >= 1 && X >= 1, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t V = (int64_t)Z * Y * X;
    char *scr = nullptr;   // persistent: [flags 8 i32 | pad to 256 | state V u8]
    MI355_TRY(device_scratch(SCR_CROP, s, 256 + (size_t)V, (void **)&scr));
    int *flags = (int *)scr;  // [0] = changed, [1..6] = bbox
    uint8_t *state = (uint8_t *)(scr + 256);
    int64_t blocks = (V + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(nonzero_state_kernel, dim3((unsigned)blocks), dim3(256), 0, s, vol_dev, C, V, state);
    hipError_t e = hipGetLastError();
    bool converged = false;
    for (int round = 0; e == hipSuccess && round < 4096; ++round) {
        (void)hipMemsetAsync(flags, 0, sizeof(int), s);
        for (int axis = 0; axis < 3; ++axis) {
            const int lines = axis == 0 ? Y * X : (axis == 1 ? Z * X : Z * Y);
            hipLaunchKernelGGL(fill_sweep_kernel, dim3((unsigned)((lines + 127) / 128)), dim3(128), 0, s, state, Z, Y, X, axis, flags);
        }
        int changed = 0;
        e = hipMemcpyAsync(&changed, flags, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess && !changed) { converged = true; break; }
    }
    int init[7] = {0, 1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
    if (e == hipSuccess) e = hipMemcpyAsync(flags, init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fill_finish_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s, state, mask_dev, Z, Y, X, flags + 1);
        e = hipGetLastError();
    }
    int out[7];
    if (e == hipSuccess) e = hipMemcpyAsync(out, flags, sizeof(out), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    MI355_HIP(e);
    MI355_REQUIRE(converged, "crop_mask: the border flood fill did not converge in 4096 rounds (%dx%dx%d)", Z, Y, X);
    MI355_REQUIRE(out[4] >= 0, "volume is all zeros: nothing to segment");
    for (int k = 0; k < 3; ++k) { bbox_host[2 * k] = out[1 + k]; bbox_host[2 * k + 1] = out[4 + k] + 1; }  // [lo, hi) per axis
    return MI355_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-label voxel statistics of a label map (feature_extraction/utils.py:167-216: get_tumor_masks,
// calculate_volume, get_centroid, get_bounding_box).  One pass yields, for every label value < K: voxel count, the
// coordinate sums along the three array axes (exact int64) and the coordinate minima / maxima; volumes, centroids and
// bounding boxes of the reference's regions (ncr, ed, et, tc, wt) are unions of labels and follow from these integers.
// out[label * 10 + {0: count, 1..3: sum of coordinate 0..2, 4..6: min, 7..9: max}]
__global__ void label_stats_kernel(const uint8_t *seg, int d0, int d1, int d2, int K, long long *out) {
    __shared__ long long acc[8 * 10];
    for (int i = threadIdx.x; i < 80; i += blockDim.x) acc[i] = (i % 10 >= 4 && i % 10 <= 6) ? (1ll << 40) : (i % 10 >= 7 ? -1 : 0);
    __syncthreads();
    const int64_t V = (int64_t)d0 * d1 * d2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = seg[i];
        if (l == 0 || l >= K) continue;  // the background is derived from the total (keeps the atomics off 97 % of voxels)
        const long long c2 = (long long)(i % d2), c1 = (long long)((i / d2) % d1), c0 = (long long)(i / ((int64_t)d2 * d1));
        long long *a = acc + l * 10;
        atomicAdd((unsigned long long *)&a[0], 1ull);
        atomicAdd((unsigned long long *)&a[1], (unsigned long long)c0);
        atomicAdd((unsigned long long *)&a[2], (unsigned long long)c1);
        atomicAdd((unsigned long long *)&a[3], (unsigned long long)c2);
        atomicMin(&a[4], c0); atomicMin(&a[5], c1); atomicMin(&a[6], c2);
        atomicMax(&a[7], c0); atomicMax(&a[8], c1); atomicMax(&a[9], c2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * 10; i += blockDim.x) {
        const int f = i % 10;
        if (i < 10) continue;
        if (f < 4) { if (acc[i]) atomicAdd((unsigned long long *)&out[i], (unsigned long long)acc[i]); }
        else if (f < 7) atomicMin(&out[i], acc[i]);
        else atomicMax(&out[i], acc[i]);
    }
}

extern "C" int mi355_label_stats(const uint8_t *seg_dev, int d0, int d1, int d2, int K, int64_t *stats_host, void *stream) {
    MI355_REQUIRE(seg_dev && stats_host && d0 >= 1 && d1 >= 1 && d2 >= 1 && K >= 2 && K <= 8, "bad argument (2 <= K <= 8)");
    hipStream_t s = (hipStream_t)stream;
    long long init[80], *dev = nullptr;
    for (int i = 0; i < 80; ++i) init[i] = (i % 10 >= 4 && i % 10 <= 6) ? (1ll << 40) : (i % 10 >= 7 ? -1 : 0);
    MI355_TRY(device_scratch(SCR_SMALL, s, 1 << 20, (void **)&dev));
    hipError_t e = hipMemcpyAsync(dev, init, sizeof(init), hipMemcpyHostToDevice, s);
    const int64_t V = (int64_t)d0 * d1 * d2;
    int64_t blocks = (V + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(label_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seg_dev, d0, d1, d2, K, dev);
        e = hipGetLastError();
    }
    long long h[80];
    if (e == hipSuccess) e = hipMemcpyAsync(h, dev, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    MI355_HIP(e);
    long long nonzero = 0;
    for (int l = 1; l < K; ++l) nonzero += h[l * 10];
    for (int i = 0; i < K * 10; ++i) stats_host[i] = h[i];
    stats_host[0] = V - nonzero;  // label 0: count only (its sums / box are not used by the reference)
    for (int f = 1; f < 10; ++f) stats_host[f] = f < 4 ? 0 : (f < 7 ? 0 : -1);
    return MI355_OK;
}
