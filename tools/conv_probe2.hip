// Micro-benchmark 2: software-pipelined conv inner loop, built up ingredient by ingredient.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: A/B prefetch distance 1 via register copies (v1 style), synchronous staging
// MODE 1: fully unrolled 27 steps, A dist 1, B ring dist 3, no staging
// MODE 2: MODE 1 + in-loop staging into the other buffer (linear addresses), 8 slots issue@2r write@2r+5
// MODE 3: MODE 2 with brick-style address arithmetic (two magic divisions + bounds + 64-bit index)
template <int MODE, int MF, bool SB>
__global__ __launch_bounds__(256, 2) void conv_like(const float *w, const float *in, float *out, int tiles,
                                                    unsigned mIX, unsigned mIY, int IX, int IY, int D, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[MF];
    for (int i = 0; i < MF; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int plane = 3264, buf_floats = 6528;
    int a_base[MF];
    for (int m = 0; m < MF; ++m) a_base[m] = (((wave * MF + m) * 32 + (lane & 31)) * 4) % 3000 + (lane >> 5) * plane;
    const int a_base0 = a_base[0], a_base1 = a_base[MF > 1 ? 1 : 0];
    for (int i = tid; i < 2 * buf_floats; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    const float *wl = w + lane * 4;
    f32x4 bq[3];
    bq[0] = *(const f32x4 *)(wl); bq[1] = *(const f32x4 *)(wl + 256); bq[2] = *(const f32x4 *)(wl + 512);
    int buf = 0;
    const int nchunks = tiles * 4;
    for (int c = 0; c < nchunks; ++c) {
        const int ch = c & 3;
        const float *bufc = lds + buf * buf_floats;
        float *bufn = lds + (buf ^ 1) * buf_floats;
        const float *wch = wl + ch * 27 * 256;
        const float *wnx = wl + ((ch + 1) & 3) * 27 * 256;
        const float *src = in + ((size_t)blockIdx.x * nchunks + c) * buf_floats;
        if (MODE == 0) {
            for (int r0 = 0; r0 < 7; r0 += 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { int i = (r0 + u) * 256 + tid; v[u] = (i < 1632) ? *(const f32x4 *)(src + i * 4) : bq[0]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { int i = (r0 + u) * 256 + tid; if (i < 1632) *(f32x4 *)(lds + i * 4) = v[u]; }
            }
            __syncthreads();
            f32x4 a0 = *(const f32x4 *)(lds + a_base0), a1 = *(const f32x4 *)(lds + a_base1), b = *(const f32x4 *)(wch);
            for (int tap = 0; tap < 27; ++tap) {
                f32x4 a0n = a0, a1n = a1, bn = b;
                if (tap + 1 < 27) {
                    const int off = ((tap + 1) % 3) * 4 + ((tap + 1) / 3) * 136;
                    a0n = *(const f32x4 *)(lds + a_base0 + off); a1n = *(const f32x4 *)(lds + a_base1 + off);
                    bn = *(const f32x4 *)(wch + (tap + 1) * 256);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a0[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a1[j], acc[1], 0, 0, 0);
                }
                a0 = a0n; a1 = a1n; b = bn;
            }
            __syncthreads();
        } else {
            f32x4 a[MF], an[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) { a[m] = *(const f32x4 *)(bufc + a_base[m]); an[m] = a[m]; }
            f32x4 st_v[8]; int st_dst[8]; bool st_in[8];
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                if (tap + 1 < 27) {
                    const int off = ((tap + 1) % 3) * 4 + ((tap + 1) / 3) * 136;
#pragma unroll
                    for (int m = 0; m < MF; ++m) an[m] = *(const f32x4 *)(bufc + a_base[m] + off);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MF; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[tap % 3][j], a[m][j], acc[m], 0, 0, 0);
                {
                    const int k = tap + 3;
                    const float *ws = (k < 27) ? wch + k * 256 : wnx + (k - 27) * 256;
                    bq[tap % 3] = *(const f32x4 *)ws;
                }
                if (MODE >= 2) {
                    if ((tap & 1) == 0 && tap / 2 < 8) {
                        const int r = tap / 2;
                        const int i = r * 256 + tid;
                        if (MODE == 2) {
                            st_dst[r] = (i < 1632) ? i * 4 : -1;
                            st_in[r] = i < 1632;
                            st_v[r] = *(const f32x4 *)(src + (st_in[r] ? i * 4 : 0));
                        } else {
                            const int bv = i >> 1;
                            const int rr = (int)((__umulhi((unsigned)bv, mIX) + bv) >> 6);
                            const int bx = bv - rr * IX;
                            const int bz = (int)((__umulhi((unsigned)rr, mIY) + rr) >> 3);
                            const int by = rr - bz * IY;
                            const int iz = (c & 63) - 1 + bz, iy = ((c >> 2) & 31) * 4 - 1 + by, ix = (c & 3) * 32 - 1 + bx;
                            st_dst[r] = (i < 1632) ? (i & 1) * plane + bv * 4 : -1;
                            st_in[r] = (i < 1632) && ((unsigned)iz < (unsigned)D) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
                            const size_t off = st_in[r] ? ((((size_t)(blockIdx.x & 7) * D + iz) * H + iy) * W + ix) * 32 + ch * 8 + (tid & 1) * 4 : 0;
                            st_v[r] = *(const f32x4 *)(in + off);
                        }
                    }
                    if (tap >= 5 && ((tap - 5) & 1) == 0 && (tap - 5) / 2 < 8) {
                        const int r = (tap - 5) / 2;
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        if (st_dst[r] >= 0) *(f32x4 *)(bufn + st_dst[r]) = st_in[r] ? st_v[r] : z;
                    }
                }
#pragma unroll
                for (int m = 0; m < MF; ++m) a[m] = an[m];
                if (SB) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            buf ^= 1;
        }
        if (ch == 3) {
            float *o = out + ((size_t)(blockIdx.x * tiles + (c >> 2)) * 128 * MF + wave * 32 * MF + (lane & 31)) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[mf][4 * g], acc[mf][4 * g + 1], acc[mf][4 * g + 2], acc[mf][4 * g + 3]};
                    *(f32x4 *)(o + mf * 32 * 32 + 8 * g) = v;
                }
            for (int i = 0; i < MF; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
    }
}

static unsigned magic(unsigned d, unsigned s) { return (unsigned)(((1ull << 32) * ((1ull << s) - d)) / d + 1); }

template <int MODE, int MF, bool SB>
void run(const char *name, const float *w, const float *in, float *out) {
    const int grid = 512, tiles = 32 / MF;
    auto k = conv_like<MODE, MF, SB>;
    const size_t ldsb = 54 * 1024;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // IX = 34 (s=6), IY = 6 (s=3)
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, w, in, out, tiles, magic(34, 6), magic(6, 3), 34, 6, 128, 128, 128);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, w, in, out, tiles, magic(34, 6), magic(6, 3), 34, 6, 128, 128, 128);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * tiles * 4 * 27 * 4 * MF * 4 * 4096.0;
    printf("%-72s %7.3f ms %7.1f TFLOP/s (%s)\n", name, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *w, *in, *out;
    hipMalloc(&w, 5 * 27 * 256 * 4 + 4096); hipMemset(w, 0, 5 * 27 * 256 * 4 + 4096);
    const size_t in_floats = (size_t)8 * 128 * 128 * 128 * 32;
    hipMalloc(&in, in_floats * 4); hipMemset(in, 0, in_floats * 4);
    hipMalloc(&out, (size_t)512 * 16 * 256 * 32 * 4);
    run<0, 2, true>("v1 style: sync staging, A/B prefetch dist 1 (register copies)", w, in, out);
    run<1, 2, true>("MF=2 unrolled, A dist 1, B ring 3, no staging, sched_barrier/step", w, in, out);
    run<1, 2, false>("MF=2 same, no sched_barrier", w, in, out);
    run<3, 2, true>("MF=2 + in-loop staging (brick addr), sched_barrier/step", w, in, out);
    run<3, 2, false>("MF=2 + in-loop staging (brick addr), no sched_barrier", w, in, out);
    run<1, 4, true>("MF=4 unrolled, no staging, sched_barrier/step", w, in, out);
    run<1, 4, false>("MF=4 unrolled, no staging, no sched_barrier", w, in, out);
    run<3, 4, true>("MF=4 + in-loop staging (brick addr), sched_barrier/step", w, in, out);
    run<3, 4, false>("MF=4 + in-loop staging (brick addr), no sched_barrier", w, in, out);
    return 0;
}
