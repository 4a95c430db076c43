// Micro-benchmark: which ingredient of the conv inner loop costs MFMA issue slots on gfx950?
// Persistent-free skeleton: each WG does `tiles` tiles x 4 chunks x 27 steps x 8 MFMA (32x32x2 f32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// flags: 1 = A from LDS each step, 2 = B from global each step, 4 = epilogue stores, 8 = barrier per chunk,
//        16 = brick staging loads (global->LDS, 7 rounds per chunk, synchronous at chunk start)
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void conv_like(const float *w, const float *in, float *out, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int a_base0 = ((wave * 2 + 0) * 32 + (lane & 31)) * 4 + (lane >> 5) * 3264;
    const int a_base1 = ((wave * 2 + 1) * 32 + (lane & 31)) * 4 + (lane >> 5) * 3264;
    f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0, b = {1e-3f, 1e-3f, 1e-3f, 1e-3f};
    for (int i = tid; i < 6528; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    for (int t = 0; t < tiles; ++t) {
        for (int ch = 0; ch < 4; ++ch) {
            if (FLAGS & 16) {
                // 1632 pieces of 16 B per chunk: 7 rounds, issue 4 / write 4
                const float *src = in + ((size_t)(blockIdx.x * tiles + t) * 4 + ch) * 6528;
                for (int r0 = 0; r0 < 7; r0 += 4) {
                    f32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { int i = (r0 + u) * 256 + tid; v[u] = (i < 1632) ? *(const f32x4 *)(src + i * 4) : b; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { int i = (r0 + u) * 256 + tid; if (i < 1632) *(f32x4 *)(lds + i * 4) = v[u]; }
                }
                __syncthreads();
            }
            const float *wch = w + ch * 27 * 256 + lane * 4;
#pragma unroll 3
            for (int tap = 0; tap < 27; ++tap) {
                if (FLAGS & 1) {
                    a0 = *(const f32x4 *)(lds + a_base0 + (tap % 3) * 4 + (tap / 3) * 136);
                    a1 = *(const f32x4 *)(lds + a_base1 + (tap % 3) * 4 + (tap / 3) * 136);
                }
                if (FLAGS & 2) b = *(const f32x4 *)(wch + tap * 256);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a0[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a1[j], acc[1], 0, 0, 0);
                }
            }
            if (FLAGS & 8) __syncthreads();
        }
        if (FLAGS & 4) {
            float *o = out + ((size_t)(blockIdx.x * tiles + t) * 256 + wave * 64 + (lane & 31)) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[mf][4 * g], acc[mf][4 * g + 1], acc[mf][4 * g + 2], acc[mf][4 * g + 3]};
                    *(f32x4 *)(o + mf * 32 * 32 + 8 * g) = v;
                }
            for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
    }
    if (!(FLAGS & 4)) {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 256 + tid] = s;
    }
}

template <int FLAGS>
void run(const char *name, const float *w, const float *in, float *out) {
    const int grid = 512, tiles = 16;
    auto k = conv_like<FLAGS>;
    const size_t ldsb = 60 * 1024;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, w, in, out, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, w, in, out, tiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * tiles * 4 * 27 * 8 * 4 * 4096.0;
    printf("%-64s %7.3f ms %7.1f TFLOP/s (%s)\n", name, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
    float *w, *in, *out;
    hipMalloc(&w, 4 * 27 * 256 * 4 + 4096); hipMemset(w, 0, 4 * 27 * 256 * 4 + 4096);
    const size_t in_floats = (size_t)512 * 16 * 4 * 6528;
    hipMalloc(&in, in_floats * 4); hipMemset(in, 0, in_floats * 4);
    hipMalloc(&out, (size_t)512 * 16 * 256 * 32 * 4);
    run<0>("MFMA only", w, in, out);
    run<8>("+ barrier per chunk", w, in, out);
    run<1 | 8>("+ A from LDS", w, in, out);
    run<2 | 8>("+ B from global (no A)", w, in, out);
    run<1 | 2 | 8>("+ A from LDS + B from global", w, in, out);
    run<1 | 2 | 4 | 8>("+ A + B + epilogue stores", w, in, out);
    run<1 | 2 | 4 | 8 | 16>("+ A + B + epilogue + synchronous brick staging", w, in, out);
    run<4 | 8>("epilogue stores only", w, in, out);
    run<16 | 8>("brick staging only", w, in, out);
    return 0;
}
