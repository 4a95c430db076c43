// Micro-benchmark: what f32-MFMA rate do different issue structures sustain on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int BARRIER_EVERY, bool SHAPE16>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f32x16 acc[NACC];
    f32x4 acc4[NACC * 4];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < NACC * 4; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (!SHAPE16) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < NACC * 4; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
            }
        }
        if (BARRIER_EVERY > 0 && (it % BARRIER_EVERY) == BARRIER_EVERY - 1) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < NACC * 4; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int BARRIER_EVERY, bool SHAPE16>
void run(const char *name, int wgs_per_cu, size_t lds_bytes, int iters) {
    float *out;
    int grid = 256 * wgs_per_cu;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    auto k = probe<NACC, BARRIER_EVERY, SHAPE16>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, out, iters, 1.0f, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, 0, out, iters, 1.0f, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // flops: per wave per iter: 8 * NACC * 4096 (32x32x2) or 8 * NACC*4 * 2048 (16x16x4) -> both 8*NACC*4096... (4*2048 = 8192 per NACC)
    double per_wave = SHAPE16 ? 8.0 * NACC * 4 * 2048 : 8.0 * NACC * 4096;
    double flops = per_wave * iters * 4.0 * grid;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s  (err %s)\n", name, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

int main() {
    const int it = 4000;
    run<4, 0, false>("32x32x2  1 WG/CU (1 wave/SIMD) 4 acc, no barrier", 1, 100 * 1024, it);
    run<2, 0, false>("32x32x2  1 WG/CU (1 wave/SIMD) 2 acc, no barrier", 1, 100 * 1024, it);
    run<1, 0, false>("32x32x2  1 WG/CU (1 wave/SIMD) 1 acc, no barrier", 1, 100 * 1024, it);
    run<2, 0, false>("32x32x2  2 WG/CU (2 waves/SIMD) 2 acc, no barrier", 2, 60 * 1024, it);
    run<2, 27, false>("32x32x2  2 WG/CU 2 acc, barrier every 27x8x2 MFMAs", 2, 60 * 1024, it);
    run<2, 27, false>("32x32x2  1 WG/CU 2 acc, barrier every 27x8x2 MFMAs", 1, 100 * 1024, it);
    run<4, 27, false>("32x32x2  2 WG/CU 4 acc, barrier every 27x8x4 MFMAs", 2, 60 * 1024, it);
    run<2, 0, false>("32x32x2  4 WG/CU (4 waves/SIMD) 2 acc, no barrier", 4, 30 * 1024, it);
    run<1, 0, true>("16x16x4  1 WG/CU (1 wave/SIMD) 4 acc, no barrier", 1, 100 * 1024, it);
    run<2, 0, true>("16x16x4  2 WG/CU (2 waves/SIMD) 8 acc, no barrier", 2, 60 * 1024, it);
    run<2, 27, true>("16x16x4  2 WG/CU 8 acc, barrier every 27 iters", 2, 60 * 1024, it);
    return 0;
}
