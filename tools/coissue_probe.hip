// Micro-benchmark: how much matrix-pipe time does a non-MFMA instruction between two f32 MFMAs cost, at 1 and 2
// waves per SIMD?  (the question behind conv3_f32_wino2_kernel's schedule)
// Build: hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o gpurun_out/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 = nothing, 1 = NV x v_pk_add_f32, 2 = NV x v_add_f32, 3 = NV x ds_read_b64, 4 = NV x v_mov_b32
template <int KIND, int NV, int WPS>
__global__ __launch_bounds__(256, WPS) void probe(float *out, int iters, float a0) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a0;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = 1e-3f;
    f32x2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k][0] = a0 * k; v[k][1] = a0 + k; }
    const float *lp = lds + (threadIdx.x & 63) * 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 15], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int k = (i * NV + n) & 7;
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(v[(k + 3) & 7]));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k][0]) : "v"(v[(k + 3) & 7][1]));
                if (KIND == 3) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[k]) : "v"((unsigned)(size_t)lp), "n"(0));
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(v[k][0]) : "v"(v[(k + 3) & 7][1]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k][0] + v[k][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NV, int WPS>
void run(const char *name, int iters) {
    float *out;
    const int grid = 256 * WPS;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    auto k = probe<KIND, NV, WPS>;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 32.0 * 4096 * iters * 4.0 * grid;
    printf("%-44s waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s  (%s)\n", name, WPS, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    hipFree(out);
}

int main() {
    const int it = 3000;
    run<0, 0, 1>("MFMA only", it);
    run<0, 0, 2>("MFMA only", it);
    run<1, 1, 1>("+1 v_pk_add_f32 per MFMA", it);
    run<1, 1, 2>("+1 v_pk_add_f32 per MFMA", it);
    run<1, 2, 1>("+2 v_pk_add_f32 per MFMA", it);
    run<1, 2, 2>("+2 v_pk_add_f32 per MFMA", it);
    run<1, 4, 1>("+4 v_pk_add_f32 per MFMA", it);
    run<1, 4, 2>("+4 v_pk_add_f32 per MFMA", it);
    run<1, 8, 1>("+8 v_pk_add_f32 per MFMA", it);
    run<2, 2, 1>("+2 v_add_f32 per MFMA", it);
    run<2, 4, 1>("+4 v_add_f32 per MFMA", it);
    run<2, 8, 1>("+8 v_add_f32 per MFMA", it);
    run<2, 8, 2>("+8 v_add_f32 per MFMA", it);
    run<4, 4, 1>("+4 v_mov_b32 per MFMA", it);
    run<4, 8, 1>("+8 v_mov_b32 per MFMA", it);
    run<3, 1, 1>("+1 ds_read_b64 per MFMA", it);
    run<3, 1, 2>("+1 ds_read_b64 per MFMA", it);
    run<3, 2, 1>("+2 ds_read_b64 per MFMA", it);
    return 0;
}
