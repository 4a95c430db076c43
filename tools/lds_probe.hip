// LDS read-pattern probe: cycles per ds_read_b128 of a wave for the address patterns a brick layout could give the MFMA
// B operand (lane = (voxel l31, channel half h)):  planar [half][voxel][16 B]  vs  interleaved [voxel][half][16 B].
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256, 1) void lds_read_kernel(float *out, unsigned long long *cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 16384; i += 256) ((float *)lds)[i] = (float)i;
    __syncthreads();
    unsigned a;
    if (PATTERN == 0) a = half * 16384 + l31 * 16;              // planar: consecutive lanes, consecutive 16-B slots
    else if (PATTERN == 1) a = l31 * 32 + half * 16;            // interleaved: voxel stride 32 B, the halves side by side
    else a = (l31 * 32 + half * 16) ^ ((l31 >> 2 & 1) * 16);    // interleaved + swap of the halves every four voxels
    a += wave * 2048;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {  // 16 reads in flight, no VALU between them: the LDS pipe is what is timed
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[u]) : "v"(a), "n"(u * 160));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" ::"v"(v[u]));
        if (it == iters - 1) acc = v[3];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int P>
static void run(const char *name) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL(lds_read_kernel<P>, dim3(256), dim3(256), 65536, 0, out, cyc, iters);
    hipLaunchKernelGGL(lds_read_kernel<P>, dim3(256), dim3(256), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-48s %6.2f cycles per wave-level ds_read_b128 (4 waves per CU reading; 8.0 = 128 B/clk per CU shared by 4 waves -> 32)\n", name, s / 256 / iters / 16);
    hipFree(out); hipFree(cyc);
}

int main() {
    hipFuncSetAttribute((const void *)lds_read_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)lds_read_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)lds_read_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<0>("planar [half][voxel][16 B]");
    run<1>("interleaved [voxel][half][16 B]");
    run<2>("interleaved, halves swapped every four voxels");
    return 0;
}
