// Micro-benchmark: which fp16 MFMA shape delivers more FLOP/s in the inner loop of the fp16 conv kernels on gfx950?
// MI355X_MICROARCH.md ('DVFS give-back' item 7): under an MFMA-dense loop the chip lowers its clock, and the clock it holds
// depends on the MFMA shape - cycles per FLOP do not decide.  Both arms below have the conv kernel's wave tile (128 voxels x
// 64 couts, one wave per SIMD, 128 accumulator registers), read the voxel operand from LDS with ds_read_b128 (the same
// bytes per FLOP in both shapes) and keep the weight operand in registers; operands are uniform random in [-1, 1).
//   arm 0: v_mfma_f32_32x32x16_f16, per K = 16 step 4 voxel fragments x 2 cout fragments = 8 MFMAs
//   arm 1: v_mfma_f32_16x16x32_f16, per K = 32 step 8 voxel fragments x 4 cout fragments = 32 MFMAs
// LDS_MODE 0: conflict-free reads, 1: the 2-way conflicts of the interleaved brick, 2: no LDS reads (operands in registers)
// Reports TFLOP/s (wall) and the in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/shape_probe.hip -o tools/shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// DMA_LINES: 0 = no LDS-DMA in the loop; else four global_load_lds_dwordx4 per 128 MFMA-equivalents (the conv kernels' brick
// rate), each touching DMA_LINES different 128-B lines (64 / 32 / 16 / 8: 1 / 2 / 4 / 8 lanes per line) of an L2-resident region
template <int SHAPE, int LDS_MODE, int DMA_LINES = 0>
__global__ __launch_bounds__(256, 1) void probe(const f16x8 *src, float *out, unsigned long long *clk, int iters, const char *stream) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 64 KiB of random halfs in LDS
    for (int i = tid; i < 4096; i += 256) ((f16x8 *)lds)[i] = src[(blockIdx.x & 15) * 4096 + i];
    f16x8 w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = src[65536 + k * 64 + lane];
    __syncthreads();
    unsigned a;  // this lane's fragment base in LDS
    if (LDS_MODE == 1) a = (lane & 31) * 32 + (lane >> 5) * 16;  // [voxel][half][16 B]: 2-way conflicts
    else a = lane * 16;                                            // consecutive 16-B slots: conflict-free
    a += wave * 4096;
    // LDS-DMA source: this wave's 24-KiB window of `stream` (3 MiB per XCD: L2 hits, L1 misses), lane pattern by DMA_LINES
    constexpr int LPL = DMA_LINES ? 64 / DMA_LINES : 1;  // lanes per line
    const char *dsrc = stream + (size_t)(blockIdx.x * 4 + wave) * 24576 + (lane / LPL) * 256 + (lane % LPL) * 16;
    char *ddst = lds + 32768 + wave * 1024;  // (never read)
    auto dma = [&](int n) {
        if constexpr (DMA_LINES != 0) {
            const char *g = dsrc + (unsigned)((n * DMA_LINES * 256) % 24576);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)ddst, 16, 0, 0);
        }
    };
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        f32x16 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        f16x8 b[2][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) b[0][m] = *(const __attribute__((address_space(3))) f16x8 *)(a + m * 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {  // 16 K-steps per iteration, fragments double-buffered one step ahead
                // (the conv kernels' order: two MFMAs, then the next step's reads + the DMA, then the other MFMAs - hipcc waits
                //  lgkmcnt(0) before a step's first MFMA, so the reads must be a step old by then)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[s & 3], b[s & 1][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(s + 1) & 3], b[s & 1][0], acc[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LDS_MODE != 2) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        b[(s + 1) & 1][m] = *(const __attribute__((address_space(3))) f16x8 *)(a + ((s * 4 + m) & 15) * 1024 + (s & 3) * 32);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) b[(s + 1) & 1][m] = b[s & 1][m];
                }
                if (s % 4 == 1) dma(it * 4 + s / 4);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[(s + n) & 3], b[s & 1][m], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[m][n][r];
        out[blockIdx.x * 256 + tid] = sum;
    } else {
        f32x4 acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
        f16x8 b[2][8];
#pragma unroll
        for (int m = 0; m < 8; ++m) b[0][m] = *(const __attribute__((address_space(3))) f16x8 *)(a + (m & 3) * 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // 8 K = 32 steps per iteration = the same FLOPs and LDS bytes as 16 K = 16 steps
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(s + n) & 3], b[s & 1][0], acc[0][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LDS_MODE != 2) {
#pragma unroll
                    for (int m = 0; m < 8; ++m)
                        b[(s + 1) & 1][m] = *(const __attribute__((address_space(3))) f16x8 *)(a + ((s * 8 + m) & 15) * 1024 + (s & 3) * 32);
                } else {
#pragma unroll
                    for (int m = 0; m < 8; ++m) b[(s + 1) & 1][m] = b[s & 1][m];
                }
                if (s % 2 == 1) dma(it * 4 + s / 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[(s + n) & 3], b[s & 1][m], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) sum += acc[m][n][r];
        out[blockIdx.x * 256 + tid] = sum;
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

static f16x8 *g_src; static float *g_out; static unsigned long long *g_clk; static char *g_stream;

template <int SHAPE, int LDS_MODE, int DMA_LINES = 0>
static void run(const char *name, int iters, int rounds) {
    auto k = probe<SHAPE, LDS_MODE, DMA_LINES>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, g_src, g_out, g_clk, iters, g_stream);  // settle the clock
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, g_src, g_out, g_clk, iters, g_stream);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= rounds;
    std::vector<unsigned long long> c(512);
    hipMemcpy(c.data(), g_clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    double cyc = 0;
    for (int i = 0; i < 256; ++i) { ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1); cyc += (double)c[2 * i]; }
    std::sort(ghz.begin(), ghz.end());
    // FLOPs per wave and iteration: 16 steps x 8 MFMAs x 32*32*16*2 = 16 x 8 x 32768 (both arms)
    const double flops = 16.0 * 8 * 32768 * (double)iters * 4 * 256;
    printf("%-64s %8.3f ms %8.1f TFLOP/s  clock %.3f GHz (median)  %.1f cycles per 32x32x16-equivalent MFMA\n", name, ms, flops / ms / 1e9,
           ghz[128], cyc / 256 / ((double)iters * 16 * 8));
    fflush(stdout);
}

int main() {
    std::vector<_Float16> h((65536 + 256) * 8);
    uint32_t sd = 777u;
    for (auto &v : h) { sd ^= sd << 13; sd ^= sd >> 17; sd ^= sd << 5; v = (_Float16)((float)(int32_t)sd * (1.0f / 2147483648.0f)); }
    hipMalloc(&g_src, h.size() * 2); hipMemcpy(g_src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&g_out, 256 * 256 * 4); hipMalloc(&g_clk, 512 * 8);
    hipMalloc(&g_stream, (size_t)1024 * 24576 + 65536); hipMemset(g_stream, 1, (size_t)1024 * 24576 + 65536);
    const int iters = 20000, rounds = 8;  // ~20-40 ms per launch
    for (int rep = 0; rep < 2; ++rep) {  // interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24)
        run<0, 0>("32x32x16  LDS conflict-free", iters, rounds);
        run<1, 0>("16x16x32  LDS conflict-free", iters, rounds);
        run<0, 1>("32x32x16  LDS 2-way conflicts (interleaved brick)", iters, rounds);
        run<1, 1>("16x16x32  LDS 2-way conflicts", iters, rounds);
        run<0, 2>("32x32x16  operands in registers", iters, rounds);
        run<1, 2>("16x16x32  operands in registers", iters, rounds);
        run<0, 1, 64>("32x32x16  2-way LDS + DMA 64 lines per instruction", iters, rounds);
        run<0, 1, 32>("32x32x16  2-way LDS + DMA 32 lines per instruction", iters, rounds);
        run<0, 1, 16>("32x32x16  2-way LDS + DMA 16 lines per instruction", iters, rounds);
        run<0, 1, 8>("32x32x16  2-way LDS + DMA  8 lines per instruction", iters, rounds);
        run<1, 0, 16>("16x16x32  conflict-free LDS + DMA 16 lines per instruction", iters, rounds);
        run<1, 0, 8>("16x16x32  conflict-free LDS + DMA  8 lines per instruction", iters, rounds);
    }
    return 0;
}
