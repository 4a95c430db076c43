// Micro-benchmark: what does one global_load_lds_dwordx4 (64 lanes x 16 B -> 1 KiB of LDS) cost the L1 / texture-address path
// as a function of how its lanes are spread over 128-B lines?  The brick staging of the fp16 conv kernels is exactly this
// instruction, and the stride-2 kernel issues five times as many per MFMA as the stride-1 kernel.
// Patterns (voxel pitch 256 B = a 128-channel fp16 tensor; the region is L2-resident, larger than the L1):
//   0: 64 lanes -> 64 voxels, 16 B each (planar brick: one 8-channel half per instruction)          64 lines, 16 B used per line
//   1: lanes 2i, 2i+1 -> the 32 B of voxel i (interleaved brick)                                      32 lines, 32 B per line
//   2: lanes i, i+32 -> the two 16-B halves of voxel i (block-planar brick)                           32 lines, 32 B per line
//   3: lanes 4i .. 4i+3 -> 64 B of voxel i (32-channel chunk)                                         16 lines, 64 B per line
//   4: lanes 8i .. 8i+7 -> the 128 B of voxel i (64-channel chunk: whole lines)                        8 lines
//   5: 64 lanes -> 1 KiB contiguous                                                                    8 lines
//   6: pattern 0 twice in a row on the same voxels (half 0, then half 1): what pairing the planes' pieces buys
// Channel-blocked tensors (16 B per voxel of a block, x-consecutive voxels contiguous; row pitch 2 KiB = 128 voxels):
//   7: rows of 10 consecutive voxels (the stride-1 brick: 160 contiguous bytes per row, 6.4 rows per instruction)
//   8: rows of 17 voxels, EVEN voxels only (16 B at a 32-B stride: the stride-2 kernel's parity-split planar image)
//   9: rows of 17 voxels, lanes in the order even voxels then odd voxels (all 272 bytes of a row by one instruction, the lane
//      order permuted within the row)
//  10: pattern 7 with the row starting 16 B before a line boundary, as a brick whose origin is one voxel left of a tile does:
//      16 + 128 + 16 bytes of three lines per row (calibration of FETCH_SIZE for the conv kernels' staging: run under
//      rocprofv3 --pmc FETCH_SIZE; bytes used per launch = 256 x 4 x 20000 KiB = 20.97 GB)
// One workgroup of 4 waves per CU, `iters` instructions per wave back to back, vmcnt(8) in flight.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_probe.hip -o tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int PATTERN>
__global__ __launch_bounds__(256, 1) void probe(const char *src, unsigned long long *clk, int iters, int window) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned off;
    if (PATTERN == 0 || PATTERN == 6) off = lane * 256;
    else if (PATTERN == 1) off = (lane >> 1) * 256 + (lane & 1) * 16;
    else if (PATTERN == 2) off = (lane & 31) * 256 + (lane >> 5) * 16;
    else if (PATTERN == 3) off = (lane >> 2) * 256 + (lane & 3) * 16;
    else if (PATTERN == 4) off = (lane >> 3) * 256 + (lane & 7) * 16;
    else if (PATTERN == 7) off = (lane / 10) * 2048 + (lane % 10) * 16;
    else if (PATTERN == 10) off = (lane / 10) * 2048 + 112 + (lane % 10) * 16;
    else if (PATTERN == 8) off = (lane / 9) * 2048 + (lane % 9) * 32;
    else if (PATTERN == 9) { const int r = lane / 17, c = lane % 17; off = r * 2048 + (c < 9 ? 2 * c : 2 * (c - 9) + 1) * 16; }
    else off = lane * 16;
    constexpr int SPAN = PATTERN >= 7 ? 8 * 2048 : PATTERN == 0 || PATTERN == 6 ? 64 * 256 : PATTERN == 1 || PATTERN == 2 ? 32 * 256 : PATTERN == 3 ? 16 * 256 : PATTERN == 4 ? 8 * 256 : 1024;
    const char *base = src + (size_t)(blockIdx.x * 4 + wave) * window + off;
    char *dst = lds + wave * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned pos = 0;
    for (int it = 0; it < iters; ++it) {
        const char *g = base + pos;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        if (PATTERN == 6) {
            const char *g2 = g + 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g2, (__attribute__((address_space(3))) void *)(dst + 4096), 16, 0, 0);
        }
        pos += SPAN;
        if (pos + SPAN > (unsigned)window) pos = 0;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

static char *g_src; static unsigned long long *g_clk;

template <int PATTERN>
static void run(const char *name, int window) {
    const int iters = 20000;
    auto k = probe<PATTERN>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, g_src, g_clk, iters, window);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, g_src, g_clk, iters, window);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(256);
    hipMemcpy(c.data(), g_clk, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double instr = (double)iters * (PATTERN == 6 ? 2 : 1);
    printf("%-66s window %6d KiB/wave: %7.1f cycles per instruction and wave (median workgroup), %6.1f B/clk per CU into LDS, %7.3f ms\n", name,
           window / 1024, (double)c[128] / instr, 4.0 * 1024.0 * instr / (double)c[128], ms);
    fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)1024 * (1 << 20) + (1 << 20);  // 1 MiB window per wave at most
    hipMalloc(&g_src, bytes); hipMemset(g_src, 1, bytes); hipMalloc(&g_clk, 256 * 8);
    for (int window : {24 * 1024, 512 * 1024}) {  // 3 MiB per XCD (L2 hits) / 64 MiB per XCD (Infinity Cache / HBM)
        run<0>("0: 64 lines x 16 B (planar)", window);
        run<1>("1: 32 lines x 32 B, adjacent lanes (interleaved)", window);
        run<2>("2: 32 lines x 32 B, lanes i and i+32 (block-planar)", window);
        run<3>("3: 16 lines x 64 B", window);
        run<4>("4: 8 whole lines (8 lanes per voxel)", window);
        run<5>("5: 1 KiB contiguous", window);
        run<6>("6: pattern 0, both halves back to back", window);
        run<7>("7: blocked layout, rows of 10 voxels x 16 B", window);
        run<8>("8: blocked layout, rows of 17: even voxels only (32-B stride)", window);
        run<9>("9: blocked layout, rows of 17: even then odd voxels (permuted)", window);
        run<10>("10: blocked layout, rows of 10 voxels starting 16 B before a line", window);
    }
    return 0;
}
