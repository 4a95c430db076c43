// Checks common.h's half32_reduce_scatter (transposing reduction of the norm statistics) against plain sums on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<pkg>/csrc tools/stat_probe.hip -o tools/stat_probe
#include "common.h"
#include <vector>
using namespace mi355;
__global__ void k(const float *in, float *out) {
    const int lane = threadIdx.x;
    stat_f32x2 s1[8], s2[8];
    for (int i = 0; i < 16; ++i) { s1[i >> 1][i & 1] = in[(0 * 16 + i) * 64 + lane]; s2[i >> 1][i & 1] = in[(1 * 16 + i) * 64 + lane]; }
    out[lane] = half32_reduce_scatter(s1, s2, lane);
}
int main() {
    std::vector<float> h(2 * 16 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 7) % 1021) - 510.f;  // integers: sums exact
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(din, dout);
    float o[64];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int r = ((lane >> 1) & 1) + 2 * (lane & 1) + (lane & 12), kk = (lane >> 4) & 1, half = lane >> 5;
        float want = 0;
        for (int l = 0; l < 32; ++l) want += h[(kk * 16 + r) * 64 + half * 32 + l];
        if (want != o[lane]) {
            ++bad;
            int fk = -1, fr = -1, fh = -1;  // which total is it, if any?
            for (int k2 = 0; k2 < 2; ++k2) for (int r2 = 0; r2 < 16; ++r2) for (int h2 = 0; h2 < 2; ++h2) {
                float t = 0; for (int l = 0; l < 32; ++l) t += h[(k2 * 16 + r2) * 64 + h2 * 32 + l];
                if (t == o[lane]) { fk = k2; fr = r2; fh = h2; }
            }
            printf("lane %2d: got %g want %g (k %d r %d half %d); got is the total of k %d r %d half %d\n", lane, o[lane], want, kk, r, half, fk, fr, fh);
        }
    }
    printf("%s: %d of 64 lanes wrong\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
