// Diagnostic harness for the fp16 stride-2 convs: drives conv3d_mfma_f16 exactly as the network does on the stride-2 layer
// shapes of bench config 3 and times it with HIP events.  MI355_F16_S2=0 in the environment selects round 2's kernel
// (conv3_f16_mfma_pipe_kernel<1,2,..,2>) for an A/B on the same box.  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I<pkg>/csrc tools/s2_probe.hip -o tools/s2_probe
#include "conv3d_f16.hip"
#include "conv3d_f16_s2.hip"

namespace mi355 {
void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int bind_device() { return MI355_OK; }
int device_scratch(int slot, hipStream_t, size_t bytes, void **out, bool zeroed) {
    static void *p[SCR_COUNT]; static size_t n[SCR_COUNT];
    if (n[slot] < bytes) { if (p[slot]) (void)hipFree(p[slot]); if (hipMalloc(&p[slot], bytes) != hipSuccess) return MI355_ERR_HIP; n[slot] = bytes; if (zeroed) (void)hipMemset(p[slot], 0, bytes); }
    *out = p[slot];
    return MI355_OK;
}
}  // namespace mi355
using namespace mi355;

static int run(int N, int D, int cin, int cout, int reps, bool stats) {
    const size_t vin = (size_t)N * D * D * D, vout = vin / 8;
    std::vector<_Float16> x(vin * cin);
    std::vector<float> w((size_t)cout * cin * 27), b(cout);
    uint32_t sd = 12345u;
    auto u = [&]() { sd ^= sd << 13; sd ^= sd >> 17; sd ^= sd << 5; return (float)(int32_t)sd * (1.0f / 2147483648.0f); };  // [-1, 1)
    for (auto &v : x) v = (_Float16)u();
    for (auto &v : w) v = u() * 0.05f;
    for (auto &v : b) v = u();
    _Float16 *xd, *yd; double *st;
    hipMalloc(&xd, x.size() * 2); hipMalloc(&yd, vout * cout * 2); hipMalloc(&st, (size_t)N * cout * 16); hipMemset(st, 0, (size_t)N * cout * 16);
    hipMemcpy(xd, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    ConvWeightsH cw;
    if (conv_weights_upload_f16(w.data(), b.data(), cin, cin, cout, 2, &cw) != MI355_OK) return 1;
    ConvCallH c;
    c.in0 = xd; c.C0 = cin; c.N = N; c.Di = D; c.Hi = D; c.Wi = D; c.out = yd; c.act = ACT_LRELU; c.slope = 0.01f; c.stats = stats ? st : nullptr;
    const char *name = nullptr;
    if (conv3d_mfma_f16(cw, c, 0, &name) != MI355_OK) return 1;
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) conv3d_mfma_f16(cw, c, 0, &name);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = 2.0 * vout * cout * (double)cin * 27.0;
    const double bytes = 2.0 * (vin * cin + vout * cout);
    printf("%-52s N=%2d D=%3d %3d->%3d  %8.3f ms  %7.1f TFLOP/s = %.3f of 2500   %6.0f GB/s algorithmic\n", name, N, D, cin, cout, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 2500, bytes / ms / 1e6);
    fflush(stdout);
    conv_weights_free_f16(&cw); hipFree(xd); hipFree(yd); hipFree(st);
    return hipGetLastError() != hipSuccess;
}

int main() {
    // the stride-2 launches of bench config 3 (16 samples per forward): model B 64->128 @128^3, 128->256 @64^3, 256->512 @32^3;
    // model A 64->128 @64^3, 128->256 @32^3
    for (int stats = 0; stats < 2; ++stats) {
        if (run(16, 128, 64, 128, 3, stats)) return 1;
        if (run(16, 64, 128, 256, 5, stats)) return 1;
        if (run(16, 32, 256, 512, 10, stats)) return 1;
        if (run(16, 64, 64, 128, 10, stats)) return 1;
        if (run(16, 32, 128, 256, 10, stats)) return 1;
    }
    return 0;
}
