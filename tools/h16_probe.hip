// Diagnostic harness for conv3_f16_dma_kernel: drives the kernel exactly as the network does (conv_weights_upload_f16 +
// conv3d_mfma_f16) on one layer shape, times it with HIP events and - built with -DMI355_H16_STAMPS - prints where wave 0
// of every workgroup spends its cycles.  Not part of the product.
// Build: tools/build_probe.sh (h16_probe, h16_probe_stamps)
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

static int run(int N, int D, int cin, int cout, int reps, bool inaff = false, bool stats = false) {
    const size_t vin = (size_t)N * D * D * D;
    std::vector<_Float16> x(vin * cin);
    std::vector<float> w((size_t)cout * cin * 27), b(cout);
    uint32_t sd = 12345u;
    auto u = [&]() { sd ^= sd << 13; sd ^= sd >> 17; sd ^= sd << 5; return (float)(int32_t)sd * (1.0f / 2147483648.0f); };  // [-1, 1)
    for (auto &v : x) v = (_Float16)u();
    for (auto &v : w) v = u() * 0.05f;
    for (auto &v : b) v = u();
    _Float16 *xd, *yd;
    hipMalloc(&xd, x.size() * 2); hipMalloc(&yd, vin * cout * 2);
    hipMemcpy(xd, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    ConvWeightsH cw;
    if (conv_weights_upload_f16(w.data(), b.data(), cin, cin, cout, 1, &cw) != MI355_OK) return 1;
    ConvCallH c;
    c.in0 = xd; c.C0 = cin; c.N = N; c.Di = D; c.Hi = D; c.Wi = D; c.out = yd; c.act = ACT_LRELU; c.slope = 0.01f;
    float *sc = nullptr, *sh = nullptr; double *stat_buf = nullptr;
    if (inaff) {  // the producer's norm applied to the brick in LDS (INAFF instantiations)
        std::vector<float> hs((size_t)N * cin), hh((size_t)N * cin);
        for (auto &v : hs) v = 1.0f + 0.25f * u();
        for (auto &v : hh) v = 0.1f * u();
        hipMalloc(&sc, hs.size() * 4); hipMalloc(&sh, hh.size() * 4);
        hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(sh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
        c.in_scale = sc; c.in_shift = sh; c.in_act = ACT_LRELU;
    }
    if (stats) { hipMalloc(&stat_buf, (size_t)N * cout * 16); hipMemset(stat_buf, 0, (size_t)N * cout * 16); c.stats = stat_buf; }
    const char *name = nullptr;
    if (conv3d_mfma_f16(cw, c, 0, &name) != MI355_OK) return 1;
    hipDeviceSynchronize();
#ifdef MI355_H16_STAMPS
    { std::vector<unsigned long long> z(1024 * 16, 0); hipMemcpyToSymbol(HIP_SYMBOL(h16_stamps), z.data(), z.size() * 8); }
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) conv3d_mfma_f16(cw, c, 0, &name);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = 2.0 * vin * cout * (double)cin * 27.0;
    printf("%-40s N=%d D=%d %3d->%3d  %8.3f ms  %7.1f TFLOP/s = %.3f of 2500\n", name, N, D, cin, cout, ms, flops / ms / 1e9, flops / ms / 1e9 / 2500);
#ifdef MI355_H16_STAMPS
    std::vector<unsigned long long> st(1024 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(h16_stamps), st.size() * 8);
    double sum[16] = {0}; int wgs = 0;
    for (int g = 0; g < 1024; ++g) if (st[g * 16 + 5]) { ++wgs; for (int k = 0; k < 16; ++k) sum[k] += (double)st[g * 16 + k]; }
    const char *names[5] = {"taps 0-8 (DMA issue)", "taps 9-17", "taps 18-26", "drain + barrier", "epilogue"};
    printf("  stamps over %d workgroups x %d launches: kernel %.0f ticks per workgroup-launch, %.1f chunks, %.1f tiles\n", wgs, reps,
           sum[5] / wgs / reps, sum[6] / wgs / reps, sum[7] / wgs / reps);
    for (int k = 0; k < 5; ++k) printf("    %-28s %6.2f %%   (%8.0f ticks per %s)\n", names[k], 100.0 * sum[k] / sum[5], sum[k] / (k < 4 ? sum[6] : sum[7]), k < 4 ? "chunk" : "tile");
    printf("    %-28s %6.2f %%   (%8.0f ticks per tile)\n", "accumulator init + set-up", 100.0 * sum[8] / sum[5], sum[8] / sum[7]);
    double acc = sum[8]; for (int k = 0; k < 5; ++k) acc += sum[k];
    printf("    %-28s %6.2f %%\n", "other (launch, first DMA)", 100.0 * (sum[5] - acc) / sum[5]);
    printf("    (ticks: __builtin_readcyclecounter = s_memtime; ms * ticks/ms gives its rate)  kernel ticks / ms = %.0f\n", sum[5] / wgs / reps / ms);
#endif
    conv_weights_free_f16(&cw); hipFree(xd); hipFree(yd);
    return 0;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (argc > 1 && argv[1][0] == 'a') {  // ablation pairs: one Cout = 32 shape, one Cout = 64 shape (build with -DMI355_H16_ABL_DMA / -DMI355_H16_ABL_W)
        if (run(8, 128, 64, 32, 3)) return 1;
        if (run(8, 128, 64, 64, 3)) return 1;
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'c') {  // the Cout = 32 layers of the full-resolution level (conv3_f16_c32_kernel, round 5)
        if (run(8, 128, 32, 32, 3)) return 1;
        if (run(8, 128, 64, 32, 3)) return 1;
        if (run(8, 128, 64, 32, 3, true, true)) return 1;
        return 0;
    }
    // the same shape plain, with IN/GN statistics, and with the producer's norm applied in LDS (what INAFF costs, and where)
    if (run(8, 64, 64, 64, 5)) return 1;
    if (run(8, 64, 64, 64, 5, false, true)) return 1;
    if (run(8, 64, 64, 64, 5, true, true)) return 1;
    if (run(8, 64, 64, 64, 5, true, false)) return 1;
    return 0;
    // the Cout % 64 == 0 stride-1 launches of bench configs 2 / 3: level 1 64->64, 128->64, level 2 128->128, 256->128
    if (run(8, 64, 64, 64, 5)) return 1;
    if (run(8, 64, 128, 64, 5)) return 1;
    if (run(8, 32, 128, 128, 10)) return 1;
    if (run(8, 32, 256, 128, 10)) return 1;
    return 0;
}
