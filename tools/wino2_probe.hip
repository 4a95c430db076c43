// Diagnostic harness for conv3_f32_wino2_kernel: drives the kernel exactly as the network does (conv_weights_upload +
// conv3d_mfma_f32) on one layer shape, times it with HIP events and - built with -DMI355_W2_STAMPS - prints where wave 0
// of every workgroup spends its cycles (s_memtime sums per phase).  Not part of the product.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMI355_W2_STAMPS] -I<pkg>/csrc tools/wino2_probe.hip -o tools/wino2_probe
#include "conv3d.hip"


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

static int run(int N, int D, int cin, int cout, int reps, bool head = false) {
    const size_t vin = (size_t)N * D * D * D;
    std::vector<float> x(vin * cin), w((size_t)cout * cin * 27), b(cout);
    uint32_t sd = 12345u;
    auto u = [&]() { sd ^= sd << 13; sd ^= sd >> 17; sd ^= sd << 5; return (float)(int32_t)sd * (1.0f / 2147483648.0f); };  // [-1, 1)
    for (auto &v : x) v = u();
    for (auto &v : w) v = u() * 0.05f;
    for (auto &v : b) v = u();
    float *xd, *yd;
    hipMalloc(&xd, x.size() * 4); hipMalloc(&yd, vin * cout * 4);
    hipMemcpy(xd, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    ConvWeights cw;
    if (conv_weights_upload(w.data(), b.data(), cin, cin, cout, 1, false, &cw) != MI355_OK) return 1;
    ConvCall c;
    c.in0 = xd; c.C0 = cin; c.N = N; c.Di = D; c.Hi = D; c.Wi = D; c.out = yd; c.act = ACT_LRELU; c.slope = 0.01f;
    float *hw = nullptr, *hb = nullptr, *hout = nullptr;
    if (head) {  // the network's last decoder conv: fused 1x1x1 head, only the 3 logits are written
        std::vector<float> w3(3 * cout, 0.01f), b3(3, 0.1f);
        hipMalloc(&hw, w3.size() * 4); hipMalloc(&hb, 16); hipMalloc(&hout, vin * 3 * 4);
        hipMemcpy(hw, w3.data(), w3.size() * 4, hipMemcpyHostToDevice); hipMemcpy(hb, b3.data(), 12, hipMemcpyHostToDevice);
        c.head_w = hw; c.head_b = hb; c.head_out = hout; c.head_ncls = 3; c.out = nullptr;
    }
    const char *name = nullptr;
    if (conv3d_mfma_f32(cw, c, 0, &name) != MI355_OK) return 1;
    hipDeviceSynchronize();
#ifdef MI355_W2_STAMPS
    { std::vector<unsigned long long> z(1024 * 16, 0); hipMemcpyToSymbol(HIP_SYMBOL(w2_stamps), z.data(), z.size() * 8); }
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) conv3d_mfma_f32(cw, c, 0, &name);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = 2.0 * vin * cout * (double)cin * 27.0;
    printf("%-28s N=%d D=%d %3d->%3d  %8.3f ms  %7.1f TFLOP/s algorithmic  (%.1f executed = %.3f of 157.3)\n", name, N, D, cin, cout, ms,
           flops / ms / 1e9, flops / ms / 1e9 * 4 / 9, flops / ms / 1e9 * 4 / 9 / 157.3);
#ifdef MI355_W2_STAMPS
    std::vector<unsigned long long> st(1024 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(w2_stamps), st.size() * 8);
    double sum[16] = {0}; int wgs = 0;
    for (int g = 0; g < 1024; ++g) if (st[g * 16 + 5]) { ++wgs; for (int k = 0; k < 16; ++k) sum[k] += (double)st[g * 16 + k]; }
    const char *names[5] = {"chunk prologue", "step loop", "chunk drain+barrier", "  output transform (in epi)", "epilogue incl. transform"};
    printf("  stamps over %d workgroups x %d launches: kernel %.0f cycles per workgroup-launch, %.1f chunks, %.1f tiles\n", wgs, reps,
           sum[5] / wgs / reps, sum[6] / wgs / reps, sum[7] / wgs / reps);
    for (int k = 0; k < 5; ++k) printf("    %-28s %6.2f %%   (%8.0f cycles per %s)\n", names[k], 100.0 * sum[k] / sum[5],
                                       sum[k] / (k < 3 ? sum[6] : sum[7]), k < 3 ? "chunk" : "tile");
    printf("    %-28s %6.2f %%   (%8.0f cycles per tile)  [part of the epilogue]\n", "  barrier after epilogue", 100.0 * sum[8] / sum[5], sum[8] / sum[7]);
    printf("    %-28s %6.2f %%   (%8.0f cycles per tile)\n", "accumulator reset + set-up", 100.0 * sum[9] / sum[5], sum[9] / sum[7]);
    double acc = sum[9] - sum[3]; for (int k = 0; k < 5; ++k) acc += sum[k];
    printf("    %-28s %6.2f %%\n", "other (launch, first DMA)", 100.0 * (sum[5] - acc) / sum[5]);
    printf("    ideal step loop = 12 x 32 x 64 = 24576 cycles per chunk\n");
#endif
    conv_weights_free(&cw); hipFree(xd); hipFree(yd);
    return 0;
}

int main() {
    // the wino2 launches of bench config 2 (8 tiles of 128^3 batched): level 0 32->32 and 64->32, level 1 64->64, 128->64, level 2 128->128
    if (getenv("W2_PROBE_PMC")) {  // one launch per shape (reps = 1 -> 2 launches incl. warm-up): traffic calibration under rocprofv3 --pmc
        if (run(8, 128, 16, 32, 1)) return 1;
        if (run(8, 128, 32, 32, 1)) return 1;
        if (run(8, 64, 64, 64, 1)) return 1;
        return 0;
    }
    if (run(8, 128, 32, 32, 3)) return 1;
    if (run(8, 128, 32, 32, 3, true)) return 1;
    if (run(8, 128, 64, 32, 3)) return 1;
    if (run(8, 64, 64, 64, 5)) return 1;
    if (run(8, 64, 128, 64, 5)) return 1;
    if (run(8, 32, 128, 128, 10)) return 1;
    return 0;
}
