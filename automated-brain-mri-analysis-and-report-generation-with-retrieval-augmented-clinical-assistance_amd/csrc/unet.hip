// Network object, activation arena, forward pass and the sliding-window predictor (host side),
// plus the extern "C" entry points declared in include/mi355_nnunet.h.
//
// Reference behaviour restated here:
//   * Generic_UNet.forward                 model_architecture/generic_UNet.py:423-446
//   * eval-mode BatchNorm folded into the preceding conv (ConvDropoutNormNonlin :68-72)
//   * nnU-Net v1 SegmentationNetwork._internal_predict_3D_3Dconv_tiled / _compute_steps_for_
//     sliding_window / _get_gaussian / _internal_maybe_mirror_and_pred_3D (un-vendored upstream;
//     SURVEY.md 8a rows T1-T5) as driven by run_brats2021_inference_singlethread.py:97-128.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"

namespace mi355 {

static thread_local std::string g_last_error;
static thread_local std::string g_last_conv_kernel;  // mi355_last_conv_kernel(): which instantiation a single-op conv call ran

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

struct ConvLayer {
    ConvWeights w;     // MI355_F32
    ConvWeightsH wh;   // MI355_F16
    StemWeights stem;  // first conv of the net when Cin <= 4 (either dtype)
    bool is_stem = false;
    float *gamma_dev = nullptr, *beta_dev = nullptr;  // Instance/GroupNorm affine, or BN scale/shift (nonlin_first)
    bool runtime_norm = false;                        // statistics needed at run time (IN / GN)
    bool post_affine = false;                         // BN that could not be folded (nonlin_first)
    int cin = 0, cout = 0, stride = 1;
};

}  // namespace mi355

using namespace mi355;

struct mi355_unet {
    int in_channels = 0, cin_pad = 0, num_classes = 0, num_pool = 0;
    int norm = 0, num_groups = 0, nonlin_first = 0, dtype = 0;
    float eps = 1e-5f, slope = 0.01f;
    std::vector<std::vector<ConvLayer>> enc;  // num_pool + 1 stages
    std::vector<std::vector<ConvLayer>> dec;  // num_pool stages
    std::vector<TConvWeights> tu;
    std::vector<TConvWeightsH> tuh;
    HeadWeights head;
    // (the activation arena is not the handle's: one per LANE = per stream the caller launches on, shared by every handle that
    //  runs on that stream - device_scratch(SCR_ARENA, stream); a forward carries its pointer in its Plan)
    // gaussian importance map cache
    float *gauss_dev = nullptr;
    int gauss_p[3] = {0, 0, 0};
    int max_channels = 0;
    // optional per-kernel HIP-event timing (bench.py roofline)
    bool prof_on = false;
    struct ProfRec { std::string name; double flops, bytes; hipEvent_t a, b; };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;  // events are recycled: nothing is created inside a timed region after warm-up
};

namespace mi355 {

static std::mutex g_mu;
static int g_bound_device = -1;
static bool is_gfx950(int dev) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

int bind_device() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0) {
        set_error("no HIP device visible (%s); this library has no CPU fallback", e == hipSuccess ? "none current" : hipGetErrorString(e));
        return MI355_ERR_NO_DEVICE;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_bound_device < 0) {
        if (!is_gfx950(dev)) {
            set_error("HIP device %d is not gfx950 (MI355X); this library has no other code path", dev);
            return MI355_ERR_NO_DEVICE;
        }
        g_bound_device = dev;
    } else if (g_bound_device != dev) {
        set_error("this process is bound to HIP device %d but device %d is current: one process per GPU (weights, arena and "
                  "scratch live on the bound device)", g_bound_device, dev);
        return MI355_ERR_INVALID;
    }
    return MI355_OK;
}

// Lanes (round 5).  Scratch is per STREAM: the first MAX_LANES distinct streams the caller launches on get a lane of their own -
// activation arena, aggregation buffers, split-K partials, small reduction scratch - so that independent pieces of the work (the two
// ensemble members, or two halves of one member's (fold, tile) list: predictor.predict_folds(lanes = 2)) can be in flight on two
// streams at once: the HBM-bound kernels of one lane (norm passes, transposed convs, first layer, aggregation: a sixth of a
// config-3 step) then run beside the matrix-bound kernels of the other instead of in front of them.  Work on ONE stream is ordered
// by that stream, as before.  A stream beyond the table takes over the least recently used lane after a device synchronise.
// SCR_ZEROS / SCR_ZERO_BIAS are read-only once cleared and shared by all lanes.
constexpr int MAX_LANES = 4;
static struct { void *p; size_t bytes; } g_scratch[MAX_LANES][SCR_COUNT];
static struct { hipStream_t s; bool used; unsigned long tick; } g_lane[MAX_LANES];
static unsigned long g_lane_tick = 0;

static int lane_of_locked(hipStream_t s, int *lane) {
    int free_lane = -1, lru = 0;
    for (int i = 0; i < MAX_LANES; ++i) {
        if (g_lane[i].used && g_lane[i].s == s) { g_lane[i].tick = ++g_lane_tick; *lane = i; return MI355_OK; }
        if (!g_lane[i].used && free_lane < 0) free_lane = i;
        if (g_lane[i].used && g_lane[i].tick < g_lane[lru].tick) lru = i;
    }
    if (free_lane < 0) {
        MI355_HIP(hipDeviceSynchronize());  // whatever still runs on the evicted stream's scratch
        free_lane = lru;
    }
    g_lane[free_lane].s = s; g_lane[free_lane].used = true; g_lane[free_lane].tick = ++g_lane_tick;
    *lane = free_lane;
    return MI355_OK;
}

int device_scratch(int slot, hipStream_t stream, size_t bytes, void **out, bool zeroed) {
    MI355_REQUIRE(slot >= 0 && slot < SCR_COUNT && out, "bad scratch slot %d", slot);
    MI355_TRY(bind_device());
    std::lock_guard<std::mutex> lk(g_mu);
    int lane = 0;
    if (slot != SCR_ZEROS && slot != SCR_ZERO_BIAS) MI355_TRY(lane_of_locked(stream, &lane));
    auto &b = g_scratch[lane][slot];
    if (b.bytes < bytes) {
        if (b.p) {
            MI355_HIP(hipDeviceSynchronize());  // work in flight may still use the old buffer
            MI355_HIP(hipFree(b.p));
            b.p = nullptr; b.bytes = 0;
        }
        MI355_HIP(hipMalloc(&b.p, bytes));
        b.bytes = bytes;
        if (zeroed) MI355_HIP(hipMemset(b.p, 0, bytes));
    }
    *out = b.p;
    return MI355_OK;
}

static int require_device() { return bind_device(); }

static int upload(const float *host, size_t n, float **dev) {
    MI355_HIP(hipMalloc(dev, n * sizeof(float)));
    MI355_HIP(hipMemcpy(*dev, host, n * sizeof(float), hipMemcpyHostToDevice));
    return MI355_OK;
}

static int build_conv(const mi355_unet &net, const mi355_conv_desc &d, int cin_phys, ConvLayer *out, bool stem = false) {
    MI355_REQUIRE(d.weight != nullptr, "conv %d->%d: null weight", d.cin, d.cout);
    MI355_REQUIRE(d.cin > 0 && d.cout > 0 && cin_phys >= d.cin, "conv: bad channel counts %d->%d (phys %d)", d.cin, d.cout, cin_phys);
    ConvLayer L;
    L.cin = d.cin; L.cout = d.cout; L.stride = d.stride;
    const size_t wn = (size_t)d.cout * d.cin * 27;
    std::vector<float> w(d.weight, d.weight + wn);
    std::vector<float> b(d.cout, 0.f);
    if (d.bias) b.assign(d.bias, d.bias + d.cout);
    if (net.norm == MI355_NORM_BATCH) {
        MI355_REQUIRE(d.running_mean && d.running_var, "BatchNorm conv %d->%d: running stats missing", d.cin, d.cout);
        std::vector<float> sc(d.cout), sh(d.cout);
        for (int co = 0; co < d.cout; ++co) {
            const double g = d.gamma ? d.gamma[co] : 1.0, be = d.beta ? d.beta[co] : 0.0;
            const double s = g / std::sqrt((double)d.running_var[co] + (double)net.eps);
            sc[co] = (float)s;
            sh[co] = (float)(be - (double)d.running_mean[co] * s);
        }
        if (!net.nonlin_first) {
            // y = lrelu(BN(conv(x)+b)) = lrelu(conv'(x) + b')  with w' = w*s, b' = b*s + shift
            for (int co = 0; co < d.cout; ++co) {
                const double s = sc[co];
                for (size_t k = 0; k < (size_t)d.cin * 27; ++k)
                    w[(size_t)co * d.cin * 27 + k] = (float)((double)w[(size_t)co * d.cin * 27 + k] * s);
                b[co] = (float)((double)b[co] * s + (double)sh[co]);
            }
        } else {
            L.post_affine = true;
            MI355_TRY(upload(sc.data(), d.cout, &L.gamma_dev));
            MI355_TRY(upload(sh.data(), d.cout, &L.beta_dev));
        }
    } else if (net.norm == MI355_NORM_INSTANCE || net.norm == MI355_NORM_GROUP) {
        L.runtime_norm = true;
        std::vector<float> g(d.cout, 1.f), be(d.cout, 0.f);
        if (d.gamma) g.assign(d.gamma, d.gamma + d.cout);
        if (d.beta) be.assign(d.beta, d.beta + d.cout);
        MI355_TRY(upload(g.data(), d.cout, &L.gamma_dev));
        MI355_TRY(upload(be.data(), d.cout, &L.beta_dev));
    }
    if (stem) { L.is_stem = true; MI355_TRY(stem_weights_upload(w.data(), b.data(), d.cin, d.cout, net.dtype, &L.stem)); }
    else if (net.dtype == MI355_F16) MI355_TRY(conv_weights_upload_f16(w.data(), b.data(), d.cin, cin_phys, d.cout, d.stride, &L.wh));
    else MI355_TRY(conv_weights_upload(w.data(), b.data(), d.cin, cin_phys, d.cout, d.stride, false, &L.w));
    *out = L;
    return MI355_OK;
}

static void free_conv(ConvLayer *L) {
    conv_weights_free(&L->w);
    conv_weights_free_f16(&L->wh);
    stem_weights_free(&L->stem);
    if (L->gamma_dev) (void)hipFree(L->gamma_dev);
    if (L->beta_dev) (void)hipFree(L->beta_dev);
}

static void destroy(mi355_unet *net) {
    if (!net) return;
    for (auto &st : net->enc) for (auto &L : st) free_conv(&L);
    for (auto &st : net->dec) for (auto &L : st) free_conv(&L);
    for (auto &t : net->tu) tconv_weights_free(&t);
    for (auto &t : net->tuh) tconv_weights_free_f16(&t);
    head_weights_free(&net->head);
    if (net->gauss_dev) (void)hipFree(net->gauss_dev);
    for (auto &r : net->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : net->event_pool) (void)hipEventDestroy(e);
    delete net;
}

// ---- arena layout for one (N, D, H, W): per level four activation buffers + norm scratch
struct Plan {
    std::vector<int64_t> vox;          // voxels per sample at each level
    std::vector<int> maxc;             // widest tensor at each level
    std::vector<size_t> off[4];        // byte offsets of buffers A, B, U, T per level
    size_t stats_off = 0, scale_off = 0, shift_off = 0, x0_off = 0, total = 0;
    size_t scale2_off = 0, shift2_off = 0;  // second scale / shift pair: a block whose normalisation is applied by its consumer
    size_t stats_bytes = 0;
    char *arena = nullptr;             // the activation arena of the lane (stream) this forward runs on, set by ensure_arena
};

static int make_plan(const mi355_unet &net, int N, int D, int H, int W, Plan *pl) {
    const int np = net.num_pool;
    MI355_REQUIRE(D % (1 << np) == 0 && H % (1 << np) == 0 && W % (1 << np) == 0,
                  "patch %dx%dx%d must be divisible by %d (generic_UNet.py:256)", D, H, W, 1 << np);
    pl->vox.resize(np + 1);
    pl->maxc.assign(np + 1, 0);
    for (int l = 0; l <= np; ++l) {
        pl->vox[l] = (int64_t)(D >> l) * (H >> l) * (W >> l);
        for (auto &L : net.enc[l]) pl->maxc[l] = std::max(pl->maxc[l], L.cout);
    }
    for (int u = 0; u < np; ++u) {
        const int l = np - 1 - u;
        pl->maxc[l] = std::max(pl->maxc[l], net.dtype == MI355_F16 ? net.tuh[u].cout : net.tu[u].cout);
        for (auto &L : net.dec[u]) pl->maxc[l] = std::max(pl->maxc[l], L.cout);
    }
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~size_t(255); return r; };
    const size_t es = net.dtype == MI355_F16 ? 2 : 4;
    pl->x0_off = take((size_t)N * pl->vox[0] * net.cin_pad * es);
    for (int k = 0; k < 4; ++k) pl->off[k].resize(np + 1);
    for (int l = 0; l <= np; ++l)
        for (int k = 0; k < 4; ++k)
            pl->off[k][l] = take((size_t)N * pl->vox[l] * pl->maxc[l] * es);
    pl->stats_bytes = (size_t)N * net.max_channels * 2 * sizeof(double);
    pl->stats_off = take(pl->stats_bytes);
    pl->scale_off = take((size_t)N * net.max_channels * sizeof(float));
    pl->shift_off = take((size_t)N * net.max_channels * sizeof(float));
    pl->scale2_off = take((size_t)N * net.max_channels * sizeof(float));
    pl->shift2_off = take((size_t)N * net.max_channels * sizeof(float));
    pl->total = o;
    return MI355_OK;
}

static int ensure_arena(Plan *pl, hipStream_t s) {
    void *p = nullptr;
    MI355_TRY(device_scratch(SCR_ARENA, s, pl->total, &p));
    pl->arena = (char *)p;
    return MI355_OK;
}

struct ProfScope {
    mi355_unet *net; hipStream_t s; bool on; size_t idx;
    ProfScope(mi355_unet *n, hipStream_t st, const std::string &name, double flops, double bytes) : net(n), s(st), on(n->prof_on), idx(0) {
        if (!on) return;
        mi355_unet::ProfRec r; r.name = name; r.flops = flops; r.bytes = bytes;
        auto take = [&](hipEvent_t *e) {
            if (!net->event_pool.empty()) { *e = net->event_pool.back(); net->event_pool.pop_back(); return true; }
            return hipEventCreate(e) == hipSuccess;
        };
        if (!take(&r.a) || !take(&r.b)) { on = false; return; }
        (void)hipEventRecord(r.a, s);
        idx = net->prof.size();
        net->prof.push_back(r);
    }
    void rename(const char *name) { if (on && name) net->prof[idx].name = name; }
    ~ProfScope() { if (on) (void)hipEventRecord(net->prof[idx].b, s); }
};

static std::string conv_kernel_name(const ConvWeights &w) {
    if (!w.wp_dev) return "conv3_direct_kernel";
    char buf[64];
    if (w.pipe) snprintf(buf, sizeof(buf), "conv3_f32_mfma_pipe_kernel<*, %d>", w.nf);  // MF (4|2) is chosen per launch
    else snprintf(buf, sizeof(buf), "conv3_f32_mfma_kernel<%d, %d, %d, %d>", w.stride, w.cc, w.stride == 1 ? 2 : 1, w.nf);
    return buf;
}

// One ConvDropoutNormNonlin / ConvDropoutNonlinNorm block.
// `defer_norm`: the block's normalisation (+ activation) is NOT applied to `out`; its per-(sample, channel) scale / shift go
// to the plan's second pair and the NEXT block applies them while staging its input (`in_norm` of that call).
static int run_block(mi355_unet *net, const Plan &pl, const ConvLayer &L, const void *in0, int C0,
                     const void *in1, int C1, int N, int Di, int Hi, int Wi, void *out, hipStream_t s,
                     float *head_logits_out = nullptr, bool defer_norm = false, bool in_norm = false) {
    const bool f16 = net->dtype == MI355_F16;
    double *stats = (double *)(pl.arena + pl.stats_off);
    float *scale = (float *)(pl.arena + (defer_norm ? pl.scale2_off : pl.scale_off)), *shift = (float *)(pl.arena + (defer_norm ? pl.shift2_off : pl.shift_off));
    int act = ACT_LRELU;
    double *stats_arg = nullptr;
    if (L.runtime_norm) {
        act = net->nonlin_first ? ACT_LRELU : ACT_NONE;
        stats_arg = stats;
        MI355_HIP(hipMemsetAsync(stats, 0, (size_t)N * L.cout * 2 * sizeof(double), s));
    }
    const int st = L.stride;
    const int64_t Vo = (int64_t)((Di - 1) / st + 1) * ((Hi - 1) / st + 1) * ((Wi - 1) / st + 1);
    {
        // algorithmic work of this launch: 2*MAC over the LOGICAL cin; input read once + output written once + weights
        const double es = f16 ? 2.0 : 4.0;
        const double flops = 2.0 * N * Vo * L.cout * (double)L.cin * 27.0;
        const double bytes = es * ((double)N * Di * Hi * Wi * (C0 + C1) + (double)N * Vo * L.cout + (double)L.cout * L.cin * 27.0);
        if (L.is_stem) {
            ProfScope ps(net, s, f16 ? "conv3_stem_f16_kernel" : "conv3_stem_f32_kernel", flops, bytes);
            MI355_TRY(conv3d_stem(L.stem, in0, N, Di, Hi, Wi, out, stats_arg, act, net->slope, s));
        } else if (f16) {
            ConvCallH c;
            c.in0 = (const _Float16 *)in0; c.in1 = (const _Float16 *)in1; c.C0 = C0; c.C1 = C1;
            c.N = N; c.Di = Di; c.Hi = Hi; c.Wi = Wi; c.out = (_Float16 *)out; c.slope = net->slope;
            c.act = act; c.stats = stats_arg;
            if (head_logits_out) { c.head_w = net->head.w_dev; c.head_b = net->head.b_dev; c.head_ncls = net->head.ncls; c.head_out = head_logits_out; }
            if (in_norm) {  // in0 is the previous block's raw conv output: normalise (+ LeakyReLU) while staging
                c.in_scale = (const float *)(pl.arena + pl.scale2_off); c.in_shift = (const float *)(pl.arena + pl.shift2_off);
                c.in_act = net->nonlin_first ? ACT_NONE : ACT_LRELU;
            }
            const char *kname = nullptr;
            ProfScope ps(net, s, "conv3_f16", flops, bytes);
            MI355_TRY(conv3d_mfma_f16(L.wh, c, s, &kname));
            ps.rename(kname);
        } else {
            ConvCall c;
            c.in0 = (const float *)in0; c.in1 = (const float *)in1; c.C0 = C0; c.C1 = C1;
            c.N = N; c.Di = Di; c.Hi = Hi; c.Wi = Wi; c.out = (float *)out; c.slope = net->slope;
            c.act = act; c.stats = stats_arg;
            if (head_logits_out) { c.head_w = net->head.w_dev; c.head_b = net->head.b_dev; c.head_ncls = net->head.ncls; c.head_out = head_logits_out; }
            if (in_norm) {  // in0 is the previous block's raw conv output: the F(2x2x2,3x3x3) kernel normalises (+ LeakyReLU) its brick in LDS
                c.in_scale = (const float *)(pl.arena + pl.scale2_off); c.in_shift = (const float *)(pl.arena + pl.shift2_off);
                c.in_act = net->nonlin_first ? ACT_NONE : ACT_LRELU;
            }
            const char *kname = nullptr;
            ProfScope ps(net, s, conv_kernel_name(L.w), flops, bytes);
            if (L.w.wp_dev) MI355_TRY(conv3d_mfma_f32(L.w, c, s, &kname));
            else MI355_TRY(conv3d_direct_f32(L.w, c, s));
            ps.rename(kname);
        }
    }
    const double es = f16 ? 2.0 : 4.0;
    if (L.runtime_norm) {
        MI355_TRY(norm_finalize(stats, N, L.cout, Vo, net->norm, net->num_groups, net->eps, L.gamma_dev, L.beta_dev,
                                scale, shift, s));
        if (defer_norm) return MI355_OK;  // the consumer applies scale / shift (and the activation) in its staging
        ProfScope ps(net, s, f16 ? "norm_apply_kernel<f16>" : "norm_apply_kernel<f32>", 2.0 * N * Vo * L.cout, 2.0 * es * N * Vo * L.cout);
        MI355_TRY(norm_apply(out, net->dtype, N, Vo, L.cout, scale, shift, net->nonlin_first ? ACT_NONE : ACT_LRELU, net->slope, s));
    } else if (L.post_affine) {
        // BN after the nonlinearity: per-channel affine, identical for every sample
        for (int n = 0; n < N; ++n)
            MI355_TRY(norm_apply((char *)out + (size_t)n * Vo * L.cout * (size_t)es, net->dtype, 1, Vo, L.cout, L.gamma_dev, L.beta_dev,
                                 ACT_NONE, net->slope, s));
    }
    return MI355_OK;
}

// Can block `L` (run-time Instance/GroupNorm) leave its normalisation to the next block `Ln` of the same stage?
// generic_UNet.py:62-72 is one expression, lrelu(instnorm(conv(x))).  fp16: the consumer must be a kernel that normalises while
// staging (conv3d_f16_fuses_input_norm: register-staged pipelined kernel or the LDS-DMA kernel).  fp32 (round 4): the consumer
// must be a launch of the F(2x2x2,3x3x3) kernel, which normalises its brick in LDS (conv3d_wino3_fuses_input_norm).
static bool can_defer_norm(const mi355_unet *net, const ConvLayer &L, const ConvLayer &Ln, int N, int Dl, int Hl, int Wl) {
    if (!L.runtime_norm || Ln.stride != 1 || Ln.is_stem || Ln.cin != L.cout) return false;
    if (net->dtype == MI355_F16) {
        if ((!L.is_stem && !L.wh.wp_dev) || !Ln.wh.wp_dev) return false;
        ConvCallH c;
        c.C0 = L.cout; c.C1 = 0; c.N = N; c.Di = Dl; c.Hi = Hl; c.Wi = Wl;
        c.stats = Ln.runtime_norm ? (double *)1 : nullptr;  // (only tested for null)
        return conv3d_f16_fuses_input_norm(Ln.wh, c);
    }
    if (!Ln.w.wp3_dev) return false;
    ConvCall c;
    c.C0 = L.cout; c.C1 = 0; c.N = N; c.Di = Dl; c.Hi = Hl; c.Wi = Wl;
    c.stats = Ln.runtime_norm ? (double *)1 : nullptr;
    return conv3d_wino3_fuses_input_norm(Ln.w, c);
}

// x0: [N,D,H,W,cin_pad] already in the arena at pl.x0_off.  Returns the last decoder feature map.
// If the last decoder block has no run-time normalisation (BN folded / no norm) and its Cout fits one workgroup, the
// 1x1x1 head is fused into its epilogue: *is_logits = true and *feat points at fp32 logits [N][ncls][V] (written to
// logits_target when given, else into the arena); the 32-channel feature map is then never written nor re-read.
// head_norm (round 3): when non-null and the last decoder block carries a run-time Instance/GroupNorm, that block's
// normalisation (+ activation) is NOT applied to the returned feature map: *head_norm receives its scale / shift and the
// caller's head kernel applies them while reading the features (head_logits / head_aggregate take a FeatNorm).
static int forward_features(mi355_unet *net, const Plan &pl, int N, int D, int H, int W, const void **feat,
                            int *feat_c, hipStream_t s, bool *is_logits = nullptr, float *logits_target = nullptr,
                            FeatNorm *head_norm = nullptr) {
    const int np = net->num_pool;
    const bool f16 = net->dtype == MI355_F16;
    auto buf = [&](int k, int l) { return (void *)(pl.arena + pl.off[k][l]); };
    const void *cur = (const void *)(pl.arena + pl.x0_off);
    int curC = net->cin_pad;
    std::vector<const void *> skip(np);
    std::vector<int> skipC(np);
    // encoder + bottleneck
    for (int l = 0; l <= np; ++l) {
        int Di = D >> l, Hi = H >> l, Wi = W >> l;
        bool pending = false;  // cur holds a raw conv output whose normalisation the next block applies
        for (size_t i = 0; i < net->enc[l].size(); ++i) {
            const ConvLayer &L = net->enc[l][i];
            int inD = Di, inH = Hi, inW = Wi;
            if (L.stride == 2) { inD = Di * 2; inH = Hi * 2; inW = Wi * 2; }
            void *out = buf((int)(i & 1), l);
            const bool defer = i + 1 < net->enc[l].size() && can_defer_norm(net, L, net->enc[l][i + 1], N, Di, Hi, Wi);
            MI355_TRY(run_block(net, pl, L, cur, curC, nullptr, 0, N, inD, inH, inW, out, s, nullptr, defer, pending));
            pending = defer;
            cur = out; curC = L.cout;
        }
        if (l < np) { skip[l] = cur; skipC[l] = curC; }
    }
    // decoder
    for (int u = 0; u < np; ++u) {
        const int l = np - 1 - u;
        const int Dl = D >> l, Hl = H >> l, Wl = W >> l;
        void *up = buf(2, l);
        const int tcin = f16 ? net->tuh[u].cin : net->tu[u].cin, tcout = f16 ? net->tuh[u].cout : net->tu[u].cout;
        MI355_REQUIRE(tcin == curC, "tu.%d expects %d channels, got %d", u, tcin, curC);
        {
            const double es = f16 ? 2.0 : 4.0;
            const double vin = (double)N * (Dl / 2) * (Hl / 2) * (Wl / 2);
            ProfScope ps(net, s, f16 ? "tconv2_f16_mfma_v2_kernel" : "tconv2_f32_mfma_v2_kernel", 2.0 * vin * tcin * tcout * 8.0,
                         es * (vin * tcin + 8.0 * vin * tcout + 8.0 * tcin * tcout));
            const char *tname = nullptr;
            if (f16) { MI355_TRY(tconv2_mfma_f16(net->tuh[u], (const _Float16 *)cur, N, Dl / 2, Hl / 2, Wl / 2, (_Float16 *)up, s, &tname)); ps.rename(tname); }
            else { MI355_TRY(tconv2_mfma_f32(net->tu[u], (const float *)cur, N, Dl / 2, Hl / 2, Wl / 2, (float *)up, s, &tname)); ps.rename(tname); }
        }
        // concat order (upsampled, skip): generic_UNet.py:438 - never materialised
        const void *in0 = up, *in1 = skip[l];
        int C0 = tcout, C1 = skipC[l];
        // outputs alternate between T and whichever of A/B is not the skip
        void *freeAB = (skip[l] == buf(0, l)) ? buf(1, l) : buf(0, l);
        bool pending_d = false;
        for (size_t i = 0; i < net->dec[u].size(); ++i) {
            const ConvLayer &L = net->dec[u][i];
            void *out = (i & 1) ? freeAB : buf(3, l);
            static int fuse = -1;
            if (fuse < 0) {
                const char *e = getenv("MI355_FUSE_HEAD"), *ci = getenv("MI355_CONV_IMPL");
                fuse = (e && e[0] == '0') ? 0 : 1;
                if (f16 && ci && ci[0] == '0') fuse = 0;  // the fp16 fused epilogue exists in the pipelined kernel only
            }
            const bool last = is_logits && (u == np - 1) && (i + 1 == net->dec[u].size());
            const int lnf = f16 ? L.wh.nf : L.w.nf;
            const bool has_pack = f16 ? (L.wh.wp_dev != nullptr) : (L.w.wp_dev != nullptr);
            if (last && fuse && !L.runtime_norm && !L.post_affine && has_pack && L.cout == 32 * lnf && net->head.ncls <= 4 &&
                net->head.cin == L.cout && (!f16 || lnf == 1)) {
                float *lg = logits_target ? logits_target : (float *)out;
                MI355_TRY(run_block(net, pl, L, in0, C0, in1, C1, N, Dl, Hl, Wl, nullptr, s, lg, false, pending_d));  // (a pending norm is refused loudly by the fused-head conv: ADVICE r2)
                *is_logits = true;
                *feat = lg; *feat_c = net->head.ncls;
                return MI355_OK;
            }
            static int fuse_norm = -1;
            if (fuse_norm < 0) { const char *e = getenv("MI355_FUSE_NORM"); fuse_norm = (e && e[0] == '0') ? 0 : 1; }
            const bool to_head = fuse_norm && head_norm && L.runtime_norm && (u == np - 1) && (i + 1 == net->dec[u].size());
            const bool defer = to_head || (i + 1 < net->dec[u].size() && can_defer_norm(net, L, net->dec[u][i + 1], N, Dl, Hl, Wl));
            MI355_TRY(run_block(net, pl, L, in0, C0, in1, C1, N, Dl, Hl, Wl, out, s, nullptr, defer, pending_d));
            if (to_head) {
                head_norm->scale = (const float *)(pl.arena + pl.scale2_off);
                head_norm->shift = (const float *)(pl.arena + pl.shift2_off);
                head_norm->slope = net->nonlin_first ? 1.0f : net->slope;  // ConvDropoutNonlinNorm: the activation came before the norm
            }
            pending_d = defer;
            in0 = out; C0 = L.cout; in1 = nullptr; C1 = 0;
        }
        cur = in0; curC = C0;
    }
    if (is_logits) *is_logits = false;
    *feat = cur; *feat_c = curC;
    return MI355_OK;
}

// ---- sliding-window helpers (nnU-Net v1)
static std::vector<int> compute_steps(int patch, int image, double step_size) {
    // target = patch*step ; n = ceil((image-patch)/target)+1 ; actual = (image-patch)/(n-1)
    const double target = patch * step_size;
    const int n = (int)std::ceil((image - patch) / target) + 1;
    const int max_step = image - patch;
    const double actual = n > 1 ? (double)max_step / (n - 1) : 99999999999.0;
    std::vector<int> steps(n);
    for (int i = 0; i < n; ++i) steps[i] = (int)std::nearbyint(actual * i);  // np.round: half to even
    return steps;
}

// scipy.ndimage.gaussian_filter(delta at patch//2, sigma = patch*sigma_scale, mode='constant') is
// separable: the filtered delta is the outer product of the three normalised 1-D kernels
// (truncate=4.0 -> radius int(4*sigma+0.5)); then /max, fp32, zeros -> smallest non-zero.
static void gaussian_map(const int p[3], double sigma_scale, std::vector<float> &out) {
    std::vector<double> ax[3];
    for (int a = 0; a < 3; ++a) {
        const double sigma = p[a] * sigma_scale;
        const int radius = (int)(4.0 * sigma + 0.5);
        double sum = 0.0;
        std::vector<double> k(2 * radius + 1);
        for (int i = -radius; i <= radius; ++i) { k[i + radius] = std::exp(-0.5 / (sigma * sigma) * (double)i * i); sum += k[i + radius]; }
        ax[a].assign(p[a], 0.0);
        const int c = p[a] / 2;
        for (int i = 0; i < p[a]; ++i) {
            const int o = i - c;
            if (o >= -radius && o <= radius) ax[a][i] = k[o + radius] / sum;
        }
    }
    const size_t n = (size_t)p[0] * p[1] * p[2];
    std::vector<double> g(n);
    double mx = 0.0;
    for (int z = 0; z < p[0]; ++z)
        for (int y = 0; y < p[1]; ++y)
            for (int x = 0; x < p[2]; ++x) {
                const double v = (ax[0][z] * ax[1][y]) * ax[2][x];
                g[((size_t)z * p[1] + y) * p[2] + x] = v;
                if (v > mx) mx = v;
            }
    out.resize(n);
    float mn = INFINITY;
    for (size_t i = 0; i < n; ++i) {
        out[i] = (float)(g[i] / mx * 1.0);
        if (out[i] != 0.f && out[i] < mn) mn = out[i];
    }
    for (size_t i = 0; i < n; ++i)
        if (out[i] == 0.f) out[i] = mn;
}

static int ensure_gaussian(mi355_unet *net, const int p[3]) {
    std::lock_guard<std::mutex> lk(g_mu);  // (one handle may be driven from two lanes)
    if (net->gauss_dev && net->gauss_p[0] == p[0] && net->gauss_p[1] == p[1] && net->gauss_p[2] == p[2])
        return MI355_OK;
    if (net->gauss_dev) { MI355_HIP(hipDeviceSynchronize()); MI355_HIP(hipFree(net->gauss_dev)); net->gauss_dev = nullptr; }
    std::vector<float> g;
    gaussian_map(p, 1.0 / 8.0, g);
    MI355_TRY(upload(g.data(), g.size(), &net->gauss_dev));
    net->gauss_p[0] = p[0]; net->gauss_p[1] = p[1]; net->gauss_p[2] = p[2];
    return MI355_OK;
}

struct SwGeom {
    int P[3], Zp[3], pad_lo[3];
    std::vector<TileDesc> tiles;   // origin of every tile, loop order axis0 outer .. axis2 inner
    std::vector<int> mirrors;      // TileDesc-style masks in nnU-Net's evaluation order
};

static int make_geom(const mi355_sw_opts &o, int Z, int Y, int X, SwGeom *g) {
    const int dims[3] = {Z, Y, X};
    std::vector<int> steps[3];
    for (int a = 0; a < 3; ++a) {
        g->P[a] = o.patch[a];
        MI355_REQUIRE(g->P[a] > 0 && dims[a] > 0, "bad patch / volume size");
        g->Zp[a] = std::max(dims[a], g->P[a]);       // pad_nd_image(..., "constant", 0)
        g->pad_lo[a] = (g->Zp[a] - dims[a]) / 2;     // pad_below = difference // 2
        steps[a] = compute_steps(g->P[a], g->Zp[a], o.step_size);
    }
    MI355_REQUIRE(o.step_size > 0.f && o.step_size <= 1.f, "step_size must be in (0, 1]");
    g->tiles.clear();
    for (int z : steps[0]) for (int y : steps[1]) for (int x : steps[2]) g->tiles.push_back(TileDesc{z, y, x, 0});
    // _internal_maybe_mirror_and_pred_3D: m = 0..7; bit0 of m flips the LAST axis (x), bit1 y, bit2 z
    g->mirrors.clear();
    for (int m = 0; m < 8; ++m) {
        const bool fx = m & 1, fy = m & 2, fz = m & 4;
        if ((fx && !(o.mirror_axes & 4)) || (fy && !(o.mirror_axes & 2)) || (fz && !(o.mirror_axes & 1))) continue;
        g->mirrors.push_back((fz ? 1 : 0) | (fy ? 2 : 0) | (fx ? 4 : 0));
    }
    return MI355_OK;
}

// Evaluates the tiles with (index % world) == rank of one net; adds into agg (and cnt if non-null,
// for ALL tiles so that every rank holds the full normaliser).
// `first_item`: index of this net's tile 0 in the caller's work list (fold f of a fold list: f * tiles): the work items
// (fold, tile) are dealt round-robin over the ranks as ONE list, so 5 folds x 8 tiles on 3 ranks still balance.
static int sw_accumulate(mi355_unet *net, const float *vol, int Z, int Y, int X, const mi355_sw_opts &o,
                         const SwGeom &g, int rank, int world, float *agg, float *cnt, hipStream_t s, long first_item = 0) {
    const int nm = (int)g.mirrors.size();
    const bool use_gauss = o.use_gaussian && g.tiles.size() > 1;
    if (use_gauss) MI355_TRY(ensure_gaussian(net, g.P));
    std::vector<int> mine;
    for (size_t t = 0; t < g.tiles.size(); ++t) if ((int)((first_item + (long)t) % world) == rank) mine.push_back((int)t);
    // samples per forward: 16 (fp32) / 32 (fp16: half the bytes per sample; config 3 fp16 297 -> 293 ms per volume with 32, the
    // deep levels' launches fill the chip better) - the arena is sized for 288 GB of HBM, not for a few GB
    int bt = o.batch_tiles > 0 ? o.batch_tiles : std::max(1, (net->dtype == MI355_F16 ? 32 : 16) / nm);
    if (bt * nm > 64) bt = std::max(1, 64 / nm);
    MI355_REQUIRE(nm <= 64, "too many mirrors");
    Plan pl;
    MI355_TRY(make_plan(*net, bt * nm, g.P[0], g.P[1], g.P[2], &pl));
    MI355_TRY(ensure_arena(&pl, s));
    if (mine.empty()) return MI355_OK;
    for (size_t b0 = 0; b0 < mine.size(); b0 += bt) {
        const int nb = (int)std::min<size_t>(bt, mine.size() - b0);
        std::vector<TileDesc> samples;
        for (int i = 0; i < nb; ++i)
            for (int m = 0; m < nm; ++m) {
                TileDesc td = g.tiles[mine[b0 + i]];
                td.mirror = g.mirrors[m];
                samples.push_back(td);
            }
        {
        const double pv = (double)samples.size() * g.P[0] * g.P[1] * g.P[2];
        ProfScope ps(net, s, "extract_tiles_kernel", 0.0, pv * (4.0 * net->in_channels + (net->dtype == MI355_F16 ? 2.0 : 4.0) * net->cin_pad));
        MI355_TRY(extract_tiles(vol, net->in_channels, Z, Y, X, g.pad_lo[0], g.pad_lo[1], g.pad_lo[2], samples.data(),
                                (int)samples.size(), g.P[0], g.P[1], g.P[2], net->cin_pad,
                                (void *)(pl.arena + pl.x0_off), net->dtype, s));
        }
        const void *feat; int fc; bool is_logits = false;
        FeatNorm head_norm;
        MI355_TRY(forward_features(net, pl, (int)samples.size(), g.P[0], g.P[1], g.P[2], &feat, &fc, s, &is_logits, nullptr, &head_norm));
        MI355_REQUIRE(is_logits || fc == net->head.cin, "head expects %d channels, decoder gives %d", net->head.cin, fc);
        for (int i = 0; i < nb; ++i) {
            const TileDesc &td = g.tiles[mine[b0 + i]];
            const double pv = (double)g.P[0] * g.P[1] * g.P[2];
            if (is_logits) {
                ProfScope ps(net, s, "logits_aggregate_kernel", 0.0, 4.0 * pv * (nm * net->num_classes + 2.0 * net->num_classes + 3.0));
                MI355_TRY(logits_aggregate((const float *)feat, net->num_classes, i * nm, g.mirrors.data(), nm, g.P[0], g.P[1], g.P[2],
                                           o.nonlin, use_gauss ? net->gauss_dev : nullptr, agg, (cnt && world == 1) ? cnt : nullptr,
                                           g.Zp[0], g.Zp[1], g.Zp[2], td.z0, td.y0, td.x0, s));
                continue;
            }
            ProfScope ps(net, s, "head_aggregate_kernel", 2.0 * pv * nm * fc * net->num_classes,
                         pv * ((net->dtype == MI355_F16 ? 2.0 : 4.0) * nm * fc + 4.0 * (2.0 * net->num_classes + 3.0)));
            MI355_TRY(head_aggregate(net->head, feat, net->dtype, i * nm, g.mirrors.data(), nm, g.P[0], g.P[1], g.P[2], o.nonlin,
                                     use_gauss ? net->gauss_dev : nullptr, agg, (cnt && world == 1) ? cnt : nullptr,
                                     g.Zp[0], g.Zp[1], g.Zp[2], td.z0, td.y0, td.x0, s, head_norm));
        }
    }
    return MI355_OK;
}

}  // namespace mi355

// =============================================================================== C ABI
extern "C" const char *mi355_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *mi355_last_conv_kernel(void) { return g_last_conv_kernel.c_str(); }
extern "C" int mi355_version(void) { return 100; }
extern "C" int mi355_device_count(void) {
    int n = 0, good = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    for (int d = 0; d < n; ++d) good += is_gfx950(d) ? 1 : 0;
    return good;
}

extern "C" int mi355_unet_create(const mi355_unet_desc *d, mi355_unet_t *out) {
    MI355_REQUIRE(d && out, "null argument");
    MI355_TRY(require_device());
    MI355_REQUIRE(d->dtype == MI355_F32 || d->dtype == MI355_F16, "dtype %d unknown", d->dtype);
    MI355_REQUIRE(d->num_pool >= 1 && d->num_pool <= 7, "num_pool %d out of range", d->num_pool);
    MI355_REQUIRE(d->in_channels >= 1 && d->in_channels <= 8, "in_channels %d unsupported (1..8)", d->in_channels);
    MI355_REQUIRE(d->norm >= MI355_NORM_NONE && d->norm <= MI355_NORM_GROUP, "norm kind %d", d->norm);
    // every kernel computes LeakyReLU as max(x, slope * x), which is LeakyReLU for 0 <= slope <= 1 only; the
    // reference's is 0.01 (generic_UNet.py:39)
    MI355_REQUIRE(d->lrelu_slope >= 0.f && d->lrelu_slope <= 1.f, "lrelu_slope %g outside [0, 1]: the kernels evaluate LeakyReLU as max(x, slope x)", (double)d->lrelu_slope);
    mi355_unet *net = new mi355_unet();
    net->in_channels = d->in_channels; net->cin_pad = d->dtype == MI355_F16 ? 16 : 8; net->num_classes = d->num_classes;
    const bool use_stem = d->in_channels <= 4 && d->n_convs > 0 && d->convs[0].cout % 32 == 0 && d->convs[0].stride == 1 &&
                          !(getenv("MI355_NO_STEM") && getenv("MI355_NO_STEM")[0] == '1');
    if (use_stem) net->cin_pad = 4;  // NDHW4 input, x-taps folded into K (conv_stem.hip)
    net->num_pool = d->num_pool; net->norm = d->norm; net->num_groups = d->num_groups;
    net->nonlin_first = d->nonlin_first; net->dtype = d->dtype; net->eps = d->eps; net->slope = d->lrelu_slope;
    int rc = MI355_OK;
    int ci = 0;
    int prevC = net->cin_pad;
    net->enc.resize(d->num_pool + 1);
    net->dec.resize(d->num_pool);
    if (d->dtype == MI355_F16) net->tuh.resize(d->num_pool); else net->tu.resize(d->num_pool);
    std::vector<int> skipC(d->num_pool, 0);
    auto fail = [&](int code) { destroy(net); return code; };
    for (int l = 0; l <= d->num_pool && rc == MI355_OK; ++l) {
        if (d->enc_convs[l] < 1) { set_error("encoder stage %d has no convs", l); return fail(MI355_ERR_INVALID); }
        for (int i = 0; i < d->enc_convs[l]; ++i) {
            if (ci >= d->n_convs) { set_error("conv list too short"); return fail(MI355_ERR_INVALID); }
            const mi355_conv_desc &cd = d->convs[ci++];
            const int want_stride = (l > 0 && i == 0) ? 2 : 1;
            if (cd.stride != want_stride) { set_error("encoder conv %d.%d: stride %d, expected %d", l, i, cd.stride, want_stride); return fail(MI355_ERR_INVALID); }
            const int logical_in = (l == 0 && i == 0) ? d->in_channels : prevC;
            if (cd.cin != logical_in) { set_error("encoder conv %d.%d: cin %d, expected %d", l, i, cd.cin, logical_in); return fail(MI355_ERR_INVALID); }
            ConvLayer L;
            rc = build_conv(*net, cd, prevC, &L, use_stem && l == 0 && i == 0);
            if (rc != MI355_OK) return fail(rc);
            net->enc[l].push_back(L);
            prevC = cd.cout;
            net->max_channels = std::max(net->max_channels, cd.cout);
        }
        if (l < d->num_pool) skipC[l] = prevC;
    }
    for (int u = 0; u < d->num_pool; ++u) {
        const int l = d->num_pool - 1 - u;
        const mi355_tconv_desc &td = d->tconvs[u];
        if (td.cin != prevC) { set_error("tu.%d: cin %d, expected %d", u, td.cin, prevC); return fail(MI355_ERR_INVALID); }
        rc = d->dtype == MI355_F16 ? tconv_weights_upload_f16(td.weight, td.cin, td.cout, &net->tuh[u])
                                   : tconv_weights_upload(td.weight, td.cin, td.cout, &net->tu[u]);
        if (rc != MI355_OK) return fail(rc);
        net->max_channels = std::max(net->max_channels, td.cout);
        int inC = td.cout + skipC[l];
        if (d->dec_convs[u] < 1) { set_error("decoder stage %d has no convs", u); return fail(MI355_ERR_INVALID); }
        for (int i = 0; i < d->dec_convs[u]; ++i) {
            if (ci >= d->n_convs) { set_error("conv list too short"); return fail(MI355_ERR_INVALID); }
            const mi355_conv_desc &cd = d->convs[ci++];
            if (cd.stride != 1 || cd.cin != inC) { set_error("decoder conv %d.%d: cin %d stride %d, expected %d / 1", u, i, cd.cin, cd.stride, inC); return fail(MI355_ERR_INVALID); }
            ConvLayer L;
            rc = build_conv(*net, cd, inC, &L);
            if (rc != MI355_OK) return fail(rc);
            net->dec[u].push_back(L);
            inC = cd.cout;
            net->max_channels = std::max(net->max_channels, cd.cout);
        }
        prevC = inC;
    }
    if (ci != d->n_convs) { set_error("%d convs given, topology uses %d", d->n_convs, ci); return fail(MI355_ERR_INVALID); }
    if (d->head.cin != prevC || d->head.num_classes != d->num_classes) { set_error("head: cin %d classes %d, expected %d / %d", d->head.cin, d->head.num_classes, prevC, d->num_classes); return fail(MI355_ERR_INVALID); }
    rc = head_weights_upload(d->head.weight, d->head.bias, d->head.cin, d->head.num_classes, &net->head);
    if (rc != MI355_OK) return fail(rc);
    *out = net;
    return MI355_OK;
}

extern "C" int mi355_unet_destroy(mi355_unet_t net) {
    destroy(net);
    return MI355_OK;
}

extern "C" int64_t mi355_unet_flops(mi355_unet_t net, int d, int h, int w) {
    if (!net) return -1;
    const int np = net->num_pool;
    int64_t total = 0;
    auto vox = [&](int l) { return (int64_t)(d >> l) * (h >> l) * (w >> l); };
    for (int l = 0; l <= np; ++l)
        for (auto &L : net->enc[l]) total += 2 * vox(l) * L.cout * L.cin * 27;
    for (int u = 0; u < np; ++u) {
        const int l = np - 1 - u;
        total += 2 * vox(l + 1) * (net->dtype == MI355_F16 ? (int64_t)net->tuh[u].cin * net->tuh[u].cout : (int64_t)net->tu[u].cin * net->tu[u].cout) * 8;
        for (auto &L : net->dec[u]) total += 2 * vox(l) * L.cout * L.cin * 27;
    }
    total += 2 * vox(0) * net->head.cin * net->head.ncls;
    return total;
}

extern "C" int mi355_unet_forward(mi355_unet_t net, const float *x_dev, int n, int d, int h, int w, float *logits_dev,
                                  void *stream) {
    MI355_REQUIRE(net && x_dev && logits_dev && n > 0, "bad argument");
    MI355_TRY(bind_device());
    hipStream_t s = (hipStream_t)stream;
    Plan pl;
    MI355_TRY(make_plan(*net, n, d, h, w, &pl));
    MI355_TRY(ensure_arena(&pl, s));
    const int64_t V = (int64_t)d * h * w;
    MI355_TRY(nchw_to_ndhwc(x_dev, n, net->in_channels, V, net->cin_pad, (void *)(pl.arena + pl.x0_off), net->dtype, s));
    const void *feat; int fc; bool is_logits = false;
    FeatNorm head_norm;
    MI355_TRY(forward_features(net, pl, n, d, h, w, &feat, &fc, s, &is_logits, logits_dev, &head_norm));
    if (is_logits) return MI355_OK;  // the last conv wrote [n][ncls][V] straight into logits_dev
    MI355_REQUIRE(fc == net->head.cin, "head expects %d channels, decoder gives %d", net->head.cin, fc);
    MI355_TRY(head_logits(net->head, feat, net->dtype, n, V, logits_dev, s, head_norm));
    return MI355_OK;
}

extern "C" int mi355_profile_enable(mi355_unet_t net, int on) {
    MI355_REQUIRE(net, "null handle");
    net->prof_on = on != 0;
    if (net->prof_on) {
        // pre-create the events here, outside any timed region (a bench run records ~120 per step)
        while (net->event_pool.size() < 4096) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) break;
            net->event_pool.push_back(e);
        }
    }
    return MI355_OK;
}

// Synchronises the device, folds the recorded launches per kernel name, clears the log.
extern "C" int mi355_profile_read(mi355_unet_t net, mi355_prof_entry *out, int max_entries) {
    MI355_REQUIRE(net && out && max_entries > 0, "bad argument");
    MI355_HIP(hipDeviceSynchronize());
    int n = 0;
    for (auto &r : net->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        net->event_pool.push_back(r.a);
        net->event_pool.push_back(r.b);
        int k = 0;
        for (; k < n; ++k) if (r.name == out[k].name) break;
        if (k == n) {
            if (n == max_entries) continue;
            memset(&out[n], 0, sizeof(out[n]));
            snprintf(out[n].name, sizeof(out[n].name), "%s", r.name.c_str());
            ++n;
        }
        out[k].launches += 1; out[k].ms += ms; out[k].flops += r.flops; out[k].bytes += r.bytes;
    }
    net->prof.clear();
    return n;
}

extern "C" int mi355_compute_steps(int patch, int image, float step_size, int32_t *steps, int max_steps) {
    MI355_REQUIRE(patch > 0 && image >= patch && step_size > 0.f && step_size <= 1.f, "compute_steps: bad arguments");
    std::vector<int> st = compute_steps(patch, image, step_size);
    MI355_REQUIRE((int)st.size() <= max_steps, "compute_steps: %zu steps > buffer %d", st.size(), max_steps);
    for (size_t i = 0; i < st.size(); ++i) steps[i] = st[i];
    return (int)st.size();
}

// Partitioning B of SURVEY.md 8e with the reference's fold list (run_brats2021_inference_singlethread.py:161 folds=(0..4),
// :112-128): the work list is (fold, tile), item f * tiles + t, dealt round-robin over the ranks; agg_dev receives
// sum over this rank's items of the Gaussian-weighted, mirror-averaged probabilities - the fold mean is linear in the
// per-fold aggregates (mean_f(agg_f / cnt) = (sum_f agg_f) / cnt / n_folds), so ONE exchange per ensemble member suffices.
extern "C" int mi355_sw_partial_folds(const mi355_unet_t *nets, int n_nets, const float *vol_dev, int Z, int Y, int X,
                                      const mi355_sw_opts *opts, int rank, int world, float *agg_dev, float *cnt_dev, void *stream) {
    MI355_REQUIRE(nets && n_nets >= 1 && vol_dev && opts && agg_dev && world >= 1 && rank >= 0 && rank < world, "bad argument");
    for (int f = 0; f < n_nets; ++f) {
        MI355_REQUIRE(nets[f] != nullptr, "fold %d: null handle", f);
        MI355_REQUIRE(nets[f]->num_classes == nets[0]->num_classes, "fold %d has %d classes, fold 0 has %d", f, nets[f]->num_classes, nets[0]->num_classes);
    }
    MI355_TRY(bind_device());
    hipStream_t s = (hipStream_t)stream;
    SwGeom g;
    MI355_TRY(make_geom(*opts, Z, Y, X, &g));
    const size_t ZYXp = (size_t)g.Zp[0] * g.Zp[1] * g.Zp[2];
    MI355_HIP(hipMemsetAsync(agg_dev, 0, ZYXp * nets[0]->num_classes * sizeof(float), s));
    if (cnt_dev) MI355_HIP(hipMemsetAsync(cnt_dev, 0, ZYXp * sizeof(float), s));
    if (cnt_dev && world > 1) {
        // full normaliser of ONE fold on every rank: cnt[tile] += gaussian for ALL tiles, in tile order (no exchange needed)
        const bool use_gauss = opts->use_gaussian && g.tiles.size() > 1;
        if (use_gauss) MI355_TRY(ensure_gaussian(nets[0], g.P));
        for (const TileDesc &td : g.tiles)
            MI355_TRY(cnt_add_tile(use_gauss ? nets[0]->gauss_dev : nullptr, g.P[0], g.P[1], g.P[2], cnt_dev, g.Zp[1], g.Zp[2],
                                   td.z0, td.y0, td.x0, s));
    }
    // (world == 1: the first fold's aggregation kernels add the normaliser themselves, exactly as mi355_sw_predict does)
    for (int f = 0; f < n_nets; ++f)
        MI355_TRY(sw_accumulate(nets[f], vol_dev, Z, Y, X, *opts, g, rank, world, agg_dev, f == 0 ? cnt_dev : nullptr, s,
                                (long)f * (long)g.tiles.size()));
    return MI355_OK;
}

extern "C" int mi355_sw_partial(mi355_unet_t net, const float *vol_dev, int Z, int Y, int X, const mi355_sw_opts *opts,
                                int rank, int world, float *agg_dev, float *cnt_dev, void *stream) {
    return mi355_sw_partial_folds(&net, 1, vol_dev, Z, Y, X, opts, rank, world, agg_dev, cnt_dev, stream);
}

extern "C" int mi355_sw_finish_folds(const float *agg_dev, const float *cnt_dev, int num_classes, int Z, int Y, int X,
                                     const int32_t patch[3], int n_folds, float *probs_dev, void *stream) {
    MI355_REQUIRE(agg_dev && cnt_dev && probs_dev && patch && n_folds >= 1, "bad argument");
    MI355_TRY(bind_device());
    const int Zp = std::max(Z, patch[0]), Yp = std::max(Y, patch[1]), Xp = std::max(X, patch[2]);
    MI355_TRY(finish_probs(agg_dev, cnt_dev, num_classes, Z, Y, X, Zp, Yp, Xp, (Zp - Z) / 2, (Yp - Y) / 2, (Xp - X) / 2,
                           probs_dev, 0, (hipStream_t)stream));
    if (n_folds > 1) MI355_TRY(scale_inplace(probs_dev, (int64_t)num_classes * Z * Y * X, (float)n_folds, (hipStream_t)stream));
    return MI355_OK;
}

extern "C" int mi355_sw_finish(const float *agg_dev, const float *cnt_dev, int num_classes, int Z, int Y, int X,
                               const int32_t patch[3], float *probs_dev, void *stream) {
    return mi355_sw_finish_folds(agg_dev, cnt_dev, num_classes, Z, Y, X, patch, 1, probs_dev, stream);
}

extern "C" int mi355_sw_predict(const mi355_unet_t *nets, int n_nets, const float *vol_dev, int Z, int Y, int X,
                                const mi355_sw_opts *opts, float *probs_dev, void *stream) {
    MI355_REQUIRE(nets && n_nets >= 1 && vol_dev && opts && probs_dev, "bad argument");
    MI355_TRY(bind_device());
    hipStream_t s = (hipStream_t)stream;
    SwGeom g;
    MI355_TRY(make_geom(*opts, Z, Y, X, &g));
    const int C = nets[0]->num_classes;
    const size_t ZYXp = (size_t)g.Zp[0] * g.Zp[1] * g.Zp[2];
    // aggregation scratch: process-wide like the activation arena, grown on demand, never freed per call - a
    // hipMalloc / synchronise / hipFree round trip per volume costs more than the small kernels of a step
    void *scratch = nullptr;
    MI355_TRY(device_scratch(SCR_SW_AGG, s, ZYXp * (size_t)(C + 1) * sizeof(float), &scratch));
    float *agg = (float *)scratch, *cnt = agg + ZYXp * C;
    int rc = MI355_OK;
    for (int f = 0; f < n_nets && rc == MI355_OK; ++f) {
        if (nets[f]->num_classes != C) { set_error("fold %d has %d classes, fold 0 has %d", f, nets[f]->num_classes, C); rc = MI355_ERR_INVALID; break; }
        if (hipMemsetAsync(agg, 0, ZYXp * C * sizeof(float), s) != hipSuccess ||
            hipMemsetAsync(cnt, 0, ZYXp * sizeof(float), s) != hipSuccess) { set_error("memset failed"); rc = MI355_ERR_HIP; break; }
        rc = sw_accumulate(nets[f], vol_dev, Z, Y, X, *opts, g, 0, 1, agg, cnt, s);
        if (rc == MI355_OK)
            rc = finish_probs(agg, cnt, C, Z, Y, X, g.Zp[0], g.Zp[1], g.Zp[2], g.pad_lo[0], g.pad_lo[1], g.pad_lo[2],
                              probs_dev, f > 0, s);
    }
    // fold mean: np.mean(list_of_fp32_arrays, axis=0) = fp32 running sum in list order, one divide
    if (rc == MI355_OK && n_nets > 1) rc = scale_inplace(probs_dev, (int64_t)C * Z * Y * X, (float)n_nets, s);
    // asynchronous on `stream` like every other entry point that takes one: probs_dev is valid in stream order
    return rc;
}

// `sums` (device, [n][cout][2] doubles, zeroed here) != nullptr: the launch also carries the Instance/GroupNorm statistics
// epilogue (sum x, sum x^2 of its output per sample and channel) exactly as a run-time-norm block of the network does.
static int conv3d_ndhwc_f32_impl(const float *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                 const float *bias_host, int cout, int stride, int act, float slope, int impl, float *y_dev,
                                 double *sums, void *stream) {
    MI355_TRY(require_device());
    MI355_REQUIRE(act != ACT_LRELU || (slope >= 0.f && slope <= 1.f), "LeakyReLU slope %g outside [0, 1]", (double)slope);
    if (sums) MI355_HIP(hipMemsetAsync(sums, 0, (size_t)n * cout * 2 * sizeof(double), (hipStream_t)stream));
    if (cin == 4 && stride == 1 && impl == 0 && cout % 32 == 0) {  // the network's first-layer kernel
        StemWeights sw;
        MI355_TRY(stem_weights_upload(weight_host, bias_host, cin, cout, MI355_F32, &sw));
        int rc = conv3d_stem(sw, x_dev, n, d, h, w, y_dev, sums, act, slope, (hipStream_t)stream);
        g_last_conv_kernel = "conv3_stem_f32_kernel";
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        stem_weights_free(&sw);
        if (rc == MI355_OK && e != hipSuccess) { set_error("stem conv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
        return rc;
    }
    ConvWeights cw;
    MI355_TRY(conv_weights_upload(weight_host, bias_host, cin, cin, cout, stride, impl == 1, &cw));
    ConvCall c;
    c.in0 = x_dev; c.C0 = cin; c.N = n; c.Di = d; c.Hi = h; c.Wi = w; c.out = y_dev; c.act = act; c.slope = slope;
    c.stats = sums;
    MI355_REQUIRE(!sums || impl != 1, "the direct cross-check kernel carries no statistics epilogue");
    const char *kname = "conv3_direct_kernel";
    int rc = (impl == 1) ? conv3d_direct_f32(cw, c, (hipStream_t)stream) : conv3d_mfma_f32(cw, c, (hipStream_t)stream, &kname);
    g_last_conv_kernel = kname ? kname : "";
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    conv_weights_free(&cw);
    if (rc == MI355_OK && e != hipSuccess) { set_error("conv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
    return rc;
}

extern "C" int mi355_conv3d_ndhwc(const float *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                  const float *bias_host, int cout, int stride, int act, float slope, int impl, float *y_dev,
                                  void *stream) {
    return conv3d_ndhwc_f32_impl(x_dev, n, d, h, w, cin, weight_host, bias_host, cout, stride, act, slope, impl, y_dev, nullptr, stream);
}

extern "C" int mi355_tconv3d_ndhwc(const float *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                   int cout, float *y_dev, void *stream) {
    MI355_TRY(require_device());
    TConvWeights tw;
    MI355_TRY(tconv_weights_upload(weight_host, cin, cout, &tw));
    const char *tname = "tconv2_f32_mfma_v2_kernel";
    int rc = tconv2_mfma_f32(tw, x_dev, n, d, h, w, y_dev, (hipStream_t)stream, &tname);
    g_last_conv_kernel = tname;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    tconv_weights_free(&tw);
    if (rc == MI355_OK && e != hipSuccess) { set_error("tconv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
    return rc;
}

// The single-op fp16 entry points take and return PLAIN NDHWC tensors (include/mi355_nnunet.h); the kernels work on
// channel-blocked ones (common.h), so the operands pass through ndhwc_to_b8 / b8_to_ndhwc here.
namespace {
struct TmpBuf {
    void *p = nullptr;
    ~TmpBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

static int conv3d_ndhwc_f16_impl(const void *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                 const float *bias_host, int cout, int stride, int act, float slope, void *y_dev,
                                 double *sums, void *stream) {
    MI355_TRY(require_device());
    MI355_REQUIRE(stride == 1 || stride == 2, "conv stride %d unsupported", stride);
    MI355_REQUIRE(act != ACT_LRELU || (slope >= 0.f && slope <= 1.f), "LeakyReLU slope %g outside [0, 1]", (double)slope);
    hipStream_t s = (hipStream_t)stream;
    if (sums) MI355_HIP(hipMemsetAsync(sums, 0, (size_t)n * cout * 2 * sizeof(double), s));
    const int64_t Vi = (int64_t)d * h * w;
    const int64_t Vo = (int64_t)((d - 1) / stride + 1) * ((h - 1) / stride + 1) * ((w - 1) / stride + 1);
    MI355_REQUIRE(cout % 8 == 0, "fp16 conv needs cout %% 8 == 0 (got %d)", cout);
    TmpBuf yb;
    MI355_HIP(hipMalloc(&yb.p, (size_t)n * Vo * cout * 2));
    int rc;
    if (cin == 4 && stride == 1 && cout % 32 == 0) {  // the network's first-layer kernel: plain NDHW4 input
        StemWeights sw;
        MI355_TRY(stem_weights_upload(weight_host, bias_host, cin, cout, MI355_F16, &sw));
        rc = conv3d_stem(sw, x_dev, n, d, h, w, yb.p, sums, act, slope, s);
        g_last_conv_kernel = "conv3_stem_f16_kernel";
        if (rc == MI355_OK) rc = b8_to_ndhwc((const _Float16 *)yb.p, n, cout, Vo, (_Float16 *)y_dev, s);
        hipError_t e = hipStreamSynchronize(s);
        stem_weights_free(&sw);
        if (rc == MI355_OK && e != hipSuccess) { set_error("stem conv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
        return rc;
    }
    MI355_REQUIRE(cin % 8 == 0, "fp16 conv needs cin %% 8 == 0 (got %d)", cin);
    TmpBuf xb;
    MI355_HIP(hipMalloc(&xb.p, (size_t)n * Vi * cin * 2));
    MI355_TRY(ndhwc_to_b8((const _Float16 *)x_dev, n, cin, Vi, (_Float16 *)xb.p, s));
    ConvWeightsH cw;
    MI355_TRY(conv_weights_upload_f16(weight_host, bias_host, cin, cin, cout, stride, &cw));
    ConvCallH c;
    c.in0 = (const _Float16 *)xb.p; c.C0 = cin; c.N = n; c.Di = d; c.Hi = h; c.Wi = w; c.out = (_Float16 *)yb.p;
    c.act = act; c.slope = slope; c.stats = sums;
    const char *kname = nullptr;
    rc = conv3d_mfma_f16(cw, c, s, &kname);
    g_last_conv_kernel = kname ? kname : "";
    if (rc == MI355_OK) rc = b8_to_ndhwc((const _Float16 *)yb.p, n, cout, Vo, (_Float16 *)y_dev, s);
    hipError_t e = hipStreamSynchronize(s);
    conv_weights_free_f16(&cw);
    if (rc == MI355_OK && e != hipSuccess) { set_error("conv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
    return rc;
}

extern "C" int mi355_conv3d_ndhwc_f16(const void *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                      const float *bias_host, int cout, int stride, int act, float slope, void *y_dev,
                                      void *stream) {
    return conv3d_ndhwc_f16_impl(x_dev, n, d, h, w, cin, weight_host, bias_host, cout, stride, act, slope, y_dev, nullptr, stream);
}

extern "C" int mi355_conv3d_sums_ndhwc(const void *x_dev, int dtype, int n, int d, int h, int w, int cin, const float *weight_host,
                                       const float *bias_host, int cout, int stride, int act, float slope, void *y_dev,
                                       double *sums_dev, void *stream) {
    MI355_REQUIRE(sums_dev != nullptr, "mi355_conv3d_sums_ndhwc: sums_dev is null");
    MI355_REQUIRE(dtype == MI355_F32 || dtype == MI355_F16, "unknown dtype %d", dtype);
    if (dtype == MI355_F16)
        return conv3d_ndhwc_f16_impl(x_dev, n, d, h, w, cin, weight_host, bias_host, cout, stride, act, slope, y_dev, sums_dev, stream);
    return conv3d_ndhwc_f32_impl((const float *)x_dev, n, d, h, w, cin, weight_host, bias_host, cout, stride, act, slope, 0, (float *)y_dev,
                                 sums_dev, stream);
}

extern "C" int mi355_tconv3d_ndhwc_f16(const void *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                                       int cout, void *y_dev, void *stream) {
    MI355_TRY(require_device());
    hipStream_t s = (hipStream_t)stream;
    const int64_t Vi = (int64_t)d * h * w;
    TmpBuf xb, yb;
    MI355_HIP(hipMalloc(&xb.p, (size_t)n * Vi * cin * 2));
    MI355_HIP(hipMalloc(&yb.p, (size_t)n * Vi * 8 * cout * 2));
    TConvWeightsH tw;
    MI355_TRY(tconv_weights_upload_f16(weight_host, cin, cout, &tw));
    int rc = ndhwc_to_b8((const _Float16 *)x_dev, n, cin, Vi, (_Float16 *)xb.p, s);
    const char *tname = nullptr;
    if (rc == MI355_OK) rc = tconv2_mfma_f16(tw, (const _Float16 *)xb.p, n, d, h, w, (_Float16 *)yb.p, s, &tname);
    g_last_conv_kernel = tname ? tname : "";
    if (rc == MI355_OK) rc = b8_to_ndhwc((const _Float16 *)yb.p, n, cout, Vi * 8, (_Float16 *)y_dev, s);
    hipError_t e = hipStreamSynchronize(s);
    tconv_weights_free_f16(&tw);
    if (rc == MI355_OK && e != hipSuccess) { set_error("tconv kernel failed: %s", hipGetErrorString(e)); rc = MI355_ERR_HIP; }
    return rc;
}
