// Host-side launchers of the gfx950 kernels (declarations).  All tensors are NDHWC.
#pragma once
#include <vector>

#include "common.h"

namespace mi355 {

enum { ACT_NONE = 0, ACT_LRELU = 1 };

// ---------------------------------------------------------------- conv 3x3x3 (implicit GEMM)
// Weights packed for the MFMA B operand; see pack_conv_weights_f32 in conv3d.hip.
struct ConvWeights {
    int cin = 0, cin_pad = 0, cout = 0, stride = 1;
    int cc = 0;               // channel chunk staged in LDS per pass
    int nf = 1;               // 32-wide cout fragments per workgroup
    bool pipe = false;        // pipelined (persistent, double-buffered) kernel; implies cc == 8
    float *wp_dev = nullptr;  // packed weights (device), chunk size cc
    float *wp16_dev = nullptr;  // second pack with 16-channel chunks for the 512-voxel-tile simple kernel (auto mode)
    float *wpw_dev = nullptr;   // Winograd pack (stride 1, 16-channel chunks, 32-cout blocks): F(2,3) along y, or
    bool wino2 = false;         // F(2x2,3x3) over (z, y) when wino2
    float *wp3_dev = nullptr;   // 3-D Winograd pack F(2x2x2,3x3x3) (conv3d_wino3.hip; stride 1, 16-channel chunks, 32-cout blocks)
    float *bias_dev = nullptr;
    float *w_plain_dev = nullptr;  // [cout][cin][27] PyTorch order (direct kernel / tests)
    size_t wp_bytes = 0;
};

int conv_weights_upload(const float *w_host, const float *bias_host, int cin, int cin_pad, int cout,
                        int stride, bool keep_plain, ConvWeights *out);
void conv_weights_free(ConvWeights *w);

struct ConvCall {
    const float *in0 = nullptr;  // [N,Di,Hi,Wi,C0]
    const float *in1 = nullptr;  // [N,Di,Hi,Wi,C1] second half of a virtual concat, or null
    int C0 = 0, C1 = 0;
    int N = 0, Di = 0, Hi = 0, Wi = 0;
    float *out = nullptr;     // [N,Do,Ho,Wo,Cout]
    double *stats = nullptr;  // [N][Cout][2] (sum, sum of squares) accumulated when non-null
    int act = ACT_NONE;
    float slope = 0.01f;
    // fused 1x1x1 segmentation head (last decoder conv, Cout = one workgroup's couts): when head_out is set the
    // feature map is NOT stored; logits [N][ncls][Vo] fp32 = head_w [ncls][Cout] . act(conv) + head_b are
    const float *head_w = nullptr, *head_b = nullptr;
    float *head_out = nullptr;
    int head_ncls = 0;
    // in0 is the RAW conv output of the previous block: apply x * in_scale[n][c] + in_shift[n][c] (+ LeakyReLU when in_act) while
    // staging it (only where conv3d_wino3_fuses_input_norm says so: the F(2x2x2,3x3x3) kernel normalises its brick in LDS)
    const float *in_scale = nullptr, *in_shift = nullptr;
    int in_act = ACT_NONE;
};
int conv3d_mfma_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s, const char **kernel_name = nullptr);
// F(2x2x2, 3x3x3) kernel (conv3d_wino3.hip): launches when the call fits it and says so in *taken
void pack_conv_weights_wino3(const float *w, int cin, int cin_pad, int cout, std::vector<float> &out);
bool conv3d_wino3_enabled();
bool conv3d_wino3_fuses_input_norm(const ConvWeights &w, const ConvCall &c);
int conv3d_wino3_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s, const char **kernel_name, bool *taken);
int conv3d_direct_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s);

// first conv of the network (Cin <= 4): x-taps folded into K, NDHW4 input (conv_stem.hip)
struct StemWeights {
    int cin = 0, cout = 0, dtype = 0;
    void *wp_dev = nullptr;
    float *bias_dev = nullptr;
};
int stem_weights_upload(const float *w_host, const float *bias_host, int cin, int cout, int dtype, StemWeights *out);
void stem_weights_free(StemWeights *w);
int conv3d_stem(const StemWeights &w, const void *in, int N, int D, int H, int W, void *out, double *stats, int act,
                float slope, hipStream_t s);

// fp16 storage / fp32 accumulate variants (conv3d_f16.hip)
struct ConvWeightsH {
    int cin = 0, cin_pad = 0, cout = 0, stride = 1, nf = 1;
    _Float16 *wp_dev = nullptr;
    float *bias_dev = nullptr;
};
int conv_weights_upload_f16(const float *w_host, const float *bias_host, int cin, int cin_pad, int cout, int stride,
                            ConvWeightsH *out);
void conv_weights_free_f16(ConvWeightsH *w);
struct ConvCallH {
    const _Float16 *in0 = nullptr, *in1 = nullptr;
    int C0 = 0, C1 = 0;
    int N = 0, Di = 0, Hi = 0, Wi = 0;
    _Float16 *out = nullptr;
    double *stats = nullptr;
    int act = ACT_NONE;
    float slope = 0.01f;
    const float *head_w = nullptr, *head_b = nullptr;  // fused segmentation head, see ConvCall
    float *head_out = nullptr;
    int head_ncls = 0;
    // in0 is the RAW conv output of the previous block: apply x * in_scale[n][c] + in_shift[n][c] (+ LeakyReLU when in_act)
    // while staging it (only where conv3d_f16_fuses_input_norm says so)
    const float *in_scale = nullptr, *in_shift = nullptr;
    int in_act = ACT_NONE;
};
bool conv3d_f16_fuses_input_norm(const ConvWeightsH &w, const ConvCallH &c);
int conv3d_mfma_f16(const ConvWeightsH &w, const ConvCallH &c, hipStream_t s, const char **kernel_name = nullptr);
// stride-2 LDS-DMA kernel (conv3d_f16_s2.hip): launches when the call fits it and says so in *taken
int conv3d_f16_s2dma(const ConvWeightsH &w, const ConvCallH &c, hipStream_t s, const char **kernel_name, bool *taken);

// ---------------------------------------------------------------- transposed conv k=2 s=2
struct TConvWeights {
    int cin = 0, cout = 0;
    float *wp_dev = nullptr;
};
int tconv_weights_upload(const float *w_host, int cin, int cout, TConvWeights *out);
void tconv_weights_free(TConvWeights *w);
// in [N,D,H,W,Cin] -> out [N,2D,2H,2W,Cout]
int tconv2_mfma_f32(const TConvWeights &w, const float *in, int N, int D, int H, int W, float *out,
                    hipStream_t s, const char **kernel_name = nullptr);
struct TConvWeightsH {
    int cin = 0, cout = 0;
    _Float16 *wp_dev = nullptr;
};
int tconv_weights_upload_f16(const float *w_host, int cin, int cout, TConvWeightsH *out);
void tconv_weights_free_f16(TConvWeightsH *w);
int tconv2_mfma_f16(const TConvWeightsH &w, const _Float16 *in, int N, int D, int H, int W, _Float16 *out,
                    hipStream_t s, const char **kernel_name = nullptr);

// ---------------------------------------------------------------- normalisation
// stats [N][C][2] doubles -> scale/shift [N][C] so that y = x*scale + shift.
int norm_finalize(const double *stats, int N, int C, int64_t count, int kind, int groups, float eps,
                  const float *gamma, const float *beta, float *scale, float *shift, hipStream_t s);
// in place: x = act(x*scale[n][c] + shift[n][c]) over [N][V][C]
int norm_apply(void *x, int dtype, int N, int64_t V, int C, const float *scale, const float *shift, int act,
               float slope, hipStream_t s);

// ---------------------------------------------------------------- tiles / head / aggregate
struct TileDesc {  // one forward sample = one (tile, mirror)
    int z0, y0, x0;  // origin in the padded volume
    int mirror;      // bit0 flip z, bit1 flip y, bit2 flip x
};
// vol [C][Z][Y][X] (unpadded; pad offsets give where it sits in the padded volume)
// -> x [n_samples][P0][P1][P2][Cpad] (channels >= C are zero)
int extract_tiles(const float *vol, int C, int Z, int Y, int X, int padz, int pady, int padx,
                  const TileDesc *tiles_host, int n_samples, int P0, int P1, int P2, int Cpad, void *x, int dtype,
                  hipStream_t s);
// plain NDHWC <-> channel-blocked [N][C / 8][V][8] fp16 (common.h)
int ndhwc_to_b8(const _Float16 *x, int N, int C, int64_t V, _Float16 *y, hipStream_t s);
int b8_to_ndhwc(const _Float16 *x, int N, int C, int64_t V, _Float16 *y, hipStream_t s);
// NCDHW -> NDHWC(Cpad) for the plain forward API
int nchw_to_ndhwc(const float *x, int N, int C, int64_t V, int Cpad, void *y, int dtype, hipStream_t s);

struct HeadWeights {
    int cin = 0, ncls = 0;
    float *w_dev = nullptr;  // [ncls][cin]
    float *b_dev = nullptr;  // [ncls]
};
int head_weights_upload(const float *w_host, const float *b_host, int cin, int ncls, HeadWeights *out);
void head_weights_free(HeadWeights *w);
// The feature map handed to the head may be the RAW output of the last decoder conv: then scale / shift [N][C] (fp32) hold
// its Instance/GroupNorm affine and the head applies act(x * scale + shift) per channel while it reads the features
// (slope = 1: no activation).  scale == nullptr: the features are final.
struct FeatNorm {
    const float *scale = nullptr, *shift = nullptr;
    float slope = 1.0f;
};
// feat [N][V][C] -> logits [N][ncls][V]
int head_logits(const HeadWeights &w, const void *feat, int dtype, int N, int64_t V, float *logits, hipStream_t s,
                const FeatNorm &fn = FeatNorm());
// One tile: result = sum_m (1/n_mirrors) * flip_back(nonlin(head(feat[first_sample+m])));
// agg[c][tile] += result * gauss ; cnt[tile] += gauss (cnt may be null).
int head_aggregate(const HeadWeights &w, const void *feat, int dtype, int first_sample, const int *mirrors_host,
                   int n_mirrors, int P0, int P1, int P2, int nonlin, const float *gauss, float *agg,
                   float *cnt, int Zp, int Yp, int Xp, int z0, int y0, int x0, hipStream_t s, const FeatNorm &fn = FeatNorm());
// same as head_aggregate for forwards whose last conv already produced logits [n_samples][ncls][PV]
int logits_aggregate(const float *logits, int ncls, int first_sample, const int *mirrors_host, int n_mirrors, int P0, int P1,
                     int P2, int nonlin, const float *gauss, float *agg, float *cnt, int Zp, int Yp, int Xp, int z0, int y0,
                     int x0, hipStream_t s);
// probs[c][z][y][x] (+)= agg[c][z+pz][y+py][x+px] / cnt[...]; then optional scale (fold mean)
int finish_probs(const float *agg, const float *cnt, int C, int Z, int Y, int X, int Zp, int Yp, int Xp,
                 int pz, int py, int px, float *probs, int accumulate, hipStream_t s);
int scale_inplace(float *x, int64_t n, float divisor, hipStream_t s);
// cnt[tile] += gauss (or 1): the normaliser of a tile this rank does not evaluate itself
int cnt_add_tile(const float *gauss, int P0, int P1, int P2, float *cnt, int Yp, int Xp, int z0, int y0, int x0,
                 hipStream_t s);
// in place: x = act(x*scale + shift), scale/shift [C] shared by every sample (un-folded BN)

}  // namespace mi355
