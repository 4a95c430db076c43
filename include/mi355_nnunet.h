/*
 * mi355_nnunet.h - C ABI of the MI355X-native nnU-Net (BraTS) sliding-window predictor.
 *
 * This is the drop-in boundary for the one hot path of the reference
 * (SURVEY.md section 8b).  The reference is pure Python; what a maintainer would bind
 * is a ctypes stub (INTEGRATION.md shows it).  Each entry point names the reference
 * interface it replaces (paths relative to the reference repo):
 *
 *   mi355_unet_create        model_architecture/generic_UNet.py:188-421  Generic_UNet.__init__
 *                            + nnunet load_model_and_checkpoint_files / trainer.load_checkpoint_ram
 *                              as called at run_brats2021_inference_singlethread.py:178-183,95,113
 *   mi355_unet_forward       model_architecture/generic_UNet.py:423-446  Generic_UNet.forward
 *   mi355_sw_predict         trainer.predict_preprocessed_data_return_seg_and_softmax(...)[1]
 *                            called at run_brats2021_inference_singlethread.py:97-106,114-123
 *                            (+ the fold mean at :128 when several handles are given)
 *   mi355_regions_to_labels  save_segmentation_nifti_from_softmax(..., region_class_order=(1,2,3))
 *                            called at run_brats2021_inference_singlethread.py:144-156
 *   mi355_label_ensemble     run_brats2021_inference_singlethread.py:299-305  np.round((s1+s2)/2)
 *   mi355_prob_mean          archived/kaist_original_inference.py:30-32 (nnUNet_ensemble: mean of two softmax volumes)
 *   mi355_zscore_masked      trainer.preprocess_patient -> nonCT + use_mask_for_norm normalisation,
 *                            called at run_brats2021_inference_singlethread.py:89
 *   mi355_resize_axis        trainer.preprocess_patient -> resample_patient (same call) and the resampling inside
 *                            save_segmentation_nifti_from_softmax (:131-138, :144-156)
 *
 * Conventions: every function returns 0 on success and a negative code on failure;
 * mi355_last_error() gives the message of the calling thread's last failure.
 * All "dev" pointers are device (HBM) pointers on the current HIP device; the caller
 * owns them.  The library owns its weights and activation arena.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  ONE PROCESS PER GPU: the library binds
 * to the HIP device that is current at its first compute call (weights, arena and scratch
 * buffers live there) and every later call fails with MI355_ERR_INVALID while another device
 * is current.  The activation arena and every scratch buffer exist once per STREAM ("lane", round 5): work
 * issued on one stream is ordered by that stream and shares its arena across handles; up to four streams may
 * carry work at the same time (a fifth takes over the least recently used lane after a device synchronise).
 * A handle may be used on two streams at once (its weights are read-only); not from two threads while its
 * profiling log is enabled.  There is NO CPU fallback: on a
 * machine without a gfx950 device every compute entry point fails with MI355_ERR_NO_DEVICE.
 */
#ifndef MI355_NNUNET_H
#define MI355_NNUNET_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_ERR_INVALID (-1)
#define MI355_ERR_HIP (-2)
#define MI355_ERR_NO_DEVICE (-3)
#define MI355_ERR_UNSUPPORTED (-4)

enum { MI355_NORM_NONE = 0, MI355_NORM_BATCH = 1, MI355_NORM_INSTANCE = 2, MI355_NORM_GROUP = 3 };
enum { MI355_F32 = 0, MI355_F16 = 1 };
enum { MI355_NONLIN_IDENTITY = 0, MI355_NONLIN_SIGMOID = 1, MI355_NONLIN_SOFTMAX = 2 };

/* One ConvDropoutNormNonlin block (generic_UNet.py:27-72): Conv3d k=3 p=1 + norm + LeakyReLU.
 * All pointers are HOST pointers to fp32 data in the PyTorch layouts; they are read
 * during mi355_unet_create only. */
typedef struct {
    int32_t cin, cout, stride;
    const float *weight;       /* [cout][cin][3][3][3] */
    const float *bias;         /* [cout] or NULL */
    const float *gamma, *beta; /* norm affine [cout]; NULL -> 1 / 0 */
    const float *running_mean, *running_var; /* MI355_NORM_BATCH only */
} mi355_conv_desc;

/* ConvTranspose3d k=2 s=2 bias=False (generic_UNet.py:363-364). */
typedef struct {
    int32_t cin, cout;
    const float *weight; /* [cin][cout][2][2][2] */
} mi355_tconv_desc;

/* seg_outputs[-1]: Conv3d 1x1x1 (generic_UNet.py:389-391). */
typedef struct {
    int32_t cin, num_classes;
    const float *weight; /* [num_classes][cin] */
    const float *bias;   /* [num_classes] or NULL (seg_output_use_bias=False) */
} mi355_head_desc;

typedef struct {
    int32_t in_channels, num_classes, num_pool;
    int32_t norm;       /* MI355_NORM_* ; BATCH is eval mode and is folded into the conv */
    int32_t num_groups; /* MI355_NORM_GROUP */
    float eps, lrelu_slope;
    int32_t nonlin_first; /* 1 = ConvDropoutNonlinNorm (generic_UNet.py:75-80) */
    int32_t dtype;        /* MI355_F32 | MI355_F16 (storage; accumulation is always fp32) */
    const int32_t *enc_convs; /* [num_pool+1] convs per encoder stage; last = bottleneck */
    const int32_t *dec_convs; /* [num_pool]   convs per decoder stage */
    const mi355_conv_desc *convs; /* execution order: encoder stages, bottleneck, decoder stages */
    int32_t n_convs;
    const mi355_tconv_desc *tconvs; /* [num_pool] */
    mi355_head_desc head;
} mi355_unet_desc;

typedef struct mi355_unet *mi355_unet_t;

typedef struct {
    int32_t patch[3];     /* (z,y,x), e.g. 128,128,128 */
    float step_size;      /* 0.5 */
    int32_t use_gaussian; /* 1 */
    int32_t mirror_axes;  /* bit0 = z, bit1 = y, bit2 = x; 7 = 8-way TTA, 0 = no TTA */
    int32_t nonlin;       /* MI355_NONLIN_* applied to the logits of every forward */
    int32_t batch_tiles;  /* tiles per forward pass (0 = choose) */
} mi355_sw_opts;

const char *mi355_last_error(void);
int mi355_version(void);
/* number of visible gfx950 devices (0 if none / no HIP runtime device); other architectures are not counted */
int mi355_device_count(void);

int mi355_unet_create(const mi355_unet_desc *desc, mi355_unet_t *out);
int mi355_unet_destroy(mi355_unet_t net);
/* 2*MAC of every conv / transposed conv evaluated per forward of one [d,h,w] patch. */
int64_t mi355_unet_flops(mi355_unet_t net, int d, int h, int w);

/* x_dev: [n][in_channels][d][h][w] fp32 (NCDHW, as the reference module takes it).
 * logits_dev: [n][num_classes][d][h][w] fp32. */
int mi355_unet_forward(mi355_unet_t net, const float *x_dev, int n, int d, int h, int w,
                       float *logits_dev, void *stream);

/* Sliding-window prediction of one preprocessed volume.
 * vol_dev: [in_channels][Z][Y][X] fp32; probs_dev: [num_classes][Z][Y][X] fp32.
 * With n_nets > 1 the result is the arithmetic mean over the handles (folds), summed in
 * handle order (run_brats2021_inference_singlethread.py:128).
 * Asynchronous on `stream`: the work is enqueued, probs_dev is valid in stream order (a device fault surfaces at the
 * caller's next synchronisation, as with any HIP launch). */
int mi355_sw_predict(const mi355_unet_t *nets, int n_nets, const float *vol_dev, int Z, int Y, int X,
                     const mi355_sw_opts *opts, float *probs_dev, void *stream);
/* Host helper: the step table the predictor uses (nnU-Net v1 _compute_steps_for_sliding_window).
 * Writes at most max_steps entries, returns the count (or <0). */
int mi355_compute_steps(int patch, int image, float step_size, int32_t *steps, int max_steps);
/* Tile-sharded variant for multi-GPU: only tiles with (index % world) == rank are evaluated;
 * agg_dev [num_classes][Zp][Yp][Xp] receives the Gaussian-weighted partial sums (to be summed
 * across ranks in rank order), cnt_dev [Zp][Yp][Xp] the full normaliser. Zp.. = max(Z, patch). */
int mi355_sw_partial(mi355_unet_t net, const float *vol_dev, int Z, int Y, int X,
                     const mi355_sw_opts *opts, int rank, int world, float *agg_dev, float *cnt_dev,
                     void *stream);
int mi355_sw_finish(const float *agg_dev, const float *cnt_dev, int num_classes, int Z, int Y, int X,
                    const int32_t patch[3], float *probs_dev, void *stream);
/* The same with the reference's FOLD LIST (run_brats2021_inference_singlethread.py:161 folds=(0,1,2,3,4); :112-128 one
 * prediction per fold, np.mean over them): the work list is (fold, tile), item f * tiles + t, and item i belongs to rank
 * i % world.  agg_dev = sum over this rank's items of the Gaussian-weighted, mirror-averaged probabilities; the fold mean is
 * linear in the per-fold aggregates - mean_f(agg_f / cnt) = (sum_f agg_f) / cnt / n_folds - so ONE exchange of agg_dev per
 * ensemble member serves all folds (SURVEY.md 8e partitioning B: "(tile x mirror [x fold x model]) work list").
 * mi355_sw_finish_folds: probs = agg / cnt / n_folds on the rank-ordered sum of the partial aggregates (equals the per-fold
 * normalise-then-average of mi355_sw_predict up to fp32 rounding of the division order). */
int mi355_sw_partial_folds(const mi355_unet_t *nets, int n_nets, const float *vol_dev, int Z, int Y, int X,
                           const mi355_sw_opts *opts, int rank, int world, float *agg_dev, float *cnt_dev,
                           void *stream);
int mi355_sw_finish_folds(const float *agg_dev, const float *cnt_dev, int num_classes, int Z, int Y, int X,
                          const int32_t patch[3], int n_folds, float *probs_dev, void *stream);

/* seg = 0; for i in 0..C-1: seg[probs[i] > 0.5] = order[i]; pasted at bbox_lo into a zeroed
 * [full_z][full_y][full_x] uint8 volume.  order == NULL: seg = argmax over the C channels (first maximum wins), what
 * the same export does for trainers without regions (region_class_order=None). */
int mi355_regions_to_labels(const float *probs_dev, int C, int Z, int Y, int X, const int32_t *order,
                            const int32_t bbox_lo[3], const int32_t full[3], uint8_t *labels_dev,
                            void *stream);
/* out = uint8(round_half_even((a + b) / 2)) elementwise. */
int mi355_label_ensemble(const uint8_t *a_dev, const uint8_t *b_dev, uint8_t *out_dev, int64_t n,
                         void *stream);
/* probs_out = (a + b) / 2 (nnUNet_ensemble mode). */
int mi355_prob_mean(const float *a_dev, const float *b_dev, float *out_dev, int64_t n, void *stream);
/* Per channel: x[m] = (x[m]-mean(x[m]))/(std(x[m])+1e-8) ; x[~m] = 0  (m = mask != 0, ddof 0). */
int mi355_zscore_masked(float *vol_dev, const uint8_t *mask_dev, int C, int64_t voxels, void *stream);

/* ---- rows SURVEY.md 8f marks "next" (consumers of the label map / config 5's retrieval step) ---- */
/* out[i] = map[in[i]] (convert_labels_to_brats.py:34-55: nnU-Net {1,2,3} -> BraTS2025 {2,1,3} / BraTS2021 {2,1,4}). */
int mi355_label_remap(const uint8_t *in_dev, uint8_t *out_dev, int64_t n, const uint8_t *map256_host, void *stream);
/* counts_host[p*K+g] = #voxels with prediction p and ground truth g: everything evaluate_segmentation.py:12-49,
 * 129-195 derives (Dice, IoU, sensitivity, specificity per label and for WT/TC/ET) follows from these integers.
 * Bin K-1 is the "other" bin: it collects every label >= K-1, so pass K = (largest label of interest) + 2 and no
 * out-of-range label is ever counted as a real one; the K*K counts always sum to n.  2 <= K <= 8. */
int mi355_label_confusion(const uint8_t *pred_dev, const uint8_t *gt_dev, int64_t n, int K, uint64_t *counts_host,
                          void *stream);
/* scores = V @ q over L2-normalised rows, top-k by score (RAG_Assistant/rag_assistant.py:197-211).
 * vectors_dev [N][D] fp32, query_dev [D]; returns the number of results written (<= k) or < 0. */
int mi355_cosine_topk(const float *vectors_dev, const float *query_dev, int N, int D, int k, int32_t *idx_host,
                      float *scores_host, void *stream);

/* crop_to_nonzero of trainer.preprocess_patient (run_brats2021_inference_singlethread.py:89; nnU-Net v1
 * cropping.crop_to_nonzero): mask = OR_c(vol[c] != 0) with holes filled (scipy.ndimage.binary_fill_holes, 6-connectivity),
 * bbox_host = {z_lo, z_hi, y_lo, y_hi, x_lo, x_hi} (hi exclusive).  vol_dev [C][Z][Y][X] fp32, mask_dev [Z][Y][X] uint8.
 * Synchronous (returns the box). */
int mi355_crop_mask(const float *vol_dev, int C, int Z, int Y, int X, uint8_t *mask_dev, int32_t *bbox_host, void *stream);

/* Resampling between voxel grids (round 4): step 4 of trainer.preprocess_patient (run_brats2021_inference_singlethread.py:89 ->
 * nnU-Net v1 resample_patient: data order 3, mask order 1, a low-resolution axis separately with order 0) and the resampling inside
 * save_segmentation_nifti_from_softmax(..., order=1, force_separate_z=None, interpolation_order_z=0) (driver :131-138, :144-156).
 * One 1-D pass: in_dev viewed as [outer][n_in][inner] fp32 -> out_dev [outer][n_out][inner], sampled on the half-pixel-centred
 * grid x_in = (x_out + 0.5) * n_in / n_out - 0.5 with edge replication - what skimage.transform.resize(order, mode='edge',
 * anti_aliasing=False) / scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) do along one axis.  order 0 (nearest), 1 (linear)
 * or 3 (cubic B-spline incl. its prefilter).  A tensor-product resize is one pass per axis, in any order.  Asynchronous on `stream`. */
int mi355_resize_axis(const float *in_dev, float *out_dev, int64_t outer, int n_in, int n_out, int64_t inner, int order,
                      void *stream);
/* x[g][0..n_per_group) clipped to [min, max] of ref[g][0..ref_per_group), g < groups (skimage's resize clips its output to the
 * range of its input image: per channel for a 3-D resize, per slice in nnU-Net's separate-z mode). */
int mi355_clip_to_range_of(float *x_dev, int64_t groups, int64_t n_per_group, const float *ref_dev, int64_t ref_per_group,
                           void *stream);
/* out[i] = x[i] >= thr (batchgenerators resize_segmentation: a label survives where its linearly resized indicator is >= 0.5;
 * here the inside-the-brain mask that use_mask_for_norm reads after resampling). */
int mi355_threshold_ge(const float *x_dev, float thr, uint8_t *out_dev, int64_t n, void *stream);
/* out[i] = mask[i] != 0 ? 1.0f : 0.0f (the indicator resize_segmentation resizes). */
int mi355_mask_to_float(const uint8_t *mask_dev, float *out_dev, int64_t n, void *stream);

/* Per-label voxel statistics of a label map [d0][d1][d2] (feature_extraction/utils.py:167-216: the integers behind
 * get_tumor_masks + calculate_volume + get_centroid + get_bounding_box).  stats_host[label * 10 + f], label < K <= 8:
 * f = 0 count, 1..3 sum of the coordinates along axis 0..2, 4..6 minimum, 7..9 maximum coordinate (-1 / 2^40 when the
 * label is absent; for label 0 only the count is filled).  Labels >= K are ignored.  Synchronous. */
int mi355_label_stats(const uint8_t *seg_dev, int d0, int d1, int d2, int K, int64_t *stats_host, void *stream);

/* Per-kernel timing with HIP events on the stream the kernels are launched on (bench.py's
 * roofline). flops / bytes are the ALGORITHMIC work of the recorded launches. */
typedef struct {
    char name[64];
    int64_t launches;
    double ms, flops, bytes;
} mi355_prof_entry;
int mi355_profile_enable(mi355_unet_t net, int on);
int mi355_profile_read(mi355_unet_t net, mi355_prof_entry *out, int max_entries);

/* Single-op entry points (used by the parity tests; same kernels the network runs).
 * NDHWC fp32 device tensors. act: 0 none, 1 LeakyReLU(slope). */
int mi355_conv3d_ndhwc(const float *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                       const float *bias_host, int cout, int stride, int act, float slope, int impl,
                       float *y_dev, void *stream);
int mi355_tconv3d_ndhwc(const float *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                        int cout, float *y_dev, void *stream);
/* fp16-storage variants: x_dev / y_dev hold IEEE half plain NDHWC tensors (cin % 16 == 0, cout % 32 == 0); inside the library
 * fp16 activations are channel-blocked ([N][C/8][D][H][W][8]) and these entry points convert on the way in and out. */
int mi355_conv3d_ndhwc_f16(const void *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                           const float *bias_host, int cout, int stride, int act, float slope, void *y_dev,
                           void *stream);
int mi355_tconv3d_ndhwc_f16(const void *x_dev, int n, int d, int h, int w, int cin, const float *weight_host,
                            int cout, void *y_dev, void *stream);
/* One convolution WITH the run-time-norm statistics epilogue: y = act(conv(x) + b) as mi355_conv3d_ndhwc[_f16] (dtype =
 * MI355_F32 / MI355_F16; plain NDHWC operands of that dtype; the kernel the network would dispatch for this shape), and
 * sums_dev[n][cout][2] = (sum over voxels of y, sum of y^2) in fp64 - the two numbers nn.InstanceNorm3d / nn.GroupNorm in
 * ConvDropoutNormNonlin (generic_UNet.py:62-72) reduce their input to, accumulated by the conv kernel's epilogue from the
 * fp32 values BEFORE any rounding to fp16.  Test aid for the statistics instantiations of every conv kernel. */
int mi355_conv3d_sums_ndhwc(const void *x_dev, int dtype, int n, int d, int h, int w, int cin, const float *weight_host,
                            const float *bias_host, int cout, int stride, int act, float slope, void *y_dev,
                            double *sums_dev, void *stream);
/* Name of the kernel instantiation the calling thread's last mi355_conv3d_ndhwc / mi355_conv3d_ndhwc_f16 call dispatched
 * (the names rocprofv3 and mi355_profile_read show).  Test aid: a parity case written for one kernel can assert that it
 * ran on that kernel.  No reference counterpart (torch.nn.Conv3d, generic_UNet.py:56, has one implementation). */
const char *mi355_last_conv_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_NNUNET_H */
