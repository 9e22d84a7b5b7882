/* liblhn -- MI355X (gfx950) kernels for litehandnet's convolutional heatmap path.
 *
 * C ABI only: raw device pointers, explicit sizes, a hipStream_t (passed as void*), int status.
 * 0 = ok; nonzero = invalid argument / unsupported shape / HIP error, text via lhn_last_error().
 * Nothing here allocates or frees caller memory; every tensor is borrowed for the duration of
 * the call on the given stream.  The reference has no native interface -- each entry point cites
 * the Python function (file:line under the reference tree) whose arithmetic it replaces.
 *
 * Activation layout inside the library: NHWC fp32.  A buffer may carry a *pending* per-channel
 * transform (BatchNorm scale/shift + leaky slope, written by lhn_bn_finalize) and a per-(n,c)
 * gate (channel attention); consumers apply   v = gate[n,c] * lrelu_slope[c](scale[c]*raw+shift[c])
 * on load, so train-mode BatchNorm costs no extra pass over HBM.
 */
#ifndef LHN_H
#define LHN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LHN_VERSION 3
/* cross-block accumulators (BN statistics, BN-backward sums) are replicated to spread atomic traffic:
 * layout double[LHN_STAT_REPLICAS][2][C]; a block adds into replica (blockIdx % LHN_STAT_REPLICAS). */
#define LHN_STAT_REPLICAS 32

typedef struct lhn_view {
  float*       data;    /* base of the [N,H,W,cstride] buffer                                        */
  const float* table;   /* [3][cstride] scale | shift | slope, absolute channel index; NULL=identity */
  const float* gate;    /* [N][cstride] or NULL                                                      */
  int32_t N, H, W;
  int32_t cstride;      /* channels of the underlying buffer                                          */
  int32_t coff;         /* first channel of the view                                                  */
  int32_t C;            /* channels of the view                                                       */
  const void* pend;      /* reserved, must be NULL (round 2's deferred BatchNorm finalize is gone; the field keeps the layout) */
} lhn_view;

/* gradient-side companion of a BatchNorm'd conv output (see DESIGN.md "backward") */
typedef struct lhn_gradview {
  const float* dz;       /* gradient w.r.t. the consumed value z, same geometry as the forward view  */
  const float* dpool;    /* [N][25][cstride] channel-attention pooled gradient per bin-overlap segment
                          * (sum over the segment's bins of d(loss)/d(pooled)/|bin|), or NULL          */
  const float* coef;     /* [3][cstride]  A | B | C :  dy = A*du + B*y + C   (NULL: dy = du)          */
} lhn_gradview;

/* Optional fused BatchNorm finalize: the last workgroup of the conv launch to finish folds the replicated
 * statistics into the table (same arithmetic as lhn_bn_finalize) -- saves one launch per BatchNorm.
 * `counter` must be zero before the launch.  NULL pointer to the struct = not fused. */
typedef struct lhn_bnfin {
  uint32_t*    counter;
  const float* gamma;
  const float* beta;
  float*       running_mean;
  float*       running_var;
  int64_t*     num_batches_tracked;
  float*       table;
  float*       save_mean_invstd;
  double       count;
  int32_t      cstride, coff, C;
  float        eps, momentum, slope;
  const float* conv_bias;   /* bias of the convolution in front of the BatchNorm, or NULL (see lhn_bn_finalize) */
} lhn_bnfin;
/* same idea for lhn_bn_bwd_reduce -> lhn_bn_bwd_finalize */
typedef struct lhn_bnbwdfin {
  uint32_t*    counter;
  const float* gamma;
  float*       coef;
  float*       dgamma;
  float*       dbeta;
  double       count;
  int32_t      cstride, coff, C;
} lhn_bnbwdfin;

/* BatchNorm-backward sums contributed by a READER.  du = dz * act'(u) is linear in dz, and dz of a convolution output is the sum
 * of what its readers' backward kernels hand back -- so every reader that has its own part of dz in registers can add that
 * part's  sum du  and  sum du * xhat  into the PRODUCER's replicated sums (the reader holds the raw value and the table of its
 * input anyway, or re-reads a tile that is still in L2).  When all readers of a convolution + BatchNorm output do, its
 * lhn_bn_bwd_reduce pass (one read of y and one of dz) is not needed.  Valid for ungated inputs without pooled gradient.
 *   sums: [LHN_STAT_REPLICAS][2][C] of the producer's BatchNorm (zeroed by the caller), save: its [2][C] mean | invstd,
 *   coff: the channel of that BatchNorm which the reader's input view starts at.  NULL pointer to the struct: no sums. */
typedef struct lhn_bnsum {
  double*      sums;
  const float* save;
  int32_t      C, coff;
} lhn_bnsum;
/* Gradient contributions of OTHER readers of x that join the store of lhn_maxpool2_bwd3 (the hourglass skip tensor is read by
 * a 2x2 max-pool, an adaptive average pool and a plain sum: litehourglass.py:139-163, pose_hg_ms_att.py -- three read-modify-write
 * passes over its gradient otherwise).  same: gradient of a sum that reads x at its own resolution with coefficient 1 and no
 * activation behind it (x's geometry); pooled: gradient of adaptive_avg_pool2d(x, (OH, OW)) as [N][OH][OW][pooled_cstride].
 * Either may be NULL.  x.H and x.W must be even (every pixel of x sits in one pooling window). */
typedef struct lhn_grad_adds {
  const float* same;
  int32_t      same_cstride, same_coff;
  const float* pooled;
  int32_t      OH, OW, pooled_cstride, pooled_coff;
} lhn_grad_adds;

int         lhn_version(void);
/* 1 when the library runs in its deterministic mode (environment LHN_DETERMINISTIC=1, read once): every cross-workgroup sum
 * has one writer per replica and replicas are folded in a fixed order, so two runs on the same inputs agree bit for bit
 * (forward, gradients, running statistics).  Grids shrink to at most 16 workgroups: expect a 20-40x slower step. */
int         lhn_deterministic(void);
const char* lhn_last_error(void);
int         lhn_device_ok(void);              /* 0 if a gfx950 device is usable                       */

/* ---------------------------------------------------------------- heatmap encode / decode / loss
 * lhn_heatmap_encode    datasets/data_pipeline/generateTarget.py:74-159 (_msra_generate_target)
 * lhn_heatmap_argmax    utils/post_processing/evaluation/top_down_eval.py:199-231 (_get_max_preds)
 * lhn_heatmap_refine    .../top_down_eval.py:440-452 ('default' +-0.25 shift); mode 1 =
 *                       utils/heatmap_post_processing.py:6-33 / utils/HeatmapParser.py:197-223 twin
 * lhn_transform_preds   datasets/data_pipeline/post_transforms.py:6-48
 * lhn_heatmap_decode    .../top_down_eval.py:375-463 (keypoints_from_heatmaps, 'default'), fused
 * lhn_heatmap_nms       utils/HeatmapParser.py:41-50 (k x k max-pool peak keep)
 * lhn_pck_accuracy      .../top_down_eval.py:12-62,129-165
 * lhn_loss_balanced_mse loss/loss.py:93-114 + loss/heatmapLoss.py:242-265
 */
int lhn_heatmap_encode(const float* joints /*[N,K,3]*/, const float* visible /*[N,K,3]*/,
                       float* target /*[N,K,H,W]*/, float* weight /*[N,K]*/, int N, int K, int H, int W,
                       float img_w, float img_h, float sigma, int unbiased /*0 MSRA patch, 1 DARK full map, 2 UDP patch (generateTarget.py:160-236)*/,
                       void* stream);
int lhn_heatmap_argmax(const float* hm /*[N,K,H,W]*/, float* preds /*[N,K,2]*/, float* maxvals /*[N,K]*/,
                       int32_t* index /*[N,K] or NULL*/, int N, int K, int H, int W, void* stream);
int lhn_heatmap_refine(const float* hm, float* preds /*[N,K,2] in/out*/, int N, int K, int H, int W,
                       int mode, void* stream);
int lhn_transform_preds(const float* coords /*[N,K,2]*/, const float* center /*[N,2]*/,
                        const float* scale /*[N,2]*/, float* out /*[N,K,2]*/, int N, int K, int W, int H,
                        int use_udp, void* stream);
int lhn_heatmap_decode(const float* hm, const float* center, const float* scale, float* hm_preds,
                       float* preds, float* maxvals, int N, int K, int H, int W, int post_process,
                       void* stream);
/* GPU input path: TopDownAffine (datasets/data_pipeline/topdown_affine.py:47-114, non-UDP) +
 * ToTensor + NormalizeTensor (shared_transform.py:3-44) in one launch: uint8 HWC source images [N,Hs,Ws,3] -> normalised
 * float CHW crops [N,3,Ho,Wo]; optionally maps joints [N,K,3] (visible ones) with the same transform.  mean3/std3 are
 * HOST pointers to 3 floats.  cv2.warpAffine parity is unpinned (cv2 absent): exact-float bilinear, uint8 rounding. */
int lhn_affine_warp_normalize(const unsigned char* img, int N, int Hs, int Ws, const float* center /*[N,2]*/,
                              const float* scale /*[N,2]*/, const float* rot_deg /*[N]*/, const float* mean3,
                              const float* std3, float* out, int Ho, int Wo, float* joints /*or NULL*/,
                              const float* visible, int vis_stride, int K,
                              int use_udp /*get_warp_matrix + warp_affine_joints, post_transforms.py:49-100*/, void* stream);
/* TopDownRandomFlip (datasets/data_pipeline/RandomFlip.py:28-100) for the samples with flipped[n] != 0: lhn_random_flip
 * exchanges the joints / visibility of every (left, right) pair, mirrors x (W - 1 - x), multiplies by the visibility and
 * mirrors center_x; lhn_affine_warp_normalize2 then reads the source image of those samples mirrored (img[:, ::-1]) --
 * the flipped image is never written.  pairs: int32 [npairs][2] on the device. */
/* HSVRandomAug (datasets/data_pipeline/random_hsv.py:20-34) in place on uint8 BGR images [N,H,W,3]: gains = int16 [N][3] (hue,
 * saturation, value), drawn by the caller exactly as the reference draws them (litehandnet_amd.pipeline.hsv_gains).  The integer
 * jitter is the reference's; the 8-bit colour conversions follow OpenCV's algorithm (cv2 absent here: parity unpinned). */
int lhn_hsv_jitter(unsigned char* img, const int16_t* gains, int N, int H, int W, void* stream);
int lhn_random_flip(float* joints /*[N,K,3]*/, float* visible /*[N,K,vis_stride]*/, int vis_stride, float* center /*[N,2]*/,
                    const unsigned char* flipped /*[N]*/, const int32_t* pairs, int npairs, int N, int K, int img_width,
                    void* stream);
int lhn_affine_warp_normalize2(const unsigned char* img, int N, int Hs, int Ws, const float* center, const float* scale,
                               const float* rot_deg, const float* mean3, const float* std3, float* out, int Ho, int Wo,
                               float* joints, const float* visible, int vis_stride, int K, int use_udp,
                               const unsigned char* flipped /*[N] or NULL*/, void* stream);
/* SimDR (cfg.PIPELINE.simdr_split_ratio = k > 0): 1-D Gaussian target vectors (generate_simder.py:9-31) and the
 * auxiliary loss on the decoded vectors (centernet_simdr_loss.py:6-71: per joint, SmoothL1 'mean' over [N, L] times the
 * MEAN of that joint's weights, x and y, averaged over joints).  The two shared Linear decoders are plain library
 * GEMMs on the caller's side; decoding the vectors is lhn_heatmap_argmax on [N,K,1,L] + lhn_transform_preds.
 * sums = double[3*K] scratch shared by fwd and bwd. */
int lhn_simdr_encode(const float* joints /*[N,K,3]*/, const float* visible, int vis_stride, float* target_x /*[N,K,Wd]*/,
                     float* target_y /*[N,K,Hd]*/, int N, int K, int Wd, int Hd, float k, float sigma, void* stream);
int lhn_simdr_loss_fwd(const float* px, const float* py, const float* tx, const float* ty, const float* weight /*[N,K]*/,
                       double* sums, float* loss, int N, int K, int Wd, int Hd, void* stream);
int lhn_simdr_loss_bwd(const float* px, const float* py, const float* tx, const float* ty, const double* sums,
                       const float* gout, float* dpx, float* dpy, int N, int K, int Wd, int Hd, void* stream);
/* DARK 'unbiased' decode: k x k Gaussian modulation + log + Taylor step (top_down_eval.py:233-272,338-372,433-439;
 * twin utils/heatmap_post_processing.py:35-91), fused with argmax and transform_preds */
int lhn_heatmap_decode_dark(const float* hm, const float* center, const float* scale, float* hm_preds,
                            float* preds, float* maxvals, int N, int K, int H, int W, int kernel, void* stream);
/* UDP + DARK decode: keypoints_from_heatmaps(use_udp=True, GaussianHeatmap) = _get_max_preds + post_dark_udp
 * (top_down_eval.py:275-337, 404-411) + transform_preds(use_udp=True) (post_transforms.py:37-43) */
int lhn_heatmap_decode_dark_udp(const float* hm, const float* center, const float* scale, float* hm_preds,
                                float* preds, float* maxvals, int N, int K, int H, int W, int kernel, void* stream);
int lhn_heatmap_nms(float* hm /*in place*/, float* scratch /*same size*/, int N, int K, int H, int W,
                    int kernel, void* stream);
/* HeatmapParser.candidate_bbox (utils/HeatmapParser.py:52-85): per centre map [N,H,W] (after lhn_heatmap_nms) the k highest
 * peaks in descending order as candidates[N,k,5] = (x, y, w, h, confidence) in image pixels; size_maps [N,2,H,W] = the
 * region-averaged width / height ratio maps (NULL: w = h = 0). */
int lhn_heatmap_topk(const float* centre_maps, const float* size_maps, float* candidates, int N, int H, int W, int k,
                     float image_size, void* stream);
int lhn_pck_accuracy(const float* pred, const float* gt, const uint8_t* mask /*[N,K]*/,
                     const float* normalize /*[N,2]*/, float thr, float* acc /*[K]*/,
                     float* avg_cnt /*[2]: avg, cnt*/, int N, int K, void* stream);
/* acc: double[68]: [0..3] = {S_pos, S_neg, n_pos (exact integer < 2^53), unused} after the call,
 * [4..67] = 16 replicated partial-sum slots (spread the atomics); zeroed by the call */
int lhn_loss_balanced_mse_fwd(const float* out, const float* target, const float* weight /*[N,K]*/,
                              double* acc, float* loss /*[1]*/, int64_t NK, int64_t HW, float loss_weight,
                              int balance, void* stream);
int lhn_loss_balanced_mse_bwd(const float* out, const float* target, const float* weight,
                              const double* acc, const float* dloss /*[1] or NULL=1*/, float* dout,
                              int64_t NK, int64_t HW, float loss_weight, int balance, void* stream);

/* ---------------------------------------------------------------- conv building blocks (NHWC)
 * lhn_conv_pw_*    1x1 convolution of RepConv / nn.Conv2d            repblocks.py:8-44
 * lhn_conv_dw_*    depthwise k x k (dilation, stride) of RepConv     repblocks.py:8-44, liteHandNet.py:8-21
 * lhn_conv_stem_*  dense k x k on the 3-channel NCHW image           litehourglass.py:170, liteHandNet.py:175
 * lhn_conv_kxk_*   dense 3x3 (BasicBlock / BottleNeck)               liteHandNet.py:23-54
 * lhn_bn_finalize  train/eval BatchNorm2d statistics -> table        torch.nn.BatchNorm2d semantics
 * stats: double[LHN_STAT_REPLICAS][2][Cout] (sum, sum of squares), accumulated with atomics; caller zeroes.
 * weight gradients: `dw` may be replica 0 of `nrep` partial copies `rep_stride` floats apart (a block adds
 * into replica blockIdx % nrep; lhn_reduce_replicas folds them); nrep = 1 adds straight into dw.
 */
int lhn_conv_pw_fwd(const lhn_view* x, const float* w /*[Cout,Cin]*/, const float* bias /*or NULL*/,
                    const lhn_view* y, double* stats /*or NULL*/, int stride, float* y_nchw /*or NULL*/,
                    const lhn_bnfin* fin /*or NULL*/, void* stream);
/* Extended 1x1 entry points.  Any channel counts that are multiples of 4 (hourglassnet.py: 256; lite_hrnet.py: 20..320): the
 * library runs (<=128) x (<=128) channel slices itself (input slices accumulate into y; bias and BatchNorm statistics on the
 * last one).  opts (NULL = defaults):
 *   w_cols / w_rows   real shape of the weight tensor when the views are padded to a multiple of 4 (the 21-joint head of
 *                     hourglassnet.py:117 stored as 24 NHWC channels, and merge_preds :119 reading them back);
 *   nchw_batch_stride floats between two images of the NCHW tensor (the stacked [N, num_stack, K, H, W] output, :136). */
typedef struct lhn_pw_opts {
  int32_t w_cols, w_rows;
  int64_t nchw_batch_stride;
  /* forward only: the consumed input is coef[0]*value(x) + sum_e coef[e+1]*value(extra[e]) -- residual sums taken on load
   * (litehourglass.py:41-49) instead of being materialised by lhn_ew_fwd.  Same geometry and channel count as x. */
  int32_t n_extra;              /* 0..2 */
  const lhn_view* extra;
  float coef[3];
  /* with extra sources: the summed input is ALSO written here (plain values; same pixels and channel count as x).  A
   * training forward leaves the sum where the backward's weight gradient reads it -- one write instead of a separate
   * elementwise pass that re-reads every operand.  NULL: not written. */
  const lhn_view* sum_out;
} lhn_pw_opts;
int lhn_conv_pw_fwd2(const lhn_view* x, const float* w, const float* bias, const lhn_view* y, double* stats, int stride,
                     float* y_nchw, const lhn_bnfin* fin, const lhn_pw_opts* opts, void* stream);
int lhn_conv_dw_fwd(const lhn_view* x, const float* w /*[C,1,k,k]*/, const lhn_view* y, double* stats,
                    int k, int stride, int pad, int dil, const lhn_bnfin* fin, void* stream);
/* depthwise 3x3 (stride 1, 'same' padding, dilation 1/2) over coef2[0]*value(x) + coef2[1]*value(extra): MSRB's second
 * round reads `out + ca(cat)` (litehourglass.py:41-45) without an elementwise pass in between.  extra == NULL: lhn_conv_dw_fwd. */
int lhn_conv_dw_fwd2(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride, int pad, int dil,
                     const lhn_bnfin* fin, const lhn_view* extra, const float* coef2 /*host, 2 floats*/, void* stream);
/* ... and sum_out (or NULL): the summed input is also written there, see lhn_pw_opts.sum_out. */
int lhn_conv_dw_fwd3(const lhn_view* x, const float* w, const lhn_view* y, double* stats, int k, int stride, int pad, int dil,
                     const lhn_bnfin* fin, const lhn_view* extra, const float* coef2, const lhn_view* sum_out, void* stream);
int lhn_conv_stem_fwd(const float* img /*[N,3,Hi,Wi]*/, const float* w /*[Cout,3,k,k]*/, const lhn_view* y,
                      double* stats, int Hi, int Wi, int k, int stride, int pad, const lhn_bnfin* fin, void* stream);
/* wt_scratch: optional 9*Cout*Cin floats of caller-owned scratch; the call re-lays the OIHW weights tap-major into it
 * (16-byte weight loads per tap instead of 36-byte-strided gathers).  NULL = read the OIHW tensor directly. */
int lhn_conv_kxk_fwd(const lhn_view* x, const float* w /*[Cout,Cin,3,3]*/, const lhn_view* y, double* stats,
                     int stride, const lhn_bnfin* fin, float* wt_scratch, void* stream);
/* conv_bias: a biased convolution followed by BatchNorm (models/pose_hg_ms_att.py:27-35,46-56) stores its output
 * WITHOUT the bias -- BatchNorm cancels it -- and the bias only enters the running mean (training) or the shift
 * (eval: beta - (running_mean - bias) * scale).  d(bias) is identically zero in training mode. */
int lhn_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* table, int cstride, int coff,
                    int C, float* save_mean_invstd /*[2][C]*/, double count, float eps, float momentum,
                    float slope, int training, const float* conv_bias /*or NULL*/, void* stream);
/* stat_channels >= C: the statistics / save arrays were laid out for a view padded to a multiple of 4 (lite_hrnet.py:84-96:
 * BatchNorm over 7 / 17 / 37 channels behind a 1x1 whose output view has 8 / 20 / 40) */
int lhn_bn_finalize2(const double* stats, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float* table, int cstride, int coff, int C, int stat_channels,
                     float* save_mean_invstd, double count, float eps, float momentum, float slope, int training,
                     const float* conv_bias, void* stream);
int lhn_table_fill(float* table, int cstride, int coff, int C, float scale, float shift, float slope,
                   void* stream);

/* elementwise / pooling (liteHandNet.py:88-113, litehourglass.py:136-163, common.py:40-66) */
/* pending transform of a deployed (biased, BN-free) convolution: table slice = (1, bias[c] or 0, slope)
 * -- RepConv.forward with rep_conv, repblocks.py:41-43 */
int lhn_table_bias(float* table, int cstride, int coff, int C, const float* bias /*or NULL*/, float slope,
                   void* stream);
/* deploy-time re-parameterisation (repblocks.py:46-73,169-236; common.py:68-90): one branch of the fused kernel.
 * out_w[Cout][cin_g][k][k] (store | +=) branch * gamma/sqrt(rvar+eps), branch = w (kb x kb centred in k x k) or the
 * identity kernel when w == NULL; out_b[Cout] (store | +=) beta - rmean*gamma/sqrt(rvar+eps).  Equals the IEEE float32
 * evaluation of the reference's formula bit for bit when the branches are added in the order 1x1, identity, k x k. */
int lhn_fold_bn(const float* w, int kb, const float* gamma, const float* beta, const float* rmean,
                const float* rvar, float eps, float* out_w, float* out_b, int Cout, int cin_g, int k,
                int accumulate, void* stream);
/* out_slope == LHN_SLOPE_SILU selects SiLU instead of a leaky ReLU as the combine's output activation (the
 * BN -> SiLU -> conv pre-activation unit of models/pose_hg_ms_att.py:76-90; single same-size source in backward) */
#define LHN_SLOPE_SILU 2.0f
/* out_slope == LHN_SLOPE_RELU_SIGMOID: sigmoid(relu(v)) -- the nn.ReLU + nn.Sigmoid pairs of lite_hrnet.py:62-70,86-97
 * (single same-size source in backward, like SiLU) */
#define LHN_SLOPE_RELU_SIGMOID 3.0f
int lhn_ew_fwd(const lhn_view* srcs, int nsrc, const lhn_view* dst, float out_slope, void* stream);
/* dst = act(sum_i coef[i] * value_i); coef == NULL: all ones (host pointer, nsrc floats) */
int lhn_ew_fwd2(const lhn_view* srcs, int nsrc, const float* coef, const lhn_view* dst, float out_slope, void* stream);
/* mode bit 0: PRODUCT of the sources instead of their sum (lite_hrnet.py:105-107, s * F.interpolate(a, nearest));
 * mode bit 1: smaller sources are resampled bilinearly with align_corners=True (lite_hrnet.py:272-275) instead of nearest.
 * Backward of the product: lhn_ew_mul_bwd per operand; of a bilinear source: lhn_bilinear_bwd (same-size sources of
 * that combine: lhn_ew_bwd2 as usual). */
int lhn_ew_fwd3(const lhn_view* srcs, int nsrc, const float* coef, const lhn_view* dst, float out_slope, int mode, void* stream);
int lhn_ew_mul_bwd(const lhn_view* src, const lhn_view* other, const lhn_view* dst, const float* ddst, float* dsrc,
                   int accumulate, void* stream);
int lhn_bilinear_bwd(const lhn_view* src, const lhn_view* dst, const float* ddst, float* dsrc, int accumulate, float out_slope,
                     void* stream);
/* channel_shuffle(torch.cat([a, b], 1), groups = 2) (lite_hrnet.py:29-52,141-142,246-247): dst[2j] = value(a)[j],
 * dst[2j+1] = value(b)[j]; dst stored plain.  Backward hands d(dst) back to the operands' gradient buffers. */
int lhn_shuffle2_fwd(const lhn_view* a, const lhn_view* b, const lhn_view* dst, void* stream);
int lhn_shuffle2_bwd(const lhn_view* a, const lhn_view* b, const lhn_view* dst, const float* ddst, float* da, int acc_a,
                     float* db, int acc_b, void* stream);
int lhn_maxpool2_fwd(const lhn_view* x, const lhn_view* y, void* stream);
int lhn_avgpool_fwd(const lhn_view* x, float* out /*[N,OH,OW,x.C]*/, int OH, int OW, void* stream);
/* ... into channels [out_coff, out_coff + x.C) of an [N,OH,OW,out_cstride] tensor: the pooled branches of
 * CrossResolutionWeighting land side by side, torch.cat needs no copy (lite_hrnet.py:99-102) */
int lhn_avgpool_fwd2(const lhn_view* x, float* out, int OH, int OW, int out_cstride, int out_coff, void* stream);
int lhn_avgpool_bwd2(const lhn_view* x, const float* dout, int OH, int OW, int out_cstride, int out_coff, float* dx,
                     int dx_accumulate, void* stream);
/* gamma == NULL: deployed attention, `beta` is the bias of the fused depthwise conv, no BatchNorm (eval only) */
int lhn_ca_mlp_fwd(const float* pooled /*[N,9,C]*/, const float* w3 /*[C,1,3,3]*/, const float* gamma,
                   const float* beta, float* rmean, float* rvar, int64_t* nbt, const float* w1 /*[C/2,C]*/,
                   const float* b1, const float* w2 /*[C,C/2]*/, const float* b2, const float* dropmask,
                   float* gate, int gate_stride, int gate_coff, float* save /*see DESIGN*/, int N, int C,
                   float eps, float momentum, int training, int stage, double* gsum, double count_scale,
                   void* stream);
/* SyncBatchNorm (train/spawn_dist.py:37-38) support in the attention kernels: stage 0 = the whole op; stage 1 = up to
 * the local statistics sums, written to gsum[2][C] for the caller to all-reduce; stage 2 = the rest, with statistics
 * over N*count_scale samples.  Backward: pgrad_scale (1/world) scales d(gamma), d(beta), which come from global sums. */

/* attention of `mynet` (models/pose_hg_ms_att.py:165-174,191-192) on the pooled [N,9,C] tensor:
 * gate = sigmoid(Linear(dropout(dw3x3(relu(BN(pooled))) + b3))); save = floats[3*N*C + 2*C] */
int lhn_att_mlp_fwd(const float* pooled, const float* gamma, const float* beta, float* rmean, float* rvar,
                    int64_t* nbt, const float* w3 /*[C,1,3,3]*/, const float* b3, const float* wl /*[C,C]*/,
                    const float* bl, const float* dropmask /*[N,C] or NULL*/, float* gate, int gate_stride,
                    int gate_coff, float* save, int N, int C, float eps, float momentum, int training, int stage,
                    double* gsum, double count_scale, void* stream);
int lhn_att_mlp_bwd(const float* pooled, const float* gamma, const float* beta, const float* w3, const float* wl,
                    const float* dropmask, float* save, const float* dgate /*[N,C]*/, float* dpool, int cstride,
                    int coff, int H, int W, float* dgamma, float* dbeta, float* dw3, float* db3, float* dwl,
                    float* dbl, int N, int C, int stage, double* gsum, double count_scale, float pgrad_scale,
                    void* stream);

/* squeeze-and-excitation gate (models/pose_estimation/liteHandNet/common.py:23-37; ca_type / msrb_ca / rbu_ca = 'se'):
 * gate = sigmoid(up(relu(down(pooled)))), pooled = lhn_avgpool_fwd(y, 1, 1) [N,C]; J = internal neurons; save = floats[N*J + N*C] */
int lhn_se_mlp_fwd(const float* pooled, const float* w1 /*[J,C]*/, const float* b1, const float* w2 /*[C,J]*/,
                   const float* b2, float* gate, int gate_stride, int gate_coff, float* save, int N, int C, int J,
                   void* stream);
/* mode 1: SpatialWeighting of lite_hrnet.py:55-74 -- sigmoid(relu(.)) after both 1x1 convolutions (mode 0 = SEBlock) */
int lhn_se_mlp_fwd2(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                    int gate_stride, int gate_coff, float* save, int N, int C, int J, int mode, void* stream);
int lhn_se_mlp_bwd2(const float* pooled, const float* w1, const float* w2, const float* save, const float* dgate,
                    float* dpool, int cstride, int coff, int H, int W, float* dw1, float* db1, float* dw2, float* db2, int N,
                    int C, int J, int mode, void* stream);
int lhn_se_mlp_bwd(const float* pooled, const float* w1, const float* w2, const float* save, const float* dgate /*[N,C]*/,
                   float* dpool /*[N,25,cstride]*/, int cstride, int coff, int H, int W, float* dw1, float* db1, float* dw2,
                   float* db2, int N, int C, int J, void* stream);

/* backward building blocks */
int lhn_bn_bwd_reduce(const lhn_view* y, const lhn_gradview* g, const float* save_mean_invstd,
                      double* sums /*[R][2][C]: sum du, sum du*xhat*/, const lhn_bnbwdfin* fin /*or NULL*/, void* stream);
int lhn_bn_bwd_finalize(const double* sums, const float* gamma, const float* save_mean_invstd,
                        float* coef, int cstride, int coff, int C, double count, float* dgamma, float* dbeta,
                        float pgrad_scale /*1, or 1/world under SyncBatchNorm*/, void* stream);
int lhn_bn_bwd_finalize2(const double* sums, const float* gamma, const float* save_mean_invstd, float* coef, int cstride,
                         int coff, int C, int stat_channels, double count, float* dgamma, float* dbeta, float pgrad_scale,
                         void* stream);
int lhn_conv_pw_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy,
                    float* dx /*grad buffer of x, same geometry, or NULL*/, int dx_accumulate, float* dw,
                    float* dbias, int stride, const float* dy_nchw, int nrep, int64_t rep_stride, void* stream);
int lhn_conv_pw_bwd2(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                     int dx_accumulate, float* dw, float* dbias, int stride, const float* dy_nchw, int nrep, int64_t rep_stride,
                     const lhn_pw_opts* opts, void* stream);
int lhn_conv_dw_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                    int dx_accumulate, float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride,
                    void* stream);
/* ... and, when this convolution is the only reader of x and x is the output of a convolution + BatchNorm, the
 * BatchNorm-backward sums of THAT producer (sum du, sum du * xhat into bn_sums[LHN_STAT_REPLICAS][2][bn_C] at channel
 * bn_coff, mean / invstd from bn_save) -- the kernel has d(x), the raw x and its table in hand; lhn_bn_bwd_reduce of the
 * producer is then skipped.  3x3, stride 1, dilation 1, dx stored (not accumulated), x ungated.  bn_sums NULL: lhn_conv_dw_bwd. */
int lhn_conv_dw_bwd2(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride, double* bn_sums,
                     const float* bn_save, int bn_C, int bn_coff, void* stream);
/* ... and with up to two gradient ADDENDS: dx = [dx +] dgrad + dx_add0 + dx_add1, where dx_add* point at buffers of dx's
 * geometry (same pixel stride; the pointer already carries any channel shift).  They are the output gradients of residual
 * sums that read x (litehourglass.py:41-49: `out = out + ca(cat)`): the sum's backward costs no pass of its own.  Needs the
 * tiled kernel (stride 1, "same" padding, W >= 8).  Both NULL: lhn_conv_dw_bwd. */
int lhn_conv_dw_bwd3(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, int k, int stride, int pad, int dil, int nrep, int64_t rep_stride, const float* dx_add0,
                     const float* dx_add1, void* stream);
int lhn_conv_stem_bwd(const float* img, const lhn_view* y, const lhn_gradview* gy, float* dw, int Hi, int Wi,
                      int k, int stride, int pad, int nrep, int64_t rep_stride, void* stream);
/* NOTE: lhn_conv_kxk_bwd and the large-channel path of lhn_conv_pw_bwd CONSUME gy->dz (it is overwritten in place
 * with dy before the MFMA kernels stream it). */
int lhn_conv_kxk_bwd(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx,
                     int dx_accumulate, float* dw, int stride, int nrep, int64_t rep_stride,
                     float* wt_scratch /*9*Cout*Cin floats or NULL, as above (dgrad layout)*/, void* stream);
/* out[i] = sum_r part[r*rep_stride + i]  (folds the replicated weight-gradient partials into the flat gradient) */
/* SyncBatchNorm (train/spawn_dist.py:37-38): run half-steps [step_begin, step_end) of a phase -- step 2*i = main
 * launches of op i, step 2*i+1 = the consumer of its statistics.  The caller all-reduces (SUM) the op's statistics
 * buffer between the two; count_scale = world size, pgrad_scale = 1/world (for d(gamma), d(beta)). */
int lhn_plan_run_range(void* plan, int phase, int64_t step_begin, int64_t step_end, void* workspace,
                       void* const* params, void* const* grads, void* const* io, int training, int grad_replicas,
                       int64_t grad_rep_stride, double count_scale, float pgrad_scale, void* stream);
/* SyncBatchNorm: stats[nrep][n] -> replica 0 holds the sum over the replicas, the others are zeroed; the caller then all-reduces
 * only the first n doubles (32x fewer bytes per BatchNorm than the replicated layout) and the finalize launches run unchanged. */
int lhn_fold_stat_replicas(double* stats, int64_t n, int nrep, void* stream);
int lhn_reduce_replicas(float* out, const float* part, int64_t n, int nrep, int64_t rep_stride, void* stream);
int lhn_ew_bwd(const lhn_view* srcs, int nsrc, const lhn_view* dst, const float* ddst, float out_slope,
               float* const* dsrcs, const int* accumulate, void* stream);
/* single-source form used by the plan: dst may carry a gate and a pooled gradient (channel attention) */
int lhn_ew_bwd2(const lhn_view* src, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope,
                float* dsrc, int accumulate, void* stream);
int lhn_maxpool2_bwd(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate,
                     void* stream);
/* the same readers' backward kernels, also adding their part of the producer's BatchNorm-backward sums (lhn_bnsum; NULL = plain) */
int lhn_maxpool2_bwd2(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate, const lhn_bnsum* bns,
                      void* stream);
int lhn_maxpool2_bwd3(const lhn_view* x, const lhn_view* y, const float* dy, float* dx, int dx_accumulate, const lhn_bnsum* bns,
                      const lhn_grad_adds* adds /*or NULL*/, void* stream);
int lhn_ew_bwd3(const lhn_view* src, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope, float* dsrc,
                int accumulate, const lhn_bnsum* bns, void* stream);
/* the same for 1..3 sources that all have dst's own geometry, in ONE pass over d(dst) / dst (residual sums); bns: per-source
 * lhn_bnsum pointers or NULL */
int lhn_ew_bwd_multi(const lhn_view* srcs, int nsrc, const lhn_view* dst, const float* ddst, const float* dst_dpool, float out_slope,
                     float* const* dsrcs, const int* accumulate, const lhn_bnsum* const* bns, void* stream);
int lhn_avgpool_bwd3(const lhn_view* x, const float* dout, int OH, int OW, int out_cstride, int out_coff, float* dx,
                     int dx_accumulate, const lhn_bnsum* bns, void* stream);
int lhn_conv_pw_bwd3(const lhn_view* x, const float* w, const lhn_view* y, const lhn_gradview* gy, float* dx, int dx_accumulate,
                     float* dw, float* dbias, int stride, const float* dy_nchw, int nrep, int64_t rep_stride,
                     const lhn_pw_opts* opts, const lhn_bnsum* bns /*fused kernel only: Cin * Cout < 64 * 128, stride 1*/, void* stream);
int lhn_avgpool_bwd(const lhn_view* x, const float* dout /*[N,OH,OW,C]*/, int OH, int OW, float* dx,
                    int dx_accumulate, void* stream);
int lhn_gate_bwd_reduce(const lhn_view* y, const float* dz, float* dgate /*[N][C] dense*/, void* stream);
/* BatchNorm-backward sums of a GATED buffer without a pass of their own (the gate gradient needs one pass over (y, dz), the
 * pooled gradient of the attention exists only after it, and lhn_bn_bwd_reduce would be a second pass over both).  With
 *   du = (g[n,c] * dz + sum over the bins containing the pixel of e[n,bin,c]) * act'(u),   e = d loss / d pooled / |bin|
 * the sums split per sample:  sum du = sum_n (g * T0 + sum_bins e * M0),  sum du * xhat = sum_n (g * T1 + sum_bins e * M1)  with
 *   T0 = sum_px dz * act'(u), T1 = sum_px dz * act'(u) * xhat     per (n, c):      lhn_gate_bwd_reduce2 (tsum[N][2][C], = dgate + N*C)
 *   M0 = sum_{px in bin} act'(u), M1 = sum_{px in bin} act'(u) * xhat  per (n, bin, c): lhn_avgpool_fwd3 (pstat[N*9][2][C], forward)
 * and lhn_ca_mlp_bwd2 assembles them right where e is formed and stores them into replica 0 of the producers' sums.
 * slices: the (up to two) convolution + BatchNorm outputs that make up the buffer (common.py:40-66 gates cat(left, right)). */
typedef struct lhn_bn_slices {
  const float* save[2];   /* [2][C[k]] mean | invstd saved by lhn_bn_finalize                      */
  double*      sums[2];   /* [LHN_STAT_REPLICAS][2][C[k]] (zeroed by the caller) or NULL; lhn_ca_mlp_bwd2 only */
  int32_t      lo[2], C[2];
  int32_t      n;         /* 0..2 */
} lhn_bn_slices;
int lhn_gate_bwd_reduce2(const lhn_view* y, const float* dz, float* dgate /*[3][N][C]: dgate | tsum*/, float* tsum,
                         const lhn_bn_slices* slices, void* stream);
/* prezeroed != 0: dgate (and tsum) are already zero (the caller zeroes many such buffers with one memset) */
int lhn_gate_bwd_reduce3(const lhn_view* y, const float* dz, float* dgate, float* tsum, const lhn_bn_slices* slices, int prezeroed,
                         void* stream);
int lhn_avgpool_fwd3(const lhn_view* x, float* out /*[N,OH,OW,C]*/, int OH, int OW, float* pstat /*[N*OH*OW][2][C] or NULL*/,
                     const lhn_bn_slices* slices, void* stream);
int lhn_ca_mlp_bwd2(const float* pooled, const float* w3, const float* gamma, const float* w1, const float* w2,
                    const float* dropmask, const float* save, const float* dgate, float* dpool, int cstride, int coff, int H,
                    int W, float* dw3, float* dgamma, float* dbeta, float* dw1, float* db1, float* dw2, float* db2, int N, int C,
                    int stage, double* gsum, double count_scale, float pgrad_scale, const float* tsum, const float* pstat,
                    const lhn_bn_slices* slices, void* stream);
/* ... and copy_src (or NULL): channels [0, copy_src->C) of the pooled buffer x have not been written yet -- they are the
 * pass-through half of a gated RepBasicUnit (litehourglass.py:74-77: ca(cat(left, conv(right)))).  The kernel takes their consumed
 * values from copy_src, pools them and stores them into x: no separate copy pass.  pstat may be NULL here. */
int lhn_avgpool_fwd4(const lhn_view* x, float* out, int OH, int OW, float* pstat, const lhn_bn_slices* slices,
                     const lhn_view* copy_src, void* stream);
int lhn_ca_mlp_bwd(const float* pooled, const float* w3, const float* gamma, const float* w1, const float* w2,
                   const float* dropmask, const float* save, const float* dgate, float* dpool /*[N,9,cs]*/,
                   int cstride, int coff, int H, int W, float* dw3, float* dgamma, float* dbeta, float* dw1,
                   float* db1, float* dw2, float* db2, int N, int C, int stage, double* gsum, double count_scale,
                   float pgrad_scale, void* stream);

/* torch.optim.Adam (no amsgrad) over one flat fp32 parameter buffer -- the optimizer of dist_train.py:64-69 for the flat parameter
 * of litehandnet_amd.train.FlatParams: exp_avg / exp_avg_sq are the optimizer state of that one tensor, step = the 1-based count of
 * this update.  Buffers 16-byte aligned. */
int lhn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, void* stream);

/* ---------------------------------------------------------------- plan executor
 * A plan is a static list of the calls above over one workspace arena, built by the Python
 * mirror of the reference's nn.Module tree (models/__init__.py:20-26 get_model).  One
 * lhn_plan_run() enqueues a whole forward (phase 0) or backward (phase 1) on the stream. */
typedef struct lhn_op {
  int32_t kind;
  int32_t in_buf[3], in_coff[3], in_C[3];
  int32_t reserved[6]; /* (round 2: deferred-finalize indices) */
  int32_t out_buf, out_coff, out_C;
  int32_t p[12];       /* parameter / state indices into the params array, -1 = none              */
  int64_t ws[12];      /* byte offsets into the workspace, -1 = none                               */
  int32_t i[8];
  float   f[8];        /* [0..3] op scalars (eps, momentum, slope, ..); [4..6] coefficients of summed input sources */
} lhn_op;

typedef struct lhn_buf {
  int64_t data_off, table_off, gate_off, grad_off, dpool_off, coef_off;  /* bytes, -1 = none */
  int32_t N, H, W, C;
} lhn_buf;

void* lhn_plan_create(const lhn_buf* bufs, int nbufs, const lhn_op* fwd, int nfwd, const lhn_op* bwd, int nbwd);
void  lhn_plan_destroy(void* plan);
/* io: phase 0 {image NCHW, heatmap NCHW out}; phase 1 {image NCHW, d(heatmap) NCHW}.
 * training: a bit set -- bit 0 = train-mode BatchNorm (batch statistics, running statistics updated); bit 1 (LHN_RUN_TABLES_CURRENT,
 * eval only) = the workspace's per-buffer BatchNorm tables were built by an earlier eval run of THIS plan and no parameter
 * or running statistic changed since: the launches that only rebuild them are skipped. */
#define LHN_RUN_TABLES_CURRENT 2
/* bit 2: the launch sequence of this (phase, pointer set) may be captured into a hipGraph on its second sighting and replayed
 * afterwards (one graph launch instead of hundreds of kernel launches: Lite-HRNet runs 2,600 small launches per step and is
 * launch-bound).  Needs a real stream (the legacy default stream cannot be captured); a plan whose pointers never repeat
 * falls back to plain launches.  LHN_GRAPH=1 in the environment sets it for every call. */
#define LHN_RUN_GRAPH 4
int   lhn_plan_run(void* plan, int phase, void* workspace, void* const* params, void* const* grads,
                   void* const* io, int training, int grad_replicas, int64_t grad_rep_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif
