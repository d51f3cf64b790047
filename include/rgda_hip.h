/*
 * rgda_hip.h -- C ABI of librgda_hip.so: the MI355X (gfx950) kernels behind the
 * RegDA self-training (SSL) step.  One process per GPU; every entry point only
 * ENQUEUES work on the caller's HIP stream (no host sync, no allocation, no
 * retained pointers, no global mutable state).  All buffers are device memory
 * owned by the caller (PyTorch tensors' data_ptr()).
 *
 * The reference (StuLiu/RegDA) has no FFI: its operator interface for this path
 * is a set of Python callables.  Each entry point below names the reference
 * callable (file:line under /root/reference) it replaces; the Python mirror of
 * those callables lives in regda_amd/ and calls these symbols through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Return value: 0 on success, negative rgda_status otherwise (rgda_strerror()).
 * Layout vocabulary: "NCHW f32" = the reference's tensors at the API boundary;
 * "PxC bf16" = the internal pixel-major activation matrix [N*H*W][ld] with
 * channels contiguous (row stride ld elements), bf16.
 */
#ifndef RGDA_HIP_H
#define RGDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGDA_ABI_VERSION 10
/* Per-channel statistics are accumulated into RGDA_STAT_REPLICAS interleaved copies (workgroup b adds to
 * copy b % 8, i.e. the copy of the XCD it runs on, so the atomics stay inside one XCD's L2); consumers
 * sum the copies.  A "stats"/"sums" buffer is therefore rgda_stat_t[RGDA_STAT_REPLICAS][2][C], zeroed by the caller.
 *
 * rgda_stat_t is 64-bit FIXED POINT (ABI 4; fp32 before): every workgroup reduces its rows in a fixed order, converts
 * the partial sum to round(v * 2^FRAC) and adds it with an integer atomic.  Integer addition is associative, so the
 * totals -- and everything computed from them: BatchNorm outputs, running statistics, BatchNorm gradients -- do not
 * depend on the order in which workgroups retire: two runs of the same step are bit-identical (fp32 atomics made the
 * batch statistics differ in the last bits from run to run, which moved thresholded pseudo labels).
 *   forward  (sum y, sum y^2 of conv outputs):  FRAC = 26 -> resolution 1.5e-8 per partial (below BatchNorm's eps by
 *            three orders of magnitude after the division by the row count), |total| < 1.4e11 (a partial is clamped at 2^61 fixed-point units)
 *   backward (sum g', sum g' xhat):             FRAC = 40 -> resolution 9e-13 per partial, |total| < 8.4e6
 * Partials outside the range are clamped, non-finite partials add nothing (the non-finite ELEMENTS still propagate
 * through the element-wise passes and the loss). */
#define RGDA_STAT_REPLICAS 8
#define RGDA_STAT_FRAC_FWD 26
#define RGDA_STAT_FRAC_BWD 40
typedef int64_t rgda_stat_t;
#define RGDA_LAYOUT_TILE 64      /* tile edge of rgda_weight_transpose_batched (block accounting of its table) */

typedef void* rgda_stream_t; /* hipStream_t */

enum rgda_status {
    RGDA_OK = 0,
    RGDA_ERR_ARG = -1,       /* bad shape / null pointer / unsupported parameter */
    RGDA_ERR_WORKSPACE = -2, /* workspace too small */
    RGDA_ERR_LAUNCH = -3,    /* hipGetLastError() after launch != hipSuccess */
    RGDA_ERR_UNSUPPORTED = -4
};

int rgda_abi_version(void);
const char* rgda_strerror(int status);

/* ------------------------------------------------------------------ labels */

/* pseudo_selection(mask, cutoff_top, cutoff_low, 'tensor', ignore_label)
 *   regda/gast/pseudo_generation.py:59-93.
 * soft: NCHW f32 (b,c,hw) probabilities.  out: (b,hw) int64.
 * ws: >= rgda_pseudo_select_workspace(b,c) bytes; on return (stream order)
 *   ws[0 .. b*c) f32 = per-image per-class max, then one int32 flag word:
 *   bit0 = some value < 0 or > 1 (the reference asserts, :71).
 * If classmax_ready != 0 the per-class maxima already sit in ws (written by
 * rgda_label_refine) and the reduction pass is skipped. */
size_t rgda_pseudo_select_workspace(int b, int c);
int rgda_pseudo_select(const float* soft, int64_t* out, int b, int c, int hw, float cutoff_top,
                       float cutoff_low, int ignore_label, int classmax_ready, void* ws,
                       size_t ws_bytes, rgda_stream_t stream);

/* Homogenizer.forward(pseudo_labels, regions)  (LRH)
 *   regda/utils/local_region_homog.py:125-152 (+ torch_scatter.scatter sum, :140).
 * labels, regions, out: (b,hw) int64.  Region ids must lie in [0, max_regions);
 * pixels with ids outside are left unchanged and flag bit0 is set; labels
 * outside [0,class_num) other than ignore_label set bit1 (the reference's
 * one_hot would raise).  ws layout: int32 hist[b][max_regions][class_num],
 * int32 ids[b][max_regions], int32 flag.  Bit-exact vs the reference. */
size_t rgda_lrh_workspace(int b, int max_regions, int class_num);
int rgda_lrh(const int64_t* labels, const int64_t* regions, int64_t* out, int b, int hw,
             int class_num, int ignore_label, float percent, int max_regions, void* ws,
             size_t ws_bytes, rgda_stream_t stream);

/* pseudo_selection followed by Homogenizer.forward in one pass over the soft labels -- the chain of the SSL step
 * (tools/train_ssl_reg.py:224-228): out = LRH(pseudo_selection(soft), regions) exactly as the two calls above give it
 * (bit-exact), without the intermediate int64 label tensor.  classmax: f32 [b][c] per-image per-class maxima of `soft`
 * (what rgda_label_refine leaves in its workspace, or rgda_pseudo_select's ws[0 .. b*c)).  class_num == 6, hw % 4 == 0,
 * max_regions <= 65535 (RGDA_ERR_UNSUPPORTED otherwise: use the two calls).  ws: rgda_pseudo_lrh_workspace bytes,
 * 16-byte aligned: int32 hist[b][R][C], int32 ids[b][R], int32 flag (bit0: a region id outside [0, R)), then scratch. */
size_t rgda_pseudo_lrh_workspace(int b, int hw, int max_regions, int class_num);
int rgda_pseudo_lrh(const float* soft, const float* classmax, const int64_t* regions, int64_t* out, int b, int hw,
                    int class_num, float cutoff_top, float cutoff_low, int ignore_label, float percent,
                    int max_regions, void* ws, size_t ws_bytes, rgda_stream_t stream);

/* SAM.get_local_regions, the region-map assembly only   regda/utils/local_region_homog.py:51-56.
 * masks: uint8 [K][HW] (non-zero = inside), the automatic mask generator's masks in ITS order; areas int64 [K];
 * regions int32 [HW] out: 1 + the last mask index with area >= area_threshold covering the pixel, 0 where none does
 * (later masks overwrite earlier ones).  The generator itself (third-party segment_anything) is not part of this
 * library.  Bit-exact. */
int rgda_masks_to_regions(const uint8_t* masks, const int64_t* areas, int32_t* regions, int K, int64_t HW,
                          int64_t area_threshold, rgda_stream_t stream);

/* Aligner.label_refine(None, feat_t, [p1,p2], soft, refine=1, mode='all', temp)
 *   regda/gast/alignment.py:194-265 (+ _pearson_dist :396-423, _softmax_T, _logits_norm).
 * feat: NCHW f32 (b,k,h,w); protos (c,k) f32; p1,p2: NCHW f32 (b,c,h,w);
 * soft/out: NCHW f32 (b,c,H,W) (out may alias soft).  ws >= workspace bytes:
 * f32 sim[b][c][h*w] then f32 classmax[b][c] (+ int32 flag) laid out so that
 * `classmax` can be handed to rgda_pseudo_select (returned offset). */
size_t rgda_label_refine_workspace(int b, int c, int h, int w);
size_t rgda_label_refine_classmax_offset(int b, int c, int h, int w);
int rgda_label_refine(const float* feat, const float* protos, const float* p1, const float* p2,
                      const float* soft, float* out, int b, int k, int c, int h, int w, int H,
                      int W, float temp, void* ws, size_t ws_bytes, rgda_stream_t stream);
/* The other `mode`s of label_refine with label_t_sup=None (alignment.py:199,212-236): `views` bit 0 = prototype view
 * ('p'), bit 1 = prediction view ('l'), 3 = 'all' (= rgda_label_refine).  A view that is not asked for needs no
 * inputs: feat / protos may be NULL for views == 2, p1 / p2 for views == 1.  A single prediction tensor
 * (alignment.py:232-234) is passed as p1 == p2: (s + s) * 0.5 is s exactly. */
int rgda_label_refine_views(const float* feat, const float* protos, const float* p1, const float* p2,
                            const float* soft, float* out, int b, int k, int c, int h, int w, int H,
                            int W, float temp, int views, void* ws, size_t ws_bytes,
                            rgda_stream_t stream);

/* label_refine WITH the superpixel view (label_t_sup given, alignment.py:238-258): label_t_sup (b, H*W) int64 superpixel
 * ids in [0, max_regions).  Per (image, superpixel, class) the maximum of `soft` over the superpixel's pixels (the
 * reference's torch_scatter.scatter(reduce='max')), gathered back per pixel, softmax_T(., temp) over the classes divided by
 * its per-pixel maximum + 1e-7 = sup_weight; the pixels of the superpixel with the LARGEST id of the whole batch are
 * `ignored` and keep the weight of the other views.  views: 3 = mode 'all' (weight * sup_weight), 0 = mode 's' (sup_weight
 * alone; feat / protos / p1 / p2 may be NULL and h, w are not used -- pass 1); 1 / 2 are served too (the reference's modes
 * 'p' / 'l' do not look at label_t_sup).  Workspace: rgda_label_refine's, then the maxima table, then two int32 at
 * rgda_label_refine_sup_flag_offset: [0] the largest id met, [1] non-zero when an id lay outside [0, max_regions) (such
 * pixels are treated as `ignored`; the reference would fault in scatter).  The per-class maxima of `out` are left at
 * rgda_label_refine_classmax_offset as by rgda_label_refine. */
size_t rgda_label_refine_sup_workspace(int b, int c, int h, int w, int max_regions);
size_t rgda_label_refine_sup_flag_offset(int b, int c, int h, int w, int max_regions);
int rgda_label_refine_sup(const float* feat, const float* protos, const float* p1, const float* p2,
                          const float* soft, const int64_t* label_t_sup, float* out, int b, int k, int c,
                          int h, int w, int H, int W, float temp, int views, int max_regions, void* ws,
                          size_t ws_bytes, rgda_stream_t stream);

/* Aligner.update_prototype(feat, label)  regda/gast/alignment.py:86-90,300-327,456-481.
 * feat NCHW f32 (b,k,h,w); label (b,H,W) int64 with H = 16h, W = 16w;
 * protos (c,k) f32 updated in place; label_ds (b,h*w) int64 out.
 * ws: f32 sums[c][k], f32 cnt[c]. */
size_t rgda_proto_update_workspace(int c, int k);
int rgda_proto_update(const float* feat, const int64_t* label, float* protos, int64_t* label_ds,
                      int b, int k, int c, int h, int w, int scale, int ignore_label,
                      float min_ratio, float decay, void* ws, size_t ws_bytes,
                      rgda_stream_t stream);
/* The two halves of rgda_proto_update, for data-parallel ranks (ABI 9).  rgda_proto_stats leaves the
 * sufficient statistics of _compute_local_prototypes (alignment.py:300-327) in `stats` (rgda_proto_update_workspace
 * bytes): f32 sums[c][k] = sum of feat over the pixels whose downscaled label is class c, f32 cnt[c] = their number,
 * then a flag word.  Both add over batches: the ranks all-reduce (sum) the first c * k + c floats and each calls
 * rgda_proto_apply -- local = sums / (cnt + 1e-7), the old prototype where cnt < 1 (:318-321), EMA (:435-438) -- which
 * gives the prototypes of the concatenated global batch on every rank.  rgda_proto_update == stats + apply. */
int rgda_proto_stats(const float* feat, const int64_t* label, int64_t* label_ds, int b, int k, int c,
                     int h, int w, int scale, int ignore_label, float min_ratio, void* stats,
                     size_t stats_bytes, rgda_stream_t stream);
int rgda_proto_apply(float* protos, const void* stats, int c, int k, float decay, rgda_stream_t stream);

/* loss_calc([p1,p2], label, CrossEntropy, multi=True) forward + d(loss)/d(logits)
 *   regda/utils/tools.py:240-254; regda/gast/balance.py:88-101.
 * p1,p2: NCHW f32 (b,c,h,w) logits; label (b,H,W) int64; class_weight: NULL or
 * f32[2][c] per-head per-class weights (ClassBalance, balance.py:27-43).
 * loss: f32[1] (mean over ALL pixels, mean over heads).  g1,g2: NULL or
 * NCHW f32 (b,c,h,w) gradients of `loss` w.r.t. p1,p2. */
size_t rgda_upsample_ce_workspace(int b, int c, int h, int w, int H, int W);
int rgda_upsample_ce(const float* p1, const float* p2, const int64_t* label,
                     const float* class_weight, float* loss, float* g1, float* g2, int b, int c,
                     int h, int w, int H, int W, int ignore_label, void* ws, size_t ws_bytes,
                     rgda_stream_t stream);

/* Deeplabv2 eval-branch output  regda/models/Encoder.py:152-155:
 * (softmax(up(x1)) + softmax(up(x2))) / 2, up = bilinear align_corners=True. */
int rgda_teacher_probs(const float* p1, const float* p2, float* probs, int b, int c, int h, int w,
                       int H, int W, rgda_stream_t stream);

/* ClassBalance._local_freq counts (balance.py:45-53): cnt[c] int32 += #pixels per class. */
int rgda_class_count(const int64_t* label, int32_t* cnt, int64_t n, int c, rgda_stream_t stream);

/* ------------------------------------------------------------- conv stack  */

/* Implicit-GEMM convolution on PxC bf16 activations, bf16 MFMA, fp32 accumulate.
 * Replaces nn.Conv2d forward (cuDNN) for every conv of regda/_resnets.py:72-112,
 * regda/models/Encoder.py:8-65, and -- with mode=1 and transposed weights -- the
 * data-gradient of the same convs.
 *   x   : [N*H*W][ldx] bf16, Cin channels used
 *   wgt : [Cout][kh*kw][Cin] bf16  (mode 1: [Cin_of_fwd][kh*kw][Cout_of_fwd])
 *   y   : [N*Ho*Wo][ldy] bf16
 *   res : NULL or [N*Ho*Wo][ldres] bf16 added to the result before the store
 *   res_relu_mask : NULL or uint8 [N*Ho*Wo][Cout/8] sign bits (rgda_bn_train_apply): res is added only where its
 *          bit is set -- the gradient of a residual connection gated by the ReLU it passed through, so that gated
 *          copy never has to be written to memory
 *   stats: NULL or rgda_stat_t[RGDA_STAT_REPLICAS][2][Cout] (FRAC_FWD); per-channel sum and sum of squares of the
 *          (bf16-rounded) outputs are accumulated order-independently (BatchNorm batch stats)
 *   mode 0: y[n,ho,wo] = sum x[n, ho*stride-pad+kh*dil, wo*stride-pad+kw*dil] * w
 *   mode 1: y[n,ho,wo] = sum x[n, (ho+pad-kh*dil)/stride, (wo+pad-kw*dil)/stride] * w
 *           (terms with a non-integer or out-of-range source are zero)
 * Requires Cin % 32 == 0, Cout % 8 == 0, ld* % 8 == 0. */
int rgda_conv2d(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res,
                int ldres, const uint8_t* res_relu_mask, rgda_stat_t* stats, int stat_groups, int N, int H, int W,
                int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int mode,
                rgda_stream_t stream);

/* Several INDEPENDENT rgda_conv2d calls as (at most) one launch per eight: a descriptor holds exactly rgda_conv2d's
 * arguments.  Same results as calling rgda_conv2d on each in turn (no output of one may be an input of another).  Problems
 * served by the small-tile kernel (conv_igemm_kernel<128, 64, 3, 2, 2, true, false>: what the tiny maps of the PPM branches
 * get, regda/models/Encoder.py:30-51 -- four scales x two statistics groups, 4 - 36 workgroups each) share launches; any
 * other is launched on its own.  Every descriptor is validated before anything is launched.
 * rgda_conv2d_grouped_launches: the number of kernel launches the list makes (negative: the status it would fail with). */
typedef struct rgda_conv2d_desc {
    const void* x; const void* wgt; void* y; const void* res; const uint8_t* res_relu_mask; rgda_stat_t* stats;
    int ldx, ldy, ldres, stat_groups;
    int N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, mode;
} rgda_conv2d_desc;
int rgda_conv2d_grouped(const rgda_conv2d_desc* descs, int n, rgda_stream_t stream);
int rgda_conv2d_grouped_launches(const rgda_conv2d_desc* descs, int n);

/* Forward conv with an INFERENCE-mode BatchNorm (+ residual + ReLU) folded into the epilogue -- the EMA teacher's
 * conv + BN + ReLU units (regda/models/Encoder.py:152-155 runs the model in eval()):
 *   y = act((conv(x) - running_mean) / sqrt(running_var + eps) * gamma + beta + res). */
int rgda_conv2d_bneval(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                       const float* running_mean, const float* running_var, const float* gamma,
                       const float* beta, float eps, int relu, int N, int H, int W, int Cin, int Ho, int Wo,
                       int Cout, int kh, int kw, int stride, int pad, int dil, rgda_stream_t stream);

/* rgda_conv2d (normally the data-gradient, mode 1) with the BatchNorm-backward REDUCTION of the layer that consumes
 * its output fused into the epilogue: with g = the stored result (after the residual add),
 *   g' = g * [bn_y > 0 if relu] * nscale[n][c],  xhat = (bn_x - mean) * invstd  (mean/invstd from bn_mi[group]),
 * ([bn_y > 0] is read from bn_relu_mask instead when that is given, see rgda_bn_train_apply)
 * sums[group][REPLICAS][2][Cout] += (sum g', sum g' * xhat) -- exactly what rgda_bn_bwd_reduce would compute
 * from the stored tensor, without re-reading it.
 * relu == 2 (ABI 6): the consumer's activation was never written (it ran on ITS consumer's operand path,
 * rgda_conv2d_bnin): [bn_y > 0] is recomputed from bn_x as [fma(bn_x, scale, shift) > 0] with the forward's own
 * (scale, shift) = (gamma * invstd, fma(-mean, scale, beta)); bn_gamma / bn_beta f32 [Cout] are needed only then. */
int rgda_conv2d_bnbwd(const void* x, int ldx, const void* wgt, void* y, int ldy, const void* res, int ldres,
                      const uint8_t* res_relu_mask, rgda_stat_t* sums, int groups, const void* bn_y, int bn_ldy, const uint8_t* bn_relu_mask,
                      const void* bn_x, int bn_ldx,
                      const float* bn_mi, const float* bn_nscale, int rows_per_image, int relu,
                      const float* bn_gamma, const float* bn_beta, int N, int H,
                      int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                      int mode, rgda_stream_t stream);

/* BatchNorm (+ ReLU) on the CONSUMER's operand path (ABI 6).  The reference's bottleneck runs conv -> bn -> relu -> conv
 * (regda/_resnets.py:92-112); as separate passes the normalised activation is written and read back once per unit.
 * Here the producing convolution leaves its RAW output and its statistic accumulators, and the consuming convolution
 * applies  a = relu(fma(x, scale, shift)),  scale = gamma * invstd, shift = fma(-mean, scale, beta)  to the operand tile
 * on its way to the matrix pipe (mean / invstd from `stats` exactly as rgda_bn_train_apply derives them: biased batch
 * variance, eps inside the square root).  The activation is never written; padding positions contribute zeros as in the
 * reference (the padding of conv2 pads the ACTIVATION).  One workgroup of the launch also writes `mi` (mean, invstd per
 * group, for the backward pass) and updates running_mean / running_var (momentum, unbiased variance) group after group
 * and num_batches_tracked += groups -- nn.BatchNorm2d's train-mode side effects (regda/resnet.py:170-181 keeps every
 * BatchNorm trainable; tools/train_ssl_reg.py:210-212 runs source then target).
 * The struct is HOST memory holding DEVICE pointers; it is consumed before the call returns. */
typedef struct rgda_bn_operand {
    const rgda_stat_t* stats;       /* [groups][REPLICAS][2][C] (FRAC_FWD) of the producing convolution, complete */
    const float* gamma;             /* f32 [C] */
    const float* beta;              /* f32 [C] */
    float* mi;                      /* NULL or out f32 [groups][2][C]: mean, invstd */
    float* running_mean;            /* NULL or f32 [C], updated in place (both or neither) */
    float* running_var;
    int64_t* num_batches_tracked;   /* NULL or += groups */
    float eps, momentum;
    int groups;                     /* equal blocks of whole images, normalised independently (= stat_groups of the call) */
    int relu;
} rgda_bn_operand;
/* rgda_conv2d (mode 0) whose operand x [N*H*W][ldx] is the RAW output of the producing convolution and bn_in describes
 * the BatchNorm (+ ReLU) between them.  Served: Cin <= 512 and the geometries rgda_conv2d_bnin_supported() reports
 * (1x1 and 3x3 convolutions with >= 512 tiles of 128 x 128, and the 3x3 / dilation 1 convolutions on 32-wide maps);
 * anything else returns RGDA_ERR_UNSUPPORTED and the caller materialises the activation with rgda_bn_train_apply.
 * rgda_conv2d_bnin_supported: 0 = not served, 1 = served, 2 = served AND measured faster than the apply pass it
 * replaces (the model defers a unit only then). */
int rgda_conv2d_bnin_supported(int64_t M, int Cout, int Cin, int kh, int kw, int stride, int pad, int dil, int H, int W,
                               int Ho, int Wo, int groups);
int rgda_conv2d_bnin(const rgda_bn_operand* bn_in, const void* x, int ldx, const void* wgt, void* y, int ldy,
                     const void* res, int ldres, rgda_stat_t* stats, int stat_groups, int N, int H, int W, int Cin,
                     int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, rgda_stream_t stream);

/* The kernel instantiation that serves a convolution call, as rocprofv3 names it ("conv_igemm_kernel<128, 128, 2, 2, 4,
 * false, false>", "conv3x3_halo_kernel<1, 4, true>", ...), decided by the library's own dispatch (nothing is launched):
 * variant 0 = rgda_conv2d (has_stats / stat_groups as in the call), 1 = rgda_conv2d_bneval, 2 = rgda_conv2d_bnbwd,
 * 3 = rgda_conv2d_bnin; rgda_conv2d_wgrad_kernel: the instantiation a layer of rgda_conv2d_wgrad_grouped maps to
 * (layers with the same name share launches).  NULL where the call would be refused.  bench.py labels its per-launch
 * HIP-event timings with these, so that they can be matched against rocprof kernel statistics. */
const char* rgda_conv2d_kernel(int variant, int N, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                               int stride, int pad, int dil, int mode, int has_stats, int stat_groups);

/* Which conv_igemm_kernel<BC, BP, STAGES, ...> instantiation rgda_conv2d picks for a problem: returns
 * BC | BP << 10 | STAGES << 20 (STAGES 82 / 83 = 8-wave workgroups with a 2 / 3 stage ring), or a negative
 * status.  rows_per_group = rows of one BatchNorm group when fused statistics with groups > 1 are requested, else 0.
 * (bench.py labels its per-launch timings with it so they can be matched against rocprof kernel names.) */
int rgda_conv2d_tile(int64_t M, int Cout, int kh, int kw, int Cin, int rows_per_group);

/* Weight gradient: dw[co][tap][ci] (f32, row stride taps*Cin) +=
 *   sum_p dy[p][co] * x[src(p,tap)][ci]   (same geometry as mode 0 above).
 * Reproducible: the pixel (K) dimension of a layer is split over several workgroups only through the workspace --
 * every split leaves its partial tile there, the LAST workgroup of a tile to arrive (a counter per tile) adds the
 * partials in split order 0, 1, ... and is the only one that adds to dw (ABI 4; every split added to dw with fp32
 * atomics before: the arrival order moved the last bits from run to run).  ws: rgda_conv2d_wgrad_workspace bytes, or NULL = never
 * split (slow for layers with few tiles).  The first 64 KiB of ws are tile counters: ZERO them once after
 * allocation; every call leaves them zero.  Calls that share a ws must be ordered on one stream. */
int rgda_conv2d_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dw, int N, int H,
                      int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride,
                      int pad, int dil, void* ws, size_t ws_bytes, rgda_stream_t stream);

/* The weight gradients of several layers in as few launches as possible (same arithmetic as n calls of
 * rgda_conv2d_wgrad, accumulated into each dw).  Layers that map to the same kernel instantiation share a
 * launch (up to 16 per launch): autograd hands the reference one conv-backward at a time
 * (tools/train_ssl_reg.py:236 loss.backward()), but the gradients are only needed by clip + SGD
 * (tools/train_ssl_reg.py:237-238), so the host may collect them while the data-gradient chain runs on.
 * `descs` is a host array; it is consumed before the call returns. */
typedef struct rgda_wgrad_desc {
    const void* x;      /* bf16 [N*H*W][ldx] */
    const void* dy;     /* bf16 [N*Ho*Wo][lddy] */
    float* dw;          /* f32 [Cout][kh*kw][Cin] */
    int ldx, lddy;
    int N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil;
    /* ABI 5: where dw's rows live, so that a gradient can be written straight into a slice of a wider tensor.
     * lddw: elements between consecutive (co, tap) rows of dw (0 = Cin: dense).  co_split (0 = off; a power of two
     * dividing Cout; 1x1 layers only): output row r of the layer is dw row (r % co_split) * (Cout / co_split) +
     * r / co_split -- a layer whose Cout stacks T filters of co_split channels, [t][co], lands in a [co][t][Cin]
     * tensor (the PPM branches of the heads' 3x3 convolution, regda/models/Encoder.py:29-37,54-60). */
    int lddw, co_split;
} rgda_wgrad_desc;
size_t rgda_conv2d_wgrad_workspace(const rgda_wgrad_desc* descs, int n);
const char* rgda_conv2d_wgrad_kernel(const rgda_wgrad_desc* desc);
int rgda_conv2d_wgrad_grouped(const rgda_wgrad_desc* descs, int n, void* ws, size_t ws_bytes, rgda_stream_t stream);

/* Stem im2col: NCHW f32 image (N,3,H,W) -> [N*Ho*Wo][Kp] bf16 patches of the
 * 7x7/2 pad-3 conv (regda/_resnets.py:150-151), k index = (kh*7+kw)*3+c, zero padded to Kp. */
int rgda_stem_im2col(const float* img, void* col, int N, int H, int W, int Ho, int Wo, int Kp,
                     rgda_stream_t stream);

/* The same convolution in ONE kernel straight from the NCHW f32 image (no patch matrix): y PxC bf16 [N*Ho*Wo][ldy],
 * wgt bf16 [64][192] (k = (kh*7+kw)*3+c, zero padded), epilogue as rgda_conv2d (per-group BatchNorm statistics) or
 * rgda_conv2d_bneval (inference BatchNorm + ReLU).  Wo % 64 must be 0 (RGDA_ERR_UNSUPPORTED otherwise: use
 * rgda_stem_im2col + rgda_conv2d).  The image batch is ONE tensor: the row groups are consecutive blocks of images. */
int rgda_stem_conv(const float* img, const void* wgt, void* y, int ldy, rgda_stat_t* stats, int stat_groups, int N,
                   int H, int W, int Ho, int Wo, rgda_stream_t stream);
int rgda_stem_conv_bneval(const float* img, const void* wgt, void* y, int ldy, const float* running_mean,
                          const float* running_var, const float* gamma, const float* beta, float eps, int relu,
                          int N, int H, int W, int Ho, int Wo, rgda_stream_t stream);

/* BatchNorm2d (train) on PxC bf16, nn.BatchNorm2d defaults (eps 1e-5, momentum .1):
 *  finalize: stats rgda_stat_t[REPLICAS][2][C] (sum,sumsq over M rows, FRAC_FWD) -> mean/invstd f32[2][C] in `mi`,
 *            running_mean/var/num_batches_tracked update.  If stats==NULL, eval mode:
 *            mi is filled from the running statistics.
 *  apply   : y = act( (x-mean)*invstd*gamma+beta [+ res] ) [* nscale[n][c]]
 *  bwd     : two passes (reduce, apply) -- see kernels.
 * `groups` (>= 1): the M rows are `groups` equal blocks of rows (the source and the target batch of one
 * SSL step run through the network together) that are normalised INDEPENDENTLY, like the reference's two
 * separate forward calls; stats / sums are laid out [groups][REPLICAS][2][C], mi [groups][2][C]; running
 * statistics are updated group after group. */
int rgda_bn_stats(const void* x, int ldx, rgda_stat_t* stats, int64_t M, int C, rgda_stream_t stream);
int rgda_bn_finalize(const rgda_stat_t* stats, float* mi, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, int64_t M, int C, int groups, float eps,
                     float momentum, rgda_stream_t stream);
int rgda_bn_apply(const void* x, int ldx, const float* mi, const float* gamma, const float* beta,
                  const void* res, int ldres, const float* nscale, int rows_per_image, void* y,
                  int ldy, int64_t M, int C, int relu, int groups, rgda_stream_t stream);
/* bn_finalize + bn_apply in one launch (the training forward): statistics -> (mean, invstd) -> y, `mi` and the
 * running statistics are written by one designated workgroup per channel block.
 * relu_mask (optional, needs relu): uint8 [M][C/8], bit e of byte k = [y[row][8k+e] > 0] -- the only thing the
 * backward pass needs from y; reading it instead of y saves 15/16 of that operand's HBM traffic there. */
int rgda_bn_train_apply(const void* x, int ldx, const rgda_stat_t* stats, float* mi, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, const float* gamma,
                        const float* beta, const void* res, int ldres, const float* nscale,
                        int rows_per_image, void* y, int ldy, uint8_t* relu_mask, int64_t M, int C, int relu,
                        int groups, float eps, float momentum, rgda_stream_t stream);
/* sums rgda_stat_t[REPLICAS][2][C] (FRAC_BWD) must be ZERO on entry (the caller clears one arena per backward pass):
 * sums[0] += sum(g'), sums[1] += sum(g' * xhat), g' = g*[y>0]*nscale.  With relu, [y>0] comes from relu_mask
 * when given, else from y.
 * relu == 2 (ABI 6): [y>0] is recomputed from x, [fma(x, gamma * invstd, fma(-mean, gamma * invstd, beta)) > 0] -- the
 * unit's activation was never written (rgda_conv2d_bnin); gamma / beta are needed only then. */
int rgda_bn_bwd_reduce(const void* g, int ldg, const void* y, int ldy, const uint8_t* relu_mask, const void* x,
                       int ldx, const float* mi, const float* nscale, int rows_per_image, rgda_stat_t* sums,
                       int64_t M, int C, int relu, const float* gamma, const float* beta, int groups,
                       rgda_stream_t stream);
/* dx = gamma*invstd*(g' - sum(g')/M - xhat*sum(g' xhat)/M); gmask (optional) = g';
 * dgamma += sum(g' xhat), dbeta += sum(g') (f32, accumulated)
 * relu == 2 (ABI 6, needs beta): the ReLU sign from x as in rgda_bn_bwd_reduce; act_out (optional, bf16 [M][ldact]) then
 * receives the activation relu(fma(x, scale, shift)) itself -- bit for bit what the consuming convolution's operand
 * path fed the matrix pipe -- for that convolution's weight gradient (rgda_conv2d_wgrad), written here beside the
 * read of x this pass does anyway instead of in the forward pass. */
int rgda_bn_bwd_apply(const void* g, int ldg, const void* y, int ldy, const uint8_t* relu_mask, const void* x,
                      int ldx, const float* mi, const float* gamma, const float* nscale,
                      int rows_per_image, const rgda_stat_t* sums, void* dx, int lddx, void* gmask,
                      int ldgm, float* dgamma, float* dbeta, int64_t M, int C, int relu, const float* beta,
                      void* act_out, int ldact, int groups, rgda_stream_t stream);

/* BatchNorm of SMALL maps, several layers per launch (the PPM branches' conv -> BatchNorm -> ReLU on s x s maps,
 * regda/models/Encoder.py:30-51; nn.BatchNorm2d in train mode per statistics group, as rgda_bn_train_apply /
 * rgda_bn_bwd_reduce + rgda_bn_bwd_apply).  One workgroup owns 128 channels of one layer and all of its rows: it forms the
 * statistics itself -- sums of the stored bf16 values in a fixed order -- and applies them; up to eight descriptors share a
 * launch.  M = all rows (groups * rows per group), rows per group in [2, 320] forward, <= 320 backward (a workgroup keeps
 * its rows in registers); larger maps take the general entry points.
 * Forward: y = [relu]((x - mean) * invstd * gamma + beta), mi[groups][2][C] = (mean, invstd), running statistics updated
 * group after group, num_batches_tracked += groups, relu_mask (optional) = sign bits of y.
 * Backward: dx, dgamma += , dbeta += (atomic adds, as rgda_bn_bwd_apply); the ReLU gate from relu_mask or y. */
typedef struct rgda_bn_small_fwd_desc {
    const void* x; void* y; uint8_t* relu_mask; const float* gamma; const float* beta; float* mi;
    float* running_mean; float* running_var; int64_t* num_batches_tracked;
    int64_t M; int ldx, ldy, C, groups, relu; float eps, momentum;
} rgda_bn_small_fwd_desc;
typedef struct rgda_bn_small_bwd_desc {
    const void* g; const void* y; const uint8_t* relu_mask; const void* x; void* dx; const float* mi; const float* gamma;
    float* dgamma; float* dbeta;
    int64_t M; int ldg, ldy, ldx, lddx, C, groups, relu;
} rgda_bn_small_bwd_desc;
int rgda_bn_train_small(const rgda_bn_small_fwd_desc* descs, int n, rgda_stream_t stream);
int rgda_bn_bwd_small(const rgda_bn_small_bwd_desc* descs, int n, rgda_stream_t stream);

/* MaxPool2d(3,2,1) on PxC bf16 (regda/_resnets.py:153); idx = argmax tap (uint8). */
int rgda_maxpool_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int Ho,
                     int Wo, rgda_stream_t stream);
int rgda_maxpool_bwd(const void* gy, const uint8_t* idx, void* gx, int N, int H, int W, int C,
                     int Ho, int Wo, rgda_stream_t stream);
/* The same pooling over relu(BatchNorm(x)) of the RAW stem convolution output x (rgda_bn_operand above; conv1 -> bn1 ->
 * relu -> maxpool, regda/_resnets.py:150-153): the stem's activation is never written. */
int rgda_maxpool_fwd_bnin(const rgda_bn_operand* bn_in, const void* x, void* y, uint8_t* idx, int N, int H, int W,
                          int C, int Ho, int Wo, rgda_stream_t stream);

/* InstanceNorm2d(C, affine=False, eps) (regda/models/Encoder.py:123,146-147).
 * x [N*HW][ldx] bf16 -> y0,y1 (optional, bf16 PxC, ld ldy) and feat NCHW f32 (optional);
 * mi f32[N][2][C] mean/invstd saved for backward. */
int rgda_instnorm_fwd(const void* x, int ldx, void* y0, void* y1, int ldy, float* feat_nchw,
                      float* mi, int N, int HW, int C, float eps, rgda_stream_t stream);
/* g = ga + gb + gc (bf16 PxC, any may be NULL; ga/gb share ldg); dx bf16 */
int rgda_instnorm_bwd(const void* ga, const void* gb, int ldg, const void* gc, int ldgc,
                      const void* x, int ldx, const float* mi, void* dx, int lddx, int N, int HW,
                      int C, rgda_stream_t stream);

/* Spatial linear map shared by AdaptiveAvgPool2d / bilinear(align_corners=False)
 * and their transposes (regda/models/Encoder.py:16-18,48-51):
 *   out[n][i][c] (+)= sum_j Mx[i][j] * in[n][j][c],  Mx f32 [I][J] row-major.
 * in: bf16 [N*J][ldin]; out: bf16 [N*I][ldout] (out_f32 != 0: f32). */
int rgda_spatial_mix(const void* in, int ldin, const float* Mx, void* out, int ldout, int N, int I,
                     int J, int C, int accumulate, int out_f32, rgda_stream_t stream);

/* out[n][i][c] = sum over nsrc <= 4 sources of sum_j mats[q][i][j] * ins[q][n][j][c]  (bf16 out, no accumulate):
 * the summed backward of the four adaptive-average-pool branches of one PPM head pair in a single pass.
 * ins/ldins/mats/Js are HOST arrays of length nsrc (device pointers inside). */
int rgda_spatial_mix_multi(int nsrc, const void* const* ins, const int* ldins, const float* const* mats,
                           const int* Js, void* out, int ldout, int N, int I, int C, rgda_stream_t stream);

/* 1x1 classifier with bias (regda/models/Encoder.py:40): hidden [M][ldh] bf16 ->
 * logits NCHW f32 (N,ncls,HW); backward gives dhidden (bf16), dW f32[ncls][C] +=, db +=.
 * ws (optional, rgda_classifier_bwd_workspace bytes): per-workgroup partial sums + a deterministic reduction instead of
 * atomics (NULL: atomics). */
int rgda_classifier_fwd(const void* hidden, int ldh, const float* w, const float* bias,
                        float* logits, int N, int HW, int C, int ncls, rgda_stream_t stream);
size_t rgda_classifier_bwd_workspace(int64_t M, int C, int ncls);
int rgda_classifier_bwd(const void* hidden, int ldh, const float* w, const float* glogits,
                        void* dhidden, int lddh, float* dw, float* db, int N, int HW, int C,
                        int ncls, void* ws, size_t ws_bytes, rgda_stream_t stream);

/* ------------------------------------------------------------- optimizer   */

/* sum of squares of a flat f32 buffer -> out[0] (f32, overwritten).  ws: f32[1024]. */
int rgda_sumsq(const float* g, int64_t n, float* out, float* ws, rgda_stream_t stream);

/* clip_grad_norm_(max_norm, 2) + SGD(momentum, weight_decay) + EMA shadow + bf16 mirror
 *   tools/train_ssl_reg.py:174-175,239-241; regda/utils/ema.py:46-51.
 * coef = min(1, max_norm / (sqrt(gnorm_sq[0]) * gscale + 1e-6)); g = g*gscale*coef (gscale = 1/world)
 * v = momentum*v + (g + wd*p); p -= lr*v; shadow = (1-d)*p + d*shadow (if shadow);
 * p_bf16 = bf16(p) (if given); shadow_bf16 = bf16(shadow) (if given: the EMA teacher's mirror).
 * lr is read from device memory (lr_dev[0]). */
int rgda_sgd_step(float* p, const float* g, float* v, float* shadow, void* p_bf16, void* shadow_bf16,
                  const float* gnorm_sq, const float* lr_dev, int64_t n, float momentum,
                  float weight_decay, float max_norm, float gscale, float ema_decay,
                  int first_step, rgda_stream_t stream);

/* The stem's weight gradient straight from the NCHW f32 image (no patch matrix): dw f32 [64][147], k = (kh*7+kw)*3+c,
 * += sum_p dy[p][co] * bf16(patch(p)[k]) (the products rgda_stem_im2col + rgda_conv2d_wgrad form, summed in a fixed order:
 * reproducible).  dy bf16 [N*Ho*Wo][lddy].  ws: rgda_stem_wgrad_workspace(N, H, W) bytes (0 = this geometry is not
 * served), 16-byte aligned; calls that share it must be ordered on one stream.  Wo % 64 must be 0
 * (RGDA_ERR_UNSUPPORTED otherwise).  Replaces the tail of regda/_resnets.py:150-151's backward. */
size_t rgda_stem_wgrad_workspace(int N, int H, int W);
int rgda_stem_wgrad(const float* img, const void* dy, int lddy, float* dw, void* ws, size_t ws_bytes, int N, int H,
                    int W, int Ho, int Wo, rgda_stream_t stream);

/* w [Co][T][Ci] f32 -> wt [Ci][T][Co] bf16 (weights for the data-gradient pass). */
int rgda_weight_transpose_bf16(const float* w, void* wt, int Co, int T, int Ci, rgda_stream_t stream);
/* every derived bf16 weight layout of a model in one launch: device table[n][8] int64 = {src f32*, dst bf16*,
 * Co, T, Ci, first_block, src_ld, mode}.  src element (co,tap,ci) = src[(co*T+tap)*src_ld + ci] (src_ld > Ci: a
 * channel slice of a wider tensor); mode 0: dst[ci][tap][co] (the data-gradient operand), 1: dst[tap][co][ci],
 * 2: dst[co][tap][ci].  A row owns ceil(Ci/RGDA_LAYOUT_TILE)*ceil(Co/RGDA_LAYOUT_TILE)*T consecutive blocks starting
 * at first_block. */
int rgda_weight_transpose_batched(const int64_t* table, int n, int64_t total_blocks, rgda_stream_t stream);
int rgda_cast_bf16(const float* src, void* dst, int64_t n, rgda_stream_t stream);
int rgda_cast_f32(const void* src_bf16, float* dst, int64_t n, rgda_stream_t stream);
/* Gradient exchange with bf16 payloads and fp32 accumulation (regda_amd/ddp.py, payload 'bf16'; the reference has no
 * data-parallel path): recv bf16 [world][shard_elems] = this rank's shard of a bucket as every rank sent
 * it (all-to-all); out bf16 [shard_elems] = bf16(sum over ranks IN RANK ORDER of float(recv[r])) -- the same bits on
 * every rank.  shard_elems % 8 == 0. */
int rgda_ddp_accumulate_bf16(const void* recv, int world, void* out, int64_t shard_elems, rgda_stream_t stream);

/* ---- the collectives themselves, through RCCL (xGMI), for a host without torch.distributed ("RCCL wrappers
 * for (e)"; the reference is single-GPU, tools/train_ssl_reg.py has no counterpart).  One communicator per process and
 * GPU: rank 0 draws the 128-byte id (rgda_comm_unique_id) and the HOST ships it to the other ranks by whatever channel it
 * has (a file, a socket, MPI, a torch.distributed store); every rank then calls rgda_comm_init on the device it has made
 * current.  The calls enqueue on the caller's stream like every other entry point; buffers are caller-owned device
 * memory.  dtype: RGDA_COMM_*.  librccl.so is resolved at the first call (a copy the process already holds is
 * preferred); without it these return RGDA_ERR_UNSUPPORTED and nothing else of the library is affected.
 *   all_reduce : buf[n] <- sum over ranks, in place (the flat fp32 gradient's buckets, the prototype statistics
 *                sums[c][k] + cnt[c] of rgda_proto_stats, ClassBalance's int64 pixel counts)
 *   all_to_all : rank r's send[j * n_per_rank ...] -> rank j's recv[r * n_per_rank ...]   (bf16 payload, phase 1)
 *   all_gather : recv[r * n_per_rank ...] <- rank r's send[n_per_rank]                    (bf16 payload, phase 2) */
typedef void* rgda_comm_t;
#define RGDA_COMM_ID_BYTES 128
#define RGDA_COMM_F32 0
#define RGDA_COMM_BF16 1
#define RGDA_COMM_I64 2
#define RGDA_COMM_F64 3
int rgda_comm_unique_id(void* id_128_bytes);
int rgda_comm_init(const void* id_128_bytes, int rank, int world, rgda_comm_t* comm);
int rgda_comm_destroy(rgda_comm_t comm);
int rgda_comm_all_reduce(rgda_comm_t comm, void* buf, int64_t n, int dtype, rgda_stream_t stream);
int rgda_comm_all_gather(rgda_comm_t comm, const void* send, void* recv, int64_t n_per_rank, int dtype,
                         rgda_stream_t stream);
int rgda_comm_all_to_all(rgda_comm_t comm, const void* send, void* recv, int64_t n_per_rank, int dtype,
                         rgda_stream_t stream);
/* rows of K f32 -> rows of Kp bf16, zero padded (stem weights [64][147] -> [64][192]); and the
 * reverse accumulation dst[R][K] f32 += src[R][Kp] f32 for the stem weight gradient. */
int rgda_pad_cast_bf16(const float* src, void* dst, int R, int K, int Kp, rgda_stream_t stream);
int rgda_unpad_acc_f32(const float* src, float* dst, int R, int K, int Kp, rgda_stream_t stream);
/* The small chores of a step as kernels of this library (so that a step launches nothing from torch or the runtime's
 * blit kernels): clear a buffer (16-byte aligned); up to four device -> device copies in one launch (16-byte aligned,
 * sizes multiples of 16; dsts / srcs / bytes are HOST arrays); one f32 word (the learning rate the optimizer reads from
 * device memory); Dropout2d(p) keep masks, scaled by 1 / (1 - p) (regda/models/Encoder.py:39), from a counter-based
 * generator: element i depends on (seed, i) only. */
int rgda_fill_zero(void* p, size_t bytes, rgda_stream_t stream);
int rgda_copy_multi(int n, void* const* dsts, const void* const* srcs, const size_t* bytes, rgda_stream_t stream);
int rgda_set_f32(float* p, float value, rgda_stream_t stream);
int rgda_dropout_mask(float* out, int64_t n, float p, uint64_t seed, rgda_stream_t stream);
/* out = a + b (bf16, PxC) */
int rgda_add_bf16(const void* a, int lda, const void* b, int ldb, void* out, int ldo, int64_t M,
                  int C, rgda_stream_t stream);

/* ------------------------------------------------------------- teacher / pseudo-label harness */

/* One view of the 8-view test-time augmentation of regda/utils/tools.py:132-152 (ttach 0.0.3:
 * Compose([HorizontalFlip(), Rotate90([0,90,180,270])]), un-vendored, restated).  With R = torch.rot90(., 1, (2,3))
 * and F = flip(3): flip_first=1 -> dst = R^k(F^f(src)) (augment_image), flip_first=0 -> dst = F^f(R^k(src))
 * (deaugment_mask with k := (4-k)%4).  dst (+)= scale * view; fp32 NCHW; output is Ws x Hs for odd k. */
int rgda_dihedral_nchw(const float* src, float* dst, int N, int C, int Hs, int Ws, int hflip, int rot_k,
                       int flip_first, float scale, int accumulate, rgda_stream_t stream);

/* Sliding-window inference of regda/utils/tools.py:61-97 (pre_slide): crop + zero-pad one window to the tile size
 * (tools.py:79-80), add a tile of predictions into the full map and bump the visit count (tools.py:91-93),
 * divide by the count (tools.py:95). */
/* pad_image exactly as written (tools.py:51-58): tnf.pad(img, (0, 0, top, bottom)) pads the ROW dimension (top by
 * rows_missing, bottom by cols_missing; negative = crop) and leaves W alone; dst has h + top + bottom rows. */
int rgda_pad_rows_nchw(const float* src, float* dst, int N, int C, int h, int w, int top, int bottom,
                       rgda_stream_t stream);
int rgda_window_crop(const float* full, float* tile, int N, int C, int Hf, int Wf, int y1, int x1, int h, int w,
                     int Th, int Tw, rgda_stream_t stream);
int rgda_window_accumulate(const float* tile, float* full, float* count, int N, int C, int Hf, int Wf, int y1,
                           int x1, int h, int w, int Th, int Tw, rgda_stream_t stream);
int rgda_window_normalise(float* full, const float* count, int N, int C, int Hf, int Wf, rgda_stream_t stream);

/* tnf.interpolate(mode='bilinear', align_corners=True) of the soft labels to the dataset size
 * (regda/gast/pseudo_generation.py:135). */
int rgda_resize_bilinear_ac(const float* src, float* dst, int N, int C, int h, int w, int H, int W,
                            rgda_stream_t stream);

/* ------------------------------------------------------------- evaluation path */

/* cls.argmax(dim=1) of the (N,C,H,W) probabilities (regda/utils/eval.py:43): int64 (N,H,W), first maximum wins. */
int rgda_argmax_nchw(const float* probs, int64_t* out, int N, int C, int64_t HW, rgda_stream_t stream);
/* Confusion matrix of the pixels with y_true >= 0 (eval.py:45-50; what ever's PixelMetric.forward accumulates):
 * cm int64 [C][C] (row = true class, column = prediction) += counts.  flag |= 1 if a label >= C or a prediction
 * outside [0,C) was seen (those pixels are not counted).  C <= 64. */
int rgda_confusion_accumulate(const int64_t* y_true, const int64_t* y_pred, int64_t* cm, int* flag, int64_t n,
                              int C, rgda_stream_t stream);

/* ------------------------------------------------------------- stage 2, "align"  */

/* PrototypeContrastiveLoss (regda/loss.py:10-47), forward + gradient w.r.t. the features in one pass:
 *   feat f32 NCHW (b,K,h,w); labels int64 (b,h,w), `ignore_label` pixels are removed; protos f32 (C,K), C = 6.
 *   logits = normalize(feat_p) . normalize(protos)^T / temperature   (tnf.normalize: x / max(||x||, 1e-12))
 *   loss[0] += weight * mean over the kept pixels of cross_entropy(logits, label)      (NaN-free only if one is kept)
 *   dfeat (optional) bf16 [b*h*w][lddf], pixel-major -- the layout rgda_instnorm_bwd consumes: (+)= weight * dloss/dfeat
 * ws (rgda_pcl_loss_workspace bytes): normalised prototypes, valid-pixel count, flag (bit 2: label outside [0,C)). */
size_t rgda_pcl_loss_workspace(int C, int K);
int rgda_pcl_loss(const float* feat, const int64_t* labels, const float* protos, float* loss, void* dfeat,
                  int lddf, int accumulate, int b, int K, int C, int h, int w, int ignore_label,
                  float temperature, float weight, void* ws, size_t ws_bytes, rgda_stream_t stream);

/* Factored form of the PPM heads' tap-shifted bilinear maps (regda/models/Encoder.py:30-51: Upsample(bilinear,
 * align_corners=False) of the s x s branches into the 3x3 / pad 1 conv_last): the map V[(y,x)][(jy,jx),(ky,kx)] =
 * Uy[y+ky-1][jy] * Ux[x+kx-1][jx] is separable, so V and V^T are applied as two short maps (csrc/mix_kernels.hip).
 *   rgda_group_mix : out[g][i][c] = sum_j W[i][j] * in[g][j][c] for G groups of J consecutive rows, W f32 [I][J]
 *                    (the x direction: one group per image row);
 *   rgda_sparse_mix: out[n][i][c] = sum_k vals[k] * ins[cols[k] >> 24][n][cols[k] & 0xffffff][c] over CSR row i
 *                    (rowptr int32 [I+1], cols int32, vals f32: DEVICE arrays; ins/ldins/Js: HOST arrays, nsrc <= 4;
 *                    the I = sum(out_rows) rows are split over ndst <= 4 image-major output tensors, out_rows[q]
 *                    rows per image each: outs/ldouts/out_rows HOST arrays)
 *                    (the y direction: the gather from the four branch tensors / the scatter back to them).
 * in/out bf16, or f32 where in_f32 / out_f32 != 0; C % 8 == 0; deterministic. */
int rgda_group_mix(const void* in, int ldin, int in_f32, const float* W, void* out, int ldout, int out_f32,
                   int G, int I, int J, int C, rgda_stream_t stream);
int rgda_sparse_mix(int nsrc, const void* const* ins, const int* ldins, const int* Js, int in_f32,
                    const int* rowptr, const int* cols, const float* vals, int ndst, void* const* outs,
                    const int* ldouts, const int* out_rows, int out_f32, int N, int C, rgda_stream_t stream);

/* ------------------------------------------------------------- ASPP head */

/* Classifier_Module (regda/models/Encoder.py:68-84): out = sum_d Conv2d(K -> C, 3x3, padding = dilation = dil[d],
 * bias)(x), for the model's two heads at once.  The 2*4 convolutions' taps are computed as ONE 1x1 convolution
 * (rgda_conv2d) Z = x @ Wstack^T, Z bf16 [N*h*w][ldz] with column ((head*4 + d)*C + c)*9 + tap -- the stacked filter
 * is the eight [C][3][3][K] weights back to back -- and these two entry points do the rest:
 *   gather : out_head f32 (N,C,h,w) = sum_d bias[head*4+d][c] + sum_{d,tap} Z[(y + dy*dil[d], x + dx*dil[d])][col]
 *   scatter: dZ (bf16 [N*h*w][lddz], zc >= 72*C columns, the pad columns zeroed) from the two logit gradients
 *            g_head f32 (N,C,h,w), and dbias[head*4+d][c] += sum g_head[:,c]   (deterministic).
 * bias / dbias: HOST arrays of eight DEVICE pointers; dil: HOST array of four dilations. */
int rgda_aspp_gather(const void* z, int ldz, const float* const* bias, float* out1, float* out2, int N, int h,
                     int w, int C, const int* dil, rgda_stream_t stream);
int rgda_aspp_scatter(const float* g1, const float* g2, void* dz, int lddz, int zc, float* const* dbias, int N,
                      int h, int w, int C, const int* dil, rgda_stream_t stream);

/* ------------------------------------------------------------------ plan replay
 * The reference drives its step from Python, one operator call at a time (tools/train_ssl_reg.py:198-241); so does
 * regda_amd in eager mode -- ~750 entry-point calls per SSL step.  The step's launch sequence is static (fixed shapes,
 * fixed buffers), so the caller may RECORD it once as a table of rows {entry point, arguments packed into 64-bit
 * slots: integers / pointers / streams by value, float and double by bit pattern} and have it replayed by one call.
 * The table, every buffer it points to (device memory and the small host arrays some entry points take) and the
 * streams are caller-owned; nothing is retained or allocated here.
 *   rgda_plan_fn_id("rgda_conv2d") -> index of that entry point in the dispatch table (-1: not replayable; only
 *     int-returning enqueue entry points are), rgda_plan_fn_count() -> table size.
 *   rgda_plan_run: calls the rows in order; stops at the first row whose entry point returns a negative status and
 *     returns that status (*failed_index = its row, when given).  A row with a bad id / argument count -> RGDA_ERR_ARG. */
#define RGDA_PLAN_MAX_ARGS 36
typedef struct rgda_plan_entry {
    int32_t fn;
    int32_t nargs;
    uint64_t args[RGDA_PLAN_MAX_ARGS];
} rgda_plan_entry;
int rgda_plan_fn_count(void);
int rgda_plan_fn_id(const char* name);
int rgda_plan_run(const rgda_plan_entry* entries, int n, int* failed_index);

#ifdef __cplusplus
}
#endif
#endif /* RGDA_HIP_H */
