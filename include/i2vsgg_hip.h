/*
 * i2vsgg_hip.h -- the drop-in boundary of libi2vsgg_hip.so (MI355X / gfx950).
 *
 * A flat C ABI over raw DEVICE pointers: no torch types, no hidden allocation, no
 * hidden synchronisation, never exit().  Every entry point
 *   - returns int32_t: 0 = ok, <0 = error (I2V_ERR_*; text via i2v_last_error()),
 *   - takes the HIP stream to launch on as an opaque `void*` (hipStream_t),
 *   - writes only into caller-owned outputs / caller-provided workspace
 *     (size from the matching *_workspace_bytes() query).
 *
 * Each group cites the reference interface it replaces (paths under
 * /root/reference/lib/model).  INTEGRATION.md shows the ctypes stubs a maintainer of
 * the reference would add in place of the torch.utils.ffi `_ext` modules / `model._C`.
 *
 * Layouts.  Feature maps are NHWC ("channels_last": (B,H,W,C), C contiguous), the
 * layout the MFMA implicit-GEMM convolutions produce and consume.  ROI outputs can
 * be written NHWC (R,PH,PW,C) for the HIP heads or NCHW (R,C,PH,PW) for reference
 * callers (flatten order of vrd.fc6, resnet_SGG_emb.py:146).  rois are (R,5) fp32
 * [batch_idx, x1, y1, x2, y2] in image coordinates, as in the reference.
 */
#ifndef I2VSGG_HIP_H
#define I2VSGG_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2V_OK               0
#define I2V_ERR_ARG         -1   /* bad shape / null pointer / unsupported size   */
#define I2V_ERR_LAUNCH      -2   /* hipGetLastError() after a launch               */
#define I2V_ERR_WORKSPACE   -3   /* workspace too small                            */
#define I2V_ERR_UNSUPPORTED -4

#define I2V_LAYOUT_NHWC 0
#define I2V_LAYOUT_NCHW 1

/* epilogue flags of i2v_conv_fwd */
#define I2V_EPI_RELU      1
#define I2V_EPI_RESIDUAL  2
#define I2V_EPI_SCALE     4      /* y = acc*scale[n] + shift[n]  (frozen BN)       */
#define I2V_EPI_BIAS      8      /* y = acc + shift[n]                             */
#define I2V_EPI_ZEROED   16      /* caller guarantees y is all zeros (skips the split-K clear) */
#define I2V_EPI_MASK     32      /* last: y = mask > 0 ? y : 0 (i2v_conv_dgrad_fused: the ReLU a data gradient flows back into) */

int32_t     i2v_version(void);
const char* i2v_last_error(void);
/* 0 (rounds 3-5: bit 0 = a build with the experiment kernels, which left the tree in round 6; kept so that old callers link) */
int32_t     i2v_build_flags(void);

/* ---- streams of the host's own ---------------------------------------------------
 * The reference launches on "the current stream" of a global THCState
 * (roi_align/src/roi_align_cuda.c:5,:31; roi_pooling_cuda.c; nms_cuda.c).  Every entry
 * point below takes its stream explicitly instead; a host that forks work (the graph
 * branches of a step) creates the streams it forks onto HERE, so that they cannot
 * alias a stream a framework deals from a shared pool.  Non-blocking streams;
 * priority 0 = normal, < 0 = higher (clamped to the device's range). */
int32_t i2v_stream_create(int32_t device, int32_t priority, void** stream);
int32_t i2v_stream_destroy(void* stream);

/* ---- ROIAlign (legacy "aligned grid incl. both ends" variant) ------------------
 * replaces roi_align/src/roi_align_cuda.h:1-5 (roi_align_forward_cuda /
 * roi_align_backward_cuda), launchers roi_align_kernel.h:13-27, and -- fused with the
 * 2x2 stride-1 average -- RoIAlignAvg (roi_align/modules/roi_align.py:18-29).
 * `pooled_*` is the FINAL grid (7x7); avg=1 samples (pooled+1)^2 points and averages,
 * avg=0 samples pooled^2 points (plain RoIAlign).  feat is NHWC (feat_layout must be
 * I2V_LAYOUT_NHWC) or NCHW.  bwd ACCUMULATES into grad_feat (caller zero-fills, as
 * roi_align/functions/roi_align.py:42-43 does); fp32 atomics, summation order free. */
int32_t i2v_roi_align_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                          const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                          float spatial_scale, int32_t avg, float* out, int32_t out_layout, void* stream);
int32_t i2v_roi_align_bwd(const float* grad_out, int32_t out_layout, const float* rois, int32_t R,
                          int32_t pooled_h, int32_t pooled_w, float spatial_scale, int32_t avg,
                          float* grad_feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                          void* stream);

/* The same backward as a GATHER (NHWC grad_out (R,PH,PW,C) and NHWC grad_feat, C % 128 == 0): every element of grad_feat is
 * WRITTEN once (no zero-fill by the caller, no atomics) as the sum of its contributions in the serial order of
 * roi_align_kernel.cu:94-143 (roi, sample row, sample column ascending): deterministic.  pooled_w + avg <= 8, any pooled_h.
 * One kernel since round 5: no workspace is needed (the query returns 0; ``workspace`` may be NULL -- both stay in the
 * signature for callers built against round 4). */
size_t  i2v_roi_align_bwd_gather_workspace_bytes(int32_t R, int32_t C, int32_t pooled_h, int32_t pooled_w, int32_t avg);
int32_t i2v_roi_align_bwd_gather(const float* grad_out, const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                                 float spatial_scale, int32_t avg, float* grad_feat, int32_t B, int32_t C, int32_t H, int32_t W,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- ROIAlign, sampled variant (roi_layers.ROIAlign) ------------------------------
 * replaces model._C.roi_align_forward / roi_align_backward as bound by roi_layers/roi_align.py:20,:31-42
 * (called as (input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio) and
 * (grad, rois, spatial_scale, pooled_h, pooled_w, B, C, H, W, sampling_ratio)); constructed with
 * sampling_ratio 0 at faster_rcnn_SGG_emb.py:47.  model._C is the maskrcnn-benchmark csrc, absent from the
 * reference tree: the published algorithm is restated (no +1 on the extent, extent >= 1, mean of a
 * sampling_ratio^2 -- or ceil(extent/pooled)^2 when sampling_ratio <= 0 -- grid of bilinear samples per bin,
 * samples more than a pixel outside the map contribute 0).  A roi whose batch index is outside [0,B) yields
 * zeros / no gradient.  bwd accumulates into grad_feat (caller pre-zeroes), fp32 atomics. */
int32_t i2v_roi_align_sampled_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                                  const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                                  float spatial_scale, int32_t sampling_ratio, float* out, int32_t out_layout,
                                  void* stream);
int32_t i2v_roi_align_sampled_bwd(const float* grad_out, int32_t out_layout, const float* rois, int32_t R,
                                  int32_t pooled_h, int32_t pooled_w, float spatial_scale, int32_t sampling_ratio,
                                  float* grad_feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                                  void* stream);

/* ---- ROIPool (Caffe max pooling) ----------------------------------------------
 * replaces roi_pooling/src/roi_pooling_cuda.c:7-8,49-50 and model._C.roi_pool_forward /
 * roi_pool_backward (roi_layers/roi_pool.py:17,30).  argmax holds h*W+w of the winning
 * input pixel or -1 (int32, same shape/layout as out).  bwd accumulates into grad_feat. */
int32_t i2v_roi_pool_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                         const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                         float spatial_scale, float* out, int32_t* argmax, int32_t out_layout, void* stream);
/* The same forward with the extent of the maps read from DEVICE memory: geom[0] = H, geom[1] = W (int32).  `feat` holds
 * B packed NHWC maps of H*W cells (map b at b*H*W*C) at the start of a buffer of at least B*cap_cells*C floats.  A launch
 * recorded in a HIP graph thereby serves every map size that fits the buffer: the captured relation-head step consumes
 * roi_data_layer batches whose padded image size differs from batch to batch (roibatchLoader.py:162-190) without one
 * graph per size.  H*W > cap_cells (or H, W <= 0) yields zeros / argmax -1, never an out-of-bounds read. */
int32_t i2v_roi_pool_fwd_geom(const float* feat, int32_t B, int32_t C, const int32_t* geom, int64_t cap_cells,
                              const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w, float spatial_scale,
                              float* out, int32_t* argmax, int32_t out_layout, void* stream);
int32_t i2v_roi_pool_bwd(const float* grad_out, const int32_t* argmax, int32_t out_layout,
                         const float* rois, int32_t R, int32_t pooled_h, int32_t pooled_w,
                         float* grad_feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                         void* stream);

/* ---- NMS -----------------------------------------------------------------------
 * replaces nms/src/nms_cuda.h:4-5 (nms_cuda) / nms_cuda_kernel.h:5-6 (nms_cuda_compute)
 * and the live CPU path nms/nms_cpu.py:6-34.  dets (n_img, n, 5) fp32 rows
 * [x1,y1,x2,y2,score] ALREADY in descending score order; one independent problem per
 * image.  keep_out (n_img, n) int32: kept row indices in order, first num_out[img]
 * valid.  max_keep>0 stops each scan after that many kept rows (post_nms_topN).
 * Fully asynchronous: counts stay on the device. */
size_t  i2v_nms_workspace_bytes(int32_t n_img, int32_t n);
int32_t i2v_nms_sorted(const float* dets, int32_t n_img, int32_t n, float thresh, int32_t max_keep,
                       int32_t* keep_out, int32_t* num_out, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- RPN proposal layer ----------------------------------------------------------
 * replaces rpn/proposal_layer.py:49-163 incl. generate_anchors.py:45-56,
 * bbox_transform.py:77-103 (bbox_transform_inv), :125-133 (clip_boxes), the sort at
 * proposal_layer.py:127 and the per-image host NMS at :150.
 *   cls   (B,H,W,2A) NHWC RPN class scores; fg score of anchor a = 2-way softmax of
 *         (cls[..,a], cls[..,A+a]) as rpn.py:69-71 (is_prob=1: cls already holds the
 *         probabilities, fg = cls[..,A+a])
 *   bbox  (B,H,W,4A) NHWC deltas;  im_info (B,3) [h,w,scale] on the device
 *   base_anchors (A,4) fp32 on the device (generate_anchors output)
 *   rois  (B,post_nms_top_n,5) zero padded, col0 = image index
 *   kept_idx (B,post_nms_top_n) int32 anchor index (y*W+x)*A+a of each roi or -1 (may be NULL)
 *   num_kept (B) int32 (may be NULL)
 * Order on exactly tied scores: descending score, then ascending anchor index. */
size_t  i2v_rpn_proposal_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t A, int32_t pre_nms_top_n);
int32_t i2v_rpn_proposal(const float* cls, int32_t is_prob, const float* bbox, const float* im_info,
                         const float* base_anchors, int32_t B, int32_t H, int32_t W, int32_t A,
                         int32_t feat_stride, int32_t pre_nms_top_n, int32_t post_nms_top_n, float nms_thresh,
                         float* rois, int32_t* kept_idx, int32_t* num_kept,
                         void* workspace, size_t workspace_bytes, void* stream);

/* pieces of the above, exported for parity tests and reuse */
int32_t i2v_rpn_decode(const float* cls, int32_t is_prob, const float* bbox, const float* im_info,
                       const float* base_anchors, int32_t B, int32_t H, int32_t W, int32_t A, int32_t feat_stride,
                       float* proposals /* (B,HWA,4) */, float* scores /* (B,HWA) */, void* stream);
size_t  i2v_sort_desc_workspace_bytes(int32_t n_seg, int32_t n);
int32_t i2v_sort_desc(const float* keys, int32_t n_seg, int32_t n, int32_t* order_out /* (n_seg,n) */,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- image front-end of the data layer ------------------------------
 * replaces, per image, roi_data_layer/minibatch.py:60-90 + model/utils/blob.py:35-52 and :19-33: RGB->BGR,
 * optional horizontal flip, float conversion, PIXEL_MEANS subtraction, cv2.resize(fx=fy=target/shorter side,
 * INTER_LINEAR) and the placement into the zero-padded batch blob.  img: decoded uint8 H x W x 3 on the device
 * (rgb_order != 0: file order R,G,B); pixel_means_bgr: 3 host floats (cfg.PIXEL_MEANS); blob: (blob_h, blob_w, 4)
 * fp32 NHWC slice of the batch tensor, caller-zeroed, channel 3 stays 0 (the stem's Cin % 4 pad, folded in).
 * i2v_image_prep_size gives the resized size and im_scale (im_info = [Ho, Wo, scale], minibatch.py:49-51). */
int32_t i2v_image_prep_size(int32_t H, int32_t W, int32_t target_size, int32_t* Ho, int32_t* Wo, float* scale);
int32_t i2v_image_prep(const uint8_t* img, int32_t H, int32_t W, int32_t rgb_order, int32_t flipped,
                       const float* pixel_means_bgr, int32_t target_size, float* blob, int32_t blob_h,
                       int32_t blob_w, void* stream);

/* ---- per-class detection post-processing (eval) --------------------
 * replaces test_net_instance_styleD_bilinear.py:151-221 for one image: de-normalise the deltas with
 * TRAIN.BBOX_NORMALIZE_STDS / _MEANS (host arrays of 4 floats, NULL = not normalised), bbox_transform_inv
 * against rois (R,5), clip to the image, divide by the image scale, and for every class j >= 1: keep
 * score > score_thresh, sort, NMS(nms_thresh); then, if more than max_per_image detections survive over all
 * classes, keep those with score >= the max_per_image-th largest.  One asynchronous pass (the reference makes
 * n_classes - 1 host NMS calls per frame).
 *   cls_prob (R,C); bbox_pred (R,4) when class_agnostic else (R,4C)
 *   dets (C,R,5) [x1,y1,x2,y2,score], rows of class j in descending score order, first counts[j] valid;
 *   counts (C) int32, counts[0] = 0.  Ties: descending score, then ascending roi index. */
size_t  i2v_det_postprocess_workspace_bytes(int32_t R, int32_t C);
int32_t i2v_det_postprocess(const float* rois, const float* cls_prob, const float* bbox_pred, int32_t class_agnostic,
                            const float* stds, const float* means, float im_h, float im_w, float im_scale,
                            int32_t R, int32_t C, float score_thresh, float nms_thresh, int32_t max_per_image,
                            float* dets, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the frame's im_info row [height, width, scale] read from DEVICE memory (3 floats) by the kernel: the
 * launch arguments are then the same for every frame, so a captured per-frame evaluation step (eval.DetectStep) serves
 * frames of any scale. */
int32_t i2v_det_postprocess_info(const float* rois, const float* cls_prob, const float* bbox_pred, int32_t class_agnostic,
                                 const float* stds, const float* means, const float* im_info,
                                 int32_t R, int32_t C, float score_thresh, float nms_thresh, int32_t max_per_image,
                                 float* dets, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- relation triplet ranking (eval) -----------------------------------
 * replaces the scoring loop and the argsort of detection_output (lib/utils.py:584-628): cell (i, r) of rel_score
 * (n_pairs, n_rel) is multiplied by conf[ixs[i]] and conf[ixo[i]] (two fp32 roundings) and the k largest cells are
 * returned in descending order (ties: ascending flat index): pair_out (k) = i, pred_out (k) = r, conf_out (k). */
size_t  i2v_relation_topk_workspace_bytes(int32_t n_pairs, int32_t n_rel);
int32_t i2v_relation_topk(const float* rel_score, const float* conf, const int64_t* ixs, const int64_t* ixo,
                          int32_t n_pairs, int32_t n_rel, int32_t k, int32_t* pair_out, int32_t* pred_out,
                          float* conf_out, void* workspace, size_t workspace_bytes, void* stream);

/* IoU of boxes (B,N,4 | stride_box floats per row, first 4 used after `box_off`) against
 * gt (B,K,5): bbox_transform.py:168-257 (bbox_overlaps_batch) incl. the zero-area
 * masks; also emits per-row max/argmax (first max).  overlaps may be NULL. */
int32_t i2v_bbox_overlaps(const float* boxes, int32_t box_stride, int32_t box_off, int32_t boxes_batched,
                          const float* gt, int32_t B, int32_t N, int32_t K,
                          float* overlaps, float* max_ov, int32_t* argmax_ov, void* stream);

/* ---- convolution / linear as MFMA implicit GEMM (fp32 in, fp32 accumulate) --------
 * replaces the cuDNN-backed nn.Conv2d + frozen nn.BatchNorm2d + ReLU + residual add of
 * Bottleneck.forward (resnet_instance_styleD_bilinear.py:197-217), the stem (:224-228),
 * the RPN convs (rpn/rpn.py:27-36), the 1x1 discriminator convs (:41-46) and -- as a
 * 1x1 conv over (M,1,1,K) -- every nn.Linear of the vrd head (resnet_SGG_emb.py:83-127).
 *   x   (B,H,W,Cin) NHWC, Cin % 4 == 0        w  (Cout,KH,KW,Cin)  (K-major per filter)
 *   y   (B,Ho,Wo,Cout) NHWC                    res same shape as y (I2V_EPI_RESIDUAL)
 *   y = epi(sum_k x*w) with epi per flags: *scale[n] +shift[n], +res, relu.
 * dgrad: gx (B,H,W,Cin) = conv_transpose(gy, w) (overwrites gx); the workspace holds the
 *        flipped/transposed filter(s).  stride>1: 1x1 filters scatter to every s-th pixel; KxK filters run
 *        one sub-filter correlation per input-pixel parity class (s*s launches, dense MAC count).
 * wgrad: gw (Cout,KH,KW,Cin) (+)= gy^T * im2col(x); beta=0 overwrites, beta=1 accumulates. */
int32_t i2v_conv_fwd(const float* x, const float* w, const float* scale, const float* shift, const float* res,
                     float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t flags,
                     void* split_workspace, size_t split_workspace_bytes, void* stream);
/* Small-M shapes split K over several workgroups per tile.  With a CALLER-PROVIDED split-K workspace
 * (i2v_conv_split_workspace_bytes for the shape; NULL / 0 = none) the partial tiles go through it and the last
 * workgroup of a tile sums them in split order and applies the epilogue: deterministic, y needs no clear.
 * Workspace contract: device memory, 256-B aligned; its first 4096 bytes (the arrival counters) must be zero before
 * the first launch that uses it and every launch leaves them zero again; launches that share a workspace must be
 * ordered on the device (one stream, or graph edges) -- two launches that may run concurrently need two workspaces.
 * The library owns no device memory and keeps no per-stream state.
 * i2v_conv_fwd_splits returns 1 when this shape, given a workspace of ws_bytes, would instead accumulate with fp32
 * atomics into y (no / too small a workspace, more than 4 splits, or a small output): y must then start at zero --
 * the call clears it unless I2V_EPI_ZEROED is passed, which lets a caller batch many clears into one.  0 otherwise,
 * < 0 on error. */
int32_t i2v_conv_fwd_splits(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                            int32_t stride, int32_t pad, size_t ws_bytes);
size_t  i2v_conv_split_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH,
                                       int32_t KW, int32_t stride, int32_t pad);
/* Tuning knobs (process-wide, host side only; the library reads no environment variable).  key = I2V_TUNE_*. */
#define I2V_TUNE_CONV_SPEC            0   /* retired (rounds 1-5: the 8-wave loader / MFMA specialisation of conv_igemm_f32): only <= 0 is accepted */
#define I2V_TUNE_SPLIT_TARGET         1   /* workgroups per CU a split-K launch aims for (default 2) */
#define I2V_TUNE_SPLIT_TARGET_SKINNY  2   /* the same for GEMMs of <= 256 rows (-1: as SPLIT_TARGET) */
#define I2V_TUNE_SPLIT_BELOW          3   /* split K only when the unsplit grid has fewer tiles than this (default 256) */
#define I2V_TUNE_SPLIT_ATOMICS        4   /* how reductions that are split across workgroups are finished.  2 (process default, round 4's rule): split-K GEMMs of up to four parts and outputs of >= 2^18 elements through the caller's workspace, summed in split order by the last workgroup to arrive; everything else (more parts, small outputs, split filter gradients, bias column sums) with fp32 atomics.  0: EVERY such reduction ordered (what both step objects select for their launch contexts: bit-reproducible results) -- any number of split-K parts; filter gradients of up to 16 parts in the kernel, of more parts and of filters with Cout % 4 != 0 through side-by-side partial filters and a reduce pass (round 6); the 36-plane filter gradient of the Winograd domain (its final transform adds the parts); bias column sums of any size in two levels of 32 row blocks (round 6).  A reduction whose workspace is missing or too small falls back to atomics and is counted (i2v_ordered_fallbacks).  1: always atomics */
#define I2V_TUNE_BIG_FC_TILE          5   /* tile index for the long skinny GEMMs (rows <= 256, K >= 16384); -1: cost model */
#define I2V_TUNE_WGRAD_V2             6   /* 0: first-generation wgrad kernel, 1: default, 2/3: larger tiles */
#define I2V_TUNE_WGRAD_FUSED_TILE     7   /* 128 (default) or 64: filters per workgroup of the fused wgrad+SGD kernel */
#define I2V_TUNE_WINO_ROWS            8   /* bit 0 / 1: row-split Winograd input / output transform; -1: by size (small launches split); default 0: never -- the split wins for a layer3 launch alone (34 against 37 us) and loses inside the step graphs, whose branches already fill the chip (4.59 against 4.61 ms) */
#define I2V_TUNE_ROIPOOL_C128         9   /* 1 (default): 128-channel ROIPool forward kernel for NHWC maps */
#define I2V_TUNE_CONV_GEMM           10   /* 1 (default): pointwise layers / plain GEMMs on the lean conv_gemm_f32 kernel */
#define I2V_TUNE_STAGGER             11   /* retired (start-time stagger of co-resident workgroups): only 0 is accepted */
#define I2V_TUNE_ROIALIGN_COLS       12   /* ROIAlign forward, NHWC: 2 (default) = one ROI x 128 channels per workgroup, sample values meet in LDS, each tap through the L2 once (NHWC output, <= 64 sample points; else as 1); 1 = column-pair workgroups (round 2); 0 = one output row per workgroup (round 1) */
#define I2V_TUNE_WGRAD_PER_CU        13   /* workgroups per CU a split-over-pixels wgrad launch aims for (default 4) */
#define I2V_TUNE_WGRAD_XCD           14   /* 1 (default): a filter-gradient split's tiles share an XCD when the split count is a multiple of 8 */
#define I2V_TUNE_FC_FOLD             15   /* retired (ablation bits of the fc-fold kernel, DESIGN_HISTORY.md 5.6): only 0 is accepted */
#define I2V_TUNE_GEMM_X3             16   /* retired (three-term bf16 split of the pointwise GEMM, DESIGN_HISTORY.md 5.7): only 0 is accepted */
#define I2V_TUNE_GEMM_PERSIST        17   /* retired (persistent form of the pointwise GEMM, profiles/r03_persistent_gemm.txt): only 0 is accepted */
#define I2V_TUNE_WGRAD_PRIO          18   /* retired (falling wave priority in the filter-gradient kernel): only 0 is accepted */
#define I2V_TUNE_STREAM_TILE         19   /* 1 (default): pointwise layers of at most four K stages over >= 16384 rows (HBM-bound) take the 80x64 tile whatever the cost model says; 0: cost model */
#define I2V_TUNE_KGROUPS             20   /* 2 (round 5): the same with TWO wave groups (8 waves, two 39 KB stage regions: the workgroup shares its CU); 1 / 4: a pointwise GEMM the plan would split over K runs as one 16-wave workgroup per tile whose four wave groups split K and meet in LDS (no partial tile through memory) when one round of such tiles covers >= 70 % of the CUs: 4-10 % faster than the split across workgroups as a kernel on its own, 9 % SLOWER inside the overlapped step (a workgroup that owns a CU's LDS and registers shuts the other branches' workgroups out: profiles/r04_kgroups.txt); 0 (default): split-K across workgroups, partials through the caller's workspace */
#define I2V_TUNE_WGRAD_ORDERED_GFLOP 21   /* a split filter gradient takes an ordered finish (I2V_TUNE_SPLIT_ATOMICS == 0) only when the problem is below this many GFLOP (default 1000000: always; round 5's default was 8 -- its only ordered form, one finisher reading every part, was a tail on the large launches; 0: never) */
#define I2V_TUNE_GEMM_DMA            22   /* how the pointwise / plain-GEMM kernel stages its operand tiles (round 6).  0: global -> registers -> ds_write_b128 (rounds 2-5).  1: LDS-DMA (buffer_load ... lds, the column swizzle on the source address), 32-k stages, same LDS image.  2: LDS-DMA with 16-k stages (64-byte LDS rows): half the LDS per workgroup, twice the barriers.  Bit-equal results in all three */
#define I2V_TUNE_WGRAD_DMA           23   /* staging of the second-generation filter-gradient kernel on pointwise / linear problems (round 6).  0: global -> registers -> transposing ds_write_b128.  1: LDS-DMA into the [pixel][column] image, the group swizzle on the source column.  Bit-equal */
#define I2V_TUNE_ROIALIGN_BWD        24   /* the gather form of the RoIAlign backward (round 6).  1: a wave owns 32 channels of the row buffer, a lane = (sample column, 4 channels), gradients from grad_out to registers, 16-byte read-add-writes.  0: round 5's form (lane = channel x cell parity, gradients and tap records staged in LDS per batch of four pairs) */
#define I2V_TUNE_NMS_SCAN            25   /* the greedy scan of NMS for up to 12288 boxes (round 6).  2 (default): super-blocks of 1024 rows whose triangle of the mask sits in LDS, resolved by fixed-point sweeps (keep <- alive & ~OR of the kept rows' words, from keep = alive: exact at its fixed point, a few sweeps), a serial resolve for a super-block that has not settled after 48 sweeps; a value v > 2: the same with v sweeps at most.  1: the super-blocks with the serial resolve only.  0: round 5's scan (64-row blocks, two barriers each).  The same keep lists in every mode */
#define I2V_TUNE_COUNT               26
/* Keys CONV_SPEC (> 0), STAGGER, FC_FOLD, GEMM_X3, GEMM_PERSIST, WGRAD_PRIO belonged to kernel variants that were measured and lost
 * (DESIGN_HISTORY.md) and left the library in round 6: I2V_ERR_UNSUPPORTED for any value but "off"; the indices stay reserved. */
int32_t i2v_set_tuning(int32_t key, int32_t value);
int32_t i2v_get_tuning(int32_t key);
/* tuning hook: cfg < 0 = cost model; else cfg = tile shape 0..5 (0xFF = cost model).  Bits above the low byte selected
 * experiment kernels in rounds 1-5: I2V_ERR_UNSUPPORTED now */
int32_t i2v_conv_set_tile(int32_t cfg);
/* diagnostic: when buf != NULL every conv workgroup writes 8 u64 to buf[8*wg ..]: {K-loop shader cycles, 100 MHz
 * ticks since kernel start, setup cycles, total cycles, ...}; the in-kernel clock is total cycles / ticks * 100 MHz */
int32_t i2v_conv_debug_clock(void* buf);
/* diagnostic: one-lane kernel writing {shader-clock counter, 100 MHz counter} to out2[0..1] on `stream`; two stamps around
 * a stretch of work give the shader clock the chip held over it (tools/step_clock.py) */
int32_t i2v_debug_clock_stamp(void* out2, void* stream);
/* Backward-protocol forms of the data / filter gradient (the instance_styleD training step, where layer1-3 and layer4
 * train behind frozen BatchNorms: trainval_net_instance_styleD_bilinear.py:262-341, resnet_instance_styleD_bilinear.py:181-217).
 * Between two convolutions of a bottleneck sit a frozen-BN scale and a ReLU; instead of one streaming pass per layer over
 * the activation gradient (g = gy * (y > 0) * scale) the factors ride in the kernels that touch the data anyway:
 *   i2v_conv_dgrad_fused   gx = mask > 0 ? (dgrad(gy * gy_scale[cout], w) * out_scale[cin] + res) : 0
 *                          gy_scale: folded into the transposed filter; out_scale / res / mask: epilogue operands, each
 *                          may be NULL; res and mask have the shape of gx.  Stride 1, or a strided 1x1 layer (then gx is zero off the stride grid and res must be too); I2V_ERR_UNSUPPORTED otherwise.
 *   i2v_conv_wgrad_scaled  gw[n] = row_scale[n] * wgrad(x, gy)[n]   (the BN scale that belongs on gy, applied once per
 *                          filter row where the reduction over the pixels ends); beta as in i2v_conv_wgrad.
 *   i2v_conv3x3_winograd4_dgrad  gx = mask > 0 ? (winograd F(4x4,3x3) data gradient * out_scale[cin]) : 0,
 *                          U = i2v_winograd4_filter_dgrad(w); workspace i2v_conv3x3_winograd4_workspace_bytes(B,H,W,Cout,Cin). */
int32_t i2v_conv_dgrad_fused(const float* gy, const float* w, const float* gy_scale, const float* out_scale,
                             const float* res, const float* mask, float* gx, int32_t B, int32_t H, int32_t W,
                             int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                             void* workspace, size_t workspace_bytes, void* split_workspace, size_t split_workspace_bytes,
                             void* stream);
int32_t i2v_conv_wgrad_scaled(const float* x, const float* gy, const float* row_scale, float* gw, int32_t B,
                              int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                              int32_t stride, int32_t pad, float beta, void* split_workspace, size_t split_workspace_bytes,
                              void* stream);      /* split_workspace: as i2v_conv_wgrad's (round 6); NULL: fp32 atomics */
/* Reductions that were asked to be ordered (I2V_TUNE_SPLIT_ATOMICS == 0) but ran on fp32 atomics because the caller's split
 * workspace was absent or too small (or a size cap was exceeded), since the last reset: a step that promises bit-reproducible
 * results asserts 0 after its warm-up.  reset != 0: zero the count after reading it. */
int32_t i2v_ordered_fallbacks(int32_t reset);
int32_t i2v_conv3x3_winograd4_dgrad(const float* gy, const float* U, const float* out_scale, const float* mask,
                                    float* gx, int32_t B, int32_t H, int32_t W, int32_t Cout, int32_t Cin,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* Filter gradient of a stride-1 / pad-1 3x3 layer in the Winograd F(4x4,3x3) domain (a quarter of the direct form's MACs:
 * 36 plane GEMMs reduced over the 4x4 tiles):  gw (Cout,3,3,Cin) = beta * gw + row_scale[cout] * wgrad(x, gy), beta 0 or 1,
 * row_scale may be NULL.  The caller's workspace holds both transformed operands and the 36 partial planes. */
size_t  i2v_conv3x3_winograd4_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int32_t i2v_conv3x3_winograd4_wgrad(const float* x, const float* gy, const float* row_scale, float* gw,
                                    int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float beta,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* A trained layer transforms its input once: the forward leaves V (i2v_conv3x3_winograd4_v_bytes) in a caller buffer,
 * the filter gradient of the same step reads it (`_wgrad_v`) instead of transforming x again. */
size_t  i2v_conv3x3_winograd4_v_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin);
int32_t i2v_conv3x3_winograd4_fwd_keep(const float* x, const float* U, const float* scale, const float* shift,
                                       float* y, float* v_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                       int32_t Cout, int32_t relu, void* workspace, size_t workspace_bytes, void* stream);
int32_t i2v_conv3x3_winograd4_wgrad_v(const float* v, const float* gy, const float* row_scale, float* gw,
                                      int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float beta,
                                      void* workspace, size_t workspace_bytes, void* stream);
/* gw[z] (N x K) = gy[z]^T (M x N) . x[z] (M x K), z < nbatch (the plane GEMMs above); N % 4 == K % 4 == 0 */
int32_t i2v_gemm_tn_batched(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K,
                            int32_t nbatch, long long stride_x, long long stride_gy, long long stride_gw, void* stream);
/* the same accumulating into gw (gw += ...: cleared or pre-loaded by the caller; no memset in front) */
int32_t i2v_gemm_tn_batched_acc(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K,
                            int32_t nbatch, long long stride_x, long long stride_gy, long long stride_gw, void* stream);
size_t  i2v_conv_dgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t KH, int32_t KW);
int32_t i2v_conv_dgrad(const float* gy, const float* w, float* gx, int32_t B, int32_t H, int32_t W,
                       int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                       void* workspace, size_t workspace_bytes, void* split_workspace, size_t split_workspace_bytes,
                       void* stream);
size_t  i2v_conv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                       int32_t KH, int32_t KW, int32_t stride, int32_t pad);
/* i2v_conv_wgrad's workspace (round 5) is the caller's SPLIT workspace (i2v_conv_fwd's: counters zero between launches + slab): a
 * reduction over the pixels that is split across workgroups (up to 16 parts; small problems are capped there) is then summed in split
 * order by the tile's last workgroup -- bit-reproducible, no clear of gw in front.  NULL: fp32 atomics into a cleared gw. */
int32_t i2v_conv_wgrad(const float* x, const float* gy, float* gw, int32_t B, int32_t H, int32_t W,
                       int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                       float beta, void* workspace, size_t workspace_bytes, void* stream);

/* nbatch independent GEMMs of one shape in one launch: C_z (M x N) = A_z (M x K) * B_z (N x K)^T (fp32; operand z at
 * base + z * stride elements; K % 4 == 0). */
int32_t i2v_gemm_nt_batched(const float* a, const float* b, float* c, int32_t M, int32_t N, int32_t K, int32_t nbatch,
                            int64_t stride_a, int64_t stride_b, int64_t stride_c,
                            void* split_workspace, size_t split_workspace_bytes, void* stream);

/* Winograd F(2x2,3x3) forward for stride-1 / pad-1 3x3 convolutions whose filter is FROZEN (the SGG_emb backbone:
 * resnet_instance_styleD_bilinear.py:197-217 under the detach of faster_rcnn_SGG_emb.py:148): 2.25x fewer MACs than
 * the direct form.  i2v_winograd_filter transforms w (Cout,3,3,Cin) into U (16,Cout,Cin) once; the forward runs the
 * input transform, ONE batched GEMM launch over the 16 planes and the output transform with the frozen-BN
 * scale/shift (either may be NULL) and optional ReLU.  Same result as i2v_conv_fwd up to fp32 rounding (~1e-6 rel). */
int32_t i2v_winograd_filter(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream);
size_t  i2v_conv3x3_winograd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int32_t i2v_conv3x3_winograd_fwd(const float* x, const float* U, const float* scale, const float* shift, float* y,
                                 int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t relu,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The same with F(4x4,3x3): 4x fewer MACs than direct, U is (36,Cout,Cin), fp32 error ~1e-5 relative (transform
 * constants up to 8 and 1/24) instead of ~1e-6. */
int32_t i2v_winograd4_filter(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream);
/* Winograd-domain filter of the data gradient of the same layer: U'[36][Cin][Cout] from w (Cout,3,3,Cin), taps flipped
 * and channels swapped, so that gx = i2v_conv3x3_winograd4_fwd(gy, U', NULL, NULL, gx, B, H, W, Cout, Cin, 0, ...) is
 * the dgrad of a stride-1 / pad-1 3x3 convolution (the 3x3 of Bottleneck.backward when the layer is trained,
 * resnet_instance_styleD_bilinear.py:203-205 under autograd). */
int32_t i2v_winograd4_filter_dgrad(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream);
size_t  i2v_conv3x3_winograd4_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout);
int32_t i2v_conv3x3_winograd4_fwd(const float* x, const float* U, const float* scale, const float* shift, float* y,
                                  int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t relu,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* wgrad with the SGD(momentum) step of that filter fused into the epilogue: w and m are updated in place and
 * the gradient is never written (g' = g + wd*w; m = mom*m + g'; w -= lr*m).  Only for shapes whose pixel
 * reduction needs no split (M <= 4096 and >= 512 filter tiles), else I2V_ERR_UNSUPPORTED. */
int32_t i2v_conv_wgrad_sgd(const float* x, const float* gy, float* w, float* m, int32_t B, int32_t H, int32_t W,
                           int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                           float lr, float momentum, float weight_decay, void* stream);

/* epilogue backward, one streaming pass: g_pre = gy * (y>0) [relu]; g = g_pre * scale[n] (scale may be NULL);
 * gbias[n] += column sums of g_pre.  g, gpre and gbias may each be NULL; in-place (g == gy or gpre == gy) allowed.
 * g_t (may be NULL): g once more, column-major (N x M) -- the operand a linear layer's data gradient reads when it
 * runs on the filter-gradient kernel with the roles swapped (saves the separate transpose of the small gradient).
 * split_ws / split_ws_bytes: the caller's split workspace (i2v_conv_fwd's: counters zero between launches + slab).  With it
 * the row blocks' column sums are added in block order by the last block to arrive (bit-reproducible); NULL: fp32 atomics. */
int32_t i2v_epilogue_bwd(const float* gy, const float* y, const float* scale, float* g, float* gpre, float* gbias,
                         int64_t M, int32_t N, int32_t relu, float* g_t, void* split_ws, size_t split_ws_bytes, void* stream);

/* 3x3 / stride 2 / pad 0 / ceil_mode max pool of the stem (resnet_instance...:228), NHWC */
int32_t i2v_maxpool3x3s2_fwd(const float* x, float* y, int32_t* argmax, int32_t B, int32_t H, int32_t W, int32_t C,
                             void* stream);

/* ---- netD_style factorised-bilinear pooling ------------------------------------------
 * replaces the x1*x2 product, the rank sum and the spatial sum of netD_style.forward
 * (resnet_instance_styleD_bilinear.py:131-136) with one streaming pass:
 *   z[b][d] = sum_{row<rows} sum_{r<rank} x1[b][row][d*rank+r] * x2[b][row][d*rank+r]
 * x1,x2 (n_img, rows, dim*rank) fp32; z (n_img, dim) (overwritten).  bwd writes
 * g1 = gz[b][d]*x2 and g2 = gz[b][d]*x1 (same shapes as x1/x2). */
int32_t i2v_dstyle_pool_fwd(const float* x1, const float* x2, float* z, int64_t rows, int32_t n_img,
                            int32_t dim, int32_t rank, void* stream);
int32_t i2v_dstyle_pool_bwd(const float* gz, const float* x1, const float* x2, float* g1, float* g2,
                            int64_t rows, int32_t n_img, int32_t dim, int32_t rank, void* stream);
/* netD_style's bilinear pooling fused into the projection GEMMs (resnet_instance_styleD_bilinear.py:122-136): one kernel
 * computes x1 = x*W1^T + b1 and x2 = x*W2^T + b2 tile by tile (x (n_img*rows, k); W (dim*rank, k) each) and reduces
 * x1*x2 over the tile's positions in its epilogue; a second small kernel sums the per-tile partial rows (fixed order:
 * reproducible) and the rank groups into z (n_img, dim).  x1 / x2 (n_img*rows, dim*rank) are written only when both
 * pointers are given (a training step keeps them for i2v_dstyle_pool_bwd); NULL, NULL = nothing but z leaves the kernel. */
size_t  i2v_dstyle_fused_workspace_bytes(int64_t rows, int32_t n_img, int32_t dim, int32_t rank);
int32_t i2v_dstyle_fused_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                             float* z, float* x1, float* x2, int64_t rows, int32_t n_img, int32_t k, int32_t dim,
                             int32_t rank, void* workspace, size_t workspace_bytes, void* stream);

/* ---- small fused pieces of the relation head's tail ---------------------------------------------
 * rows here are 64 x 300, so each aten op of the reference expression is one launch-bound kernel.
 * l2norm_rows: y = x / max(||x||_2, eps) per row = F.normalize(x, p=2, dim=1) (resnet_SGG_emb.py:210-213) and its
 *   backward gx = (g - y (g.y)) / max(||x||, eps); inv_norm (rows) is kept for the backward.
 * bce_rows: loss = sum_r w[r] * mean_c BCEWithLogits(z[r][c], t[r][c]) (faster_rcnn_SGG_emb.py:269 with the per-frame
 *   mean folded into w) -> one device scalar, written (not accumulated) by a single workgroup in a fixed summation
 *   order; backward gz = (sigmoid(z) - t) * w[r] / cols * gloss[0]. */
int32_t i2v_l2norm_rows_fwd(const float* x, float* y, float* inv_norm, int32_t rows, int32_t cols, float eps, void* stream);
int32_t i2v_l2norm_rows_bwd(const float* g, const float* y, const float* inv_norm, float* gx, int32_t rows, int32_t cols,
                            float eps, void* stream);
int32_t i2v_bce_rows_fwd(const float* z, const float* t, const float* w, float* loss, int32_t rows, int32_t cols,
                         void* stream);
int32_t i2v_bce_rows_bwd(const float* z, const float* t, const float* w, const float* gloss, float* gz, int32_t rows,
                         int32_t cols, void* stream);

/* Subject / object rows of the relation pairs (resnet_SGG_emb.py:170-176: two index_selects + cat):
 * out[p] = [obj[ixs[p]] | obj[ixo[p]]] (n_pairs, 2*emb) from obj (n_box, emb); an index outside [0, n_box) yields zeros.
 * Backward: gobj (n_box, emb) written in full (no clear needed), pair contributions summed in pair order (deterministic). */
int32_t i2v_pair_gather_fwd(const float* obj, const int64_t* ixs, const int64_t* ixo, float* out, int32_t n_pairs,
                            int32_t n_box, int32_t emb, void* stream);
int32_t i2v_pair_gather_bwd(const float* g, const int64_t* ixs, const int64_t* ixo, float* gobj, int32_t n_pairs,
                            int32_t n_box, int32_t emb, void* stream);

/* ---- the small arithmetic of the detector's losses and target layers, one kernel per direction (each replaces 6-25 launch-bound
 * aten kernels).  Scalar outputs are WRITTEN by one workgroup in a fixed summation order (no clear in front, no atomics).
 *   half_mse:    out = 0.5 * mean((d - target)^2): the discriminator terms of trainval_net_instance_styleD_bilinear.py:276-296
 *                (target 0: 0.5*mean(d^2); target 1: 0.5*mean((1-d)^2)); backward gd = gout * (d - target) / n.
 *   smooth_l1:   net_utils.py:122-136 (_smooth_l1_loss): sum over all n elements of outw * huber_sigma(inw * (pred - tgt)) / rows
 *                (= the mean over the batch axis of the per-image sums); inw / outw carry one weight per `per_weight` consecutive
 *                elements (4: the (B,N,1) weights of rpn.py:100-101; 1: the (R,4) weights of the RCNN box loss).  Gradient
 *                w.r.t. pred only.
 *   bbox_transform: bbox_transform.py:36-75 (bbox_transform_batch): targets (B,N,4) of gt boxes (row stride gt_stride >= 4 floats)
 *                against ex boxes (N,4) shared by the images or (B,N,4); means4 / stds4 non-NULL: (t - mean) / std
 *                (proposal_target_layer_cascade.py:104-106).  means4 / stds4 are HOST pointers.
 *   signed_sqrt: sqrt(relu(z)) - sqrt(relu(-z)) of netD_style.forward (resnet_instance_styleD_bilinear.py:137) and its backward. */
int32_t i2v_half_mse_fwd(const float* d, int64_t n, float target, float* out, void* stream);
int32_t i2v_half_mse_bwd(const float* d, int64_t n, float target, const float* gout, float* gd, void* stream);
int32_t i2v_smooth_l1_fwd(const float* pred, const float* tgt, const float* inw, const float* outw, int64_t n, int32_t per_weight,
                          int32_t rows, float sigma, float* out, void* stream);
int32_t i2v_smooth_l1_bwd(const float* pred, const float* tgt, const float* inw, const float* outw, int64_t n, int32_t per_weight,
                          int32_t rows, float sigma, const float* gout, float* gpred, void* stream);
int32_t i2v_bbox_transform(const float* ex, int32_t ex_batched, const float* gt, int32_t gt_stride, float* out, int32_t B, int32_t N,
                           const float* means4, const float* stds4, void* stream);
int32_t i2v_signed_sqrt_fwd(const float* z, float* y, int64_t n, void* stream);
int32_t i2v_signed_sqrt_bwd(const float* z, const float* g, float* gz, int64_t n, void* stream);

/* ---- netD_pixel, fused (instance-level discriminator) ---------------------------------
 * replaces netD_pixel.forward (resnet_instance_styleD_bilinear.py:38-83: GRL, conv1 1024->512 + ReLU, conv2
 * 512->128 + ReLU, conv3 128->1, sigmoid, optional context vector = mean of the 128-d features over the ROI's
 * pixels) and its autograd backward incl. GradReverse (net_utils.py:52-61) with one kernel per direction: the 512-
 * and 128-wide activations of a 32-row tile stay in LDS between the layers.
 *   x (M,1024) rows = ROI pixels (NHWC order, M = R * pix_per_roi); w1 (512,1024), w2 (128,512), w3 (128); no biases.
 *   fwd writes h1 (M,512), h2 (M,128) (post-ReLU, kept for the backward), d (M) and, if feat != NULL, feat (R,128).
 *   bwd takes gd (M) (may be NULL) and gfeat (R,128) (may be NULL) and writes g3 (M) = grad at the conv3 output,
 *   gh2 (M,128), gh1 (M,512) -- the `gy` operands of the three filter gradients, which are plain
 *   i2v_conv_wgrad calls with KH = KW = 1 -- and gx (M,1024) = -lambda * d loss / d x. */
int32_t i2v_dpixel_fwd(const float* x, const float* w1, const float* w2, const float* w3, float* h1, float* h2,
                       float* d, float* feat, int32_t M, int32_t pix_per_roi, void* stream);
size_t  i2v_dpixel_bwd_workspace_bytes(void);
int32_t i2v_dpixel_bwd(const float* gd, const float* gfeat, const float* d, const float* h1, const float* h2,
                       const float* w1, const float* w2, const float* w3, float* g3, float* gh2, float* gh1,
                       float* gx, int32_t M, int32_t pix_per_roi, float lambda, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---- fused SGD(momentum) step over a flat parameter buffer -------------------------
 * replaces torch.optim.SGD.step of trainval_net_SGG_emb.py:145-150,255 for one param
 * group: g' = g + wd*p ; m = mom*m + g' ; p -= lr*m   (m==first step handled by caller
 * zero-init, which equals torch's "buf = clone(g')" on the first step). */
int32_t i2v_sgd_momentum(float* p, const float* g, float* m, int64_t n, float lr, float momentum,
                         float weight_decay, void* stream);
/* The same update for `count` tensors in one launch (host arrays of device pointers / sizes / per-tensor lr and
 * weight decay, read at call time): for the dozens of biases and small filters of the vrd head, where the launch
 * is the cost.  Element order and rounding are those of i2v_sgd_momentum. */
int32_t i2v_sgd_momentum_multi(float* const* p, const float* const* g, float* const* m, const int64_t* n,
                               const float* lr, const float* weight_decay, int32_t count, float momentum,
                               void* stream);

/* torch.optim.Adam (amsgrad off; trainval_net_*.py `--o adam`, :143-145 / :146-147) for `count` tensors, same calling form as
 * i2v_sgd_momentum_multi: g' = g + wd p; m += (1 - b1)(g' - m); v = b2 v + (1 - b2) g'^2;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The step count t is an int32 in DEVICE memory that
 * i2v_adam_step increments (once per optimizer step, before the i2v_adam_multi launches of that step): a captured
 * training step then replays with the right bias corrections. */
int32_t i2v_adam_step(int32_t* step_counter, void* stream);
/* lr, beta1, beta2, eps are DOUBLES, as torch.optim.Adam holds them: its per-step scalars (1 - beta^t, lr / (1 - beta1^t),
 * sqrt(1 - beta2^t)) are computed in double and rounded to float once; so are they here. */
int32_t i2v_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                       const double* lr, const float* weight_decay, int32_t count, double beta1, double beta2, double eps,
                       const int32_t* step_counter, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I2VSGG_HIP_H */
