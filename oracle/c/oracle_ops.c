/*
 * oracle_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the integer/byte-level ops of the hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (i2vsgg_amd/) never links or calls it.
 *
 * Every function cites the reference file:line (under /root/reference/lib/model)
 * whose arithmetic it restates.  Arithmetic type and evaluation order follow the
 * reference's CPU path exactly (float vs double promotion, no FMA contraction:
 * build with -ffp-contract=off).
 *
 * Pinning status:
 *   nms_greedy          pinned  : golden vectors made by the reference nms_cpu.py
 *   roi_align_fwd       pinned  : bit-equal to the reference's own ROIAlignForwardCpu
 *                       (roi_align/src/roi_align.c:80-136, compiled from its own text by
 *                       oracle/build_ref.py -> tests/golden/roi_align_fwd.npz, tier "extracted").
 *   roi_align_bwd       pinned through the forward: restated from roi_align_kernel.cu:94-143
 *                       (the CPU twin roi_align.c:138-190 inverts its bounds test, :175) and held
 *                       to the transpose of the pinned forward, element by element
 *                       (tests/test_oracle_golden.py).
 *   roi_pool_*          UNPINNED: model._C source is absent from the reference tree;
 *                       restated from roi_pooling_kernel.cu:24-93,128-203.
 *   roi_align_sampled_* UNPINNED: model._C (roi_layers/roi_align.py:20,:31) is the csrc of
 *                       facebookresearch/maskrcnn-benchmark via jwyang/faster-rcnn.pytorch
 *                       (pytorch-1.0 branch), no commit pinned, source absent from the
 *                       reference tree; its published algorithm
 *                       (ROIAlign_cuda.cu: RoIAlignForward / bilinear_interpolate /
 *                       RoIAlignBackwardFeature) is restated here in fp32.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ NMS ---- */
/* nms/nms_cpu.py:6-34.  dets (n,5) fp32 [x1,y1,x2,y2,score], rows already in
 * descending score order (the caller sorts; proposal_layer.py:127-146) so the
 * reference's argsort()[::-1] is the identity on tie-free input.  Keeps i, drops
 * j>i with IoU > thresh (`ovr <= thresh` survives, nms_cpu.py:31); IoU in fp32,
 * thresh compared as fp32 (numpy weak-scalar promotion).  Returns count kept. */
int oracle_nms_sorted(const float* dets, int n, float thresh, int32_t* keep)
{
    unsigned char* dead = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
    float* area = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        const float* d = dets + 5 * (size_t)i;
        area[i] = (d[2] - d[0] + 1.0f) * (d[3] - d[1] + 1.0f);
    }
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        const float* a = dets + 5 * (size_t)i;
        for (int j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float* b = dets + 5 * (size_t)j;
            float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
            float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
            float w = fmaxf(0.0f, xx2 - xx1 + 1.0f);
            float h = fmaxf(0.0f, yy2 - yy1 + 1.0f);
            float inter = w * h;
            float ovr = inter / (area[i] + area[j] - inter);
            if (!(ovr <= thresh)) dead[j] = 1;
        }
    }
    free(dead); free(area);
    return nk;
}

/* ------------------------------------------------------------ ROIAlign ---- */
/* Shared sample-point geometry of roi_align.c:99-118 (fwd) and
 * roi_align_kernel.cu:103-125 (bwd).  Returns 0 when the sample is outside. */
static int ra_sample(const float* roi, float scale, int H, int W, int AH, int AW,
                     int ph, int pw, int* hs, int* ws, float* hr, float* wr)
{
    float x1 = roi[1] * scale, y1 = roi[2] * scale;
    float x2 = roi[3] * scale, y2 = roi[4] * scale;
    float rw = fmaxf(x2 - x1 + 1., 0.);
    float rh = fmaxf(y2 - y1 + 1., 0.);
    float bh = rh / (AH - 1.);
    float bw = rw / (AW - 1.);
    float h = (float)(ph) * bh + y1;
    float w = (float)(pw) * bw + x1;
    int hstart = fminf(floor(h), H - 2);
    int wstart = fminf(floor(w), W - 2);
    if (h < 0 || h >= H || w < 0 || w >= W) return 0;
    *hs = hstart; *ws = wstart;
    *hr = h - (float)(hstart);
    *wr = w - (float)(wstart);
    return 1;
}

/* roi_align.c:80-136 (ROIAlignForwardCpu).  feat NCHW (B,C,H,W); rois (R,5);
 * out (R,C,AH,AW).  Double-precision tap products exactly as the `1.` literals
 * promote them. */
void oracle_roi_align_fwd(const float* feat, const float* rois, int R, int C, int H,
                          int W, int AH, int AW, float scale, float* out)
{
    for (int n = 0; n < R; ++n)
        for (int ph = 0; ph < AH; ++ph)
            for (int pw = 0; pw < AW; ++pw) {
                int hs, ws; float hr, wr;
                int ok = ra_sample(rois + 5 * n, scale, H, W, AH, AW, ph, pw, &hs, &ws, &hr, &wr);
                int b = (int)rois[5 * n];
                for (int c = 0; c < C; ++c) {
                    size_t o = (((size_t)n * C + c) * AH + ph) * AW + pw;
                    if (!ok) { out[o] = 0.f; continue; }
                    const float* p = feat + (((size_t)b * C + c) * H + hs) * W + ws;
                    out[o] = p[0] * (1. - hr) * (1. - wr) + p[1] * (1. - hr) * wr
                           + p[W] * hr * (1. - wr) + p[W + 1] * hr * wr;
                }
            }
}

/* roi_align_kernel.cu:94-143 (ROIAlignBackward; the CPU twin roi_align.c:138-190
 * has an inverted bounds test and is not followed).  Serial order replaces the
 * CUDA atomics: index order (n,c,ph,pw).  grad_in NCHW, pre-zeroed by caller. */
void oracle_roi_align_bwd(const float* gout, const float* rois, int R, int C, int H,
                          int W, int AH, int AW, float scale, float* gin)
{
    for (int n = 0; n < R; ++n) {
        int b = (int)rois[5 * n];
        for (int c = 0; c < C; ++c)
            for (int ph = 0; ph < AH; ++ph)
                for (int pw = 0; pw < AW; ++pw) {
                    int hs, ws; float hr, wr;
                    if (!ra_sample(rois + 5 * n, scale, H, W, AH, AW, ph, pw, &hs, &ws, &hr, &wr))
                        continue;
                    float g = gout[(((size_t)n * C + c) * AH + ph) * AW + pw];
                    float* p = gin + (((size_t)b * C + c) * H + hs) * W + ws;
                    p[0]     += (float)(g * (1. - hr) * (1 - wr));
                    p[1]     += (float)(g * (1. - hr) * wr);
                    p[W]     += (float)(g * hr * (1 - wr));
                    p[W + 1] += (float)(g * hr * wr);
                }
    }
}

/* ------------------------------------------- ROIAlign, sampled (model._C) ---- */
/* maskrcnn-benchmark bilinear_interpolate: a sample more than one pixel outside the
 * map contributes 0; coordinates are clamped at 0 and at the last row / column.
 * Returns 0 for an outside sample, else the four tap weights and the two corners. */
static int ras_taps(int H, int W, float y, float x, int* yl, int* xl, int* yh, int* xh, float w[4])
{
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    *yl = (int)y; *xl = (int)x;
    if (*yl >= H - 1) { *yh = *yl = H - 1; y = (float)*yl; } else *yh = *yl + 1;
    if (*xl >= W - 1) { *xh = *xl = W - 1; x = (float)*xl; } else *xh = *xl + 1;
    float ly = y - (float)*yl, lx = x - (float)*xl, hy = 1.f - ly, hx = 1.f - lx;
    w[0] = hy * hx; w[1] = hy * lx; w[2] = ly * hx; w[3] = ly * lx;
    return 1;
}

struct ras_geom { float y0, x0, bh, bw; int gh, gw; };

/* RoIAlignForward prologue: no +1 on the extent, extent clamped to >= 1, bin = extent / pooled,
 * sampling grid = sampling_ratio or ceil(extent / pooled) when sampling_ratio <= 0. */
static struct ras_geom ras_geometry(const float* roi, float scale, int PH, int PW, int sampling)
{
    struct ras_geom g;
    float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
    float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    g.x0 = x1; g.y0 = y1;
    g.bh = rh / (float)PH; g.bw = rw / (float)PW;
    g.gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)PH);
    g.gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)PW);
    return g;
}

/* feat NCHW (B,C,H,W), rois (R,5) -> out (R,C,PH,PW): mean of the gh x gw bilinear samples of a bin. */
void oracle_roi_align_sampled_fwd(const float* feat, const float* rois, int R, int C, int H, int W,
                                  int PH, int PW, float scale, int sampling, float* out)
{
    for (int n = 0; n < R; ++n) {
        struct ras_geom g = ras_geometry(rois + 5 * n, scale, PH, PW, sampling);
        int b = (int)rois[5 * n];
        float count = (float)(g.gh * g.gw);
        for (int c = 0; c < C; ++c) {
            const float* f = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < g.gh; ++iy) {
                        float y = g.y0 + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
                        for (int ix = 0; ix < g.gw; ++ix) {
                            float x = g.x0 + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                            int yl, xl, yh, xh; float w[4];
                            if (!ras_taps(H, W, y, x, &yl, &xl, &yh, &xh, w)) continue;
                            acc += w[0] * f[yl * W + xl] + w[1] * f[yl * W + xh] + w[2] * f[yh * W + xl]
                                 + w[3] * f[yh * W + xh];
                        }
                    }
                    out[(((size_t)n * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
}

/* RoIAlignBackwardFeature in serial (n,c,ph,pw,iy,ix) order; gin NCHW pre-zeroed by the caller. */
void oracle_roi_align_sampled_bwd(const float* gout, const float* rois, int R, int C, int H, int W,
                                  int PH, int PW, float scale, int sampling, float* gin)
{
    for (int n = 0; n < R; ++n) {
        struct ras_geom g = ras_geometry(rois + 5 * n, scale, PH, PW, sampling);
        int b = (int)rois[5 * n];
        float count = (float)(g.gh * g.gw);
        for (int c = 0; c < C; ++c) {
            float* f = gin + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    float top = gout[(((size_t)n * C + c) * PH + ph) * PW + pw];
                    for (int iy = 0; iy < g.gh; ++iy) {
                        float y = g.y0 + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
                        for (int ix = 0; ix < g.gw; ++ix) {
                            float x = g.x0 + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                            int yl, xl, yh, xh; float w[4];
                            if (!ras_taps(H, W, y, x, &yl, &xl, &yh, &xh, w)) continue;
                            f[yl * W + xl] += top * w[0] / count;
                            f[yl * W + xh] += top * w[1] / count;
                            f[yh * W + xl] += top * w[2] / count;
                            f[yh * W + xh] += top * w[3] / count;
                        }
                    }
                }
        }
    }
}

/* roi_align/modules/roi_align.py:26-29: avg_pool2d(kernel 2, stride 1) over the
 * (PH+1)x(PW+1) aligned grid; fp32 running sum in window raster order then /4
 * (torch CPU avg_pool2d contiguous kernel). */
void oracle_avgpool2x2_fwd(const float* x, int N, int AH, int AW, float* y)
{
    int PH = AH - 1, PW = AW - 1;
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < PH; ++i)
            for (int j = 0; j < PW; ++j) {
                const float* p = x + ((size_t)n * AH + i) * AW + j;
                float s = 0.f;
                s += p[0]; s += p[1]; s += p[AW]; s += p[AW + 1];
                y[((size_t)n * PH + i) * PW + j] = s / 4.f;
            }
}

void oracle_avgpool2x2_bwd(const float* gy, int N, int AH, int AW, float* gx)
{
    int PH = AH - 1, PW = AW - 1;
    memset(gx, 0, sizeof(float) * (size_t)N * AH * AW);
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < PH; ++i)
            for (int j = 0; j < PW; ++j) {
                float d = gy[((size_t)n * PH + i) * PW + j] / 4.f;
                float* p = gx + ((size_t)n * AH + i) * AW + j;
                p[0] += d; p[1] += d; p[AW] += d; p[AW + 1] += d;
            }
}

/* ------------------------------------------------------------- ROIPool ---- */
/* roi_pooling_kernel.cu:24-93 (ROIPoolForward), same Caffe algorithm that
 * model._C.roi_pool_forward implements (roi_layers/roi_pool.py:17).  feat NCHW.
 * argmax = index inside the (b,c) plane (h*W+w) or -1. */
void oracle_roi_pool_fwd(const float* feat, const float* rois, int R, int C, int H,
                         int W, int PH, int PW, float scale, float* out, int32_t* argmax)
{
    for (int n = 0; n < R; ++n) {
        const float* r = rois + 5 * n;
        int b = (int)r[0];
        int x1 = (int)roundf(r[1] * scale), y1 = (int)roundf(r[2] * scale);
        int x2 = (int)roundf(r[3] * scale), y2 = (int)roundf(r[4] * scale);
        int rw = (int)fmaxf((float)(x2 - x1 + 1), 1.f);
        int rh = (int)fmaxf((float)(y2 - y1 + 1), 1.f);
        float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
        for (int c = 0; c < C; ++c) {
            const float* plane = feat + ((size_t)b * C + c) * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
                    int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
                    hs = (int)fminf(fmaxf((float)(hs + y1), 0.f), (float)H);
                    he = (int)fminf(fmaxf((float)(he + y1), 0.f), (float)H);
                    ws = (int)fminf(fmaxf((float)(ws + x1), 0.f), (float)W);
                    we = (int)fminf(fmaxf((float)(we + x1), 0.f), (float)W);
                    int empty = (he <= hs) || (we <= ws);
                    float m = empty ? 0.f : -FLT_MAX;
                    int mi = -1;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w)
                            if (plane[h * W + w] > m) { m = plane[h * W + w]; mi = h * W + w; }
                    size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
                    out[o] = m;
                    argmax[o] = mi;
                }
        }
    }
}

/* roi_pooling_kernel.cu:128-203 computes the same sum in gather form; here in
 * scatter form over (n,c,ph,pw) order.  gin NCHW pre-zeroed by caller. */
void oracle_roi_pool_bwd(const float* gout, const float* rois, const int32_t* argmax,
                         int R, int C, int H, int W, int PH, int PW, float* gin)
{
    for (int n = 0; n < R; ++n) {
        int b = (int)rois[5 * n];
        for (int c = 0; c < C; ++c) {
            float* plane = gin + ((size_t)b * C + c) * H * W;
            for (int k = 0; k < PH * PW; ++k) {
                size_t o = ((size_t)n * C + c) * PH * PW + k;
                if (argmax[o] >= 0) plane[argmax[o]] += gout[o];
            }
        }
    }
}

/* --------------------------------------------------------- RPN decode ---- */
/* bbox_transform.py:77-103 + :125-133 on one image.  anchors (N,4), deltas (N,4),
 * each op rounded separately in fp32; exp evaluated in double and rounded once
 * (<=1 ulp from torch's fp32 exp; see DESIGN.md "decode exp"). */
void oracle_decode_clip(const float* anchors, const float* deltas, int N,
                        float im_h, float im_w, float* out)
{
    float xmax = im_w - 1.0f, ymax = im_h - 1.0f;
    for (int i = 0; i < N; ++i) {
        const float* a = anchors + 4 * (size_t)i;
        const float* d = deltas + 4 * (size_t)i;
        float w = a[2] - a[0] + 1.0f, h = a[3] - a[1] + 1.0f;
        float cx = a[0] + 0.5f * w, cy = a[1] + 0.5f * h;
        float pcx = d[0] * w + cx, pcy = d[1] * h + cy;
        float pw = (float)exp((double)d[2]) * w, ph = (float)exp((double)d[3]) * h;
        float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph;
        float x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
        float* o = out + 4 * (size_t)i;
        o[0] = fminf(fmaxf(x1, 0.f), xmax); o[1] = fminf(fmaxf(y1, 0.f), ymax);
        o[2] = fminf(fmaxf(x2, 0.f), xmax); o[3] = fminf(fmaxf(y2, 0.f), ymax);
    }
}
