// ROIAlign (legacy aligned-grid variant, optionally fused with the 2x2/s1 average of
// RoIAlignAvg) and Caffe ROIPool for gfx950.  HBM-bound gather kernels: the channel
// axis is the lane axis (NHWC), so every tap is a coalesced 16 B/lane read and every
// gradient atomic wave-instruction covers 256 contiguous bytes.
//
// Arithmetic follows roi_align/src/roi_align.c:91-134 exactly (float / double
// promotion, one rounding per op; this file is built with -ffp-contract=off).
#include "common.h"
#include <float.h>

namespace {

struct Strides { long long b, c, h, w; };

__host__ __device__ inline Strides feat_strides(int layout, int C, int H, int W) {
    Strides s;
    if (layout == I2V_LAYOUT_NHWC) { s.c = 1; s.w = C; s.h = (long long)W * C; s.b = (long long)H * W * C; }
    else { s.w = 1; s.h = W; s.c = (long long)H * W; s.b = (long long)C * H * W; }
    return s;
}

struct Sample { int ok, hs, ws; float hr, wr; };

// roi_align.c:99-118: geometry of sample point (ph,pw) of the AHxAW aligned grid.
__device__ inline Sample ra_sample(const float* roi, float scale, int H, int W, int AH, int AW, int ph, int pw) {
    Sample s;
    float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
    float rw = fmaxf((float)((double)(x2 - x1) + 1.), 0.f);
    float rh = fmaxf((float)((double)(y2 - y1) + 1.), 0.f);
    float bh = (float)((double)rh / (AH - 1.));
    float bw = (float)((double)rw / (AW - 1.));
    float h = (float)ph * bh + y1;
    float w = (float)pw * bw + x1;
    s.hs = (int)fminf(floorf(h), (float)(H - 2));
    s.ws = (int)fminf(floorf(w), (float)(W - 2));
    s.ok = !(h < 0 || h >= H || w < 0 || w >= W);
    s.hr = h - (float)s.hs;
    s.wr = w - (float)s.ws;
    return s;
}

__device__ inline float bilinear(float p00, float p01, float p10, float p11, float hr, float wr) {
    return (float)(p00 * (1. - hr) * (1. - wr) + p01 * (1. - hr) * wr + p10 * hr * (1. - wr) + p11 * hr * wr);
}

// ---------------------------------------------------------------- forward, NHWC
// grid = R*PH blocks (one output row of one ROI), 256 threads, 4 channels per thread.
template <int AVG>
__global__ void __launch_bounds__(256)
roi_align_fwd_nhwc(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                   int C, int H, int W, int PH, int PW, float scale, Strides os) {
    const int r = blockIdx.x / PH, ph = blockIdx.x % PH;
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const int AH = PH + AVG, AW = PW + AVG;
    const float* fb = feat + (long long)b * H * W * C;
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
        float4 tp = make_float4(0, 0, 0, 0), bp = tp;   // previous column: top / bottom sample row
        for (int aw = 0; aw < AW; ++aw) {
            float4 v[2];
#pragma unroll
            for (int k = 0; k <= AVG; ++k) {
                Sample s = ra_sample(roi, scale, H, W, AH, AW, ph + k, aw);
                if (s.ok) {
                    const float* p = fb + ((long long)s.hs * W + s.ws) * C + c;
                    float4 a = *(const float4*)p, bq = *(const float4*)(p + C);
                    float4 cq = *(const float4*)(p + (long long)W * C), d = *(const float4*)(p + (long long)W * C + C);
                    v[k].x = bilinear(a.x, bq.x, cq.x, d.x, s.hr, s.wr);
                    v[k].y = bilinear(a.y, bq.y, cq.y, d.y, s.hr, s.wr);
                    v[k].z = bilinear(a.z, bq.z, cq.z, d.z, s.hr, s.wr);
                    v[k].w = bilinear(a.w, bq.w, cq.w, d.w, s.hr, s.wr);
                } else {
                    v[k] = make_float4(0, 0, 0, 0);
                }
            }
            float4 o;
            int pw;
            if (AVG) {
                if (aw == 0) { tp = v[0]; bp = v[1]; continue; }
                pw = aw - 1;
                // avg_pool2d(2, stride 1): fp32 running sum in window raster order, then /4
                o.x = (((tp.x + v[0].x) + bp.x) + v[1].x) / 4.f;
                o.y = (((tp.y + v[0].y) + bp.y) + v[1].y) / 4.f;
                o.z = (((tp.z + v[0].z) + bp.z) + v[1].z) / 4.f;
                o.w = (((tp.w + v[0].w) + bp.w) + v[1].w) / 4.f;
                tp = v[0]; bp = v[1];
            } else {
                pw = aw;
                o = v[0];
            }
            float* q = out + r * os.b + ph * os.h + pw * os.w + c * os.c;
            if (os.c == 1) {
                *(float4*)q = o;
            } else {
                q[0] = o.x; q[os.c] = o.y; q[2 * os.c] = o.z; q[3 * os.c] = o.w;
            }
        }
    }
}

// ---------------------------------------------------------------- forward, NHWC, column-pair blocks
// The row kernel above walks the AW sample columns of its output row one after the other: 8 dependent HBM round trips
// per workgroup and only R*PH workgroups (224 at 32 ROIs: fewer than the chip has CUs) -- 18.7 us for 16 MB, 0.11 of
// the HBM roofline (profiles/r01_roi_nms_bench.txt).  Here a workgroup owns TWO output columns of one output row: its
// (2 x 3) samples x 4 taps = 24 16-byte loads per lane are all issued before the first one is used (ONE round trip),
// and the grid is R*PH*ceil(PW/2) workgroups (896 at 32 ROIs).  Out-of-map samples load tap (0,0) and are zeroed by a
// select, so no load sits under a condition.  Same arithmetic, same order: bit-equal to the row kernel and the oracle.
template <int AVG>
__global__ void __launch_bounds__(256)
roi_align_fwd_nhwc_cols(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                        int R, int C, int H, int W, int PH, int PW, float scale, Strides os) {
    const int groups = (PW + 1) >> 1;
    // XCD-aware placement: workgroups b and b+8 share an XCD (and its L2), so ALL workgroups of one ROI get the same
    // b % 8 -- the ROI's window of the map then crosses the fabric once instead of once per XCD (PMC, 4 frames x 32 ROIs:
    // 173 MB of HBM traffic for 65 MB of algorithmic bytes with the plain order).  Placement only; R is padded to a
    // multiple of 8 by the launcher and the surplus workgroups exit.
    const int per_roi = PH * groups;
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int r = x + 8 * (i / per_roi), inner = i % per_roi;
    if (r >= R) return;
    const int g = inner % groups, ph = inner / groups, pw0 = 2 * g;
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const int AH = PH + AVG, AW = PW + AVG;
    const float* fb = feat + (long long)b * H * W * C;
    constexpr int NR = 1 + AVG, NC = 2 + AVG;          // sample rows / columns this workgroup needs
    Sample sm[NR][NC];
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            sm[k][j] = ra_sample(roi, scale, H, W, AH, AW, ph + k, min(pw0 + j, AW - 1));
            if (pw0 + j >= AW) sm[k][j].ok = 0;
            if (!sm[k][j].ok) { sm[k][j].hs = 0; sm[k][j].ws = 0; }
        }
    for (int c = threadIdx.x * 4; c < C; c += blockDim.x * 4) {
        float4 t[NR][NC][4];
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float* p = fb + ((long long)sm[k][j].hs * W + sm[k][j].ws) * C + c;
                t[k][j][0] = *(const float4*)p;
                t[k][j][1] = *(const float4*)(p + C);
                t[k][j][2] = *(const float4*)(p + (long long)W * C);
                t[k][j][3] = *(const float4*)(p + (long long)W * C + C);
            }
        float4 v[NR][NC];
#pragma unroll
        for (int k = 0; k < NR; ++k)
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const Sample s = sm[k][j];
                float4 q;
                q.x = bilinear(t[k][j][0].x, t[k][j][1].x, t[k][j][2].x, t[k][j][3].x, s.hr, s.wr);
                q.y = bilinear(t[k][j][0].y, t[k][j][1].y, t[k][j][2].y, t[k][j][3].y, s.hr, s.wr);
                q.z = bilinear(t[k][j][0].z, t[k][j][1].z, t[k][j][2].z, t[k][j][3].z, s.hr, s.wr);
                q.w = bilinear(t[k][j][0].w, t[k][j][1].w, t[k][j][2].w, t[k][j][3].w, s.hr, s.wr);
                v[k][j] = s.ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pw = pw0 + j;
            if (pw >= PW) break;
            float4 o;
            if (AVG) {      // avg_pool2d(2, stride 1): fp32 running sum in window raster order, then /4
                o.x = (((v[0][j].x + v[0][j + AVG].x) + v[AVG][j].x) + v[AVG][j + AVG].x) / 4.f;
                o.y = (((v[0][j].y + v[0][j + AVG].y) + v[AVG][j].y) + v[AVG][j + AVG].y) / 4.f;
                o.z = (((v[0][j].z + v[0][j + AVG].z) + v[AVG][j].z) + v[AVG][j + AVG].z) / 4.f;
                o.w = (((v[0][j].w + v[0][j + AVG].w) + v[AVG][j].w) + v[AVG][j + AVG].w) / 4.f;
            } else {
                o = v[0][j];
            }
            float* q = out + r * os.b + ph * os.h + pw * os.w + c * os.c;
            if (os.c == 1) {
                *(float4*)q = o;
            } else {
                q[0] = o.x; q[os.c] = o.y; q[2 * os.c] = o.z; q[3 * os.c] = o.w;
            }
        }
    }
}

// ---------------------------------------------------------------- forward, NHWC in and out, one ROI x 128 channels
// The column-pair kernel above issues (2 x 3) samples x 4 taps per TWO outputs: 588 tap loads per ROI and channel for the
// 256 the ROI's 8 x 8 sample points need (every sample feeds up to four outputs and is fetched again by each of their
// workgroups) -- 86 MB through the L2s for one frame's 16 MB of algorithmic bytes, and the kernel runs at the L2s' rate.
// Here a workgroup owns ONE ROI x 128 channels: each 32-lane group fetches whole samples (4 taps x 16 bytes per lane, 16 loads
// in flight per lane and pass), the AH x AW sample values meet in LDS (32 KB) and every output is the 2 x 2 mean of four of
// them -- each tap crosses the L2 once.  blockIdx % (C / 128) picks the channel chunk: with C = 1024 an XCD (blockIdx % 8)
// serves ONE 128-channel slice of the maps for every ROI, 1.2 MB per frame, resident in its L2.
// Same sample geometry (ra_sample), same fp64 tap products (bilinear), same order of the 2 x 2 mean: bit-equal to the kernels above.
// CH = 64 (launches of fewer than 512 workgroups at 128: one frame's 32 ROIs): twice the workgroups, and the sixteen 16-lane groups
// fetch all 64 samples in ONE pass instead of two.
template <int AVG, int CH = 128>
__global__ void __launch_bounds__(256)
roi_align_fwd_roi_kernel(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                         int R, int C, int H, int W, int PH, int PW, float scale) {
    constexpr int L = CH / 4, G = 256 / L, LG = CH == 128 ? 7 : 6;     // lanes per sample, sample groups, log2(CH)
    __shared__ __attribute__((aligned(16))) float4 sv[64 * L];         // [sample][lane]: CH channels of up to 64 samples
    const int nchunk = C >> LG;
    const int chunk = blockIdx.x % nchunk, r = blockIdx.x / nchunk;
    if (r >= R) return;
    const int l = threadIdx.x & (L - 1), g = threadIdx.x / L;
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const int AH = PH + AVG, AW = PW + AVG, NS = AH * AW;
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, 0x7FFFFFFCu, 0x00020000);
    const unsigned row = (unsigned)C * 4u, lane_off = (unsigned)((chunk << LG) + 4 * l) * 4u;
    const unsigned img = (unsigned)b * (unsigned)(H * W);
    for (int s0 = g; s0 < NS; s0 += 4 * G) {             // four samples (16 loads) per lane and pass
        Sample sm[4];
        float4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + G * j;
            const int ph = min(s, NS - 1) / AW, pw = min(s, NS - 1) % AW;
            sm[j] = ra_sample(roi, scale, H, W, AH, AW, ph, pw);
            if (s >= NS) sm[j].ok = 0;
            const unsigned dead = sm[j].ok ? 0u : 0x80000000u;      // outside the map: zeros without traffic
            const unsigned at = (img + (unsigned)(sm[j].hs * W + sm[j].ws)) * row + lane_off;
            t[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, at | dead, 0, 0));
            t[j][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, (at + row) | dead, 0, 0));
            t[j][2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, (at + (unsigned)W * row) | dead, 0, 0));
            t[j][3] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(fr, (at + (unsigned)W * row + row) | dead, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + G * j;
            if (s >= NS) continue;
            float4 q;
            q.x = bilinear(t[j][0].x, t[j][1].x, t[j][2].x, t[j][3].x, sm[j].hr, sm[j].wr);
            q.y = bilinear(t[j][0].y, t[j][1].y, t[j][2].y, t[j][3].y, sm[j].hr, sm[j].wr);
            q.z = bilinear(t[j][0].z, t[j][1].z, t[j][2].z, t[j][3].z, sm[j].hr, sm[j].wr);
            q.w = bilinear(t[j][0].w, t[j][1].w, t[j][2].w, t[j][3].w, sm[j].hr, sm[j].wr);
            sv[s * L + l] = sm[j].ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    float* ob = out + (long long)r * PH * PW * C + (chunk << LG);
    for (int o = threadIdx.x; o < PH * PW * L; o += 256) {
        const int p = o / L, ll = o & (L - 1);
        const int ph = p / PW, pw = p - ph * PW;
        float4 v;
        if (AVG) {          // avg_pool2d(2, stride 1): fp32 running sum in window raster order, then /4
            const float4 a = sv[(ph * AW + pw) * L + ll], bb = sv[(ph * AW + pw + 1) * L + ll];
            const float4 c = sv[((ph + 1) * AW + pw) * L + ll], d = sv[((ph + 1) * AW + pw + 1) * L + ll];
            v.x = (((a.x + bb.x) + c.x) + d.x) / 4.f;
            v.y = (((a.y + bb.y) + c.y) + d.y) / 4.f;
            v.z = (((a.z + bb.z) + c.z) + d.z) / 4.f;
            v.w = (((a.w + bb.w) + c.w) + d.w) / 4.f;
        } else {
            v = sv[p * L + ll];
        }
        *(float4*)(ob + (long long)p * C + 4 * ll) = v;
    }
}

// ---------------------------------------------------------------- forward, any layout
// one thread per output element; the reference's own decomposition
// (roi_align_kernel.cu:15-70).  Used for NCHW features (inner-boundary drop-in).
template <int AVG>
__global__ void roi_align_fwd_generic(const float* __restrict__ feat, const float* __restrict__ rois,
                                      float* __restrict__ out, long long total, int C, int H, int W, int PH,
                                      int PW, float scale, Strides fs, Strides os) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int pw = idx % PW, ph = (idx / PW) % PH, c = (idx / PW / PH) % C, r = idx / PW / PH / C;
        const float* roi = rois + 5 * (long long)r;
        const float* fb = feat + (long long)roi[0] * fs.b + c * fs.c;
        const int AH = PH + AVG, AW = PW + AVG;
        float v[2][2];
#pragma unroll
        for (int i = 0; i <= AVG; ++i)
#pragma unroll
            for (int j = 0; j <= AVG; ++j) {
                Sample s = ra_sample(roi, scale, H, W, AH, AW, ph + i, pw + j);
                float val = 0.f;
                if (s.ok) {
                    const float* p = fb + s.hs * fs.h + s.ws * fs.w;
                    val = bilinear(p[0], p[fs.w], p[fs.h], p[fs.h + fs.w], s.hr, s.wr);
                }
                v[i][j] = val;
            }
        float o = AVG ? (((v[0][0] + v[0][AVG]) + v[AVG][0]) + v[AVG][AVG]) / 4.f : v[0][0];
        out[r * os.b + c * os.c + ph * os.h + pw * os.w] = o;
    }
}

// ---------------------------------------------------------------- backward (atomics)
// grid = R*AH blocks (one SAMPLE row of one ROI), lane = channel.  Follows
// roi_align_kernel.cu:94-143 (the CPU twin roi_align.c:175 has an inverted test); the
// avg_pool2d backward (grad/4 summed over the <=4 windows holding the sample, raster
// order) is folded in front of the scatter.
template <int AVG>
__global__ void __launch_bounds__(256)
roi_align_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat,
                     int C, int H, int W, int PH, int PW, float scale, Strides fs, Strides os) {
    const int AH = PH + AVG, AW = PW + AVG;
    const int r = blockIdx.x / AH, ah = blockIdx.x % AH;
    const float* roi = rois + 5 * (long long)r;
    float* gb = gfeat + (long long)roi[0] * fs.b;
    for (int aw = 0; aw < AW; ++aw) {
        Sample s = ra_sample(roi, scale, H, W, AH, AW, ah, aw);
        if (!s.ok) continue;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float* go = gout + r * os.b + c * os.c;
            float g;
            if (AVG) {
                g = 0.f;
                for (int ph = ah - 1; ph <= ah; ++ph)
                    for (int pw = aw - 1; pw <= aw; ++pw)
                        if (ph >= 0 && ph < PH && pw >= 0 && pw < PW) g += go[ph * os.h + pw * os.w] / 4.f;
            } else {
                g = go[ah * os.h + aw * os.w];
            }
            float* p = gb + c * fs.c + s.hs * fs.h + s.ws * fs.w;
            atomicAdd(p, (float)(g * (1. - s.hr) * (1 - s.wr)));
            atomicAdd(p + fs.w, (float)(g * (1. - s.hr) * s.wr));
            atomicAdd(p + fs.h, (float)(g * s.hr * (1 - s.wr)));
            atomicAdd(p + fs.h + fs.w, (float)(g * s.hr * s.wr));
        }
    }
}

// ---------------------------------------------------------------- backward as a GATHER (NHWC in and out)
// The scatter above costs four float atomics per sample and channel: 134 MB of atomically added bytes for 4 frames x 32 ROIs,
// and the chip adds ~1.3 TB/s of them -- 102 us, 0.07 of the HBM roofline on algorithmic bytes, summation order free.
// Here every element of the gradient map is written ONCE, by the workgroup that owns it, as the sum of its contributions in
// the reference's serial order (roi, sample row, sample column ascending -- roi_align_kernel.cu:94-143 run as a loop):
// deterministic, no atomics, no zero-fill in front.  ONE kernel (round 5; round 4 ran a second kernel in front that wrote
// the sample gradients and the sample geometry to a workspace -- 2.0x the algorithmic bytes): per (frame, map row h,
// 128 channels)
//   list    the (roi, sample row) pairs of the frame whose taps touch row h (hs == h: the upper taps; hs + 1 == h: the lower
//           taps), in index order, by a deterministic ballot compaction; the row geometry is computed from the roi table on
//           the spot (a thread looks at the frame index first: only the frame's own pairs cost the fp64 arithmetic);
//   stage   per batch of listed pairs: the 128-channel sample gradients straight from grad_out -- the 2x2 mean's backward
//           (avg_pool2d: grad / 4 summed over the <= 4 windows holding the sample, raster order) folded into the load -- and
//           one tap record per (pair, sample column, cell parity): the cell of the row buffer the tap goes to and its weight;
//   add     wave w adds them for ITS 32 channels into the row buffer in list order; the row is then stored with 16-byte stores.
// A tap's weight is (1 - hr | hr) * (1 - wr | wr) rounded to fp32 once per (pair, column) instead of the scatter's
// double-promoted product per element (roi_align_kernel.cu:137-140): a tap differs from the scatter's by <= 1 ulp.
struct AxisGeom { int ok, s; float r; int pad; };        // one axis of ra_sample: in range?, floor (clamped to n - 2), fraction

// ra_sample's arithmetic for one axis (roi_align.c:99-118): the sample grid is separable, a sample row shares hs / hr, a
// sample column ws / wr, and a sample is inside the map when both are
__device__ inline AxisGeom ra_axis(float lo, float hi, float scale, int n, int A, int p) {
    const float a1 = lo * scale, a2 = hi * scale;
    const float len = fmaxf((float)((double)(a2 - a1) + 1.), 0.f);
    const float bin = (float)((double)len / (A - 1.));
    const float v = (float)p * bin + a1;
    AxisGeom g;
    g.s = (int)fminf(floorf(v), (float)(n - 2));
    g.ok = !(v < 0 || v >= n);
    g.r = v - (float)g.s;
    g.pad = 0;
    return g;
}

constexpr int RAB_SEG = 1024;           // (roi, sample row) pairs examined per list pass: 256 per wave
constexpr int RAB_BATCH = 4;            // listed pairs whose sample gradients are staged in LDS together (one per wave's record lanes)
constexpr int RAB_MAXA = 8;             // sample columns per row this kernel takes

struct TapRec { int cell0; float w0; int cell1; float w1; };    // a sample column's tap on the even / odd cell: row-buffer cell, weight

#ifdef RAB_CLOCKS           // tools/micro/rab_clock.hip only: per-workgroup phase stamps (s_memtime), never in the library
__device__ unsigned long long g_rab_clk[8192][6];
#define RAB_T(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#define RAB_ADD(acc, a, b) acc += (b) - (a)
#else
#define RAB_T(x)
#define RAB_ADD(acc, a, b)
#endif

// lane = (channel, cell parity) -- of a sample column's two taps (cells ws, ws + 1) one is even and one is odd, so every lane
// has exactly one tap per column and there is not a branch in the add loop; the cells a lane meets along a pair are
// non-decreasing, a repeated cell (ws + 1 == ws') is forwarded in a register, and the current values of all eight are requested
// at once.  No two waves touch one channel, so there is nothing to synchronise but the batches.
// P7: pooled grid 7 x 7 (the path's own: RoIAlignAvg(7, 7) -> 8 x 8 samples) with the grid sizes as constants.
template <int AVG, int P7>
__global__ void __launch_bounds__(256, 3)           // three workgroups per CU (LDS allows three): <= 168 VGPRs
roi_align_bwd_row_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int R, int PH_,
                         int PW_, float scale, int B, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rowbuf = lds;                                                    // [W + 2][128]: cell x, channel c at x * 128 + (c ^ ((x & 1) << 5))
                                                                            // (odd cells swap their 32-channel halves: the two half-waves
                                                                            // of a tap hit different banks); rows W, W + 1: a bin for taps outside the map
    float* stage = rowbuf + (size_t)(W + 2) * 128;                          // [RAB_BATCH][RAB_MAXA][128]
    unsigned short* list = (unsigned short*)(stage + RAB_BATCH * RAB_MAXA * 128);   // [4][256]: wave w's hits among ITS 256 candidates,
                                                                            // (index in the segment) * 2 + (hs + 1 == h); the list = the four in wave order
    __shared__ int s_cnt[2][4];          // by segment parity: a segment without a hit has ONE barrier, so a fast wave may write the
                                         // next segment's count while a slow one still reads this segment's (round-5 advice)
    __shared__ TapRec s_rec[RAB_BATCH][RAB_MAXA];
    const int PH = P7 ? 7 : PH_, PW = P7 ? 7 : PW_;
    const int AH = PH + AVG, AW = PW + AVG;
    const int n_pairs = R * AH;
    const int nchunk = C >> 7;
    // rows are dealt from the middle of the map outwards, all frames of a row together: boxes crowd the middle, so the rows
    // with the longest lists start first and the launch ends on short ones
    const int chunk = blockIdx.x % nchunk, b = (blockIdx.x / nchunk) % B, hk = blockIdx.x / (nchunk * B);
    const int h = (hk & 1) ? H / 2 - 1 - (hk >> 1) : H / 2 + (hk >> 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = 32 * wave + (lane & 31), par = lane >> 5, chs = ch ^ (par << 5);
#ifdef RAB_CLOCKS
    unsigned long long c_list = 0, c_stage = 0, c_add = 0, c_req = 0, c_n = 0;
#endif
    RAB_T(c_begin);
    for (int i = threadIdx.x; i < (W + 2) * 32; i += 256) ((float4*)rowbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < n_pairs; base += RAB_SEG) {
        const int sp = (base / RAB_SEG) & 1;
        RAB_T(c_l0);
        // ---- the list of this segment.  Wave w looks at candidates [256 w, 256 w + 256) in four rounds and appends its hits to
        // its own quarter of the list (a ballot and a running count: no barrier); a pair of another frame is dismissed on its
        // frame index before any arithmetic.  One barrier, then every thread knows the four counts.  No load sits under a
        // condition (a candidate past the end re-reads the last roi and is masked).
        {
            int cnt = 0;
            float f0[4], y1[4], y2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(base + 256 * wave + 64 * k + lane, n_pairs - 1);
                const float* roi = rois + 5 * (long long)(i / AH);
                f0[k] = roi[0]; y1[k] = roi[2]; y2[k] = roi[4];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int il = 256 * wave + 64 * k + lane, i = base + il;
                int hit = 0, dy = 0;
                if (i < n_pairs && (int)f0[k] == b) {
                    const AxisGeom q = ra_axis(y1[k], y2[k], scale, H, AH, i % AH);
                    dy = q.s + 1 == h;
                    hit = q.ok && (q.s == h || dy);
                }
                const unsigned long long m = __ballot(hit);
                if (hit) list[256 * wave + cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(2 * il + dy);
                cnt += __popcll(m);
            }
            if (lane == 0) s_cnt[sp][wave] = cnt;
        }
        __syncthreads();
        const int p1 = s_cnt[sp][0], p2 = p1 + s_cnt[sp][1], p3 = p2 + s_cnt[sp][2], n = p3 + s_cnt[sp][3];
        auto entry = [&](int e) {           // the e-th listed pair: global pair index * 2 + dy
            const int w = (e >= p1) + (e >= p2) + (e >= p3);
            const int off = w == 0 ? 0 : (w == 1 ? p1 : (w == 2 ? p2 : p3));
            return 2 * base + (int)list[256 * w + e - off];
        };
        RAB_T(c_l1);
        RAB_ADD(c_list, c_l0, c_l1);
#ifdef RAB_CLOCKS
        c_n += n;
#endif
        // the batch's operands travel while the PREVIOUS batch is added: the grad_out rows its samples are made of (16 bytes per
        // lane; with the 2x2 mean four loads per sample) and, on eight lanes of every wave, the roi of one pair -- requested here
        // through a buffer resource, none under a condition (a sample / window cell that does not exist gets the out-of-range
        // bit: zeros, no traffic), first used when the batch is staged.  The tap records are made from the roi AFTER the previous
        // batch's chain, when the loads have long landed.
        constexpr int TL = RAB_BATCH * RAB_MAXA * 32 / 256;
        constexpr int NQ = AVG ? 4 : 1;
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, 0x7FFFFFFCu, 0x00020000);
        float4 q4[TL][NQ];
        float rq[4];
        int rcode = 0;
        TapRec recn;
        auto request = [&](int e0) {
            const int nb = min(RAB_BATCH, n - e0);
#pragma unroll
            for (int j = 0; j < TL; ++j) {
                const int k = threadIdx.x + 256 * j;
                const int u = k / (RAB_MAXA * 32), aw = (k / 32) % RAB_MAXA, ll = k & 31;
                const int live = u < nb && aw < AW;
                const int pair = entry(e0 + min(u, nb - 1)) >> 1;
                const int r = pair / AH, ah = pair - r * AH;
                const unsigned at = (unsigned)(r * PH * PW) * (unsigned)C + (unsigned)((chunk << 7) + 4 * ll);
                if (AVG) {              // the (<= 4) pooled cells whose 2x2 window holds the sample, raster order
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int ph = ah - 1 + (d >> 1), pw = aw - 1 + (d & 1);
                        const unsigned dead = (live && ph >= 0 && ph < PH && pw >= 0 && pw < PW) ? 0u : 0x80000000u;
                        q4[j][d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                            gr, ((at + (unsigned)((ph * PW + pw) * C)) * 4u) | dead, 0, 0));
                    }
                } else {
                    q4[j][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        gr, ((at + (unsigned)((ah * PW + aw) * C)) * 4u) | (live ? 0u : 0x80000000u), 0, 0));
                }
            }
            rcode = entry(e0 + min(wave, nb - 1));
            const float* roi = rois + 5 * (long long)((rcode >> 1) / AH);
            rq[0] = roi[1]; rq[1] = roi[2]; rq[2] = roi[3]; rq[3] = roi[4];
            if (wave >= nb) rcode = -1;
        };
        auto make_rec = [&]() {
            recn.cell0 = W; recn.cell1 = W + 1; recn.w0 = 0.f; recn.w1 = 0.f;
            if (lane < RAB_MAXA && lane < AW && rcode >= 0) {
                const AxisGeom qr = ra_axis(rq[1], rq[3], scale, H, AH, (rcode >> 1) % AH);
                const AxisGeom qc = ra_axis(rq[0], rq[2], scale, W, AW, lane);
                if (qc.ok) {
                    const float fh = (rcode & 1) ? qr.r : (1 - qr.r);
                    const float wl = fh * (1 - qc.r), wr = fh * qc.r;     // taps on cells ws, ws + 1
                    const int odd = qc.s & 1;                            // the even cell is ws (odd == 0) or ws + 1
                    recn.cell0 = qc.s + odd;      recn.w0 = odd ? wr : wl;
                    recn.cell1 = qc.s + 1 - odd;  recn.w1 = odd ? wl : wr;
                }
            }
        };
        if (n > 0) { request(0); make_rec(); }
        for (int e0 = 0; e0 < n; e0 += RAB_BATCH) {
            const int nb = min(RAB_BATCH, n - e0);
            RAB_T(c_s0);
#pragma unroll
            for (int j = 0; j < TL; ++j) {
                float4 v = q4[j][0];
                if (AVG) {              // avg_pool2d backward: grad / 4 summed over the windows, raster order
                    v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        v.x += q4[j][d].x / 4.f; v.y += q4[j][d].y / 4.f; v.z += q4[j][d].z / 4.f; v.w += q4[j][d].w / 4.f;
                    }
                }
                ((float4*)stage)[threadIdx.x + 256 * j] = v;
            }
            if (lane < RAB_MAXA) s_rec[wave][lane] = recn;
            __syncthreads();
            RAB_T(c_s1);
            if (e0 + RAB_BATCH < n) request(e0 + RAB_BATCH);
            RAB_T(c_s2);
            // ---- add: pairs in list order, sample columns ascending; a pair's eight tap records, staged gradients and current
            // cell values come into registers with one LDS wait each, a tap is then a multiply-add and one store.  The next
            // pair's records and gradients are read under this pair's chain.
            float sv[RAB_MAXA], wt[RAB_MAXA];
            int at[RAB_MAXA];
            auto fetch = [&](int u) {
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) {
                    const int2 q = *(const int2*)((const char*)&s_rec[u][aw] + 8 * par);
                    at[aw] = q.x * 128 + chs;
                    wt[aw] = __int_as_float(q.y);
                    sv[aw] = stage[(u * RAB_MAXA + aw) * 128 + ch];
                }
            };
            fetch(0);
            // (round 6: the read-add-write below as ONE ds_add_f32 per tap -- a lane's LDS operations execute in order, so the
            // sums would keep their order -- measured 4.5-6x SLOWER: 105 / 275 us against 23.5 / 46.9 at 1 / 4 frames; the LDS
            // float atomic costs far more than the three instructions it replaces)
            for (int u = 0; u < nb; ++u) {
                float pre[RAB_MAXA], tap[RAB_MAXA];
                int a[RAB_MAXA];
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) { a[aw] = at[aw]; pre[aw] = rowbuf[a[aw]]; tap[aw] = sv[aw] * wt[aw]; }
                if (u + 1 < nb) fetch(u + 1);
                float cur = 0.f;
                int prev = -1;
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) {
                    const float v = (a[aw] == prev) ? cur : pre[aw];
                    cur = v + tap[aw];
                    rowbuf[a[aw]] = cur;
                    prev = a[aw];
                }
            }
            if (e0 + RAB_BATCH < n) make_rec();
            __syncthreads();
            RAB_T(c_s3);
            RAB_ADD(c_stage, c_s0, c_s1); RAB_ADD(c_req, c_s1, c_s2); RAB_ADD(c_add, c_s2, c_s3);
        }
    }
#ifdef RAB_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned long long* o6 = g_rab_clk[blockIdx.x];
        o6[0] = __builtin_amdgcn_s_memtime() - c_begin; o6[1] = c_list; o6[2] = c_stage; o6[3] = c_req; o6[4] = c_add; o6[5] = c_n;
    }
#endif
    float* o = gfeat + (((long long)b * H + h) * W) * C + (chunk << 7);
    for (int i = threadIdx.x; i < W * 32; i += 256) {
        const int x = i >> 5, c4 = i & 31;
        *(float4*)(o + (long long)x * C + 4 * c4) = ((const float4*)rowbuf)[x * 32 + (c4 ^ ((x & 1) << 3))];
    }
}

// ---------------------------------------------------------------- backward as a gather, round 6: waves that never meet
// Same ownership (a workgroup = one frame, one map row, 128 channels; every gradient element written once, no atomics) and the
// same list phase as roi_align_bwd_row_kernel.  What changed is the add.  Round 5 gave a lane ONE channel and one cell parity:
// 32 LDS instructions per (pair, wave) for 16 taps x 32 channels, four-byte accesses, the staged gradients and tap records
// passed between waves through LDS with two barriers per batch of four pairs (~1000 cycles per pair).  Here
//   * a wave owns 32 channels of the row buffer for the whole launch and a lane is (sample column, 4 channels): a pair's two
//     taps per column are two 16-byte read-add-writes, 1 + 4 LDS instructions per (pair, wave); nothing is handed between waves
//     but the tap records, made once per chunk of 32 listed pairs (one record per thread, one barrier);
//   * the sample gradients go from grad_out to registers, two pairs in flight per wave (deeper is slower: more registers and
//     dead requests behind the end of a 13-pair list buy nothing).  With the 2x2 mean a sample is the
//     sum of a 2x2 window of pooled cells, and the windows of neighbouring sample columns share a pooled column: a lane loads
//     ITS pooled column only (two 16-byte loads: the window's rows) and takes the left column's row sum from the lane two to
//     its left (DPP row_shr:2 -- the 16 lanes of a DPP row hold all eight columns of two channel groups).  Loading all four
//     window cells per lane asked the CU's vector L1 for 4 KB per (pair, wave), 1 MB per CU and launch: 64 B per clock made
//     that ~70 % of the add phase (tools/micro/rab_clock.hip);
//   * everything of a pair that is wave-uniform (roi, sample row, the byte offsets of its pooled rows) is scalar arithmetic on a
//     list entry read one step ahead and rides in the scalar offset of the loads; the per-lane part is a loop constant;
//   * sample columns of a narrow box may share a cell: a record carries the ROUND its column is added in (columns of one round
//     have distinct cells; one round unless a bin is narrower than a cell).  The sums keep a fixed order: pair, round, left tap
//     before right tap (a column's right cell is often the next column's left cell: the left taps of a round are all stored
//     before its right taps are read).  Two refinements measured no gain and are not in the kernel (tools/micro/rab_clock.hip,
//     same box): handing such a right tap to the neighbour's lane by DPP so that a pair is ONE LDS round trip (1 % slower), and
//     summing the pairs of one box in registers before a single read-add-write (30.8 against 30.8 us).  What a pair costs
//     (~1200 cycles per wave, sixteen waves per CU) is the bytes it brings into registers -- 2 KB of gradients through the
//     vector L1 and 3 KB of records and cells from the LDS, ~64 bytes per clock and CU between them -- not a dependent chain.
// A sample gradient is now (g(ah-1, aw-1) + g(ah, aw-1)) + (g(ah-1, aw) + g(ah, aw)) with the mean's 1/4 (exact) folded into
// the tap weights -- columns first, where the scatter and round 5's gather add in raster order: <= 1 ulp of the sample.
constexpr int RB2_CHUNK = 32;           // listed pairs whose tap records are made together (8 threads per pair)
#if defined(RAB_CLOCKS) && defined(RAB_DEPTH)          // tools/micro/rab_clock.hip only
constexpr int RB2_DEPTH = RAB_DEPTH;
#else
constexpr int RB2_DEPTH = 2;            // pairs in flight per wave (same box, 4 x 32: depth 1 35.0 us, 2 32.9, 4 34.2, 8 36.4)
#endif
struct TapRec2 { unsigned off; float wl, wr; int meta; };   // row-buffer byte offsets of the two taps' cells (left | right << 16; slot 0's),
                                                             // their weights, round | rounds of the pair << 8

__device__ inline float4 rab_row_shr2(float4 v) {           // lane i <- lane i - 2 of its 16-lane DPP row (zeros shifted in)
    float4 o;
    o.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0x112, 0xF, 0xF, true));
    o.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0x112, 0xF, 0xF, true));
    o.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.z), 0x112, 0xF, 0xF, true));
    o.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0x112, 0xF, 0xF, true));
    return o;
}

template <int AVG, int P7, int GROUPS>
__global__ void __launch_bounds__(256, 4)
roi_align_bwd_row2_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat, int R, int PH_,
                          int PW_, float scale, int B, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // GROUPS = 1: the workgroup owns 128 channels, a wave 32 of them, every wave walks the whole list.
    // GROUPS = 2: the workgroup owns 64 channels in TWO row buffers; the listed pairs are dealt to two groups of two waves (even /
    // odd list index), each adding into ITS buffer, and the store adds the two (buffer 0 + buffer 1: a fixed order).  A
    // workgroup's pairs form one dependent chain per wave (~1000 cycles per pair, 40+ pairs on a crowded row) and a launch that
    // fits the chip at once waits for its longest chain: two groups halve it (one frame: 15.8 -> 13.5 us).  A launch of several
    // rounds is bound by the sum instead, and twice the workgroups pay the list and the records twice (four frames: 30.1 -> 32.8
    // us): the host picks by the launch's size.
    // [GROUPS][W + 2][128 / GROUPS channels]: cell x, 4-channel group j (wave-in-group j >> 3) in slot j ^ ((x & 7) << 1) -- the
    // cells of a 16-lane access spread over the banks; rows W, W + 1: a bin for taps outside the map
    constexpr int CHB = 512 / GROUPS, WPG = 4 / GROUPS;            // bytes of a cell in one buffer; waves per group
    float* rowbuf = lds;
    unsigned short* list = (unsigned short*)(rowbuf + (size_t)(W + 2) * 128);       // [4][256] per-wave sublists, then compacted to [n]
    __shared__ int s_cnt[2][4];
    __shared__ __attribute__((aligned(16))) TapRec2 s_rec[RB2_CHUNK][8];
    const int PH = P7 ? 7 : PH_, PW = P7 ? 7 : PW_;
    const int AH = PH + AVG, AW = PW + AVG;
    const int n_pairs = R * AH;
    const int nchunk = C / (128 / GROUPS);
    const int chunk = blockIdx.x % nchunk, b = (blockIdx.x / nchunk) % B, hk = blockIdx.x / (nchunk * B);
    const int h = (hk & 1) ? H / 2 - 1 - (hk >> 1) : H / 2 + (hk >> 1);    // rows from the middle of the map outwards (longest lists first)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sxl = (lane >> 1) & 7, c4g = ((lane >> 4) << 1) | (lane & 1);        // my sample column; my 4 of the wave's 32 channels
    const int grp = wave / WPG, wch = wave % WPG;                                   // my group (list index mod GROUPS); my 32 of the buffer's channels
    constexpr int NQ = AVG ? 2 : 1;
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, 0x7FFFFFFCu, 0x00020000);
    // loop constants of a lane: the byte offset of my channels in my pooled column (the out-of-range bit where it does not exist),
    // the XOR that turns a record's cell offset into my slot of the cell
    const unsigned lcol = (sxl < AW && sxl < PW) ? ((unsigned)(sxl * C) + (unsigned)(chunk * (128 / GROUPS) + (wch << 5) + (c4g << 2))) * 4u : 0x80000000u;
    const unsigned lxor = (unsigned)(((wch << 3) | c4g) << 4);
    char* const mybuf = (char*)rowbuf + (size_t)grp * (W + 2) * CHB;
#ifdef RAB_CLOCKS
    unsigned long long c_list = 0, c_rec = 0, c_add = 0, c_n = 0;
#endif
    RAB_T(c_begin);
    for (int base = 0; base < n_pairs; base += RAB_SEG) {
        const int sp = (base / RAB_SEG) & 1;
        RAB_T(c_l0);
        int cnt = 0;
        {       // ---- the list of this segment (roi_align_bwd_row_kernel's: per-wave sublists, a ballot and a running count, ONE barrier).
                // The candidates are dealt to the waves in blocks of 64 (block 4 k + wave): rois usually arrive frame by frame, and
                // with 256 consecutive candidates per wave ONE wave did the whole frame's row geometry (fp64) while three idled
            float f0[4], y1[4], y2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(base + 64 * (4 * k + wave) + lane, n_pairs - 1);
                const float* roi = rois + 5 * (long long)(i / AH);
                f0[k] = roi[0]; y1[k] = roi[2]; y2[k] = roi[4];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int il = 64 * (4 * k + wave) + lane, i = base + il;
                int hit = 0, dy = 0;
                if (i < n_pairs && (int)f0[k] == b) {
                    const AxisGeom q = ra_axis(y1[k], y2[k], scale, H, AH, i % AH);
                    dy = q.s + 1 == h;
                    hit = q.ok && (q.s == h || dy);
                }
                const unsigned long long m = __ballot(hit);
                if (hit) list[256 * wave + cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(2 * il + dy);
                cnt += __popcll(m);
            }
            if (lane == 0) s_cnt[sp][wave] = cnt;
        }
        __syncthreads();
        const int p1 = s_cnt[sp][0], p2 = p1 + s_cnt[sp][1], p3 = p2 + s_cnt[sp][2], n = p3 + s_cnt[sp][3];
        if (n > 0) {
            // the four sublists close up into ONE list (a wave's entries move down, never up: read, barrier, write): an entry is then
            // one LDS read away for everyone, instead of a search through the four counts per use
            const int mine = wave == 0 ? 0 : (wave == 1 ? p1 : (wave == 2 ? p2 : p3));
            unsigned short keep[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) keep[k] = list[256 * wave + min(64 * k + lane, 255)];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (64 * k + lane < cnt) list[mine + 64 * k + lane] = keep[k];
            __syncthreads();
        }
        RAB_T(c_l1);
        RAB_ADD(c_list, c_l0, c_l1);
#ifdef RAB_CLOCKS
        c_n += n;
#endif
        // the sample-gradient operands of listed pair e (its list entry `code` read a step ahead) for my column and channels: always
        // issued (a pair past the end, a column or a pooled row that does not exist gets the out-of-range bit: zeros, no
        // traffic), so the loads in flight are countable.  A dead row's scalar offset must not be negative: the range check
        // subtracts it from the buffer size, and the out-of-range bit of the vector offset would land in range again.
        auto issue = [&](float4 (&q)[NQ], int e, int code) {
            const int pair = (2 * base + __builtin_amdgcn_readfirstlane(code)) >> 1;        // scalar from here on
            const int r = pair / AH, ah = pair - r * AH;
#if defined(RAB_CLOCKS) && defined(RAB_ABL) && (RAB_ABL & 2)      // tools/micro/rab_clock.hip only: no loads (every request out of range)
            const bool live = false;
#else
            const bool live = e < n;
#endif
            if (AVG) {                  // the two pooled rows of the sample's 2x2 window, my pooled column
                const bool d0 = !(live && ah >= 1), d1 = !(live && ah < PH);
                const unsigned b0 = d0 ? 0u : (unsigned)(((r * PH + ah - 1) * PW) * C) * 4u, b1 = d1 ? 0u : (unsigned)(((r * PH + ah) * PW) * C) * 4u;
                q[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, lcol | (d0 ? 0x80000000u : 0u), b0, 0));
                q[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, lcol | (d1 ? 0x80000000u : 0u), b1, 0));
            } else {
                q[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(gr, lcol | (live ? 0u : 0x80000000u),
                                                                                         live ? (unsigned)(((r * PH + ah) * PW) * C) * 4u : 0u, 0));
            }
        };
        auto process = [&](const float4 (&q)[NQ], const TapRec2 rec) {     // a listed pair: my column's two taps
            float4 v = q[0];
            if (AVG) {                  // avg_pool2d backward: my pooled column's two rows + the left column's (grad / 4 is in the weights)
                v.x += q[1].x; v.y += q[1].y; v.z += q[1].z; v.w += q[1].w;
                const float4 l = rab_row_shr2(v);
                v.x = l.x + v.x; v.y = l.y + v.y; v.z = l.z + v.z; v.w = l.w + v.w;
            }
            const int meta = __builtin_amdgcn_readfirstlane(rec.meta), nr = (meta >> 8) & 255, round = rec.meta & 255;
            float4* pl = (float4*)(mybuf + ((rec.off & 0xFFFFu) ^ lxor));
            float4* pr = (float4*)(mybuf + ((rec.off >> 16) ^ lxor));
            const float4 tl = make_float4(v.x * rec.wl, v.y * rec.wl, v.z * rec.wl, v.w * rec.wl);
            const float4 tr = make_float4(v.x * rec.wr, v.y * rec.wr, v.z * rec.wr, v.w * rec.wr);
#if defined(RAB_CLOCKS) && defined(RAB_ABL) && (RAB_ABL & 1)      // tools/micro/rab_clock.hip only: no read-add-write (results wrong)
            if (tl.x + tr.y == 12345.678f) *pl = tl;
            return;
#endif
            for (int s = 0; s < nr; ++s) {
                if (round == s) {
                    float4 a = *pl;
                    a.x += tl.x; a.y += tl.y; a.z += tl.z; a.w += tl.w;
                    *pl = a;
                    float4 c = *pr;         // (a neighbour's left cell may be my right cell: after the left stores, in order)
                    c.x += tr.x; c.y += tr.y; c.z += tr.z; c.w += tr.w;
                    *pr = c;
                }
            }
        };
        float4 q[RB2_DEPTH][NQ];
        if (n > 0) {            // my group's first pairs: list entries grp, grp + GROUPS, ...
#pragma unroll
            for (int k = 0; k < RB2_DEPTH; ++k) issue(q[k], GROUPS * k + grp, list[min(GROUPS * k + grp, n - 1)]);
        }
        if (base == 0) {        // the row buffer is cleared under the first requests' round trip (a barrier follows before its first use)
            for (int i = threadIdx.x; i < (W + 2) * 32; i += 256) ((float4*)rowbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e0 = 0; e0 < n; e0 += RB2_CHUNK) {
            RAB_T(c_r0);
            {       // ---- the chunk's tap records: thread = (pair of the chunk, sample column)
                const int u = threadIdx.x >> 3, sx = threadIdx.x & 7, e = e0 + u;
                int cell = W;           // (the bin rows W, W + 1)
                float wl = 0.f, wr = 0.f;
                if (e < n && sx < AW) {
                    const int code = 2 * base + (int)list[e], pair = code >> 1;
                    const int r = pair / AH, ah = pair - r * AH;
                    const float* roi = rois + 5 * (long long)r;
                    const AxisGeom qr = ra_axis(roi[2], roi[4], scale, H, AH, ah);
                    const AxisGeom qc = ra_axis(roi[1], roi[3], scale, W, AW, sx);
                    if (qc.ok) {
                        const float fh = (code & 1) ? qr.r : (1 - qr.r);
                        cell = qc.s; wl = fh * (1 - qc.r); wr = fh * qc.r;               // taps on cells ws, ws + 1
                    }
                }
                int round = 0;          // columns before mine that share my cell (cells do not decrease along a pair)
#pragma unroll
                for (int k = 1; k < 8; ++k) {
                    const int c = __shfl_up(cell, k, 8);
                    if (sx >= k && c == cell && cell < W) ++round;
                }
                // does my right cell belong to the next column as its left cell?  (then the pair's taps are added left, then right)
                int nr = round + 1;                              // over the pair's 8 threads: the most rounds
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) nr = max(nr, __shfl_xor(nr, m, 8));
                TapRec2 rec;
                rec.off = (unsigned)(cell * CHB + ((cell & 7) << 5)) | ((unsigned)((cell + 1) * CHB + (((cell + 1) & 7) << 5)) << 16);
                rec.wl = AVG ? wl * 0.25f : wl; rec.wr = AVG ? wr * 0.25f : wr;
                rec.meta = round | (nr << 8);
                s_rec[u][sx] = rec;
            }
            __syncthreads();
            RAB_T(c_r1);
            const int ne = min(RB2_CHUNK, n - e0);
            // my group's pairs of the chunk (u = grp, grp + GROUPS, ...), RB2_DEPTH of them in flight per wave; no wave waits for
            // another.  A step's LDS reads that do not depend on it -- the NEXT pair's record, the list entry of the pair
            // requested next -- are issued in front of the step's read-add-write, so they share its round trip (the LDS returns
            // in order) instead of heading chains of their own
            TapRec2 rec = s_rec[grp][sxl];
            int code = list[min(e0 + GROUPS * RB2_DEPTH + grp, n - 1)];
            for (int u0 = 0; u0 < ne; u0 += GROUPS * RB2_DEPTH) {
#pragma unroll
                for (int k = 0; k < RB2_DEPTH; ++k) {
                    const int u = u0 + GROUPS * k + grp;
                    const TapRec2 recn = s_rec[min(u + GROUPS, RB2_CHUNK - 1)][sxl];
                    const int coden = list[min(e0 + u + GROUPS * RB2_DEPTH + GROUPS, n - 1)];
                    if (u < ne) process(q[k], rec);
                    issue(q[k], e0 + u + GROUPS * RB2_DEPTH, code);
                    rec = recn;
                    code = coden;
                }
            }
            __syncthreads();            // the records (and, at the end of a segment, the list) are rewritten next
            RAB_T(c_r2);
            RAB_ADD(c_rec, c_r0, c_r1); RAB_ADD(c_add, c_r1, c_r2);
        }
    }
    __syncthreads();
    RAB_T(c_st0);
    constexpr int C4B = 32 / GROUPS;            // 4-channel groups of a cell in one buffer
    float* o = gfeat + (((long long)b * H + h) * W) * C + chunk * (128 / GROUPS);
    const float4* buf1 = (const float4*)rowbuf + (size_t)(W + 2) * C4B;
    for (int i = threadIdx.x; i < W * C4B; i += 256) {
        const int x = i / C4B, c4 = i % C4B, at = x * C4B + (c4 ^ ((x & 7) << 1));
        float4 a = ((const float4*)rowbuf)[at];
        if (GROUPS == 2) { const float4 c = buf1[at]; a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; }
        *(float4*)(o + (long long)x * C + 4 * c4) = a;
    }
#ifdef RAB_CLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned long long* o6 = g_rab_clk[blockIdx.x];
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        o6[0] = c_end - c_begin; o6[1] = c_list; o6[2] = c_rec; o6[3] = c_end - c_st0; o6[4] = c_add; o6[5] = c_n;
    }
#endif
}

// ---------------------------------------------------------------- ROIPool
// roi_pooling_kernel.cu:24-93.  One workgroup per (roi, 64-channel chunk); lane = channel, the four
// waves split the pooled rows (ph % 4).  The PHxPW results of the chunk are staged in LDS so that an
// NCHW output (the flatten order vrd.fc6 expects) is written as one contiguous 64*PH*PW run.
__global__ void __launch_bounds__(256)
roi_pool_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                    int* __restrict__ argmax, int C, int H, int W, int PH, int PW, float scale, Strides fs,
                    Strides os, int out_nchw, const int* __restrict__ geom, long long cap_cells) {
    extern __shared__ float lds[];
    const int P = PH * PW;
    float* sval = lds;
    int* sarg = (int*)(lds + 64 * P);
    if (geom) {                          // extent of the (NHWC, packed) maps read from device memory: i2v_roi_pool_fwd_geom
        H = geom[0]; W = geom[1];
        if (H <= 0 || W <= 0 || (long long)H * W > cap_cells) H = W = 0;       // does not fit the buffer: every bin empty
        fs = feat_strides(I2V_LAYOUT_NHWC, C, H, W);
    }
    const int chunks = (C + 63) / 64;
    const int r = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = c0 + lane;
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const int x1 = (int)roundf(roi[1] * scale), y1 = (int)roundf(roi[2] * scale);
    const int x2 = (int)roundf(roi[3] * scale), y2 = (int)roundf(roi[4] * scale);
    const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    const bool live = c < C;
    const float* fb = feat + (long long)b * fs.b + (live ? c : 0) * fs.c;
    for (int ph = wave; ph < PH; ph += 4) {
        int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
        hs = min(max(hs + y1, 0), H); he = min(max(he + y1, 0), H);
        for (int pw = 0; pw < PW; ++pw) {
            int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
            ws = min(max(ws + x1, 0), W); we = min(max(we + x1, 0), W);
            bool empty = (he <= hs) || (we <= ws);
            float m = empty ? 0.f : -FLT_MAX;
            int mi = -1;
            if (live)
                for (int h = hs; h < he; ++h)
                    for (int w = ws; w < we; ++w) {
                        float v = fb[h * fs.h + w * fs.w];
                        if (v > m) { m = v; mi = h * W + w; }
                    }
            sval[lane * P + ph * PW + pw] = m;
            sarg[lane * P + ph * PW + pw] = mi;
        }
    }
    __syncthreads();
    const int nch = min(64, C - c0);
    if (out_nchw) {                      // (r, c0..c0+nch, :, :) is contiguous
        float* o = out + (long long)r * C * P + (long long)c0 * P;
        int* a = argmax + (long long)r * C * P + (long long)c0 * P;
        for (int e = threadIdx.x; e < nch * P; e += 256) { o[e] = sval[e]; a[e] = sarg[e]; }
    } else {                             // NHWC: (r, p, c) with c contiguous
        for (int p = wave; p < P; p += 4)
            if (live) {
                long long o = (long long)r * P * C + (long long)p * C + c;
                out[o] = sval[lane * P + p];
                argmax[o] = sarg[lane * P + p];
            }
    }
}

// NHWC map, C % 128 == 0: a workgroup owns 128 channels of one ROI; a 32-lane group takes one bin at a time with four
// channels per lane (one 16-B load per window cell instead of four 4-B ones), eight bins in flight per workgroup.  Window
// cells are visited in the same (h, w) order per channel as above: same maxima, same first-maximum argmax.
__global__ void __launch_bounds__(512)
roi_pool_fwd_c128_kernel(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                         int* __restrict__ argmax, int C, int H, int W, int PH, int PW, float scale, int out_nchw,
                         const int* __restrict__ geom, long long cap_cells) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int P = PH * PW;
    float* sval = lds;
    int* sarg = (int*)(lds + 128 * P);
    if (geom) {                          // i2v_roi_pool_fwd_geom: H, W live in device memory
        H = geom[0]; W = geom[1];
        if (H <= 0 || W <= 0 || (long long)H * W > cap_cells) H = W = 0;
    }
    const int chunks = C / 128;
    const int r = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * 128;
    const int l32 = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const int x1 = (int)roundf(roi[1] * scale), y1 = (int)roundf(roi[2] * scale);
    const int x2 = (int)roundf(roi[3] * scale), y2 = (int)roundf(roi[4] * scale);
    const int rw = max(x2 - x1 + 1, 1), rh = max(y2 - y1 + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    // Window cells are fetched EIGHT at a time through a buffer resource, unconditionally (a slot beyond the window gets the
    // out-of-range bit: zeros, no traffic, skipped by the compare): one memory round trip per eight cells instead of one per
    // cell -- the scan is a dependent chain (first maximum wins), the loads need not be.  Cells are visited in (h, w) order.
    const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, 0x7FFFFFFCu, 0x00020000);
    const unsigned lane_off = (unsigned)(c0 + 4 * l32) * 4u, row_bytes = (unsigned)C * 4u;
    const unsigned img = (unsigned)b * (unsigned)(H * W);
    const int ngrp = blockDim.x >> 5;            // 16 bins in flight per workgroup (512 threads): 49 bins in four rounds
    for (int p = grp; p < P; p += ngrp) {
        const int ph = p / PW, pw = p - ph * PW;
        int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
        hs = min(max(hs + y1, 0), H); he = min(max(he + y1, 0), H);
        int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
        ws = min(max(ws + x1, 0), W); we = min(max(we + x1, 0), W);
        const bool empty = (he <= hs) || (we <= ws);
        float4 m = empty ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
        int4 mi = make_int4(-1, -1, -1, -1);
        const int nw = we - ws, cells = empty ? 0 : (he - hs) * nw;
        int h = hs, w = ws;
        for (int k0 = 0; k0 < cells; k0 += 8) {
            float4 v[8];
            int id[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = k0 + u < cells;
                id[u] = live ? h * W + w : -1;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    fr, ((img + (unsigned)(h * W + w)) * row_bytes + lane_off) | (live ? 0u : 0x80000000u), 0, 0));
                if (++w == we) { w = ws; ++h; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (id[u] < 0) continue;                      // uniform across the 32-lane group
                if (v[u].x > m.x) { m.x = v[u].x; mi.x = id[u]; }
                if (v[u].y > m.y) { m.y = v[u].y; mi.y = id[u]; }
                if (v[u].z > m.z) { m.z = v[u].z; mi.z = id[u]; }
                if (v[u].w > m.w) { m.w = v[u].w; mi.w = id[u]; }
            }
        }
        const int e = (4 * l32) * P + p;
        sval[e] = m.x; sval[e + P] = m.y; sval[e + 2 * P] = m.z; sval[e + 3 * P] = m.w;
        sarg[e] = mi.x; sarg[e + P] = mi.y; sarg[e + 2 * P] = mi.z; sarg[e + 3 * P] = mi.w;
    }
    __syncthreads();
    if (out_nchw) {                      // (r, c0..c0+128, :, :) is contiguous: 16-byte rows when the chunk's base allows
        float* o = out + (long long)r * C * P + (long long)c0 * P;
        int* a = argmax + (long long)r * C * P + (long long)c0 * P;
        if ((((long long)r * C + c0) * P) % 4 == 0 && (128 * P) % 4 == 0) {
            for (int e = threadIdx.x * 4; e < 128 * P; e += 4 * blockDim.x) {
                *(float4*)(o + e) = *(const float4*)(sval + e);
                *(int4*)(a + e) = *(const int4*)(sarg + e);
            }
        } else {
            for (int e = threadIdx.x; e < 128 * P; e += blockDim.x) { o[e] = sval[e]; a[e] = sarg[e]; }
        }
    } else {                             // NHWC: (r, p, c) with c contiguous
        for (int e = threadIdx.x; e < 128 * P; e += blockDim.x) {
            const int p = e >> 7, ch = e & 127;
            const long long o = (long long)r * P * C + (long long)p * C + c0 + ch;
            out[o] = sval[ch * P + p];
            argmax[o] = sarg[ch * P + p];
        }
    }
}

__global__ void roi_pool_bwd_kernel(const float* __restrict__ gout, const int* __restrict__ argmax,
                                    const float* __restrict__ rois, float* __restrict__ gfeat, long long total,
                                    int C, int W, int P, Strides fs, int out_nchw) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        int c, r;
        if (out_nchw) { c = (idx / P) % C; r = idx / P / C; }
        else { c = idx % C; r = idx / C / P; }
        int a = argmax[idx];
        if (a < 0) continue;
        int b = (int)rois[5 * (long long)r];
        atomicAdd(gfeat + b * fs.b + c * fs.c + (a / W) * fs.h + (a % W) * fs.w, gout[idx]);
    }
}

// ---------------------------------------------------------------- ROIAlign, sampled (model._C)
// roi_layers.ROIAlign of the SGG_emb model (roi_layers/roi_align.py:20,:31 -> model._C, the maskrcnn-benchmark
// csrc; source absent from the reference tree, algorithm restated from the published ROIAlign_cuda.cu): no +1 on
// the ROI extent, extent clamped to >= 1, each bin is the mean of a gh x gw grid of bilinear samples (gh =
// sampling_ratio, or ceil(extent / pooled) when sampling_ratio <= 0), a sample more than a pixel outside the map
// contributes 0 and coordinates clamp at the borders.  fp32 throughout, one rounding per op.
struct RasGeom { float y0, x0, bh, bw; int gh, gw; };

__device__ inline RasGeom ras_geometry(const float* roi, float scale, int PH, int PW, int sampling) {
    RasGeom g;
    const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    g.x0 = x1; g.y0 = y1;
    g.bh = rh / (float)PH; g.bw = rw / (float)PW;
    g.gh = sampling > 0 ? sampling : (int)ceilf(rh / (float)PH);
    g.gw = sampling > 0 ? sampling : (int)ceilf(rw / (float)PW);
    return g;
}

struct RasTaps { int ok, yl, xl, yh, xh; float w0, w1, w2, w3; };

__device__ inline RasTaps ras_taps(int H, int W, float y, float x) {
    RasTaps t;
    t.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    t.yl = (int)y; t.xl = (int)x;
    if (t.yl >= H - 1) { t.yh = t.yl = H - 1; y = (float)t.yl; } else t.yh = t.yl + 1;
    if (t.xl >= W - 1) { t.xh = t.xl = W - 1; x = (float)t.xl; } else t.xh = t.xl + 1;
    const float ly = y - (float)t.yl, lx = x - (float)t.xl, hy = 1.f - ly, hx = 1.f - lx;
    t.w0 = hy * hx; t.w1 = hy * lx; t.w2 = ly * hx; t.w3 = ly * lx;
    if (!t.ok) t.yl = t.xl = t.yh = t.xh = 0;
    return t;
}

// grid = R*PH*PW blocks (one output bin), lanes run over channels (VEC = 4: NHWC map, 16 B per lane and tap).
template <int VEC>
__global__ void __launch_bounds__(256)
roi_align_sampled_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
                             int B, int C, int H, int W, int PH, int PW, float scale, int sampling, Strides fs,
                             Strides os) {
    const int pw = blockIdx.x % PW, ph = (blockIdx.x / PW) % PH, r = blockIdx.x / (PW * PH);
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    const bool live = b >= 0 && b < B;
    const RasGeom g = ras_geometry(roi, scale, PH, PW, sampling);
    const float count = (float)(g.gh * g.gw);
    const float* fb = feat + (long long)(live ? b : 0) * fs.b;
    for (int c = threadIdx.x * VEC; c < C; c += blockDim.x * VEC) {
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int iy = 0; live && iy < g.gh; ++iy) {
            const float y = g.y0 + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.x0 + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                const RasTaps t = ras_taps(H, W, y, x);
                if (!t.ok) continue;
                const float* p = fb + c * fs.c;
                float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                if (VEC == 4) {
                    *(float4*)v0 = *(const float4*)(p + t.yl * fs.h + t.xl * fs.w);
                    *(float4*)v1 = *(const float4*)(p + t.yl * fs.h + t.xh * fs.w);
                    *(float4*)v2 = *(const float4*)(p + t.yh * fs.h + t.xl * fs.w);
                    *(float4*)v3 = *(const float4*)(p + t.yh * fs.h + t.xh * fs.w);
                } else {
                    v0[0] = p[t.yl * fs.h + t.xl * fs.w]; v1[0] = p[t.yl * fs.h + t.xh * fs.w];
                    v2[0] = p[t.yh * fs.h + t.xl * fs.w]; v3[0] = p[t.yh * fs.h + t.xh * fs.w];
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] += t.w0 * v0[k] + t.w1 * v1[k] + t.w2 * v2[k] + t.w3 * v3[k];
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) out[r * os.b + (c + k) * os.c + ph * os.h + pw * os.w] = acc[k] / count;
    }
}

// backward: same decomposition, fp32 atomics on the map gradient (RoIAlignBackwardFeature); summation order free.
__global__ void __launch_bounds__(256)
roi_align_sampled_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gfeat,
                             int B, int C, int H, int W, int PH, int PW, float scale, int sampling, Strides fs,
                             Strides os) {
    const int pw = blockIdx.x % PW, ph = (blockIdx.x / PW) % PH, r = blockIdx.x / (PW * PH);
    const float* roi = rois + 5 * (long long)r;
    const int b = (int)roi[0];
    if (b < 0 || b >= B) return;
    const RasGeom g = ras_geometry(roi, scale, PH, PW, sampling);
    const float count = (float)(g.gh * g.gw);
    float* fb = gfeat + (long long)b * fs.b;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float top = gout[r * os.b + c * os.c + ph * os.h + pw * os.w];
        float* p = fb + c * fs.c;
        for (int iy = 0; iy < g.gh; ++iy) {
            const float y = g.y0 + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
                const float x = g.x0 + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
                const RasTaps t = ras_taps(H, W, y, x);
                if (!t.ok) continue;
                atomicAdd(p + t.yl * fs.h + t.xl * fs.w, top * t.w0 / count);
                atomicAdd(p + t.yl * fs.h + t.xh * fs.w, top * t.w1 / count);
                atomicAdd(p + t.yh * fs.h + t.xl * fs.w, top * t.w2 / count);
                atomicAdd(p + t.yh * fs.h + t.xh * fs.w, top * t.w3 / count);
            }
        }
    }
}

inline Strides out_strides(int layout, int C, int PH, int PW) {
    Strides s;
    if (layout == I2V_LAYOUT_NHWC) { s.c = 1; s.w = C; s.h = (long long)PW * C; s.b = (long long)PH * PW * C; }
    else { s.w = 1; s.h = PW; s.c = (long long)PH * PW; s.b = (long long)C * PH * PW; }
    return s;
}

}  // namespace

extern "C" int32_t i2v_roi_align_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H,
                                     int32_t W, const float* rois, int32_t R, int32_t PH, int32_t PW,
                                     float scale, int32_t avg, float* out, int32_t out_layout, void* stream) {
    if (R == 0) return I2V_OK;          // empty roi set: nothing to do (tensors may have null storage)
    I2V_CHECK_ARG(feat && rois && out, "roi_align_fwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H >= 2 && W >= 2 && R >= 0 && PH > 0 && PW > 0, "roi_align_fwd: bad shape");
    I2V_CHECK_ARG(avg == 0 || avg == 1, "roi_align_fwd: avg must be 0/1");
    I2V_CHECK_ARG(avg || (PH > 1 && PW > 1), "roi_align_fwd: aligned grid needs >=2 points per side");
    hipStream_t st = (hipStream_t)stream;
    Strides os = out_strides(out_layout, C, PH, PW);
    if (feat_layout == I2V_LAYOUT_NHWC && (C % 4) == 0) {
        const int groups = (PW + 1) / 2;
        // one ROI x 128 channels per workgroup (each tap through the L2 once): NHWC output, <= 64 sample points, 32-bit offsets;
        // I2V_ROIALIGN_COLS = 1 keeps the column-pair kernel, 0 the round-1 row kernel
        const bool per_roi = g_i2v_tuning[I2V_TUNE_ROIALIGN_COLS] == 2 && out_layout == I2V_LAYOUT_NHWC && (C % 128) == 0 &&
                             (PH + avg) * (PW + avg) <= 64 && (long long)B * H * W * C * 4 < (1ll << 31);
        if (per_roi) {
            if (R * (C / 128) < 512) {          // a small launch: 64 channels per workgroup, one pass
                if (avg) roi_align_fwd_roi_kernel<1, 64><<<R * (C / 64), 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale);
                else roi_align_fwd_roi_kernel<0, 64><<<R * (C / 64), 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale);
            } else if (avg) roi_align_fwd_roi_kernel<1><<<R * (C / 128), 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale);
            else roi_align_fwd_roi_kernel<0><<<R * (C / 128), 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale);
        } else if (!g_i2v_tuning[I2V_TUNE_ROIALIGN_COLS]) {
            if (avg) roi_align_fwd_nhwc<1><<<R * PH, 256, 0, st>>>(feat, rois, out, C, H, W, PH, PW, scale, os);
            else roi_align_fwd_nhwc<0><<<R * PH, 256, 0, st>>>(feat, rois, out, C, H, W, PH, PW, scale, os);
        } else if (avg) {
            roi_align_fwd_nhwc_cols<1><<<(R + 7) / 8 * 8 * PH * groups, 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale, os);
        } else {
            roi_align_fwd_nhwc_cols<0><<<(R + 7) / 8 * 8 * PH * groups, 256, 0, st>>>(feat, rois, out, R, C, H, W, PH, PW, scale, os);
        }
    } else {
        Strides fs = feat_strides(feat_layout, C, H, W);
        long long total = (long long)R * C * PH * PW;
        int grid = (int)fmin((double)i2v_cdiv(total, 256), 65535.0 * 4);
        if (avg) roi_align_fwd_generic<1><<<grid, 256, 0, st>>>(feat, rois, out, total, C, H, W, PH, PW, scale, fs, os);
        else roi_align_fwd_generic<0><<<grid, 256, 0, st>>>(feat, rois, out, total, C, H, W, PH, PW, scale, fs, os);
    }
    I2V_CHECK_LAUNCH("roi_align_fwd");
    return I2V_OK;
}

extern "C" int32_t i2v_roi_align_bwd(const float* gout, int32_t out_layout, const float* rois, int32_t R,
                                     int32_t PH, int32_t PW, float scale, int32_t avg, float* gfeat,
                                     int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                                     void* stream) {
    if (R == 0) return I2V_OK;          // empty roi set: nothing to do (tensors may have null storage)
    I2V_CHECK_ARG(gout && rois && gfeat, "roi_align_bwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H >= 2 && W >= 2 && R >= 0 && PH > 0 && PW > 0, "roi_align_bwd: bad shape");
    I2V_CHECK_ARG(avg == 0 || avg == 1, "roi_align_bwd: avg must be 0/1");
    hipStream_t st = (hipStream_t)stream;
    Strides os = out_strides(out_layout, C, PH, PW), fs = feat_strides(feat_layout, C, H, W);
    if (avg) roi_align_bwd_kernel<1><<<R * (PH + 1), 256, 0, st>>>(gout, rois, gfeat, C, H, W, PH, PW, scale, fs, os);
    else roi_align_bwd_kernel<0><<<R * PH, 256, 0, st>>>(gout, rois, gfeat, C, H, W, PH, PW, scale, fs, os);
    I2V_CHECK_LAUNCH("roi_align_bwd");
    return I2V_OK;
}

extern "C" size_t i2v_roi_align_bwd_gather_workspace_bytes(int32_t R, int32_t C, int32_t PH, int32_t PW, int32_t avg) {
    (void)R; (void)C; (void)PH; (void)PW; (void)avg;
    return 0;                           // round 5: one kernel, nothing staged in global memory (the entry stays for its callers)
}

extern "C" int32_t i2v_roi_align_bwd_gather(const float* gout, const float* rois, int32_t R, int32_t PH, int32_t PW, float scale,
                                            int32_t avg, float* gfeat, int32_t B, int32_t C, int32_t H, int32_t W, void* ws,
                                            size_t ws_bytes, void* stream) {
    (void)ws; (void)ws_bytes;
    I2V_CHECK_ARG(gout && rois && gfeat, "roi_align_bwd_gather: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H >= 2 && W >= 2 && R > 0 && PH > 0 && PW > 0, "roi_align_bwd_gather: bad shape");
    I2V_CHECK_ARG(avg == 0 || avg == 1, "roi_align_bwd_gather: avg must be 0/1");
    I2V_CHECK_ARG(C % 128 == 0, "roi_align_bwd_gather: C must be a multiple of 128 (NHWC in and out)");
    const size_t AH = PH + avg, AW = PW + avg;
    I2V_CHECK_ARG(AW <= RAB_MAXA, "roi_align_bwd_gather: at most 8 sample columns (pooled_w + avg <= 8)");
    const bool form2 = g_i2v_tuning[I2V_TUNE_ROIALIGN_BWD] != 0;        // round 6: waves that never meet (0: round 5's staged batches)
    const size_t lds = (size_t)(W + 2) * 128 * sizeof(float) + (form2 ? 0 : (size_t)RAB_BATCH * RAB_MAXA * 128 * sizeof(float)) + RAB_SEG * sizeof(unsigned short);
    I2V_CHECK_ARG(lds <= 60 * 1024 && (long long)R * AH < (1ll << 29), "roi_align_bwd_gather: map too wide for the LDS row buffer");
    I2V_CHECK_ARG((long long)B * H * (C / 128) < (1ll << 31), "roi_align_bwd_gather: grid too large");
    I2V_CHECK_ARG((long long)R * PH * PW * C * 4 < (1ll << 31), "roi_align_bwd_gather: grad_out beyond the 2 GiB a 32-bit buffer offset reaches");
    hipStream_t st = (hipStream_t)stream;
    // the second form with one pair group per workgroup (128 channels) or two (64 channels, half the dependent chain): two where
    // the launch fits the chip at once (its time is then its longest list's), one where it runs in rounds (the sum counts)
    const bool two = form2 && (long long)B * H * (C / 64) <= 4ll * 256;       // four workgroups on each of the 256 CUs
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<0, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<1, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_row2_kernel<1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        return true;
    }();
    (void)once;
    const dim3 grid(B * H * (C / 128)), grid2(B * H * (C / 64));
    const int p7 = avg && PH == 7 && PW == 7;
#define RAB2(A, P, G, GRID) roi_align_bwd_row2_kernel<A, P, G><<<GRID, 256, lds, st>>>(gout, rois, gfeat, R, PH, PW, scale, B, C, H, W)
    if (form2 && two) { if (p7) RAB2(1, 1, 2, grid2); else if (avg) RAB2(1, 0, 2, grid2); else RAB2(0, 0, 2, grid2); }
    else if (form2) { if (p7) RAB2(1, 1, 1, grid); else if (avg) RAB2(1, 0, 1, grid); else RAB2(0, 0, 1, grid); }
#undef RAB2
    else if (avg && PH == 7 && PW == 7) roi_align_bwd_row_kernel<1, 1><<<grid, 256, lds, st>>>(gout, rois, gfeat, R, PH, PW, scale, B, C, H, W);
    else if (avg) roi_align_bwd_row_kernel<1, 0><<<grid, 256, lds, st>>>(gout, rois, gfeat, R, PH, PW, scale, B, C, H, W);
    else roi_align_bwd_row_kernel<0, 0><<<grid, 256, lds, st>>>(gout, rois, gfeat, R, PH, PW, scale, B, C, H, W);
    I2V_CHECK_LAUNCH("roi_align_bwd_gather");
    return I2V_OK;
}

namespace {
int32_t roi_pool_fwd_launch(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                            const int32_t* geom, long long cap_cells, const float* rois, int32_t R, int32_t PH, int32_t PW,
                            float scale, float* out, int32_t* argmax, int32_t out_layout, void* stream) {
    I2V_CHECK_ARG(PH * PW <= 256, "roi_pool_fwd: pooled grid too large for the LDS stage");
    Strides fs = feat_strides(feat_layout, C, H, W), os = out_strides(out_layout, C, PH, PW);
    const int c128 = g_i2v_tuning[I2V_TUNE_ROIPOOL_C128];
    const long long feat_bytes = (long long)B * C * 4 * (geom ? cap_cells : (long long)H * W);       // 32-bit byte offsets in the kernel
    if (c128 && feat_layout == I2V_LAYOUT_NHWC && C % 128 == 0 && (size_t)128 * PH * PW * 8 <= 64 * 1024 && feat_bytes < (1ll << 31)) {
        roi_pool_fwd_c128_kernel<<<R * (C / 128), 512, (size_t)128 * PH * PW * 8, (hipStream_t)stream>>>(
            feat, rois, out, argmax, C, H, W, PH, PW, scale, out_layout == I2V_LAYOUT_NCHW, geom, cap_cells);
        I2V_CHECK_LAUNCH("roi_pool_fwd");
        return I2V_OK;
    }
    size_t lds = (size_t)64 * PH * PW * 8;
    roi_pool_fwd_kernel<<<R * ((C + 63) / 64), 256, lds, (hipStream_t)stream>>>(
        feat, rois, out, argmax, C, H, W, PH, PW, scale, fs, os, out_layout == I2V_LAYOUT_NCHW, geom, cap_cells);
    I2V_CHECK_LAUNCH("roi_pool_fwd");
    return I2V_OK;
}
}  // namespace

extern "C" int32_t i2v_roi_pool_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H,
                                    int32_t W, const float* rois, int32_t R, int32_t PH, int32_t PW, float scale,
                                    float* out, int32_t* argmax, int32_t out_layout, void* stream) {
    if (R == 0) return I2V_OK;          // empty roi set: nothing to do (tensors may have null storage)
    I2V_CHECK_ARG(feat && rois && out && argmax, "roi_pool_fwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "roi_pool_fwd: bad shape");
    return roi_pool_fwd_launch(feat, feat_layout, B, C, H, W, nullptr, 0, rois, R, PH, PW, scale, out, argmax, out_layout,
                               stream);
}

extern "C" int32_t i2v_roi_pool_fwd_geom(const float* feat, int32_t B, int32_t C, const int32_t* geom, int64_t cap_cells,
                                         const float* rois, int32_t R, int32_t PH, int32_t PW, float scale, float* out,
                                         int32_t* argmax, int32_t out_layout, void* stream) {
    if (R == 0) return I2V_OK;
    I2V_CHECK_ARG(feat && geom && rois && out && argmax, "roi_pool_fwd_geom: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && cap_cells > 0 && R >= 0 && PH > 0 && PW > 0, "roi_pool_fwd_geom: bad shape");
    return roi_pool_fwd_launch(feat, I2V_LAYOUT_NHWC, B, C, 1, 1, geom, (long long)cap_cells, rois, R, PH, PW, scale, out,
                               argmax, out_layout, stream);
}

extern "C" int32_t i2v_roi_pool_bwd(const float* gout, const int32_t* argmax, int32_t out_layout,
                                    const float* rois, int32_t R, int32_t PH, int32_t PW, float* gfeat,
                                    int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                                    void* stream) {
    if (R == 0) return I2V_OK;          // empty roi set: nothing to do (tensors may have null storage)
    I2V_CHECK_ARG(gout && argmax && rois && gfeat, "roi_pool_bwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "roi_pool_bwd: bad shape");
    Strides fs = feat_strides(feat_layout, C, H, W);
    long long total = (long long)R * C * PH * PW;
    int grid = (int)fmin((double)i2v_cdiv(total, 256), 65535.0 * 4);
    roi_pool_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gout, argmax, rois, gfeat, total, C, W, PH * PW, fs,
                                                                out_layout == I2V_LAYOUT_NCHW);
    I2V_CHECK_LAUNCH("roi_pool_bwd");
    return I2V_OK;
}

extern "C" int32_t i2v_roi_align_sampled_fwd(const float* feat, int32_t feat_layout, int32_t B, int32_t C, int32_t H,
                                             int32_t W, const float* rois, int32_t R, int32_t PH, int32_t PW,
                                             float scale, int32_t sampling_ratio, float* out, int32_t out_layout,
                                             void* stream) {
    if (R == 0) return I2V_OK;
    I2V_CHECK_ARG(feat && rois && out, "roi_align_sampled_fwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "roi_align_sampled_fwd: bad shape");
    I2V_CHECK_ARG((long long)R * PH * PW < (1ll << 31), "roi_align_sampled_fwd: too many bins for one launch");
    Strides fs = feat_strides(feat_layout, C, H, W), os = out_strides(out_layout, C, PH, PW);
    const int grid = R * PH * PW;
    if (feat_layout == I2V_LAYOUT_NHWC && (C % 4) == 0)
        roi_align_sampled_fwd_kernel<4><<<grid, 256, 0, (hipStream_t)stream>>>(feat, rois, out, B, C, H, W, PH, PW, scale,
                                                                               sampling_ratio, fs, os);
    else
        roi_align_sampled_fwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(feat, rois, out, B, C, H, W, PH, PW, scale,
                                                                               sampling_ratio, fs, os);
    I2V_CHECK_LAUNCH("roi_align_sampled_fwd");
    return I2V_OK;
}

extern "C" int32_t i2v_roi_align_sampled_bwd(const float* gout, int32_t out_layout, const float* rois, int32_t R,
                                             int32_t PH, int32_t PW, float scale, int32_t sampling_ratio, float* gfeat,
                                             int32_t feat_layout, int32_t B, int32_t C, int32_t H, int32_t W,
                                             void* stream) {
    if (R == 0) return I2V_OK;
    I2V_CHECK_ARG(gout && rois && gfeat, "roi_align_sampled_bwd: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H > 0 && W > 0 && R >= 0 && PH > 0 && PW > 0, "roi_align_sampled_bwd: bad shape");
    I2V_CHECK_ARG((long long)R * PH * PW < (1ll << 31), "roi_align_sampled_bwd: too many bins for one launch");
    Strides fs = feat_strides(feat_layout, C, H, W), os = out_strides(out_layout, C, PH, PW);
    roi_align_sampled_bwd_kernel<<<R * PH * PW, 256, 0, (hipStream_t)stream>>>(gout, rois, gfeat, B, C, H, W, PH, PW, scale,
                                                                              sampling_ratio, fs, os);
    I2V_CHECK_LAUNCH("roi_align_sampled_bwd");
    return I2V_OK;
}
