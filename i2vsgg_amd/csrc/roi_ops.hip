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
// deterministic, no atomics, no zero-fill in front.  Two kernels:
//   prep    per (roi, 128 channels): the sample gradients gs[r][s][c] (the 2x2 mean's backward folded in, same expression as
//           the scatter kernel) and, once per ROI, the sample geometry table {frame, ok, hs, ws, hr, wr};
//   gather  per (frame, map row h, 128 channels): the samples of that frame whose taps touch row h (hs == h or hs + 1 == h)
//           are listed in index order (a deterministic compaction, 2048 samples per pass); wave group g owns the cells
//           w = g (mod 8) of the row, walks the list and adds its taps -- (float)(g * (1. - hr) * (1 - wr)) ... exactly as the
//           scatter -- into a 128-channel row buffer in LDS; the row is then stored with 16-byte stores.
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

// prep: per (roi, 128 channels) the sample gradients gs[r][s][c]; once per ROI the geometry of its AH sample rows
// (ok = frame index or -1) and AW sample columns
template <int AVG>
__global__ void __launch_bounds__(256)
roi_align_bwd_prep_kernel(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ gs,
                          AxisGeom* __restrict__ rowg, AxisGeom* __restrict__ colg, int R, int C, int H, int W, int PH, int PW,
                          float scale) {
    const int nchunk = C >> 7;
    const int chunk = blockIdx.x % nchunk, r = blockIdx.x / nchunk;
    const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int AH = PH + AVG, AW = PW + AVG, NS = AH * AW;
    const float* roi = rois + 5 * (long long)r;
    if (chunk == 0) {
        if ((int)threadIdx.x < AH) {
            AxisGeom q = ra_axis(roi[2], roi[4], scale, H, AH, threadIdx.x);
            q.ok = q.ok ? (int)roi[0] : -1;
            rowg[(long long)r * AH + threadIdx.x] = q;
        } else if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + AW) {
            colg[(long long)r * AW + (threadIdx.x - 64)] = ra_axis(roi[1], roi[3], scale, W, AW, threadIdx.x - 64);
        }
    }
    // the ROI's PH x PW gradients of this chunk into LDS with every load in flight at once (one round trip), then the samples
    __shared__ __attribute__((aligned(16))) float4 sg[64 * 32];
    const float* go = gout + (long long)r * PH * PW * C + (chunk << 7);
    float* o = gs + ((long long)r * NS) * C + (chunk << 7) + 4 * l;
    {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = threadIdx.x + 256 * j;
            t[j] = (k < PH * PW * 32) ? *(const float4*)(go + (long long)(k >> 5) * C + 4 * (k & 31)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = threadIdx.x + 256 * j;
            if (k < PH * PW * 32) sg[k] = t[j];
        }
    }
    __syncthreads();
    for (int s = g; s < NS; s += 8) {
        const int ah = s / AW, aw = s - ah * AW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AVG) {          // avg_pool2d backward: grad / 4 summed over the <= 4 windows holding the sample, raster order
#pragma unroll
            for (int dy = -1; dy <= 0; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 0; ++dx) {
                    const int ph = ah + dy, pw = aw + dx;
                    if (ph >= 0 && ph < PH && pw >= 0 && pw < PW) {
                        const float4 t = sg[(ph * PW + pw) * 32 + l];
                        v.x += t.x / 4.f; v.y += t.y / 4.f; v.z += t.z / 4.f; v.w += t.w / 4.f;
                    }
                }
        } else {
            v = sg[s * 32 + l];
        }
        *(float4*)(o + (long long)s * C) = v;
    }
}

constexpr int RAB_SEG = 1024;           // (roi, sample row) pairs examined per list pass
constexpr int RAB_BATCH = 4;            // listed pairs whose sample gradients are staged in LDS together
constexpr int RAB_MAXA = 8;             // sample columns per row this kernel takes

// gather: per (frame b, map row h, 128 channels).  The (roi, sample row) pairs of frame b whose taps touch row h (hs == h:
// the upper taps; hs + 1 == h: the lower taps) are listed in index order -- the reference loop's order -- by a deterministic
// compaction; the 128-channel gradients of a batch of listed pairs' samples are staged in LDS by all threads at once (one
// memory round trip per batch, requested while the previous batch is added), then WAVE w adds them for ITS 32 channels into the row
// buffer, in list order: lane = (channel, cell parity) -- of a sample column's two taps (cells ws, ws + 1) one is even and one
// is odd, so every lane has exactly one tap per column and there is not a branch in the loop; the cells a lane meets along a
// pair are non-decreasing, a repeated cell (ws + 1 == ws') is forwarded in a register, and the current values of all eight are
// requested at once.  (Round 4's first form gave wave w the CELLS of class cell % 4 == w: every wave walked every column to
// find its one tap in four, ~300 instructions and four dependent LDS round trips per pair -- 1.8k cycles, 76 % of the kernel.)
// No two waves touch one channel, so there is nothing to synchronise but the batches.
__global__ void __launch_bounds__(256)
roi_align_bwd_gather_kernel(const float* __restrict__ gs, const AxisGeom* __restrict__ rowg, const AxisGeom* __restrict__ colg,
                            float* __restrict__ gfeat, int n_pairs, int AH, int AW, int C, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* rowbuf = lds;                                                    // [W + 2][128]: cell x, channel c at x * 128 + (c ^ ((x & 1) << 5))
                                                                            // (odd cells swap their 32-channel halves: the two half-waves
                                                                            // of a tap hit different banks); rows W, W + 1: a bin for taps outside the map
    float* stage = rowbuf + (size_t)(W + 2) * 128;                          // [RAB_BATCH][RAB_MAXA][128]
    int* list = (int*)(stage + RAB_BATCH * RAB_MAXA * 128);                 // [RAB_SEG]: pair index * 2 + (hs + 1 == h)
    __shared__ int s_wave_cnt[4], s_total;
    __shared__ AxisGeom s_col[RAB_BATCH][RAB_MAXA];
    __shared__ float s_hr[RAB_BATCH];
    __shared__ int s_dy[RAB_BATCH];
    const int nchunk = C >> 7;
    const int chunk = blockIdx.x % nchunk, h = (blockIdx.x / nchunk) % H, b = blockIdx.x / (nchunk * H);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ch = 32 * wave + (lane & 31), par = lane >> 5, chs = ch ^ (par << 5);
    for (int i = threadIdx.x; i < (W + 2) * 32; i += 256) ((float4*)rowbuf)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < n_pairs; base += RAB_SEG) {
        if (threadIdx.x == 0) s_total = 0;
        // ---- the keys of this segment: requested at once (RAB_SEG / 256 = 4 per thread), compacted from registers
        AxisGeom key[RAB_SEG / 256];
#pragma unroll
        for (int k = 0; k < RAB_SEG / 256; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            key[k].ok = -1; key[k].s = 0; key[k].r = 0.f;
            if (i < n_pairs) key[k] = rowg[i];
        }
        __syncthreads();
        const int rounds = (min(RAB_SEG, n_pairs - base) + 255) >> 8;
#pragma unroll
        for (int k = 0; k < RAB_SEG / 256; ++k) {
            if (k >= rounds) break;
            const int i = base + k * 256 + threadIdx.x;
            const int dy = key[k].s + 1 == h;
            const int hit = i < n_pairs && key[k].ok == b && (key[k].s == h || dy);
            const unsigned long long m = __ballot(hit);
            if (lane == 0) s_wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int off = s_total;
            for (int w = 0; w < wave; ++w) off += s_wave_cnt[w];
            if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = 2 * i + dy;
            __syncthreads();
            if (threadIdx.x == 0) s_total += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
            __syncthreads();
        }
        const int n = s_total;
        // the batch's operands travel while the PREVIOUS batch is added: the 128-channel gradients of its samples (16 bytes per
        // lane, 4 loads per thread) and, on the first threads, the column geometry / row fraction of its pairs
        constexpr int TL = RAB_BATCH * RAB_MAXA * 32 / 256;
        float4 t[TL];
        AxisGeom qn; float hrn = 0.f; int dyn = 0;
        auto request = [&](int e0) {
            const int nb = min(RAB_BATCH, n - e0);
#pragma unroll
            for (int j = 0; j < TL; ++j) {
                const int k = threadIdx.x + 256 * j;
                const int u = k / (RAB_MAXA * 32), aw = (k / 32) % RAB_MAXA, ll = k & 31;
                t[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < nb && aw < AW) {
                    const int pair = list[e0 + u] >> 1;
                    t[j] = *(const float4*)(gs + ((long long)pair * AW + aw) * C + (chunk << 7) + 4 * ll);
                }
            }
            qn.ok = 0; qn.s = 0; qn.r = 0.f; qn.pad = 0;
            if (threadIdx.x < nb * RAB_MAXA) {
                const int u = threadIdx.x / RAB_MAXA, aw = threadIdx.x % RAB_MAXA;
                const int pair = list[e0 + u] >> 1, r = pair / AH;
                if (aw < AW) qn = colg[(long long)r * AW + aw];
                if (aw == 0) { hrn = rowg[pair].r; dyn = list[e0 + u] & 1; }
            }
        };
        if (n > 0) request(0);
        for (int e0 = 0; e0 < n; e0 += RAB_BATCH) {
            const int nb = min(RAB_BATCH, n - e0);
#pragma unroll
            for (int j = 0; j < TL; ++j) ((float4*)stage)[threadIdx.x + 256 * j] = t[j];
            if (threadIdx.x < nb * RAB_MAXA) {
                const int u = threadIdx.x / RAB_MAXA, aw = threadIdx.x % RAB_MAXA;
                s_col[u][aw] = qn;
                if (aw == 0) { s_hr[u] = hrn; s_dy[u] = dyn; }
            }
            __syncthreads();
            if (e0 + RAB_BATCH < n) request(e0 + RAB_BATCH);
            // ---- add: pairs in list order, sample columns ascending.  WAVE w owns the cells of its class (cell % 4 == w) with
            // all 64 lanes (2 channels each): no two waves touch one cell.  The pair's eight column entries and its eight staged
            // gradients come into registers with one LDS wait each; a tap is then register arithmetic and one read-modify-write
            // of the row buffer
            for (int u = 0; u < nb; ++u) {
                const double fh = s_dy[u] ? (double)s_hr[u] : (1. - s_hr[u]);
                AxisGeom q[RAB_MAXA];
                float sv[RAB_MAXA], pre[RAB_MAXA], wx[RAB_MAXA];
                int at[RAB_MAXA];
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) q[aw] = s_col[u][aw];
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) sv[aw] = stage[(u * RAB_MAXA + aw) * 128 + ch];
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) {
                    const int t = (q[aw].s ^ par) & 1;                        // my tap of this column: the cell of my parity
                    const int cell = q[aw].ok ? q[aw].s + t : W + par;        // a column outside the map goes to the bin
                    wx[aw] = t ? q[aw].r : (1 - q[aw].r);                     // the scatter kernel's (1 - wr) / wr, in float as there
                    at[aw] = cell * 128 + chs;
                    pre[aw] = rowbuf[at[aw]];
                }
                float cur = 0.f;
                int prev = -1;
#pragma unroll
                for (int aw = 0; aw < RAB_MAXA; ++aw) {
                    const float v = (at[aw] == prev) ? cur : pre[aw];
                    // the scatter kernel's expressions: g * (1. - hr) * (1 - wr), g * (1. - hr) * wr, g * hr * (1 - wr), g * hr * wr
                    cur = v + (float)(sv[aw] * fh * wx[aw]);
                    rowbuf[at[aw]] = cur;
                    prev = at[aw];
                }
            }
            __syncthreads();
        }
    }
    float* o = gfeat + (((long long)b * H + h) * W) * C + (chunk << 7);
    for (int i = threadIdx.x; i < W * 32; i += 256) {
        const int x = i >> 5, c4 = i & 31;
        *(float4*)(o + (long long)x * C + 4 * c4) = ((const float4*)rowbuf)[x * 32 + (c4 ^ ((x & 1) << 3))];
    }
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
    const size_t AH = PH + avg, AW = PW + avg;
    return i2v_align((size_t)R * AH * AW * C * sizeof(float)) + i2v_align((size_t)R * AH * sizeof(AxisGeom)) +
           i2v_align((size_t)R * AW * sizeof(AxisGeom));
}

extern "C" int32_t i2v_roi_align_bwd_gather(const float* gout, const float* rois, int32_t R, int32_t PH, int32_t PW, float scale,
                                            int32_t avg, float* gfeat, int32_t B, int32_t C, int32_t H, int32_t W, void* ws,
                                            size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(gout && rois && gfeat && ws, "roi_align_bwd_gather: null pointer");
    I2V_CHECK_ARG(B > 0 && C > 0 && H >= 2 && W >= 2 && R > 0 && PH > 0 && PW > 0, "roi_align_bwd_gather: bad shape");
    I2V_CHECK_ARG(avg == 0 || avg == 1, "roi_align_bwd_gather: avg must be 0/1");
    I2V_CHECK_ARG(C % 128 == 0, "roi_align_bwd_gather: C must be a multiple of 128 (NHWC in and out)");
    const size_t AH = PH + avg, AW = PW + avg;
    I2V_CHECK_ARG(AW <= RAB_MAXA && AH <= 64, "roi_align_bwd_gather: at most 8 sample columns");
    if (ws_bytes < i2v_roi_align_bwd_gather_workspace_bytes(R, C, PH, PW, avg)) {
        i2v_set_error("roi_align_bwd_gather: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    const size_t lds = (size_t)(W + 2) * 128 * sizeof(float) + (size_t)RAB_BATCH * RAB_MAXA * 128 * sizeof(float) + RAB_SEG * sizeof(int);
    I2V_CHECK_ARG(lds <= 60 * 1024 && (long long)R * AH < (1ll << 30), "roi_align_bwd_gather: map too wide for the LDS row buffer");
    hipStream_t st = (hipStream_t)stream;
    float* gs = (float*)ws;
    AxisGeom* rowg = (AxisGeom*)((char*)ws + i2v_align((size_t)R * AH * AW * C * sizeof(float)));
    AxisGeom* colg = (AxisGeom*)((char*)rowg + i2v_align((size_t)R * AH * sizeof(AxisGeom)));
    if (avg) roi_align_bwd_prep_kernel<1><<<R * (C / 128), 256, 0, st>>>(gout, rois, gs, rowg, colg, R, C, H, W, PH, PW, scale);
    else roi_align_bwd_prep_kernel<0><<<R * (C / 128), 256, 0, st>>>(gout, rois, gs, rowg, colg, R, C, H, W, PH, PW, scale);
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)roi_align_bwd_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024);
        return true;
    }();
    (void)once;
    roi_align_bwd_gather_kernel<<<B * H * (C / 128), 256, lds, st>>>(gs, rowg, colg, gfeat, (int)(R * AH), (int)AH, (int)AW, C, H, W);
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
