// Winograd F(2x2, 3x3) for the stride-1 / pad-1 3x3 convolutions of a FROZEN backbone (SGG_emb: the reference detaches
// it, so the filters need no gradient and are transformed once).  The step is MFMA-throughput bound and the 23 layer3
// 3x3 convolutions are 29 % of its MACs; Winograd does them with 2.25x fewer: per 2x2 output tile 16 multiplies per
// (cin, cout) pair instead of 36.  fp32 throughout; the transform constants are 0, +-1, +-0.5, so the result differs
// from the direct convolution by a few ulp-scale roundings (measured ~1e-6 relative).
//
//   U[xi][n][c]  = (G g G^T)[xi]            filter transform, once per filter (i2v_winograd_filter)
//   V[xi][t][c]  = (B^T d B)[xi]            input transform of the 4x4 patch of tile t  (HBM/L2 streaming)
//   M[xi][t][n]  = sum_c V[xi][t][c] U[xi][n][c]      16 independent (T x Cin) x (Cin x Cout) GEMMs, ONE launch
//   y(tile t)    = A^T M[.][t][n] A          output transform + BN scale/shift + ReLU  (streaming)
// xi = 4*row + col of the 4x4 transform domain.  Only the GEMM uses the matrix cores; the two transforms are
// streaming kernels that overlap with MFMA work of the other stream.
#include "common.h"
#include <algorithm>

extern "C" int32_t i2v_gemm_tn_batched(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K,
                                       int32_t nbatch, long long stride_x, long long stride_gy, long long stride_gw,
                                       void* stream);
extern "C" int32_t i2v_gemm_nt_batched(const float* a, const float* b, float* c, int32_t M, int32_t N, int32_t K,
                                       int32_t nbatch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                                       void* split_ws, size_t split_ws_bytes, void* stream);

namespace {

__device__ inline float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Loads and stores of the F(4x4,3x3) transforms go through buffer resources, UNCONDITIONALLY: a tap outside the image (or a
// pixel outside it on the way out) gets the 2 GiB bit OR-ed into its byte offset, the hardware returns 0 for it (drops the
// store) and no traffic results.  Written as ``inside ? x[..] : 0.f`` every one of a thread's 36 loads sat under a branch of its
// own (s_cbranch_execz) with an s_waitcnt vmcnt(0) in front of most of them -- six to twelve exposed round trips per thread
// (round 4, from the ISA; 10.1 -> 5.9 us for the input transform of a layer3 frame pair).  Byte offsets are 32-bit: the
// launchers refuse tensors of 2 GiB and more.
constexpr unsigned WINV = 0x80000000u;
__device__ inline __amdgpu_buffer_rsrc_t wrsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFCu, 0x00020000);
}
__device__ inline float wload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ inline void wstore(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}
// 0 when 0 <= iy < H and 0 <= ix < W, else the out-of-range bit
__device__ inline unsigned outside(int iy, int H, int ix, int W) {
    return (unsigned)((iy | (H - 1 - iy) | ix | (W - 1 - ix)) >> 31) & WINV;
}

// one thread per (filter n, channel c): 9 taps -> 16 transform-domain values
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int c = (int)(idx % Cin), n = (int)(idx / Cin);
    float g[3][3];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(((long long)n * 3 + ky) * 3 + kx) * Cin + c];
    float t[4][3];                      // G g : G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
    for (int r = 0; r < 4; ++r) {       // (G g) G^T
        const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]),
                    u3 = t[r][2];
        const long long plane = (long long)Cout * Cin;
        float* o = U + (long long)(4 * r) * plane + (long long)n * Cin + c;
        o[0] = u0; o[plane] = u1; o[2 * plane] = u2; o[3 * plane] = u3;
    }
}

// one thread per (tile, 4 channels): V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
__global__ void __launch_bounds__(256)
wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C, int th, int tw) {
    const int c4n = C >> 2;
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long T = (long long)B * th * tw;
    if (idx >= T * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    const long long t = idx / c4n;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), b = (int)(t / ((long long)tw * th));
    float4 d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ix = 2 * tx - 1 + q;
            d[r][q] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? *(const float4*)(x + (((long long)b * H + iy) * W + ix) * C + c)
                                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 m[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // B^T d (rows)
        m[0][q] = f4sub(d[0][q], d[2][q]);
        m[1][q] = f4add(d[1][q], d[2][q]);
        m[2][q] = f4sub(d[2][q], d[1][q]);
        m[3][q] = f4sub(d[1][q], d[3][q]);
    }
    const long long plane = T * C;
    float* o = V + t * C + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {       // (B^T d) B (columns)
        *(float4*)(o + (long long)(4 * r + 0) * plane) = f4sub(m[r][0], m[r][2]);
        *(float4*)(o + (long long)(4 * r + 1) * plane) = f4add(m[r][1], m[r][2]);
        *(float4*)(o + (long long)(4 * r + 2) * plane) = f4sub(m[r][2], m[r][1]);
        *(float4*)(o + (long long)(4 * r + 3) * plane) = f4sub(m[r][1], m[r][3]);
    }
}

// one thread per (tile, 4 filters): Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; then scale/shift (frozen BN) and ReLU
__global__ void __launch_bounds__(256)
wino_output_kernel(const float* __restrict__ Mx, const float* __restrict__ scale, const float* __restrict__ shift,
                   float* __restrict__ y, int B, int H, int W, int N, int th, int tw, int relu) {
    const int n4n = N >> 2;
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long T = (long long)B * th * tw;
    if (idx >= T * n4n) return;
    const int n = (int)(idx % n4n) * 4;
    const long long t = idx / n4n;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), b = (int)(t / ((long long)tw * th));
    const long long plane = T * N;
    const float* src = Mx + t * N + n;
    float4 s[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // A^T M (rows)
        const float4 m0 = *(const float4*)(src + (long long)(0 + q) * plane), m1 = *(const float4*)(src + (long long)(4 + q) * plane);
        const float4 m2 = *(const float4*)(src + (long long)(8 + q) * plane), m3 = *(const float4*)(src + (long long)(12 + q) * plane);
        s[0][q] = f4add(f4add(m0, m1), m2);
        s[1][q] = f4sub(f4sub(m1, m2), m3);
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale) sc = *(const float4*)(scale + n);
    if (shift) sh = *(const float4*)(shift + n);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int oy = 2 * ty + r;
        if (oy >= H) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ox = 2 * tx + q;
            if (ox >= W) continue;
            float4 v = q == 0 ? f4add(f4add(s[r][0], s[r][1]), s[r][2]) : f4sub(f4sub(s[r][1], s[r][2]), s[r][3]);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *(float4*)(y + (((long long)b * H + oy) * W + ox) * N + n) = v;
        }
    }
}


// ---------------------------------------------------------------- F(4x4, 3x3): 36 multiplies per 16 outputs (4x fewer
// than direct).  Lavin & Gray's matrices; fp32 error ~1e-5 relative (measured on the backbone shapes) because the
// transform constants reach 8 and 1/24.  One thread per (tile, channel): a 6x6 patch held as scalars.
__device__ inline void bt6(const float d[6], float t[6]) {           // B^T d
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    t[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ inline void at6(const float m[6], float s[4]) {           // A^T m
    s[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    s[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
    s[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
    s[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
__device__ inline void g6(const float g[3], float u[6]) {            // G g
    u[0] = 0.25f * g[0];
    u[1] = (-1.f / 6.f) * (g[0] + g[1] + g[2]);
    u[2] = (-1.f / 6.f) * (g[0] - g[1] + g[2]);
    u[3] = (1.f / 24.f) * g[0] + (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
    u[4] = (1.f / 24.f) * g[0] - (1.f / 12.f) * g[1] + (1.f / 6.f) * g[2];
    u[5] = g[2];
}

__global__ void wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int c = (int)(idx % Cin), n = (int)(idx / Cin);
    float t[6][3];
    for (int kx = 0; kx < 3; ++kx) {
        float g[3], u[6];
        for (int ky = 0; ky < 3; ++ky) g[ky] = w[(((long long)n * 3 + ky) * 3 + kx) * Cin + c];
        g6(g, u);
        for (int r = 0; r < 6; ++r) t[r][kx] = u[r];
    }
    const long long plane = (long long)Cout * Cin;
    for (int r = 0; r < 6; ++r) {
        float u[6];
        g6(t[r], u);
        for (int q = 0; q < 6; ++q) U[(long long)(6 * r + q) * plane + (long long)n * Cin + c] = u[q];
    }
}

// Filter of the DATA-GRADIENT convolution in the Winograd domain: gx = conv3x3(gy, w') with w'[c][ky][kx][n] =
// w[n][2-ky][2-kx][c] (taps flipped, channels swapped) -> U'[36][Cin][Cout].  n is the fast thread index: the 36 plane
// stores are coalesced, the 9 tap reads are strided (the filter is the small side).
__global__ void wino4_filter_dgrad_kernel(const float* __restrict__ w, float* __restrict__ U, int Cout, int Cin) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int n = (int)(idx % Cout), c = (int)(idx / Cout);
    float t[6][3];
    for (int kx = 0; kx < 3; ++kx) {
        float g[3], u[6];
        for (int ky = 0; ky < 3; ++ky) g[ky] = w[(((long long)n * 3 + (2 - ky)) * 3 + (2 - kx)) * Cin + c];
        g6(g, u);
        for (int r = 0; r < 6; ++r) t[r][kx] = u[r];
    }
    const long long plane = (long long)Cout * Cin;
    for (int r = 0; r < 6; ++r) {
        float u[6];
        g6(t[r], u);
        for (int q = 0; q < 6; ++q) U[(long long)(6 * r + q) * plane + (long long)c * Cout + n] = u[q];
    }
}

// One thread per (tile, channel).  All index arithmetic is 32-bit (the launchers refuse tensors of 2 GiB and more; the 64-bit
// form spent ~660 VALU instructions per thread on addresses); the 36 planes of V / M are reached through the buffer
// instruction's SCALAR offset (plane * k is uniform), so a plane access costs no vector arithmetic at all.
__global__ void __launch_bounds__(256)
wino4_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C, int th, int tw) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned T = (unsigned)B * th * tw;
    if (idx >= T * (unsigned)C) return;
    const unsigned c = idx % (unsigned)C, t = idx / (unsigned)C;
    const int tx = (int)(t % (unsigned)tw), ty = (int)((t / (unsigned)tw) % (unsigned)th), b = (int)(t / ((unsigned)tw * th));
    const __amdgpu_buffer_rsrc_t xr = wrsrc(x), vr = wrsrc(V);
    float d[6][6];                              // d[q][r]: all 36 taps requested before the first is used
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int ix = 4 * tx - 1 + q;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = 4 * ty - 1 + r;
            d[q][r] = wload(xr, (unsigned)((((b * H + iy) * W + ix) * C + (int)c) * 4) | outside(iy, H, ix, W));
        }
    }
    float m[6][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {               // column q of the patch through B^T (rows)
        float tt[6];
        bt6(d[q], tt);
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r][q] = tt[r];
    }
    const unsigned plane4 = T * (unsigned)C * 4u, o4 = (t * (unsigned)C + c) * 4u;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float v[6];
        bt6(m[r], v);
#pragma unroll
        for (int q = 0; q < 6; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q]), vr, o4, plane4 * (unsigned)(6 * r + q), 0);
    }
}

__global__ void __launch_bounds__(256)
wino4_output_kernel(const float* __restrict__ Mx, const float* __restrict__ scale, const float* __restrict__ shift,
                    float* __restrict__ y, int B, int H, int W, int N, int th, int tw, int relu,
                    const float* __restrict__ mask = nullptr) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    const unsigned T = (unsigned)B * th * tw;
    if (idx >= T * (unsigned)N) return;
    const unsigned n = idx % (unsigned)N, t = idx / (unsigned)N;
    const int tx = (int)(t % (unsigned)tw), ty = (int)((t / (unsigned)tw) % (unsigned)th), b = (int)(t / ((unsigned)tw * th));
    const __amdgpu_buffer_rsrc_t sr = wrsrc(Mx), yr = wrsrc(y), mr = wrsrc(mask ? mask : y);
    const __amdgpu_buffer_rsrc_t scr = wrsrc(scale ? scale : Mx), shr = wrsrc(shift ? shift : Mx);
    const unsigned plane4 = T * (unsigned)N * 4u, s4 = (t * (unsigned)N + n) * 4u;
    const unsigned no_mask = mask ? 0u : WINV;
    // every load of the thread is requested here, before the first use: the mask tile (data gradients; absent: the
    // out-of-range bit, zeros without traffic), the 36 plane values, scale and shift
    float mk[4][4], mv[6][6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = 4 * ty + r, ox = 4 * tx + q;
            mk[r][q] = wload(mr, (unsigned)((((b * H + oy) * W + ox) * N + (int)n) * 4) | outside(oy, H, ox, W) | no_mask);
        }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < 6; ++r)
            mv[q][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, s4, plane4 * (unsigned)(6 * r + q), 0));
    const float sc_l = wload(scr, (n * 4u) | (scale ? 0u : WINV)), sh_l = wload(shr, (n * 4u) | (shift ? 0u : WINV));
    const float sc = scale ? sc_l : 1.f, sh = shift ? sh_l : 0.f;
    float s[4][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float ss[4];
        at6(mv[q], ss);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r][q] = ss[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int oy = 4 * ty + r;
        float o[4];
        at6(s[r], o);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ox = 4 * tx + q;
            float v = o[q] * sc + sh;
            if (relu) v = fmaxf(v, 0.f);
            if (mask && !(mk[r][q] > 0.f)) v = 0.f;       // data gradient: the ReLU of the tensor it flows into
            wstore(yr, (unsigned)((((b * H + oy) * W + ox) * N + (int)n) * 4) | outside(oy, H, ox, W), v);
        }
    }
}

// ---- row-split forms of the two F(4x4,3x3) transforms.  One thread per (tile, channel) is 320 x 256 threads on
// layer3 -- five waves per CU, each with 36 loads and 36 stores: one exposed round trip each way and nothing to hide it
// behind (10 + 7 us per layer, 30 layers per step).  Here blockIdx.y picks ONE row of the transformed tile: a thread
// loads only the patch rows that row's B^T (A^T) coefficients touch (3-4 of 6 for the input, 4-5 of 6 for the output),
// the launch has 6x (4x) the threads, and every value is computed by the same expression as in the one-thread form.
template <int R> __device__ inline float bt6_row(const float d[6]) {
    if (R == 0) return 4.f * d[0] - 5.f * d[2] + d[4];
    if (R == 1) return -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    if (R == 2) return 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    if (R == 3) return -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    if (R == 4) return 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    return 4.f * d[1] - 5.f * d[3] + d[5];
}
template <int R> __device__ inline bool bt6_needs(int k) {
    return R == 0 ? (k == 0 || k == 2 || k == 4) : R == 5 ? (k == 1 || k == 3 || k == 5) : (k >= 1 && k <= 4);
}
template <int R> __device__ inline float at6_row(const float m[6]) {
    if (R == 0) return m[0] + m[1] + m[2] + m[3] + m[4];
    if (R == 1) return m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
    if (R == 2) return m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
    return m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
template <int R> __device__ inline bool at6_needs(int k) {
    return R == 0 ? k <= 4 : R == 3 ? k >= 1 : (k >= 1 && k <= 4);
}

template <int R>
__device__ inline void wino4_input_row(const float* __restrict__ x, float* __restrict__ V, int H, int W, int C, int b,
                                       int ty, int tx, long long t, int c, long long plane) {
    float mrow[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int ix = 4 * tx - 1 + q;
        float d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            d[k] = 0.f;
            if (!bt6_needs<R>(k)) continue;
            const int iy = 4 * ty - 1 + k;
            d[k] = wload(wrsrc(x), (unsigned)((((b * H + iy) * W + ix) * C + c) * 4) | outside(iy, H, ix, W));
        }
        mrow[q] = bt6_row<R>(d);
    }
    float v[6];
    bt6(mrow, v);
    float* o = V + t * C + c;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[(long long)(6 * R + q) * plane] = v[q];
}

__global__ void __launch_bounds__(256)
wino4_input_rows_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int H, int W, int C, int th, int tw) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long T = (long long)B * th * tw;
    if (idx >= T * C) return;
    const int c = (int)(idx % C);
    const long long t = idx / C;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), b = (int)(t / ((long long)tw * th));
    const long long plane = T * C;
    switch (blockIdx.y) {                   // uniform per workgroup
    case 0: wino4_input_row<0>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    case 1: wino4_input_row<1>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    case 2: wino4_input_row<2>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    case 3: wino4_input_row<3>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    case 4: wino4_input_row<4>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    default: wino4_input_row<5>(x, V, H, W, C, b, ty, tx, t, c, plane); break;
    }
}

template <int R>
__device__ inline void wino4_output_row(const float* __restrict__ src, long long plane, float s[6]) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float m[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) m[k] = at6_needs<R>(k) ? src[(long long)(6 * k + q) * plane] : 0.f;
        s[q] = at6_row<R>(m);
    }
}

__global__ void __launch_bounds__(256)
wino4_output_rows_kernel(const float* __restrict__ Mx, const float* __restrict__ scale, const float* __restrict__ shift,
                         float* __restrict__ y, int B, int H, int W, int N, int th, int tw, int relu,
                         const float* __restrict__ mask = nullptr) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long T = (long long)B * th * tw;
    if (idx >= T * N) return;
    const int n = (int)(idx % N);
    const long long t = idx / N;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), b = (int)(t / ((long long)tw * th));
    const int r = blockIdx.y, oy = 4 * ty + r;
    if (oy >= H) return;
    const long long plane = T * N;
    const float* src = Mx + t * N + n;
    float s[6];
    switch (r) {                            // uniform per workgroup
    case 0: wino4_output_row<0>(src, plane, s); break;
    case 1: wino4_output_row<1>(src, plane, s); break;
    case 2: wino4_output_row<2>(src, plane, s); break;
    default: wino4_output_row<3>(src, plane, s); break;
    }
    float o[4];
    at6(s, o);
    const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
    const __amdgpu_buffer_rsrc_t yr = wrsrc(y), mr = wrsrc(mask ? mask : y);
    const unsigned no_mask = mask ? 0u : WINV;
    float mk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ox = 4 * tx + q;
        mk[q] = wload(mr, (unsigned)((((b * H + oy) * W + ox) * N + n) * 4) | outside(oy, H, ox, W) | no_mask);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ox = 4 * tx + q;
        float v = o[q] * sc + sh;
        if (relu) v = fmaxf(v, 0.f);
        if (mask && !(mk[q] > 0.f)) v = 0.f;
        wstore(yr, (unsigned)((((b * H + oy) * W + ox) * N + n) * 4) | outside(oy, H, ox, W), v);
    }
}

// ---- filter gradient in the F(4x4,3x3) domain.  With U = G g G^T the forward is y = A^T [U (.) B^T d B] A per tile, so
//   dL/dg = G^T [ sum over tiles (A gy A^T) (.) (B^T d B) ] G
// 36 plane GEMMs  X[p] (Cout x Cin) = Ygy[p]^T (T x Cout) . V[p] (T x Cin)  with a quarter of the direct form's MACs (the
// reduction runs over tiles, not pixels x taps), one transform of gy (A gy A^T: 4x4 -> 6x6, zero beyond the image), the
// input transform of the forward, and a 36 -> 9 transform per (filter, channel) at the end.
__device__ inline void a6(const float v[4], float o[6]) {            // A v  (A = (A^T)^T, 6 x 4)
    o[0] = v[0];
    o[1] = v[0] + v[1] + v[2] + v[3];
    o[2] = v[0] - v[1] + v[2] - v[3];
    o[3] = v[0] + 2.f * v[1] + 4.f * v[2] + 8.f * v[3];
    o[4] = v[0] - 2.f * v[1] + 4.f * v[2] - 8.f * v[3];
    o[5] = v[3];
}
__device__ inline void gt6(const float x[6], float o[3]) {           // G^T x
    o[0] = 0.25f * x[0] + (-1.f / 6.f) * (x[1] + x[2]) + (1.f / 24.f) * (x[3] + x[4]);
    o[1] = (-1.f / 6.f) * (x[1] - x[2]) + (1.f / 12.f) * (x[3] - x[4]);
    o[2] = (-1.f / 6.f) * (x[1] + x[2]) + (1.f / 6.f) * (x[3] + x[4]) + x[5];
}

__global__ void __launch_bounds__(256)
wino4_gy_kernel(const float* __restrict__ gy, float* __restrict__ Y, int B, int H, int W, int N, int th, int tw,
                float* __restrict__ zero_ptr, long long zero_n4) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long T = (long long)B * th * tw;
    // side job: clear the accumulator of the 36 plane GEMMs that follow (they accumulate over pixel splits with atomics) --
    // a memset node per trained 3x3 layer and step otherwise (41 per instance_styleD step)
    for (long long i = idx; i < zero_n4; i += (long long)gridDim.x * blockDim.x) ((float4*)zero_ptr)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx >= T * N) return;
    const int n = (int)(idx % N);
    const long long t = idx / N;
    const int tx = (int)(t % tw), ty = (int)((t / tw) % th), b = (int)(t / ((long long)tw * th));
    float m[6][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {               // column q of the 4x4 tile through A (rows)
        const int ox = 4 * tx + q;
        float d[4], o[6];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = 4 * ty + r;
            d[r] = wload(wrsrc(gy), (unsigned)((((b * H + oy) * W + ox) * N + n) * 4) | outside(oy, H, ox, W));
        }
        a6(d, o);
#pragma unroll
        for (int r = 0; r < 6; ++r) m[r][q] = o[r];
    }
    const long long plane = T * N;
    float* o = Y + t * N + n;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float v[6];
        a6(m[r], v);
#pragma unroll
        for (int q = 0; q < 6; ++q) o[(long long)(6 * r + q) * plane] = v[q];
    }
}

// gw[n][ky][kx][c] = beta * gw + row_scale[n] * (G^T X G)[ky][kx],  X[6r+q][n][c]
// nparts > 1 (round 6, ordered sums): X holds the pixel split's parts side by side ([part][36][Cout x Cin]); they are added
// here in part order -- the plane GEMMs' second pass costs no launch of its own
__global__ void wino4_wgrad_final_kernel(const float* __restrict__ X, float* __restrict__ gw, const float* __restrict__ row_scale,
                                         int Cout, int Cin, float beta, int nparts) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)Cout * Cin) return;
    const int c = (int)(idx % Cin), n = (int)(idx / Cin);
    const long long plane = (long long)Cout * Cin;
    float t[3][6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float x[6], o[3];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float v = X[(long long)(6 * r + q) * plane + idx];
            for (int s = 1; s < nparts; ++s) v += X[((long long)s * 36 + (6 * r + q)) * plane + idx];
            x[r] = v;
        }
        gt6(x, o);
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k][q] = o[k];
    }
    const float rs = row_scale ? row_scale[n] : 1.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float o[3];
        gt6(t[ky], o);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* dst = gw + (((long long)n * 3 + ky) * 3 + kx) * Cin + c;
            *dst = (beta != 0.f ? beta * *dst : 0.f) + rs * o[kx];
        }
    }
}

}  // namespace

extern "C" int32_t i2v_winograd_filter(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream) {
    I2V_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "winograd_filter: bad argument");
    const long long n = (long long)Cout * Cin;
    wino_filter_kernel<<<(unsigned)i2v_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w, U, Cout, Cin);
    I2V_CHECK_LAUNCH("winograd_filter");
    return I2V_OK;
}

extern "C" size_t i2v_conv3x3_winograd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 256;
    const size_t T = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    return i2v_align(16 * T * Cin * sizeof(float)) + i2v_align(16 * T * Cout * sizeof(float));
}

extern "C" int32_t i2v_conv3x3_winograd_fwd(const float* x, const float* U, const float* scale, const float* shift,
                                            float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                            int32_t relu, void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(x && U && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_winograd_fwd: bad argument");
    I2V_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0, "conv3x3_winograd_fwd: Cin and Cout must be multiples of 4");
    if (!ws || ws_bytes < i2v_conv3x3_winograd_workspace_bytes(B, H, W, Cin, Cout)) {
        i2v_set_error("conv3x3_winograd_fwd: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int th = (H + 1) / 2, tw = (W + 1) / 2;
    const long long T = (long long)B * th * tw;
    float* V = (float*)ws;
    float* Mx = (float*)((char*)ws + i2v_align(16 * (size_t)T * Cin * sizeof(float)));
    wino_input_kernel<<<(unsigned)i2v_cdiv(T * (Cin / 4), 256), 256, 0, st>>>(x, V, B, H, W, Cin, th, tw);
    int rc = i2v_gemm_nt_batched(V, U, Mx, (int32_t)T, Cout, Cin, 16, T * Cin, (long long)Cout * Cin, T * Cout, nullptr, 0, stream);
    if (rc) return rc;
    wino_output_kernel<<<(unsigned)i2v_cdiv(T * (Cout / 4), 256), 256, 0, st>>>(Mx, scale, shift, y, B, H, W, Cout, th, tw, relu);
    I2V_CHECK_LAUNCH("conv3x3_winograd_fwd");
    return I2V_OK;
}

// ---- F(4x4,3x3) entry points (variant = 4): same contract, U is (36, Cout, Cin)
extern "C" int32_t i2v_winograd4_filter(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream) {
    I2V_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "winograd4_filter: bad argument");
    const long long n = (long long)Cout * Cin;
    wino4_filter_kernel<<<(unsigned)i2v_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w, U, Cout, Cin);
    I2V_CHECK_LAUNCH("winograd4_filter");
    return I2V_OK;
}

extern "C" int32_t i2v_winograd4_filter_dgrad(const float* w, float* U, int32_t Cout, int32_t Cin, void* stream) {
    I2V_CHECK_ARG(w && U && Cout > 0 && Cin > 0, "winograd4_filter_dgrad: bad argument");
    const long long n = (long long)Cout * Cin;
    wino4_filter_dgrad_kernel<<<(unsigned)i2v_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(w, U, Cout, Cin);
    I2V_CHECK_LAUNCH("winograd4_filter_dgrad");
    return I2V_OK;
}

extern "C" size_t i2v_conv3x3_winograd4_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 256;
    const size_t T = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    return i2v_align(36 * T * Cin * sizeof(float)) + i2v_align(36 * T * Cout * sizeof(float));
}

static int winograd4_impl(const float* x, const float* U, const float* scale, const float* shift, const float* mask,
                          float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                          int32_t relu, void* ws, size_t ws_bytes, void* stream, float* v_keep = nullptr) {
    I2V_CHECK_ARG(x && U && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_winograd4_fwd: bad argument");
    I2V_CHECK_ARG(Cin % 4 == 0, "conv3x3_winograd4_fwd: Cin must be a multiple of 4");
    // the transform kernels reach the 36 planes of V / M with 32-bit byte offsets (plane (6 r + q) at (6 r + q) * T * C * 4, through a
    // 0x7FFFFFFC-byte descriptor): the WORKSPACE side is ~2.25x the padded activation and is what must stay below 2 GiB
    I2V_CHECK_ARG((long long)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31) &&
                  36ll * B * ((H + 3) / 4) * ((W + 3) / 4) * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31),
                  "conv3x3_winograd4_fwd: the 36-plane transform workspace (or the activation) reaches 2 GiB (32-bit byte offsets)");
    if (!ws || ws_bytes < i2v_conv3x3_winograd4_workspace_bytes(B, H, W, Cin, Cout)) {
        i2v_set_error("conv3x3_winograd4_fwd: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)B * th * tw;
    float* V = v_keep ? v_keep : (float*)ws;        // v_keep: the transformed input outlives the call (the filter gradient reads it)
    float* Mx = (float*)((char*)ws + i2v_align(36 * (size_t)T * Cin * sizeof(float)));
    // row-split transforms (I2V_TUNE_WINO_ROWS; -1 = by size) pay where the one-thread-per-(tile, channel) launch cannot fill
    // the chip and runs ALONE (layer3: 34 vs 37 us per layer); on layer1/2 their redundant loads cost more than the parallelism
    // buys (79 vs 57 us), and inside the step graphs co-running branches fill the chip anyway (headline step 4.59 vs 4.61 ms
    // without them): the default is the plain form
    const int rows_env = g_i2v_tuning[I2V_TUNE_WINO_ROWS];   // bit 0: input, bit 1: output; -1: by size
    const int rows = rows_env >= 0 ? rows_env : (T * (Cin > Cout ? Cin : Cout) <= 98304 ? 3 : 0);
    if (rows & 1) wino4_input_rows_kernel<<<dim3((unsigned)i2v_cdiv(T * Cin, 256), 6), 256, 0, st>>>(x, V, B, H, W, Cin, th, tw);
    else wino4_input_kernel<<<(unsigned)i2v_cdiv(T * Cin, 256), 256, 0, st>>>(x, V, B, H, W, Cin, th, tw);
    int rc = i2v_gemm_nt_batched(V, U, Mx, (int32_t)T, Cout, Cin, 36, T * Cin, (long long)Cout * Cin, T * Cout, nullptr, 0, stream);
    if (rc) return rc;
    if (rows & 2) wino4_output_rows_kernel<<<dim3((unsigned)i2v_cdiv(T * Cout, 256), 4), 256, 0, st>>>(Mx, scale, shift, y, B, H, W, Cout, th, tw, relu, mask);
    else wino4_output_kernel<<<(unsigned)i2v_cdiv(T * Cout, 256), 256, 0, st>>>(Mx, scale, shift, y, B, H, W, Cout, th, tw, relu, mask);
    I2V_CHECK_LAUNCH("conv3x3_winograd4_fwd");
    return I2V_OK;
}

extern "C" int32_t i2v_conv3x3_winograd4_fwd(const float* x, const float* U, const float* scale, const float* shift,
                                             float* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                             int32_t relu, void* ws, size_t ws_bytes, void* stream) {
    return winograd4_impl(x, U, scale, shift, nullptr, y, B, H, W, Cin, Cout, relu, ws, ws_bytes, stream);
}

// Forward that leaves the transformed input V (36 x tiles x Cin floats, i2v_conv3x3_winograd4_v_bytes) in the caller's
// buffer: a trained layer hands it to i2v_conv3x3_winograd4_wgrad instead of transforming x a second time.
extern "C" size_t i2v_conv3x3_winograd4_v_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0) return 0;
    return 36 * (size_t)B * ((H + 3) / 4) * ((W + 3) / 4) * Cin * sizeof(float);
}

extern "C" int32_t i2v_conv3x3_winograd4_fwd_keep(const float* x, const float* U, const float* scale, const float* shift,
                                                  float* y, float* v_out, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                                  int32_t Cout, int32_t relu, void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(v_out, "conv3x3_winograd4_fwd_keep: null V buffer");
    return winograd4_impl(x, U, scale, shift, nullptr, y, B, H, W, Cin, Cout, relu, ws, ws_bytes, stream, v_out);
}

extern "C" int32_t i2v_conv3x3_winograd4_dgrad(const float* gy, const float* U, const float* out_scale, const float* mask,
                                               float* gx, int32_t B, int32_t H, int32_t W, int32_t Cout, int32_t Cin,
                                               void* ws, size_t ws_bytes, void* stream) {
    // the data gradient of a stride-1 / pad-1 3x3 layer is the same convolution with U = i2v_winograd4_filter_dgrad(w):
    // gy has Cout channels, gx has Cin
    return winograd4_impl(gy, U, out_scale, nullptr, mask, gx, B, H, W, Cout, Cin, 0, ws, ws_bytes, stream);
}

extern "C" size_t i2v_conv3x3_winograd4_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 256;
    const size_t T = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);
    // the last region holds the 36 plane gradients -- once per part of the pixel split, so that an ordered sum
    // (I2V_TUNE_SPLIT_ATOMICS == 0) can leave the parts side by side for the final transform to add in order
    const size_t parts = (size_t)i2v_internal_wgrad_plane_splits((long long)T, Cout, Cin);
    return i2v_align(36 * T * Cin * sizeof(float)) + i2v_align(36 * T * Cout * sizeof(float)) +
           i2v_align(36 * (size_t)Cout * Cin * sizeof(float) * (parts > 1 ? parts : 1));
}

// Filter gradient of a stride-1 / pad-1 3x3 layer in the F(4x4,3x3) domain: gw (Cout,3,3,Cin) = beta * gw +
// row_scale[n] * wgrad(x, gy).  x (B,H,W,Cin), gy (B,H,W,Cout) NHWC; Cin % 4 == 0 and Cout % 4 == 0.
static int winograd4_wgrad_impl(const float* x, const float* v_in, const float* gy, const float* row_scale, float* gw,
                                int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float beta,
                                void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG((x || v_in) && gy && gw && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_winograd4_wgrad: bad argument");
    I2V_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0, "conv3x3_winograd4_wgrad: Cin and Cout must be multiples of 4");
    I2V_CHECK_ARG(beta == 0.f || beta == 1.f, "conv3x3_winograd4_wgrad: beta must be 0 or 1");
    I2V_CHECK_ARG((long long)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31) &&
                  36ll * B * ((H + 3) / 4) * ((W + 3) / 4) * (Cin > Cout ? Cin : Cout) * 4 < (1ll << 31),
                  "conv3x3_winograd4_wgrad: the 36-plane transform workspace (or the activation) reaches 2 GiB (32-bit byte offsets)");
    if (!ws || ws_bytes < i2v_conv3x3_winograd4_wgrad_workspace_bytes(B, H, W, Cin, Cout)) {
        i2v_set_error("conv3x3_winograd4_wgrad: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int th = (H + 3) / 4, tw = (W + 3) / 4;
    const long long T = (long long)B * th * tw;
    float* V = (float*)ws;
    float* Y = (float*)((char*)ws + i2v_align(36 * (size_t)T * Cin * sizeof(float)));
    float* X = (float*)((char*)Y + i2v_align(36 * (size_t)T * Cout * sizeof(float)));
    const int rows_env = g_i2v_tuning[I2V_TUNE_WINO_ROWS];
    const int rows = rows_env >= 0 ? rows_env : (T * Cin <= 98304 ? 1 : 0);
    if (v_in) V = const_cast<float*>(v_in);         // the forward's transformed input (i2v_conv3x3_winograd4_fwd_keep)
    else if (rows & 1) wino4_input_rows_kernel<<<dim3((unsigned)i2v_cdiv(T * Cin, 256), 6), 256, 0, st>>>(x, V, B, H, W, Cin, th, tw);
    else wino4_input_kernel<<<(unsigned)i2v_cdiv(T * Cin, 256), 256, 0, st>>>(x, V, B, H, W, Cin, th, tw);
    if (g_i2v_tuning[I2V_TUNE_SPLIT_ATOMICS] == 0) {
        // ordered: the parts of the pixel split side by side in X (the workspace has room for the planned count; fewer is legal),
        // no clear of X, summed in part order by the final transform
        const size_t room = ws_bytes - (size_t)((char*)X - (char*)ws);
        const int cap = (int)std::min<size_t>(room / (36 * (size_t)Cout * Cin * sizeof(float)), 1 << 20);
        int nparts = 1;
        wino4_gy_kernel<<<(unsigned)i2v_cdiv(T * Cout, 256), 256, 0, st>>>(gy, Y, B, H, W, Cout, th, tw, nullptr, 0);
        int rc = i2v_internal_gemm_tn_batched_parts(V, Y, X, (int32_t)T, Cout, Cin, 36, T * Cin, T * Cout, cap, &nparts, stream);
        if (rc) return rc;
        // two parts (layer3 / layer4) are added by the final transform itself; more (layer1: 28, layer2: 7 at eight frames) first
        // meet in slot 0 through the parallel reduce pass -- the final transform has one thread per filter element and would read
        // 36 x parts values each in a dependent chain (measured: 42 us per call instead of 9)
        if (nparts > 2) {
            i2v_internal_reduce_parts(X, nparts, 36, (long long)Cout * Cin, stream);
            nparts = 1;
        }
        wino4_wgrad_final_kernel<<<(unsigned)i2v_cdiv((long long)Cout * Cin, 256), 256, 0, st>>>(X, gw, row_scale, Cout, Cin, beta, nparts);
        I2V_CHECK_LAUNCH("conv3x3_winograd4_wgrad");
        return I2V_OK;
    }
    wino4_gy_kernel<<<(unsigned)i2v_cdiv(T * Cout, 256), 256, 0, st>>>(gy, Y, B, H, W, Cout, th, tw, X, 9ll * Cout * Cin);
    int rc = i2v_gemm_tn_batched_acc(V, Y, X, (int32_t)T, Cout, Cin, 36, T * Cin, T * Cout, (long long)Cout * Cin, stream);
    if (rc) return rc;
    wino4_wgrad_final_kernel<<<(unsigned)i2v_cdiv((long long)Cout * Cin, 256), 256, 0, st>>>(X, gw, row_scale, Cout, Cin, beta, 1);
    I2V_CHECK_LAUNCH("conv3x3_winograd4_wgrad");
    return I2V_OK;
}

extern "C" int32_t i2v_conv3x3_winograd4_wgrad(const float* x, const float* gy, const float* row_scale, float* gw,
                                               int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float beta,
                                               void* ws, size_t ws_bytes, void* stream) {
    return winograd4_wgrad_impl(x, nullptr, gy, row_scale, gw, B, H, W, Cin, Cout, beta, ws, ws_bytes, stream);
}

// Same with the input already in the Winograd domain (V of i2v_conv3x3_winograd4_fwd_keep on the same x).
extern "C" int32_t i2v_conv3x3_winograd4_wgrad_v(const float* v, const float* gy, const float* row_scale, float* gw,
                                                 int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, float beta,
                                                 void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(v, "conv3x3_winograd4_wgrad_v: null V");
    return winograd4_wgrad_impl(nullptr, v, gy, row_scale, gw, B, H, W, Cin, Cout, beta, ws, ws_bytes, stream);
}
