// Discriminator-head pieces that are not plain GEMMs.
//
// netD_style (resnet_instance_styleD_bilinear.py:122-136) pools a factorised bilinear
// feature: z[b][d] = sum_pos sum_r x1[b,pos,d*rank+r] * x2[b,pos,d*rank+r], where x1/x2
// are the two 512 -> dim*rank projections.  The reference materialises x1*x2 and reduces
// it in three full-size passes; here product, rank-sum and spatial sum are one streaming
// pass over x1 and x2 (HBM-bound: 2 reads, no intermediate), and the backward is one
// elementwise pass that writes both projection gradients.
#include "common.h"

namespace {

constexpr int POOL_ROWS = 64;

// grid (row_chunks, n_img); blockDim = N/4 threads (N = dim*rank), one float4 column group each
__global__ void dstyle_pool_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                       float* __restrict__ z, long long rows, int dim, int rank) {
    extern __shared__ float part[];           // N floats
    const int N = dim * rank;
    const int img = blockIdx.y;
    const long long r0 = (long long)blockIdx.x * POOL_ROWS;
    const long long r1 = r0 + POOL_ROWS < rows ? r0 + POOL_ROWS : rows;
    const float* a = x1 + ((long long)img * rows) * N + threadIdx.x * 4;
    const float* b = x2 + ((long long)img * rows) * N + threadIdx.x * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0; r < r1; ++r) {
        float4 u = *(const float4*)(a + r * N), v = *(const float4*)(b + r * N);
        s.x += u.x * v.x; s.y += u.y * v.y; s.z += u.z * v.z; s.w += u.w * v.w;
    }
    *(float4*)&part[threadIdx.x * 4] = s;
    __syncthreads();
    for (int d = threadIdx.x; d < dim; d += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < rank; ++r) t += part[d * rank + r];
        atomicAdd(z + (long long)img * dim + d, t);
    }
}

__global__ void dstyle_pool_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ x1,
                                       const float* __restrict__ x2, float* __restrict__ g1, float* __restrict__ g2,
                                       long long rows, int n_img, int dim, int rank) {
    const int N = dim * rank;
    const long long total = (long long)n_img * rows * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % N);
        const int img = (int)(i / ((long long)rows * N));
        const float g = gz[(long long)img * dim + j / rank];
        const float a = x1[i], b = x2[i];
        g1[i] = g * b;
        g2[i] = g * a;
    }
}


// ------------------------------------------------------------------------------------------------
// netD_pixel (resnet_instance_styleD_bilinear.py:38-83): per ROI pixel, 1024 -> 512 ReLU -> 128 ReLU -> 1 -> sigmoid,
// no biases.  ONE kernel per direction: a workgroup owns 32 rows (ROI pixels); the 512- and 128-wide activations of
// those rows never leave the CU between layers -- they stay in LDS in exactly the swizzled [k-stage][row][32] image
// the MFMA fragment reads want, and are written to HBM once, for the backward.  The reference runs 3 convs + 3
// pointwise kernels forward (and their autograd backward), each a round trip through HBM.
//
// Both directions are the same machine: rows x K  ->(GEMM, weights N x K streamed through LDS)->  rows x N, twice.
//   forward :  x[.,1024] -> relu -> h1[.,512] -> relu -> h2[.,128] -> dot(w3) -> sigmoid
//   backward:  gh2[.,128] -> (W2^T) mask(h1>0) -> gh1[.,512] -> (W1^T) * (-lambda) -> gx[.,1024]   (GRL folded in)
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DP_ROWS = 32, DP_BK = 32, DP_CHUNK = 128, DP_THREADS = 256;

__device__ inline int dp_swz(int row, int kc) { return (kc ^ ((row >> 1) & 7)) << 2; }

// One layer for the workgroup's 32 rows: out[32][N] = A[32][K] * Bw[N][K]^T, N in chunks of 128 columns (wave w owns
// 32 of them).  A comes from global rows (A_LDS == false: `a_rows`, leading dimension K, rows >= m_valid read 0) or
// from an LDS image [K/32][32][32] (swizzled).  epi(row, col, value) is called once per output element.
template <int K, int N, bool A_LDS, class Epi>
__device__ inline void dp_layer(const float* __restrict__ a_rows, int m_valid, const float* a_img,
                                const float* __restrict__ Bw, float* stA, float* stB, Epi epi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int srow = tid >> 3, kc = tid & 7;
    constexpr int STAGES = K / DP_BK;
    for (int n0 = 0; n0 < N; n0 += DP_CHUNK) {
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 ra, rb[4];
        auto gload = [&](int s) {
            const int k0 = s * DP_BK + kc * 4;
            if (!A_LDS) ra = srow < m_valid ? *(const float4*)(a_rows + (long long)srow * K + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + srow + q * 32;
                rb[q] = n < N ? *(const float4*)(Bw + (long long)n * K + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto sstore = [&](int buf) {
            if (!A_LDS) *(float4*)&stA[buf * (DP_ROWS * DP_BK) + srow * DP_BK + dp_swz(srow, kc)] = ra;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = srow + q * 32;
                *(float4*)&stB[buf * (DP_CHUNK * DP_BK) + row * DP_BK + dp_swz(row, kc)] = rb[q];
            }
        };
        gload(0);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int s = 0; s < STAGES; ++s) {
            if (s + 1 < STAGES) gload(s + 1);
            const float* As = A_LDS ? a_img + s * (DP_ROWS * DP_BK) : stA + buf * (DP_ROWS * DP_BK);
            const float* Bs = stB + buf * (DP_CHUNK * DP_BK);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = i * 16 + fr;
                    av[i] = *(const float4*)&As[row * DP_BK + dp_swz(row, h * 4 + fg)];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wave * 32 + j * 16 + fr;
                    bv[j] = *(const float4*)&Bs[row * DP_BK + dp_swz(row, h * 4 + fg)];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                            const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                        }
            }
            if (s + 1 < STAGES) sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
        // C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) epi(i * 16 + 4 * fg + r, n0 + wave * 32 + j * 16 + fr, acc[i][j][r]);
        __syncthreads();          // the stage buffers are reused by the next chunk / layer
    }
}

// LDS: stage A 2x4 KB | stage B 2x16 KB | image1 [16][32][32] 64 KB | image2 [4][32][32] 16 KB   = 120 KB
constexpr int DP_LDS_FLOATS = 2 * DP_ROWS * DP_BK + 2 * DP_CHUNK * DP_BK + 16 * DP_ROWS * DP_BK + 4 * DP_ROWS * DP_BK;

__device__ inline void dp_img_store(float* img, int row, int col, float v) {
    img[((col >> 5) * DP_ROWS + row) * DP_BK + dp_swz(row, (col & 31) >> 2) + (col & 3)] = v;
}
__device__ inline float dp_img_load(const float* img, int row, int col) {
    return img[((col >> 5) * DP_ROWS + row) * DP_BK + dp_swz(row, (col & 31) >> 2) + (col & 3)];
}

__global__ void __launch_bounds__(DP_THREADS)
dpixel_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ w2,
                  const float* __restrict__ w3, float* __restrict__ h1, float* __restrict__ h2, float* __restrict__ d,
                  float* __restrict__ feat, int M, int pix_per_roi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stA = lds;
    float* stB = stA + 2 * DP_ROWS * DP_BK;
    float* img1 = stB + 2 * DP_CHUNK * DP_BK;
    float* img2 = img1 + 16 * DP_ROWS * DP_BK;
    const int m0 = blockIdx.x * DP_ROWS;
    const int mv = min(DP_ROWS, M - m0);
    dp_layer<1024, 512, false>(x + (long long)m0 * 1024, mv, nullptr, w1, stA, stB, [&](int row, int col, float v) {
        v = fmaxf(v, 0.f);
        dp_img_store(img1, row, col, v);
        if (row < mv) h1[(long long)(m0 + row) * 512 + col] = v;
    });
    dp_layer<512, 128, true>(nullptr, mv, img1, w2, stA, stB, [&](int row, int col, float v) {
        v = fmaxf(v, 0.f);
        dp_img_store(img2, row, col, v);
        if (row < mv) h2[(long long)(m0 + row) * 128 + col] = v;
    });
    // (dp_layer ended with a barrier: img2 is complete)  d = sigmoid(h2 . w3): 8 lanes per row, 16 columns each
    const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = part * 16 + c;
        s += dp_img_load(img2, row, col) * w3[col];
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0 && row < mv) d[m0 + row] = 1.f / (1.f + expf(-s));
    if (feat) {                   // context vector: mean of h2 over the pixels of each ROI (:70-74)
        const float inv = 1.f / (float)pix_per_roi;
        for (int e = threadIdx.x; e < DP_ROWS * 128; e += DP_THREADS) {
            const int r = e >> 7, col = e & 127;
            if (r < mv) atomicAdd(feat + (long long)((m0 + r) / pix_per_roi) * 128 + col, dp_img_load(img2, r, col) * inv);
        }
    }
}

// backward chain.  w1t = W1^T (1024 x 512), w2t = W2^T (512 x 128): reduction-major copies made by the launcher.
__global__ void __launch_bounds__(DP_THREADS)
dpixel_bwd_kernel(const float* __restrict__ gd, const float* __restrict__ gfeat, const float* __restrict__ dsig,
                  const float* __restrict__ h1, const float* __restrict__ h2, const float* __restrict__ w1t,
                  const float* __restrict__ w2t, const float* __restrict__ w3, float* __restrict__ g3,
                  float* __restrict__ gh2, float* __restrict__ gh1, float* __restrict__ gx, int M, int pix_per_roi,
                  float neg_lambda) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* stA = lds;
    float* stB = stA + 2 * DP_ROWS * DP_BK;
    float* img1 = stB + 2 * DP_CHUNK * DP_BK;        // gh1 image
    float* img2 = img1 + 16 * DP_ROWS * DP_BK;       // gh2 image
    const int m0 = blockIdx.x * DP_ROWS;
    const int mv = min(DP_ROWS, M - m0);
    const float inv = 1.f / (float)pix_per_roi;
    // g3 = gd * d * (1 - d);  gh2 = (g3 * w3 + gfeat / pixels) * (h2 > 0)
    for (int e = threadIdx.x; e < DP_ROWS * 128; e += DP_THREADS) {
        const int r = e >> 7, col = e & 127;
        float v = 0.f;
        if (r < mv) {
            const int m = m0 + r;
            const float ds = dsig[m];
            const float g = (gd ? gd[m] : 0.f) * ds * (1.f - ds);
            if (col == 0) g3[m] = g;
            v = g * w3[col];
            if (gfeat) v += gfeat[(long long)(m / pix_per_roi) * 128 + col] * inv;
            v = h2[(long long)m * 128 + col] > 0.f ? v : 0.f;
            gh2[(long long)m * 128 + col] = v;
        }
        dp_img_store(img2, r, col, v);
    }
    __syncthreads();
    dp_layer<128, 512, true>(nullptr, mv, img2, w2t, stA, stB, [&](int row, int col, float v) {
        const bool live = row < mv;
        if (live) v = h1[(long long)(m0 + row) * 512 + col] > 0.f ? v : 0.f;
        dp_img_store(img1, row, col, live ? v : 0.f);
        if (live) gh1[(long long)(m0 + row) * 512 + col] = v;
    });
    dp_layer<512, 1024, true>(nullptr, mv, img1, w1t, stA, stB, [&](int row, int col, float v) {
        if (row < mv) gx[(long long)(m0 + row) * 1024 + col] = v * neg_lambda;      // gradient reversal (net_utils.py:52-61)
    });
}

// ------------------------------------------------------------------------------------------------
// Small fused pieces of the relation head's tail (resnet_SGG_emb.py:207-215, faster_rcnn_SGG_emb.py:269): these tensors
// are 64 x 300, so every aten op is one launch-bound kernel; F.normalize + its autograd backward are ~9 launches, the
// BCE-with-logits + per-frame mean + weighting ~12.
//
// y = x / max(||x||_2, eps) per row (F.normalize(p=2, dim=1)); one wave per row.
__global__ void __launch_bounds__(256)
l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm, int rows, int cols,
                  float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += xr[c] * xr[c];
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
    const float inv = 1.f / fmaxf(sqrtf(s), eps);
    if (lane == 0) inv_norm[row] = inv;
    for (int c = lane; c < cols; c += 64) y[(long long)row * cols + c] = xr[c] * inv;
}
// gx = (g - y * (g . y)) * inv   (norm > eps; at the clamp the reference's gradient is g / eps, which this also gives
// because inv = 1/eps and the projection term is dropped only when ||x|| <= eps)
__global__ void __launch_bounds__(256)
l2norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ inv_norm,
                  float* __restrict__ gx, int rows, int cols, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* gr = g + (long long)row * cols;
    const float* yr = y + (long long)row * cols;
    float d = 0.f;
    for (int c = lane; c < cols; c += 64) d += gr[c] * yr[c];
    for (int sh = 32; sh > 0; sh >>= 1) d += __shfl_xor(d, sh);
    const float inv = inv_norm[row];
    const bool clamped = inv >= 1.f / eps;
    for (int c = lane; c < cols; c += 64)
        gx[(long long)row * cols + c] = clamped ? gr[c] * inv : (gr[c] - yr[c] * d) * inv;
}

// loss = sum_r w[r] * mean_c bce(z[r][c], t[r][c]),  bce = max(z,0) - z*t + log1p(exp(-|z|))   (BCEWithLogitsLoss)
// ONE workgroup: a wave per row, rows dealt round robin, the four per-wave sums folded in a fixed order -- no atomics, no
// clear of the scalar in front of the kernel, the same bits every run (the tensor is ~64 x 62).
__global__ void __launch_bounds__(256)
bce_rows_fwd_kernel(const float* __restrict__ z, const float* __restrict__ t, const float* __restrict__ w,
                    float* __restrict__ loss, int rows, int cols) {
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int row = wave; row < rows; row += 4) {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) {
            const float zz = z[(long long)row * cols + c], tt = t[(long long)row * cols + c];
            s += fmaxf(zz, 0.f) - zz * tt + log1pf(expf(-fabsf(zz)));
        }
        for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
        acc += s / (float)cols * w[row];
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = ((part[0] + part[1]) + part[2]) + part[3];
}
__global__ void bce_rows_bwd_kernel(const float* __restrict__ z, const float* __restrict__ t, const float* __restrict__ w,
                                    const float* __restrict__ gloss, float* __restrict__ gz, int rows, int cols) {
    const long long n = (long long)rows * cols;
    const float gl = gloss[0];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / cols);
        const float zz = z[i];
        const float sg = 1.f / (1.f + expf(-zz));
        gz[i] = (sg - t[i]) * (w[row] / (float)cols) * gl;
    }
}

// Subject / object rows of the relation pairs (resnet_SGG_emb.py:170-176: index_select twice, cat): out[p] = [obj[ixs[p]] |
// obj[ixo[p]]].  Backward as a GATHER over the pairs, one wave per box row: gobj[b] = sum_{p: ixs[p] == b} g[p][:E] +
// sum_{p: ixo[p] == b} g[p][E:] in pair order -- no atomics, no clear, the same bits every run (tens of pairs per frame).
__global__ void __launch_bounds__(256)
pair_gather_fwd_kernel(const float* __restrict__ obj, const long long* __restrict__ ixs, const long long* __restrict__ ixo,
                       float* __restrict__ out, int n_pairs, int n_box, int E) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= n_pairs) return;
    const long long s = ixs[p], o = ixo[p];
    const bool sok = s >= 0 && s < n_box, ook = o >= 0 && o < n_box;
    for (int c = lane; c < E; c += 64) {
        out[(long long)p * 2 * E + c] = sok ? obj[s * E + c] : 0.f;
        out[(long long)p * 2 * E + E + c] = ook ? obj[o * E + c] : 0.f;
    }
}
__global__ void __launch_bounds__(256)
pair_gather_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ ixs, const long long* __restrict__ ixo,
                       float* __restrict__ gobj, int n_pairs, int n_box, int E) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= n_box) return;
    for (int c0 = 0; c0 < E; c0 += 64 * 4) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < n_pairs; ++p) {
            const bool hs = ixs[p] == b, ho = ixo[p] == b;          // uniform per wave
            if (!hs && !ho) continue;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 64 + lane;
                if (c < E) a[u] += (hs ? g[(long long)p * 2 * E + c] : 0.f) + (ho ? g[(long long)p * 2 * E + E + c] : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * 64 + lane;
            if (c < E) gobj[(long long)b * E + c] = a[u];
        }
    }
}

// wt[k][n] = w[n][k]
__global__ void dp_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int N, int K) {
    const long long total = (long long)N * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N), k = (int)(i / N);
        wt[i] = w[(long long)n * K + k];
    }
}

}  // namespace

extern "C" int32_t i2v_dstyle_pool_fwd(const float* x1, const float* x2, float* z, int64_t rows, int32_t n_img,
                                       int32_t dim, int32_t rank, void* stream) {
    I2V_CHECK_ARG(x1 && x2 && z && rows > 0 && n_img > 0 && dim > 0 && rank > 0, "dstyle_pool_fwd: bad argument");
    const int N = dim * rank;
    I2V_CHECK_ARG(N % 4 == 0 && N / 4 <= 1024, "dstyle_pool_fwd: dim*rank must be a multiple of 4 and <= 4096");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(z, 0, sizeof(float) * (size_t)n_img * dim, st);
    dstyle_pool_fwd_kernel<<<dim3(i2v_cdiv(rows, POOL_ROWS), n_img), N / 4, (size_t)N * 4, st>>>(x1, x2, z, rows, dim,
                                                                                                   rank);
    I2V_CHECK_LAUNCH("dstyle_pool_fwd");
    return I2V_OK;
}

extern "C" int32_t i2v_dstyle_pool_bwd(const float* gz, const float* x1, const float* x2, float* g1, float* g2,
                                       int64_t rows, int32_t n_img, int32_t dim, int32_t rank, void* stream) {
    I2V_CHECK_ARG(gz && x1 && x2 && g1 && g2 && rows > 0 && n_img > 0 && dim > 0 && rank > 0,
                  "dstyle_pool_bwd: bad argument");
    const long long total = (long long)n_img * rows * dim * rank;
    dstyle_pool_bwd_kernel<<<(int)fmin((double)i2v_cdiv(total, 256), 16384.0), 256, 0, (hipStream_t)stream>>>(
        gz, x1, x2, g1, g2, rows, n_img, dim, rank);
    I2V_CHECK_LAUNCH("dstyle_pool_bwd");
    return I2V_OK;
}

// ---- netD_pixel, fused ------------------------------------------------------------------------------
extern "C" int32_t i2v_dpixel_fwd(const float* x, const float* w1, const float* w2, const float* w3, float* h1, float* h2,
                                  float* d, float* feat, int32_t M, int32_t pix_per_roi, void* stream) {
    I2V_CHECK_ARG(x && w1 && w2 && w3 && h1 && h2 && d && M >= 0 && pix_per_roi > 0, "dpixel_fwd: bad argument");
    I2V_CHECK_ARG(!feat || M % pix_per_roi == 0, "dpixel_fwd: M must be a whole number of ROIs when feat is requested");
    if (M == 0) return I2V_OK;
    hipStream_t st = (hipStream_t)stream;
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)dpixel_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_FLOATS * 4);
        (void)hipFuncSetAttribute((const void*)dpixel_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_FLOATS * 4);
        return true;
    }();
    (void)attr;
    if (feat) hipMemsetAsync(feat, 0, sizeof(float) * (size_t)(M / pix_per_roi) * 128, st);
    dpixel_fwd_kernel<<<i2v_cdiv(M, DP_ROWS), DP_THREADS, DP_LDS_FLOATS * 4, st>>>(x, w1, w2, w3, h1, h2, d, feat, M,
                                                                                 pix_per_roi);
    I2V_CHECK_LAUNCH("dpixel_fwd");
    return I2V_OK;
}

extern "C" size_t i2v_dpixel_bwd_workspace_bytes(void) { return sizeof(float) * (size_t)(1024 * 512 + 512 * 128); }

extern "C" int32_t i2v_dpixel_bwd(const float* gd, const float* gfeat, const float* d, const float* h1, const float* h2,
                                  const float* w1, const float* w2, const float* w3, float* g3, float* gh2, float* gh1,
                                  float* gx, int32_t M, int32_t pix_per_roi, float lambda, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    I2V_CHECK_ARG((gd || gfeat) && d && h1 && h2 && w1 && w2 && w3 && g3 && gh2 && gh1 && gx && M >= 0 && pix_per_roi > 0,
                  "dpixel_bwd: bad argument");
    if (!workspace || workspace_bytes < i2v_dpixel_bwd_workspace_bytes()) {
        i2v_set_error("dpixel_bwd: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    if (M == 0) return I2V_OK;
    hipStream_t st = (hipStream_t)stream;
    float* w1t = (float*)workspace;
    float* w2t = w1t + 1024 * 512;
    dp_transpose_kernel<<<512, 256, 0, st>>>(w1, w1t, 512, 1024);
    dp_transpose_kernel<<<128, 256, 0, st>>>(w2, w2t, 128, 512);
    static bool attr = [] {
        (void)hipFuncSetAttribute((const void*)dpixel_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_FLOATS * 4);
        (void)hipFuncSetAttribute((const void*)dpixel_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS_FLOATS * 4);
        return true;
    }();
    (void)attr;
    dpixel_bwd_kernel<<<i2v_cdiv(M, DP_ROWS), DP_THREADS, DP_LDS_FLOATS * 4, st>>>(gd, gfeat, d, h1, h2, w1t, w2t, w3, g3,
                                                                                 gh2, gh1, gx, M, pix_per_roi, -lambda);
    I2V_CHECK_LAUNCH("dpixel_bwd");
    return I2V_OK;
}

// ---- small fused pieces of the relation head's tail ------------------------------------------------
extern "C" int32_t i2v_l2norm_rows_fwd(const float* x, float* y, float* inv_norm, int32_t rows, int32_t cols, float eps,
                                       void* stream) {
    I2V_CHECK_ARG(x && y && inv_norm && rows >= 0 && cols > 0 && eps > 0.f, "l2norm_rows_fwd: bad argument");
    if (rows == 0) return I2V_OK;
    l2norm_fwd_kernel<<<i2v_cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(x, y, inv_norm, rows, cols, eps);
    I2V_CHECK_LAUNCH("l2norm_rows_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_l2norm_rows_bwd(const float* g, const float* y, const float* inv_norm, float* gx, int32_t rows,
                                       int32_t cols, float eps, void* stream) {
    I2V_CHECK_ARG(g && y && inv_norm && gx && rows >= 0 && cols > 0 && eps > 0.f, "l2norm_rows_bwd: bad argument");
    if (rows == 0) return I2V_OK;
    l2norm_bwd_kernel<<<i2v_cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(g, y, inv_norm, gx, rows, cols, eps);
    I2V_CHECK_LAUNCH("l2norm_rows_bwd");
    return I2V_OK;
}
extern "C" int32_t i2v_bce_rows_fwd(const float* z, const float* t, const float* w, float* loss, int32_t rows,
                                    int32_t cols, void* stream) {
    I2V_CHECK_ARG(z && t && w && loss && rows >= 0 && cols > 0, "bce_rows_fwd: bad argument");
    bce_rows_fwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(z, t, w, loss, rows, cols);
    I2V_CHECK_LAUNCH("bce_rows_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_bce_rows_bwd(const float* z, const float* t, const float* w, const float* gloss, float* gz,
                                    int32_t rows, int32_t cols, void* stream) {
    I2V_CHECK_ARG(z && t && w && gloss && gz && rows >= 0 && cols > 0, "bce_rows_bwd: bad argument");
    if (rows == 0) return I2V_OK;
    const long long n = (long long)rows * cols;
    bce_rows_bwd_kernel<<<(int)fmin((double)i2v_cdiv(n, 256), 4096.0), 256, 0, (hipStream_t)stream>>>(z, t, w, gloss, gz,
                                                                                                   rows, cols);
    I2V_CHECK_LAUNCH("bce_rows_bwd");
    return I2V_OK;
}

extern "C" int32_t i2v_pair_gather_fwd(const float* obj, const int64_t* ixs, const int64_t* ixo, float* out, int32_t n_pairs,
                                       int32_t n_box, int32_t emb, void* stream) {
    I2V_CHECK_ARG(obj && ixs && ixo && out && n_pairs >= 0 && n_box > 0 && emb > 0, "pair_gather_fwd: bad argument");
    if (n_pairs == 0) return I2V_OK;
    pair_gather_fwd_kernel<<<i2v_cdiv(n_pairs, 4), 256, 0, (hipStream_t)stream>>>(obj, (const long long*)ixs, (const long long*)ixo,
                                                                                 out, n_pairs, n_box, emb);
    I2V_CHECK_LAUNCH("pair_gather_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_pair_gather_bwd(const float* g, const int64_t* ixs, const int64_t* ixo, float* gobj, int32_t n_pairs,
                                       int32_t n_box, int32_t emb, void* stream) {
    I2V_CHECK_ARG(g && ixs && ixo && gobj && n_pairs >= 0 && n_box > 0 && emb > 0, "pair_gather_bwd: bad argument");
    pair_gather_bwd_kernel<<<i2v_cdiv(n_box, 4), 256, 0, (hipStream_t)stream>>>(g, (const long long*)ixs, (const long long*)ixo,
                                                                               gobj, n_pairs, n_box, emb);
    I2V_CHECK_LAUNCH("pair_gather_bwd");
    return I2V_OK;
}

// ------------------------------------------------------------------------------------------------
// The small arithmetic of the detector's losses and target layers as one kernel per direction (round 3).  Each of these is
// 6-25 aten launches on tensors of a few thousand elements (trainval_net_instance_styleD_bilinear.py:276-296,
// net_utils.py:122-136, bbox_transform.py:36-75, resnet_instance_styleD_bilinear.py:137-139): launch-bound.  Reductions run in
// ONE workgroup with a fixed summation order: no atomics, no clear in front, the same bits every run.
namespace {

__device__ inline float block_sum_1024(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;                                   // valid in thread 0
}

// out = 0.5 * mean((d - target)^2): 0.5*mean(d^2) (target 0) and 0.5*mean((1-d)^2) (target 1) of :276-296
__global__ void __launch_bounds__(1024) half_mse_fwd_kernel(const float* __restrict__ d, long long n, float target, float* __restrict__ out) {
    __shared__ float sh[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) { const float e = d[i] - target; s += e * e; }
    const float t = block_sum_1024(s, sh);
    if (threadIdx.x == 0) out[0] = 0.5f * (t / (float)n);
}
__global__ void half_mse_bwd_kernel(const float* __restrict__ d, long long n, float target, const float* __restrict__ gout, float* __restrict__ gd) {
    const float k = gout[0] / (float)n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) gd[i] = k * (d[i] - target);
}

// net_utils.py:122-136: d = inw * (pred - tgt); l = outw * (|d| < 1/s2 ? 0.5 s2 d^2 : |d| - 0.5/s2); sum over everything but the
// batch axis, mean over the batch = total / rows.  inw / outw hold one weight per ``wdiv`` consecutive elements.
__global__ void __launch_bounds__(1024)
smooth_l1_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ inw,
                     const float* __restrict__ outw, long long n, int wdiv, float s2, float inv_rows, float* __restrict__ out) {
    __shared__ float sh[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const long long wi = i / wdiv;
        const float dd = inw[wi] * (pred[i] - tgt[i]), ad = fabsf(dd);
        s += outw[wi] * (ad < 1.f / s2 ? dd * dd * (s2 * 0.5f) : ad - 0.5f / s2);
    }
    const float t = block_sum_1024(s, sh);
    if (threadIdx.x == 0) out[0] = t * inv_rows;
}
__global__ void smooth_l1_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const float* __restrict__ inw,
                                     const float* __restrict__ outw, long long n, int wdiv, float s2, float inv_rows,
                                     const float* __restrict__ gout, float* __restrict__ gpred) {
    const float k = gout[0] * inv_rows;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long wi = i / wdiv;
        const float w = inw[wi], dd = w * (pred[i] - tgt[i]), ad = fabsf(dd);
        const float dl = ad < 1.f / s2 ? s2 * dd : (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
        gpred[i] = k * outw[wi] * w * dl;
    }
}

// bbox_transform.py:36-75: regression targets of (B,N,4) gt boxes against (N,4) or (B,N,4) example boxes, optionally
// normalised (t - mean) / std (proposal_target_layer_cascade.py:104-106)
__global__ void bbox_transform_kernel(const float* __restrict__ ex, int ex_batched, const float* __restrict__ gt, int gt_stride,
                                      float* __restrict__ out, long long rows, int N, float4 mean, float4 stdv, int normalize) {
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        const float* e = ex + (ex_batched ? r : r % N) * 4;
        const float* g = gt + r * gt_stride;
        const float ew = e[2] - e[0] + 1.f, eh = e[3] - e[1] + 1.f, ecx = e[0] + 0.5f * ew, ecy = e[1] + 0.5f * eh;
        const float gw = g[2] - g[0] + 1.f, gh = g[3] - g[1] + 1.f, gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
        float4 t = make_float4((gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh));
        if (normalize) { t.x = (t.x - mean.x) / stdv.x; t.y = (t.y - mean.y) / stdv.y; t.z = (t.z - mean.z) / stdv.z; t.w = (t.w - mean.w) / stdv.w; }
        *(float4*)(out + r * 4) = t;
    }
}

// resnet_instance_styleD_bilinear.py:137: sqrt(relu(z)) - sqrt(relu(-z)) = sign(z) sqrt(|z|); backward g * 0.5 / sqrt(|z|), 0 at z = 0
__global__ void signed_sqrt_fwd_kernel(const float* __restrict__ z, float* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = z[i];
        y[i] = v > 0.f ? sqrtf(v) : (v < 0.f ? -sqrtf(-v) : 0.f);
    }
}
__global__ void signed_sqrt_bwd_kernel(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ gz, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = fabsf(z[i]);
        gz[i] = v > 0.f ? g[i] * (0.5f / sqrtf(v)) : 0.f;
    }
}

inline int ew_grid(long long n) { return (int)fmin((double)i2v_cdiv(n, 256), 2048.0); }

}  // namespace

extern "C" int32_t i2v_half_mse_fwd(const float* d, int64_t n, float target, float* out, void* stream) {
    I2V_CHECK_ARG(d && out && n > 0, "half_mse_fwd: bad argument");
    half_mse_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(d, n, target, out);
    I2V_CHECK_LAUNCH("half_mse_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_half_mse_bwd(const float* d, int64_t n, float target, const float* gout, float* gd, void* stream) {
    I2V_CHECK_ARG(d && gout && gd && n > 0, "half_mse_bwd: bad argument");
    half_mse_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(d, n, target, gout, gd);
    I2V_CHECK_LAUNCH("half_mse_bwd");
    return I2V_OK;
}
extern "C" int32_t i2v_smooth_l1_fwd(const float* pred, const float* tgt, const float* inw, const float* outw, int64_t n,
                                     int32_t per_weight, int32_t rows, float sigma, float* out, void* stream) {
    I2V_CHECK_ARG(pred && tgt && inw && outw && out && n > 0 && per_weight > 0 && rows > 0 && sigma > 0.f, "smooth_l1_fwd: bad argument");
    smooth_l1_fwd_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(pred, tgt, inw, outw, n, per_weight, sigma * sigma, 1.f / (float)rows, out);
    I2V_CHECK_LAUNCH("smooth_l1_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_smooth_l1_bwd(const float* pred, const float* tgt, const float* inw, const float* outw, int64_t n,
                                     int32_t per_weight, int32_t rows, float sigma, const float* gout, float* gpred, void* stream) {
    I2V_CHECK_ARG(pred && tgt && inw && outw && gout && gpred && n > 0 && per_weight > 0 && rows > 0 && sigma > 0.f, "smooth_l1_bwd: bad argument");
    smooth_l1_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(pred, tgt, inw, outw, n, per_weight, sigma * sigma,
                                                                     1.f / (float)rows, gout, gpred);
    I2V_CHECK_LAUNCH("smooth_l1_bwd");
    return I2V_OK;
}
extern "C" int32_t i2v_bbox_transform(const float* ex, int32_t ex_batched, const float* gt, int32_t gt_stride, float* out, int32_t B,
                                      int32_t N, const float* means4, const float* stds4, void* stream) {
    I2V_CHECK_ARG(ex && gt && out && B > 0 && N > 0 && gt_stride >= 4, "bbox_transform: bad argument");
    const int norm = means4 && stds4;
    const float4 m = norm ? make_float4(means4[0], means4[1], means4[2], means4[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 sd = norm ? make_float4(stds4[0], stds4[1], stds4[2], stds4[3]) : make_float4(1.f, 1.f, 1.f, 1.f);
    const long long rows = (long long)B * N;
    bbox_transform_kernel<<<ew_grid(rows), 256, 0, (hipStream_t)stream>>>(ex, ex_batched, gt, gt_stride, out, rows, N, m, sd, norm);
    I2V_CHECK_LAUNCH("bbox_transform");
    return I2V_OK;
}
extern "C" int32_t i2v_signed_sqrt_fwd(const float* z, float* y, int64_t n, void* stream) {
    I2V_CHECK_ARG(z && y && n > 0, "signed_sqrt_fwd: bad argument");
    signed_sqrt_fwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(z, y, n);
    I2V_CHECK_LAUNCH("signed_sqrt_fwd");
    return I2V_OK;
}
extern "C" int32_t i2v_signed_sqrt_bwd(const float* z, const float* g, float* gz, int64_t n, void* stream) {
    I2V_CHECK_ARG(z && g && gz && n > 0, "signed_sqrt_bwd: bad argument");
    signed_sqrt_bwd_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(z, g, gz, n);
    I2V_CHECK_LAUNCH("signed_sqrt_bwd");
    return I2V_OK;
}
