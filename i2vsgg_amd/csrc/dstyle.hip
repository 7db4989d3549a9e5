// netD_style (resnet_instance_styleD_bilinear.py:85-146): the factorised bilinear pooling as ONE GEMM-shaped kernel.
//
//   x1 = rows * W1^T + b1,  x2 = rows * W2^T + b2        rows (n_img * P, 512), W (dim*rank, 512)
//   z[b][d] = sum_{pos < P} sum_{r < rank} x1[b,pos,d*rank+r] * x2[b,pos,d*rank+r]
//
// The reference materialises both projections (2 x 96 MB per 600x1000 frame) and reduces the product in three more
// full-size passes.  Here a workgroup owns 64 positions of ONE image and 64 of the dim*rank columns: the activation
// tile is staged once and feeds TWO accumulator sets (W1 and W2 tiles of the same columns), and the epilogue multiplies
// the two tiles in registers and reduces the product over the tile's positions: what leaves the kernel is one partial
// row per (image, position tile) -- 0.6 % of a projection -- which a second, tiny kernel sums over position tiles
// (in a fixed order: no atomics, the result is reproducible) and over the rank groups.
// HBM traffic of the forward: the 19.2 MB tap + 10 MB of weights, instead of + 4 x 96 MB.
//
// STORE = true additionally writes x1 and x2: a training step needs them for the backward (g1 = gz * x2, g2 = gz * x1;
// recomputing them there would cost two more 24.6-GMAC GEMMs), inference and no-grad calls do not.
//
// Tile GEMM: the machinery of conv_gemm_f32 (conv.hip): 32-deep swizzled LDS stages, register prefetch, 16x16x4 fp32
// MFMA with the weights as the row operand, so that a lane's four accumulator registers are four consecutive columns.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int DS_BM = 64, DS_BN = 64, DS_BK = 32, DS_THREADS = 256;

template <bool STORE>
__global__ void __launch_bounds__(DS_THREADS)
dstyle_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                        const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ part,
                        float* __restrict__ x1o, float* __restrict__ x2o, int P, int K, int N, int mtiles) {
    // grid: x = column tile (fastest: the column tiles of one position tile share the activation tile in L2),
    //       y = position tile within the image, z = image
    __shared__ __attribute__((aligned(16))) float As[2][DS_BM * DS_BK];
    __shared__ __attribute__((aligned(16))) float B1s[2][DS_BN * DS_BK];
    __shared__ __attribute__((aligned(16))) float B2s[2][DS_BN * DS_BK];
    __shared__ float colsum[2][DS_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;             // 2 x 2 waves, 32 x 32 outputs each (2 x 2 fragments)
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * DS_BN, p0 = blockIdx.y * DS_BM, img = blockIdx.z;
    const long long row0 = (long long)img * P;
    const int kc = tid & 7, kg = kc * 4;
    constexpr unsigned INV = 0x80000000u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)(x + row0 * K), 0, (unsigned)((long long)P * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc((void*)w1, 0, (unsigned)((long long)N * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2r = __builtin_amdgcn_make_buffer_rsrc((void*)w2, 0, (unsigned)((long long)N * K * 4), 0x00020000);
    unsigned a_v[2], b_v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (tid >> 3) + q * 32;
        a_v[q] = (p0 + row < P) ? ((unsigned)((p0 + row) * K) + (unsigned)kg) * 4u : INV;
        b_v[q] = (n0 + row < N) ? ((unsigned)((n0 + row) * K) + (unsigned)kg) * 4u : INV;
    }
    float4 ra[2], rb1[2], rb2[2];
    auto gload = [&](int k0) {
        const unsigned so = (unsigned)k0 * 4u;
        const unsigned kinv = ~(unsigned)((k0 + kg - K) >> 31) & INV;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, a_v[q] | kinv, so, 0));
            rb1[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w1r, b_v[q] | kinv, so, 0));
            rb2[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w2r, b_v[q] | kinv, so, 0));
        }
    };
    auto sstore = [&](int S) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = (tid >> 3) + q * 32;
            const int o = row * DS_BK + ((kc ^ ((row >> 1) & 7)) << 2);
            *(float4*)&As[S][o] = ra[q];
            *(float4*)&B1s[S][o] = rb1[q];
            *(float4*)&B2s[S][o] = rb2[q];
        }
    };
    gload(0);
    // biases of this lane's columns: n = n0 + (wn*2 + j)*16 + 4*fg .. +3
    float4 bb1[2], bb2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wn * 2 + j) * 16 + 4 * fg;
        bb1[j] = n < N ? *(const float4*)(b1 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bb2[j] = n < N ? *(const float4*)(b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc1[2][2], acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 av[2], u1[2], u2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wm * 2 + i) * 16 + fr;
                av[i] = *(const float4*)&As[buf][row * DS_BK + (((h * 4 + fg) ^ ((row >> 1) & 7)) << 2)];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wn * 2 + j) * 16 + fr;
                const int o = row * DS_BK + (((h * 4 + fg) ^ ((row >> 1) & 7)) << 2);
                u1[j] = *(const float4*)&B1s[buf][o];
                u2[j] = *(const float4*)&B2s[buf][o];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                        const float c1 = t == 0 ? u1[j].x : t == 1 ? u1[j].y : t == 2 ? u1[j].z : u1[j].w;
                        const float c2 = t == 0 ? u2[j].x : t == 1 ? u2[j].y : t == 2 ? u2[j].z : u2[j].w;
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1, a, acc1[i][j], 0, 0, 0);
                        acc2[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(c2, a, acc2[i][j], 0, 0, 0);
                    }
        }
    };
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += DS_BK) {
        const bool more = k0 + DS_BK < K;
        if (more) gload(k0 + DS_BK);
        compute(buf);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: x1 = acc1 + b1, x2 = acc2 + b2 (written when STORE), product summed over the tile's valid positions.
    // A lane holds columns n..n+3 of position p = p0 + (wm*2 + i)*16 + fr for i = 0, 1.
    float4 s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        s[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = n0 + (wn * 2 + j) * 16 + 4 * fg;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pos = p0 + (wm * 2 + i) * 16 + fr;
            const float4 v1 = make_float4(acc1[i][j][0] + bb1[j].x, acc1[i][j][1] + bb1[j].y, acc1[i][j][2] + bb1[j].z, acc1[i][j][3] + bb1[j].w);
            const float4 v2 = make_float4(acc2[i][j][0] + bb2[j].x, acc2[i][j][1] + bb2[j].y, acc2[i][j][2] + bb2[j].z, acc2[i][j][3] + bb2[j].w);
            if (pos < P && n < N) {
                if (STORE) {
                    *(float4*)(x1o + (row0 + pos) * N + n) = v1;
                    *(float4*)(x2o + (row0 + pos) * N + n) = v2;
                }
                s[j].x += v1.x * v2.x; s[j].y += v1.y * v2.y; s[j].z += v1.z * v2.z; s[j].w += v1.w * v2.w;
            }
        }
        // sum over the 16 positions held by the lanes of one 16-lane group (fr): butterfly inside the group
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            s[j].x += __shfl_xor(s[j].x, m, 64); s[j].y += __shfl_xor(s[j].y, m, 64);
            s[j].z += __shfl_xor(s[j].z, m, 64); s[j].w += __shfl_xor(s[j].w, m, 64);
        }
        if (fr == 0) *(float4*)&colsum[wm][(wn * 2 + j) * 16 + 4 * fg] = s[j];
    }
    __syncthreads();
    if (tid < DS_BN && n0 + tid < N)          // the two position halves of the tile, in a fixed order
        part[((long long)img * mtiles + blockIdx.y) * N + n0 + tid] = colsum[0][tid] + colsum[1][tid];
}

// z[b][d] = sum_{tile} sum_{r < rank} part[b][tile][d*rank + r]   (tiles in index order: reproducible)
__global__ void dstyle_reduce_kernel(const float* __restrict__ part, float* __restrict__ z, int mtiles, int dim, int rank) {
    const int img = blockIdx.y, d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    const int N = dim * rank;
    float acc = 0.f;
    for (int t = 0; t < mtiles; ++t) {
        const float* p = part + ((long long)img * mtiles + t) * N + d * rank;
        float u = 0.f;
        for (int r = 0; r < rank; ++r) u += p[r];
        acc += u;
    }
    z[(long long)img * dim + d] = acc;
}

}  // namespace

extern "C" size_t i2v_dstyle_fused_workspace_bytes(int64_t rows, int32_t n_img, int32_t dim, int32_t rank) {
    return i2v_align((size_t)n_img * (size_t)i2v_cdiv(rows, DS_BM) * (size_t)dim * rank * sizeof(float));
}

extern "C" int32_t i2v_dstyle_fused_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                        float* z, float* x1, float* x2, int64_t rows, int32_t n_img, int32_t k,
                                        int32_t dim, int32_t rank, void* workspace, size_t workspace_bytes, void* stream) {
    I2V_CHECK_ARG(x && w1 && b1 && w2 && b2 && z && rows > 0 && n_img > 0 && dim > 0 && rank > 0 && k > 0,
                  "dstyle_fused_fwd: bad argument");
    I2V_CHECK_ARG((x1 == nullptr) == (x2 == nullptr), "dstyle_fused_fwd: x1 and x2 are stored together or not at all");
    const int N = dim * rank;
    I2V_CHECK_ARG(k % 4 == 0 && N % 4 == 0, "dstyle_fused_fwd: k and dim*rank must be multiples of 4");
    I2V_CHECK_ARG(rows * (int64_t)k * 4 < (1ll << 31) && (int64_t)N * k * 4 < (1ll << 31), "dstyle_fused_fwd: operand larger than 2 GiB");
    if (!workspace || workspace_bytes < i2v_dstyle_fused_workspace_bytes(rows, n_img, dim, rank)) {
        i2v_set_error("dstyle_fused_fwd: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int mtiles = i2v_cdiv(rows, DS_BM);
    const dim3 grid(i2v_cdiv(N, DS_BN), mtiles, n_img);
    if (x1) dstyle_fused_fwd_kernel<true><<<grid, DS_THREADS, 0, st>>>(x, w1, b1, w2, b2, (float*)workspace, x1, x2, (int)rows, k, N, mtiles);
    else dstyle_fused_fwd_kernel<false><<<grid, DS_THREADS, 0, st>>>(x, w1, b1, w2, b2, (float*)workspace, nullptr, nullptr, (int)rows, k, N, mtiles);
    dstyle_reduce_kernel<<<dim3(i2v_cdiv(dim, 128), n_img), 128, 0, st>>>((const float*)workspace, z, mtiles, dim, rank);
    I2V_CHECK_LAUNCH("dstyle_fused_fwd");
    return I2V_OK;
}
