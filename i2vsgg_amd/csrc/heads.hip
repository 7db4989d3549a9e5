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
