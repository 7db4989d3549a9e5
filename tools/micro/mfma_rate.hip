// Micro-benchmark: cycles per fp32 MFMA (16x16x4 vs 32x32x2) for one or two waves per SIMD and 1..8 independent
// accumulators.  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
void run(const char* name, F kern, int nacc, int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double per = (double)h[blocks / 2] / ((double)iters * 8 * nacc);
    // per = cycles per MFMA issued by ONE wave; with W waves per SIMD the pipe sees per / W per MFMA
    int waves_per_simd = threads / 256;
    printf("%-10s acc=%d  waves/SIMD=%d  cycles per MFMA per wave %6.1f  -> per-SIMD pipe interval %6.1f\n", name, nacc,
           waves_per_simd, per, per / waves_per_simd);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int threads : {256, 512}) {
        run("16x16x4", k16<1>, 1, threads, out, cyc); run("16x16x4", k16<2>, 2, threads, out, cyc);
        run("16x16x4", k16<4>, 4, threads, out, cyc); run("16x16x4", k16<5>, 5, threads, out, cyc);
        run("16x16x4", k16<8>, 8, threads, out, cyc);
        run("32x32x2", k32<1>, 1, threads, out, cyc); run("32x32x2", k32<2>, 2, threads, out, cyc);
        run("32x32x2", k32<4>, 4, threads, out, cyc);
    }
    return 0;
}
