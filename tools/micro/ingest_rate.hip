// Micro-benchmark: how many bytes per clock one CU can pull from an L2-resident buffer with
// buffer_load_dwordx4, as a function of the access pattern of a wave-instruction and of the number of
// loads kept in flight.  One 256-thread workgroup per CU.  Build: hipcc --offload-arch=gfx950 -O3
// Pattern: a wave-load covers 8 "rows" of 128 B (8 lanes x 16 B each); rows are `row_stride` bytes apart.
//   row_stride = 128   -> 1 KiB contiguous per wave-load
//   row_stride = 1024  -> NHWC activations with 256 channels (3x3 256->256)
//   row_stride = 4096  -> 1024 channels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int DEPTH>
__global__ void __launch_bounds__(256) ingest(const float* __restrict__ src, unsigned bytes, unsigned row_stride,
                                              unsigned wg_span, int iters, unsigned long long* clk, float* sink) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const unsigned tid = threadIdx.x;
    const unsigned row = tid >> 3, kc = tid & 7;        // 32 rows per 256-thread load
    // each workgroup walks its own window (like one GEMM tile: 80 rows re-visited with a moving k offset)
    unsigned base = (blockIdx.x * wg_span) % (bytes / 2);
    float4 v[DEPTH];
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned k = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        v[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, base + row * row_stride + kc * 16 + k, 0, 0));
        k += 128; if (k >= row_stride) { k = 0; base += 32 * row_stride; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += v[d].x + v[d].w;
            v[d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, base + row * row_stride + kc * 16 + k, 0, 0));
            k += 128; if (k >= row_stride) { k = 0; base += 32 * row_stride; if (base >= bytes / 2) base = 0; }
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += v[d].x;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int DEPTH>
void run(const float* d, unsigned bytes, unsigned stride, int wgs, unsigned long long* dclk, float* sink) {
    const int iters = 400 / DEPTH * 4;
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(ingest<DEPTH>, dim3(wgs), dim3(256), 0, 0, d, bytes, stride, 80 * stride, iters, dclk, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs);
    hipMemcpy(h.data(), dclk, wgs * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double loads = (double)(iters + 1) * DEPTH * 4.0 * 1024.0;   // bytes per workgroup (4 waves x 1 KiB)
    printf("  stride %5u  in flight %2d x 4 KiB  wgs %4d : %6.1f B/clk/CU (median)\n", stride, DEPTH, wgs, loads / (double)h[wgs / 2]);
}

int main() {
    const unsigned bytes = 24u << 20;      // 24 MiB: the windows of 256 WGs together stay L2/MALL resident
    float* d; unsigned long long* dclk; float* sink;
    hipMalloc(&d, bytes); hipMemset(d, 0, bytes);
    hipMalloc(&dclk, 4096 * 8); hipMalloc(&sink, 4);
    for (int wgs : {1, 32, 256, 512}) {
        for (unsigned stride : {128u, 1024u, 4096u}) {
            run<1>(d, bytes, stride, wgs, dclk, sink);
            run<5>(d, bytes, stride, wgs, dclk, sink);
            run<10>(d, bytes, stride, wgs, dclk, sink);
            run<20>(d, bytes, stride, wgs, dclk, sink);
        }
    }
    return 0;
}
