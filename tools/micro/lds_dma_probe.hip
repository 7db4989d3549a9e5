#include <hip/hip_runtime.h>
__global__ void k(const float* src, float* dst, unsigned bytes) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // each wave DMA-loads 1 KiB: lane l fetches 16 B at voffset, lands at lds_base + l*16
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, (wave * 64 + (lane ^ 3)) * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[threadIdx.x * 4 + 0] = lds[threadIdx.x * 4 + 0];
    dst[threadIdx.x * 4 + 1] = lds[threadIdx.x * 4 + 1];
    dst[threadIdx.x * 4 + 2] = lds[threadIdx.x * 4 + 2];
    dst[threadIdx.x * 4 + 3] = lds[threadIdx.x * 4 + 3];
}
int main() {
    float *s, *d; hipMalloc(&s, 8192); hipMalloc(&d, 8192);
    float h[2048]; for (int i = 0; i < 2048; ++i) h[i] = i;
    hipMemcpy(s, h, 8192, hipMemcpyHostToDevice);
    k<<<1, 256>>>(s, d, 4096);       // second half of the lanes of waves 1.. partly OOB? bytes = 4096 -> all in range; try 2048 for OOB zeros
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("%g %g %g %g | %g %g\n", h[0], h[4], h[8], h[12], h[16], h[1020]);
    k<<<1, 256>>>(s, d, 2048);
    hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
    printf("oob: %g %g (expect 0 0) in-range %g\n", h[600], h[1020], h[4]);
    return 0;
}
