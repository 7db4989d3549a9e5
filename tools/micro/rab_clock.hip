// Where a workgroup of the RoIAlign backward spends its time: the kernels of csrc/roi_ops.hip compiled with phase stamps
// (RAB_CLOCKS), 4 frames x 32 ROIs x 1024 channels on a 38 x 63 map (configs[2]).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/micro/rab_clock.hip -o tools/micro/rab_clock
//   tools/micro/rab_clock [frames] [form: 1 = round 6 (waves that never meet), 0 = round 5]
#define RAB_CLOCKS 1
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../i2vsgg_amd/csrc/roi_ops.hip"

void i2v_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
int g_i2v_tuning[64];

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, R = 32 * B, C = 1024, H = 38, W = 63, PH = 7, PW = 7;
    const int form = argc > 2 ? atoi(argv[2]) : 1;
    g_i2v_tuning[I2V_TUNE_ROIALIGN_BWD] = form;
    std::vector<float> rois(5 * R), g((size_t)R * PH * PW * C);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int r = 0; r < R; ++r) {
        float x1 = rnd() * (1000 - 33), y1 = rnd() * (600 - 33), w = 32 + rnd() * 368, h = 32 + rnd() * 368;
        rois[5 * r] = (float)(r / 32); rois[5 * r + 1] = x1; rois[5 * r + 2] = y1;
        rois[5 * r + 3] = std::min(x1 + w, 999.f); rois[5 * r + 4] = std::min(y1 + h, 599.f);
    }
    for (auto& v : g) v = rnd() - 0.5f;
    float *d_rois, *d_g, *d_o;
    hipMalloc(&d_rois, rois.size() * 4); hipMalloc(&d_g, g.size() * 4); hipMalloc(&d_o, (size_t)B * H * W * C * 4);
    hipMemcpy(d_rois, rois.data(), rois.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_g, g.data(), g.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) i2v_roi_align_bwd_gather(d_g, d_rois, R, PH, PW, 1 / 16.f, 1, d_o, B, C, H, W, nullptr, 0, nullptr);
    hipEventRecord(e0);
    const int N = 20;
    for (int i = 0; i < N; ++i) i2v_roi_align_bwd_gather(d_g, d_rois, R, PH, PW, 1 / 16.f, 1, d_o, B, C, H, W, nullptr, 0, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nwg = B * H * (C / 128);
    std::vector<unsigned long long> clk((size_t)8192 * 6);
    hipMemcpyFromSymbol(clk.data(), HIP_SYMBOL(g_rab_clk), clk.size() * 8);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    std::vector<unsigned long long> tot;
    for (int i = 0; i < std::min(nwg, 8192); ++i) { for (int k = 0; k < 6; ++k) sum[k] += clk[(size_t)i * 6 + k]; tot.push_back(clk[(size_t)i * 6]); }
    std::sort(tot.begin(), tot.end());
    printf("%d frames: %.2f us per launch (with stamps), %d workgroups\n", B, 1e3 * ms / N, nwg);
    printf("form %d, per workgroup (s_memtime ticks, 100 MHz => x21 for 2.1 GHz cycles): total %.0f (median %llu, max %llu)  list %.0f  %s %.0f  %s %.0f  add %.0f  pairs %.1f\n",
           form, sum[0] / nwg, tot[tot.size() / 2], tot.back(), sum[1] / nwg, form ? "records+barrier" : "stage+barrier", sum[2] / nwg,
           form ? "store" : "request", sum[3] / nwg, sum[4] / nwg, sum[5] / nwg);
    return 0;
}
