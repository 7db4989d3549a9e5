// Image front-end of the data layer on the device.
//
// Replaces, for one image, the host work of roi_data_layer/minibatch.py:60-90 + model/utils/blob.py:35-52:
//   RGB -> BGR (minibatch.py:74), optional horizontal flip (:76-77), float32 conversion and PIXEL_MEANS
//   subtraction (blob.py:38-39), cv2.resize(fx = fy = target / shorter side, INTER_LINEAR) (:49-50), and the
//   zero-padded placement into the batch blob (blob.py:19-33).
// Input is the decoded uint8 image as it comes from the file (H x W x 3): a quarter of the float blob's bytes
// cross PCIe.  Output is the NHWC blob with FOUR channels (the fourth is zero): the stem convolution wants
// Cin % 4 == 0, so the channel pad the backbone otherwise does on every step is folded in here.
//
// cv2's float INTER_LINEAR, restated (cv2 is not in this image, so this follows its documented algorithm; parity
// with cv2 itself is unpinned): dsize = round-half-even(src * f); source position (dx + 0.5) / f - 0.5 evaluated in
// double and rounded to float; s = floor(pos), a = pos - s; s < 0 -> (0, a = 0); s >= n-1 -> (n-1, a = 0);
// horizontal pass first, v = p[s]*(1-a) + p[s+1]*a, then the same vertically; one fp32 rounding per operation.
#include "common.h"

namespace {

__device__ inline void axis_coef(int d, double inv_f, int n, int& s, float& a) {
    float pos = (float)(((double)d + 0.5) * inv_f - 0.5);
    s = (int)floorf(pos);
    a = pos - (float)s;
    if (s < 0) { s = 0; a = 0.f; }
    if (s >= n - 1) { s = n - 1; a = 0.f; }
}

__global__ void image_prep_kernel(const unsigned char* __restrict__ img, int H, int W, int rgb, int flip, float m0,
                                  float m1, float m2, double inv_f, int Ho, int Wo, float* __restrict__ out,
                                  int out_w) {
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= Wo) return;
    int sx, sy;
    float ax, ay;
    axis_coef(dx, inv_f, W, sx, ax);
    axis_coef(dy, inv_f, H, sy, ay);
    const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
    // file order is RGB when rgb != 0: blob channel c (B,G,R) reads file channel 2-c; a flipped image reads column W-1-x
    const int x0 = flip ? W - 1 - sx : sx, x1 = flip ? W - 1 - sx1 : sx1;
    const float mean[3] = {m0, m1, m2};
    float4 o;
    float* op = &o.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int fc = rgb ? 2 - c : c;
        const float p00 = (float)img[((long long)sy * W + x0) * 3 + fc] - mean[c];
        const float p01 = (float)img[((long long)sy * W + x1) * 3 + fc] - mean[c];
        const float p10 = (float)img[((long long)sy1 * W + x0) * 3 + fc] - mean[c];
        const float p11 = (float)img[((long long)sy1 * W + x1) * 3 + fc] - mean[c];
        const float r0 = p00 * (1.f - ax) + p01 * ax;
        const float r1 = p10 * (1.f - ax) + p11 * ax;
        op[c] = r0 * (1.f - ay) + r1 * ay;
    }
    o.w = 0.f;
    *(float4*)(out + ((long long)dy * out_w + dx) * 4) = o;
}

}  // namespace

extern "C" int32_t i2v_image_prep_size(int32_t H, int32_t W, int32_t target_size, int32_t* Ho, int32_t* Wo, float* scale) {
    I2V_CHECK_ARG(H > 0 && W > 0 && target_size > 0 && Ho && Wo && scale, "image_prep_size: bad argument");
    const double f = (double)target_size / (double)(H < W ? H : W);    // blob.py:44: float(target)/float(min), a Python double
    *scale = (float)f;
    *Ho = (int32_t)nearbyint((double)H * f);      // cv2: dsize = saturate_cast<int>(ssize * f), round half to even
    *Wo = (int32_t)nearbyint((double)W * f);
    return I2V_OK;
}

extern "C" int32_t i2v_image_prep(const uint8_t* img, int32_t H, int32_t W, int32_t rgb_order, int32_t flipped,
                                  const float* pixel_means_bgr, int32_t target_size, float* blob, int32_t blob_h,
                                  int32_t blob_w, void* stream) {
    I2V_CHECK_ARG(img && pixel_means_bgr && blob && H > 0 && W > 0 && target_size > 0, "image_prep: bad argument");
    int32_t Ho, Wo;
    float scale;
    i2v_image_prep_size(H, W, target_size, &Ho, &Wo, &scale);
    I2V_CHECK_ARG(Ho <= blob_h && Wo <= blob_w, "image_prep: the blob is smaller than the resized image");
    const double f = (double)target_size / (double)(H < W ? H : W);
    image_prep_kernel<<<dim3(i2v_cdiv(Wo, 256), Ho), 256, 0, (hipStream_t)stream>>>(
        img, H, W, rgb_order, flipped, pixel_means_bgr[0], pixel_means_bgr[1], pixel_means_bgr[2], 1.0 / f, Ho, Wo, blob,
        blob_w);
    I2V_CHECK_LAUNCH("image_prep");
    return I2V_OK;
}
