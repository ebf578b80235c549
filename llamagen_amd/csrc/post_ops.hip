// Post-processing of decoded samples, the step right after the hot path in the reference's FID sampling
// loop (autoregressive/sample/sample_c2i_ddp.py:141-143):
//   lgen_resize_bicubic   F.interpolate(samples, size=(E, E), mode='bicubic')  (align_corners=False, A=-0.75,
//                         no antialias; 384 -> 256 by default: --image-size-eval 256)
//   lgen_to_uint8_hwc     torch.clamp(127.5 * x + 128.0, 0, 255).permute(0, 2, 3, 1).to(uint8)   (truncating cast)
// Both are HBM-bound elementwise/gather kernels: one thread per output pixel (all channels), coalesced along W.
#include "lgen_common.h"
#include "../../include/lgen.h"

// ATen UpSample.h cubic_convolution1 / cubic_convolution2 with A = -0.75, same operation order
LGEN_DEV float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
LGEN_DEV float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(256) void resize_bicubic_kernel(const float* __restrict__ in, float* __restrict__ out, int BC,
                                                             int Hi, int Wi, int Ho, int Wo, float sy, float sx) {
    const long long total = (long long)BC * Ho * Wo;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(t % Wo);
        const int oy = (int)((t / Wo) % Ho);
        const long long bc = t / ((long long)Wo * Ho);
        // area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=true): no clamp at 0
        const float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy), ix = (int)floorf(fx);
        const float ty = fy - (float)iy, tx = fx - (float)ix;
        const float wx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
        const float wy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
        const float* src = in + bc * (long long)Hi * Wi;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int yy = iy - 1 + i;
            yy = yy < 0 ? 0 : (yy > Hi - 1 ? Hi - 1 : yy);
            float row = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int xx = ix - 1 + j;
                xx = xx < 0 ? 0 : (xx > Wi - 1 ? Wi - 1 : xx);
                row += src[(long long)yy * Wi + xx] * wx[j];
            }
            acc += row * wy[i];
        }
        out[t] = acc;
    }
}

extern "C" int lgen_resize_bicubic(const float* in_nchw, float* out_nchw, int BC, int Hi, int Wi, int Ho, int Wo,
                                   void* stream) {
    if (BC < 0 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return LGEN_ERR_BAD_ARG;
    const long long total = (long long)BC * Ho * Wo;
    if (total == 0) return 0;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in_nchw, out_nchw, BC, Hi, Wi, Ho, Wo,
                       (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    LGEN_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void to_uint8_hwc_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int B,
                                                           int C, int HW) {
    const long long total = (long long)B * HW;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long b = t / HW, p = t - b * HW;
        for (int c = 0; c < C; ++c) {
            float v = 127.5f * in[(b * C + c) * HW + p] + 128.0f;
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[t * C + c] = (unsigned char)v;  // truncation, like .to(torch.uint8) of a non-negative float
        }
    }
}

extern "C" int lgen_to_uint8_hwc(const float* in_nchw, unsigned char* out_nhwc, int B, int C, int H, int W, void* stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return LGEN_ERR_BAD_ARG;
    const long long total = (long long)B * H * W;
    if (total == 0) return 0;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(to_uint8_hwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in_nchw, out_nhwc, B, C, H * W);
    LGEN_CHECK_LAUNCH();
    return 0;
}
