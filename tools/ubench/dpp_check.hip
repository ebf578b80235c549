// checks lgen_common.h's DPP / permlane butterflies against __shfl_xor on the GPU (development aid)
#include "../../llamagen_amd/csrc/lgen_common.h"
#include <cstdio>
#include <vector>
template <int LPK>
__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    float v = in[l];
    float a = group_sum<LPK>(v);
    float b = v;
    for (int o = 1; o < LPK; o <<= 1) b += __shfl_xor(b, o, 64);
    float m1 = across_groups<LPK>(a, [](float x, float y) { return fmaxf(x, y); });
    float m2 = b;
    for (int o = LPK; o < 64; o <<= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
    float s1 = across_groups<LPK>(v, [](float x, float y) { return x + y; });
    float s2 = v;
    for (int o = LPK; o < 64; o <<= 1) s2 += __shfl_xor(s2, o, 64);
    out[l * 6 + 0] = a; out[l * 6 + 1] = b; out[l * 6 + 2] = m1; out[l * 6 + 3] = m2; out[l * 6 + 4] = s1; out[l * 6 + 5] = s2;
}
template <int LPK> int run(const float* din, float* dout) {
    hipLaunchKernelGGL(k<LPK>, dim3(1), dim3(64), 0, 0, din, dout);
    std::vector<float> h(64 * 6);
    hipMemcpy(h.data(), dout, sizeof(float) * 64 * 6, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 3; ++j)
            if (!(h[l * 6 + 2 * j] == h[l * 6 + 2 * j + 1])) { if (bad < 6) printf("LPK %d lane %d kind %d: %g vs %g\n", LPK, l, j, h[l * 6 + 2 * j], h[l * 6 + 2 * j + 1]); ++bad; }
    printf("LPK %d: %d mismatches\n", LPK, bad);
    return bad;
}
int main() {
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 101) * 0.37f - 11.f;
    float *din, *dout;
    hipMalloc(&din, 256); hipMalloc(&dout, 64 * 6 * 4);
    hipMemcpy(din, h.data(), 256, hipMemcpyHostToDevice);
    int bad = run<8>(din, dout) + run<16>(din, dout) + run<32>(din, dout);
    return bad != 0;
}
