#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    auto q = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    out[l * 4 + 0] = r[0]; out[l * 4 + 1] = r[1]; out[l * 4 + 2] = q[0]; out[l * 4 + 3] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j) { printf("%s: ", j == 0 ? "swap16[0]" : j == 1 ? "swap16[1]" : j == 2 ? "swap32[0]" : "swap32[1]"); for (int l = 0; l < 64; l += 8) printf("%u ", h[l * 4 + j]); printf("\n"); }
    return 0;
}
