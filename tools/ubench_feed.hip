// Microbenchmark (round 4): how fast can ONE CU pull L2-resident operand bytes, by path?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), DEPTH pieces in flight per wave, counted vmcnt
//   mode 1: global_load_dwordx4 -> VGPR, DEPTH loads in flight per wave (xor-reduced so the loads stay live)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128 into an LDS ring (register-staged fill)
// Every workgroup streams `kb_per_wg` KiB per pass from a region shared by `share` workgroups of its XCD, `passes` times.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_feed tools/ubench_feed.hip && tools/ubench_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void feed(const uint4* __restrict__ buf, unsigned* __restrict__ out, int pieces_per_wave, int passes,
                                            int share, int region_pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    // region of this workgroup: shared by `share` consecutive slots of its XCD
    const size_t region = ((size_t)(slot / share) * 8 + xcd) * region_pieces;   // in 1 KiB pieces
    const uint4* base = buf + region * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < passes; ++p) {
        if constexpr (MODE == 0) {
            unsigned char* dst = smem + w * (DEPTH * 1024);
            int i = 0;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)((w + (i + d) * nw) % region_pieces) * 64), (lptr_t)(dst + d * 1024), 16, 0, 0);
            for (i = DEPTH; i + DEPTH <= pieces_per_wave; i += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    wait_vm<DEPTH - 1>();
                    __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)((w + (i + d) * nw) % region_pieces) * 64), (lptr_t)(dst + d * 1024), 16, 0, 0);
                }
            }
            wait_vm<0>();
        } else {
            uint4 r[DEPTH];
            int i = 0;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = base[(size_t)((w + (i + d) * nw) % region_pieces) * 64];
            for (i = DEPTH; i + DEPTH <= pieces_per_wave; i += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if constexpr (MODE == 1) {
                        acc[0] ^= r[d].x; acc[1] ^= r[d].y; acc[2] ^= r[d].z; acc[3] ^= r[d].w;
                    } else {
                        *(uint4*)(smem + (w * DEPTH + d) * 1024 + lane * 16) = r[d];
                    }
                    r[d] = base[(size_t)((w + (i + d) * nw) % region_pieces) * 64];
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc[0] ^= r[d].x; acc[1] ^= r[d].y; acc[2] ^= r[d].z; acc[3] ^= r[d].w; }
        }
    }
    if (MODE == 2) acc[0] ^= *(unsigned*)(smem + lane * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int MODE, int DEPTH>
static void run(const char* name, const uint4* buf, unsigned* out, int wgs, int waves, int kb_per_wg, int passes, int share, int region_kb) {
    const int ppw = kb_per_wg / waves;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute((const void*)feed<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((feed<MODE, DEPTH>), dim3(wgs), dim3(64 * waves), lds, 0, buf, out, ppw, passes, share, region_kb);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * kb_per_wg * 1024.0 * passes;
    const double cus = wgs < 256 ? wgs : 256;
    printf("%-8s depth %2d wgs %4d waves %d  %4d KiB/wg x %3d passes share %d region %4d KiB: %8.1f us  %7.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU(2.4GHz)\n",
           name, DEPTH, wgs, waves, kb_per_wg, passes, share, region_kb, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / cus,
           bytes / (best * 1e-3) / cus / 2.4e9);
}

int main() {
    const size_t total = 512ull << 20;
    uint4* buf; unsigned* out;
    hipMalloc(&buf, total); hipMalloc(&out, 4);
    hipMemset(buf, 1, total);
    hipDeviceSynchronize();
    // L2-resident regime: per XCD (wgs/8/share) regions of region_kb; 256 wgs, share 4, region 256 KiB -> 8 regions = 2 MiB per XCD
    for (int waves : {4, 8}) {
        for (int wgs : {16, 256, 512}) {
            run<0, 4>("dma", buf, out, wgs, waves, 256, 16, 4, 256);
            run<0, 8>("dma", buf, out, wgs, waves, 256, 16, 4, 256);
            run<0, 16>("dma", buf, out, wgs, waves, 256, 16, 4, 256);
            run<1, 4>("vgpr", buf, out, wgs, waves, 256, 16, 4, 256);
            run<1, 8>("vgpr", buf, out, wgs, waves, 256, 16, 4, 256);
            run<1, 16>("vgpr", buf, out, wgs, waves, 256, 16, 4, 256);
            run<2, 8>("vgpr+lds", buf, out, wgs, waves, 256, 16, 4, 256);
            run<2, 16>("vgpr+lds", buf, out, wgs, waves, 256, 16, 4, 256);
        }
    }
    // one pass only (cold-ish, the decode GEMM regime: every byte touched once per launch): 256 KiB per wg, regions do not repeat
    for (int waves : {4, 8}) {
        run<0, 8>("dma-1p", buf, out, 256, waves, 256, 1, 4, 256);
        run<1, 8>("vgpr-1p", buf, out, 256, waves, 256, 1, 4, 256);
        run<1, 16>("vgpr-1p", buf, out, 256, waves, 256, 1, 4, 256);
        run<1, 16>("vgpr-1p", buf, out, 256, waves, 256, 1, 1, 256);   // nothing shared: 64 MiB streamed from HBM/MALL
        run<0, 16>("dma-1p", buf, out, 256, waves, 256, 1, 1, 256);
    }
    return 0;
}
