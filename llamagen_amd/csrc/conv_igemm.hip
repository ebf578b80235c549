// Implicit-GEMM convolution / batched NT-GEMM on MFMA with hi/lo-split bf16 operands.
//
// Replaces, in the VQ decoder (tokenizer/tokenizer_image/vq_model.py): nn.Conv2d 3x3 / 1x1
// (:288-291, 321-324, 134, 167), F.interpolate(nearest, 2x) + conv (:374-378, folded into the
// input indexing), the residual add of ResnetBlock / AttnBlock (:314, 351, fused epilogue) and
// the two torch.bmm of AttnBlock (:337, 346).
//
//   out[b][p][n] = alpha * sum_{tap, c} A[b][p + tap][c] * W[tap][n][c]  (+ bias[n]) (+ res[b][p][n])
//
// A and W arrive as (hi, lo) bf16 plane pairs (x = hi + lo); each product is accumulated as
// hi*hi + hi*lo + lo*hi with `v_mfma_f32_16x16x32_bf16` into fp32 (the lo*lo term, < 2^-16
// relative, is dropped): 3 MFMA passes at the bf16 rate (2.5 PF/3) instead of fp32 MFMA (157 TF).
//
// Tiling: workgroup = 4 waves, BM = WMW*JM*16 pixels x BN = WNW*JN*16 output channels, K-step =
// one tap x 32 input channels.  The pixel tile is 128 consecutive NHWC pixels of one image; the
// nine taps re-read shifted rows through L2.  Global -> registers -> LDS double buffer; the PIXEL tile has
// two register staging sets (its loads for step s+2 are issued before the MFMAs of step s and written to
// LDS after the MFMAs of step s+1), the L2-hot weight tile one (a second full set spills at 128x128); one
// barrier per step.  All staging loads are unconditional (border lanes are zeroed at the LDS write):
// with per-lane conditional loads 70 % of the wave cycles were spent waiting, 37 % without.  LDS
// tiles are [rows][32] bf16 with a 16-byte-slot XOR swizzle that makes both the staging
// `ds_write_b128` and the fragment `ds_read_b128` conflict-free for the 16x16x32 operand layout.
// The weight tile is the MFMA A operand and the pixel tile the B operand, so a lane ends up with
// 4 consecutive output channels of one pixel: 16-byte NHWC stores.
#include "lgen_common.h"
#include "../../include/lgen.h"

struct IgemmArgs {
    const uint16_t* a_hi; const uint16_t* a_lo;  // [B][Hs][Ws][Cin]
    const uint16_t* w_hi; const uint16_t* w_lo;  // [taps][Npad][Cin]
    const float* bias;   // [Cout] or null
    const float* res;    // like out (NHWC) or null
    float* out;
    int H, W, Cin, Cout, Npad, ks, ups, out_nchw;
    int nt;      // streaming (non-temporal) output stores: the fp32 result is far larger than any cache
    int stride;  // 1, or 2 = Downsample (vq_model.py:389-393): input (2H) x (2W), zero pad right/bottom only
    long long a_bstride, w_bstride, o_bstride;
    float alpha;
};

LGEN_DEV int swz(int row, int seg) { return row * 64 + ((seg ^ ((row >> 2) & 2)) << 4); }  // byte offset in a [rows][32]bf16 tile

template <int JN, int JM, int WNW, int DS>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs a) {
    constexpr int WMW = 4 / WNW;
    constexpr int BM = WMW * JM * 16, BN = WNW * JN * 16;
    constexpr int A_IT = (BM * 4 + 255) / 256, B_IT = (BN * 4 + 255) / 256;
    constexpr int STAGE = (BM + BN) * 64 * 2;  // bytes: (pixel tile + weight tile) x (hi, lo)
    constexpr bool A_FULL = A_IT * 256 == BM * 4, B_FULL = B_IT * 256 == BN * 4;  // every thread stages every slot
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wv % WNW, wm = wv / WNW;
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int HW = a.H * a.W;
    const int Hs = a.stride == 2 ? a.H * 2 : (a.H >> a.ups), Ws = a.stride == 2 ? a.W * 2 : (a.W >> a.ups);
    const int Hin = a.stride == 2 ? Hs : a.H, Win = a.stride == 2 ? Ws : a.W;  // bounds of the (virtual) conv input
    const uint16_t* ahi = a.a_hi + (size_t)b * a.a_bstride;
    const uint16_t* alo = a.a_lo + (size_t)b * a.a_bstride;
    const uint16_t* whi = a.w_hi + (size_t)b * a.w_bstride;
    const uint16_t* wlo = a.w_lo + (size_t)b * a.w_bstride;

    // per-thread staging coordinates
    int apy[A_IT], apx[A_IT];
    bool apv[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = t + i * 256;
        const int p = p0 + (idx >> 2);
        apv[i] = (idx < BM * 4) && p < HW;
        apy[i] = p / a.W;
        apx[i] = p - apy[i] * a.W;
    }
    const int kchunks = a.Cin >> 5;
    const int taps = a.ks * a.ks;
    const int nsteps = taps * kchunks;
    const int pad = a.stride == 2 ? 0 : (a.ks >> 1);

    // Two register staging sets: the loads of step s+2 are issued while step s computes and are written
    // to LDS one iteration later, so every global load has two compute phases to land.  All loads are
    // UNCONDITIONAL (clamped to a valid address; border / tail lanes are zeroed when the registers are
    // written to LDS): a per-lane "load or zero" select makes hipcc branch around each load and drain
    // vmcnt per element.
    uint4 s0_ah[A_IT], s0_al[A_IT], s0_bh[B_IT], s0_bl[B_IT], s1_ah[A_IT], s1_al[A_IT], s1_bh[B_IT], s1_bl[B_IT];
    bool s0_ok[A_IT], s1_ok[A_IT];
#define IG_GLOAD_A(step_, P)                                                                            \
    {                                                                                                   \
        const int st_ = (step_) < nsteps ? (step_) : nsteps - 1; /* tail: harmless re-load */           \
        const int tap = st_ / kchunks, kc = st_ - tap * kchunks;                                        \
        const int dy = tap / a.ks - pad, dx = tap % a.ks - pad;                                         \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                              \
            const int idx = t + i * 256;                                                                \
            int sy = apy[i] * a.stride + dy, sx = apx[i] * a.stride + dx;                               \
            P##ok[i] = apv[i] && sy >= 0 && sy < Hin && sx >= 0 && sx < Win;                            \
            sy = P##ok[i] ? sy >> a.ups : 0;                                                            \
            sx = P##ok[i] ? sx >> a.ups : 0;                                                            \
            const size_t off = (((size_t)sy * Ws + sx) * a.Cin + kc * 32 + (idx & 3) * 8);              \
            P##ah[i] = *(const uint4*)(ahi + off);                                                      \
            P##al[i] = *(const uint4*)(alo + off);                                                      \
        }                                                                                               \
    }
#define IG_GLOAD_B(step_, P)                                                                            \
    {                                                                                                   \
        const int st_ = (step_) < nsteps ? (step_) : nsteps - 1;                                        \
        const int tap = st_ / kchunks, kc = st_ - tap * kchunks;                                        \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                              \
            const int idx = t + i * 256;                                                                \
            const int row = idx < BN * 4 ? (idx >> 2) : 0;                                              \
            const size_t off = (((size_t)tap * a.Npad + n0 + row) * a.Cin + kc * 32 + (idx & 3) * 8);   \
            P##bh[i] = *(const uint4*)(whi + off);                                                      \
            P##bl[i] = *(const uint4*)(wlo + off);                                                      \
        }                                                                                               \
    }
#define IG_LSTORE_A(buf_, P)                                                                            \
    {                                                                                                   \
        unsigned char* base = smem + (buf_) * STAGE;                                                    \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                              \
            const int idx = t + i * 256;                                                                \
            if (A_FULL || idx < BM * 4) {                                                               \
                const int o = swz(idx >> 2, idx & 3);                                                   \
                const unsigned m_ = P##ok[i] ? 0xffffffffu : 0u; /* scalar mask: no lvalue select */   \
                *(uint4*)(base + o) = make_uint4(P##ah[i].x & m_, P##ah[i].y & m_, P##ah[i].z & m_, P##ah[i].w & m_); \
                *(uint4*)(base + BM * 64 + o) = make_uint4(P##al[i].x & m_, P##al[i].y & m_, P##al[i].z & m_, P##al[i].w & m_); \
            }                                                                                           \
        }                                                                                               \
    }
#define IG_LSTORE_B(buf_, P)                                                                            \
    {                                                                                                   \
        unsigned char* base = smem + (buf_) * STAGE;                                                    \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                              \
            const int idx = t + i * 256;                                                                \
            if (B_FULL || idx < BN * 4) {                                                               \
                const int o = swz(idx >> 2, idx & 3);                                                   \
                *(uint4*)(base + BM * 128 + o) = P##bh[i];                                              \
                *(uint4*)(base + BM * 128 + BN * 64 + o) = P##bl[i];                                    \
            }                                                                                           \
        }                                                                                               \
    }

    f32x4_t acc[JN][JM];
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < JM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* base = smem + buf * STAGE;
        uint4 phi[JM], plo[JM];
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            const int o = swz((wm * JM + i) * 16 + fr, fg);
            phi[i] = *(const uint4*)(base + o);
            plo[i] = *(const uint4*)(base + BM * 64 + o);
        }
#pragma unroll
        for (int j = 0; j < JN; ++j) {  // weight fragments one n-tile at a time (register pressure)
            const int o = swz((wn * JN + j) * 16 + fr, fg);
            const uint4 wh = *(const uint4*)(base + BM * 128 + o);
            const uint4 wl = *(const uint4*)(base + BM * 128 + BN * 64 + o);
#pragma unroll
            for (int i = 0; i < JM; ++i) {
                acc[j][i] = BF16::mma(wl, phi[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh, plo[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh, phi[i], acc[j][i]);
            }
        }
    };

#define IG_GLOAD(step_, P) { IG_GLOAD_A(step_, P); IG_GLOAD_B(step_, P); }
#define IG_LSTORE(buf_, P) { IG_LSTORE_A(buf_, P); IG_LSTORE_B(buf_, P); }
    if constexpr (DS == 0) {  // one staging set: loads of step s+1 fly during the MFMAs of step s only
        IG_GLOAD(0, s0_);
        IG_LSTORE(0, s0_);
        __syncthreads();
        for (int step = 0; step < nsteps; ++step) {
            IG_GLOAD(step + 1, s0_);
            compute(step & 1);
            IG_LSTORE((step & 1) ^ 1, s0_);
            __syncthreads();
        }
    } else if constexpr (DS == 1) {  // two full staging sets (spills at the 128x128 tile)
        IG_GLOAD(0, s0_);
        IG_GLOAD(1, s1_);
        IG_LSTORE(0, s0_);
        __syncthreads();
        for (int step = 0; step < nsteps; step += 2) {
            IG_GLOAD(step + 2, s0_);   // set 0 (step) is already in LDS buffer 0
            compute(0);
            IG_LSTORE(1, s1_);         // step + 1, requested one iteration ago
            __syncthreads();
            if (step + 1 >= nsteps) break;
            IG_GLOAD(step + 3, s1_);
            compute(1);
            IG_LSTORE(0, s0_);         // step + 2
            __syncthreads();
        }
    } else {  // DS == 2: two sets for the pixel tile (HBM/L2 latency), one for the (L2-hot) weight tile
        IG_GLOAD_A(0, s0_);
        IG_GLOAD_B(0, s0_);
        IG_GLOAD_A(1, s1_);
        IG_LSTORE_A(0, s0_);
        IG_LSTORE_B(0, s0_);
        __syncthreads();
        for (int step = 0; step < nsteps; step += 2) {
            IG_GLOAD_A(step + 2, s0_);
            IG_GLOAD_B(step + 1, s0_);
            compute(0);
            IG_LSTORE_A(1, s1_);
            IG_LSTORE_B(1, s0_);
            __syncthreads();
            if (step + 1 >= nsteps) break;
            IG_GLOAD_A(step + 3, s1_);
            IG_GLOAD_B(step + 2, s0_);
            compute(1);
            IG_LSTORE_A(0, s0_);
            IG_LSTORE_B(0, s0_);
            __syncthreads();
        }
    }
#undef IG_GLOAD
#undef IG_LSTORE
#undef IG_GLOAD_A
#undef IG_GLOAD_B
#undef IG_LSTORE_A
#undef IG_LSTORE_B

    // epilogue: lane holds out channels n = nb + fg*4 + {0..3} of pixel p = pb + fr
    float* outb = a.out + (size_t)b * a.o_bstride;
    const float* resb = a.res ? a.res + (size_t)b * a.o_bstride : nullptr;
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int n = n0 + (wn * JN + j) * 16 + fg * 4;
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            const int p = p0 + (wm * JM + i) * 16 + fr;
            if (p >= HW || n >= a.Cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[j][i][e] * a.alpha;
                if (a.bias && n + e < a.Cout) v[e] += a.bias[n + e];
            }
            if (!a.out_nchw && n + 3 < a.Cout && (a.Cout & 3) == 0) {
                const size_t o = (size_t)p * a.Cout + n;
                if (resb) {
                    const float4 r = *(const float4*)(resb + o);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (a.nt) stg_nt_f4((float4*)(outb + o), v[0], v[1], v[2], v[3]);
                else *(float4*)(outb + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= a.Cout) continue;
                    const size_t o = a.out_nchw ? (size_t)(n + e) * HW + p : (size_t)p * a.Cout + n + e;
                    outb[o] = v[e] + (resb ? resb[o] : 0.f);
                }
            }
        }
    }
}

template <int JN, int JM, int WNW, int DS>
static int launch_igemm(const IgemmArgs& a, int B, hipStream_t st) {
    constexpr int BM = (4 / WNW) * JM * 16, BN = WNW * JN * 16;
    const size_t lds = 2 * (size_t)(BM + BN) * 64 * 2;
    if (a.Npad % BN) return LGEN_ERR_BAD_ARG;
    static bool attr = false;
    if (!attr && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_kernel<JN, JM, WNW, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    dim3 grid((a.H * a.W + BM - 1) / BM, a.Npad / BN, B);
    hipLaunchKernelGGL((igemm_kernel<JN, JM, WNW, DS>), grid, dim3(256), lds, st, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}

int lgen_vq_nt();  // vq_ops.hip
// tuning knob (tools/): 0 = 128x128 tile, one staging set; 1 = 128x128, two staging sets; 2 = 128x64 tiles, two sets;
// 3 = 128x128, two sets for the pixel tile only
static int g_igemm_variant = 3;
extern "C" int lgen_debug_set_igemm_variant(int v) {
    if (v == 1) return LGEN_ERR_UNSUPPORTED;  // double-staged both operands: 144 B of scratch per lane, removed in round 3
    g_igemm_variant = v;
    return 0;
}

extern "C" int lgen_conv_igemm(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                               const float* res, float* out, int B, int H, int W, int Cin, int Cout, int Npad, int ksize,
                               int upsample, int out_nchw, long long w_bstride, float alpha, void* stream) {
    const int stride = upsample == 2 ? 2 : 1;  // upsample: 0 none, 1 nearest-2x input, 2 stride-2 Downsample conv
    if (stride == 2) upsample = 0;
    if (Cin % 32 || (ksize != 1 && ksize != 3) || Npad % 16 || Npad < Cout || (upsample && ((H | W) & 1)) ||
        (stride == 2 && ksize != 3))
        return LGEN_ERR_BAD_ARG;
    if (B == 0 || H * W == 0) return 0;
    IgemmArgs a{(const uint16_t*)a_hi, (const uint16_t*)a_lo, (const uint16_t*)w_hi, (const uint16_t*)w_lo, bias, res, out,
                H, W, Cin, Cout, Npad, ksize, upsample ? 1 : 0, out_nchw, lgen_vq_nt(), stride,
                stride == 2 ? (long long)(2 * H) * (2 * W) * Cin
                            : (long long)(H >> (upsample ? 1 : 0)) * (W >> (upsample ? 1 : 0)) * Cin,
                w_bstride, (long long)H * W * Cout, alpha};
    hipStream_t st = (hipStream_t)stream;
    if (g_igemm_variant == 2 && Npad % 64 == 0) return launch_igemm<4, 2, 1, 1>(a, B, st);
    if (Npad % 128 == 0) {                                          // 128 px x 128 ch
        if (g_igemm_variant == 3) return launch_igemm<4, 4, 2, 2>(a, B, st);
        return launch_igemm<4, 4, 2, 0>(a, B, st);
    }
    if (Npad % 64 == 0) return launch_igemm<4, 2, 1, 1>(a, B, st);    // 128 px x 64 ch
    return launch_igemm<1, 2, 1, 1>(a, B, st);                        // 128 px x 16 ch (conv_out, Cout = 3)
}
