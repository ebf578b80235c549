// Fused VQ-decoder convolution: GroupNorm-apply + swish + hi/lo split on the way INTO the tile, 3x3 / 1x1
// convolution on MFMA from an LDS-resident halo tile, bias + residual + next-GroupNorm statistics on the way OUT.
//
// Replaces, per convolution of the decoder (tokenizer/tokenizer_image/vq_model.py):
//   Normalize (GroupNorm(32, C, 1e-6), :359-364) + nonlinearity (x*sigmoid(x), :354-356) applied to the conv
//   input (ResnetBlock.forward :299-306, Decoder.forward :190-192), nn.Conv2d 3x3 pad 1 / 1x1 (:288-291, :167),
//   F.interpolate(2.0, nearest) in front of Upsample.conv (:374-378), the residual add of ResnetBlock (:314) and
//   the statistics pass of the NEXT GroupNorm (per-(image, group) sum / sum of squares of this conv's output).
//
// Why (round-1 profile): the un-fused form read every conv input three times as fp32 (statistics, apply/split,
// then the (hi, lo) planes again) and staged each pixel tile from L2 once per filter tap.  Here
//   * the input is read ONCE per K-chunk as fp32 (tile + 1-pixel halo), normalised / activated / split in
//     registers and kept in LDS as (hi, lo) bf16 for all nine taps: staging traffic per tap is the weight tile only;
//   * the 3-pass split-bf16 product (hi*hi + hi*lo + lo*hi, fp32 accumulate on v_mfma_f32_16x16x32_bf16) is
//     unchanged: <= 1.3e-4 abs against the fp32 reference decoder (single-pass bf16: 9e-2);
//   * the epilogue reduces per-(tile, 4-channel quad) (sum, M2) of the stored fp32 values, so the consumer's
//     GroupNorm needs no pass over the activation (lgen_gn_finalize turns the partials into per-channel
//     (scale, shift) pairs).
//
// Tile: 8 rows x 16 columns of one image x BN output channels per 256-thread workgroup (4 waves as
// WNW x WMW; each wave JN x JM 16x16 MFMA tiles), two workgroups per CU so that one's HBM-bound prologue /
// epilogue overlaps the other's MFMA phase (at C = 128 these convolutions sit near the HBM/MFMA balance point:
// 7.2 GB of fp32 traffic vs 4.2 TFLOP of split-bf16 MFMA work per 32-image 384 px conv).
// LDS: halo tile [plane][k-slice fg][pixel][8 ch] (pixel-major inside a k-slice slab whose size is a
// multiple of 256 B: the B-operand `ds_read_b128` of 16 consecutive pixels is bank-conflict free for EVERY
// tap shift), weights in MFMA fragment order (lane-linear, conflict free), double-buffered per tap.
#include <cstdint>
#include <type_traits>

#include "lgen_common.h"
#include "../../include/lgen.h"

struct ConvFArgs {
    const float* x;      // [B][Hs][Ws][Cin] fp32 NHWC (Hs = H >> ups)
    const float2* coef;  // [B][Cin] (scale, shift) of the fused GroupNorm, or null
    const uint4* w;      // [Npad/BN][Cin/32][taps][2 planes][BN/16][64 lanes] x 16 B (MFMA A-fragment order)
    const float* bias;   // [Cout] or null
    const float* res;    // NHWC like out, or null
    float* out;          // NHWC [B][H][W][Cout] (or NCHW)
    float* part;         // [B][ntiles][Npad/4][2] partial (sum, sumsq) of the stored output, or null
    int H, W, Cin, Cout, Npad, ups, swish, out_nchw, tiles_x, ntiles;
};

LGEN_DEV void split8(const float (&f)[8], uint4& hi, uint4& lo) {
    float r[8];
    hi = BF16::pack(f);
    float h[8];
    BF16::unpack(hi, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f[e] - h[e];
    lo = BF16::pack(r);
}

template <int N>
LGEN_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DMA = true: the weight tiles go HBM/L2 -> LDS directly (`global_load_lds_dwordx4`, one 1 KiB fragment per wave-instruction; the
// fragment-packed global layout IS the LDS image), issued right after a step's fragment reads and waited for before the step's
// closing barrier: no staging registers, no ds_write pass.
// PIPE (round 6, with DMA, 3x3 only): the LDS fragment reads of a step no longer all sit in front of its MFMAs.  The pixel fragments of tap
// t + 1 are requested while tap t's MFMAs run (the halo tile is static for the nine taps of a 32-channel chunk), the weight
// fragments of a step are waited for pair by pair (counted `lgkmcnt`: LDS operations return in order), so a wave's MFMAs start
// after two fragment reads instead of sixteen.  The reads are inline asm for the reason gemm_tile.hip gives: behind a
// `global_load_lds` hipcc drains vmcnt in front of every LDS read it can see.
template <int OFF>
LGEN_DEV u32x4_t cf_lds_rd(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
LGEN_DEV void cf_touch(u32x4_t& v) { asm volatile("" : "+v"(v)); }
LGEN_DEV uint4 cf_u4(const u32x4_t& v) { return make_uint4(v[0], v[1], v[2], v[3]); }
template <int N>
LGEN_DEV void cf_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <int I, int N, typename F>
LGEN_DEV void cf_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cf_static_for<I + 1, N>(f);
    }
}

template <int JN, int WNW, int KS, int NWV, bool DMA, int PIPE = 0>
__global__ __launch_bounds__(64 * NWV, NWV / 2) void conv_fused_kernel(ConvFArgs a) {
    constexpr int PAD = KS / 2, TH = 8, TW = 16, TAPS = KS * KS, NT = 64 * NWV;  // NT threads, NWV waves (2 workgroups / CU)
    constexpr int WMW = NWV / WNW, JM = TH / WMW;
    constexpr int BN = WNW * JN * 16;
    constexpr int HR = TH + 2 * PAD, HC = TW + 2 * PAD, NP = HR * HC, NPP = (NP + 15) / 16 * 16;
    constexpr int ITER = (NP * 4 + NT - 1) / NT;
    constexpr int FGS = NPP * 16;               // bytes of one k-slice slab
    constexpr int PLANE = 4 * FGS;              // bytes of one (hi | lo) plane
    constexpr int HALO = 2 * PLANE;
    constexpr int WT = 2 * (BN / 16) * 1024;    // bytes of one tap's weight tile (hi + lo fragments)
    constexpr int W_IT = (WT / 16 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALO;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wv % WNW, wm = wv / WNW;
    const int b = blockIdx.z, nb = blockIdx.y;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give each XCD a
    // contiguous run of tiles (neighbouring tiles share halo rows and all share the weights in that XCD's L2)
    int tile = blockIdx.x;
    {
        const int nt = a.ntiles, q = nt >> 3, r = nt & 7, xcd = tile & 7, k = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int y0 = (tile / a.tiles_x) * TH, x0 = (tile % a.tiles_x) * TW;
    const int Hs = a.H >> a.ups, Ws = a.W >> a.ups;
    const int nkc = a.Cin >> 5;
    const int nsteps = nkc * TAPS;
    const float* xb = a.x + (size_t)b * Hs * Ws * a.Cin;

    // per-thread halo staging items: it = t + i*NT -> (pixel P = it >> 2, k-slice fg = it & 3)
    int soff[ITER];
    bool sok[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
        const int it = t + i * NT;
        const int P = (it >> 2) < NP ? (it >> 2) : NP - 1;
        const int hy = P / HC, hx = P - hy * HC;
        const int yy = y0 - PAD + hy, xx = x0 - PAD + hx;
        sok[i] = (it < NP * 4) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        const int sy = sok[i] ? (yy >> a.ups) : 0, sx = sok[i] ? (xx >> a.ups) : 0;
        soff[i] = (sy * Ws + sx) * a.Cin + (it & 3) * 8;
    }
    const int fg_t = t & 3;
    float4 raw[ITER][2];
    float4 cf[4];  // 8 (scale, shift) pairs of this thread's channels in the chunk
#define CF_GLOAD_HALO(kc_)                                                                          \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < ITER; ++i) {                                          \
            const float4* p = (const float4*)(xb + soff[i] + (kc_) * 32);                           \
            raw[i][0] = p[0];                                                                       \
            raw[i][1] = p[1];                                                                       \
        }                                                                                           \
        if (a.coef) {                                                                               \
            const float4* c = (const float4*)(a.coef + (size_t)b * a.Cin + (kc_) * 32 + fg_t * 8);  \
            cf[0] = c[0]; cf[1] = c[1]; cf[2] = c[2]; cf[3] = c[3];                                 \
        }                                                                                           \
    }
    auto store_halo = [&]() {
        const float sc[8] = {cf[0].x, cf[0].z, cf[1].x, cf[1].z, cf[2].x, cf[2].z, cf[3].x, cf[3].z};
        const float sh[8] = {cf[0].y, cf[0].w, cf[1].y, cf[1].w, cf[2].y, cf[2].w, cf[3].y, cf[3].w};
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int it = t + i * NT;
            if (i + 1 < ITER || it < NP * 4) {
                float f[8] = {raw[i][0].x, raw[i][0].y, raw[i][0].z, raw[i][0].w, raw[i][1].x, raw[i][1].y, raw[i][1].z, raw[i][1].w};
                if (a.coef) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                }
                if (a.swish) {
#pragma unroll
                    // x * sigmoid(x) with the hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~1e-7 relative, far below
                    // the 2^-16 of the hi/lo split that follows); IEEE expf + division cost ~30 VALU ops per element and
                    // were a third of this kernel's issue cycles
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * __frcp_rn(1.0f + __expf(-f[e]));
                }
                uint4 hi, lo;
                split8(f, hi, lo);
                const unsigned m_ = sok[i] ? 0xffffffffu : 0u;  // conv zero padding applies AFTER norm / swish
                const int o = (it & 3) * FGS + (it >> 2) * 16;
                *(uint4*)(sH + o) = make_uint4(hi.x & m_, hi.y & m_, hi.z & m_, hi.w & m_);
                *(uint4*)(sH + PLANE + o) = make_uint4(lo.x & m_, lo.y & m_, lo.z & m_, lo.w & m_);
            }
        }
    };
    const uint4* wbase = a.w + (size_t)nb * nkc * TAPS * (WT / 16);
    // Two register staging sets for the (L2-resident) weight tiles: the loads of step s+2 are issued before the MFMAs
    // of step s and written to LDS after the MFMAs of step s+1, so a load has two compute phases to land (with one set
    // every step ended waiting out the L2 latency of the load it had just issued: 3900 cycles per step for 770 cycles
    // of MFMA).  Scalars, not an indexed array: hipcc leaves such an array in scratch memory across the two loop levels.
    uint4 wa0 = make_uint4(0, 0, 0, 0), wa1 = wa0, wa2 = wa0, wa3 = wa0, wb0 = wa0, wb1 = wa0, wb2 = wa0, wb3 = wa0;
    auto gload_w = [&](int step_, uint4& r0, uint4& r1, uint4& r2, uint4& r3) {
        const int st_ = step_ < nsteps ? step_ : nsteps - 1;  // tail: harmless re-load
        const uint4* src = wbase + (size_t)st_ * (WT / 16);
        if constexpr (W_IT == 4) {
            r0 = src[t]; r1 = src[t + NT]; r2 = src[t + 2 * NT]; r3 = src[t + 3 * NT];
        } else if constexpr (W_IT == 2) {
            r0 = src[t]; r1 = src[t + NT];
        } else {
            r0 = src[t < WT / 16 ? t : 0];
        }
    };
    auto lstore_w = [&](int buf_, const uint4& r0, const uint4& r1, const uint4& r2, const uint4& r3) {
        unsigned char* dst = sW + buf_ * WT + t * 16;
        if constexpr (W_IT == 4) {
            *(uint4*)(dst) = r0; *(uint4*)(dst + 16 * NT) = r1; *(uint4*)(dst + 32 * NT) = r2; *(uint4*)(dst + 48 * NT) = r3;
        } else if constexpr (W_IT == 2) {
            *(uint4*)(dst) = r0; *(uint4*)(dst + 16 * NT) = r1;
        } else {
            if (t < WT / 16) *(uint4*)(dst) = r0;
        }
    };
    static_assert((W_IT == 4 && WT / 16 == 4 * NT) || (W_IT == 2 && WT / 16 == 2 * NT) || (W_IT == 1 && WT / 16 <= NT),
                  "weight tile staging shape");

    f32x4_t acc[JN][JM];
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int i = 0; i < JM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    int kc = 0, tap = 0;
    const int fr = lane & 15, fg = lane >> 4;
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma_w = [&](int step_, int buf_) {  // weight tile of `step_` -> LDS buffer `buf_`, one fragment per wave-instruction
        if (step_ < nsteps) {
            const uint4* src = wbase + (size_t)step_ * (WT / 16) + lane;
#pragma unroll
            for (int i = 0; i < (WT / 1024 + NWV - 1) / NWV; ++i) {
                const int c = wv + i * NWV;
                if ((WT / 1024) % NWV == 0 || c < WT / 1024)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + c * 64), (lptr_t)(sW + buf_ * WT + c * 1024), 16, 0, 0);
            }
        }
    };
    // DMA form of one step: ALL fragment reads first, then the next step's weight DMA (+ the next chunk's halo loads on the
    // first tap: issued after the DMA so that the counted wait below covers the DMA only), then the MFMAs
    auto compute_dma = [&](int wbuf, int tap, int s, bool halo_next) {
        const int ty = tap / KS, tx = tap - ty * KS;
        const unsigned char* hb = sH + fg * FGS + (size_t)((wm * JM + ty) * HC + fr + tx) * 16;
        uint4 phi[JM], plo[JM], wh[JN], wl[JN];
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            phi[i] = *(const uint4*)(hb + i * HC * 16);
            plo[i] = *(const uint4*)(hb + PLANE + i * HC * 16);
        }
        const unsigned char* wb = sW + wbuf * WT + lane * 16;
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            wh[j] = *(const uint4*)(wb + (wn * JN + j) * 1024);
            wl[j] = *(const uint4*)(wb + (BN / 16 + wn * JN + j) * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads above are done before the DMA below may overwrite
        dma_w(s + 1, wbuf ^ 1);                             // (buffer wbuf^1 was last read one step ago, by every wave)
        if (halo_next) CF_GLOAD_HALO(kc + 1);
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int i = 0; i < JM; ++i) {
                acc[j][i] = BF16::mma(wl[j], phi[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh[j], plo[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh[j], phi[i], acc[j][i]);
            }
    };
    auto compute = [&](int wbuf, int tap) {
        const int ty = tap / KS, tx = tap - ty * KS;
        const unsigned char* hb = sH + fg * FGS + (size_t)((wm * JM + ty) * HC + fr + tx) * 16;
        uint4 phi[JM], plo[JM];
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            phi[i] = *(const uint4*)(hb + i * HC * 16);
            plo[i] = *(const uint4*)(hb + PLANE + i * HC * 16);
        }
        const unsigned char* wb = sW + wbuf * WT + lane * 16;
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            const uint4 wh = *(const uint4*)(wb + (wn * JN + j) * 1024);
            const uint4 wl = *(const uint4*)(wb + (BN / 16 + wn * JN + j) * 1024);
#pragma unroll
            for (int i = 0; i < JM; ++i) {
                acc[j][i] = BF16::mma(wl, phi[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh, plo[i], acc[j][i]);
                acc[j][i] = BF16::mma(wh, phi[i], acc[j][i]);
            }
        }
    };

    if constexpr (DMA && PIPE != 0) {
        static_assert(KS == 3 && 2 * (JN - 1) + 2 * JM <= 15, "pipelined form: 3x3 taps; the largest counted wait must fit the 4-bit lgkmcnt");
        u32x4_t px[2][2 * JM];   // [step parity][phi 0..JM-1 | plo 0..JM-1]
        // what is requested one tap ahead: PIPE = 1 the hi fragments only (224 VGPRs), PIPE = 3 both planes (234): both fit the 256 of two
        // waves per SIMD once the nine taps of a chunk are unrolled with the tap known at compile time (a runtime tap counter left
        // hipcc unable to see that nothing is requested across a chunk boundary: 256 VGPRs + 300-400 spilled)
        constexpr int PFW = PIPE, PFN = (PFW == 3 ? 2 : 1) * JM;
        u32x4_t wf[2][2];        // two (wh, wl) pairs in flight: the pair of n-tile j + 2 is requested behind the MFMAs of n-tile j
        const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
        const unsigned aH = lds0 + fg * FGS + (unsigned)((wm * JM) * HC + fr) * 16;
        const unsigned aW = lds0 + HALO + lane * 16 + wn * JN * 1024;
        // which = 1: the hi fragments, 2: the lo fragments, 3: both (hi first); T = tap (compile time: the nine taps of a chunk are unrolled)
        auto rd_px = [&](auto set_, auto which_, auto t_) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value, WH = decltype(which_)::value, T = decltype(t_)::value;
            constexpr int OFFT = ((T / KS) * HC + (T % KS)) * 16;
            if constexpr (WH & 1) cf_static_for<0, JM>([&](auto i_) { px[S][decltype(i_)::value] = cf_lds_rd<OFFT + decltype(i_)::value * HC * 16>(aH); });
            if constexpr (WH & 2) cf_static_for<0, JM>([&](auto i_) { px[S][JM + decltype(i_)::value] = cf_lds_rd<OFFT + PLANE + decltype(i_)::value * HC * 16>(aH); });
        };
        // one step = tap T of the current chunk; S = T & 1 is the pixel set it consumes, PF = the next tap's fragments are requested here.
        // LDS operations in issue order: [this tap's pixels not requested ahead] w0 w1 [next tap's] w2 w3; they return in order, so
        // "pair j landed" is lgkmcnt <= what was issued behind it.
        auto step = [&](auto t_, int wbuf, int s, bool halo_next) __attribute__((always_inline)) {
            constexpr int T = decltype(t_)::value, S = T & 1, PF = T + 1 < TAPS ? 1 : 0;
            static_assert(JN == 4, "pipelined form: four n-tiles per wave");
            const unsigned a0 = aW + (unsigned)wbuf * WT;
            wf[0][0] = cf_lds_rd<0 * 1024>(a0);
            wf[0][1] = cf_lds_rd<(BN / 16 + 0) * 1024>(a0);
            wf[1][0] = cf_lds_rd<1 * 1024>(a0);
            wf[1][1] = cf_lds_rd<(BN / 16 + 1) * 1024>(a0);
            dma_w(s + 1, wbuf ^ 1);                 // (buffer wbuf^1 was last read one step ago, by every wave)
            if constexpr (PF) rd_px(std::integral_constant<int, S ^ 1>{}, std::integral_constant<int, PFW>{}, std::integral_constant<int, T + 1>{});
            if (halo_next) CF_GLOAD_HALO(kc + 1);
            cf_static_for<0, JN>([&](auto j_) {
                constexpr int J = decltype(j_)::value, SL = J & 1;
                cf_wait_lgkm<J == 0 ? 2 + PFN * PF : (J == 1 ? PFN * PF + 2 : (J == 2 ? 2 : 0))>();
                if constexpr (J == 0) cf_static_for<0, 2 * JM>([&](auto i_) { cf_touch(px[S][decltype(i_)::value]); });
                cf_touch(wf[SL][0]);
                cf_touch(wf[SL][1]);
                const uint4 wh = cf_u4(wf[SL][0]), wl = cf_u4(wf[SL][1]);
#pragma unroll
                for (int i = 0; i < JM; ++i) {
                    const uint4 phi = cf_u4(px[S][i]), plo = cf_u4(px[S][JM + i]);
                    acc[J][i] = BF16::mma(wl, phi, acc[J][i]);
                    acc[J][i] = BF16::mma(wh, plo, acc[J][i]);
                    acc[J][i] = BF16::mma(wh, phi, acc[J][i]);
                }
                if constexpr (J + 2 < JN) {          // the pair two n-tiles ahead, into the registers this n-tile's MFMAs have consumed
                    wf[SL][0] = cf_lds_rd<(J + 2) * 1024>(a0);
                    wf[SL][1] = cf_lds_rd<(BN / 16 + J + 2) * 1024>(a0);
                }
            });
        };
        dma_w(0, 0);
        CF_GLOAD_HALO(0);
        int s = 0;
        for (kc = 0; kc < nkc; ++kc) {
            store_halo();      // (waits for the raw halo loads and, in program order before them, the DMA of this step)
            __syncthreads();
            const bool more = kc + 1 < nkc;
            rd_px(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
            cf_static_for<0, TAPS>([&](auto t_) {
                constexpr int T = decltype(t_)::value;
                if constexpr (T > 0 && (3 & ~PFW) != 0)   // what tap T - 1 did not request ahead
                    rd_px(std::integral_constant<int, T & 1>{}, std::integral_constant<int, 3 & ~PFW>{}, t_);
                step(t_, s & 1, s, more && T == 0);
                // own DMA of step s+1 landed (the halo loads issued after it may stay in flight), then everybody's
                if (more && T == 0) {
                    if (a.coef) wait_vmcnt<2 * ITER + 4>(); else wait_vmcnt<2 * ITER>();
                } else {
                    wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();
                ++s;
            });
        }
    } else if constexpr (DMA) {
        dma_w(0, 0);
        CF_GLOAD_HALO(0);
        for (int s = 0; s < nsteps; ++s) {
            bool halo_next = false;
            if (tap == 0) {  // every wave is past the barrier that ended the previous chunk's last tap
                store_halo();      // (waits for the raw halo loads and, in program order before them, the DMA of this step)
                __syncthreads();
                halo_next = kc + 1 < nkc;
            }
            compute_dma(s & 1, tap, s, halo_next);
            // own DMA of step s+1 landed (the halo loads issued after it may stay in flight), then everybody's
            if (halo_next) {
                if (a.coef) wait_vmcnt<2 * ITER + 4>(); else wait_vmcnt<2 * ITER>();
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            if (++tap == TAPS) { tap = 0; ++kc; }
        }
    } else {
    gload_w(0, wa0, wa1, wa2, wa3);
    CF_GLOAD_HALO(0);
    gload_w(1, wb0, wb1, wb2, wb3);
    lstore_w(0, wa0, wa1, wa2, wa3);
    // one step = one filter tap of one 32-channel chunk; LD* = the set whose contents (step s) are already in LDS,
    // ST* = the set holding step s+1
#define CF_STEP(s_, cur_, LD0, LD1, LD2, LD3, ST0, ST1, ST2, ST3)                                   \
    {                                                                                               \
        if (tap == 0) { /* every wave is past the barrier that ended the previous chunk's last tap */ \
            store_halo();                                                                           \
            __syncthreads();                                                                        \
            if (kc + 1 < nkc) CF_GLOAD_HALO(kc + 1); /* lands during the taps of this chunk */     \
        }                                                                                           \
        gload_w((s_) + 2, LD0, LD1, LD2, LD3);                                                      \
        compute(cur_, tap);                                                                         \
        lstore_w((cur_) ^ 1, ST0, ST1, ST2, ST3);                                                   \
        __syncthreads();                                                                            \
        if (++tap == TAPS) { tap = 0; ++kc; }                                                       \
    }
    for (int s = 0; s < nsteps; s += 2) {
        CF_STEP(s, 0, wa0, wa1, wa2, wa3, wb0, wb1, wb2, wb3);
        if (s + 1 < nsteps) CF_STEP(s + 1, 1, wb0, wb1, wb2, wb3, wa0, wa1, wa2, wa3);
    }
    }
#undef CF_STEP
#undef CF_GLOAD_HALO

    // epilogue: lane holds channels n = n0 + g*4 + {0..3} of pixel (y0 + row, x0 + fr).  W need not be a multiple of the
    // 16-pixel tile width (24 x 24 maps of a 384 px decode): columns >= W of the last tile column are computed and dropped.
    const size_t HW = (size_t)a.H * a.W;
    float* outb = a.out + (size_t)b * HW * a.Cout;
    const float* resb = a.res ? a.res + (size_t)b * HW * a.Cout : nullptr;
    float* red = (float*)smem;  // [WMW][BN/4][2] after the main loop (all LDS reads are behind the last barrier)
    const int cntx = (a.W - x0) < TW ? (a.W - x0) : TW;   // valid columns of this tile (uniform over the workgroup)
    const bool inx = fr < cntx;
    const int xs = x0 + (inx ? fr : cntx - 1);             // clamped column: loads of dropped pixels stay inside the image
    // All residual loads of the wave's tiles first (the halo / fragment registers are dead here): inside the store loop below each
    // load would sit behind the previous tile's store (`out` and `res` may alias as far as the compiler knows), i.e. JN x JM
    // dependent memory round trips per tile -- measured as 10 % of the kernel.
    const bool res_fast = resb && !a.out_nchw && (nb + 1) * BN <= a.Cout;
    float4 rres[JN][JM];
    if (res_fast) {
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int i = 0; i < JM; ++i)
                rres[j][i] = *(const float4*)(resb + ((size_t)(y0 + wm * JM + i) * a.W + xs) * a.Cout + nb * BN + (wn * JN + j) * 16 + fg * 4);
    }
    const float nsub = 4.0f * JM * cntx;   // values per (lane group of 16 pixels x JM rows x 4 channels) statistic
#pragma unroll
    for (int j = 0; j < JN; ++j) {
        const int nl = (wn * JN + j) * 16 + fg * 4;
        const int n = nb * BN + nl;
        float bs[4] = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) {
            if (n + 3 < a.Cout) {
                const float4 bv = *(const float4*)(a.bias + n);
                bs[0] = bv.x; bs[1] = bv.y; bs[2] = bv.z; bs[3] = bv.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[e] = n + e < a.Cout ? a.bias[n + e] : 0.f;
            }
        }
        float vals[JM][4];
#pragma unroll
        for (int i = 0; i < JM; ++i) {
            const int y = y0 + wm * JM + i;
            const size_t p = (size_t)y * a.W + xs;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[j][i][e] + bs[e];
            if (!a.out_nchw && n + 3 < a.Cout) {
                const size_t o = p * a.Cout + n;
                if (resb) {
                    const float4 r = res_fast ? rres[j][i] : *(const float4*)(resb + o);
                    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
                }
                if (inx) *(float4*)(outb + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e >= a.Cout) { v[e] = 0.f; continue; }
                    const size_t o = a.out_nchw ? (size_t)(n + e) * HW + p : p * a.Cout + n + e;
                    if (resb) v[e] += resb[o];
                    if (inx) outb[o] = v[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) vals[i][e] = v[e];
        }
        if (a.part) {
            // Statistics of the stored values for the NEXT GroupNorm as (sum, M2 = sum of squared deviations from the local mean):
            // first over the wave's 16 pixels x JM rows x 4 channels (fixed-order shuffles), then over wm through LDS with the
            // pairwise update of Chan et al.  (A plain sum of squares loses the variance to cancellation when |mean| >> std.)
            float ssum = 0.f;
#pragma unroll
            for (int i = 0; i < JM; ++i) ssum += inx ? (vals[i][0] + vals[i][1]) + (vals[i][2] + vals[i][3]) : 0.f;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) ssum += __shfl_xor(ssum, o, 64);
            const float mean = ssum / nsub;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < JM; ++i) {
                const float d0 = vals[i][0] - mean, d1 = vals[i][1] - mean, d2 = vals[i][2] - mean, d3 = vals[i][3] - mean;
                m2 += inx ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m2 += __shfl_xor(m2, o, 64);
            if (fr == 0) {
                red[(wm * (BN / 4) + (nl >> 2)) * 2 + 0] = ssum;
                red[(wm * (BN / 4) + (nl >> 2)) * 2 + 1] = m2;
            }
        }
    }
    if (a.part) {
        __syncthreads();
        if (t < BN / 4) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) s += red[(w * (BN / 4) + t) * 2 + 0];
            const float mean = s / (nsub * WMW);
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WMW; ++w) {
                const float dm = red[(w * (BN / 4) + t) * 2 + 0] / nsub - mean;
                m2 += red[(w * (BN / 4) + t) * 2 + 1] + nsub * dm * dm;
            }
            float* dst = a.part + (((size_t)b * a.ntiles + tile) * (a.Npad / 4) + nb * (BN / 4) + t) * 2;
            dst[0] = s;
            dst[1] = m2;
        }
    }
}

template <int JN, int WNW, int KS, int NWV, bool DMA = false, int PIPE = 0>
static int launch_cf(const ConvFArgs& a, int B, hipStream_t st) {
    constexpr int PAD = KS / 2, NP = (8 + 2 * PAD) * (16 + 2 * PAD), NPP = (NP + 15) / 16 * 16;
    constexpr int BN = WNW * JN * 16;
    constexpr size_t lds = 2 * 4 * NPP * 16 + 2 * 2 * (BN / 16) * 1024;
    if (a.Npad % BN) return LGEN_ERR_BAD_ARG;
    dim3 grid(a.ntiles, a.Npad / BN, B);
    hipLaunchKernelGGL((conv_fused_kernel<JN, WNW, KS, NWV, DMA, PIPE>), grid, dim3(64 * NWV), lds, st, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}

static int g_cf_dma = 3;  // (default 3 since round 6: 128 -> 128 at 384 px 4.2-4.45 -> 3.95 ms, decode_code 52.4-53.7 -> 50.8 ms, bit-identical: profiles/r06_conv_pipe_ab.log) weight tiles by LDS-DMA (1, 2, 3) or through staging registers (0); 2 / 3: the pipelined fragment reads (3x3, 128 channels) with the hi / both pixel planes requested one tap ahead; lgen_debug_set_conv_fused_variant
extern "C" int lgen_debug_set_conv_fused_variant(int v) { g_cf_dma = v < 0 || v > 3 ? 3 : v; return 0; }

// weight tile width (output channels per workgroup) this library uses for a given Cout: the host packs to it
extern "C" int lgen_conv_fused_bn(int Cout) { return Cout >= 128 ? 128 : (Cout > 16 ? 64 : 16); }

extern "C" int lgen_conv_fused(const float* x_nhwc, const float* gn_coef, int swish, const void* w_frag, const float* bias,
                               const float* res, float* out, float* stats_partial, int B, int H, int W, int Cin, int Cout,
                               int Npad, int ksize, int upsample, int out_nchw, void* stream) {
    if ((ksize != 1 && ksize != 3) || Cin % 32 || H % 8 || W < 1 || Npad < Cout || (upsample && ((H | W) & 1)) ||
        upsample < 0 || upsample > 1)
        return LGEN_ERR_BAD_ARG;
    if (B == 0) return 0;
    const int bn = lgen_conv_fused_bn(Cout);
    if (Npad % bn) return LGEN_ERR_BAD_ARG;
    ConvFArgs a{x_nhwc, (const float2*)gn_coef, (const uint4*)w_frag, bias, res, out, stats_partial,
                H, W, Cin, Cout, Npad, upsample, swish ? 1 : 0, out_nchw, (W + 15) / 16, (H / 8) * ((W + 15) / 16)};
    hipStream_t st = (hipStream_t)stream;
    // 4 waves x (64 ch x 64 px).  Measured alternative: 8 waves x (32 ch x 64 px) per workgroup (4 waves per SIMD instead of 2)
    // runs at the same speed (2.54 vs 2.57 ms, 16 x 384 px, 128 -> 128): the kernel is not short of waves to hide latency.
    if (bn == 128) {
        if (g_cf_dma == 2 && ksize == 3) return launch_cf<4, 2, 3, 4, true, 1>(a, B, st);
        if (g_cf_dma == 3 && ksize == 3) return launch_cf<4, 2, 3, 4, true, 3>(a, B, st);
        if (g_cf_dma) return ksize == 3 ? launch_cf<4, 2, 3, 4, true>(a, B, st) : launch_cf<4, 2, 1, 4, true>(a, B, st);
        return ksize == 3 ? launch_cf<4, 2, 3, 4>(a, B, st) : launch_cf<4, 2, 1, 4>(a, B, st);
    }
    if (bn == 64) return ksize == 3 ? launch_cf<4, 1, 3, 4>(a, B, st) : launch_cf<4, 1, 1, 4>(a, B, st);
    return ksize == 3 ? launch_cf<1, 1, 3, 4>(a, B, st) : launch_cf<1, 1, 1, 4>(a, B, st);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics -> per-channel (scale, shift): coef[b][c] = (rstd*gamma_c, beta_c - rstd*gamma_c*mean)
// (vq_model.py:359-362, eps inside the sqrt, biased variance).  Input is either the per-tile partials written by
// lgen_conv_fused (ntiles > 0: part[b][tile][C/4][2] = (sum, M2), fp32, combined here in fp64 in a fixed order) or finished
// statistics stats[b][32][2] = (mean, rstd) from lgen_gn_stats (ntiles == 0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float2* __restrict__ coef, int ntiles, int C, int Cq_stride,
                                                         double count, float eps, int width) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int gs = C >> 5;
    float mean, rstd;
    if (ntiles > 0) {
        // partials are (sum, M2 about the partial's own mean) over n_i = 4 channels x 8 rows x (valid columns of the tile) values:
        // total mean first, then M2 = sum_i [M2_i + n_i (mean_i - mean)^2] (Chan et al.), in fp64, fixed order
        const int qpg = gs >> 2;  // 4-channel quads per group
        const int tiles_x = (width + 15) >> 4;
        double S = 0.0;
        for (int i = lane; i < ntiles * qpg; i += 64) {
            const int tile = i / qpg, qq = i - tile * qpg;
            S += (double)part[(((size_t)b * ntiles + tile) * Cq_stride + g * qpg + qq) * 2];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) S += __shfl_xor(S, o, 64);
        const double m = S / count;
        double Q = 0.0;
        for (int i = lane; i < ntiles * qpg; i += 64) {
            const int tile = i / qpg, qq = i - tile * qpg;
            const float* p = part + (((size_t)b * ntiles + tile) * Cq_stride + g * qpg + qq) * 2;
            const int tx = tile % tiles_x;
            const int cnt = width - 16 * tx < 16 ? width - 16 * tx : 16;
            const double n_i = 32.0 * cnt;
            const double dm = (double)p[0] / n_i - m;
            Q += (double)p[1] + n_i * dm * dm;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) Q += __shfl_xor(Q, o, 64);
        double var = Q / count;
        var = var < 0.0 ? 0.0 : var;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        mean = stats[((size_t)b * 32 + g) * 2];
        rstd = stats[((size_t)b * 32 + g) * 2 + 1];
    }
    for (int c = g * gs + lane; c < (g + 1) * gs; c += 64) {
        const float sc = rstd * gamma[c];
        coef[(size_t)b * C + c] = make_float2(sc, beta[c] - sc * mean);
    }
}

extern "C" int lgen_gn_finalize(const float* partial, const float* stats, const float* gamma, const float* beta, float* coef,
                                int B, int C, int ntiles, int quad_stride, int hw, int width, float eps, void* stream) {
    if (C % 128 || (!partial && !stats) || (partial && (ntiles < 1 || width < 1 || ntiles % ((width + 15) / 16)))) return LGEN_ERR_BAD_ARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, B), dim3(64), 0, (hipStream_t)stream, partial, stats, gamma, beta,
                       (float2*)coef, partial ? ntiles : 0, C, quad_stride, (double)hw * (C / 32), eps, width);
    LGEN_CHECK_LAUNCH();
    return 0;
}
