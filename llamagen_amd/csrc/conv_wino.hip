// Winograd F(2x2, 3x3) form of the VQ decoder's 3x3 convolutions (round 5): the same operation and the same fusions as
// conv_fused.hip -- GroupNorm-apply + swish on the way in (vq_model.py:299-306, 354-364), nn.Conv2d 3x3 pad 1 (:288-291),
// nearest-2x upsampling in front of Upsample.conv (:374-378), bias + residual add (:314) and the NEXT GroupNorm's (sum, M2)
// partials on the way out -- with 16 multiplies per 2x2 output block and input channel instead of 36.
//
// Why: decode_code() is 20 % of a bench step and its 3x3 convolutions (95 % of its FLOPs at maps >= 48 x 48) ran at 0.89 of what
// the vendor GEMM reaches on this chip: MFMA-bound, power-limited.  Nothing is left in the schedule, so the arithmetic has to
// shrink: Y = A^T [ (G g G^T) (.) (B^T d B) ] A (Lavin & Gray; correlation form, as nn.Conv2d) needs 2.25 x fewer MFMA products.
// Numerics: both transforms are exact small-integer / half-integer combinations evaluated in fp32; the products keep the 3-pass
// split-bf16 form (hi*hi + hi*lo + lo*hi, fp32 accumulate) on the TRANSFORMED operands.  CPU emulation of the whole decoder
// (tools/wino_emulate.py): 7.7e-5 max abs against the fp32 decoder, 5.8e-5 for the direct split-bf16 form; F(4x4, 3x3) would
// give 4 x fewer products but 3.3e-4 -- too close to the 1e-3 bar, not used.
//
// One 512-thread workgroup (8 waves, one per CU) owns 8 x 16 output pixels (= 4 x 8 Winograd tiles of 2 x 2) x 128 output
// channels -- the tile of conv_fused.hip, so the statistics partials keep their layout.  Per 32-channel chunk of the input:
//   1. the 10 x 18 halo pixels are loaded ONCE (fp32, prefetched a chunk ahead), normalised / activated in registers and parked
//      in LDS as fp32;
//   2. transform: every (tile, 4-channel quad) item computes its 16 values B^T d B in registers, splits them and writes them as
//      MFMA B-operand fragments [position][hi|lo][tile group][k-slice][tile][8 ch] (two threads per item, two rows of the 4 x 4
//      each);
//   3. 16 position GEMMs [128 cout x 32 cin] x [32 cin x 32 tiles], two per step: wave (wn, wm, ps) owns cout tiles 4 wn .. 4 wn+3,
//      tile group wm and the positions of parity ps; the transformed weights (hi | lo, fragment-packed on the host) arrive by
//      LDS-DMA one step ahead; accumulators of ALL 16 positions stay in registers over the whole K loop (8 positions x 4 tiles per
//      wave = 128 VGPRs).
// Epilogue: A^T M A from the accumulators (the two position parities meet through LDS), bias, residual, float4 NHWC stores and the
// (sum, M2) partials of every 8 x 16 tile x channel quad.
#include <cstdlib>
#include <type_traits>

#include "lgen_common.h"
#include "../../include/lgen.h"

namespace {

struct ConvWArgs {
    const float* x;      // [B][Hs][Ws][Cin] fp32 NHWC (Hs = H >> ups)
    const float2* coef;  // [B][Cin] (scale, shift) of the fused GroupNorm, or null
    const uint4* w;      // [Cout/128][Cin/32][16 positions][2 planes][8 cout tiles][64 lanes] x 16 B (MFMA A-fragment order)
    const float* bias;   // [Cout] or null
    const float* res;    // NHWC like out, or null
    float* out;          // NHWC [B][H][W][Cout]
    float* part;         // [B][ntiles][Cout/4][2] partial (sum, M2) of the stored output, or null
    int H, W, Cin, Cout, ups, swish, tiles_x, ntiles, dbg;   // dbg: development ablation mask (LGEN_WINO_ABLATE), 0 in production
};

constexpr int WN_HC = 18, WN_NP = 10 * 18;           // halo columns / pixels of an 8 x 16 output tile
constexpr int WN_RS = 144;                           // bytes per halo pixel in LDS: 32 channels fp32 + 16 B (bank spread)
constexpr int WN_SV = 16 * 2 * 2 * 1024;             // transformed input: [16 positions][hi|lo][2 tile groups] x 1 KiB fragments
constexpr int WN_WT = 2 * 8 * 1024;                  // one position's weight tile: [hi|lo][8 cout tiles] x 1 KiB fragments
constexpr int WN_STEP = 2 * WN_WT;                   // a step = two positions
constexpr int WN_SW = 2 * WN_STEP;                   // double-buffered
constexpr int WN_SR = WN_NP * WN_RS;
constexpr int WN_LDS = WN_SV + WN_SW + WN_SR;        // 157 056 B of the CU's 160 KiB
constexpr int WN_NT = 512;
constexpr int WN_ITER = (WN_NP * 8 + WN_NT - 1) / WN_NT;   // float4 staging items per thread and chunk (3)

template <int N>
LGEN_DEV void wn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
LGEN_DEV void wn_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// operand reads of the position GEMMs as inline asm: hipcc cannot tell which LDS bytes a pending `global_load_lds` writes and
// drains vmcnt in front of every LDS read it can see (gemm_tile.hip has the long version of this note)
typedef __attribute__((ext_vector_type(4))) unsigned wn_u4;
template <int OFF>
LGEN_DEV wn_u4 wn_lds_rd(unsigned addr) {
    wn_u4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
LGEN_DEV void wn_touch(wn_u4& v) { asm volatile("" : "+v"(v)); }
LGEN_DEV f32x4_t wn_mma(const wn_u4& a, const wn_u4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int I, int N, typename F>
LGEN_DEV void wn_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wn_static_for<I + 1, N>(f);
    }
}

LGEN_DEV float4 f4sub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
LGEN_DEV float4 f4add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// fp32 x 4 -> (hi, lo) bf16 x 4 with hi + lo = x to ~2^-17 relative
LGEN_DEV void wn_split4(const float4& v, uint2& hi, uint2& lo) {
    hi.x = f2bf2(v.x, v.y);
    hi.y = f2bf2(v.z, v.w);
    lo.x = f2bf2(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u));
    lo.y = f2bf2(v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
}

typedef __attribute__((address_space(1))) const void* wn_gptr_t;
typedef __attribute__((address_space(3))) void* wn_lptr_t;

__global__ __launch_bounds__(WN_NT, 2) void conv_wino_kernel(ConvWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned char* sW = smem + WN_SV;
    unsigned char* sR = smem + WN_SV + WN_SW;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wv & 1, wm = (wv >> 1) & 1, ps = wv >> 2;
    const int b = blockIdx.z, nb = blockIdx.y;
    int tile = blockIdx.x;
    {   // XCD-aware tile order (conv_fused.hip): workgroup ids go round-robin over the 8 XCDs; give each a contiguous run of tiles
        const int nt = a.ntiles, q = nt >> 3, r = nt & 7, xcd = tile & 7, k = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int y0 = (tile / a.tiles_x) * 8, x0 = (tile % a.tiles_x) * 16;
    const int Hs = a.H >> a.ups, Ws = a.W >> a.ups;
    const int nkc = a.Cin >> 5;
    const int nsteps = nkc * 8;
    const float* xb = a.x + (size_t)b * Hs * Ws * a.Cin;
    const unsigned lds0 = (unsigned)(uintptr_t)(wn_lptr_t)smem;
    const bool has_coef = a.coef != nullptr;

    // ---- halo staging items of this thread: it = t + i * 512 -> (halo pixel it >> 3, channel quad it & 7 = t & 7) ----
    const int cq_s = t & 7;
    int soff[WN_ITER];
    bool sok[WN_ITER];
#pragma unroll
    for (int i = 0; i < WN_ITER; ++i) {
        const int it = t + i * WN_NT;
        const int P = (it >> 3) < WN_NP ? (it >> 3) : WN_NP - 1;
        const int hy = P / WN_HC, hx = P - hy * WN_HC;
        const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
        sok[i] = (it < WN_NP * 8) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        const int sy = sok[i] ? (yy >> a.ups) : 0, sx = sok[i] ? (xx >> a.ups) : 0;
        soff[i] = (sy * Ws + sx) * a.Cin + cq_s * 4;
    }
    float4 raw[WN_ITER];
    float4 cf0 = make_float4(1.f, 0.f, 1.f, 0.f), cf1 = cf0;
    auto gload_halo = [&](int kc) {
#pragma unroll
        for (int i = 0; i < WN_ITER; ++i) raw[i] = *(const float4*)(xb + soff[i] + kc * 32);
        if (has_coef) {
            const float4* c = (const float4*)(a.coef + (size_t)b * a.Cin + kc * 32 + cq_s * 4);
            cf0 = c[0];
            cf1 = c[1];
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < WN_ITER; ++i) {
            const int it = t + i * WN_NT;
            if (i + 1 < WN_ITER || it < WN_NP * 8) {
                float f[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
                if (has_coef) {
                    f[0] = fmaf(f[0], cf0.x, cf0.y);
                    f[1] = fmaf(f[1], cf0.z, cf0.w);
                    f[2] = fmaf(f[2], cf1.x, cf1.y);
                    f[3] = fmaf(f[3], cf1.z, cf1.w);
                }
                if (a.swish) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = f[e] * __frcp_rn(1.0f + __expf(-f[e]));   // as conv_fused.hip
                }
                // (the conv's zero padding applies AFTER norm / swish)
                *(float4*)(sR + (it >> 3) * WN_RS + cq_s * 16) = sok[i] ? make_float4(f[0], f[1], f[2], f[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    // ---- transform item of this thread: (tile ty, tx; channel quad cq; half of the 4 x 4: rows pi = 2 hrow, 2 hrow + 1) ----
    // lane bits chosen so that the 16 lanes of a ds_write_b64 group cover one contiguous 128 B: bit 0 = half of the 16-B slot,
    // bits 1-3 = tx, bits 4-5 = k-slice g, bits 6-7 = ty
    const int q = t & 255, hrow = t >> 8;
    const int half = q & 1, ttx = (q >> 1) & 7, tg_g = (q >> 4) & 3, tty = (q >> 6) & 3;
    const int cq_t = tg_g * 2 + half;
    const unsigned char* rbase = sR + ((2 * tty + hrow) * WN_HC + 2 * ttx) * WN_RS + cq_t * 16;
    unsigned char* vbase = sV + (tty >> 1) * 1024 + tg_g * 256 + ((tty & 1) * 8 + ttx) * 16 + half * 8 + (2 * hrow) * 4 * 4096;
    auto transform = [&]() {
        // rows r0 .. r0+2 of the 4 x 4 input patch, r0 = hrow: hrow 0 -> (d0 - d2, d1 + d2), hrow 1 -> (d2 - d1, d1 - d3)
        float4 tA[4], tB[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 R0 = *(const float4*)(rbase + c * WN_RS);
            const float4 R1 = *(const float4*)(rbase + (WN_HC + c) * WN_RS);
            const float4 R2 = *(const float4*)(rbase + (2 * WN_HC + c) * WN_RS);
            if (hrow == 0) {
                tA[c] = f4sub(R0, R2);
                tB[c] = f4add(R1, R2);
            } else {
                tA[c] = f4sub(R1, R0);
                tB[c] = f4sub(R0, R2);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const float4* tt = rr ? tB : tA;
            float4 v[4];
            v[0] = f4sub(tt[0], tt[2]);
            v[1] = f4add(tt[1], tt[2]);
            v[2] = f4sub(tt[2], tt[1]);
            v[3] = f4sub(tt[1], tt[3]);
#pragma unroll
            for (int pj = 0; pj < 4; ++pj) {
                uint2 hi, lo;
                wn_split4(v[pj], hi, lo);
                unsigned char* dst = vbase + (rr * 4 + pj) * 4096;   // position p = (2 hrow + rr) * 4 + pj; [p][plane][tg] x 1 KiB
                *(uint2*)(dst) = hi;
                *(uint2*)(dst + 2048) = lo;
            }
        }
    };

    // ---- weight DMA: step s = two positions = 32 x 1 KiB fragments, four per wave ----
    const uint4* wbase = a.w + (size_t)nb * nkc * 16 * (WN_WT / 16) + lane;
    auto dma_w = [&](int s, int buf) {
        if (s < nsteps && !(a.dbg & 4)) {
            const uint4* src = wbase + (size_t)s * (WN_STEP / 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = wv + i * 8;
                __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + c * 64), (wn_lptr_t)(sW + buf * WN_STEP + c * 1024), 16, 0, 0);
            }
        }
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[s][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned aV0 = lds0 + ps * 4096 + wm * 1024 + lane * 16;                       // + st * 8192 (+ 2048: lo plane)
    const unsigned aW0 = lds0 + WN_SV + ps * WN_WT + (wn * 4) * 1024 + lane * 16;        // + buf * WN_STEP (+ 8192: lo plane)

    dma_w(0, 0);
    gload_halo(0);
    for (int kc = 0; kc < nkc; ++kc) {
        if (kc == 0) {
            store_halo();
            wn_wait_vm<0>();
        }
        __syncthreads();           // halo of chunk kc complete (kc > 0: stored during the previous chunk's steps); weights of step 8 kc landed
        if (!(a.dbg & 1)) transform();
        __syncthreads();
        const bool halo_next = kc + 1 < nkc;
        wn_static_for<0, 8>([&](auto st_) {
            constexpr int ST = decltype(st_)::value;
            const int s = kc * 8 + ST;
            const unsigned aV = aV0 + ST * 8192, aW = aW0 + (ST & 1) * WN_STEP;
            if (a.dbg & 2) {
                dma_w(s + 1, (ST & 1) ^ 1);
                if (ST == 0 && halo_next) gload_halo(kc + 1);
                wn_wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                if (ST == 1 && halo_next) store_halo();
                return;
            }
            wn_u4 bh = wn_lds_rd<0>(aV), bl = wn_lds_rd<2048>(aV);
            wn_u4 wl[4], wh[4];
            wn_static_for<0, 4>([&](auto j_) {
                constexpr int J = decltype(j_)::value;
                wl[J] = wn_lds_rd<8192 + J * 1024>(aW);
                wh[J] = wn_lds_rd<J * 1024>(aW);
            });
            dma_w(s + 1, (ST & 1) ^ 1);        // (that buffer was last read in step s - 1, which every wave has left)
            asm volatile("" ::: "memory");     // the halo loads stay BEHIND the DMA pieces in the wave's vmcnt queue (counted wait below)
            if (ST == 0 && halo_next) gload_halo(kc + 1);
            wn_static_for<0, 4>([&](auto j_) {
                constexpr int J = decltype(j_)::value;
                wn_wait_lgkm<6 - 2 * J>();
                if constexpr (J == 0) { wn_touch(bh); wn_touch(bl); }
                wn_touch(wl[J]);
                wn_touch(wh[J]);
                acc[ST][J] = wn_mma(wl[J], bh, acc[ST][J]);
                acc[ST][J] = wn_mma(wh[J], bl, acc[ST][J]);
                acc[ST][J] = wn_mma(wh[J], bh, acc[ST][J]);
            });
            // own pieces of step s + 1 landed (the halo loads issued behind them in step 0 may stay in flight), then everybody's
            if (ST == 0 && halo_next) {
                if (has_coef) wn_wait_vm<WN_ITER + 2>(); else wn_wait_vm<WN_ITER>();
            } else {
                wn_wait_vm<0>();
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ST == 1 && halo_next) store_halo();   // sR was last read by transform(kc); the loads are two steps old
        });
    }

    // ---- epilogue: Y = A^T M A.  This wave holds M[pi][pj] for pj of parity ps: acc[2 pi] = M[pi][ps], acc[2 pi + 1] = M[pi][ps + 2]
    // A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]: column sums first (over the wave's two pj), then the rows.  The two position parities
    // of a (wn, wm) pair meet through LDS: wave ps finalises cout tiles j = 2 ps, 2 ps + 1 and hands the other two over.
    // (ps is wave-uniform; the parity is a template argument so that every array index below is static -- a runtime select between
    // register arrays goes through scratch memory)
    auto epilogue = [&](auto ps_) {
        constexpr int PS = decltype(ps_)::value;
        float z[4][2][2][4];   // [j][i][jj][e]: partial (over this wave's positions) of output pixel (i, jj) of the tile, cout tile j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float c0[4][4], c1[4][4];   // [pi][e]: sum over own pj of M[pi][pj] * AT[jj][pj], jj = 0 / 1
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float A_ = acc[2 * pi][j][e], B_ = acc[2 * pi + 1][j][e];
                    c0[pi][e] = PS == 0 ? A_ + B_ : A_;      // pj 0, 2: AT[0] = 1, 1   | pj 1, 3: AT[0] = 1, 0
                    c1[pi][e] = PS == 0 ? -B_ : A_ - B_;     // pj 0, 2: AT[1] = 0, -1  | pj 1, 3: AT[1] = 1, -1
                }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[j][0][0][e] = (c0[0][e] + c0[1][e]) + c0[2][e];
                z[j][0][1][e] = (c1[0][e] + c1[1][e]) + c1[2][e];
                z[j][1][0][e] = (c0[1][e] - c0[2][e]) - c0[3][e];
                z[j][1][1][e] = (c1[1][e] - c1[2][e]) - c1[3][e];
            }
        }
        float4* xch = (float4*)smem;   // [8 waves][8 float4][64 lanes] (all LDS reads of the main loop are behind the last barrier)
        constexpr int JS = 2 * (1 - PS), JF = 2 * PS;   // tiles handed over / finalised here
#pragma unroll
        for (int jj2 = 0; jj2 < 2; ++jj2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    xch[(wv * 8 + jj2 * 4 + i * 2 + c) * 64 + lane] =
                        make_float4(z[JS + jj2][i][c][0], z[JS + jj2][i][c][1], z[JS + jj2][i][c][2], z[JS + jj2][i][c][3]);
        __syncthreads();
        const int pw = wv ^ 4;   // the partner wave (same wn, wm, other parity)
        const int fr = lane & 15, fg = lane >> 4;
        const int ty = wm * 2 + (fr >> 3), tx = fr & 7;
        const size_t HW = (size_t)a.H * a.W;
        float* outb = a.out + (size_t)b * HW * a.Cout;
        const float* resb = a.res ? a.res + (size_t)b * HW * a.Cout : nullptr;
        float* red = (float*)(smem + 8 * 8 * 64 * 16);   // [2 tile groups][128 / 4 quads][2] behind the exchange area
#pragma unroll
        for (int jj2 = 0; jj2 < 2; ++jj2) {
            const int nl = (wn * 4 + JF + jj2) * 16 + fg * 4;   // first of this lane's 4 output channels inside the 128-channel block
            const int n = nb * 128 + nl;
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) bs = *(const float4*)(a.bias + n);
            float vals[2][2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float4 o4 = xch[(pw * 8 + jj2 * 4 + i * 2 + c) * 64 + lane];
                    const size_t o = ((size_t)(y0 + 2 * ty + i) * a.W + x0 + 2 * tx + c) * a.Cout + n;
                    float4 v = make_float4((z[JF + jj2][i][c][0] + o4.x) + bs.x, (z[JF + jj2][i][c][1] + o4.y) + bs.y,
                                           (z[JF + jj2][i][c][2] + o4.z) + bs.z, (z[JF + jj2][i][c][3] + o4.w) + bs.w);
                    if (resb) {
                        const float4 r = *(const float4*)(resb + o);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    *(float4*)(outb + o) = v;
                    vals[i][c][0] = v.x; vals[i][c][1] = v.y; vals[i][c][2] = v.z; vals[i][c][3] = v.w;
                }
            if (a.part) {
                // (sum, M2 about the partial's own mean) of the stored values: this lane's 4 pixels x 4 channels, then the 16 tiles of
                // the wave's tile group (fixed-order shuffles), then the two tile groups through LDS with the pairwise update of Chan et al.
                float ssum = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) ssum += (vals[i][c][0] + vals[i][c][1]) + (vals[i][c][2] + vals[i][c][3]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) ssum += __shfl_xor(ssum, o, 64);
                const float mean = ssum * (1.0f / 256.0f);
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float d0 = vals[i][c][0] - mean, d1 = vals[i][c][1] - mean, d2 = vals[i][c][2] - mean, d3 = vals[i][c][3] - mean;
                        m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) m2 += __shfl_xor(m2, o, 64);
                if (fr == 0) {
                    red[(wm * 32 + (nl >> 2)) * 2 + 0] = ssum;
                    red[(wm * 32 + (nl >> 2)) * 2 + 1] = m2;
                }
            }
        }
    };
    if (a.dbg & 16) return;
    if (ps == 0) epilogue(std::integral_constant<int, 0>{});
    else epilogue(std::integral_constant<int, 1>{});
    if (a.part) {
        __syncthreads();
        float* red = (float*)(smem + 8 * 8 * 64 * 16);
        if (t < 32) {
            const float s0 = red[t * 2], s1 = red[(32 + t) * 2];
            const float s = s0 + s1;
            const float mean = s * (1.0f / 512.0f);
            const float dm0 = s0 * (1.0f / 256.0f) - mean, dm1 = s1 * (1.0f / 256.0f) - mean;
            const float m2 = (red[t * 2 + 1] + 256.0f * dm0 * dm0) + (red[(32 + t) * 2 + 1] + 256.0f * dm1 * dm1);
            float* dst = a.part + (((size_t)b * a.ntiles + tile) * (a.Cout / 4) + nb * 32 + t) * 2;
            dst[0] = s;
            dst[1] = m2;
        }
    }
}

}  // namespace

// 3x3 convolution, stride 1, pad 1, in Winograd F(2x2, 3x3) form: same contract as lgen_conv_fused (fused GroupNorm coefficients /
// swish on the input, nearest-2x upsampling, bias, residual, statistics partials) for the shapes the form covers: H % 8 == 0,
// W % 16 == 0, Cin % 32 == 0, Cout % 128 == 0, NHWC output.  `u_frag`: the transformed weights G g G^T, (hi, lo)-split and
// fragment-packed [Cout/128][Cin/32][16][2][8][64 lanes] x 16 B (llamagen_amd/vq_engine.py: _WinoW).  LGEN_ERR_UNSUPPORTED for
// other shapes (the caller keeps lgen_conv_fused for them).
extern "C" int lgen_conv_wino(const float* x_nhwc, const float* gn_coef, int swish, const void* u_frag, const float* bias,
                              const float* res, float* out, float* stats_partial, int B, int H, int W, int Cin, int Cout,
                              int upsample, void* stream) {
    if (upsample < 0 || upsample > 1 || B < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return LGEN_ERR_BAD_ARG;
    if (H % 8 || W % 16 || Cin % 32 || Cout % 128) return LGEN_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    ConvWArgs a{x_nhwc, (const float2*)gn_coef, (const uint4*)u_frag, bias, res, out, stats_partial,
                H, W, Cin, Cout, upsample, swish ? 1 : 0, W / 16, (H / 8) * (W / 16), 0};
    if (const char* e = getenv("LGEN_WINO_ABLATE")) a.dbg = atoi(e);
    static unsigned long long attr_set_mask = 0;   // per device
    const int dev_i = lgen_cur_dev();
    if (!((attr_set_mask >> dev_i) & 1)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set_mask |= 1ull << dev_i;
    }
    hipLaunchKernelGGL(conv_wino_kernel, dim3(a.ntiles, Cout / 128, B), dim3(WN_NT), WN_LDS, (hipStream_t)stream, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}
