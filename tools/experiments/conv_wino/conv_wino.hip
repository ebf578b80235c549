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
// Shape of the kernel (second form; the first one -- a workgroup per tile, a step's weights shared through LDS behind a barrier --
// was correct and 1.3 x SLOWER than conv_fused: profiles/r05_attn_clamp_ab_and_wino_ablate.log says where the time went: 1.6 of
// 5.8 ms launch / prologue skeleton, 1.7 ms epilogue and 1.8 ms weight-DMA round trips, none of them overlapped with anything
// because one 157 KB workgroup fills a CU; the MFMAs themselves were 0.6 ms):
//   * PERSISTENT: one 512-thread workgroup per CU walks (image, 128-channel block, 8 x 16 pixel tile) items; the next item's first
//     halo chunk and weight piece are requested during the current item's last chunk, the output stores drain under the next
//     item's prologue.  An item is the tile of conv_fused.hip, so the statistics partials keep their layout.
//   * per 32-channel chunk: (1) the 10 x 18 halo pixels, loaded once as fp32 a chunk ahead, are normalised / activated in
//     registers and parked in LDS; (2) every (Winograd tile, 4-channel quad) item computes its 16 values B^T d B in registers,
//     splits them and writes MFMA B-operand fragments [position][hi|lo][tile group][k-slice][tile][8 ch]; (3) 16 position GEMMs
//     [128 cout x 32 cin] x [32 cin x 32 tiles]: wave (wn, pq) owns cout tiles 4 wn .. 4 wn + 3, BOTH tile groups and row pq of
//     the 4 x 4 positions; accumulators of all its positions stay in registers over the whole K loop (4 x 4 x 2 tiles = 128 VGPRs).
//   * weights are PRIVATE to a wave: no other wave reads the fragments of (its positions, its cout tiles), so every wave DMAs its
//     own 4 KiB per sub-step into its own double buffer and waits on its own vmcnt -- two workgroup barriers per chunk, none
//     inside the position GEMMs.
// Epilogue: A^T M A -- the column half in registers, the row half across the four waves of a cout half through LDS -- then bias,
// residual, float4 NHWC stores and the (sum, M2) partial of every (8 x 16 tile, 4-channel quad).
#include <cstdlib>
#include <type_traits>

#include "lgen_common.h"
#include "../../include/lgen.h"

namespace {

struct ConvWArgs {
    const float* x;      // [B][Hs][Ws][Cin] fp32 NHWC (Hs = H >> ups)
    const float2* coef;  // [B][Cin] (scale, shift) of the fused GroupNorm, or null
    const uint4* w;      // [Cout/128][Cin/32][8 sub-steps][8 waves][hi|lo][2 cout tiles][64 lanes] x 16 B (MFMA A-fragment order)
    const float* bias;   // [Cout] or null
    const float* res;    // NHWC like out, or null
    float* out;          // NHWC [B][H][W][Cout]
    float* part;         // [B][ntiles][Cout/4][2] partial (sum, M2) of the stored output, or null
    int H, W, Cin, Cout, ups, swish, tiles_x, ntiles, nitems, dbg;   // dbg: development ablation mask (LGEN_WINO_ABLATE), 0 in production
};

constexpr int WN_HC = 18, WN_NP = 10 * 18;           // halo columns / pixels of an 8 x 16 output tile
constexpr int WN_RS = 144;                           // bytes per halo pixel in LDS: 32 channels fp32 + 16 B (bank spread)
constexpr int WN_SV = 16 * 2 * 2 * 1024;             // transformed input: [16 positions][hi|lo][2 tile groups] x 1 KiB fragments
constexpr int WN_PW = 2 * 1024;                      // one sub-step's private weights of a wave: (hi | lo) fragment of ONE cout tile
constexpr int WN_RING = 4;                           // slots of a wave's private ring: three sub-steps of DMA in flight
constexpr int WN_SW = 8 * WN_RING * WN_PW;           // 8 waves
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

struct WnItem { int b, nb, tile, y0, x0; };

__global__ __launch_bounds__(WN_NT, 2) void conv_wino_kernel(ConvWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned char* sW = smem + WN_SV;
    unsigned char* sR = smem + WN_SV + WN_SW;
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wv & 1, pq = wv >> 1;          // cout half (tiles 4 wn .. 4 wn + 3) / row pi of the 4 x 4 positions
    const int nkc = a.Cin >> 5;
    const int Hs = a.H >> a.ups, Ws = a.W >> a.ups;
    const int per_img = a.ntiles * (a.Cout >> 7);
    const unsigned lds0 = (unsigned)(uintptr_t)(wn_lptr_t)smem;
    const bool has_coef = a.coef != nullptr;
    const int dbg = a.dbg;

    int item = blockIdx.x;
    if (item >= a.nitems) return;
    // development (dbg bit 5): shader-clock totals per phase of every wave of workgroup 0, written over part[0 .. 63] at the end
    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    const bool timing = (dbg & 32) != 0;
    auto lap = [&](int k) {
        if (timing) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm[k] += now - t_last;
            t_last = now;
        }
    };
    if (timing) t_last = __builtin_amdgcn_s_memtime();

    // ---- work items (image b, 128-channel block nb, 8 x 16 tile), tiles fastest: neighbours in x run at the same time ----
    auto decode_item = [&](int it_) {
        WnItem r;
        r.b = it_ / per_img;
        const int q_ = it_ - r.b * per_img;
        r.nb = q_ / a.ntiles;
        r.tile = q_ - r.nb * a.ntiles;
        r.y0 = (r.tile / a.tiles_x) * 8;
        r.x0 = (r.tile % a.tiles_x) * 16;
        return r;
    };

    // ---- halo staging items of this thread: it = t + i * 512 -> (halo pixel it >> 3, channel quad it & 7 = t & 7).  Addresses are
    // recomputed per chunk (a few dozen integer operations against a chunk's ~100 MFMAs per wave) instead of being held in VGPRs ----
    const int cq_s = t & 7;
    float4 raw[WN_ITER];
    unsigned rok = 0;    // bit i: pixel of raw[i] lies inside the image
    float4 cf0 = make_float4(1.f, 0.f, 1.f, 0.f), cf1 = cf0;
    auto gload_halo = [&](const WnItem& im, int kc) {
        const float* xb = a.x + (size_t)im.b * Hs * Ws * a.Cin + kc * 32 + cq_s * 4;
        rok = 0;
#pragma unroll
        for (int i = 0; i < WN_ITER; ++i) {
            const int it = t + i * WN_NT;
            const int P = (it >> 3) < WN_NP ? (it >> 3) : WN_NP - 1;
            const int hy = P / WN_HC, hx = P - hy * WN_HC;
            const int yy = im.y0 - 1 + hy, xx = im.x0 - 1 + hx;
            const bool ok = (it < WN_NP * 8) && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
            const int sy = ok ? (yy >> a.ups) : 0, sx = ok ? (xx >> a.ups) : 0;
            raw[i] = *(const float4*)(xb + (sy * Ws + sx) * a.Cin);
            rok |= ok ? (1u << i) : 0u;
        }
        if (has_coef) {
            const float4* c = (const float4*)(a.coef + (size_t)im.b * a.Cin + kc * 32 + cq_s * 4);
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
                *(float4*)(sR + (it >> 3) * WN_RS + cq_s * 16) = ((rok >> i) & 1) ? make_float4(f[0], f[1], f[2], f[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        // rows r0 .. r0+2 of the 4 x 4 input patch, r0 = hrow: hrow 0 -> (d0 - d2, d1 + d2), hrow 1 -> (d2 - d1, d1 - d3).  One output
        // row at a time (its two input rows are re-read from LDS): holding both rows' intermediates cost 20 more live registers
        // than the 256 two waves per SIMD leave.
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            // (row a, row b, sign) of t = R[a] +- R[b], relative to r0: hrow 0: (0, 2, -), (1, 2, +); hrow 1: (1, 0, -), (0, 2, -)
            const int ra = hrow == 0 ? rr : 1 - rr;
            const int rb = (hrow == 1 && rr == 0) ? 0 : 2;
            const bool plus = hrow == 0 && rr == 1;
            float4 tt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 Ra = *(const float4*)(rbase + (ra * WN_HC + c) * WN_RS);
                const float4 Rb = *(const float4*)(rbase + (rb * WN_HC + c) * WN_RS);
                tt[c] = plus ? f4add(Ra, Rb) : f4sub(Ra, Rb);
            }
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

    // ---- weights: private to the wave.  Sub-step ss = pj * 4 + jt: position 4 pq + pj, cout tile 4 wn + jt -- one (hi | lo) pair of
    // 1 KiB fragments nobody else reads, DMA'd FOUR sub-steps ahead into a four-slot ring (the slot of the pair whose fragments have just been
    // read into registers; 6 KiB per wave in flight: the second form's one-sub-step lookahead left every sub-step waiting out an L2 round trip).  Host layout [nb][kc][ss][wave][hi|lo] x 1 KiB. ----
    unsigned char* myW = sW + wv * (WN_RING * WN_PW);
    const unsigned aW0 = lds0 + WN_SV + wv * (WN_RING * WN_PW) + lane * 16;
    const uint4* wsrc = a.w + lane;
    auto dma_w = [&](int inb, int kc, int ss, int slot) {
        if (dbg & 4) return;
        const uint4* src = wsrc + ((((size_t)inb * nkc + kc) * 16 + ss) * 8 + wv) * (WN_PW / 16);
        __builtin_amdgcn_global_load_lds((wn_gptr_t)(src), (wn_lptr_t)(myW + slot * WN_PW), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((wn_gptr_t)(src + 64), (wn_lptr_t)(myW + slot * WN_PW + 1024), 16, 0, 0);
    };
    const unsigned aV0 = lds0 + pq * 4 * 4096 + lane * 16;   // + pj * 4096 (+ 1024: tile group 1) (+ 2048: lo plane)

    f32x4_t acc[4][4][2];   // [pj][cout tile of the wave's half][tile group]

    WnItem cur = decode_item(item);
    dma_w(cur.nb, 0, 0, 0);
    dma_w(cur.nb, 0, 1, 1);
    dma_w(cur.nb, 0, 2, 2);
    dma_w(cur.nb, 0, 3, 3);
    gload_halo(cur, 0);
    store_halo();

    // ---- epilogue of one item (PQ = the wave's position row, static so that every register-array index is static) ----
    auto epilogue = [&](auto pq_, const WnItem& im) {
        constexpr int PQ = decltype(pq_)::value;
        // column half of A^T M A over this wave's row: c[jj2] = sum_pj M[PQ][pj] * AT[jj2][pj], AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
        float c[2][4][2][4];   // [jj2][cout tile][tile group][e]
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float m0 = acc[0][jt][tg][e], m1 = acc[1][jt][tg][e], m2 = acc[2][jt][tg][e], m3 = acc[3][jt][tg][e];
                    c[0][jt][tg][e] = (m0 + m1) + m2;
                    c[1][jt][tg][e] = (m1 - m2) - m3;
                }
        // this wave finalises cout tile jt = PQ of its half (both tile groups): its 16 channels x the whole 8 x 16 pixel tile
        const int fr = lane & 15, fg = lane >> 4;
        const int n = im.nb * 128 + (wn * 4 + PQ) * 16 + fg * 4;   // first of this lane's 4 output channels
        const size_t HW = (size_t)a.H * a.W;
        float* outb = a.out + (size_t)im.b * HW * a.Cout;
        const float* resb = a.res ? a.res + (size_t)im.b * HW * a.Cout : nullptr;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) bs = *(const float4*)(a.bias + n);
        const int ty0 = fr >> 3, tx = fr & 7;
        const int obase = ((im.y0 + 2 * ty0) * a.W + im.x0 + 2 * tx) * a.Cout + n;   // pixel (0, 0) of tile (tile group 0); < 2^31 per image
        float4* xch = (float4*)sV;   // [wn][finaliser d][source index][jj2] x 64 lanes: 48 KiB per tile-group round
        float vals[2][2][2][4];
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            // this round's residual loads first: their latency runs under the exchange (tile group tg = tile rows 2 tg, 2 tg + 1)
            float4 rres[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj2 = 0; jj2 < 2; ++jj2)
                    rres[i][jj2] = resb ? *(const float4*)(resb + obase + ((4 * tg + i) * a.W + jj2) * a.Cout) : make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();   // sV free: the position GEMMs' reads (tg 0) / the previous round's reads (tg 1) are done
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (d != PQ) {
                    const int si = PQ < d ? PQ : PQ - 1;
#pragma unroll
                    for (int jj2 = 0; jj2 < 2; ++jj2)
                        xch[(((wn * 4 + d) * 3 + si) * 2 + jj2) * 64 + lane] =
                            make_float4(c[jj2][d][tg][0], c[jj2][d][tg][1], c[jj2][d][tg][2], c[jj2][d][tg][3]);
                }
            }
            __syncthreads();
            // row half: Y[i] = sum_pi AT[i][pi] c^(pi), pi ascending (fixed order); c^(PQ) is this wave's own
#pragma unroll
            for (int jj2 = 0; jj2 < 2; ++jj2) {
                float y0v[4] = {0.f, 0.f, 0.f, 0.f}, y1v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float cs[4];
                    if (s == PQ) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs[e] = c[jj2][PQ][tg][e];
                    } else {
                        const int si = s < PQ ? s : s - 1;
                        const float4 o4 = xch[(((wn * 4 + PQ) * 3 + si) * 2 + jj2) * 64 + lane];
                        cs[0] = o4.x; cs[1] = o4.y; cs[2] = o4.z; cs[3] = o4.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (s == 0) { y0v[e] = cs[e]; }
                        if (s == 1) { y0v[e] += cs[e]; y1v[e] = cs[e]; }
                        if (s == 2) { y0v[e] += cs[e]; y1v[e] -= cs[e]; }
                        if (s == 3) { y1v[e] -= cs[e]; }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* yv = i ? y1v : y0v;
                    const float4 r = rres[i][jj2];
                    const float4 v = make_float4((yv[0] + bs.x) + r.x, (yv[1] + bs.y) + r.y, (yv[2] + bs.z) + r.z, (yv[3] + bs.w) + r.w);
                    *(float4*)(outb + obase + ((4 * tg + i) * a.W + jj2) * a.Cout) = v;
                    vals[tg][i][jj2][0] = v.x; vals[tg][i][jj2][1] = v.y; vals[tg][i][jj2][2] = v.z; vals[tg][i][jj2][3] = v.w;
                }
            }
        }
        if (a.part) {
            // (sum, M2 about the partial's own mean) of the stored values of the whole tile x this lane group's 4-channel quad:
            // 8 pixels x 4 channels per lane, then the 16 lanes of the group in a fixed order
            float ssum = 0.f;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj2 = 0; jj2 < 2; ++jj2)
                        ssum += (vals[tg][i][jj2][0] + vals[tg][i][jj2][1]) + (vals[tg][i][jj2][2] + vals[tg][i][jj2][3]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) ssum += __shfl_xor(ssum, o, 64);
            const float mean = ssum * (1.0f / 512.0f);
            float m2 = 0.f;
#pragma unroll
            for (int tg = 0; tg < 2; ++tg)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj2 = 0; jj2 < 2; ++jj2) {
                        const float d0 = vals[tg][i][jj2][0] - mean, d1 = vals[tg][i][jj2][1] - mean;
                        const float d2 = vals[tg][i][jj2][2] - mean, d3 = vals[tg][i][jj2][3] - mean;
                        m2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) m2 += __shfl_xor(m2, o, 64);
            if (fr == 0) {
                float* dst = a.part + (((size_t)im.b * a.ntiles + im.tile) * (a.Cout / 4) + (n >> 2)) * 2;
                dst[0] = ssum;
                dst[1] = m2;
            }
        }
    };

    while (true) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[s][j][0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc[s][j][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        const int nitem = item + gridDim.x;
        const bool have_next = nitem < a.nitems;
        WnItem nxt = cur;
        for (int kc = 0; kc < nkc; ++kc) {
            // what follows this chunk: the next chunk of this item, or chunk 0 of the next item
            const bool last = kc + 1 == nkc;
            const bool more = !last || have_next;
            int n_nb = cur.nb, n_kc = kc + 1;
            if (last && have_next) {
                nxt = decode_item(nitem);
                n_nb = nxt.nb;
                n_kc = 0;
            }
            lap(7);
            __syncthreads();       // sR holds this chunk's halo; every wave has left the previous chunk's position GEMMs (sV may be rewritten)
            lap(0);
            if (more) gload_halo(last ? nxt : cur, n_kc);   // the NEXT chunk's halo: in flight under the transform and the first sub-steps
            if (!(dbg & 1)) transform();
            lap(1);
            __syncthreads();       // sV complete; sR is free again
            lap(2);
            // Operand reads run one sub-step ahead of the MFMAs (two register sets): without that a sub-step was an LDS round trip plus
            // six MFMAs, and two waves per SIMD hid only half of it (third form: 4.4 ms per 384 px conv against 4.6 for conv_fused).
            // Queue of this wave's in-order loads at the top of a chunk: pairs of sub-steps 0 .. 3 (requested by the previous chunk),
            // then the next chunk's halo loads (issued above), then whatever this chunk requests.
            wn_u4 wA[2][2], bB[4];   // [set][hi, lo] / one set [hi tg0, lo tg0, hi tg1, lo tg1] (re-read behind a position's last MFMAs)
            auto rd_a = [&](auto set_, int slot) {
                constexpr int S = decltype(set_)::value;
                const unsigned aW = aW0 + slot * WN_PW;
                wA[S][0] = wn_lds_rd<0>(aW);
                wA[S][1] = wn_lds_rd<1024>(aW);
            };
            auto rd_b = [&](int pj) {
                const unsigned aV = aV0 + pj * 4096;
                bB[0] = wn_lds_rd<0>(aV);
                bB[1] = wn_lds_rd<2048>(aV);
                bB[2] = wn_lds_rd<1024>(aV);
                bB[3] = wn_lds_rd<3072>(aV);
            };
            if (more) { if (has_coef) wn_wait_vm<6 + WN_ITER + 2>(); else wn_wait_vm<6 + WN_ITER>(); }
            else wn_wait_vm<6>();
            if (!(dbg & 2)) {
                rd_a(std::integral_constant<int, 0>{}, 0);
                rd_b(0);
            }
            wn_static_for<0, 16>([&](auto ss_) {
                constexpr int SS = decltype(ss_)::value, PJ = SS >> 2, JT = SS & 3, CS = SS & 1;
                constexpr bool NEXT_B = JT == 3 && SS < 15;          // the next sub-step starts a new position: its B fragments are read
                constexpr int NRD = SS < 15 ? 2 : 0;                 // behind this sub-step's MFMAs; A reads issued below, ahead of them
                if constexpr (SS < 15) {
                    // the pair of sub-step SS + 1 has landed: behind it the queue holds the pairs of SS + 2 and SS + 3 (and, up to
                    // sub-step 2, the halo loads: older than everything this chunk requests, so from sub-step 3 on they are in)
                    if (!more && SS >= 13) wn_wait_vm<0>();
                    else if (SS <= 2 && more) { if (has_coef) wn_wait_vm<4 + WN_ITER + 2>(); else wn_wait_vm<4 + WN_ITER>(); }
                    else wn_wait_vm<4>();
                }
                if (SS == 3) lap(3);
                if (SS == 3 && more) store_halo();   // next chunk's halo -> sR (its last reader, this chunk's transform, is behind the barrier)
                if (SS == 3) lap(4);
                if (!(dbg & 2)) {
                    if constexpr (SS < 15) rd_a(std::integral_constant<int, CS ^ 1>{}, (SS + 1) & 3);
                    wn_wait_lgkm<NRD>();     // this sub-step's operands (read one sub-step ago) are in registers
                }
                // the pair four sub-steps ahead goes into the slot this sub-step's pair has just left
                if (SS + 4 < 16) dma_w(cur.nb, kc, SS + 4, SS & 3);
                else if (more) dma_w(n_nb, n_kc, SS + 4 - 16, SS & 3);
                if (dbg & 2) return;
                wn_touch(wA[CS][0]); wn_touch(wA[CS][1]);
                if constexpr (JT == 0) { wn_touch(bB[0]); wn_touch(bB[1]); wn_touch(bB[2]); wn_touch(bB[3]); }
                acc[PJ][JT][0] = wn_mma(wA[CS][1], bB[0], acc[PJ][JT][0]);
                acc[PJ][JT][0] = wn_mma(wA[CS][0], bB[1], acc[PJ][JT][0]);
                acc[PJ][JT][0] = wn_mma(wA[CS][0], bB[0], acc[PJ][JT][0]);
                acc[PJ][JT][1] = wn_mma(wA[CS][1], bB[2], acc[PJ][JT][1]);
                acc[PJ][JT][1] = wn_mma(wA[CS][0], bB[3], acc[PJ][JT][1]);
                acc[PJ][JT][1] = wn_mma(wA[CS][0], bB[2], acc[PJ][JT][1]);
                if constexpr (NEXT_B) rd_b(PJ + 1);   // (the MFMAs above have issued: the registers are free)
            });
        }
        lap(5);
        if (!(dbg & 16)) {
            if (pq == 0) epilogue(std::integral_constant<int, 0>{}, cur);
            else if (pq == 1) epilogue(std::integral_constant<int, 1>{}, cur);
            else if (pq == 2) epilogue(std::integral_constant<int, 2>{}, cur);
            else epilogue(std::integral_constant<int, 3>{}, cur);
        }
        lap(6);
        if (!have_next) break;
        item = nitem;
        cur = nxt;
    }
    if (timing && blockIdx.x == 0 && lane == 0 && a.part) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a.part[wv * 8 + k] = (float)tm[k];
    }
}

}  // namespace

// 3x3 convolution, stride 1, pad 1, in Winograd F(2x2, 3x3) form: same contract as lgen_conv_fused (fused GroupNorm coefficients /
// swish on the input, nearest-2x upsampling, bias, residual, statistics partials) for the shapes the form covers: H % 8 == 0,
// W % 16 == 0, Cin % 32 == 0, Cout % 128 == 0, NHWC output.  `u_frag`: the transformed weights G g G^T, (hi, lo)-split and
// fragment-packed [Cout/128][Cin/32][8 sub-steps][8 waves][hi|lo][2 cout tiles][64 lanes] x 16 B (llamagen_amd/vq_engine.py:
// _ConvW.wino).  LGEN_ERR_UNSUPPORTED for other shapes (the caller keeps lgen_conv_fused for them).
extern "C" int lgen_conv_wino(const float* x_nhwc, const float* gn_coef, int swish, const void* u_frag, const float* bias,
                              const float* res, float* out, float* stats_partial, int B, int H, int W, int Cin, int Cout,
                              int upsample, void* stream) {
    if (upsample < 0 || upsample > 1 || B < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return LGEN_ERR_BAD_ARG;
    if (H % 8 || W % 16 || Cin % 32 || Cout % 128) return LGEN_ERR_UNSUPPORTED;
    if (B == 0) return 0;
    const int ntiles = (H / 8) * (W / 16);
    const long long nitems = (long long)B * (Cout / 128) * ntiles;
    if (nitems > 0x7fffffffLL) return LGEN_ERR_BAD_ARG;
    ConvWArgs a{x_nhwc, (const float2*)gn_coef, (const uint4*)u_frag, bias, res, out, stats_partial,
                H, W, Cin, Cout, upsample, swish ? 1 : 0, W / 16, ntiles, (int)nitems, 0};
    if (const char* e = getenv("LGEN_WINO_ABLATE")) a.dbg = atoi(e);
    static unsigned long long attr_set_mask = 0;   // per device
    const int dev_i = lgen_cur_dev();
    if (!((attr_set_mask >> dev_i) & 1)) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set_mask |= 1ull << dev_i;
    }
    const int n_cu = lgen_cu_count();   // one 157 KB workgroup per CU, walking its items
    const int grid = nitems < n_cu ? (int)nitems : n_cu;
    hipLaunchKernelGGL(conv_wino_kernel, dim3(grid), dim3(WN_NT), WN_LDS, (hipStream_t)stream, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}
