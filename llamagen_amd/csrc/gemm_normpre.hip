// RMSNorm-fused skinny GEMM, "normalise while the weights fly" form.
//
// Same operation as gemm_kernel<..., NORM = true> (gemm_skinny.hip): out = RMSNorm(x) . W^T with the
// qkv+RoPE+append / SwiGLU / rows epilogues, i.e. gpt.py:253-256 `attention(attention_norm(x))`,
// `feed_forward(ffn_norm(h))` and gpt.py:367-368 `output(norm(h))`.  The difference is the schedule.
// The measured cost of the NORM prologue in the ring-buffer kernel is ~4 us per GEMM, because every
// workgroup normalises the whole activation panel (M x K elements, ~5 VALU ops each) AFTER its operand
// loads have landed, serially in front of the MFMAs.  Here a wave's K range is small enough (CPW <= 6
// k-chunks, i.e. d <= 1536 with 8 K-splitting waves) to hold ALL of its operands in registers:
//   1. request the x chunks, the norm weights and the row statistics   (L2 / MALL resident)
//   2. request ALL weight chunks of the wave (non-temporal, from HBM)   <- the long latency
//   3. normalise x in registers while (2) is in flight
//   4. MFMAs, cross-wave reduction, epilogue (shared with gemm_skinny.hip)
// Loads retire in issue order, so step 3 waits only for the loads of step 1.
#include "gemm_epilogue.h"

template <typename D, int MT, int NT, int EPI, int CPW>
__global__ __launch_bounds__(512) void gemm_normpre_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    constexpr int TILES = NT * MT;
    constexpr int UNITS = EPI == EPI_SWIGLU ? (TILES / 2 > 0 ? TILES / 2 : 1) : TILES;
    constexpr int UPW = (UNITS + 1) / 2;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KW = blockDim.x >> 6;  // KCH == KW * CPW (checked by the launcher)
    const int nt0 = blockIdx.x * NT;
    const int mt0 = blockIdx.y * MT;
    const int k0 = w * CPW;
    int posr[MT];
    load_row_pos<MT, EPI>(a, mt0, lane, posr);

    const unsigned pf_token = prefetch_lines(a.pf, a.pf_bytes, (blockIdx.y * gridDim.x + blockIdx.x) * KW + w,
                                             gridDim.x * gridDim.y * KW, lane);
    // 1. activations, norm weights, row statistics
    uint4 B[CPW][MT], WN[CPW];
    const uint4* xbase = a.xp + (size_t)mt0 * 64 + lane;
    const size_t xstride = (size_t)a.MTs * 64;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
#pragma unroll
        for (int i = 0; i < MT; ++i) B[c][i] = xbase[(size_t)(k0 + c) * xstride + i * 64];
        WN[c] = a.nw[(size_t)(k0 + c) * 4 + (lane >> 4)];
    }
    SsqLoads<ssq_nv<MT>()> sl[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) sl[i] = ssq_issue<ssq_nv<MT>()>(a.ssq_in, a.parts, (mt0 + i) * 16 + (lane & 15), lane);
    float ssum[MT];  // rows too wide to hold their partials across the weight requests: reduce them now (one or two L2 round trips)
    if (!sl[0].fast) ssq_rows_now<MT>(a.ssq_in, a.parts, mt0, lane, ssum);
    // 2. every weight chunk of this wave + the epilogue's memory operands
    uint4 A[CPW][NT];
    const uint4* wbase = a.wp + ((size_t)nt0 * a.KCH) * 64 + lane;
    const size_t wstride = (size_t)a.KCH * 64;
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) A[c][j] = ldg_w(wbase + j * wstride + (size_t)(k0 + c) * 64);
    uint4 aux[UNITS];
#pragma unroll
    for (int q = 0; q < UNITS; ++q) aux[q] = make_uint4(0, 0, 0, 0);
    if constexpr (epi_has_aux<EPI>()) {
#pragma unroll
        for (int q = 0; q < UNITS; ++q) {
            const int u = w + q * KW;
            if (u < UNITS) {
                const int j = u / MT, i = u - j * MT;
                aux[q] = epi_prefetch<D, EPI>(a, nt0 + j, mt0 + i, lane, pick_pos<MT>(posr, i));
            }
        }
    }
    // 3. RMSNorm in registers (gpt.py:143-148), fixed-order statistics
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float s = sl[i].fast ? ssq_finish(sl[i]) : ssum[i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float ri = 1.0f / sqrtf(s * a.inv_k + a.eps);
#pragma unroll
        for (int c = 0; c < CPW; ++c) B[c][i] = D::norm_chunk(B[c][i], ri, WN[c]);
    }
    // 4. MFMAs
    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j][i] = D::mma(A[c][j], B[c][i], acc[j][i]);

    prefetch_retire(a.pf, pf_token);

    if (KW == 1) {
#pragma unroll
        for (int q = 0; q < UNITS; ++q) {
            if constexpr (EPI == EPI_SWIGLU) {
                const int jp = q / MT, i = q - jp * MT;
                epilogue<D, EPI>(a, nt0 + 2 * jp, mt0 + i, lane, acc[2 * jp][i], acc[(2 * jp + 1) % NT][i], aux[q], pick_pos<MT>(posr, i));
            } else {
                const int j = q / MT, i = q - j * MT;
                epilogue<D, EPI>(a, nt0 + j, mt0 + i, lane, acc[j][i], acc[j][i], aux[q], pick_pos<MT>(posr, i));
            }
        }
        return;
    }
    // cross-wave K reduction through LDS, fixed summation order (wave 0, 1, 2, ...)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x4_t v = acc[j][i];
            red[((size_t)w * TILES + j * MT + i) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    __syncthreads();
    auto rsum = [&](int t) {
        float4 s = red[(size_t)t * 64 + lane];
        for (int ww = 1; ww < KW; ++ww) {
            float4 p = red[((size_t)ww * TILES + t) * 64 + lane];
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        return f32x4_t{s.x, s.y, s.z, s.w};
    };
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
        const int u = w + q * KW;
        if (u < UNITS) {
            if constexpr (EPI == EPI_SWIGLU) {
                const int jp = u / MT, i = u - jp * MT;
                epilogue<D, EPI>(a, nt0 + 2 * jp, mt0 + i, lane, rsum((2 * jp) * MT + i), rsum((2 * jp + 1) * MT + i), aux[q], pick_pos<MT>(posr, i));
            } else {
                const int j = u / MT, i = u - j * MT;
                const f32x4_t v = rsum(u);
                epilogue<D, EPI>(a, nt0 + j, mt0 + i, lane, v, v, aux[q], pick_pos<MT>(posr, i));
            }
        }
    }
}

template <int MT, int NT, int EPI, int CPW>
static int launch_np(const GemmArgs& a, int kw, hipStream_t st) {
    // shapes whose operand set does not fit 256 VGPRs (they spill; found by compiling everything once):
    constexpr bool spills = (MT == 4 && CPW == 6) || (MT == 4 && NT == 4 && CPW == 5) ||
                            (MT == 4 && NT >= 2 && EPI == EPI_QKV && CPW >= 5) ||
                            (MT == 4 && NT == 4 && EPI == EPI_QKV) || (MT == 2 && NT == 4 && EPI == EPI_QKV && CPW == 6);
    if constexpr (spills) {
        return LGEN_ERR_UNSUPPORTED;
    } else {
        dim3 grid((a.N / 16) / NT, a.MTs / MT);
        const size_t lds = kw > 1 ? (size_t)kw * NT * MT * 64 * sizeof(float4) : 0;
        hipLaunchKernelGGL((gemm_normpre_kernel<BF16, MT, NT, EPI, CPW>), grid, dim3(64 * kw), lds, st, a);
        LGEN_CHECK_LAUNCH();
        return 0;
    }
}

template <int MT, int NT, int EPI>
static int dispatch_cpw(const GemmArgs& a, int kw, int cpw, hipStream_t st) {
    switch (cpw) {
        case 3: return launch_np<MT, NT, EPI, 3>(a, kw, st);
        case 4: return launch_np<MT, NT, EPI, 4>(a, kw, st);
        case 5: return launch_np<MT, NT, EPI, 5>(a, kw, st);
        case 6: return launch_np<MT, NT, EPI, 6>(a, kw, st);
        default: return LGEN_ERR_UNSUPPORTED;
    }
}

template <int EPI>
static int dispatch_tiles(const GemmArgs& a, int mt, int nt, int kw, int cpw, hipStream_t st) {
#define LGEN_CASE(MT_, NT_) if (mt == MT_ && nt == NT_) return dispatch_cpw<MT_, NT_, EPI>(a, kw, cpw, st);
    LGEN_CASE(1, 1) LGEN_CASE(1, 2) LGEN_CASE(1, 4)
    LGEN_CASE(2, 1) LGEN_CASE(2, 2) LGEN_CASE(2, 4)
    LGEN_CASE(4, 1) LGEN_CASE(4, 2) LGEN_CASE(4, 4)
#undef LGEN_CASE
    return LGEN_ERR_UNSUPPORTED;
}

// Entry used by gemm_skinny.hip's dispatcher: LGEN_ERR_UNSUPPORTED -> caller falls back to the
// ring-buffer NORM kernel.  bf16 only, KCH == kw * cpw with 3 <= cpw <= 6, kw <= 8.
int lgen_gemm_normpre_try(const GemmArgs& a, int epi, int dtype, int mt, int nt, int kw, hipStream_t st) {
    if (dtype != LGEN_BF16 || kw < 1 || kw > 8 || a.KCH % kw) return LGEN_ERR_UNSUPPORTED;
    if ((a.N / 16) % nt != 0 || a.MTs % mt != 0) return LGEN_ERR_UNSUPPORTED;
    if (epi == EPI_SWIGLU && (nt & 1)) return LGEN_ERR_UNSUPPORTED;
    const int cpw = a.KCH / kw;
    switch (epi) {
        case EPI_QKV: return dispatch_tiles<EPI_QKV>(a, mt, nt, kw, cpw, st);
        case EPI_SWIGLU: return dispatch_tiles<EPI_SWIGLU>(a, mt, nt, kw, cpw, st);
        case EPI_ROWS: return dispatch_tiles<EPI_ROWS>(a, mt, nt, kw, cpw, st);
        default: return LGEN_ERR_UNSUPPORTED;
    }
}
