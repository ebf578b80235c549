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

// Round 3: the kernel is PERSISTENT ALONG N ("x-stationary").  A workgroup keeps its normalised activation fragments in
// registers and walks `passes` consecutive n-groups (NT 16-row tiles of N each): the activation / statistics loads, the RMSNorm
// VALU work and the workgroup's start-up latency are paid once per workgroup instead of once per n-group, and the weight loads
// of n-group g+1 are in flight while n-group g is reduced through LDS and stored.  (A second weight register set requested one
// whole n-group ahead was measured no faster and bimodal in round 3 and removed in round 4.)  Measured motivation (profiles/r03_sq_pmc.csv): waves of the one-pass form
// spend 58-63 % of their cycles parked on s_waitcnt / barriers with one workgroup per CU, i.e. the load phase and the compute
// phase of a CU never overlap.  passes == 1 is the round-2 kernel.
//
// Workgroup -> (n-group range, m-group): a 1-D grid, decoded so that the m-groups that read the SAME weights sit on the same XCD
// (block id b runs on XCD b % 8) in adjacent dispatch slots: one L2 fill serves all of them.
template <int MT, int NT, int CPW>
LGEN_DEV void np_load_w(uint4 (&A)[CPW][NT], const uint4* wbase, size_t wstride) {
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) A[c][j] = ldg_w(wbase + j * wstride + (size_t)c * 64);
}

// MFMAs of one n-group, `between()` (the caller's next weight request), cross-wave K reduction through LDS in a fixed order
// (wave 0, 1, 2, ...) and the fused epilogue (gemm_epilogue.h).  `red` is this pass's reduction buffer.
template <typename D, int MT, int NT, int EPI, int CPW, typename F>
LGEN_DEV void np_pass(const GemmArgs& a, const uint4 (&A)[CPW][NT], const uint4 (&B)[CPW][MT], float4* red, int w, int KW, int lane,
                      int nt0, int mt0, const int (&posr)[MT], F&& between) {
    uint4 aux[gemm_units<MT, NT, EPI>()];
    gemm_aux_prefetch<D, MT, NT, EPI>(a, aux, w, KW, lane, nt0, mt0, posr);
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
    between();
    gemm_reduce_epilogue<D, MT, NT, EPI>(a, acc, red, w, KW, lane, nt0, mt0, posr, aux);
}

// reduction buffers: two (ping-pong, one barrier per pass) when both fit the 160 KiB of LDS with 8 K-splitting waves
template <int MT, int NT>
constexpr bool np_red2() { return 2 * 8 * NT * MT <= 160; }

// MODE 0: one n-group per workgroup (the round-2 kernel); 1: `passes` n-groups, weights reloaded after each group's MFMAs
template <typename D, int MT, int NT, int EPI, int CPW, int MODE>
__global__ __launch_bounds__(512) void gemm_normpre_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    constexpr int TILES = NT * MT;
    constexpr bool RED2 = np_red2<MT, NT>();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KW = blockDim.x >> 6;  // KCH == KW * CPW (checked by the launcher)
    // block id -> (bx = n-group range, by = m-group), the m-groups of one range on one XCD
    const int gy = a.MTs / MT;
    const int ngroups = (a.N / 16) / NT;
    const int passes = MODE == 0 ? 1 : a.passes;
    const int gx = (ngroups + passes - 1) / passes;
    const int bid = blockIdx.x, slot = bid >> 3;
    const int by = slot % gy, bx = (slot / gy) * 8 + (bid & 7);
    if (bx >= gx) return;  // padding of the decoded grid (whole workgroup, before any barrier)
    const int g0 = bx * passes;
    const int np = (ngroups - g0) < passes ? (ngroups - g0) : passes;
    const int mt0 = by * MT;
    const int k0 = w * CPW;
    int posr[MT];
    load_row_pos<MT, EPI>(a, mt0, lane, posr);

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
    // 2. every weight chunk of this wave for the first n-group
    const size_t wstride = (size_t)a.KCH * 64;
    const size_t gstride = wstride * NT;                  // uint4s between consecutive n-groups
    const uint4* wbase = a.wp + (size_t)g0 * gstride + (size_t)k0 * 64 + lane;
    uint4 A0[CPW][NT];
    np_load_w<MT, NT, CPW>(A0, wbase, wstride);
    // 3. RMSNorm in registers (gpt.py:143-148), fixed-order statistics, while (2) is in flight
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        float s = sl[i].fast ? ssq_finish(sl[i]) : ssum[i];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float ri = 1.0f / sqrtf(s * a.inv_k + a.eps);
#pragma unroll
        for (int c = 0; c < CPW; ++c) B[c][i] = D::norm_chunk(B[c][i], ri, WN[c]);
    }
    // 4. the n-groups of this workgroup
    const size_t rstride = RED2 ? (size_t)KW * TILES * 64 : 0;
    if constexpr (MODE == 0) {
        np_pass<D, MT, NT, EPI, CPW>(a, A0, B, red, w, KW, lane, g0 * NT, mt0, posr, []() {});
    } else {
        for (int p = 0; p < np; ++p) {
            np_pass<D, MT, NT, EPI, CPW>(a, A0, B, red + (p & 1) * rstride, w, KW, lane, (g0 + p) * NT, mt0, posr, [&]() {
                if (p + 1 < np) np_load_w<MT, NT, CPW>(A0, wbase + (size_t)(p + 1) * gstride, wstride);  // registers just consumed
            });
            if (!RED2 && p + 1 < np) __syncthreads();  // next pass rewrites `red`
        }
    }
}

// shapes whose operand set does not fit 256 VGPRs or that leave objects in scratch (found by compiling everything once and
// reading csrc/gemm_normpre.usage); the multi-pass modes exist for the tile shapes the engine's heuristics can pick
template <int MT, int NT, int EPI, int CPW, int MODE>
constexpr bool np_spills() {
    if (MODE > 0) {
        if (!((MT == 1 && NT == 4) || (MT == 2 && NT == 2) || (MT == 2 && NT == 4) || (MT == 4 && NT == 2))) return true;
        if (MT == 4 && (EPI == EPI_QKV || CPW >= 5)) return true;
        if (MT == 2 && NT == 4 && EPI == EPI_QKV && CPW >= 5) return true;
        return false;
    }
    if (MT == 2 && NT == 1 && EPI == EPI_QKV) return true;  // 40 B of scratch; a tuning-only shape (fused qkv runs nt = 4)
    return (MT == 4 && CPW == 6) || (MT == 4 && NT == 4 && CPW == 5) || (MT == 4 && NT >= 2 && EPI == EPI_QKV && CPW >= 5) ||
           (MT == 4 && NT == 4 && EPI == EPI_QKV) || (MT == 2 && NT == 4 && EPI == EPI_QKV && CPW == 6);
}

template <int MT, int NT, int EPI, int CPW, int MODE>
static int launch_np2(const GemmArgs& a, int kw, hipStream_t st) {
    if constexpr (np_spills<MT, NT, EPI, CPW, MODE>()) {
        return LGEN_ERR_UNSUPPORTED;
    } else {
        const int passes = MODE == 0 ? 1 : a.passes;
        const int gy = a.MTs / MT, ngroups = (a.N / 16) / NT;
        const int gx = (ngroups + passes - 1) / passes;
        dim3 grid(8 * ((gx + 7) / 8) * gy);   // decoded in the kernel: m-groups of one n-range on one XCD, padding exits
        const size_t one = kw > 1 ? (size_t)kw * NT * MT * 64 * sizeof(float4) : 0;
        const size_t lds = (np_red2<MT, NT>() && MODE > 0) ? 2 * one : one;
        if (lds > 160 * 1024) return LGEN_ERR_UNSUPPORTED;
        auto kern = gemm_normpre_kernel<BF16, MT, NT, EPI, CPW, MODE>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(64 * kw), lds, st, a);
        LGEN_CHECK_LAUNCH();
        return 0;
    }
}

template <int MT, int NT, int EPI, int CPW>
static int launch_np(const GemmArgs& a, int kw, hipStream_t st) {
    if (a.passes > 1) {  // shapes without a multi-pass instantiation fall back to one n-group per workgroup
        const int rc = launch_np2<MT, NT, EPI, CPW, 1>(a, kw, st);
        if (rc != LGEN_ERR_UNSUPPORTED) return rc;
    }
    return launch_np2<MT, NT, EPI, CPW, 0>(a, kw, st);
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
