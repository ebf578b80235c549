// Skinny weight-streaming GEMM for the decode step: out[M, N] = x[M, K] . W[N, K]^T with
// M = CFG-doubled batch (1..256), i.e. HBM-bound on the weight stream.
//
// Replaces the nn.Linear calls of the reference decode step
// (autoregressive/models/gpt.py:161-163 w1/w3/w2, :199-200 wqkv/wo, :287 output) and fuses
// what the reference does around them:
//   EPI_QKV    : wqkv + apply_rotary_emb(q, k) + KVCache.update   (gpt.py:214-226, 177-185, 420-430)
//   EPI_RES    : wo / w2 + residual add                           (gpt.py:238-240, 255-256)
//   EPI_SWIGLU : silu(w1 x) * w3 x on a row-interleaved w1||w3    (gpt.py:167)
//   EPI_ROWS   : row-major output (lm_head logits, gpt.py:367-368)
//   EPI_PACKED / EPI_GELU : plain / gelu(tanh) packed output (CaptionEmbedder MLP, gpt.py:118-131)
//
// Design (MI355X-first): both operands live in HBM in MFMA-fragment order, so
// every wave-level load is one fully coalesced 1 KiB `global_load_dwordx4`, straight into the
// VGPRs that feed `v_mfma_f32_16x16x32_bf16` (or 4 x `v_mfma_f32_16x16x4_f32` in fp32 mode):
// no LDS staging, no barriers in the main loop.  A workgroup owns NT 16-row tiles of N for all
// M rows; its KW waves split K between them and combine through LDS in a fixed order
// (deterministic, no atomics).  Weights are loaded non-temporally (read once per step).
#include "lgen_common.h"
#include "../../include/lgen.h"

enum { EPI_ROWS = 0, EPI_PACKED = 1, EPI_GELU = 2, EPI_RES = 3, EPI_SWIGLU = 4, EPI_QKV = 5 };

struct GemmArgs {
    const uint4* wp;     // packed weights [N/16][KCH][64] x 16 B
    const uint4* xp;     // packed activations [KCH][MTs][64] x 16 B
    void* out;           // EPI_ROWS: [M][N]; packed epilogues: XP of width N (or N/2 for SWIGLU); QKV: q rows
    void* kc;            // QKV: k cache
    void* vc;            // QKV: v cache
    const float* freqs;  // QKV: [P][hd/2][2] fp32 (cos, sin)
    const int* pos_ptr;  // QKV: device scalar, position of this token
    int N, KCH, MTs, M;
    int d, hd, hdp, H, S8;
};

LGEN_DEV float silu_f(float x) { return x / (1.0f + expf(-x)); }
LGEN_DEV float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// one 16x16 output tile: lane (g = lane>>4, r = lane&15) holds n = nt*16 + g*4 + {0..3}, m = mt*16 + r
template <typename D, int EPI>
LGEN_DEV void epilogue(const GemmArgs& a, int nt, int mt, int lane, f32x4_t v, f32x4_t v2) {
    const int r = lane & 15, g = lane >> 4;
    const int m = mt * 16 + r;
    const int n = nt * 16 + g * 4;
    float x0 = D::rnd(v[0]), x1 = D::rnd(v[1]), x2 = D::rnd(v[2]), x3 = D::rnd(v[3]);  // nn.Linear output rounding
    if constexpr (EPI == EPI_ROWS) {
        if (m < a.M) D::st4(a.out, (size_t)m * a.N + n, x0, x1, x2, x3);
    } else if constexpr (EPI == EPI_PACKED) {
        D::st4(a.out, D::xp_off(n, mt, r, a.MTs), x0, x1, x2, x3);
    } else if constexpr (EPI == EPI_GELU) {
        D::st4(a.out, D::xp_off(n, mt, r, a.MTs), gelu_tanh_f(x0), gelu_tanh_f(x1), gelu_tanh_f(x2), gelu_tanh_f(x3));
    } else if constexpr (EPI == EPI_RES) {
        size_t o = D::xp_off(n, mt, r, a.MTs);
        float h0, h1, h2, h3;
        D::ld4(a.out, o, h0, h1, h2, h3);
        D::st4(a.out, o, h0 + x0, h1 + x1, h2 + x2, h3 + x3);
    } else if constexpr (EPI == EPI_SWIGLU) {
        // nt is the w1 tile (even), v2 the matching w3 tile; output feature f = (nt/2)*16 + g*4
        float y0 = D::rnd(v2[0]), y1 = D::rnd(v2[1]), y2 = D::rnd(v2[2]), y3 = D::rnd(v2[3]);
        int f = (nt >> 1) * 16 + g * 4;
        D::st4(a.out, D::xp_off(f, mt, r, a.MTs),
               D::rnd(silu_f(x0)) * y0, D::rnd(silu_f(x1)) * y1, D::rnd(silu_f(x2)) * y2, D::rnd(silu_f(x3)) * y3);
    } else if constexpr (EPI == EPI_QKV) {
        if (m >= a.M) return;
        const int pos = *a.pos_ptr;
        const int sec = n / a.d;
        const int c = n - sec * a.d;
        const int head = c / a.hd;
        const int dd = c - head * a.hd;
        if (sec < 2) {  // 2-D RoPE on interleaved (even, odd) pairs, fp32, one rounding
            const float4 f = *(const float4*)(a.freqs + ((size_t)pos * (a.hd >> 1) + (dd >> 1)) * 2);
            float y0 = x0 * f.x - x1 * f.y, y1 = x1 * f.x + x0 * f.y;
            float y2 = x2 * f.z - x3 * f.w, y3 = x3 * f.z + x2 * f.w;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        }
        if (sec == 0) {
            D::st4(a.out, ((size_t)m * a.H + head) * a.hdp + dd, x0, x1, x2, x3);
        } else {
            void* cache = sec == 1 ? a.kc : a.vc;
            D::st4(cache, (((size_t)m * a.H + head) * a.S8 + pos) * a.hdp + dd, x0, x1, x2, x3);
        }
    }
}

// VGPR budget: accumulators + two register stages of operands; big tiles run <= 8 waves.
template <int MT, int NT>
constexpr int gemm_max_threads() { return (NT * MT * 4 + 2 * (NT + MT) * 4 + 24 > 112) ? 512 : 1024; }

template <typename D, int MT, int NT, int EPI>
__global__ __launch_bounds__((gemm_max_threads<MT, NT>())) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KW = blockDim.x >> 6;
    const int nt0 = blockIdx.x * NT;
    const int mt0 = blockIdx.y * MT;
    const int k0 = (int)(((long)a.KCH * w) / KW), k1 = (int)(((long)a.KCH * (w + 1)) / KW);

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const uint4* wbase = a.wp + ((size_t)nt0 * a.KCH) * 64 + lane;
    const uint4* xbase = a.xp + (size_t)mt0 * 64 + lane;
    const size_t wstride = (size_t)a.KCH * 64;
    const size_t xstride = (size_t)a.MTs * 64;

    uint4 A0[NT], B0[MT], A1[NT], B1[MT];
#define LGEN_LOAD(A, B, kk)                                                                   \
    {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) A[j] = ldg_nt(wbase + j * wstride + (size_t)(kk) * 64); \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) B[i] = xbase[(size_t)(kk) * xstride + i * 64];          \
    }
#define LGEN_MMA(A, B)                                                                        \
    {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                        \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[j][i] = D::mma(A[j], B[i], acc[j][i]); \
    }
    int k = k0;
    if (k < k1) {
        LGEN_LOAD(A0, B0, k);
        while (true) {
            if (k + 1 < k1) LGEN_LOAD(A1, B1, k + 1);
            LGEN_MMA(A0, B0);
            if (++k >= k1) break;
            if (k + 1 < k1) LGEN_LOAD(A0, B0, k + 1);
            LGEN_MMA(A1, B1);
            if (++k >= k1) break;
        }
    }
#undef LGEN_LOAD
#undef LGEN_MMA

    constexpr int TILES = NT * MT;
    if (KW == 1) {
        if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
            for (int j = 0; j < NT; j += 2)
#pragma unroll
                for (int i = 0; i < MT; ++i) epilogue<D, EPI>(a, nt0 + j, mt0 + i, lane, acc[j][i], acc[(j + 1) % NT][i]);
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) epilogue<D, EPI>(a, nt0 + j, mt0 + i, lane, acc[j][i], acc[j][i]);
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
    if constexpr (EPI == EPI_SWIGLU) {
        constexpr int UNITS = (NT / 2) * MT;
        for (int u = w; u < UNITS; u += KW) {
            int jp = u / MT, i = u - jp * MT;
            epilogue<D, EPI>(a, nt0 + 2 * jp, mt0 + i, lane, rsum((2 * jp) * MT + i), rsum((2 * jp + 1) * MT + i));
        }
    } else {
        for (int t = w; t < TILES; t += KW) {
            int j = t / MT, i = t - j * MT;
            f32x4_t v = rsum(t);
            epilogue<D, EPI>(a, nt0 + j, mt0 + i, lane, v, v);
        }
    }
}

template <typename D, int MT, int NT, int EPI>
static int launch(const GemmArgs& a, int kw, hipStream_t st) {
    dim3 grid((a.N / 16) / NT, a.MTs / MT);
    size_t lds = kw > 1 ? (size_t)kw * NT * MT * 64 * sizeof(float4) : 0;
    if (lds > 160 * 1024 || kw * 64 > gemm_max_threads<MT, NT>()) return LGEN_ERR_BAD_ARG;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<D, MT, NT, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((gemm_kernel<D, MT, NT, EPI>), grid, dim3(64 * kw), lds, st, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}

template <typename D, int EPI>
static int dispatch(const GemmArgs& a, int mt, int nt, int kw, hipStream_t st) {
    if ((a.N / 16) % nt != 0 || a.MTs % mt != 0 || kw < 1 || kw > 16) return LGEN_ERR_BAD_ARG;
    if (EPI == EPI_SWIGLU && (nt & 1)) return LGEN_ERR_BAD_ARG;
#define LGEN_CASE(MT_, NT_) if (mt == MT_ && nt == NT_) return launch<D, MT_, NT_, EPI>(a, kw, st);
    LGEN_CASE(1, 1) LGEN_CASE(1, 2) LGEN_CASE(1, 4)
    LGEN_CASE(2, 1) LGEN_CASE(2, 2) LGEN_CASE(2, 4)
    LGEN_CASE(4, 1) LGEN_CASE(4, 2) LGEN_CASE(4, 4)
    LGEN_CASE(8, 1) LGEN_CASE(8, 2)
#undef LGEN_CASE
    return LGEN_ERR_BAD_ARG;
}

template <int EPI>
static int dispatch_dt(const GemmArgs& a, int dtype, int mt, int nt, int kw, hipStream_t st) {
    if (dtype == LGEN_BF16) return dispatch<BF16, EPI>(a, mt, nt, kw, st);
    if (dtype == LGEN_F32) return dispatch<F32, EPI>(a, mt, nt, kw, st);
    return LGEN_ERR_BAD_ARG;
}

extern "C" int lgen_gemm(const void* wp, const void* xp, void* out, int M, int MTs, int N, int K, int epilogue_kind,
                         int dtype, int mt, int nt, int kw, void* stream) {
    const int kcsz = dtype == LGEN_BF16 ? 32 : 16;
    if (N % 16 || K % kcsz || M > MTs * 16) return LGEN_ERR_BAD_ARG;
    GemmArgs a{};
    a.wp = (const uint4*)wp; a.xp = (const uint4*)xp; a.out = out;
    a.N = N; a.KCH = K / kcsz; a.MTs = MTs; a.M = M;
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue_kind) {
        case LGEN_EPI_ROWS: return dispatch_dt<EPI_ROWS>(a, dtype, mt, nt, kw, st);
        case LGEN_EPI_PACKED: return dispatch_dt<EPI_PACKED>(a, dtype, mt, nt, kw, st);
        case LGEN_EPI_GELU: return dispatch_dt<EPI_GELU>(a, dtype, mt, nt, kw, st);
        case LGEN_EPI_RES: return dispatch_dt<EPI_RES>(a, dtype, mt, nt, kw, st);
        case LGEN_EPI_SWIGLU: return dispatch_dt<EPI_SWIGLU>(a, dtype, mt, nt, kw, st);
        default: return LGEN_ERR_BAD_ARG;
    }
}

extern "C" int lgen_gemm_qkv_rope(const void* wp, const void* xp, void* q_out, void* k_cache, void* v_cache,
                                  const float* freqs, const int* pos_ptr, int M, int MTs, int d, int n_head, int hd,
                                  int hdp, int S8, int dtype, int mt, int nt, int kw, void* stream) {
    const int kcsz = dtype == LGEN_BF16 ? 32 : 16;
    if (d % kcsz || (3 * d) % 16 || hd % 4 || d != n_head * hd || M > MTs * 16) return LGEN_ERR_BAD_ARG;
    GemmArgs a{};
    a.wp = (const uint4*)wp; a.xp = (const uint4*)xp; a.out = q_out; a.kc = k_cache; a.vc = v_cache;
    a.freqs = freqs; a.pos_ptr = pos_ptr;
    a.N = 3 * d; a.KCH = d / kcsz; a.MTs = MTs; a.M = M;
    a.d = d; a.hd = hd; a.hdp = hdp; a.H = n_head; a.S8 = S8;
    return dispatch_dt<EPI_QKV>(a, dtype, mt, nt, kw, (hipStream_t)stream);
}
