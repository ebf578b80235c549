// Wave-reduction kernels of the GPT decode step (HBM/latency-bound, no MFMA):
//   lgen_embed_pack   token / class-embedding gather -> fragment-packed residual stream
//                     (autoregressive/models/gpt.py:78-83 LabelEmbedder, :351 tok_embeddings)
//   lgen_rmsnorm      RMSNorm on the packed layout (gpt.py:137-148), two bf16 rounding points
//   lgen_attn_decode  KV-cached single-query attention over kv_len = pos+1 keys only
//                     (gpt.py:229-236 repeat_interleave + math-SDPA with causal_mask[:, pos])
#include "gemm_epilogue.h"

// ---------------------------------------------------------------------------------------------
// embedding gather: hp[k-chunk][mt][lane] <- table[idx[m]][k..]   (16 B per thread), plus
//   * ssq_out[row][LGEN_SSQ_STRIDE]: per-(row, 16 columns) sums of squares = the partials the first fused
//     RMSNorm (gemm_skinny.hip NORM prologue) sums in a fixed order
//   * state advance: (pos, step) += 1 before anything of this decode step reads them (the sampler
//     of the previous step is complete at this kernel's launch boundary)
// ---------------------------------------------------------------------------------------------
// one 16-byte chunk piece per lane -> this row's partial sums of squares, one per 16 columns (ssq[row][LGEN_SSQ_STRIDE]): a
// 2-byte chunk (KC = 32) yields two partials (lane groups {0,1} and {2,3}), an fp32 chunk (KC = 16) one
template <typename D>
LGEN_DEV void ssq_store(const uint4& v, float* ssq_out, int kc, int m, int lane) {
    float f[D::EPL];
    D::unpack(v, f);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < D::EPL; ++e) ss += f[e] * f[e];
    ss += __shfl_xor(ss, 16, 64);
    if constexpr (D::KC == 32) {
        if ((lane & 16) == 0) ssq_out[(size_t)m * LGEN_SSQ_STRIDE + 2 * kc + (lane >> 5)] = ss;
    } else {
        ss += __shfl_xor(ss, 32, 64);
        if (lane < 16) ssq_out[(size_t)m * LGEN_SSQ_STRIDE + kc] = ss;
    }
}

template <typename D>
__global__ __launch_bounds__(256) void embed_pack_kernel(const uint4* __restrict__ table, const int* __restrict__ idx,
                                                         uint4* __restrict__ hp, float* __restrict__ ssq_out,
                                                         int* __restrict__ state, int M, int MTs, int d, int rows,
                                                         const uint4* __restrict__ table0 = nullptr,
                                                         const int* __restrict__ idx0 = nullptr,
                                                         const int* __restrict__ row_pos = nullptr, int rows0 = 0) {
    const int KCH = d / D::KC;
    const int total = KCH * MTs * 64;  // a multiple of 64: every wave is one (kc, mt) chunk
    if (state && blockIdx.x == 0 && threadIdx.x == 0) {
        state[0] += 1;
        state[1] += 1;
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int lane = t & 63;
        const int mt = (t >> 6) % MTs;
        const int kc = (t >> 6) / MTs;
        const int m = mt * 16 + (lane & 15);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) {
            // continuous batching: a row at position 0 is a fresh request -> its conditioning embedding (table0[idx0[m]])
            const bool first = row_pos && row_pos[m] == 0;
            const uint4* tb = first ? table0 : table;
            const int nr = first ? rows0 : rows;
            int row = first ? idx0[m] : idx[m];
            row = row < 0 ? 0 : (row >= nr ? nr - 1 : row);
            v = tb[((size_t)row * d + kc * D::KC + (lane >> 4) * D::EPL) / D::EPL];
        }
        hp[t] = v;
        if (ssq_out) ssq_store<D>(v, ssq_out, kc, m, lane);
    }
}

extern "C" int lgen_embed_pack(const void* table, const int* idx, void* hp, float* ssq_out, int* state_advance, int M,
                               int MTs, int d, int rows, int dtype, void* stream) {
    if (M > MTs * 16 || d % 32 || (ssq_out && d / 16 > LGEN_SSQ_STRIDE)) return LGEN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LGEN_BF16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<BF16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)table, idx,
                           (uint4*)hp, ssq_out, state_advance, M, MTs, d, rows);
    } else if (dtype == LGEN_F16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<F16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)table, idx,
                           (uint4*)hp, ssq_out, state_advance, M, MTs, d, rows);
    } else if (dtype == LGEN_F32) {
        int total = (d / 16) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<F32>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)table, idx,
                           (uint4*)hp, ssq_out, state_advance, M, MTs, d, rows);
    } else {
        return LGEN_ERR_BAD_ARG;
    }
    LGEN_CHECK_LAUNCH();
    return 0;
}

// Continuous batching (autoregressive/serve/: every row of the step batch is its own request at its own position): row m
// takes cls_table[cond[m]] when row_pos[m] == 0 (the request's prefill token, gpt.py:348-349) and tok_table[cur_tok[m]]
// otherwise (gpt.py:351); the sampler advances row_pos.
extern "C" int lgen_embed_rows(const void* tok_table, const void* cls_table, const int* cur_tok, const int* cond,
                               const int* row_pos, void* hp, float* ssq_out, int M, int MTs, int d, int tok_rows, int cls_rows,
                               int dtype, void* stream) {
    if (M > MTs * 16 || d % 32 || !row_pos) return LGEN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LGEN_BF16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<BF16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)tok_table, cur_tok,
                           (uint4*)hp, ssq_out, (int*)nullptr, M, MTs, d, tok_rows, (const uint4*)cls_table, cond, row_pos, cls_rows);
    } else if (dtype == LGEN_F16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<F16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)tok_table, cur_tok,
                           (uint4*)hp, ssq_out, (int*)nullptr, M, MTs, d, tok_rows, (const uint4*)cls_table, cond, row_pos, cls_rows);
    } else if (dtype == LGEN_F32) {
        int total = (d / 16) * MTs * 64;
        hipLaunchKernelGGL(embed_pack_kernel<F32>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)tok_table, cur_tok,
                           (uint4*)hp, ssq_out, (int*)nullptr, M, MTs, d, tok_rows, (const uint4*)cls_table, cond, row_pos, cls_rows);
    } else {
        return LGEN_ERR_BAD_ARG;
    }
    LGEN_CHECK_LAUNCH();
    return 0;
}

// Row sums of squares of an already packed residual stream (t2i prefix rows enter the decode loop
// from the CaptionEmbedder MLP, not from an embedding gather): ssq_out[row][LGEN_SSQ_STRIDE], d/16 partials per row.
template <typename D>
__global__ __launch_bounds__(256) void ssq_pack_kernel(const uint4* __restrict__ hp, float* __restrict__ ssq_out, int total,
                                                       int MTs) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int lane = t & 63;
        const int mt = (t >> 6) % MTs;
        const int kc = (t >> 6) / MTs;
        ssq_store<D>(hp[t], ssq_out, kc, mt * 16 + (lane & 15), lane);
    }
}

extern "C" int lgen_ssq_pack(const void* hp, float* ssq_out, int MTs, int d, int dtype, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (d % 32 || !ssq_out || d / 16 > LGEN_SSQ_STRIDE) return LGEN_ERR_BAD_ARG;
    if (dtype == LGEN_BF16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(ssq_pack_kernel<BF16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)hp, ssq_out, total, MTs);
    } else if (dtype == LGEN_F16) {
        int total = (d / 32) * MTs * 64;
        hipLaunchKernelGGL(ssq_pack_kernel<F16>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)hp, ssq_out, total, MTs);
    } else if (dtype == LGEN_F32) {
        int total = (d / 16) * MTs * 64;
        hipLaunchKernelGGL(ssq_pack_kernel<F32>, dim3((total + 255) / 256), dim3(256), 0, st, (const uint4*)hp, ssq_out, total, MTs);
    } else {
        return LGEN_ERR_BAD_ARG;
    }
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone RMSNorm on packed activations (the decode loop uses the GEMM-fused form; this one
// serves the unfused call sites and the per-kernel parity tests).  One workgroup per 16-row
// m-tile; wave w owns k-chunks w, w+NW, ... (each a coalesced 1 KiB load kept in registers; all
// loads are issued unconditionally up front, out-of-range chunks re-read the last one and are
// masked), row sums of squares reduce over the 4 lane groups (xor 16, 32) and then over waves
// through LDS.
// ---------------------------------------------------------------------------------------------
template <typename D, int RMS_MAXC>
__global__ __launch_bounds__(1024) void rmsnorm_kernel(const uint4* __restrict__ hp, const void* __restrict__ w,
                                                       uint4* __restrict__ xn, int MTs, int d, float eps) {
    __shared__ float part[16][16];
    __shared__ float rinv[16];
    const int lane = threadIdx.x & 63, NW = blockDim.x >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt = blockIdx.x;
    const int KCH = d / D::KC;
    uint4 v[RMS_MAXC];
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        int kc = wv + i * NW;
        kc = kc < KCH ? kc : KCH - 1;
        v[i] = hp[((size_t)kc * MTs + mt) * 64 + lane];
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        float f[D::EPL];
        D::unpack(v[i], f);
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < D::EPL; ++e) s1 += f[e] * f[e];
        ss += (wv + i * NW < KCH) ? s1 : 0.f;
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lane < 16) part[wv][lane] = ss;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.f;
        for (int i = 0; i < NW; ++i) t += part[i][threadIdx.x];
        rinv[threadIdx.x] = 1.0f / sqrtf(t / (float)d + eps);
    }
    __syncthreads();
    const float ri = rinv[lane & 15];
#pragma unroll
    for (int i = 0; i < RMS_MAXC; ++i) {
        const int kc = wv + i * NW;
        if (kc < KCH) {
            float f[D::EPL], o[D::EPL];
            D::unpack(v[i], f);
            const int k = kc * D::KC + (lane >> 4) * D::EPL;
#pragma unroll
            for (int e = 0; e < D::EPL; ++e) o[e] = D::rnd(f[e] * ri) * D::ld(w, k + e);
            xn[((size_t)kc * MTs + mt) * 64 + lane] = D::pack(o);
        }
    }
}

extern "C" int lgen_rmsnorm(const void* hp, const void* weight, void* xnp, int MTs, int d, float eps, int dtype,
                            void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int kcsz = dtype != LGEN_F32 ? 32 : 16;
    if (d % kcsz) return LGEN_ERR_BAD_ARG;
    const int KCH = d / kcsz;
    int nw = 4;
    while (nw < 16 && nw * 4 < KCH) nw *= 2;  // ~4 chunks per wave, at most 8 (16 for very wide rows)
    if (nw * 16 < KCH) return LGEN_ERR_BAD_ARG;
    const bool wide = nw * 8 < KCH;
    if (wide && dtype != LGEN_F32) return LGEN_ERR_UNSUPPORTED;  // 16-bit rows wider than 4096: 16 chunks per wave would spill (no registry model)
#define LGEN_RMS(DT, C) hipLaunchKernelGGL((rmsnorm_kernel<DT, C>), dim3(MTs), dim3(64 * nw), 0, st, (const uint4*)hp, weight, \
                                           (uint4*)xnp, MTs, d, eps)
    if (dtype == LGEN_BF16) LGEN_RMS(BF16, 8);
    else if (dtype == LGEN_F16) LGEN_RMS(F16, 8);
    else if (dtype == LGEN_F32) { if (wide) LGEN_RMS(F32, 16); else LGEN_RMS(F32, 8); }
    else return LGEN_ERR_BAD_ARG;
#undef LGEN_RMS
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Decode attention.  One workgroup (4 waves) per (batch row, head).  K and V of that (b, h) are
// contiguous [S8][hdp] streams; a wave-wide 16-byte-per-lane load covers KPL = 64/LPK keys
// (LPK lanes per key).  Each wave walks its share of the keys in groups of ATT_CH loads of K and V
// with the NEXT group's loads already in flight (register double buffer); the first group is
// requested before the device-side position is even known (slots < S8 are always valid memory and
// masked afterwards), so q, pos and the first 2 x ATT_CH KiB of K/V arrive in one latency.  Each
// wave keeps an online-softmax state (m, l, acc) and the four states merge through LDS.  fp32 math
// from storage-dtype inputs, one rounding at the output (math-SDPA semantics, incl. ATen's
// sqrt(scale) pre-scaling of both q and k).  Only kv_len = pos+1 keys are read -- the reference
// reads (and copies) all S8 slots and masks.
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const void* q;       // [M][H][hdp]
    const void* kc;      // [B2][H][S8][hdp]
    const void* vc;
    void* out;           // packed [d/KC][MTs][64][EPL]
    const int* pos_ptr;
    int pos_stride;      // 0: every row at *pos_ptr; 1: row b at pos_ptr[b] (continuous batching)
    const unsigned char* mask;  // null (pure causal) or causal_mask [B2][S8][S8] bytes: row `pos` is read
    int H, hd, hdp, S8, MTs;
    float sf;            // sqrt(1/sqrt(hd))
    int kvs;             // elements between consecutive cache rows (hdp, or 2*hdp for an interleaved K|V slab)
    int mask_len;        // only keys < mask_len consult the mask row (t2i: the caption prefix); the rest is pure causal
};

// HPW (round 3): (batch row, head) pairs per workgroup.  At 256 chain rows the grid is 4096 (b, h) pairs; dispatching that many
// 128-thread workgroups is 9 us of a 52 us average launch (kv_len 1: 9.0 us at 256 rows against 4.6 us at 64), so wide chains put
// 2 or 4 heads of one row into a workgroup (same kv_len: the waves stay balanced across the one barrier).
template <typename D, int LPK, int ATT_CH, int ATT_NW, int HPW>
__global__ __launch_bounds__(64 * ATT_NW * HPW, (ATT_CH <= 2 ? 4 : 2)) void attn_decode_kernel(AttnArgs a) {
    constexpr int KPL = 64 / LPK;            // keys per wave-load
    constexpr int EPL = D::EPL;
    constexpr int GK = ATT_CH * KPL;         // keys per group
    __shared__ float s_m[HPW][ATT_NW], s_l[HPW][ATT_NW];
    __shared__ float s_acc[HPW][ATT_NW][LPK * EPL];
    const int lane = threadIdx.x & 63;
    const int wvall = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hg = wvall / ATT_NW, wv = wvall - hg * ATT_NW;   // head of this wave inside the workgroup, wave inside the head
    const int bh = blockIdx.x * HPW + hg;
    const int b = bh / a.H, h = bh - b * a.H;
    const int part = lane % LPK, kin = lane / LPK;
    const size_t rowbase = ((size_t)b * a.H + h) * a.S8;
    const int lpr = a.hdp / EPL;             // 16-byte pieces per key row (== LPK)
    const int rpr = a.kvs / EPL;             // 16-byte pieces between consecutive rows of one cache
    const uint4* kp = (const uint4*)a.kc + rowbase * rpr + part;
    const uint4* vp = (const uint4*)a.vc + rowbase * rpr + part;
    // Row clamp of the key loads.  Until the position is known: the last slot of the slab (the first group is requested before the
    // position load resolves).  From then on: the last VISIBLE key, so that the lanes of a partly filled last group re-read one
    // line instead of fetching up to GK - 1 rows nobody looks at (round 5: on average 15.5 of 288.5 keys per launch, 5 % of the
    // kernel's HBM bytes -- the live PMC ratio 1.07 at positions 64 / 288 / 575 was exactly this).
    int smax = a.S8 - 1;

    uint4 k0[ATT_CH], v0[ATT_CH], k1[ATT_CH], v1[ATT_CH];
#define ATT_LOAD(KB, VB, g)                                                 \
    {                                                                       \
        _Pragma("unroll") for (int j = 0; j < ATT_CH; ++j) {                \
            int kk = (g) * GK + j * KPL + kin;                              \
            kk = kk < smax ? kk : smax;                                     \
            KB[j] = ldg_nt(kp + (size_t)kk * rpr);                          \
            VB[j] = ldg_nt(vp + (size_t)kk * rpr);                          \
        }                                                                   \
    }
    // group g of this wave: g = wv, wv + NW, ...; first group requested before pos is known
    int g = wv;
    const uint4 qv = ((const uint4*)a.q)[((size_t)b * a.H + h) * lpr + part];
    ATT_LOAD(k0, v0, g);
    const int pos = a.pos_ptr[b * a.pos_stride];
    const int kvlen = pos + 1;
    const int ngroups = (kvlen + GK - 1) / GK;
    smax = pos < smax ? pos : smax;
    float qf[EPL];
    D::unpack(qv, qf);
#pragma unroll
    for (int e = 0; e < EPL; ++e) qf[e] *= a.sf;
    const unsigned char* pm = a.mask ? a.mask + ((size_t)b * a.S8 + pos) * a.S8 : nullptr;

    float m_run = -1e30f, l_run = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;

#define ATT_COMPUTE(KB, VB, g)                                                                     \
    {                                                                                              \
        float s[ATT_CH];                                                                           \
        float tmax = -1e30f;                                                                       \
        _Pragma("unroll") for (int j = 0; j < ATT_CH; ++j) {                                       \
            const int key = (g) * GK + j * KPL + kin;                                              \
            float kf[EPL];                                                                         \
            D::unpack(KB[j], kf);                                                                  \
            float dot = 0.f;                                                                       \
            _Pragma("unroll") for (int e = 0; e < EPL; ++e) dot = fmaf(qf[e], kf[e] * a.sf, dot);  \
            dot = group_sum<LPK>(dot);                                                             \
            bool vis = key < kvlen;                                                                \
            if (pm && vis && key < a.mask_len) vis = pm[key] != 0;                                 \
            s[j] = vis ? dot : -1e30f;                                                             \
            tmax = fmaxf(tmax, s[j]);                                                              \
        }                                                                                          \
        tmax = across_groups<LPK>(tmax, [](float x, float y) { return fmaxf(x, y); });              \
        const float m_new = fmaxf(m_run, tmax);                                                    \
        const float scale = D::fexp(m_run - m_new);                                                \
        l_run *= scale;                                                                            \
        _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] *= scale;                           \
        _Pragma("unroll") for (int j = 0; j < ATT_CH; ++j) {                                       \
            const float p = s[j] > -1e29f ? D::fexp(s[j] - m_new) : 0.f;                           \
            l_run += p;                                                                            \
            float vf[EPL];                                                                         \
            D::unpack(VB[j], vf);                                                                  \
            _Pragma("unroll") for (int e = 0; e < EPL; ++e) acc[e] = fmaf(p, vf[e], acc[e]);       \
        }                                                                                          \
        m_run = m_new;                                                                             \
    }
    while (g < ngroups) {
        if (g + ATT_NW < ngroups) ATT_LOAD(k1, v1, g + ATT_NW);
        ATT_COMPUTE(k0, v0, g);
        g += ATT_NW;
        if (g >= ngroups) break;
        if (g + ATT_NW < ngroups) ATT_LOAD(k0, v0, g + ATT_NW);
        ATT_COMPUTE(k1, v1, g);
        g += ATT_NW;
    }
#undef ATT_LOAD
#undef ATT_COMPUTE
    // combine the KPL key groups of this wave (lanes with equal `part`)
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
        l_run += __shfl_xor(l_run, o, 64);
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (lane < LPK) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) s_acc[hg][wv][lane * EPL + e] = acc[e];
    }
    if (lane == 0) { s_m[hg][wv] = m_run; s_l[hg][wv] = l_run; }
    __syncthreads();
    for (int t = wv * 64 + lane; t < a.hd; t += 64 * ATT_NW) {
        float M = s_m[hg][0];
#pragma unroll
        for (int i = 1; i < ATT_NW; ++i) M = fmaxf(M, s_m[hg][i]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int i = 0; i < ATT_NW; ++i) {
            const float f = D::fexp(s_m[hg][i] - M);
            L += s_l[hg][i] * f;
            o += s_acc[hg][i][t] * f;
        }
        o = o / L;
        D::st(a.out, D::xp_off(h * a.hd + t, b >> 4, b & 15, a.MTs), o);
    }
}

// Round 6: the 4 lane-group sums of a statistics row, computed ONCE per row instead of once per consumer workgroup.  A fused-norm
// consumer (gemm_tile.hip prologue, ssq_rows_now) sums a row's `parts` partials as 4 lane-group sums s_g = sum of the partials q = g,
// g + 4, g + 8, ... in ascending order, then (s_0 + s_1) + (s_2 + s_3) by two cross-lane adds.  For `parts` that are not a multiple
// of 16 (GPT-3B: d / 16 = 200) that loop is 50 dependent-batch loads per wave, repeated by every one of the 400-728 workgroups of a
// launch: 25 of the 99 us of GPT-3B's wqkv at 512 rows and 38 of the 140 us of w1||w3 (profiles/r06_tile_ablations_3b.log, mask 16).
// This kernel writes s_0 .. s_3 as a 4-partial row; a consumer given parts = 4 then reads s_g alone (0.f + s_g = s_g) and finishes
// with the same two cross-lane adds: the SAME bits, whatever `parts` was.  One thread per row, the row's float4s loaded in
// batches of 32 (independent loads), added component-wise in ascending order (component g of float4 k is partial 4k + g).
__global__ __launch_bounds__(64) void ssq_group4_kernel(const float* __restrict__ ssq_in, float* __restrict__ ssq_out, int rows, int parts) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    const float4* p = (const float4*)(ssq_in + (size_t)row * LGEN_SSQ_STRIDE);
    const int n4 = parts >> 2;   // launcher: parts % 4 == 0
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k0 = 0; k0 < n4; k0 += 32) {   // <= 2 round trips per row (parts <= 256)
        float4 v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = p[k0 + j < n4 ? k0 + j : n4 - 1];
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (k0 + j < n4) { s0 += v[j].x; s1 += v[j].y; s2 += v[j].z; s3 += v[j].w; }
    }
    *(float4*)(ssq_out + (size_t)row * LGEN_SSQ_STRIDE) = make_float4(s0, s1, s2, s3);
}

extern "C" int lgen_ssq_group4(const float* ssq_in, float* ssq_out, int rows, int parts, void* stream) {
    if (!ssq_in || !ssq_out || ssq_in == ssq_out || rows < 1 || parts < 4 || parts % 4 || parts > LGEN_SSQ_STRIDE) return LGEN_ERR_BAD_ARG;
    hipLaunchKernelGGL(ssq_group4_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, ssq_in, ssq_out, rows, parts);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---- persistent form (round 4, variant 8) ---------------------------------------------------------------------------------
// Same arithmetic per (batch row, head) as attn_decode_kernel with one wave per item (the wave-level online softmax of variants
// 4 / 5: every group of CH x KPL keys updates one wave-uniform running maximum, so the result does not depend on how many waves
// share the chip), different schedule: a fixed grid of NWV-wave workgroups, ONE per CU, whose waves each walk whole (b, h) items
// gw, gw + total_waves, ... as ONE flat stream of K/V groups -- the loads of the next group (of the next item, at an item
// boundary) are in flight while the current group is reduced, 2 x 2 x CH KiB per wave.  Why (measured, round 4):
//   * at 256 rows attn_decode_kernel is 2048 workgroups of 4 waves that fill every SIMD to 5 waves: ~9 us of a 52 us launch are
//     dispatch + the cold first round trip of every workgroup, and
//   * while such a grid drains, a GEMM / conv workgroup of ANOTHER chain (2 waves on all four SIMDs + > 100 KB of LDS at once)
//     cannot be placed -- small workgroups keep taking the slots that free up -- so chains in flight barely overlap
//     (three 256-row chains: 105 img/s against 85 with one).  Eight resident waves per CU at <= 128 VGPRs leave 256 registers per
//     SIMD and all of the LDS to whatever else is in flight.
// No LDS, no barriers; one (pos, kv_len) for all rows (generate()); per-row positions keep attn_decode_kernel.
template <typename D, int LPK, int CH, int NWV>
__global__ __launch_bounds__(64 * NWV) void attn_decode_persist_kernel(AttnArgs a, int items, int total_waves) {
    constexpr int KPL = 64 / LPK;            // keys per wave-load
    constexpr int EPL = D::EPL;
    constexpr int GK = CH * KPL;             // keys per group
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * NWV + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (gw >= items) return;
    const int part = lane % LPK, kin = lane / LPK;
    const int lpr = a.hdp / EPL;             // 16-byte pieces per key row (== LPK)
    const int rpr = a.kvs / EPL;             // 16-byte pieces between consecutive rows of one cache
    int smax = a.S8 - 1;                    // row clamp of the key loads: the last slot until the position is known, then the last visible key (see attn_decode_kernel)
    const uint4* const dummy = (const uint4*)a.q + part;   // where the look-ahead loads past the last item point (finite, L2-hot)

    uint4 k0[CH], v0[CH], k1[CH], v1[CH];
    auto load = [&](uint4 (&KB)[CH], uint4 (&VB)[CH], int item, int g) {
        const bool live = item < items;      // wave-uniform; loads stay unconditional so that the waits stay counted
        const uint4* kp = live ? (const uint4*)a.kc + (size_t)item * a.S8 * rpr + part : dummy;
        const uint4* vp = live ? (const uint4*)a.vc + (size_t)item * a.S8 * rpr + part : dummy;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            int kk = g * GK + j * KPL + kin;
            kk = kk < smax ? kk : smax;
            const size_t o = live ? (size_t)kk * rpr : 0;
            KB[j] = ldg_nt(kp + o);
            VB[j] = ldg_nt(vp + o);
        }
    };
    // the first group and q of the first item are requested before the position is known
    load(k0, v0, gw, 0);
    uint4 qv = ((const uint4*)a.q)[(size_t)gw * lpr + part];
    const int pos = *a.pos_ptr;
    const int kvlen = pos + 1;
    const int ngroups = (kvlen + GK - 1) / GK;
    smax = pos < smax ? pos : smax;
    int li = gw, lg = 0;                      // load cursor: the group after the one just requested
    auto advance = [&](int& item, int& g) {
        if (++g == ngroups) { g = 0; item += total_waves; }
    };
    advance(li, lg);

    int ci = gw, cg = 0;                      // compute cursor
    float qf[EPL], m_run = -1e30f, l_run = 0.f, acc[EPL];
    uint4 qn = qv, qraw = qv;
    // 16-bit storage (round 6): q . k as packed dot products on the RAW operands (v_dot2c_f32_bf16 / v_dot2_f32_f16: 4 instructions
    // per 8 elements instead of 8 unpacks + 8 multiplies + 8 fmas), scaled once by sf^2 -- the products of two 8- / 11-bit
    // mantissas are exact in fp32, so this differs from the reference's sum of fl(q sf) * fl(k sf) (ATen's sqrt(scale) on both operands)
    // by fp32 round-off only, like any other summation order.  The kernel spent 44 % of its wave cycles issuing VALU
    // (profiles/r06_sq_pmc.csv) -- cycles the other chain's GEMM waves on the same SIMDs wait for.
    const float sf2 = a.sf * a.sf;
    auto begin_item = [&]() {
        if constexpr (D::ESZ == 2) {
            qraw = qn;
        } else {
            D::unpack(qn, qf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) qf[e] *= a.sf;
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
        m_run = -1e30f; l_run = 0.f;
        const int nxt = ci + total_waves;     // q of the following item: one whole item ahead
        qn = ((const uint4*)a.q)[(size_t)(nxt < items ? nxt : ci) * lpr + part];
    };
    begin_item();
    auto compute = [&](const uint4 (&KB)[CH], const uint4 (&VB)[CH]) {
        const unsigned char* pm = a.mask ? a.mask + ((size_t)(ci / a.H) * a.S8 + pos) * a.S8 : nullptr;
        float s[CH];
        float tmax = -1e30f;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int key = cg * GK + j * KPL + kin;
            float dot = 0.f;
            if constexpr (D::ESZ == 2) {
                dot = D::dot8(KB[j], qraw) * sf2;
            } else {
                float kf[EPL];
                D::unpack(KB[j], kf);
#pragma unroll
                for (int e = 0; e < EPL; ++e) dot = fmaf(qf[e], kf[e] * a.sf, dot);
            }
            dot = group_sum<LPK>(dot);
            bool vis = key < kvlen;
            if (pm && vis && key < a.mask_len) vis = pm[key] != 0;
            s[j] = vis ? dot : -1e30f;
            tmax = fmaxf(tmax, s[j]);
        }
        tmax = across_groups<LPK>(tmax, [](float x, float y) { return fmaxf(x, y); });
        const float m_new = fmaxf(m_run, tmax);
        const float scale = D::fexp(m_run - m_new);
        l_run *= scale;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] *= scale;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float p = s[j] > -1e29f ? D::fexp(s[j] - m_new) : 0.f;
            l_run += p;
            float vf[EPL];
            D::unpack(VB[j], vf);
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
        }
        m_run = m_new;
    };
    auto end_group = [&]() {                  // returns false when this wave has no item left
        if (++cg < ngroups) return true;
        // item done: combine the KPL key groups of the wave (lanes with equal `part`), normalise, store
        auto add = [](float x, float y) { return x + y; };
        l_run = across_groups<LPK>(l_run, add);
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = across_groups<LPK>(acc[e], add);
        if (lane < LPK) {
            const int b = ci / a.H, h = ci - b * a.H;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int t = lane * EPL + e;
                if (t < a.hd) D::st(a.out, D::xp_off(h * a.hd + t, b >> 4, b & 15, a.MTs), acc[e] / l_run);
            }
        }
        cg = 0;
        ci += total_waves;
        if (ci >= items) return false;
        begin_item();
        return true;
    };
    while (true) {
        load(k1, v1, li, lg);
        advance(li, lg);
        compute(k0, v0);
        if (!end_group()) break;
        load(k0, v0, li, lg);
        advance(li, lg);
        compute(k1, v1);
        if (!end_group()) break;
    }
}

// Kernel variants (explicit `variant` argument of lgen_attn_decode; -1 = the library's choice by shape):
//   (K/V loads per buffer, waves per (b, h) workgroup): 0 = (4, 4); 1 = (2, 4); 2 = (2, 2): 128-thread workgroups, lowest fixed cost at
//   <= 128 rows (4.4 us at kv_len 1 vs 6.1 us with 4 waves, same 6.4 TB/s incremental rate); 3 = (4, 2); 4 = (2, 1); 5 = (4, 1);
//   6 / 7 (round 3): (2, 2) with 2 / 4 heads of a row per workgroup (fewer workgroups to dispatch at >= 256 rows);
//   8 .. 13 (round 4): the persistent form, (loads per buffer, waves per workgroup) = 8: (4, 8); 9: (4, 8) x two workgroups per CU;
//   10: (4, 4); 11: (2, 8); 12: (2, 4); 13: (1, 8); 14: (3, 4) -- one position for all rows only.
// K/V loads are non-temporal (ldg_nt): every cache row is read once per step, and keeping 75-150 MB per layer out of the
// memory-side cache leaves the (chain-shared) weights there: 26.5 vs 28.5 us at kv_len 576, 71.6 vs 68.6 img/s.  Fixed at
// compile time: a per-load run-time choice costs the compiler its count of loads in flight (vmcnt(0) before every compute).
static int attn_decode_impl(const void* q, const void* k_cache, const void* v_cache, void* out_packed,
                            const int* pos_ptr, int pos_stride, const unsigned char* mask, int mask_len, int B2, int MTs,
                            int n_head, int hd, int hdp, int S8, int kv_row_stride, int dtype, int variant_arg, void* stream) {
    AttnArgs a{q, k_cache, v_cache, out_packed, pos_ptr, pos_stride, mask, n_head, hd, hdp, S8, MTs, 0.f,
               kv_row_stride > 0 ? kv_row_stride : hdp, mask_len > 0 ? mask_len : S8};
    if (variant_arg < -1 || variant_arg > 14) return LGEN_ERR_BAD_ARG;
    a.sf = sqrtf(1.0f / sqrtf((float)hd));
    hipStream_t st = (hipStream_t)stream;
    const int epl = dtype != LGEN_F32 ? 8 : 4;
    if (dtype != LGEN_BF16 && dtype != LGEN_F32 && dtype != LGEN_F16) return LGEN_ERR_BAD_ARG;
    if (hdp % epl || hd > hdp || B2 > MTs * 16 || S8 < 1 || a.kvs < (hd + epl - 1) / epl * epl || a.kvs % epl) return LGEN_ERR_BAD_ARG;
    const int lpk = hdp / epl;
    // default variant 2; chains of >= 256 rows put two heads of a row into a workgroup (variant 6: 49.8 vs 50.7 us average launch
    // at 256 rows, 27.3 vs 27.0 at 128: tools/attn_sweep.py, profiles/r03_attn_sweep.log)
    // library's choice (variant_arg == -1): variant 2 below 256 rows; chains of >= 256 rows at one position take the persistent form with 4 waves x 16 KiB in flight per
    // CU (variant 10).  Alone it is ~6 % slower than variant 6 (54.3 vs 51.1 us per average launch at 256 rows: one wave per SIMD
    // has nothing to hide its own VALU latency behind), but with two chains in flight the bench gains 8 % (110.1 -> 119.0 img/s at
    // 512-row chains, gpurun_out/attn_ab3.log): 64 KB per CU in flight instead of 160 leaves the HBM queue short enough for the other
    // chain's latency-bound kernels, and 4 resident waves leave every SIMD room for them (tools/overlap_probe.py: the GEMM chain
    // hidden under an attention chain 0.30 -> 0.57).  Per-row positions (continuous batching): variant 6 (two heads per workgroup).
    int variant = variant_arg;
    if (variant < 0) variant = B2 >= 256 ? (pos_stride == 0 ? 10 : (n_head % 2 == 0 ? 6 : 2)) : 2;
    if (variant >= 8 && pos_stride == 0) {
        // persistent form, (K/V loads per buffer CH, waves per workgroup) = 8: (4, 8), one workgroup per CU; 9: (4, 8) x two per CU;
        // 10: (4, 4); 11: (2, 8); 12: (2, 4); 13: (1, 8).  Bytes in flight per CU = waves x 4 x CH KiB: what the HBM queue holds
        // beyond the bandwidth-delay product (~10 MB chip-wide) only adds latency for every kernel that shares the chip.
        const int n_cu = lgen_cu_count();
        const int items = B2 * n_head;
        const int nwv = (variant == 10 || variant == 12 || variant == 14) ? 4 : 8;
        const int ch = variant == 14 ? 3 : (variant <= 10 ? 4 : (variant <= 12 ? 2 : 1));
        int wgs = n_cu * (variant == 9 ? 2 : 1);
        if (wgs * nwv > items) wgs = (items + nwv - 1) / nwv;
        const int total_waves = wgs * nwv;
#define LGEN_ATTP(DT, L)                                                                                                   \
        do {                                                                                                               \
            if (ch == 4 && nwv == 8) hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 4, 8>), dim3(wgs), dim3(512), 0, st, a, items, total_waves); \
            else if (ch == 4) hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 4, 4>), dim3(wgs), dim3(256), 0, st, a, items, total_waves);        \
            else if (ch == 3) hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 3, 4>), dim3(wgs), dim3(256), 0, st, a, items, total_waves);        \
            else if (ch == 2 && nwv == 8) hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 2, 8>), dim3(wgs), dim3(512), 0, st, a, items, total_waves); \
            else if (ch == 2) hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 2, 4>), dim3(wgs), dim3(256), 0, st, a, items, total_waves);        \
            else hipLaunchKernelGGL((attn_decode_persist_kernel<DT, L, 1, 8>), dim3(wgs), dim3(512), 0, st, a, items, total_waves);                     \
        } while (0)
        if (dtype == LGEN_BF16 && lpk == 8) LGEN_ATTP(BF16, 8);
        else if (dtype == LGEN_BF16 && lpk == 16) LGEN_ATTP(BF16, 16);
        else if (dtype == LGEN_F16 && lpk == 8) LGEN_ATTP(F16, 8);
        else if (dtype == LGEN_F16 && lpk == 16) LGEN_ATTP(F16, 16);
        else if (dtype == LGEN_F32 && lpk == 16) LGEN_ATTP(F32, 16);
        else if (dtype == LGEN_F32 && lpk == 32) LGEN_ATTP(F32, 32);
        else return LGEN_ERR_BAD_ARG;
#undef LGEN_ATTP
        LGEN_CHECK_LAUNCH();
        return 0;
    }
    const int hpw = variant == 6 ? 2 : (variant == 7 ? 4 : 1);
    if (variant >= 8) return LGEN_ERR_UNSUPPORTED;   // per-row positions: the persistent form has one kv_len for all rows
    if (n_head % hpw) return LGEN_ERR_BAD_ARG;
    const int nw = variant >= 6 ? 2 : (variant >= 4 ? 1 : (variant >= 2 ? 2 : 4));
    dim3 grid(B2 * n_head / hpw), block(64 * nw * hpw);
#define LGEN_ATT(DT, L)                                                                              \
    do {                                                                                             \
        if (variant == 1) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 2, 4, 1>), grid, block, 0, st, a);      \
        else if (variant == 2) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 2, 2, 1>), grid, block, 0, st, a); \
        else if (variant == 3) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 4, 2, 1>), grid, block, 0, st, a); \
        else if (variant == 4) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 2, 1, 1>), grid, block, 0, st, a); \
        else if (variant == 5) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 4, 1, 1>), grid, block, 0, st, a); \
        else if (variant == 6) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 2, 2, 2>), grid, block, 0, st, a); \
        else if (variant == 7) hipLaunchKernelGGL((attn_decode_kernel<DT, L, 2, 2, 4>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((attn_decode_kernel<DT, L, 4, 4, 1>), grid, block, 0, st, a);                          \
    } while (0)
    if (dtype == LGEN_BF16 && lpk == 8) LGEN_ATT(BF16, 8);
    else if (dtype == LGEN_BF16 && lpk == 16) LGEN_ATT(BF16, 16);
    else if (dtype == LGEN_F16 && lpk == 8) LGEN_ATT(F16, 8);
    else if (dtype == LGEN_F16 && lpk == 16) LGEN_ATT(F16, 16);
    else if (dtype == LGEN_F32 && lpk == 16) LGEN_ATT(F32, 16);
    else if (dtype == LGEN_F32 && lpk == 32) LGEN_ATT(F32, 32);
    else return LGEN_ERR_BAD_ARG;
#undef LGEN_ATT
    LGEN_CHECK_LAUNCH();
    return 0;
}

extern "C" int lgen_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out_packed,
                                const int* pos_ptr, const unsigned char* mask, int mask_len, int B2, int MTs,
                                int n_head, int hd, int hdp, int S8, int kv_row_stride, int dtype, int variant, void* stream) {
    return attn_decode_impl(q, k_cache, v_cache, out_packed, pos_ptr, 0, mask, mask_len, B2, MTs, n_head, hd, hdp, S8,
                            kv_row_stride, dtype, variant, stream);
}

extern "C" int lgen_attn_decode_rows(const void* q, const void* k_cache, const void* v_cache, void* out_packed,
                                     const int* row_pos, const unsigned char* mask, int mask_len, int B2, int MTs,
                                     int n_head, int hd, int hdp, int S8, int kv_row_stride, int dtype, int variant, void* stream) {
    return attn_decode_impl(q, k_cache, v_cache, out_packed, row_pos, 1, mask, mask_len, B2, MTs, n_head, hd, hdp, S8,
                            kv_row_stride, dtype, variant, stream);
}

// ---------------------------------------------------------------------------------------------
// Prefill of a conditioning prefix (t2i: T = cls_token_num caption tokens, gpt.py:348-349 / generate.py:77-86
// with the emb_masks folded into causal_mask, generate.py:154-163): all T positions of all B2 rows go
// through each layer at once (rows r = t * B2 + b of the packed activations) instead of T decode steps.
//   lgen_rope_append_prefill : packed qkv rows -> RoPE(q, k) at position t (gpt.py:220-226, 420-430),
//                              q rows [R][H][hdp], K/V appended at cache slot t of (b, h)
//   lgen_attn_prefill        : masked causal attention of the prefix onto itself (gpt.py:229-236), fp32
//                              math-SDPA semantics, output packed for the wo GEMM
// T <= 128 (the 120-token caption prefix): K/V of one (b, h) staged whole in LDS as fp32; longer sequences
// take the key-tiled kernel below.  Plain VALU kernels (an evaluation path, not the sampling loop).
// ---------------------------------------------------------------------------------------------
template <typename D>
__global__ __launch_bounds__(256) void rope_append_prefill_kernel(const uint4* __restrict__ qkvp, void* __restrict__ qrows,
                                                                  void* __restrict__ kc, void* __restrict__ vc,
                                                                  const float* __restrict__ freqs, int R, int B2, int MTs, int d,
                                                                  int H, int hd, int hdp, int S8, int kvs, int pos0) {
    constexpr int EPL = D::EPL;
    const int KCH3 = 3 * d / D::KC;
    const long long total = (long long)KCH3 * MTs * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const int mt = (int)((idx >> 6) % MTs);
        const int kcx = (int)((idx >> 6) / MTs);
        const int r = mt * 16 + (lane & 15);
        if (r >= R) continue;
        const int t = r / B2, b = r - t * B2;
        float f[EPL];
        D::unpack(qkvp[idx], f);
        const int n0 = kcx * D::KC + (lane >> 4) * EPL;
#pragma unroll
        for (int e = 0; e < EPL; e += 2) {
            const int n = n0 + e;
            const int sec = n / d, c = n - sec * d;
            const int head = c / hd, dd = c - head * hd;
            float x0 = f[e], x1 = f[e + 1];
            if (sec < 2) {  // interleaved (even, odd) pairs, fp32, one rounding at the store
                const float2 cs = *(const float2*)(freqs + ((size_t)(pos0 + t) * (hd >> 1) + (dd >> 1)) * 2);
                rope_pair(x0, x1, cs.x, cs.y);
            }
            if (sec == 0) {
                D::st(qrows, ((size_t)r * H + head) * hdp + dd, x0);
                D::st(qrows, ((size_t)r * H + head) * hdp + dd + 1, x1);
            } else {
                void* cache = sec == 1 ? kc : vc;
                const size_t o = (((size_t)b * H + head) * S8 + pos0 + t) * kvs + dd;
                D::st(cache, o, x0);
                D::st(cache, o + 1, x1);
            }
        }
    }
}

extern "C" int lgen_rope_append_prefill(const void* qkv_packed, void* q_rows, void* k_cache, void* v_cache, const float* freqs,
                                        int R, int B2, int MTs, int d, int n_head, int hd, int hdp, int S8, int kv_row_stride,
                                        int pos0, int dtype, void* stream) {
    const int kcsz = dtype != LGEN_F32 ? 32 : 16;
    if ((3 * d) % kcsz || d != n_head * hd || hd % 2 || R > MTs * 16 || B2 < 1 || R % B2) return LGEN_ERR_BAD_ARG;
    const int kvs = kv_row_stride > 0 ? kv_row_stride : hdp;
    {   // same key-row contract as lgen_attn_decode: whole 16-byte pieces, at least the head's valid elements apart
        const int epl = dtype != LGEN_F32 ? 8 : 4;
        if (kvs % epl || kvs < (hd + epl - 1) / epl * epl) return LGEN_ERR_BAD_ARG;
    }
    const long long total = (long long)(3 * d / kcsz) * MTs * 64;
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LGEN_BF16)
        hipLaunchKernelGGL(rope_append_prefill_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const uint4*)qkv_packed, q_rows, k_cache,
                           v_cache, freqs, R, B2, MTs, d, n_head, hd, hdp, S8, kvs, pos0);
    else if (dtype == LGEN_F16)
        hipLaunchKernelGGL(rope_append_prefill_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const uint4*)qkv_packed, q_rows, k_cache,
                           v_cache, freqs, R, B2, MTs, d, n_head, hd, hdp, S8, kvs, pos0);
    else if (dtype == LGEN_F32)
        hipLaunchKernelGGL(rope_append_prefill_kernel<F32>, dim3(blocks), dim3(256), 0, st, (const uint4*)qkv_packed, q_rows, k_cache,
                           v_cache, freqs, R, B2, MTs, d, n_head, hd, hdp, S8, kvs, pos0);
    else
        return LGEN_ERR_BAD_ARG;
    LGEN_CHECK_LAUNCH();
    return 0;
}

#define PF_MAXT 128
template <typename D>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const void* __restrict__ qrows, const void* __restrict__ kc,
                                                           const void* __restrict__ vc, void* __restrict__ out,
                                                           const unsigned char* __restrict__ mask, int T, int B2, int MTs, int H,
                                                           int hd, int hdp, int S8, int kvs, float sf) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = hd + 1;                       // padded row stride: lanes walk rows
    float* Ks = sm;                              // [T][ld], pre-scaled by sf
    float* Vs = Ks + (size_t)T * ld;             // [T][ld]
    float* Ps = Vs + (size_t)T * ld;             // [4 waves][PF_MAXT]
    float* Qs = Ps + 4 * PF_MAXT;                // [4 waves][hd], pre-scaled by sf
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < T * hd; i += 256) {
        const int s = i / hd, dd = i - s * hd;
        const size_t o = (((size_t)b * H + h) * S8 + s) * kvs + dd;
        Ks[s * ld + dd] = D::ld(kc, o) * sf;
        Vs[s * ld + dd] = D::ld(vc, o);
    }
    __syncthreads();
    for (int t = wv; t < T; t += 4) {
        const int r = t * B2 + b;
        for (int dd = lane; dd < hd; dd += 64) Qs[wv * hd + dd] = D::ld(qrows, ((size_t)r * H + h) * hdp + dd) * sf;
        __builtin_amdgcn_wave_barrier();
        const unsigned char* mrow = mask ? mask + ((size_t)b * S8 + t) * S8 : nullptr;
        float sc[2];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int s = lane + u * 64;
            float v = -INFINITY;
            if (s < T && s <= t && (!mrow || mrow[s])) {
                float dot = 0.f;
                for (int dd = 0; dd < hd; ++dd) dot = fmaf(Qs[wv * hd + dd], Ks[s * ld + dd], dot);
                v = dot;
            }
            sc[u] = v;
            mx = fmaxf(mx, v);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float p = sc[u] > -INFINITY ? expf(sc[u] - mx) : 0.f;
            sc[u] = p;
            sum += p;
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (lane + u * 64 < PF_MAXT) Ps[wv * PF_MAXT + lane + u * 64] = sc[u] / sum;
        __builtin_amdgcn_wave_barrier();
        for (int dd = lane; dd < hd; dd += 64) {
            float acc = 0.f;
            for (int s = 0; s <= t; ++s) acc = fmaf(Ps[wv * PF_MAXT + s], Vs[s * ld + dd], acc);
            D::st(out, D::xp_off(h * hd + dd, r >> 4, r & 15, MTs), acc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Any-length variant (T > PF_MAXT: full-sequence teacher-forced forward, gpt.py:341-346 + is_causal SDPA of
// gpt.py:232-236): one workgroup per (b, h, tile of `qt` query rows); K/V stream through LDS in tiles of
// PFT_KT keys (fp32, scaled like the short kernel) with the running (max, sum, accumulator) of every query
// row kept in LDS -- the usual online-softmax recurrence, fp32 throughout, one rounding at the store.
#define PFT_KT 128
template <typename D>
__global__ __launch_bounds__(256) void attn_prefill_tiled_kernel(const void* __restrict__ qrows, const void* __restrict__ kc,
                                                                 const void* __restrict__ vc, void* __restrict__ out,
                                                                 const unsigned char* __restrict__ mask, int T, int B2, int MTs,
                                                                 int H, int hd, int hdp, int S8, int kvs, float sf, int qt) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = hd + 1;
    float* Ks = sm;                              // [PFT_KT][ld], pre-scaled by sf
    float* Vs = Ks + (size_t)PFT_KT * ld;        // [PFT_KT][ld]
    float* Qs = Vs + (size_t)PFT_KT * ld;        // [qt][hd], pre-scaled by sf
    float* Os = Qs + (size_t)qt * hd;            // [qt][hd] running sum of p * v
    float* Ps = Os + (size_t)qt * hd;            // [4 waves][PFT_KT]
    float* Ms = Ps + 4 * PFT_KT;                 // [qt] running max
    float* Ls = Ms + qt;                         // [qt] running sum of p
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int q0 = blockIdx.y * qt;
    const int qn = min(qt, T - q0);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < qt * hd; i += 256) {
        const int ti = i / hd, dd = i - ti * hd;
        float q = 0.f;
        if (ti < qn) q = D::ld(qrows, ((size_t)((size_t)(q0 + ti) * B2 + b) * H + h) * hdp + dd) * sf;
        Qs[i] = q;
        Os[i] = 0.f;
    }
    if (tid < qt) {
        Ms[tid] = -INFINITY;
        Ls[tid] = 0.f;
    }
    const int kend = q0 + qn;  // causal: keys 0 .. kend-1 can matter to this tile
    for (int k0 = 0; k0 < kend; k0 += PFT_KT) {
        const int kn = min(PFT_KT, kend - k0);
        __syncthreads();  // previous tile consumed (first pass: Q / O / M / L staged)
        for (int i = tid; i < kn * hd; i += 256) {
            const int s = i / hd, dd = i - s * hd;
            const size_t o = (((size_t)b * H + h) * S8 + k0 + s) * kvs + dd;
            Ks[s * ld + dd] = D::ld(kc, o) * sf;
            Vs[s * ld + dd] = D::ld(vc, o);
        }
        __syncthreads();
        for (int ti = wv; ti < qn; ti += 4) {
            const int t = q0 + ti;
            if (k0 > t) continue;  // the whole tile lies in this row's future (wave-uniform)
            const unsigned char* mrow = mask ? mask + ((size_t)b * S8 + t) * S8 : nullptr;
            const float m_old = Ms[ti], l_old = Ls[ti];
            float sc[2];
            float mx = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int s = lane + u * 64;
                const int key = k0 + s;
                float v = -INFINITY;
                if (s < kn && key <= t && (!mrow || mrow[key])) {
                    float dot = 0.f;
                    for (int dd = 0; dd < hd; ++dd) dot = fmaf(Qs[ti * hd + dd], Ks[s * ld + dd], dot);
                    v = dot;
                }
                sc[u] = v;
                mx = fmaxf(mx, v);
            }
            mx = wave_max(mx);
            const float m_new = fmaxf(m_old, mx);
            if (!(m_new > -INFINITY)) continue;  // nothing visible so far (wave-uniform)
            const float alpha = m_old > -INFINITY ? expf(m_old - m_new) : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float p = sc[u] > -INFINITY ? expf(sc[u] - m_new) : 0.f;
                Ps[wv * PFT_KT + lane + u * 64] = p;
                sum += p;
            }
            sum = wave_sum(sum);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nk = min(kn, t - k0 + 1);
            for (int dd = lane; dd < hd; dd += 64) {
                float acc = Os[ti * hd + dd] * alpha;
                for (int s = 0; s < nk; ++s) acc = fmaf(Ps[wv * PFT_KT + s], Vs[s * ld + dd], acc);
                Os[ti * hd + dd] = acc;
            }
            if (lane == 0) {
                Ms[ti] = m_new;
                Ls[ti] = l_old * alpha + sum;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // rows ti = wv (mod 4) were only ever touched by this wave
    for (int ti = wv; ti < qn; ti += 4) {
        const int r = (q0 + ti) * B2 + b;
        const float l = Ls[ti];
        for (int dd = lane; dd < hd; dd += 64) D::st(out, D::xp_off(h * hd + dd, r >> 4, r & 15, MTs), Os[ti * hd + dd] / l);
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA form of the prefill attention (bf16 storage): QK^T and PV on v_mfma_f32_16x16x32_bf16, flash-style.
// One workgroup = (b, h, 64 queries); wave w owns 16 of them.  Per 32-key step:
//   S^T[key][q] = K . Q^T      A = K rows straight from the cache (16 B per lane), B = Q (registers), fp32 accumulate;
//                              the two operand roundings of ATen's math path ((q*sf).(k*sf)) become one fp32 scale by sf^2
//   online softmax per query   a lane holds 4 keys x 2 tiles of one query column; max / sum across the 4 lane groups
//   O^T[d][q] += V^T . P       A = V^T from an LDS tile written transposed ([d][key-slot], the only layout an MFMA can
//                              contract over keys with), B = P as (hi, lo) bf16 -- two MFMA passes keep the fp32 softmax
//                              weights to 2^-17, so the output carries one bf16 rounding like math-SDPA's (gpt.py:232-236)
// Causal + optional per-row key mask exactly like the VALU kernels below.  fp32 storage keeps those kernels.
// ---------------------------------------------------------------------------------------------
template <int HDP>
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(const uint16_t* __restrict__ qrows, const uint16_t* __restrict__ kc,
                                                                const uint16_t* __restrict__ vc, void* __restrict__ out,
                                                                const unsigned char* __restrict__ mask, int T, int B2, int MTs,
                                                                int H, int hd, int S8, int kvs, float scale) {
    constexpr int KCH = HDP / 32, DT = HDP / 16, VLD = 40;  // V^T rows: 32 key slots + 8 pad (bank spread), bf16
    __shared__ __attribute__((aligned(16))) uint16_t vt[HDP * VLD];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int q0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r = lane & 15;
    const int tq = q0 + wv * 16 + r;                 // this lane's query (column r of the wave's tiles)
    const bool qok = tq < T;
    const size_t kvbase = ((size_t)b * H + h) * S8;
    // Q as the MFMA B operand: lane (g, r) holds query r, k-slice g*8.. of every 32-wide chunk
    uint4 qf[KCH];
#pragma unroll
    for (int c = 0; c < KCH; ++c)
        qf[c] = qok ? *(const uint4*)(qrows + ((size_t)((size_t)tq * B2 + b) * H + h) * HDP + c * 32 + g * 8) : make_uint4(0, 0, 0, 0);
    const unsigned char* mrow = (mask && qok) ? mask + ((size_t)b * S8 + tq) * S8 : nullptr;
    f32x4_t o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int kend = min(T, q0 + 64);                // causal: no key beyond the tile's last query
    const int wmax = q0 + wv * 16 + 15;              // last query of this wave
    for (int k0 = 0; k0 < kend; k0 += 32) {
        __syncthreads();                             // everybody is done with the previous V^T tile
        // stage V^T: thread -> (key = tid / (HDP/8), 8 consecutive d); key slot e = (key & 3) + 4 * ((key >> 4) & 1) within lane
        // group gq = (key >> 2) & 3, i.e. LDS column gq * 8 + e -- the order in which the P fragment holds its keys
        for (int it = tid; it < 32 * (HDP / 8); it += 256) {
            const int key = it / (HDP / 8), d8 = it - key * (HDP / 8);
            const int kk = k0 + key;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (kk < kend) v = *(const uint4*)(vc + (kvbase + kk) * kvs + d8 * 8);
            const int col = ((key >> 2) & 3) * 8 + (key & 3) + 4 * (key >> 4);
            const uint16_t* pv = (const uint16_t*)&v;
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[(d8 * 8 + e) * VLD + col] = pv[e];
        }
        __syncthreads();
        if (k0 > wmax) continue;                     // the whole key tile lies in this wave's future (wave-uniform)
        // S^T for two 16-key tiles
        f32x4_t s2[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
            int krow = k0 + kt * 16 + r;
            krow = krow < S8 ? krow : S8 - 1;        // clamp (masked below); slots < S8 are always valid memory
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                const uint4 kf = *(const uint4*)(kc + (kvbase + krow) * kvs + c * 32 + g * 8);
                acc = BF16::mma(kf, qf[c], acc);
            }
            s2[kt] = acc;
        }
        // scale + mask; lane holds keys k0 + kt*16 + g*4 + j of query tq
        float sv[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = k0 + kt * 16 + g * 4 + j;
                bool vis = qok && key <= tq && key < T;
                if (vis && mrow) vis = mrow[key] != 0;
                const float x = vis ? s2[kt][j] * scale : -INFINITY;
                sv[kt * 4 + j] = x;
                tmax = fmaxf(tmax, x);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run > -INFINITY) ? __expf(m_run - m_new) : 0.f;   // m_new = -inf only if nothing visible yet
        float p[8], psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = sv[e] > -INFINITY ? __expf(sv[e] - m_new) : 0.f;
            psum += p[e];
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // P as (hi, lo) bf16 B operands: lane (g, r) = query r, key slots g*8 + e in the order p[] holds them
        uint4 ph = BF16::pack(p);
        float ph_f[8], pl_f[8];
        BF16::unpack(ph, ph_f);
#pragma unroll
        for (int e = 0; e < 8; ++e) pl_f[e] = p[e] - ph_f[e];
        const uint4 pl = BF16::pack(pl_f);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            const uint4 vf = *(const uint4*)(vt + (d * 16 + r) * VLD + g * 8);   // V^T rows d*16 + r, key slots g*8..
            f32x4_t od = o[d];
            od[0] *= alpha; od[1] *= alpha; od[2] *= alpha; od[3] *= alpha;
            od = BF16::mma(vf, pl, od);
            od = BF16::mma(vf, ph, od);
            o[d] = od;
        }
    }
    if (!qok) return;
    const float inv = 1.0f / l_run;
    const int rg = tq * B2 + b;                       // global row of this query in the packed activations
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const int dd = d * 16 + g * 4;               // lane holds output features dd .. dd+3 of query tq
        if (dd + 3 < hd)
            BF16::st4(out, BF16::xp_off(h * hd + dd, rg >> 4, rg & 15, MTs), o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (dd + j < hd) BF16::st(out, BF16::xp_off(h * hd + dd + j, rg >> 4, rg & 15, MTs), o[d][j] * inv);
    }
}

static int g_prefill_mfma = 1;
extern "C" int lgen_debug_set_prefill_mfma(int v) { g_prefill_mfma = v ? 1 : 0; return 0; }

template <typename D>
static int launch_attn_prefill_tiled(const void* q_rows, const void* k_cache, const void* v_cache, void* out_packed,
                                     const unsigned char* mask, int T, int B2, int MTs, int n_head, int hd, int hdp, int S8, int kvs,
                                     float sf, hipStream_t st) {
    int qt = 32;
    auto lds_for = [&](int q) { return ((size_t)2 * PFT_KT * (hd + 1) + (size_t)2 * q * hd + 4 * PFT_KT + 2 * q) * sizeof(float); };
    while (qt > 4 && lds_for(qt) > 160 * 1024) qt >>= 1;
    const size_t lds = lds_for(qt);
    if (lds > 160 * 1024) return LGEN_ERR_BAD_ARG;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_prefill_tiled_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(attn_prefill_tiled_kernel<D>, dim3(B2 * n_head, (T + qt - 1) / qt), dim3(256), lds, st, q_rows, k_cache, v_cache,
                       out_packed, mask, T, B2, MTs, n_head, hd, hdp, S8, kvs, sf, qt);
    LGEN_CHECK_LAUNCH();
    return 0;
}

extern "C" int lgen_attn_prefill(const void* q_rows, const void* k_cache, const void* v_cache, void* out_packed,
                                 const unsigned char* mask, int T, int B2, int MTs, int n_head, int hd, int hdp, int S8,
                                 int kv_row_stride, int dtype, void* stream) {
    if (T < 1 || T > S8 || (long long)B2 * T > (long long)MTs * 16) return LGEN_ERR_BAD_ARG;
    const int kvs = kv_row_stride > 0 ? kv_row_stride : hdp;
    {
        const int epl = dtype != LGEN_F32 ? 8 : 4;
        if (kvs % epl || kvs < (hd + epl - 1) / epl * epl) return LGEN_ERR_BAD_ARG;
    }
    const float sf = sqrtf(1.0f / sqrtf((float)hd));
    if (dtype == LGEN_BF16 && g_prefill_mfma && (hdp == 64 || hdp == 128) && kvs % 8 == 0 && (hd & 3) == 0) {
        dim3 grid(B2 * n_head, (T + 63) / 64);
        if (hdp == 64)
            hipLaunchKernelGGL(attn_prefill_mfma_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)q_rows,
                               (const uint16_t*)k_cache, (const uint16_t*)v_cache, out_packed, mask, T, B2, MTs, n_head, hd, S8, kvs, sf * sf);
        else
            hipLaunchKernelGGL(attn_prefill_mfma_kernel<128>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)q_rows,
                               (const uint16_t*)k_cache, (const uint16_t*)v_cache, out_packed, mask, T, B2, MTs, n_head, hd, S8, kvs, sf * sf);
        LGEN_CHECK_LAUNCH();
        return 0;
    }
    if (T > PF_MAXT) {
        if (dtype == LGEN_BF16)
            return launch_attn_prefill_tiled<BF16>(q_rows, k_cache, v_cache, out_packed, mask, T, B2, MTs, n_head, hd, hdp, S8, kvs, sf,
                                                   (hipStream_t)stream);
        if (dtype == LGEN_F32)
            return launch_attn_prefill_tiled<F32>(q_rows, k_cache, v_cache, out_packed, mask, T, B2, MTs, n_head, hd, hdp, S8, kvs, sf,
                                                  (hipStream_t)stream);
        if (dtype == LGEN_F16)
            return launch_attn_prefill_tiled<F16>(q_rows, k_cache, v_cache, out_packed, mask, T, B2, MTs, n_head, hd, hdp, S8, kvs, sf,
                                                  (hipStream_t)stream);
        return LGEN_ERR_BAD_ARG;
    }
    const size_t lds = ((size_t)2 * T * (hd + 1) + 4 * PF_MAXT + 4 * hd) * sizeof(float);
    if (lds > 160 * 1024) return LGEN_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == LGEN_BF16) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_prefill_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(attn_prefill_kernel<BF16>, dim3(B2 * n_head), dim3(256), lds, st, q_rows, k_cache, v_cache, out_packed, mask,
                           T, B2, MTs, n_head, hd, hdp, S8, kvs, sf);
    } else if (dtype == LGEN_F16) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_prefill_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(attn_prefill_kernel<F16>, dim3(B2 * n_head), dim3(256), lds, st, q_rows, k_cache, v_cache, out_packed, mask,
                           T, B2, MTs, n_head, hd, hdp, S8, kvs, sf);
    } else if (dtype == LGEN_F32) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)attn_prefill_kernel<F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(attn_prefill_kernel<F32>, dim3(B2 * n_head), dim3(256), lds, st, q_rows, k_cache, v_cache, out_packed, mask,
                           T, B2, MTs, n_head, hd, hdp, S8, kvs, sf);
    } else {
        return LGEN_ERR_BAD_ARG;
    }
    LGEN_CHECK_LAUNCH();
    return 0;
}
