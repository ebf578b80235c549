// Shared by the skinny-GEMM translation units (gemm_skinny.hip, gemm_normpre.hip): kernel arguments,
// early epilogue operand fetch and the fused epilogues.  See gemm_skinny.hip for the design notes.
#pragma once
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
    const int* pos_ptr;  // QKV: position of this token: device scalar, or one per row (pos_stride = 1)
    int pos_stride;      // QKV: 0 = all rows at *pos_ptr; 1 = row m at pos_ptr[m] (continuous batching: every row its own position)
    const uint4* nw;     // NORM: RMSNorm weight [K] storage dtype
    const float* ssq_in; // NORM: [MTs*16][LGEN_SSQ_STRIDE], `parts` partial sums of squares per x row
    float* ssq_out;      // RES: [MTs*16][LGEN_SSQ_STRIDE], N/16 partials per new row (nullable)
    int N, KCH, MTs, M;
    int d, hd, hdp, H, S8;
    int kvs;             // QKV: elements between consecutive cache rows (hdp, or 2*hdp for an interleaved K|V slab)
    int parts;
    float eps, inv_k;
    int passes;          // fused-norm GEMMs: consecutive n-groups one workgroup walks with its activations kept in registers (>= 1)
    int db;              // gemm_tile.hip: development ablation mask (LGEN_TILE_ABLATE), 0 in production
};
// weight chunk load: default cache policy (chains in flight share the weights through the memory-side cache; measured better than
// non-temporal).  One plain load, NOT a run-time choice of policy: a branch per load makes the compiler lose count of the loads in
// flight and wait for all of them (s_waitcnt vmcnt(0)) in front of the first MFMA, which turns the operand ring into load-all /
// wait-all / compute-all (round 2, found in the ISA of gemm_kernel<BF16,2,1,EPI_RES,false,6>).
LGEN_DEV uint4 ldg_w(const uint4* p) { return *p; }

// This lane's share of one row's sum of squares: lane group q4 = lane >> 4 sums a contiguous quarter of the row's `parts`
// partials; the caller adds the four groups (xor 16, 32).  Fast path (parts a multiple of 16, <= 128: every registry model):
// up to 8 UNCONDITIONAL 16-byte loads, so that the caller can issue them -- and its weight loads behind them -- without a
// single wait in between (the round-1 form, a runtime-length loop of 4-byte loads, made the compiler wait for every group of
// four before the weight stream was even requested: 2-3 us per fused-norm GEMM).
template <int NV>
struct SsqLoads { float4 v[NV > 0 ? NV : 1]; int n4; bool fast; };
// NV (16-byte loads a lane keeps in flight per m-tile) shrinks with the m-tiles of a workgroup: 8 / 4 / 2 for MT = 1 / 2 / >= 4,
// i.e. the no-wait form covers d <= 2048 / 1024 / 512; wider rows are reduced before the weights are requested (ssq_rows_now).
template <int MT>
constexpr int ssq_nv() { return MT == 1 ? 8 : (MT == 2 ? 4 : 2); }
template <int NV>
LGEN_DEV SsqLoads<NV> ssq_issue(const float* ssq_in, int parts, int row, int lane) {
    SsqLoads<NV> r;
    r.fast = (parts & 15) == 0 && parts <= 16 * NV;
    r.n4 = parts >> 4;                       // float4 loads per lane group
    const int q4 = lane >> 4;
    const float4* p = (const float4*)(ssq_in + (size_t)row * LGEN_SSQ_STRIDE) + q4 * r.n4;
    if (r.fast) {
        constexpr int H = NV > 4 ? 4 : NV;
#pragma unroll
        for (int j = 0; j < H; ++j) r.v[j] = p[j < r.n4 ? j : 0];
        if constexpr (NV > 4) {
            if (r.n4 > 4) {
#pragma unroll
                for (int j = 4; j < NV; ++j) r.v[j] = p[j < r.n4 ? j : 0];
            }
        }
    }
    return r;
}
template <int NV>
LGEN_DEV float ssq_finish(const SsqLoads<NV>& r) {
    float s = 0.f;
    if (r.fast) {
        constexpr int H = NV > 4 ? 4 : NV;
#pragma unroll
        for (int j = 0; j < H; ++j) s += j < r.n4 ? (r.v[j].x + r.v[j].y) + (r.v[j].z + r.v[j].w) : 0.f;
        if constexpr (NV > 4) {
            if (r.n4 > 4) {
#pragma unroll
                for (int j = 4; j < NV; ++j) s += j < r.n4 ? (r.v[j].x + r.v[j].y) + (r.v[j].z + r.v[j].w) : 0.f;
            }
        }
    }
    return s;  // !fast: the caller takes ssq_rows_now() instead
}

// The same statistic, reduced at once (the caller has NOT requested its weights yet, or does not mind waiting): this lane's
// share of the sums of squares of rows (mt0 + i) * 16 + (lane & 15), i < MT.  parts a multiple of 16: 16-byte loads in rounds
// of 4 per m-tile, every round's 4 * MT loads issued before its one wait (d = 1024: one round); else a scalar loop.
template <int MT>
LGEN_DEV void ssq_rows_now(const float* ssq_in, int parts, int mt0, int lane, float (&s)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) s[i] = 0.f;
    const float* base = ssq_in + (size_t)(mt0 * 16 + (lane & 15)) * LGEN_SSQ_STRIDE;
    if ((parts & 15) == 0) {
        const int n4 = parts >> 4;
        const float4* p = (const float4*)base + (lane >> 4) * n4;
        for (int c = 0; c < n4; c += 4) {
            float4 v[MT][4];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = p[(size_t)i * (16 * LGEN_SSQ_STRIDE / 4) + (c + j < n4 ? c + j : c)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i] += c + j < n4 ? (v[i][j].x + v[i][j].y) + (v[i][j].z + v[i][j].w) : 0.f;
        }
    } else {
        for (int q = lane >> 4; q < parts; q += 4) {
#pragma unroll
            for (int i = 0; i < MT; ++i) s[i] += base[(size_t)i * 16 * LGEN_SSQ_STRIDE + q];
        }
    }
}

// Position of the rows a lane works on (QKV epilogue: RoPE angles, KV-cache slot): one device scalar (generate(): every row at
// the same position, a scalar-cache load) or one per row (continuous batching: a vector load, issued FIRST in the kernel so that
// the dependent RoPE-table loads never wait behind the weight stream).
template <int MT, int EPI>
LGEN_DEV void load_row_pos(const GemmArgs& a, int mt0, int lane, int (&posr)[MT]) {
#pragma unroll
    for (int i = 0; i < MT; ++i) posr[i] = 0;
    if constexpr (EPI == 5 /* EPI_QKV */) {
        if (a.pos_stride) {
#pragma unroll
            for (int i = 0; i < MT; ++i) posr[i] = a.pos_ptr[(mt0 + i) * 16 + (lane & 15)];
        } else {
            const int p0 = *a.pos_ptr;
#pragma unroll
            for (int i = 0; i < MT; ++i) posr[i] = p0;
        }
    }
}
template <int MT>
LGEN_DEV int pick_pos(const int (&posr)[MT], int i) {  // posr[i] for a runtime i without indexing the register array
    int p = 0;
#pragma unroll
    for (int k = 0; k < MT; ++k) p |= posr[k] & -(int)(i == k);  // (a select chain is turned back into an indexed load)
    return p;
}

LGEN_DEV float silu_f(float x) { return x / (1.0f + expf(-x)); }
LGEN_DEV float gelu_tanh_f(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float inner = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}

// memory operands of an epilogue, requested early: RES -> the residual tile, QKV -> RoPE (cos, sin) x 2
template <typename D, int EPI>
LGEN_DEV uint4 epi_prefetch(const GemmArgs& a, int nt, int mt, int lane, int pos) {
    uint4 aux = make_uint4(0, 0, 0, 0);
    const int r = lane & 15, g = lane >> 4;
    const int n = nt * 16 + g * 4;
    if constexpr (EPI == EPI_RES) {
        const size_t o = D::xp_off(n, mt, r, a.MTs);
        if constexpr (D::ESZ == 2) {
            const uint2 v = *(const uint2*)((const uint16_t*)a.out + o);
            aux.x = v.x; aux.y = v.y;
        } else {
            aux = *(const uint4*)((const float*)a.out + o);
        }
    } else if constexpr (EPI == EPI_QKV) {
        const int sec = n / a.d;
        if (sec < 2) {
            const int c = n - sec * a.d;
            const int dd = c % a.hd;
            aux = *(const uint4*)(a.freqs + ((size_t)pos * (a.hd >> 1) + (dd >> 1)) * 2);
        }
    }
    return aux;
}

// one 16x16 output tile: lane (g = lane>>4, r = lane&15) holds n = nt*16 + g*4 + {0..3}, m = mt*16 + r
template <typename D, int EPI>
LGEN_DEV void epilogue(const GemmArgs& a, int nt, int mt, int lane, f32x4_t v, f32x4_t v2, const uint4& aux, int pos) {
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
        float h0, h1, h2, h3;
        if constexpr (D::ESZ == 2) {
            D::unpack4(make_uint2(aux.x, aux.y), h0, h1, h2, h3);
        } else {
            h0 = __uint_as_float(aux.x); h1 = __uint_as_float(aux.y);
            h2 = __uint_as_float(aux.z); h3 = __uint_as_float(aux.w);
        }
        h0 = D::rnd(h0 + x0); h1 = D::rnd(h1 + x1); h2 = D::rnd(h2 + x2); h3 = D::rnd(h3 + x3);
        D::st4(a.out, D::xp_off(n, mt, r, a.MTs), h0, h1, h2, h3);
        if (a.ssq_out) {  // fixed-order partial of sum(h^2) over this tile's 16 columns, per row
            float ss = ((h0 * h0 + h1 * h1) + h2 * h2) + h3 * h3;
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            if (lane < 16) a.ssq_out[(size_t)m * LGEN_SSQ_STRIDE + nt] = ss;
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
        // nt is the w1 tile (even), v2 the matching w3 tile; output feature f = (nt/2)*16 + g*4
        float y0 = D::rnd(v2[0]), y1 = D::rnd(v2[1]), y2 = D::rnd(v2[2]), y3 = D::rnd(v2[3]);
        int f = (nt >> 1) * 16 + g * 4;
        D::st4(a.out, D::xp_off(f, mt, r, a.MTs),
               D::rnd(silu_f(x0)) * y0, D::rnd(silu_f(x1)) * y1, D::rnd(silu_f(x2)) * y2, D::rnd(silu_f(x3)) * y3);
    } else if constexpr (EPI == EPI_QKV) {
        if (m >= a.M) return;
        const int sec = n / a.d;
        const int c = n - sec * a.d;
        const int head = c / a.hd;
        const int dd = c - head * a.hd;
        if (sec < 2) {  // 2-D RoPE on interleaved (even, odd) pairs, fp32, one rounding
            const float fx = __uint_as_float(aux.x), fy = __uint_as_float(aux.y);
            const float fz = __uint_as_float(aux.z), fw = __uint_as_float(aux.w);
            rope_pair(x0, x1, fx, fy);
            rope_pair(x2, x3, fz, fw);
        }
        // (pointers and strides read into values FIRST: selecting between the ADDRESSES of kernel-argument fields made the compiler
        // spill them to 40 B of scratch once `a` arrived through a helper's reference)
        void* const po = a.out;
        void* const pk = a.kc;
        void* const pv = a.vc;
        const int hdp = a.hdp, kvs = a.kvs;
        const size_t rh = (size_t)m * a.H + head;
        void* const dst = sec == 0 ? po : (sec == 1 ? pk : pv);
        const size_t off = sec == 0 ? rh * hdp + dd : (rh * a.S8 + pos) * kvs + dd;
        D::st4(dst, off, x0, x1, x2, x3);
    }
}

template <int EPI> constexpr bool epi_has_aux() { return EPI == EPI_RES || EPI == EPI_QKV; }


// ---- what every K-splitting GEMM kernel does around its MFMA loop (gemm_kernel, gemm_steady_kernel, np_pass) ----------------
// Epilogue work items ("units") of a workgroup tile: one per 16x16 output tile, one per (w1, w3) tile pair for SwiGLU.  With
// KW > 1 waves unit u belongs to wave u % KW (its q-th unit is u = w + q * KW); with one wave all units are its own.
template <int MT>
struct RowPos { int v[MT]; };   // by value: an `int (&)[MT]` parameter kept the array in scratch for (MT, NT, EPI) = (2, 1, QKV)
template <int MT>
LGEN_DEV int pick_pos(const RowPos<MT>& p, int i) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < MT; ++k) r |= p.v[k] & -(int)(i == k);
    return r;
}

template <int MT, int NT, int EPI>
constexpr int gemm_units() { return EPI == EPI_SWIGLU ? (NT * MT / 2 > 0 ? NT * MT / 2 : 1) : NT * MT; }

// memory operands of this wave's units, requested before the main loop
template <typename D, int MT, int NT, int EPI>
LGEN_DEV void gemm_aux_prefetch(const GemmArgs& a, uint4 (&aux)[gemm_units<MT, NT, EPI>()], int w, int KW, int lane, int nt0, int mt0,
                                const int (&posr_)[MT]) {
    RowPos<MT> posr;
#pragma unroll
    for (int k = 0; k < MT; ++k) posr.v[k] = posr_[k];
    constexpr int UNITS = gemm_units<MT, NT, EPI>();
#pragma unroll
    for (int q = 0; q < UNITS; ++q) aux[q] = make_uint4(0, 0, 0, 0);
    if constexpr (epi_has_aux<EPI>()) {
#pragma unroll
        for (int q = 0; q < UNITS; ++q) {
            const int u = w + q * KW;   // (q >= ceil(UNITS / 2) never matches when KW >= 2)
            if (u < UNITS) {
                const int j = u / MT, i = u - j * MT;
                aux[q] = epi_prefetch<D, EPI>(a, nt0 + j, mt0 + i, lane, pick_pos<MT>(posr, i));
            }
        }
    }
}

// cross-wave K reduction through LDS in a FIXED order (wave 0, 1, 2, ...: deterministic, no atomics) + the fused epilogue of this
// wave's units.  `red` = this pass's reduction buffer ([KW][NT * MT][64] float4); KW == 1: no LDS, no barrier.
template <typename D, int MT, int NT, int EPI>
LGEN_DEV void gemm_reduce_epilogue(const GemmArgs& a, const f32x4_t (&acc)[NT][MT], float4* red, int w, int KW, int lane, int nt0,
                                   int mt0, const int (&posr_)[MT], const uint4 (&aux)[gemm_units<MT, NT, EPI>()]) {
    constexpr int TILES = NT * MT, UNITS = gemm_units<MT, NT, EPI>(), UPW = (UNITS + 1) / 2;
    RowPos<MT> posr;
#pragma unroll
    for (int k = 0; k < MT; ++k) posr.v[k] = posr_[k];
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
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const f32x4_t v = acc[j][i];
            red[((size_t)w * TILES + j * MT + i) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    __syncthreads();
    auto rsum = [&](int t) {
        float4 s = red[(size_t)t * 64 + lane];
        for (int ww = 1; ww < KW; ++ww) {
            const float4 p = red[((size_t)ww * TILES + t) * 64 + lane];
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

// gemm_normpre.hip: RMSNorm-fused GEMM that normalises while its weights are in flight (small K per wave);
// returns LGEN_ERR_UNSUPPORTED when the shape is outside its envelope (caller falls back).
int lgen_gemm_normpre_try(const GemmArgs& a, int epi, int dtype, int mt, int nt, int kw, hipStream_t st);
