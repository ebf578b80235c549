// Big-M GEMM family of the decode step (round 4): out[M, N] = x[M, K] . W[N, K]^T for chains of 128 .. 512 rows.
//
// Same operation, operand layouts and fused epilogues as gemm_skinny.hip / gemm_normpre.hip (the five nn.Linear of a
// reference layer, autoregressive/models/gpt.py:161-167,199-200,238-240,367-368, with RMSNorm gpt.py:143-148, RoPE + KV append
// gpt.py:214-226, SwiGLU gpt.py:167 and the residual adds gpt.py:255-256 fused around them), but shaped for the rows a wide
// chain carries.  The skinny kernels split K over the 8 waves of a workgroup and own 16-32 rows: at 256 rows every weight byte
// is pulled through L2 by 8 m-groups, every (n-group, m-group) unit pays a cross-wave LDS reduction, and the load phase of a CU
// never overlaps its compute phase (profiles/r03_pmc.json: fetch 1.4-3.3 x algorithmic; r03_sq_pmc.csv: waves parked 58-78 %).
//
// Here a workgroup owns a (WM*MTV*16 rows) x (WN*NTV*16 columns) output tile over the WHOLE K range:
//   * waves split the TILE (WM x WN), never K: no cross-wave reduction, accumulators stay in registers until the epilogue;
//   * both operands stream HBM/L2 -> LDS through `global_load_lds_dwordx4` (the fragment-packed global layout IS the LDS
//     image: one 1 KiB chunk per wave-instruction, lane-linear, conflict-free for the `ds_read_b128` operand reads) into a
//     ring of STAGES slots of KB k-chunks each, with counted `s_waitcnt vmcnt` and ONE raw `s_barrier` per stage, so that
//     STAGES-1 stages of loads stay in flight under the MFMAs of the oldest; every operand byte enters a CU once;
//   * NORM: the RMSNorm of the rows is applied to the B fragments on their way from LDS to the MFMA (rnd(rnd(x * rinv) * w), the
//     reference's two roundings); with WN == 1 every fragment is normalised by exactly one wave of the workgroup.  Row statistics
//     (the producer's partial sums of squares) and the norm weight arrive through the same DMA path, summed in the fixed order of
//     the skinny kernels (bit-identical scales);
//   * block id -> (n-group, m-group) so that the m-groups reading the same weights sit on one XCD (one L2 fill).
//
// Per-CU traffic is the tile perimeter: (rows + cols) x K x 2 B.  GPT-L at 256 rows: wqkv 64 x 48 tiles -> 224 KB per CU
// against 448 KB for the (2, 4, 8) x 3-pass skinny form.
#include <cstdlib>
#include <type_traits>

#include "gemm_epilogue.h"

typedef __attribute__((address_space(1))) const void* gt_gptr_t;
typedef __attribute__((address_space(3))) void* gt_lptr_t;

template <int N>
LGEN_DEV void gt_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
LGEN_DEV void gt_wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }

// LDS reads of the DMA-filled ring as inline asm.  hipcc (ROCm 7.2) cannot tell which ring slot a `ds_read` touches, so behind
// every `global_load_lds` it puts `s_waitcnt vmcnt(0)` in front of the next LDS read it can see -- which drains the whole ring at
// every stage (found in the ISA of the first version of this kernel).  Reads it cannot see are ordered by hand instead: DMA
// pieces by the counted vmcnt + barrier of the stage loop, the reads themselves by a counted lgkmcnt (LDS operations return in
// order) followed by gt_touch() on every destination, which ties the registers' uses behind the wait.
template <int OFF>
LGEN_DEV u32x4_t gt_lds_rd(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
LGEN_DEV void gt_touch(u32x4_t& v) { asm volatile("" : "+v"(v)); }
LGEN_DEV uint4 gt_u4(const u32x4_t& v) { return make_uint4(v[0], v[1], v[2], v[3]); }

template <int I, int N, typename F>
LGEN_DEV void gt_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        gt_static_for<I + 1, N>(f);
    }
}

// This lane group's share (g = lane >> 4) of one row's partial sums of squares, read from global memory, in the fixed order of
// ssq_rows_now / gemm_normpre.hip: `parts` a multiple of 16 and <= 128 -> a contiguous quarter as float4s, else every fourth
// partial.  The loads of a batch are issued together (clamped addresses, out-of-range terms add +0.0f, which leaves a sum of
// squares unchanged): a chain of dependent load -> add pairs costs 50 L2 round trips for GPT-3B's 200 partials.
LGEN_DEV float gt_row_ssq_global(const float* rowg, int parts, int g) {
    float s = 0.f;
    if ((parts & 15) == 0 && parts <= 128) {
        const int n4 = parts >> 4;
        float4 v[8];
#pragma unroll
        for (int J = 0; J < 8; ++J) v[J] = ((const float4*)rowg)[g * n4 + (J < n4 ? J : 0)];
#pragma unroll
        for (int J = 0; J < 8; ++J)
            if (J < n4) s += (v[J].x + v[J].y) + (v[J].z + v[J].w);
    } else {
        for (int q0 = g; q0 < parts; q0 += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + 4 * j;
                t[j] = rowg[q < parts ? q : parts - 1];
                t[j] = q < parts ? t[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
    }
    return s;
}

// running (section, head, dd) of a q|k|v output column n, advanced 16 columns at a time (one division per wave, not per tile)
struct QkvCol {
    int sec, head, dd;
    LGEN_DEV void init(int n, int d, int hd) {
        sec = n / d;
        const int c = n - sec * d;
        head = c / hd;
        dd = c - head * hd;
    }
    LGEN_DEV void next16(int hd, int H) {
        dd += 16;
        if (dd >= hd) { dd -= hd; ++head; }
        if (head >= H) { head -= H; ++sec; }
    }
};

// LW = 0: every wave issues its share of the DMA pieces AND computes (round-4 first form: the ~90 cycles a wave spends per
// 1 KiB piece while four waves issue sit in front of its own MFMAs).  LW = 4: wave specialisation -- WM*WN consumer waves (LDS
// reads, RMSNorm, MFMA, epilogue) + LW loader waves (DMA issue and the counted waits only); a workgroup's waves are dealt to
// the SIMDs round-robin, so every SIMD holds one consumer and one loader and the DMA issue overlaps the MFMA / VALU work.
//
// Measured and dropped in round 5 (profiles/r05_first_ab.log, r05_tile_sweep_640_ln_touch.log; code in git history, commit "Merge
// branch r5-prep"): (a) loader waves that ALSO apply the RMSNorm (activation fragments through registers, normalised once per
// workgroup, ds_write into the ring; plain consumers with 64 x 64 wave tiles = half the LDS reads per MFMA): bit-identical, no
// faster alone (w1||w3 19.6 against 17.9 us, lm_head 55.5 against 51.5) and 107 against 122 img/s in the bench -- the K loop is not
// bound by its LDS reads; (b) an L2 "touch-ahead" (one dword load per 128-byte line of the later stages right behind the first
// ring stages): every touch pulls its line through the CU's vector L1, the same path the DMA pieces take, so the feed traffic
// doubles -- wo 6.7 -> 9.8 us, w2 10.5 -> 15.6, w1||w3 17.9 -> 23.2, bench 112 against 122 img/s.  What that says: these kernels
// are bound by the L2 -> CU feed (~45 B/clk/CU by LDS-DMA), i.e. by tile perimeter x K bytes per CU, not by LDS reads or MFMA.
// (c) deeper rings -- 5 / 6 stages for wqkv, 6 / 9 half-size stages for wo / w2: the same alone, 122.3-122.9 against 121.6-123.4 img/s
// in the bench over two boxes (r05_first_ab.log, r05_ab3.log): inside the box spread, instantiations removed again.
template <typename D, int WM, int WN, int MTV, int NTV, int KB, int STAGES, int EPI, bool NORM, int LW>
__global__ __launch_bounds__(64 * (WM * WN + LW)) void gemm_tile_kernel(GemmArgs a) {
    constexpr int NC = WM * WN, NL = LW ? LW : NC, MTW = WM * MTV, NTW = WN * NTV;
    constexpr int CPK = MTW + NTW;      // 1 KiB chunks per k-chunk of the workgroup tile
    constexpr int CPS = KB * CPK;       // chunks per ring stage
    static_assert(CPS % NL == 0, "every DMA-issuing wave issues the same number of pieces per stage");
    constexpr int P = CPS / NL;         // DMA pieces per issuing wave per stage
    constexpr int STAGE_B = CPS * 1024;
    static_assert((STAGES - 1) * P <= 60, "vmcnt is a 6-bit counter");
    static_assert(EPI != EPI_SWIGLU || (NTV % 2 == 0), "w1 / w3 tiles come in pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool consumer = LW == 0 || w < NC, loader = LW == 0 || w >= NC;
    // (round 6, measured and dropped: `s_setprio 2` for all waves / `s_setprio 3` for the loader waves of this kernel, like `s_setprio 3` in the
    // persistent attention kernel before it: 124.5-125.7 / 124.6 against 124.9-125.4 img/s in the two-chain bench, profiles/r06_c2_bench_ab.log)
    const int lw = LW ? w - NC : w;     // index among the DMA-issuing waves
    const int cw = w < NC ? w : 0;      // index among the consumers
    const int wm = cw / WN, wn = cw - wm * WN;
    const int ntiles = a.N >> 4;
    const int gx = (ntiles + NTW - 1) / NTW, gy = a.MTs / MTW;
    // Block id -> tile.  Workgroups are dealt to the 8 XCDs round-robin (XCD = bid & 7), each XCD has its own L2: XCD x owns the
    // column groups bx = xc (mod XC) of the row groups by = xrow (mod XR), x = xrow * XC + xc, XR * XC = 8, row groups fastest (the
    // workgroups that share a weight tile run together).  XR = 1 (every XCD reads ALL activation rows, each weight byte enters
    // one L2): right while the weights outweigh the row panel; the launcher picks XR = 2 / 4 where 8 copies of the row panel cost
    // more than 2 / 4 copies of the weights (wo / w2 of a 640-row chain: 34.6 -> 25.9 MB of L2 fills per launch).
    const int bid = blockIdx.x, xrl = (a.db >> 8) & 3, XR = 1 << xrl, XC = 8 >> xrl;
    const int xcd = bid & 7, slot0 = bid >> 3;
    const int gyl = (gy + XR - 1) >> xrl;
    const int by = (slot0 % gyl) * XR + (xcd >> (3 - xrl)), bx = (slot0 / gyl) * XC + (xcd & (XC - 1));
    if (bx >= gx || by >= gy) return;  // padding of the decoded grid (whole workgroup, before any barrier)
    const int nt0 = bx * NTW, mt0 = by * MTW;
    const int NI = a.KCH / KB;          // ring stages over K (launcher: KCH % KB == 0, NI >= STAGES - 1)
    const int mtw0 = mt0 + wm * MTV, ntw0 = nt0 + wn * NTV;   // this wave's first m-tile / n-tile
    const unsigned lds0 = (unsigned)(uintptr_t)(gt_lptr_t)smem;

    // stage loop of a DMA-issuing wave: own pieces of stage `it` landed (younger stages stay in flight) -> barrier (everybody's
    // pieces landed; every consumer has finished reading the slot that stage it+STAGES-1 overwrites: it was consumed in iteration
    // it-1) -> issue stage it+STAGES-1
    auto stage_wait = [&](int it) {
        const int rem = NI - 1 - it;
        if (rem >= STAGES - 2) gt_wait_vm<(STAGES - 2) * P>();
        else if (STAGES > 3 && rem == 1) gt_wait_vm<P>();
        else gt_wait_vm<0>();
    };

    if (LW != 0 && !consumer) {
        // ================= loader wave =================
        const uint4* src[P];
        unsigned step[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int c = lw + p * NL, kk = c / CPK, cc = c - kk * CPK;
            if (cc < MTW) {
                src[p] = a.xp + ((size_t)kk * a.MTs + mt0 + cc) * 64 + lane;
                step[p] = (unsigned)(KB * a.MTs * 64);
            } else {
                int nt = nt0 + cc - MTW;
                nt = nt < ntiles ? nt : ntiles - 1;   // ragged last n-group: re-reads a valid tile, its epilogue is skipped
                src[p] = a.wp + ((size_t)nt * a.KCH + kk) * 64 + lane;
                step[p] = (unsigned)(KB * 64);
            }
        }
        auto issue = [&](int slot) {
            if (a.db & 4) return;   // ablation (LGEN_TILE_ABLATE): no operand DMA
            unsigned char* dst = smem + slot * STAGE_B + lw * 1024;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                __builtin_amdgcn_global_load_lds((gt_gptr_t)src[p], (gt_lptr_t)(dst + p * NL * 1024), 16, 0, 0);
                src[p] += step[p];
            }
        };
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t) issue(t);
        int slot = 0;
        for (int it = 0; it < NI; ++it) {
            stage_wait(it);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (it + STAGES - 1 < NI) issue(slot == 0 ? STAGES - 1 : slot - 1);
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }
        return;
    }

    // ================= consumer wave (LW == 0: also issues its share of the DMA) =================
    // ---- epilogue operands first (oldest requests: they never disturb the counted waits below) ----
    int pos = 0;
    if constexpr (EPI == EPI_QKV) pos = *a.pos_ptr;   // per-row positions (continuous batching) stay on the skinny kernels
    uint2 res[NTV][MTV];
    uint4 rope[EPI == EPI_QKV ? NTV : 1];
    if constexpr (EPI == EPI_RES) {
#pragma unroll
        for (int j = 0; j < NTV; ++j)
#pragma unroll
            for (int i = 0; i < MTV; ++i) {
                const int nt = ntw0 + j < ntiles ? ntw0 + j : ntiles - 1;
                res[j][i] = *(const uint2*)((const uint16_t*)a.out + D::xp_off(nt * 16 + (lane >> 4) * 4, mtw0 + i, lane & 15, a.MTs));
            }
    }
    // (wave tiles of >= 20 MFMA tiles -- the 128 x 160 shape of GPT-3B's wqkv -- request the RoPE factors in the epilogue instead,
    // where the operand fragments are dead: 40 more registers across the K loop spill at 256)
    constexpr bool ROPE_LATE = EPI == EPI_QKV && MTV * NTV >= 20;
    auto load_rope = [&]() {
        QkvCol qc;
        qc.init((ntw0 < ntiles ? ntw0 : ntiles - 1) * 16 + (lane >> 4) * 4, a.d, a.hd);
#pragma unroll
        for (int j = 0; j < (EPI == EPI_QKV ? NTV : 1); ++j) {
            rope[j] = make_uint4(0, 0, 0, 0);
            if (qc.sec < 2) rope[j] = *(const uint4*)(a.freqs + ((size_t)pos * (a.hd >> 1) + (qc.dd >> 1)) * 2);
            qc.next16(a.hd, a.H);
        }
    };
    if constexpr (EPI == EPI_QKV && !ROPE_LATE) load_rope();

    // ---- DMA bookkeeping (LW == 0 only): piece p of this wave is chunk c = w + p * NL of a stage ----
    const uint4* src[LW == 0 ? P : 1];
    unsigned step[LW == 0 ? P : 1];
    if constexpr (LW == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int c = w + p * NL, kk = c / CPK, cc = c - kk * CPK;
            if (cc < MTW) {
                src[p] = a.xp + ((size_t)kk * a.MTs + mt0 + cc) * 64 + lane;
                step[p] = (unsigned)(KB * a.MTs * 64);
            } else {
                int nt = nt0 + cc - MTW;
                nt = nt < ntiles ? nt : ntiles - 1;
                src[p] = a.wp + ((size_t)nt * a.KCH + kk) * 64 + lane;
                step[p] = (unsigned)(KB * 64);
            }
        }
    }
    auto issue = [&](int slot) {
        if constexpr (LW == 0) {
            unsigned char* dst = smem + slot * STAGE_B + w * 1024;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                __builtin_amdgcn_global_load_lds((gt_gptr_t)src[p], (gt_lptr_t)(dst + p * NL * 1024), 16, 0, 0);
                src[p] += step[p];
            }
        }
    };

    // ---- NORM prologue requests: norm weight (shared, every consumer writes the same bytes) + this wave's row statistics ----
    // row statistics staging: the last ring slot (idle until stage STAGES-1 is issued) when a wave's share fits there, else a
    // dedicated region behind the norm weight (launcher: a.passes = 1 selects the slot, 2 the dedicated region, 3 no staging at
    // all: wide rows, read from global memory where the row scales are computed)
    unsigned char* s_nw = smem + STAGES * STAGE_B;
    const int nw_bytes = (a.KCH * D::KC * D::ESZ + 1023) / 1024 * 1024;
    const int ssq_share = (MTV * 16 * a.parts * 4 + 1023) & ~1023;   // whole 1 KiB DMA rounds: the clamped tail lanes stay inside the share
    const unsigned ssq_off = a.passes == 2 ? (unsigned)(STAGES * STAGE_B + nw_bytes + cw * ssq_share)
                                           : (unsigned)((STAGES - 1) * STAGE_B + cw * (STAGE_B / NC));
    unsigned char* s_ssq = smem + ssq_off;
    const int q4 = a.parts >> 2;  // float4s per statistics row (launcher: parts % 4 == 0)
    if constexpr (NORM) {
        const int nwb = a.KCH * D::KC * D::ESZ;  // bytes of the norm weight
        for (int o = 0; o < nwb; o += 1024) {
            int off = o + lane * 16;
            off = off < nwb - 16 ? off : nwb - 16;
            __builtin_amdgcn_global_load_lds((gt_gptr_t)((const char*)a.nw + off), (gt_lptr_t)(s_nw + o), 16, 0, 0);
        }
        const int tot = a.passes == 3 ? 0 : MTV * 16 * q4;   // passes == 3: the statistics are read straight from global memory below
        for (int o = 0; o < tot; o += 64) {
            int e = o + lane;
            e = e < tot ? e : tot - 1;
            const int row = e / q4, c4 = e - row * q4;
            __builtin_amdgcn_global_load_lds((gt_gptr_t)(a.ssq_in + (size_t)(mtw0 * 16 + row) * LGEN_SSQ_STRIDE + c4 * 4),
                                             (gt_lptr_t)(s_ssq + o * 16), 16, 0, 0);
        }
    }
    // ---- fill the ring: stages 0 .. STAGES-2 ----
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t) issue(t);

    // ---- RMSNorm row scales, fixed summation order of gemm_normpre.hip / ssq_rows_now (bit-identical) ----
    float ri[MTV];
    if constexpr (NORM) {
        if (a.db & 16) {   // ablation: no statistics prologue
#pragma unroll
            for (int i = 0; i < MTV; ++i) ri[i] = 1.0f;
        } else {
        if constexpr (LW == 0) gt_wait_vm<(STAGES - 1) * P>();   // the prologue pieces are older than every stage piece
        else gt_wait_vm<0>();
        const int r = lane & 15, g = lane >> 4;
        const unsigned ssq0 = lds0 + ssq_off;
#pragma unroll
        for (int i = 0; i < MTV; ++i) {
            const unsigned rowa = ssq0 + (unsigned)((i * 16 + r) * a.parts) * 4;
            const float* rowg = a.ssq_in + (size_t)((mtw0 + i) * 16 + r) * LGEN_SSQ_STRIDE;
            float s = 0.f;
            if (a.passes == 3) {   // wide rows (GPT-3B: 200 partials x 128 rows = 100 KB would not fit beside the ring): same order, from L2
                s = gt_row_ssq_global(rowg, a.parts, g);
            } else if ((a.parts & 15) == 0 && a.parts <= 128) {
                const int n4 = a.parts >> 4;
                u32x4_t v[8];
                gt_static_for<0, 8>([&](auto j) {
                    constexpr int J = decltype(j)::value;
                    v[J] = gt_lds_rd<0>(rowa + (unsigned)(g * n4 + (J < n4 ? J : 0)) * 16);
                });
                gt_wait_lgkm<0>();
                gt_static_for<0, 8>([&](auto j) {
                    constexpr int J = decltype(j)::value;
                    gt_touch(v[J]);
                    if (J < n4)
                        s += (__uint_as_float(v[J][0]) + __uint_as_float(v[J][1])) + (__uint_as_float(v[J][2]) + __uint_as_float(v[J][3]));
                });
            } else {
                for (int q = g; q < a.parts; q += 4) {
                    float t;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(rowa + (unsigned)q * 4) : "memory");
                    s += t;
                }
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            ri[i] = 1.0f / sqrtf(s * a.inv_k + a.eps);
        }
        }
    }

    f32x4_t acc[NTV][MTV];
#pragma unroll
    for (int j = 0; j < NTV; ++j)
#pragma unroll
        for (int i = 0; i < MTV; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const unsigned ldsB = lds0 + lane * 16 + (wm * MTV) * 1024;
    const unsigned ldsA = lds0 + lane * 16 + (MTW + wn * NTV) * 1024;
    const unsigned ldsN = lds0 + STAGES * STAGE_B + (lane >> 4) * 16;
    constexpr int NRD = MTV + NTV + (NORM ? 1 : 0);   // LDS reads per k-chunk
    static_assert(NRD <= 15, "lgkmcnt is a 4-bit counter");
    int slot = 0;
    for (int it = 0; it < NI; ++it) {
        if constexpr (LW == 0) stage_wait(it);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + STAGES - 1 < NI) issue(slot == 0 ? STAGES - 1 : slot - 1);
        const unsigned aB = ldsB + slot * STAGE_B, aA = ldsA + slot * STAGE_B, aN = ldsN + it * (KB * 64);
        // fragments of k-chunk kk+1 are requested before the MFMAs of k-chunk kk (two register sets)
        u32x4_t Bf[2][MTV], Af[2][NTV], Wn[2];
        auto rd = [&](auto kk_) {
            constexpr int KK = decltype(kk_)::value, S = KK & 1;
            gt_static_for<0, MTV>([&](auto i) { Bf[S][decltype(i)::value] = gt_lds_rd<(KK * CPK + decltype(i)::value) * 1024>(aB); });
            gt_static_for<0, NTV>([&](auto j) { Af[S][decltype(j)::value] = gt_lds_rd<(KK * CPK + decltype(j)::value) * 1024>(aA); });
            if constexpr (NORM) Wn[S] = gt_lds_rd<KK * 64>(aN);
        };
        if (a.db & 2) { slot = slot + 1 == STAGES ? 0 : slot + 1; continue; }   // ablation: consumers only keep the barriers
        rd(std::integral_constant<int, 0>{});
        gt_static_for<0, KB>([&](auto kk_) {
            constexpr int KK = decltype(kk_)::value, S = KK & 1;
            if constexpr (KK + 1 < KB) {
                rd(std::integral_constant<int, KK + 1>{});
                gt_wait_lgkm<NRD>();
            } else {
                gt_wait_lgkm<0>();
            }
            gt_static_for<0, MTV>([&](auto i) { gt_touch(Bf[S][decltype(i)::value]); });
            gt_static_for<0, NTV>([&](auto j) { gt_touch(Af[S][decltype(j)::value]); });
            uint4 B[MTV];
            if constexpr (NORM) {
                gt_touch(Wn[S]);
                const uint4 wn4 = gt_u4(Wn[S]);
#pragma unroll
                for (int i = 0; i < MTV; ++i) B[i] = (a.db & 1) ? gt_u4(Bf[S][i]) : D::norm_chunk(gt_u4(Bf[S][i]), ri[i], wn4);
            } else {
#pragma unroll
                for (int i = 0; i < MTV; ++i) B[i] = gt_u4(Bf[S][i]);
            }
#pragma unroll
            for (int i = 0; i < MTV; ++i)
#pragma unroll
                for (int j = 0; j < NTV; ++j) acc[j][i] = D::mma(gt_u4(Af[S][j]), B[i], acc[j][i]);
        });
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }

    if (a.db & 8) return;   // ablation: no epilogue
    // ---- epilogue: this wave's NTV x MTV tiles ----
    if constexpr (EPI == EPI_QKV) {
        if constexpr (ROPE_LATE) load_rope();
        QkvCol qc;
        qc.init((ntw0 < ntiles ? ntw0 : ntiles - 1) * 16 + (lane >> 4) * 4, a.d, a.hd);
        const int r = lane & 15;
#pragma unroll
        for (int j = 0; j < NTV; ++j) {
            if (ntw0 + j < ntiles) {
#pragma unroll
                for (int i = 0; i < MTV; ++i) {
                    const int m = (mtw0 + i) * 16 + r;
                    const f32x4_t v = acc[j][i];
                    float x0 = D::rnd(v[0]), x1 = D::rnd(v[1]), x2 = D::rnd(v[2]), x3 = D::rnd(v[3]);
                    if (qc.sec < 2) {  // 2-D RoPE on interleaved (even, odd) pairs, fp32, one rounding (gpt.py:420-430)
                        const float fx = __uint_as_float(rope[j].x), fy = __uint_as_float(rope[j].y);
                        const float fz = __uint_as_float(rope[j].z), fw = __uint_as_float(rope[j].w);
                        rope_pair(x0, x1, fx, fy);
                        rope_pair(x2, x3, fz, fw);
                    }
                    if (m < a.M) {
                        if (qc.sec == 0) {
                            D::st4(a.out, ((size_t)m * a.H + qc.head) * a.hdp + qc.dd, x0, x1, x2, x3);
                        } else {
                            void* cache = qc.sec == 1 ? a.kc : a.vc;
                            D::st4(cache, (((size_t)m * a.H + qc.head) * a.S8 + pos) * a.kvs + qc.dd, x0, x1, x2, x3);
                        }
                    }
                }
            }
            qc.next16(a.hd, a.H);
        }
    } else if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int j = 0; j < NTV; j += 2) {
            if (ntw0 + j < ntiles) {
#pragma unroll
                for (int i = 0; i < MTV; ++i) {
                    // silu(w1 x) * (w3 x) with the hardware exp / reciprocal (v_exp_f32, v_rcp_f32: ~1e-7 relative, far below the
                    // 2^-9 half-ulp of the bf16 rounding that follows; the IEEE expf + division of the skinny kernels' epilogue
                    // cost ~30 VALU instructions per element: 3.3 of the 8 us this kernel spends outside its K loop at 640 rows)
                    const f32x4_t v = acc[j][i], v2 = acc[j + 1][i];
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = D::rnd(v[e]), y = D::rnd(v2[e]);
                        o[e] = D::rnd(x * __frcp_rn(1.0f + __expf(-x))) * y;
                    }
                    D::st4(a.out, D::xp_off(((ntw0 + j) >> 1) * 16 + (lane >> 4) * 4, mtw0 + i, lane & 15, a.MTs), o[0], o[1], o[2], o[3]);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NTV; ++j) {
            if (ntw0 + j < ntiles) {
#pragma unroll
                for (int i = 0; i < MTV; ++i) {
                    uint4 aux = make_uint4(0, 0, 0, 0);
                    if constexpr (EPI == EPI_RES) { aux.x = res[j][i].x; aux.y = res[j][i].y; }
                    epilogue<D, EPI>(a, ntw0 + j, mtw0 + i, lane, acc[j][i], acc[j][i], aux, 0);
                }
            }
        }
    }
}

static int gt_xr_override() {   // development switch LGEN_TILE_XR=0|1|2: log2 of the row split over XCDs (default: chosen per launch)
    static const int v = [] { const char* e = getenv("LGEN_TILE_XR"); return e ? atoi(e) : -1; }();
    return v < 0 || v > 2 ? -1 : v;
}

// ---- launch table ---------------------------------------------------------------------------------------------------------
template <typename D, int WM, int WN, int MTV, int NTV, int KB, int STAGES, int EPI, bool NORM, int LW>
static int gt_launch(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    constexpr int NW = WM * WN, MTW = WM * MTV, NTW = WN * NTV, CPS = KB * (MTW + NTW), STAGE_B = CPS * 1024;
    if (a.MTs % MTW || a.KCH % KB || a.KCH / KB < STAGES - 1) return LGEN_ERR_UNSUPPORTED;
    size_t lds = (size_t)STAGES * STAGE_B;
    a.passes = 1;
    if (NORM) {
        if (a.parts % 4) return LGEN_ERR_UNSUPPORTED;
        lds += ((size_t)a.KCH * D::KC * D::ESZ + 1023) / 1024 * 1024;
        const size_t share = ((size_t)MTV * 16 * a.parts * 4 + 1023) & ~(size_t)1023;   // as in the kernel: whole DMA rounds
        if (share > (size_t)STAGE_B / NW) {   // the statistics do not fit the idle ring slot: a region of their own ...
            a.passes = 2;
            if (lds + share * NW > 160 * 1024) a.passes = 3;   // ... or, when that does not fit either, read from global memory
            else lds += share * NW;
        }
    }
    if (lds > 160 * 1024) return LGEN_ERR_UNSUPPORTED;
    const int ntiles = a.N / 16;
    const int gx = (ntiles + NTW - 1) / NTW, gy = a.MTs / MTW;
    auto kern = gemm_tile_kernel<D, WM, WN, MTV, NTV, KB, STAGES, EPI, NORM, LW>;
    if (lds > 64 * 1024) {   // once per (instantiation, device): the most this kernel ever asks for
        static unsigned long long attr_set_mask = 0;
        const int dev_i = lgen_cur_dev();
        if (!((attr_set_mask >> dev_i) & 1)) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set_mask |= 1ull << dev_i;
        }
    }
    // rows split over XR groups of XCDs (see the block-id decode in the kernel): the balanced split with the fewest L2 fills
    int xrl = gt_xr_override();
    if (xrl < 0) {
        const double xb = (double)a.MTs * 16 * a.KCH, wb = (double)a.N * a.KCH;   // row panel / weights, in k-chunk rows
        double best = 8 * xb + wb;
        xrl = 0;
        for (int l = 1; l <= 2; ++l)
            if (gy % (1 << l) == 0 && (8 >> l) * xb + (1 << l) * wb < 0.95 * best) {
                best = (8 >> l) * xb + (1 << l) * wb;
                xrl = l;
            }
    } else if (gy % (1 << xrl)) {
        xrl = 0;
    }
    a.db = (a.db & ~0x300) | (xrl << 8);
    const int XR = 1 << xrl, XC = 8 >> xrl;
    hipLaunchKernelGGL(kern, dim3(8 * ((gx + XC - 1) / XC) * ((gy + XR - 1) / XR)), dim3(64 * (NW + LW)), lds, st, a);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// shapes (WM, WN, MTV, NTV, KB, STAGES, LW) with an instantiation; X(...) is expanded per (EPI, NORM) below
#define GT_SHAPES_NORM(X)                                                                                        \
    X(4, 1, 1, 3, 4, 4, 4) X(4, 1, 1, 4, 4, 4, 4) X(4, 1, 1, 6, 2, 4, 4) X(4, 1, 1, 6, 4, 3, 4) X(4, 1, 1, 8, 2, 4, 4) \
    X(4, 1, 2, 4, 2, 4, 4) X(4, 1, 2, 6, 2, 4, 4) X(4, 1, 2, 8, 2, 4, 4) X(4, 1, 1, 2, 4, 4, 4) X(4, 1, 1, 3, 4, 4, 0) \
    X(8, 1, 1, 8, 2, 4, 4) X(8, 1, 1, 6, 2, 4, 4) X(8, 1, 1, 4, 2, 4, 4) X(4, 1, 2, 10, 2, 4, 4)
#define GT_SHAPES_PLAIN(X)                                                                                       \
    X(2, 2, 1, 1, 4, 4, 4) X(2, 2, 1, 2, 4, 4, 4) X(2, 2, 2, 1, 4, 4, 4) X(2, 2, 2, 2, 4, 4, 4) X(2, 2, 2, 2, 2, 4, 4) \
    X(4, 1, 1, 2, 4, 4, 4) X(2, 2, 1, 1, 4, 4, 0) X(2, 2, 4, 1, 4, 4, 4) X(2, 2, 4, 2, 2, 4, 4)

template <typename D, int EPI, bool NORM>
static int gt_dispatch(const GemmArgs& a, int wm, int wn, int mtv, int ntv, int kb, int stages, int lw, hipStream_t st) {
#define GT_CASE(WM_, WN_, MTV_, NTV_, KB_, ST_, LW_)                                                                 \
    if (wm == WM_ && wn == WN_ && mtv == MTV_ && ntv == NTV_ && kb == KB_ && stages == ST_ && lw == LW_) {            \
        if constexpr ((EPI == EPI_SWIGLU && (NTV_ % 2)) || (EPI == EPI_QKV && WM_ * WN_ == 8 && NTV_ == 8)) /* (spills at 168 VGPRs) */ \
            return LGEN_ERR_UNSUPPORTED;                                                                              \
        else return gt_launch<D, WM_, WN_, MTV_, NTV_, KB_, ST_, EPI, NORM, LW_>(a, st);                               \
    }
    if constexpr (NORM) {
        GT_SHAPES_NORM(GT_CASE)
    } else {
        GT_SHAPES_PLAIN(GT_CASE)
    }
#undef GT_CASE
    return LGEN_ERR_UNSUPPORTED;
}

static int gt_ablate() {   // development switch: bit 0 no RMSNorm VALU work, bit 1 no LDS reads / MFMAs, bit 2 no operand DMA, bit 3 no epilogue, bit 4 no statistics prologue
    static const int v = [] { const char* e = getenv("LGEN_TILE_ABLATE"); return (e ? atoi(e) : 0) & 0xff; }();   // read once per process
    return v;
}

static int gt_dispatch_epi(const GemmArgs& a, int epi, int wm, int wn, int mtv, int ntv, int kb, int stages, int lw, hipStream_t st) {
    if (a.nw) {
        if (!a.ssq_in || a.parts < 1) return LGEN_ERR_BAD_ARG;
        switch (epi) {
            case EPI_QKV: return gt_dispatch<BF16, EPI_QKV, true>(a, wm, wn, mtv, ntv, kb, stages, lw, st);
            case EPI_SWIGLU: return gt_dispatch<BF16, EPI_SWIGLU, true>(a, wm, wn, mtv, ntv, kb, stages, lw, st);
            case EPI_ROWS: return gt_dispatch<BF16, EPI_ROWS, true>(a, wm, wn, mtv, ntv, kb, stages, lw, st);
            default: return LGEN_ERR_UNSUPPORTED;
        }
    }
    switch (epi) {
        case EPI_RES: return gt_dispatch<BF16, EPI_RES, false>(a, wm, wn, mtv, ntv, kb, stages, lw, st);
        default: return LGEN_ERR_UNSUPPORTED;
    }
}

// LGEN_GEMM_TILE_* entry points (include/lgen.h): same operand contract as lgen_gemm / lgen_gemm_qkv_rope, the workgroup shape
// is explicit.  bf16 storage only; LGEN_ERR_UNSUPPORTED for a shape without an instantiation (the caller picks another one).
extern "C" int lgen_gemm_tile(const void* wp, const void* xp, void* out, int M, int MTs, int N, int K, int epilogue_kind, int dtype,
                              int wm, int wn, int mtv, int ntv, int kb, int stages, int lw, const void* norm_w, const float* ssq_in,
                              int ssq_parts, float eps, float* ssq_out, void* stream) {
    if (dtype != LGEN_BF16) return LGEN_ERR_UNSUPPORTED;
    if (N % 16 || K % 32 || M > MTs * 16 || M < 1) return LGEN_ERR_BAD_ARG;
    if ((norm_w && (ssq_parts < 1 || ssq_parts > LGEN_SSQ_STRIDE)) || (ssq_out && N / 16 > LGEN_SSQ_STRIDE)) return LGEN_ERR_BAD_ARG;
    if (ssq_out && epilogue_kind != LGEN_EPI_RES) return LGEN_ERR_BAD_ARG;
    if (epilogue_kind == LGEN_EPI_QKV) return LGEN_ERR_BAD_ARG;
    GemmArgs a{};
    a.wp = (const uint4*)wp; a.xp = (const uint4*)xp; a.out = out;
    a.N = N; a.KCH = K / 32; a.MTs = MTs; a.M = M;
    a.nw = (const uint4*)norm_w; a.ssq_in = ssq_in; a.parts = ssq_parts; a.eps = eps; a.inv_k = 1.0f / (float)K;
    a.ssq_out = ssq_out;
    a.passes = 1;
    a.db = gt_ablate();
    return gt_dispatch_epi(a, epilogue_kind, wm, wn, mtv, ntv, kb, stages, lw, (hipStream_t)stream);
}

extern "C" int lgen_gemm_qkv_rope_tile(const void* wp, const void* xp, void* q_out, void* k_cache, void* v_cache, const float* freqs,
                                       const int* pos_ptr, int M, int MTs, int d, int n_head, int hd, int hdp, int S8,
                                       int kv_row_stride, int dtype, int wm, int wn, int mtv, int ntv, int kb, int stages, int lw,
                                       const void* norm_w, const float* ssq_in, int ssq_parts, float eps, void* stream) {
    if (dtype != LGEN_BF16) return LGEN_ERR_UNSUPPORTED;
    if (d % 32 || (3 * d) % 16 || hd % 4 || hd < 16 || d != n_head * hd || M > MTs * 16 || M < 1) return LGEN_ERR_BAD_ARG;
    if (norm_w && (ssq_parts < 1 || ssq_parts > LGEN_SSQ_STRIDE)) return LGEN_ERR_BAD_ARG;
    GemmArgs a{};
    a.wp = (const uint4*)wp; a.xp = (const uint4*)xp; a.out = q_out; a.kc = k_cache; a.vc = v_cache;
    a.freqs = freqs; a.pos_ptr = pos_ptr; a.pos_stride = 0;
    a.N = 3 * d; a.KCH = d / 32; a.MTs = MTs; a.M = M;
    a.d = d; a.hd = hd; a.hdp = hdp; a.H = n_head; a.S8 = S8;
    a.kvs = kv_row_stride > 0 ? kv_row_stride : hdp;
    if (a.kvs < ((hd + 7) & ~7) || a.kvs % 8) return LGEN_ERR_BAD_ARG;   // (rows may be packed tighter than the lane group hdp: lgen.h)
    a.nw = (const uint4*)norm_w; a.ssq_in = ssq_in; a.parts = ssq_parts; a.eps = eps; a.inv_k = 1.0f / (float)d;
    a.passes = 1;
    a.db = gt_ablate();
    if (!norm_w) return LGEN_ERR_UNSUPPORTED;
    return gt_dispatch<BF16, EPI_QKV, true>(a, wm, wn, mtv, ntv, kb, stages, lw, (hipStream_t)stream);
}
