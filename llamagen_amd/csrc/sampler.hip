// Fused sampler: classifier-free-guidance mix -> /temperature -> top-k threshold (k-th largest,
// ties kept) -> [top-p nucleus] -> softmax -> argmax(p / q) with q ~ Exp(1).
//
// Replaces autoregressive/models/generate.py:79-86, 94-99 (CFG split/mix), :57-66 sample(),
// :16-54 top_k_top_p_filtering and torch.multinomial(num_samples=1), which ATen evaluates as
// argmax(p / q), q = empty_like(p).exponential_(1) (pinned in
// tests/test_oracle_golden.py).  The Exp(1) draw itself stays with torch (same Philox
// consumption as the reference); everything else is one kernel with no host sync -- the
// reference's boolean-mask assignment and multinomial validity checks each sync the host.
//
// One workgroup (1024 threads) per image row; the whole row (V <= 16384 fp32 = 64 KiB) lives in
// LDS.  k-th largest by MSB-first radix select on order-preserving uint keys (4 passes of 8
// bits; pass 0 uses wave-aggregated histogram updates because logits share few exponents).
#include "lgen_common.h"
#include "../../include/lgen.h"

#define SMP_THREADS 1024
#define SMP_MAXV 16384

LGEN_DEV uint32_t fkey(float f) {  // ascending order-preserving map float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SampleArgs {
    const void* logits;    // [>=B2][V] storage dtype, rows [0,B) cond, [B,2B) uncond when cfg
    const float* noise;    // Exp(1) draws of this step at noise + step*noise_stride, [B][V] (null when greedy)
    long long noise_stride;
    int* cur_tok;          // [2B or B] token fed to the next step (both CFG halves)
    int* seq;              // [B][seq_stride] output ids, column = step
    const int* state;      // [0] = pos, [1] = step (read only; the embed kernel of the next step advances them)
    int B, V, seq_stride, use_cfg;
    float cfg_scale, temperature, top_p;
    int cfg_interval, top_k, greedy;
};

// Ownership: thread t owns the 8 consecutive vocabulary entries of slot s at i = (s*1024 + t)*8
// (s < NS = 2): every wave-level global access is a contiguous 1 KiB (bf16 logits) / 2 KiB (fp32)
// run, and ALL of a thread's loads (cond, uncond, noise) are issued before anything is consumed.
// The radix-select passes in between walk the LDS copy with the bank-conflict-free stride-1024
// ownership instead.
#define SMP_NS 2
template <typename D>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float vals[];  // [V]
    __shared__ unsigned hist[256];
    __shared__ float red_f[SMP_THREADS / 64];
    __shared__ int red_i[SMP_THREADS / 64];
    __shared__ unsigned sel_prefix, sel_k;
    __shared__ float sh_f;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x, V = a.V, B = a.B;
    const int step = a.state[1];
    // generate.py:113-114: decode iteration i = step-1 drops guidance once i > cfg_interval
    const bool mix = a.use_cfg && !(step > 0 && a.cfg_interval > -1 && (step - 1) > a.cfg_interval);
    const float tdiv = fmaxf(a.temperature, 1e-5f);

    // 0. request everything this thread will ever read from HBM
    float lc[SMP_NS][8], lu[SMP_NS][8], nz[SMP_NS][8];
    bool own[SMP_NS];
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
        const int i0 = (s * SMP_THREADS + tid) * 8;
        own[s] = i0 < V;  // V % 8 == 0
        const int ic = own[s] ? i0 : 0;
        D::ld8(a.logits, (size_t)b * V + ic, lc[s]);
        if (mix) D::ld8(a.logits, (size_t)(B + b) * V + ic, lu[s]);
        if (!a.greedy) {
            const float4* np = (const float4*)(a.noise + (size_t)step * a.noise_stride + (size_t)b * V + ic);
            const float4 n0 = np[0], n1 = np[1];
            nz[s][0] = n0.x; nz[s][1] = n0.y; nz[s][2] = n0.z; nz[s][3] = n0.w;
            nz[s][4] = n1.x; nz[s][5] = n1.y; nz[s][6] = n1.z; nz[s][7] = n1.w;
        }
    }

    // 1. CFG mix + temperature -> LDS, row max
    float lmax = -INFINITY;
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
        if (own[s]) {
            float l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = lc[s][e];
                if (mix) v = lu[s][e] + (v - lu[s][e]) * a.cfg_scale;
                v = v / tdiv;
                l[e] = v;
                lmax = fmaxf(lmax, v);
            }
            float4* vp = (float4*)(vals + (s * SMP_THREADS + tid) * 8);
            vp[0] = make_float4(l[0], l[1], l[2], l[3]);
            vp[1] = make_float4(l[4], l[5], l[6], l[7]);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red_f[wv] = lmax;
    __syncthreads();
    if (tid == 0) {
        float m = red_f[0];
        for (int i = 1; i < SMP_THREADS / 64; ++i) m = fmaxf(m, red_f[i]);
        sh_f = m;
    }
    __syncthreads();
    const float rmax = sh_f;

    // 2. top-k: key of the k-th largest value (strict '<' removal keeps ties, generate.py:35)
    uint32_t thr_key = 0;  // keep everything
    int k = a.top_k;
    if (k > 0) k = k < 1 ? 1 : (k > V ? V : k);
    if (k > 0 && k < V) {
        if (tid == 0) { sel_prefix = 0; sel_k = (unsigned)k; }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = sel_prefix;
            const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < V; i += SMP_THREADS) {
                const uint32_t key = fkey(vals[i]);
                bool act = (key & pmask) == prefix;
                const unsigned bin = (key >> shift) & 0xffu;
                if (pass == 0) {  // wave-aggregated: few distinct exponent bins
                    unsigned long long todo = __ballot(act);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned lb = __shfl(bin, leader, 64);
                        const unsigned long long same = __ballot(act && bin == lb);
                        if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(same));
                        todo &= ~same;
                        if (bin == lb) act = false;
                    }
                } else if (act) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
            if (wv == 0) {  // one wave scans the 256 bins from the top (4 bins per lane)
                const unsigned c0 = hist[255 - 4 * lane], c1 = hist[254 - 4 * lane];
                const unsigned c2 = hist[253 - 4 * lane], c3 = hist[252 - 4 * lane];
                const unsigned s4 = c0 + c1 + c2 + c3;
                unsigned incl = s4;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += t;
                }
                const unsigned excl = incl - s4;
                const unsigned kk = sel_k;
                if (excl < kk && kk <= incl) {  // exactly one lane: the k-th largest is in its bins
                    unsigned cum = excl;
                    int bsel = 255 - 4 * lane;
                    if (cum + c0 < kk) {
                        cum += c0; bsel -= 1;
                        if (cum + c1 < kk) {
                            cum += c1; bsel -= 1;
                            if (cum + c2 < kk) { cum += c2; bsel -= 1; }
                        }
                    }
                    sel_k = kk - cum;
                    sel_prefix = prefix | ((unsigned)bsel << shift);
                }
            }
            __syncthreads();
        }
        thr_key = sel_prefix;
    }

    // 3. softmax over kept entries (max of kept == row max) and argmax(p / q), on the owned slots
    float ex[SMP_NS][8];
    float lsum = 0.f;
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
        if (own[s]) {
            const float4* vp = (const float4*)(vals + (s * SMP_THREADS + tid) * 8);
            const float4 a0 = vp[0], a1 = vp[1];
            const float l[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ex[s][e] = fkey(l[e]) >= thr_key ? expf(l[e] - rmax) : 0.f;
                lsum += ex[s][e];
            }
        }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red_f[wv] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < SMP_THREADS / 64; ++i) s += red_f[i];
        sh_f = s;
    }
    __syncthreads();
    const float tot = sh_f;

    float best = -1.f;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
        if (own[s]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float p = ex[s][e] / tot;
                const float r = a.greedy ? p : p / nz[s][e];
                if (r > best) { best = r; bidx = (s * SMP_THREADS + tid) * 8 + e; }  // ascending index: first maximum wins
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    __syncthreads();
    if (lane == 0) { red_f[wv] = best; red_i[wv] = bidx; }
    __syncthreads();
    if (tid == 0) {
        float bb = red_f[0];
        int bi = red_i[0];
        for (int i = 1; i < SMP_THREADS / 64; ++i)
            if (red_f[i] > bb || (red_f[i] == bb && red_i[i] < bi)) { bb = red_f[i]; bi = red_i[i]; }
        a.cur_tok[b] = bi;
        if (a.use_cfg) a.cur_tok[B + b] = bi;
        a.seq[(size_t)b * a.seq_stride + step] = bi;
    }
}

// (pos, step) += 1 outside a decode step (the decode step's embed kernel advances them itself).
__global__ void advance_state_kernel(int* state) {
    state[0] += 1;
    state[1] += 1;
}

extern "C" int lgen_sample(const void* logits, const float* noise, long long noise_step_stride, int* cur_tok, int* seq,
                           const int* state, int B, int V, int seq_stride, int use_cfg, float cfg_scale,
                           int cfg_interval, float temperature, int top_k, float top_p, int greedy, int dtype,
                           void* stream) {
    if (V > SMP_MAXV || V < 8 || (V & 7) || B < 1) return LGEN_ERR_BAD_ARG;
    if (top_p < 1.0f) return LGEN_ERR_UNSUPPORTED;
    if (!greedy && !noise) return LGEN_ERR_BAD_ARG;
    SampleArgs a{logits, noise, noise_step_stride, cur_tok, seq, state, B, V, seq_stride, use_cfg, cfg_scale,
                 temperature, top_p, cfg_interval, top_k, greedy};
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)V * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {  // row (64 KiB) + small statics exceeds the default 64 KiB LDS cap
        hipError_t e1 = hipFuncSetAttribute((const void*)sample_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SMP_MAXV * (int)sizeof(float));
        hipError_t e2 = hipFuncSetAttribute((const void*)sample_kernel<F32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SMP_MAXV * (int)sizeof(float));
        if (e1 != hipSuccess) return (int)e1;
        if (e2 != hipSuccess) return (int)e2;
        attr_set = true;
    }
    if (dtype == LGEN_BF16)
        hipLaunchKernelGGL(sample_kernel<BF16>, dim3(B), dim3(SMP_THREADS), lds, st, a);
    else if (dtype == LGEN_F32)
        hipLaunchKernelGGL(sample_kernel<F32>, dim3(B), dim3(SMP_THREADS), lds, st, a);
    else
        return LGEN_ERR_BAD_ARG;
    LGEN_CHECK_LAUNCH();
    return 0;
}

extern "C" int lgen_advance_state(int* state, void* stream) {
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    LGEN_CHECK_LAUNCH();
    return 0;
}
