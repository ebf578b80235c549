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
// One workgroup (1024 threads) per image row; the row (V <= 16384) lives in registers, 16 values
// per thread.  k-th largest by a one-pass value-bin histogram + exact rank inside the critical bin
// (MSB-first radix select on an LDS copy as the exact fallback for degenerate rows).
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
    // continuous batching (all null / 0 otherwise): slot b is its own request at step row_step[b] of max_steps; its noise block
    // is noise[(b * max_steps + step) * V ...]; a finished / empty slot (step >= max_steps) is skipped; after sampling the
    // slot's step and the positions of its (cond, uncond) rows advance
    int* row_step;         // [B]
    int* row_pos;          // [2B or B]
    int max_steps;
};

// Ownership: thread t owns the 8 consecutive vocabulary entries of slot s at i = (s*1024 + t)*8
// (s < NS = 2): every wave-level global access is a contiguous 1 KiB (bf16 logits) / 2 KiB (fp32)
// run, ALL of a thread's loads (cond, uncond, noise) are issued before anything is consumed, and the
// row stays in registers.
//
// k-th largest value (top-k threshold, ties kept): ONE histogram pass over NB = 4096 equal-width value
// bins between the row minimum and maximum (monotone binning: a higher bin holds strictly larger
// values), a suffix scan that finds the bin holding the k-th largest, and an exact rank among the few
// values of that bin.  Logits are spread over many bins, so the LDS atomics rarely collide -- the
// bit-radix select this replaces spent ~30 us in its first pass, where all values share 3-4 exponent
// bins.  Rows the histogram cannot resolve (more than SMP_CAND values in the critical bin, e.g. constant
// rows, or non-finite values) take the exact MSB-first radix select on an LDS copy instead.
#define SMP_NS 2
static_assert(SMP_NS == 2, "the top-p tie ranking (tie_w[0] / tie_w[1], index order (s, thread, e)) is written for two slices per thread");
#define SMP_NB 4096
#define SMP_CAND 1024

// exact k-th largest key of vals[0..V) (order-preserving uint keys), 4 x 8-bit MSB-first radix passes
LGEN_DEV uint32_t radix_kth_key(const float* vals, int V, int k, unsigned* hist, unsigned* sel /* [2] */) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { sel[0] = 0; sel[1] = (unsigned)k; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = sel[0];
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
            const unsigned kk = sel[1];
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
                sel[1] = kk - cum;
                sel[0] = prefix | ((unsigned)bsel << shift);
            }
        }
        __syncthreads();
    }
    return sel[0];
}

template <typename D>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float vals[];  // [V]: only the radix fallback uses it
    __shared__ unsigned hist[SMP_NB];
    __shared__ float cand[SMP_CAND];
    __shared__ float red_f[SMP_THREADS / 64], red_g[SMP_THREADS / 64];
    __shared__ int red_i[SMP_THREADS / 64];
    __shared__ unsigned wtot[SMP_THREADS / 64];
    __shared__ unsigned sel[2], ncand;
    __shared__ float sh_f, sh_g, sh_thr;
    __shared__ unsigned long long hm[SMP_NB];       // top-p: probability mass per value bin, 2^-40 fixed point
    __shared__ unsigned long long wtot64[SMP_THREADS / 64], sh_gab;
    __shared__ unsigned sh_key;
    __shared__ unsigned tie_w[SMP_NS][SMP_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x, V = a.V, B = a.B;
    const int step = a.row_step ? a.row_step[b] : a.state[1];
    if (a.row_step && step >= a.max_steps) return;  // empty / finished slot (uniform over the workgroup)
    // generate.py:113-114: decode iteration i = step-1 drops guidance once i > cfg_interval
    const bool mix = a.use_cfg && !(step > 0 && a.cfg_interval > -1 && (step - 1) > a.cfg_interval);
    const float tdiv = fmaxf(a.temperature, 1e-5f);

    // 0. request everything this thread will ever read from HBM
    float l[SMP_NS][8], lu[SMP_NS][8], nz[SMP_NS][8];
    bool own[SMP_NS];
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
        const int i0 = (s * SMP_THREADS + tid) * 8;
        own[s] = i0 < V;  // V % 8 == 0
        const int ic = own[s] ? i0 : 0;
        D::ld8(a.logits, (size_t)b * V + ic, l[s]);
        if (mix) D::ld8(a.logits, (size_t)(B + b) * V + ic, lu[s]);
        if (!a.greedy) {
            const float4* np = a.row_step ? (const float4*)(a.noise + ((size_t)b * a.max_steps + step) * V + ic)
                                          : (const float4*)(a.noise + (size_t)step * a.noise_stride + (size_t)b * V + ic);
            const float4 n0 = np[0], n1 = np[1];
            nz[s][0] = n0.x; nz[s][1] = n0.y; nz[s][2] = n0.z; nz[s][3] = n0.w;
            nz[s][4] = n1.x; nz[s][5] = n1.y; nz[s][6] = n1.z; nz[s][7] = n1.w;
        }
    }
    for (int i = tid; i < SMP_NB; i += SMP_THREADS) hist[i] = 0;
    if (tid == 0) ncand = 0;

    // 1. CFG mix + temperature (in registers), row max / min
    float lmax = -INFINITY, lmin = INFINITY;
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = l[s][e];
            if (mix) v = lu[s][e] + (v - lu[s][e]) * a.cfg_scale;
            v = v / tdiv;
            l[s][e] = v;
            if (own[s]) { lmax = fmaxf(lmax, v); lmin = fminf(lmin, v); }
        }
    }
    lmax = wave_max(lmax);
    lmin = -wave_max(-lmin);
    if (lane == 0) { red_f[wv] = lmax; red_g[wv] = lmin; }
    __syncthreads();
    if (tid == 0) {
        float m = red_f[0], n = red_g[0];
        for (int i = 1; i < SMP_THREADS / 64; ++i) { m = fmaxf(m, red_f[i]); n = fminf(n, red_g[i]); }
        sh_f = m; sh_g = n;
    }
    __syncthreads();
    const float rmax = sh_f, rmin = sh_g;

    // 2. top-k threshold = k-th largest value (strict '<' removal keeps ties, generate.py:35)
    float thr = -INFINITY;  // keep everything
    int k = a.top_k;
    if (k > 0) k = k < 1 ? 1 : (k > V ? V : k);
    if (k > 0 && k < V) {
        const float range = rmax - rmin;
        bool ok = range > 0.f && range < 3.0e38f;  // finite, non-constant row
        const float scale = ok ? (float)SMP_NB / range : 0.f;
        int bins[SMP_NS][8];
        if (ok) {
#pragma unroll
            for (int s = 0; s < SMP_NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int bi = (int)((l[s][e] - rmin) * scale);
                    bi = bi > SMP_NB - 1 ? SMP_NB - 1 : bi;
                    bins[s][e] = bi;
                    if (own[s]) atomicAdd(&hist[bi], 1u);
                }
        }
        __syncthreads();
        if (ok) {
            // suffix scan over bins, 4 per thread: above = #values in bins higher than this thread's
            const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const unsigned s4 = c0 + c1 + c2 + c3;
            unsigned incl = s4;  // inclusive suffix sum within the wave (lanes >= this one)
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += t;
            }
            if (lane == 0) wtot[wv] = incl;
            __syncthreads();
            unsigned above = incl - s4;
            for (int w2 = wv + 1; w2 < SMP_THREADS / 64; ++w2) above += wtot[w2];
            const unsigned kk = (unsigned)k;
            if (above < kk && kk <= above + s4) {  // exactly one thread
                unsigned cum = above;
                int bsel = 4 * tid + 3;
                if (cum + c3 < kk) {
                    cum += c3; bsel -= 1;
                    if (cum + c2 < kk) {
                        cum += c2; bsel -= 1;
                        if (cum + c1 < kk) { cum += c1; bsel -= 1; }
                    }
                }
                sel[0] = (unsigned)bsel;
                sel[1] = kk - cum;  // rank (1 = largest) inside the critical bin
            }
            __syncthreads();
            const int bsel = (int)sel[0];
#pragma unroll
            for (int s = 0; s < SMP_NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (own[s] && bins[s][e] == bsel) {
                        const unsigned slot = atomicAdd(&ncand, 1u);
                        if (slot < SMP_CAND) cand[slot] = l[s][e];
                    }
            __syncthreads();
            const unsigned nc = ncand;
            if (nc <= SMP_CAND) {
                const unsigned need = sel[1];
                if ((unsigned)tid < nc) {
                    const float ci = cand[tid];
                    unsigned gt = 0, ge = 0;
                    for (unsigned j = 0; j < nc; ++j) {
                        const float cj = cand[j];
                        gt += cj > ci;
                        ge += cj >= ci;
                    }
                    if (gt < need && need <= ge) sh_thr = ci;  // equal candidates write the same value
                }
                __syncthreads();
                thr = sh_thr;
            } else {
                ok = false;  // block-uniform: nc comes from LDS
            }
        }
        if (!ok) {  // exact radix select on an LDS copy of the row
#pragma unroll
            for (int s = 0; s < SMP_NS; ++s)
                if (own[s]) {
                    float4* vp = (float4*)(vals + (s * SMP_THREADS + tid) * 8);
                    vp[0] = make_float4(l[s][0], l[s][1], l[s][2], l[s][3]);
                    vp[1] = make_float4(l[s][4], l[s][5], l[s][6], l[s][7]);
                }
            __syncthreads();
            const uint32_t key = radix_kth_key(vals, V, k, hist, sel);
            // invert fkey: the threshold value itself
            const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            thr = __uint_as_float(u);
        }
    }

    // 3. softmax over kept entries (max of kept == row max) and argmax(p / q), on the owned slots
    float ex[SMP_NS][8];
    float lsum = 0.f;
#pragma unroll
    for (int s = 0; s < SMP_NS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ex[s][e] = (own[s] && l[s][e] >= thr) ? expf(l[s][e] - rmax) : 0.f;
            lsum += ex[s][e];
        }
    lsum = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) red_f[wv] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < SMP_THREADS / 64; ++i) s += red_f[i];
        sh_f = s;
    }
    __syncthreads();
    float tot = sh_f;

    // 3b. nucleus (top-p) filter on the top-k survivors (generate.py:38-53): an entry stays iff the probability
    // mass of the strictly larger entries is <= top_p (the reference's shifted `cumsum > top_p` removal); of the entries
    // equal to that threshold value the lowest-index ones stay, as many as the reference's sort order keeps.  Masses are
    // 2^-40 fixed point, so every sum is order-independent: value-bin mass histogram -> critical bin by suffix
    // scan -> exact mass ranking inside that bin.
    if (a.top_p < 1.0f) {
        const float range = rmax - rmin;
        if (range > 0.f && range < 3.0e38f) {  // a constant row keeps everything (no strictly larger entry)
            const float scale = (float)SMP_NB / range;
            const unsigned long long P = (unsigned long long)((double)a.top_p * 1099511627776.0);
            auto mass = [&](float e) { return (unsigned long long)((double)(e / tot) * 1099511627776.0); };
            for (int i = tid; i < SMP_NB; i += SMP_THREADS) hm[i] = 0ull;
            if (tid == 0) { ncand = 0; sh_key = 0xffffffffu; sel[0] = 0xffffffffu; }
            __syncthreads();
            int pb[SMP_NS][8];
#pragma unroll
            for (int s = 0; s < SMP_NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int bi = (int)((l[s][e] - rmin) * scale);
                    bi = bi > SMP_NB - 1 ? SMP_NB - 1 : bi;
                    pb[s][e] = bi;
                    if (ex[s][e] > 0.f) atomicAdd(&hm[bi], mass(ex[s][e]));
                }
            __syncthreads();
            const unsigned long long m0 = hm[4 * tid], m1 = hm[4 * tid + 1], m2 = hm[4 * tid + 2], m3 = hm[4 * tid + 3];
            const unsigned long long s4 = m0 + m1 + m2 + m3;
            unsigned long long incl = s4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long t = __shfl_down(incl, o, 64);
                if (lane + o < 64) incl += t;
            }
            if (lane == 0) wtot64[wv] = incl;
            __syncthreads();
            unsigned long long above = incl - s4;
            for (int w2 = wv + 1; w2 < SMP_THREADS / 64; ++w2) above += wtot64[w2];
            {   // critical bin b: mass above b <= P < mass above b + mass of b (walk this thread's bins from the top)
                unsigned long long g = above;
                const unsigned long long mm[4] = {m0, m1, m2, m3};
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (g <= P && P < g + mm[q]) { sel[0] = (unsigned)(4 * tid + q); sh_gab = g; }
                    g += mm[q];
                }
            }
            __syncthreads();
            const unsigned bstar = sel[0];
            if (bstar != 0xffffffffu) {  // else: total mass <= top_p (rounding), nothing to remove
                float* cv = vals;         // candidate values of the critical bin (capacity V)
#pragma unroll
                for (int s = 0; s < SMP_NS; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ex[s][e] > 0.f && (unsigned)pb[s][e] == bstar) cv[atomicAdd(&ncand, 1u)] = l[s][e];
                __syncthreads();
                const unsigned nc = ncand;
                const unsigned long long gab = sh_gab;
                for (unsigned i = tid; i < nc; i += SMP_THREADS) {
                    const float ci = cv[i];
                    unsigned long long g = gab;
                    for (unsigned j = 0; j < nc; ++j) {
                        const float cj = cv[j];
                        if (cj > ci) g += mass(expf(cj - rmax));
                    }
                    if (g <= P) atomicMin(&sh_key, fkey(ci));
                }
                __syncthreads();
                const uint32_t key = sh_key;
                const float thr_p = __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
                // Ties AT the threshold value.  The reference sorts descending and its shifted `cumsum > top_p` removes the tail of a
                // tie group in SORT order; torch's GPU sort is a stable radix sort (ties in ascending index -- the CPU sort of the same
                // call is unstable and keeps an arbitrary subset of the same SIZE), so: of the c entries equal to thr_p keep the
                // n_keep = floor((P - G) / m) + 1 lowest-index ones (G = mass strictly above thr_p, m = mass of one such entry).
                unsigned long long gpart = 0ull;
                unsigned cnt[SMP_NS];
#pragma unroll
                for (int s = 0; s < SMP_NS; ++s) {
                    cnt[s] = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ex[s][e] > 0.f) {
                            if (l[s][e] > thr_p) gpart += mass(ex[s][e]);
                            else if (l[s][e] == thr_p) cnt[s] += 1;
                        }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gpart += __shfl_xor(gpart, o, 64);
                unsigned incl_c[SMP_NS];  // inclusive scan of the tie counts over the lanes of this wave (index order = (s, thread, e))
#pragma unroll
                for (int s = 0; s < SMP_NS; ++s) {
                    unsigned v = cnt[s];
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned u = __shfl_up(v, o, 64);
                        if (lane >= o) v += u;
                    }
                    incl_c[s] = v;
                }
                __syncthreads();
                if (lane == 63) { tie_w[0][wv] = incl_c[0]; tie_w[1][wv] = incl_c[1]; }
                if (lane == 0) wtot64[wv] = gpart;
                __syncthreads();
                unsigned long long G = 0ull;
                unsigned before[SMP_NS], total_c = 0, tot0 = 0;
                for (int i = 0; i < SMP_THREADS / 64; ++i) { G += wtot64[i]; tot0 += tie_w[0][i]; total_c += tie_w[0][i] + tie_w[1][i]; }
                before[0] = incl_c[0] - cnt[0];
                before[1] = tot0 + incl_c[1] - cnt[1];
                for (int i = 0; i < wv; ++i) { before[0] += tie_w[0][i]; before[1] += tie_w[1][i]; }
                const unsigned long long m1 = mass(expf(thr_p - rmax));
                unsigned long long nk = m1 > 0ull ? (P >= G ? (P - G) / m1 + 1ull : 1ull) : (unsigned long long)total_c;
                const unsigned n_keep = nk > (unsigned long long)total_c ? total_c : (unsigned)nk;
                // re-normalise over the nucleus
                lsum = 0.f;
#pragma unroll
                for (int s = 0; s < SMP_NS; ++s) {
                    unsigned rank = before[s];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (!(l[s][e] >= thr_p)) ex[s][e] = 0.f;
                        else if (ex[s][e] > 0.f && l[s][e] == thr_p) {
                            if (rank >= n_keep) ex[s][e] = 0.f;
                            rank += 1;
                        }
                        lsum += ex[s][e];
                    }
                }
                lsum = wave_sum(lsum);
                if (lane == 0) red_f[wv] = lsum;
                __syncthreads();
                if (tid == 0) {
                    float sacc = 0.f;
                    for (int i = 0; i < SMP_THREADS / 64; ++i) sacc += red_f[i];
                    sh_f = sacc;
                }
                __syncthreads();
                tot = sh_f;
            }
        }
    }

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
        // a row whose every ratio is NaN (non-finite logits or noise, e.g. a filler row fed uninitialised memory) has no maximum:
        // token 0 instead of the 0x7fffffff sentinel, which the next step's embedding lookup would follow out of bounds
        bi = (unsigned)bi < (unsigned)a.V ? bi : 0;
        a.cur_tok[b] = bi;
        if (a.use_cfg) a.cur_tok[B + b] = bi;
        a.seq[(size_t)b * a.seq_stride + step] = bi;
        if (a.row_step) {
            a.row_step[b] = step + 1;
            a.row_pos[b] += 1;
            if (a.use_cfg) a.row_pos[B + b] += 1;
        }
    }
}

// (pos, step) += 1 outside a decode step (the decode step's embed kernel advances them itself).
__global__ void advance_state_kernel(int* state) {
    state[0] += 1;
    state[1] += 1;
}

static int sample_impl(const void* logits, const float* noise, long long noise_step_stride, int* cur_tok, int* seq,
                       const int* state, int B, int V, int seq_stride, int use_cfg, float cfg_scale,
                       int cfg_interval, float temperature, int top_k, float top_p, int greedy, int dtype,
                       int* row_step, int* row_pos, int max_steps, void* stream) {
    if (V > SMP_MAXV || V < 8 || (V & 7) || B < 1) return LGEN_ERR_BAD_ARG;
    if (!greedy && !noise) return LGEN_ERR_BAD_ARG;
    SampleArgs a{logits, noise, noise_step_stride, cur_tok, seq, state, B, V, seq_stride, use_cfg, cfg_scale,
                 temperature, top_p, cfg_interval, top_k, greedy, row_step, row_pos, max_steps};
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)SMP_MAXV * sizeof(float);
    static unsigned long long attr_set_mask = 0;   // per device
    const int dev_i = lgen_cur_dev();
    if (!((attr_set_mask >> dev_i) & 1)) {  // row (64 KiB) + small statics exceeds the default 64 KiB LDS cap
        hipError_t e1 = hipFuncSetAttribute((const void*)sample_kernel<BF16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SMP_MAXV * (int)sizeof(float));
        hipError_t e2 = hipFuncSetAttribute((const void*)sample_kernel<F32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SMP_MAXV * (int)sizeof(float));
        hipError_t e3 = hipFuncSetAttribute((const void*)sample_kernel<F16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            SMP_MAXV * (int)sizeof(float));
        if (e3 != hipSuccess) return (int)e3;
        if (e1 != hipSuccess) return (int)e1;
        if (e2 != hipSuccess) return (int)e2;
        attr_set_mask |= 1ull << dev_i;
    }
    if (dtype == LGEN_BF16)
        hipLaunchKernelGGL(sample_kernel<BF16>, dim3(B), dim3(SMP_THREADS), lds, st, a);
    else if (dtype == LGEN_F32)
        hipLaunchKernelGGL(sample_kernel<F32>, dim3(B), dim3(SMP_THREADS), lds, st, a);
    else if (dtype == LGEN_F16)
        hipLaunchKernelGGL(sample_kernel<F16>, dim3(B), dim3(SMP_THREADS), lds, st, a);
    else
        return LGEN_ERR_BAD_ARG;
    LGEN_CHECK_LAUNCH();
    return 0;
}

extern "C" int lgen_sample(const void* logits, const float* noise, long long noise_step_stride, int* cur_tok, int* seq,
                           const int* state, int B, int V, int seq_stride, int use_cfg, float cfg_scale,
                           int cfg_interval, float temperature, int top_k, float top_p, int greedy, int dtype,
                           void* stream) {
    return sample_impl(logits, noise, noise_step_stride, cur_tok, seq, state, B, V, seq_stride, use_cfg, cfg_scale, cfg_interval,
                       temperature, top_k, top_p, greedy, dtype, nullptr, nullptr, 0, stream);
}

// Continuous batching form (autoregressive/serve/sampler.py:54-58,106-108: paired cond / uncond sequences, every request at its
// own step): slot b samples step row_step[b] < max_steps with noise[(b*max_steps + step)*V ..], writes seq[b][step] and
// cur_tok[b] (and cur_tok[B+b]), then advances row_step[b] and row_pos[b] (and row_pos[B+b]); other slots are left alone.
extern "C" int lgen_sample_rows(const void* logits, const float* noise, int* cur_tok, int* seq, int* row_step, int* row_pos,
                                int max_steps, int B, int V, int seq_stride, int use_cfg, float cfg_scale, int cfg_interval,
                                float temperature, int top_k, float top_p, int greedy, int dtype, void* stream) {
    if (!row_step || !row_pos || max_steps < 1) return LGEN_ERR_BAD_ARG;
    return sample_impl(logits, noise, 0, cur_tok, seq, nullptr, B, V, seq_stride, use_cfg, cfg_scale, cfg_interval, temperature,
                       top_k, top_p, greedy, dtype, row_step, row_pos, max_steps, stream);
}

extern "C" int lgen_advance_state(int* state, void* stream) {
    hipLaunchKernelGGL(advance_state_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    LGEN_CHECK_LAUNCH();
    return 0;
}
