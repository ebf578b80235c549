// VQ-VAE tokenizer kernels that are not matmul-shaped (tokenizer/tokenizer_image/vq_model.py):
//   lgen_vq_codebook_prep   F.normalize(codebook) + row ||e||^2            (vq_model.py:221-224, 263-264)
//   lgen_vq_lookup_pqconv   get_codebook_entry gather + post_quant_conv 1x1 (vq_model.py:261-276, 47-48)
//   lgen_vq_argmin          VectorQuantizer.forward nearest-entry search    (vq_model.py:215-232)
//   lgen_gn_stats           GroupNorm(32, C, eps=1e-6) statistics           (vq_model.py:359-362)
//   lgen_gn_swish_split     GroupNorm-apply [+ swish] + hi/lo bf16 split    (vq_model.py:299-306, 354-356)
//   lgen_softmax_split      AttnBlock softmax + hi/lo split                  (vq_model.py:341)
//   lgen_split_t            plain hi/lo split with [HW][C] -> [C][HW] transpose (AttnBlock v operand)
// Activations are NHWC fp32.  "hi/lo split": x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the
// convolutions run 3 bf16 MFMA passes (hi*hi + hi*lo + lo*hi) with fp32 accumulation, which holds
// the decoder to ~1e-4 of the fp32 reference (single-pass bf16 does not: SURVEY.md section 7).
#include "lgen_common.h"
#include "../../include/lgen.h"

LGEN_DEV uint16_t bf_hi(float x) { return f2bf(x); }
LGEN_DEV void split2(float x, uint16_t& hi, uint16_t& lo) {
    hi = f2bf(x);
    lo = f2bf(x - bf2f(hi));
}

// ---------------------------------------------------------------------------------------------
__global__ void codebook_prep_kernel(const float* __restrict__ cb, float* __restrict__ cbn, float* __restrict__ esq,
                                     int n_e, int dim, int l2norm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_e) return;
    float ss = 0.f;
    for (int j = 0; j < dim; ++j) ss += cb[(size_t)e * dim + j] * cb[(size_t)e * dim + j];
    const float den = l2norm ? fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
    float s2 = 0.f;
    for (int j = 0; j < dim; ++j) {
        const float v = cb[(size_t)e * dim + j] / den;
        cbn[(size_t)e * dim + j] = v;
        s2 += v * v;
    }
    esq[e] = s2;
}

extern "C" int lgen_vq_codebook_prep(const float* codebook, float* cb_norm, float* e_sq, int n_e, int dim, int l2norm,
                                     void* stream) {
    hipLaunchKernelGGL(codebook_prep_kernel, dim3((n_e + 255) / 256), dim3(256), 0, (hipStream_t)stream, codebook, cb_norm,
                       e_sq, n_e, dim, l2norm);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// gather + 1x1 conv (dim -> Cout): thread = output channel, block walks PIX pixels
#define PQ_PIX 32
__global__ __launch_bounds__(512) void lookup_pqconv_kernel(const float* __restrict__ cbn, const long long* __restrict__ idx,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int npix, int n_e, int cout) {
    const int c = threadIdx.x;
    float wr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wr[j] = c < cout ? w[(size_t)c * 8 + j] : 0.f;
    const float bc = c < cout ? bias[c] : 0.f;
    const int p0 = blockIdx.x * PQ_PIX;
    for (int p = p0; p < p0 + PQ_PIX && p < npix; ++p) {
        long long e = idx[p];
        e = e < 0 ? 0 : (e >= n_e ? n_e - 1 : e);
        const float4 a = *(const float4*)(cbn + (size_t)e * 8);
        const float4 b = *(const float4*)(cbn + (size_t)e * 8 + 4);
        float acc = 0.f;
        acc = fmaf(wr[0], a.x, acc); acc = fmaf(wr[1], a.y, acc); acc = fmaf(wr[2], a.z, acc); acc = fmaf(wr[3], a.w, acc);
        acc = fmaf(wr[4], b.x, acc); acc = fmaf(wr[5], b.y, acc); acc = fmaf(wr[6], b.z, acc); acc = fmaf(wr[7], b.w, acc);
        if (c < cout) out[(size_t)p * cout + c] = acc + bc;
    }
}

extern "C" int lgen_vq_lookup_pqconv(const float* cb_norm, const long long* indices, const float* w, const float* bias,
                                     float* out_nhwc, int npix, int n_e, int dim, int cout, void* stream) {
    if (dim != 8 || cout > 512 || cout < 1) return LGEN_ERR_UNSUPPORTED;
    if (npix == 0) return 0;
    const int threads = (cout + 63) / 64 * 64;
    hipLaunchKernelGGL(lookup_pqconv_kernel, dim3((npix + PQ_PIX - 1) / PQ_PIX), dim3(threads), 0, (hipStream_t)stream, cb_norm,
                       indices, w, bias, out_nhwc, npix, n_e, cout);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Codebook argmin: thread = one latent vector (8 floats in registers), the normalised codebook
// streams through LDS in 2048-entry tiles (broadcast reads), distance d = |z|^2 + |e|^2 - 2 z.e as
// the reference writes it, running (min, first index).  Never materialises the [N, 16384] matrix.
// ---------------------------------------------------------------------------------------------
#define AM_TILE 2048
__global__ __launch_bounds__(256) void argmin_kernel(const float* __restrict__ z, const float* __restrict__ cbn,
                                                     const float* __restrict__ esq, long long* __restrict__ out, int nvec,
                                                     int hw, int n_e, int l2norm) {
    __shared__ float4 s_e[AM_TILE * 2];
    __shared__ float s_q[AM_TILE];
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float zv[8];
    float zsq = 0.f;
    const bool live = v < nvec;
    if (live) {  // z is NCHW: element (b, c, p) at (b*8 + c)*hw + p
        const int b = v / hw, p = v - b * hw;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { zv[j] = z[((size_t)b * 8 + j) * hw + p]; ss += zv[j] * zv[j]; }
        if (l2norm) {
            const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
            for (int j = 0; j < 8; ++j) zv[j] = zv[j] / den;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) zsq += zv[j] * zv[j];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[j] = 0.f;
    }
    float best = INFINITY;
    int bidx = 0;
    for (int e0 = 0; e0 < n_e; e0 += AM_TILE) {
        const int cnt = min(AM_TILE, n_e - e0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 2; i += blockDim.x) s_e[i] = ((const float4*)cbn)[(size_t)e0 * 2 + i];
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) s_q[i] = esq[e0 + i];
        __syncthreads();
        for (int i = 0; i < cnt; ++i) {
            const float4 a = s_e[2 * i], b = s_e[2 * i + 1];
            float dot = 0.f;
            dot = fmaf(zv[0], a.x, dot); dot = fmaf(zv[1], a.y, dot); dot = fmaf(zv[2], a.z, dot); dot = fmaf(zv[3], a.w, dot);
            dot = fmaf(zv[4], b.x, dot); dot = fmaf(zv[5], b.y, dot); dot = fmaf(zv[6], b.z, dot); dot = fmaf(zv[7], b.w, dot);
            const float d = (zsq + s_q[i]) - 2.0f * dot;
            if (d < best) { best = d; bidx = e0 + i; }
        }
    }
    if (live) out[v] = bidx;
}

extern "C" int lgen_vq_argmin(const float* z_nchw, const float* cb_norm, const float* e_sq, long long* indices, int nvec,
                              int hw, int n_e, int dim, int l2norm, void* stream) {
    if (dim != 8) return LGEN_ERR_UNSUPPORTED;
    if (nvec == 0) return 0;
    hipLaunchKernelGGL(argmin_kernel, dim3((nvec + 255) / 256), dim3(256), 0, (hipStream_t)stream, z_nchw, cb_norm, e_sq,
                       indices, nvec, hw, n_e, l2norm);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics on NHWC fp32: stage 1 = per-(image, pixel-chunk) partial (sum, sumsq) per
// group, fixed-order reductions; stage 2 = fp64 combine -> (mean, rstd) per (image, group).
// ---------------------------------------------------------------------------------------------
#define GN_THREADS 256
static int g_vq_nt = 0;
extern "C" int lgen_debug_set_vq_nt(int v) { g_vq_nt = v ? 1 : 0; return 0; }
int lgen_vq_nt() { return g_vq_nt; }

__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part, int hw,
                                                                int C, int ppc, int nt) {
    __shared__ float s_s[GN_THREADS], s_q[GN_THREADS];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int cq = C >> 2;                       // float4 columns per pixel
    const int t = threadIdx.x;
    const int col = t % cq, prow = t / cq, rows = GN_THREADS / cq;
    const int p0 = chunk * ppc, p1 = min(hw, p0 + ppc);
    float s = 0.f, q = 0.f;
    if (prow < rows) {
        for (int p = p0 + prow; p < p1; p += rows) {
            const float4* vp = (const float4*)x + ((size_t)b * hw + p) * cq + col;
            const float4 v = nt ? ldg_nt_f4(vp) : *vp;
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    }
    s_s[t] = s; s_q[t] = q;
    __syncthreads();
    if (t < 32) {  // group g owns float4 columns [g*gq, (g+1)*gq), gq = C/128
        const int gq = cq >> 5;
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int c = t * gq; c < (t + 1) * gq; ++c) { S += (double)s_s[r * cq + c]; Q += (double)s_q[r * cq + c]; }
        part[(((size_t)b * nchunk + chunk) * 32 + t) * 2 + 0] = S;
        part[(((size_t)b * nchunk + chunk) * 32 + t) * 2 + 1] = Q;
    }
}

__global__ void gn_final_kernel(const double* __restrict__ part, float* __restrict__ stats, int nchunk, double count, float eps) {
    const int b = blockIdx.x, g = threadIdx.x;
    double S = 0.0, Q = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        S += part[(((size_t)b * nchunk + c) * 32 + g) * 2 + 0];
        Q += part[(((size_t)b * nchunk + c) * 32 + g) * 2 + 1];
    }
    const double mean = S / count;
    double var = Q / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[((size_t)b * 32 + g) * 2 + 0] = (float)mean;
    stats[((size_t)b * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

extern "C" int lgen_gn_stats(const float* x_nhwc, double* partial_ws, float* stats, int B, int hw, int C, float eps,
                             int nchunk, void* stream) {
    if (C % 128 || C > 1024 || nchunk < 1) return LGEN_ERR_BAD_ARG;  // 32 groups of >= 4 channels, float4 aligned
    if (B == 0) return 0;
    const int ppc = (hw + nchunk - 1) / nchunk;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(GN_THREADS), 0, st, x_nhwc, partial_ws, hw, C, ppc, g_vq_nt);
    LGEN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_final_kernel, dim3(B), dim3(32), 0, st, partial_ws, stats, nchunk, (double)hw * (C / 32), eps);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// y = GN(x) [* sigmoid] -> (hi, lo) bf16 planes, same NHWC indexing.  8 channels per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_swish_split_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             uint4* __restrict__ hi, uint4* __restrict__ lo, size_t n8, int hw,
                                                             int C, int mode, int nt) {
    const int c8 = C >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % c8);
        const size_t pix = i / c8;
        const int b = (int)(pix / hw);
        const int c0 = col * 8;
        const float4* xp4 = (const float4*)x + i * 2;
        const float4 v0 = nt ? ldg_nt_f4(xp4) : xp4[0], v1 = nt ? ldg_nt_f4(xp4 + 1) : xp4[1];
        float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (mode & 1) {
            const int gs = C >> 5;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e;
                const float mean = stats[((size_t)b * 32 + c / gs) * 2], rstd = stats[((size_t)b * 32 + c / gs) * 2 + 1];
                const float a = rstd * gamma[c];
                f[e] = fmaf(f[e], a, beta[c] - a * mean);
            }
        }
        if (mode & 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] / (1.0f + expf(-f[e]));
        }
        uint16_t h[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2(f[e], h[e], l[e]);
        hi[i] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        lo[i] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    }
}

extern "C" int lgen_gn_swish_split(const float* x_nhwc, const float* stats, const float* gamma, const float* beta, void* hi,
                                   void* lo, int B, int hw, int C, int mode, void* stream) {
    if (C % 8 || ((mode & 1) && C % 32)) return LGEN_ERR_BAD_ARG;
    const size_t n8 = (size_t)B * hw * (C / 8);
    if (n8 == 0) return 0;
    const int blocks = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipLaunchKernelGGL(gn_swish_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x_nhwc, stats, gamma, beta,
                       (uint4*)hi, (uint4*)lo, n8, hw, C, mode, g_vq_nt);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// plain split with transpose: x [B][R][C] fp32 -> hi/lo [B][C][R] bf16 (R padded to ldr in the output)
__global__ void split_t_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int R, int C,
                               int ldr) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? x[((size_t)b * R + r) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < C && r < ldr) {
            uint16_t h, l;
            split2(r < R ? tile[threadIdx.x][j] : 0.f, h, l);
            hi[((size_t)b * C + c) * ldr + r] = h;
            lo[((size_t)b * C + c) * ldr + r] = l;
        }
    }
}

extern "C" int lgen_split_t(const float* x, void* hi, void* lo, int B, int R, int C, int ldr, void* stream) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(split_t_kernel, dim3((C + 31) / 32, (ldr + 31) / 32, B), dim3(32, 8), 0, (hipStream_t)stream, x,
                       (uint16_t*)hi, (uint16_t*)lo, R, C, ldr);
    LGEN_CHECK_LAUNCH();
    return 0;
}

// row softmax (fp32) + hi/lo split; one workgroup per row, row length n, output row stride ldo (zero padded)
__global__ __launch_bounds__(256) void softmax_split_kernel(const float* __restrict__ s, uint16_t* __restrict__ hi,
                                                            uint16_t* __restrict__ lo, int n, int ldo) {
    __shared__ float red[4];
    __shared__ float bc;
    const size_t row = blockIdx.x;
    const float* sr = s + row * n;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    float m = -INFINITY;
    for (int i = t; i < n; i += 256) m = fmaxf(m, sr[i]);
    m = wave_max(m);
    if (lane == 0) red[wv] = m;
    __syncthreads();
    if (t == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bc;
    float sum = 0.f;
    for (int i = t; i < n; i += 256) sum += expf(sr[i] - m);
    sum = wave_sum(sum);
    __syncthreads();
    if (lane == 0) red[wv] = sum;
    __syncthreads();
    if (t == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    sum = bc;
    for (int i = t; i < ldo; i += 256) {
        uint16_t h = 0, l = 0;
        if (i < n) split2(expf(sr[i] - m) / sum, h, l);
        hi[row * ldo + i] = h;
        lo[row * ldo + i] = l;
    }
}

extern "C" int lgen_softmax_split(const float* scores, void* hi, void* lo, int rows, int n, int ldo, void* stream) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(softmax_split_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, scores, (uint16_t*)hi,
                       (uint16_t*)lo, n, ldo);
    LGEN_CHECK_LAUNCH();
    return 0;
}
