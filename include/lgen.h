/* lgen.h -- C ABI of the MI355X-native LlamaGen sampling library (liblgen_hip.so).
 *
 * The reference (FoundationVision/LlamaGen) has no native/FFI layer on this path: the boundary a
 * replacement sits behind is its Python API.  This header is the C ABI
 * underneath our Python host mirror (llamagen_amd/): `extern "C"`, raw device pointers + sizes,
 * a `hipStream_t` passed as void*, no torch types, no allocation, no synchronisation; every entry
 * point only enqueues kernels on the given stream (hipGraph-capture safe) and returns 0 or a
 * hipError_t / LGEN_ERR_* code.  ABI v7: every schedule / kernel-variant choice of the product path is an explicit ARGUMENT of the
 * call it applies to (workgroup shape, `passes`, attention `variant`); the one-shot per-thread hints of v4-v6 are gone.  What is left
 * inside the library: (1) one-shot `hipFuncSetAttribute` flags for the kernels that need more than 64 KiB of LDS (set on first use
 * for the current device) and the cached CU count of the persistent attention form; (2) the development switches of the
 * `lgen_debug_*` section at the end of this header (process-wide ints read at launch time, never set by the product path) and the
 * LGEN_GEMM_STEADY / LGEN_TILE_ABLATE environment variables (read at launch = capture time; bit-identical results / timing
 * ablations).  ABI v8 = v7 + a K/V row stride that may be smaller than the lane group hdp (lgen_gemm_qkv_rope); ABI v9 = v8 minus
 * lgen_conv_wino (the Winograd experiment of round 5 left the product library: tools/experiments/conv_wino/); ABI v10 = v9 +
 * lgen_ssq_group4 (row statistics pre-grouped for the fused-norm consumers) and key-row stride checks in the prefill entry points.  Each entry point
 * cites the reference op sequence it replaces (paths relative to the reference repository root).
 *
 * Fragment-packed layouts ("chunk" = 1 KiB = [16 rows][KC k] in MFMA operand order, lane =
 * g*16 + r holds row r, k-slice g*EPL..+EPL; bf16: KC=32, EPL=8; fp32: KC=16, EPL=4):
 *   weights      WP[N/16][K/KC][64][EPL]
 *   activations  XP[K/KC][MTs][64][EPL]      (row m lives in m-tile m/16, lane-row m%16)
 */
#ifndef LGEN_H
#define LGEN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LGEN_ABI_VERSION 10
#define LGEN_BF16 0
#define LGEN_F32 1
#define LGEN_F16 2   /* fp16 storage: BF16's layouts (KC = 32, EPL = 8), IEEE half rounding, v_mfma_f32_16x16x32_f16 */

/* Row statistics of the fused RMSNorm: ssq arrays are [MTs*16 rows][LGEN_SSQ_STRIDE] fp32; row m holds `parts` partial sums of
 * squares (one per 16 columns of the producer: parts = d/16), contiguous, so a consumer reads them as a few 16-byte loads. */
#define LGEN_SSQ_STRIDE 256

#define LGEN_ERR_BAD_ARG (-1)
#define LGEN_ERR_UNSUPPORTED (-2)

/* epilogues of lgen_gemm */
#define LGEN_EPI_ROWS 0   /* out[M][N] row-major, storage dtype (lm_head: gpt.py:367-368)            */
#define LGEN_EPI_PACKED 1 /* out = XP of width N                                                    */
#define LGEN_EPI_GELU 2   /* out = XP of gelu_tanh(.) (CaptionEmbedder MLP fc1, gpt.py:118-131)      */
#define LGEN_EPI_RES 3    /* out (XP, in/out) += result: wo / w2 + residual (gpt.py:238-240,255-256) */
#define LGEN_EPI_SWIGLU 4 /* W = row-tile-interleaved w1||w3, out = XP of silu(w1x)*w3x (gpt.py:167) */
#define LGEN_EPI_QKV 5    /* lgen_gemm_qkv_rope only (lgen_gemm_max_kw query)                         */

int lgen_abi_version(void);

/* ---- GPT decode step ------------------------------------------------------------------- */

/* nn.Embedding gather (gpt.py:78-83 LabelEmbedder / :351 tok_embeddings) -> packed residual stream.
 * table [rows][d] storage dtype, idx int32 [M] (device), hp = XP[d/KC][MTs]; ssq_out (nullable):
 * [MTs*16][LGEN_SSQ_STRIDE] fp32, d/16 partial row sums of squares per row, for the first fused RMSNorm; state_advance (nullable):
 * device {pos, step}, both incremented before the rest of the decode step reads them (the
 * `input_pos += 1` of generate.py:118). */
int lgen_embed_pack(const void* table, const int* idx, void* hp, float* ssq_out, int* state_advance, int M, int MTs,
                    int d, int rows, int dtype, void* stream);

/* Row sums of squares of an already packed residual stream hp = XP[d/KC][MTs] (t2i prefix rows produced
 * by the CaptionEmbedder MLP): ssq_out [MTs*16][LGEN_SSQ_STRIDE] fp32 (d/16 partials per row), the ssq_in of a fused RMSNorm. */
int lgen_ssq_pack(const void* hp, float* ssq_out, int MTs, int d, int dtype, void* stream);

/* Round 6 (ABI v10): the four lane-group sums of every statistics row -- ssq_out[row][0..3] = sum of ssq_in[row][q], q = g, g + 4, ...
 * ascending, g = 0..3 -- i.e. what every fused-norm consumer computes first from a `ssq_parts`-partial row.  A consumer given
 * ssq_out with ssq_parts = 4 produces the SAME bits as with ssq_in and the original count, and skips the per-workgroup reduction
 * (GPT-3B, 200 partials: 25-38 us of a 99-140 us launch).  rows = MTs * 16, parts % 4 == 0, ssq_out != ssq_in. */
int lgen_ssq_group4(const float* ssq_in, float* ssq_out, int rows, int parts, void* stream);

/* RMSNorm.forward (gpt.py:143-148) on XP -> XP, fp32 math, two storage roundings. */
int lgen_rmsnorm(const void* hp, const void* weight, void* xnp, int MTs, int d, float eps, int dtype, void* stream);

/* Bias-free nn.Linear (gpt.py:161-163,199-200,287) out = x . W^T with fused epilogue, fp32
 * accumulate on MFMA, one storage rounding of the linear output.  (mt, nt, kw) = tile shape:
 * m-tiles per workgroup (divides MTs), n-tiles per workgroup, waves splitting K.
 * RMSNorm fusion (gpt.py:137-148, the norm that precedes wqkv / w1,w3 / output in gpt.py:253-256,367):
 *   norm_w != NULL (ROWS, SWIGLU): x is the un-normalised residual stream and the kernel applies
 *     rnd(rnd(x * rsqrt(mean(x^2) + eps)) * norm_w) to the operand on the fly; mean(x^2) comes from
 *     ssq_in[MTs*16][LGEN_SSQ_STRIDE] fp32: ssq_parts partial row sums of squares per row (summed in a fixed order);
 *   ssq_out != NULL (RES only): also writes ssq_out[row][N/16 partials], the per-(row, 16-column tile) sums of
 *     squares of the updated residual stream, i.e. the ssq_in (ssq_parts = N/16) of the next norm.
 *   passes (1..64): fused-norm form only (csrc/gemm_normpre.hip) -- consecutive n-groups one workgroup walks with its normalised
 *     rows kept in registers; any value gives bit-identical results (a scheduling choice); 1 elsewhere. */
int lgen_gemm(const void* wp, const void* xp, void* out, int M, int MTs, int N, int K, int epilogue_kind, int dtype,
              int mt, int nt, int kw, const void* norm_w, const float* ssq_in, int ssq_parts, float eps,
              float* ssq_out, int passes, void* stream);

/* Largest kw (K-splitting waves per workgroup) the (mt, nt) tile shape of that kernel variant admits. */
int lgen_gemm_max_kw(int epilogue_kind, int fused_norm, int mt, int nt);

/* Attention.forward front half (gpt.py:214-226): [attention_norm, gpt.py:254 +] wqkv GEMM +
 * apply_rotary_emb(q,k) (gpt.py:420-430) + KVCache.update at *pos_ptr (gpt.py:177-185).
 * q_out [MTs*16][H][hdp]; caches [B2][H][S8] rows of hd valid elements, kv_row_stride elements apart (0 = hdp;
 * 2*hdp with v_cache = k_cache + hdp is an interleaved K|V slab: one HBM stream per (b, h)).  Round 5: kv_row_stride may be
 * SMALLER than the lane group hdp, down to hd rounded up to 8 elements (GPT-3B: hd 100 -> rows 104 elements = 13 x 16 B apart
 * instead of 128): the attention kernels still read hdp elements per key -- the lanes past the row read the first bytes of the
 * NEXT row, which multiply q's zero pad lanes (QK^T) or land in output elements >= hd that are never stored (PV) -- so every
 * cache must be followed by >= (hdp - kv_row_stride) readable, FINITE elements (the engine allocates that slack), every cache
 * element a packed row's neighbours can be must be finite too (an Inf / NaN there times a zero pad lane is NaN: with fp16 storage
 * keep appended keys finite), and the q rows given to lgen_attn_decode / lgen_attn_prefill must hold EXACT ZEROS in elements
 * [hd, hdp) (the QKV epilogues of this library write them so; a caller that fills q itself must too).  kv_row_stride must be a
 * multiple of the 16-byte piece (8 elements, 4 for fp32) and >= hd rounded up to it: every entry point that takes it returns
 * LGEN_ERR_BAD_ARG otherwise (round 6: also lgen_rope_append_prefill / lgen_attn_prefill);
 * freqs [P][hd/2][2] fp32 from precompute_freqs_cis_2d (gpt.py:404-417); norm_w / ssq_in / ssq_parts / eps as
 * in lgen_gemm; passes as in lgen_gemm. */
int lgen_gemm_qkv_rope(const void* wp, const void* xp, void* q_out, void* k_cache, void* v_cache, const float* freqs,
                       const int* pos_ptr, int M, int MTs, int d, int n_head, int hd, int hdp, int S8,
                       int kv_row_stride, int dtype, int mt, int nt, int kw, const void* norm_w, const float* ssq_in,
                       int ssq_parts, float eps, int passes, void* stream);

/* The same two operations for wide chains (>= 128 rows), bf16 storage: the big-M tile family (csrc/gemm_tile.hip).  A workgroup of
 * wm x wn waves owns (wm*mtv*16 rows) x (wn*ntv*16 columns) of the output over the whole K range -- waves split the tile, never K --
 * and both operands stream through an LDS ring of `stages` slots of `kb` k-chunks (global_load_lds, counted waits; lw = 4: issued
 * by four extra loader waves, one per SIMD, so that DMA issue overlaps the MFMA work; lw = 0: by the computing waves): every weight
 * byte enters a CU once per (row block), no cross-wave reduction.  Same operands, epilogues, RMSNorm fusion and summation-order
 * contract for the statistics as lgen_gemm / lgen_gemm_qkv_rope (gpt.py:161-167,199-226,238-240,253-257,367-368); the accumulation
 * order over K differs from the skinny kernels (one wave walks all of K), so results agree with them to fp32-accumulation
 * round-off, not bit for bit.  LGEN_ERR_UNSUPPORTED: no instantiation of that shape / shape does not divide (MTs % (wm*mtv),
 * (K/32) % kb) / LDS budget exceeded: the caller picks another shape or the skinny kernel.  EPI_RES, and with norm_w EPI_ROWS /
 * EPI_SWIGLU (lgen_gemm_tile) and the fused-norm qkv form (lgen_gemm_qkv_rope_tile; one position for all rows). */
int lgen_gemm_tile(const void* wp, const void* xp, void* out, int M, int MTs, int N, int K, int epilogue_kind, int dtype,
                   int wm, int wn, int mtv, int ntv, int kb, int stages, int lw, const void* norm_w, const float* ssq_in,
                   int ssq_parts, float eps, float* ssq_out, void* stream);
int lgen_gemm_qkv_rope_tile(const void* wp, const void* xp, void* q_out, void* k_cache, void* v_cache, const float* freqs,
                            const int* pos_ptr, int M, int MTs, int d, int n_head, int hd, int hdp, int S8,
                            int kv_row_stride, int dtype, int wm, int wn, int mtv, int ntv, int kb, int stages, int lw,
                            const void* norm_w, const float* ssq_in, int ssq_parts, float eps, void* stream);

/* Attention.forward back half (gpt.py:229-236): repeat_interleave + math-backend SDPA with
 * causal_mask[:, pos] -- here: single-query attention over the first *pos_ptr+1 cache slots.
 * mask: null = pure causal, else the reference's causal_mask [B2][S8][S8] (1 byte per entry, as
 * modified by generate.py:154-163 for t2i emb_masks); row *pos_ptr of it gates the keys < mask_len (0 = all
 * S8; generate.py only ever clears columns of the T-token prefix, so mask_len = T skips the byte loads for
 * the image tokens).  variant: -1 = the library's choice by shape (one head per 128-thread workgroup below 256 rows; from 256
 * rows the PERSISTENT form: one 4-wave workgroup per CU whose waves walk whole (row, head) items with 16 KiB of K/V loads in
 * flight each -- 64 KB per CU instead of 160, so that the HBM queue stays short for the other chains' kernels), else 0..13
 * (csrc/gpt_ops.hip lists them; every variant computes the same wave-level online softmax). */
int lgen_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out_packed, const int* pos_ptr,
                     const unsigned char* mask, int mask_len, int B2, int MTs, int n_head, int hd, int hdp, int S8,
                     int kv_row_stride, int dtype, int variant, void* stream);

/* ---- sequence prefill (t2i prefix: all T = cls_token_num caption positions of all B2 rows per layer at once; rows
 * r = t * B2 + b of the packed activations; generate.py:77-86 + gpt.py:348-349 with emb_masks folded into
 * causal_mask, generate.py:154-163) ---- */

/* packed wqkv output (lgen_gemm EPI_PACKED, width 3d, R = B2*T rows) -> apply_rotary_emb(q, k) at position
 * pos0 + t (gpt.py:220-226, 420-430); q rows [R][H][hdp]; K/V into slot pos0 + t of cache row b (gpt.py:177-185). */
int lgen_rope_append_prefill(const void* qkv_packed, void* q_rows, void* k_cache, void* v_cache, const float* freqs, int R,
                             int B2, int MTs, int d, int n_head, int hd, int hdp, int S8, int kv_row_stride, int pos0,
                             int dtype, void* stream);

/* masked causal attention of the prefix onto itself (gpt.py:229-236, math-SDPA semantics): query row (b, t)
 * sees keys s <= t with mask[b][t][s] != 0 (mask = causal_mask [B2][S8][S8] bytes or null); out = XP of width d.
 * Any 1 <= T <= S8: up to 128 positions keep K/V of one (b, h) whole in LDS; longer sequences (the whole-sequence
 * `is_causal` forward of gpt.py:232-236, 341-346) take a key-tiled online-softmax kernel. */
int lgen_attn_prefill(const void* q_rows, const void* k_cache, const void* v_cache, void* out_packed,
                      const unsigned char* mask, int T, int B2, int MTs, int n_head, int hd, int hdp, int S8,
                      int kv_row_stride, int dtype, void* stream);

/* generate.py:79-86,94-99 (CFG mix) + :57-66 sample() + :16-54 top_k_top_p_filtering +
 * torch.multinomial(1) == argmax(p / noise).  logits [>=2B][V] storage dtype (rows [0,B) cond,
 * [B,2B) uncond when use_cfg); noise fp32 Exp(1) draws: row b of this call is at
 * noise[(step * noise_step_stride) + b*V] with step = state[1] (stride 0: one [B][V] block reused);
 * state = {pos, step} device ints (read only).  Writes cur_tok[b] (and cur_tok[B+b]), seq[b][step]. */
int lgen_sample(const void* logits, const float* noise, long long noise_step_stride, int* cur_tok, int* seq,
                const int* state, int B, int V, int seq_stride, int use_cfg, float cfg_scale, int cfg_interval,
                float temperature, int top_k, float top_p, int greedy, int dtype, void* stream);

int lgen_advance_state(int* state, void* stream);

/* ---- continuous batching (autoregressive/serve/: the vLLM fork's token-level scheduling -- every row of the step batch is its
 * own request at its own position; serve/sampler.py:54-58,106-108 pairs a conditional and an unconditional sequence per
 * request).  Same kernels as above with PER-ROW positions: row_pos[m] instead of one device scalar. ---- */

/* lgen_embed_pack with a per-row source: row m at position 0 is a fresh request and takes cls_table[cond[m]] (its prefill
 * token, gpt.py:348-349), any other row takes tok_table[cur_tok[m]] (gpt.py:351). */
int lgen_embed_rows(const void* tok_table, const void* cls_table, const int* cur_tok, const int* cond, const int* row_pos,
                    void* hp, float* ssq_out, int M, int MTs, int d, int tok_rows, int cls_rows, int dtype, void* stream);

/* lgen_gemm_qkv_rope with RoPE angles and the KV-cache slot of row m taken at row_pos[m] ([MTs*16] device ints). */
int lgen_gemm_qkv_rope_rows(const void* wp, const void* xp, void* q_out, void* k_cache, void* v_cache, const float* freqs,
                            const int* row_pos, int M, int MTs, int d, int n_head, int hd, int hdp, int S8,
                            int kv_row_stride, int dtype, int mt, int nt, int kw, const void* norm_w, const float* ssq_in,
                            int ssq_parts, float eps, int passes, void* stream);

/* lgen_attn_decode with kv_len of row b = row_pos[b] + 1. */
int lgen_attn_decode_rows(const void* q, const void* k_cache, const void* v_cache, void* out_packed, const int* row_pos,
                          const unsigned char* mask, int mask_len, int B2, int MTs, int n_head, int hd, int hdp, int S8,
                          int kv_row_stride, int dtype, int variant, void* stream);

/* lgen_sample per slot: slot b is at step row_step[b] of max_steps (>= max_steps: empty / finished, skipped); its Exp(1) draws
 * are noise[(b*max_steps + step)*V ..]; writes seq[b][step], cur_tok[b] (and cur_tok[B+b]), then advances row_step[b] and
 * row_pos[b] (and row_pos[B+b]). */
int lgen_sample_rows(const void* logits, const float* noise, int* cur_tok, int* seq, int* row_step, int* row_pos, int max_steps,
                     int B, int V, int seq_stride, int use_cfg, float cfg_scale, int cfg_interval, float temperature, int top_k,
                     float top_p, int greedy, int dtype, void* stream);

/* ---- VQ-VAE tokenizer (tokenizer/tokenizer_image/vq_model.py), fp32 NHWC activations ------------- */

/* F.normalize(embedding.weight) (vq_model.py:223,264) + |e|^2 per row; cb_norm [n_e][dim], e_sq [n_e]. */
int lgen_vq_codebook_prep(const float* codebook, float* cb_norm, float* e_sq, int n_e, int dim, int l2norm,
                          void* stream);

/* VectorQuantizer.get_codebook_entry (vq_model.py:261-276) + post_quant_conv 1x1 (vq_model.py:48):
 * indices int64 [npix] -> out NHWC [npix][cout]; w [cout][dim], bias [cout]. */
int lgen_vq_lookup_pqconv(const float* cb_norm, const long long* indices, const float* w, const float* bias,
                          float* out_nhwc, int npix, int n_e, int dim, int cout, void* stream);

/* VectorQuantizer.forward eval path (vq_model.py:215-232): z NCHW [B][dim][hw] -> int64 [B*hw] nearest
 * entry (first index wins ties); never materialises the [N][n_e] distance matrix. */
int lgen_vq_argmin(const float* z_nchw, const float* cb_norm, const float* e_sq, long long* indices, int nvec, int hw,
                   int n_e, int dim, int l2norm, void* stream);

/* nn.GroupNorm(32, C, eps) statistics (vq_model.py:359-362): stats [B][32][2] = (mean, rstd);
 * partial_ws: B*nchunk*64 doubles. */
int lgen_gn_stats(const float* x_nhwc, double* partial_ws, float* stats, int B, int hw, int C, float eps, int nchunk,
                  void* stream);

/* mode bit0: apply GroupNorm (stats/gamma/beta), bit1: swish x*sigmoid(x) (vq_model.py:354-356);
 * writes x as (hi, lo) bf16 planes with the same NHWC indexing. */
int lgen_gn_swish_split(const float* x_nhwc, const float* stats, const float* gamma, const float* beta, void* hi,
                        void* lo, int B, int hw, int C, int mode, void* stream);

/* x [B][R][C] fp32 -> (hi, lo) bf16 [B][C][ldr] (transposed, zero padded to ldr >= R). */
int lgen_split_t(const float* x, void* hi, void* lo, int B, int R, int C, int ldr, void* stream);

/* F.softmax(dim=-1) of AttnBlock (vq_model.py:341) on [rows][n] fp32 -> (hi, lo) bf16 [rows][ldo]. */
int lgen_softmax_split(const float* scores, void* hi, void* lo, int rows, int n, int ldo, void* stream);

/* nn.Conv2d 3x3 (pad 1) / 1x1, stride 1 (vq_model.py:288-291,321-324) as implicit GEMM on MFMA with
 * hi/lo-split bf16 operands (3 passes, fp32 accumulate); `upsample` = 1: nearest-2x upsample of the input
 * (Upsample, vq_model.py:374-378: a_* is then [B][H/2][W/2][Cin]); `upsample` = 2: stride-2 Downsample conv
 * (vq_model.py:389-393: a_* is [B][2H][2W][Cin], zero pad right/bottom); bias, residual add (NHWC like out),
 * NCHW output, scale alpha.  With ksize 1 and w_bstride != 0 it is a batched NT GEMM (torch.bmm of
 * AttnBlock, vq_model.py:337,346).  w_* planes: [taps][Npad][Cin] bf16. */
int lgen_conv_igemm(const void* a_hi, const void* a_lo, const void* w_hi, const void* w_lo, const float* bias,
                    const float* res, float* out, int B, int H, int W, int Cin, int Cout, int Npad, int ksize,
                    int upsample, int out_nchw, long long w_bstride, float alpha, void* stream);

/* Fused decoder convolution (vq_model.py:299-314 ResnetBlock body, :374-378 Upsample, :190-192 norm_out/swish/conv_out):
 * out = conv_{ksize x ksize, pad ksize/2}( swish?( x * scale_c + shift_c ) ) + bias (+ res), with the GroupNorm-apply
 * (gn_coef [B][Cin][2] = (scale, shift) from lgen_gn_finalize, or NULL), the activation and the hi/lo bf16 split done
 * on the tile load (x is read once, as fp32 NHWC [B][H>>upsample][W>>upsample][Cin]), the 3x3 taps served from an
 * LDS-resident halo tile, and -- stats_partial != NULL -- the next GroupNorm's per-(8x16 tile, 4-channel quad)
 * (sum, M2) of the stored output written to stats_partial [B][(H/8)*ceil(W/16)][Npad/4][2].  upsample = 1 folds
 * F.interpolate(2.0, nearest).  w_frag: hi/lo bf16 weights in MFMA fragment order
 * [Npad/BN][Cin/32][ksize^2][2][BN/16][64 lanes][8], BN = lgen_conv_fused_bn(Cout).  Needs H % 8 == 0 and Cin % 32 == 0; W is
 * free (tiles are 8 x 16 pixels, the columns past W of the last tile column are computed and dropped: 24-wide maps of a 384 px
 * decode).  Other shapes: lgen_gn_stats + lgen_gn_swish_split + lgen_conv_igemm. */
int lgen_conv_fused_bn(int Cout);
int lgen_conv_fused(const float* x_nhwc, const float* gn_coef, int swish, const void* w_frag, const float* bias,
                    const float* res, float* out, float* stats_partial, int B, int H, int W, int Cin, int Cout, int Npad,
                    int ksize, int upsample, int out_nchw, void* stream);

/* GroupNorm(32, C, eps) statistics -> per-channel (scale, shift) = (rstd*gamma, beta - rstd*gamma*mean), coef [B][C][2].
 * Source: partial != NULL: the tile partials of lgen_conv_fused ([B][ntiles][quad_stride][2] = (sum, M2 about the partial's
 * own mean), combined in fp64 in a fixed order with the pairwise update of Chan et al.; hw = pixels per image, width = W of the
 * map the partials tile (ABI v6: a partial of the last tile column covers W - 16 * (tiles_x - 1) columns)); else stats
 * [B][32][2] = (mean, rstd) from lgen_gn_stats (width ignored). */
int lgen_gn_finalize(const float* partial, const float* stats, const float* gamma, const float* beta, float* coef, int B,
                     int C, int ntiles, int quad_stride, int hw, int width, float eps, void* stream);

/* ---- post-processing of decoded samples (autoregressive/sample/sample_c2i_ddp.py:141-143) ------------- */

/* F.interpolate(x, size=(Ho, Wo), mode='bicubic') (align_corners=False, A = -0.75): fp32 NCHW [BC][Hi][Wi] -> [BC][Ho][Wo]. */
int lgen_resize_bicubic(const float* in_nchw, float* out_nchw, int BC, int Hi, int Wi, int Ho, int Wo, void* stream);

/* torch.clamp(127.5 * x + 128.0, 0, 255).permute(0, 2, 3, 1).to(uint8): fp32 NCHW -> uint8 NHWC. */
int lgen_to_uint8_hwc(const float* in_nchw, unsigned char* out_nhwc, int B, int C, int H, int W, void* stream);

/* ---- lgen_debug_*: development switches (process-wide ints, read at launch time; the product path never calls them and every
 * default is the measured-best choice; kept for A/B measurements and for the parity tests that hold both forms to the oracle) ---- */
int lgen_debug_set_vq_nt(int v);              /* VQ decoder: non-temporal fp32 activation stores / GroupNorm-pass loads (default 0 = off) */
int lgen_debug_set_prefill_mfma(int v);       /* lgen_attn_prefill, bf16: 1 (default) MFMA flash kernel; 0 the VALU kernels (always used for fp32) */
int lgen_debug_set_conv_fused_variant(int v); /* lgen_conv_fused weight tiles: 3 (default since round 6) / 2: global -> LDS DMA + pipelined fragment reads in the 3x3, 128-channel-tile kernel (both / the hi pixel planes of the next tap requested under the current tap's MFMAs); 1 DMA, all fragment reads in front of a tap's MFMAs (rounds 3-5); 0 through staging registers -- every variant gives the same bits */
int lgen_debug_set_igemm_variant(int v);      /* 3 (default): 128x128 tile, 2 staging sets for pixels / 1 for weights; 0: 1 set; 2: 128x64 tiles; 1 is refused */

#ifdef __cplusplus
}
#endif
#endif /* LGEN_H */
