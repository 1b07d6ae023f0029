/* libtextflux_hip.so -- C ABI of the MI355X (gfx950) TextFlux / FLUX.1-Fill denoising engine.
 *
 * The reference (yyyyyxie/textflux @ 2025-09-26) has no FFI: its hot path sits behind Python object protocols of the
 * vendored diffusers 0.32.0.dev0.  Each entry point below names the reference call site(s) it
 * replaces; `D/` = diffusers/src/diffusers in the reference tree.  INTEGRATION.md shows the ctypes binding and how
 * the reference-side classes would call it.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; tfx_last_error() then describes it (thread local);
 *   - all tensor arguments are DEVICE pointers to bf16 (uint16 bit pattern) unless typed otherwise; nothing is
 *     allocated or freed by the library, no pointer is retained after a call returns;
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued asynchronously on it and is safe to
 *     capture into a hipGraph (no allocation, no synchronisation, no host read-back inside any entry point);
 *   - row-major matrices with explicit leading dimensions (`ld*`, in elements) and batch strides (`*_bstride`).
 */
#ifndef TEXTFLUX_HIP_H
#define TEXTFLUX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tfx_stream;

/* ---- library -------------------------------------------------------------------------------------------- */
const char* tfx_version(void);
const char* tfx_last_error(void);
/* ABI stamp.  TFX_ABI_VERSION is bumped whenever a struct below grows or the meaning of a field / entry point changes;
 * tfx_abi_info writes {TFX_ABI_VERSION the library was built with, sizeof(tfx_gemm_args), sizeof(tfx_attn_args),
 * sizeof(tfx_dit_desc), sizeof(tfx_step_desc), sizeof(tfx_double_block), sizeof(tfx_single_block)} (as many as fit in n; the last
 * two since ABI 6) and returns how many values there are.  A binding
 * compares them with its own view of this header BEFORE the first call that passes a struct: a library built from an older
 * header would otherwise ignore the tail fields of a grown struct silently (no reference counterpart: the reference has no FFI). */
#define TFX_ABI_VERSION 8
int tfx_abi_info(int32_t* out, int n);
/* Writes the gcnArchName of the current device (e.g. "gfx950:sramecc+:xnack-") into buf.  Needs a GPU. */
int tfx_query_arch(char* buf, int buflen);

/* ---- GEMM:  C[b] = epi(A[b] @ W^T + bias)  (every nn.Linear on the path: D/models/attention_processor.py:1990-1992,
 *      2009-2011, 2052, 2056; D/models/attention.py:1217-1232; transformer_flux.py:724, 732, 1086, 1099, 1203;
 *      D/models/normalization.py:168, 200, 364; D/models/embeddings.py:1008-1021, 1926-1931).
 *      A [batch][M,K] (lda, a_bstride), W [N,K] nn.Linear layout (ldw), bias [N] or NULL, C [batch][M,N].
 *      epilogue: 0 bias | 1 bias, then tanh-GELU on columns >= gelu_from_col (activations.py:83; the split form is the
 *      fused [k|v|q|mlp] projection of FluxSingleTransformerBlock) | 3 C = res + (A@W^T + bias) | 2 C = res + gate[b,:] * (A@W^T + bias)
 *      (gated residual, transformer_flux.py:733-735, 817-818, 824-826, 830-831, 837; res may alias C).
 *      variant: -1 auto, 0 generic FMA kernel (any shape), 1 MFMA path (K % 64 == 0, N % 8 == 0, 16-byte aligned;
 *      the persistent kernel when K % 128 == 0, else the one-tile kernel), 2 / 3 force the one-tile / persistent MFMA
 *      kernel (tests: the two agree bit for bit). */
typedef struct tfx_gemm_args {
  const void* A; int64_t lda; int64_t a_bstride;
  const void* W; int64_t ldw;
  const void* bias;
  void* C; int64_t ldc; int64_t c_bstride;
  int32_t M, N, K, batch;
  int32_t epilogue;
  int32_t gelu_from_col;
  const void* gate; int64_t gate_bstride;
  const void* res; int64_t ldr; int64_t r_bstride;
  /* optional caller-owned scratch (>= 16-byte aligned): when a GEMM has fewer 256x256 tiles than the device has CUs, the
   * auto path splits K into 2-4 slices, writes fp32 partials [slices][batch][M][N] here and finishes in a second pass
   * (deterministic: slices are summed in order).  NULL / too small = no split. */
  void* workspace; int64_t workspace_bytes;
  /* ABI 7 (round 6): per-batch weights, W [batch][N, K] with batch stride w_bstride elements (a multiple of 8; 0 = ONE weight matrix
   * for every batch sample, every nn.Linear).  bf16 entry points only; what the VAE mid-block attention's q k^T and P v products are
   * (k and v^T differ per image): one launch per query-row chunk for the whole batch instead of one per image. */
  int64_t w_bstride;
} tfx_gemm_args;
int tfx_gemm_bf16(const tfx_gemm_args* args, int variant, tfx_stream stream);

/* Round 6 (ABI 7).  The fused q | k | v (| mlp) projection of a FLUX block as ONE operation: tfx_gemm_bf16 (epilogue 0 or 1) whose
 * epilogue also applies the per-head RMSNorm (weights norm_q / norm_k, bf16 [128]) and the interleaved-pair RoPE to the q and k column
 * ranges [q0, q1) / [k0, k1) of the output (multiples of 256: whole pairs of 128-wide heads), rounding for rounding as tfx_rmsnorm_rope
 * does it after the fact (D/models/attention_processor.py:1990-1992, 2001-2004, 2023-2037; transformer_flux.py:715-739).  rope_cs: the
 * rotary table as fp32 (cos, sin) pairs [rows, 64, 2]; output row m of a batch sample uses table row pos0 + m.  This is the kernel
 * tfx_dit_forward launches internally for every block projection with at least as many 256 x 256 tiles as the device has CUs; shapes it
 * cannot take (K % 128 != 0, unaligned ranges, fewer tiles than CUs with a workspace: the K-sliced path has no such epilogue) are
 * refused -- run tfx_gemm_bf16 + tfx_rmsnorm_rope instead. */
typedef struct tfx_qkn_args {
  const void* norm_q; const void* norm_k;
  const float* rope_cs;
  int32_t pos0, q0, q1, k0, k1;
  float eps;
} tfx_qkn_args;
int tfx_gemm_bf16_qkn(const tfx_gemm_args* args, const tfx_qkn_args* qkn, tfx_stream stream);

/* Same operands, C = fp32 raw accumulators [batch][M, N] (ldc / c_bstride in floats, C 16-byte aligned); bias must be
 * NULL and epilogue 0.  Used where a product must reach its consumer unrounded: the q k^T scores of the VAE mid-block
 * attention (D/models/attention_processor.py:2858-2862) on their way to tfx_row_softmax. */
int tfx_gemm_bf16_f32(const tfx_gemm_args* args, tfx_stream stream);

/* ---- fp8 (OCP e4m3) variant of the same Linear, BASELINE config 5 "fp8 weights (CDNA4 fp8 MFMA)"; no reference
 *      counterpart (the reference computes in bf16).  A [batch][M, K] and W [N, K] hold one e4m3 byte per element
 *      (lda / ldw / a_bstride count elements = bytes), C = (A . W^T) * a_scale[b][m] * w_scale[n] + bias with the same
 *      epilogues, bf16 output.  K % 256 == 0, rows 16-byte aligned.  tfx_quantize_rows_fp8 produces both operands:
 *      out = round_e4m3(x / scale), scale[b * s_bstride + row] = max|x[row]| / 448 (1.0 for an all-zero row). */
int tfx_gemm_fp8(const tfx_gemm_args* args, const float* a_scale, int64_t a_scale_bstride, const float* w_scale,
                 tfx_stream stream);
int tfx_quantize_rows_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride,
                          float* scale, int64_t s_bstride, int32_t rows, int32_t batch, int32_t K, tfx_stream stream);
/* tfx_ln_modulate whose result is written as the e4m3 quantisation of the bf16 row (== tfx_ln_modulate followed by
 * tfx_quantize_rows_fp8, bit for bit): q8 [B][rows, D] bytes with row stride ldq / batch stride q_bstride (bytes),
 * q8_scale[b * s_bstride + row]. */
int tfx_ln_modulate_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* q8, int64_t ldq, int64_t q_bstride,
                        float* q8_scale, int64_t s_bstride, const void* shift, const void* scale, int64_t mod_bstride,
                        int32_t rows_per_batch, int32_t batch, int32_t D, float eps, tfx_stream stream);

/* ---- LayerNorm(no affine, eps) * (1 + scale[b]) + shift[b]   (AdaLayerNormZero / ZeroSingle / Continuous,
 *      D/models/normalization.py:170, 202, 365; norm2 + modulation, transformer_flux.py:820-821, 833-834).
 *      x, out: [batch][rows_per_batch, D]; shift, scale: [batch][D] with stride mod_bstride.  D % 8 == 0, D <= 3072. */
int tfx_ln_modulate(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride,
                    const void* shift, const void* scale, int64_t mod_bstride, int32_t rows_per_batch, int32_t batch,
                    int32_t D, float eps, tfx_stream stream);

/* ---- per-head RMSNorm * weight, then RoPE, in place on the q and k column ranges of a fused projection buffer
 *      [B][Ntok, ld] (heads of 128) (FluxAttnProcessor2_0, D/models/attention_processor.py:2001-2004, 2023-2037;
 *      RMSNorm D/models/normalization.py:534-549; apply_rotary_emb D/models/embeddings.py:899-918).
 *      Rows < T use w*_txt (norm_added_q/k), rows >= T use w*_img (norm_q/k).  cos/sin: fp32 [Ntok,128]. */
int tfx_rmsnorm_rope(void* buf, int64_t ld, int64_t bstride, int32_t q_off, int32_t k_off, int32_t H, int32_t Ntok,
                     int32_t T, int32_t B, const void* wq_img, const void* wk_img, const void* wq_txt,
                     const void* wk_txt, const float* cos_tab, const float* sin_tab, float eps, tfx_stream stream);

/* the name SURVEY.md §8(b) lists for the same entry point (q AND k of one fused buffer): identical arguments and behaviour */
int tfx_rmsnorm_rope_qk(void* buf, int64_t ld, int64_t bstride, int32_t q_off, int32_t k_off, int32_t H, int32_t Ntok,
                        int32_t T, int32_t B, const void* wq_img, const void* wk_img, const void* wq_txt,
                        const void* wk_txt, const float* cos_tab, const float* sin_tab, float eps, tfx_stream stream);

/* ---- gated residual out[b][r, :] = res[b][r, :] + bf16(gate[b][:] * x[b][r, :])   (transformer_flux.py:733-735, 817-818, 824-826,
 *      830-831, 837: `hidden_states + gate.unsqueeze(1) * attn_output`; the product is rounded to bf16 before the add, as the
 *      reference's two bf16 ops do).  Standalone form of tfx_gemm_args.epilogue 2 for a caller that keeps its own Linear (the
 *      attention-processor plugin level); inside tfx_dit_forward the same arithmetic runs in the producing GEMM's epilogue.
 *      x, res, out: bf16 [batch][rows_per_batch, D] with row strides ldx / ldr / ldo and batch strides; gate: bf16 [batch][D] with
 *      batch stride gate_bstride.  D % 8 == 0; out may alias res or x. */
int tfx_gate_residual(const void* x, int64_t ldx, int64_t x_bstride, const void* gate, int64_t gate_bstride, const void* res,
                      int64_t ldr, int64_t r_bstride, void* out, int64_t ldo, int64_t o_bstride, int32_t rows_per_batch,
                      int32_t batch, int32_t D, tfx_stream stream);

/* ---- seam blend of the tiled VAE (AutoencoderKL.blend_v / blend_h, D/models/autoencoders/autoencoder_kl.py:334-344), NHWC bf16,
 *      in place on b: for t in [0, extent), u in [0, len):  b[n][t][u][:] = bf16(bf16(a[n][t][u][:] * (1 - t/extent)) +
 *      bf16(b[n][t][u][:] * (t/extent)))  (the reference's python-float weights in fp32, its three bf16 roundings).  t walks rows for
 *      the vertical blend and columns for the horizontal one: *_tstride / *_ustride are the element strides of t and u, `a` points at
 *      the first of the neighbour tile's last `extent` rows (columns).  C % 8 == 0. */
int tfx_blend_edge_nhwc(const void* a, int64_t a_bstride, int64_t a_tstride, int64_t a_ustride, void* b, int64_t b_bstride,
                        int64_t b_tstride, int64_t b_ustride, int32_t batch, int32_t extent, int32_t len, int32_t C, tfx_stream stream);

/* ---- joint attention softmax(q k^T * scale) v, head_dim 128, no mask (F.scaled_dot_product_attention,
 *      D/models/attention_processor.py:2039-2041).  Element (b, n, h, d) of q is q[b*q_bstride + n*ldq + h*128 + d];
 *      same for k, v, o.  o may alias q. */
typedef struct tfx_attn_args {
  const void* q; const void* k; const void* v; void* o;
  int64_t ldq, ldk, ldv, ldo;
  int64_t q_bstride, k_bstride, v_bstride, o_bstride;
  int32_t B, H, N;
  float scale;
  /* optional promise of the caller: |scale * q . k| <= score_bound for every (query, key) pair, 0 = unknown.  Softmax does not
   * depend on the reference that is subtracted before the exponential; when score_bound * log2(e) + log2(N) + 24 <= 126 (P1024's
   * N = 4608: score_bound <= 62.2; every weight 2^+-90 at most, and a row's sum over N keys times |v| up to 2^24 stays inside the
   * exponent range fp32 and bf16 share) the kernel subtracts none: no running row maximum, no rescaling branch.  tfx_dit_forward
   * takes the bound per block from that block's q / k RMSNorm weights (after the norm |q| <= sqrt(128) max|w_q|, RoPE preserves
   * it).  A wrong promise can overflow to inf / NaN; 0 (or a bound beyond the limit) keeps the kernel's own overflow guard. */
  float score_bound;
  /* optional scratch (ABI 8; NULL = none), 16-byte aligned, 2 * 256 * 256 * 132 * 4 = 69.2 MB for any shape: lets the head-dim-128 kernel deal
   * (item, 64-key tile) units to the CUs instead of whole (b, h, 256-query) items ("attention_streamk", tfx_set_option) when that fills
   * the chip better; contents are scratch, the launch's own stream orders its uses.  tfx_dit_forward passes the split-K scratch. */
  void* workspace; int64_t workspace_bytes;
} tfx_attn_args;
int tfx_joint_attention(const tfx_attn_args* args, tfx_stream stream);

/* ---- scheduler steps on packed latents x [rows, C] with model output v [rows, C]; the result is also written
 *      into columns [0, C) of xin [rows, ldxin] when xin != NULL (next x_embedder input; replaces the torch.cat of
 *      D/pipelines/flux/pipeline_flux_fill.py:2085).  The step index is *step_ptr (device int) when step_ptr != NULL,
 *      else `step`.
 *      Euler: coef[step] = sigma_{i+1} - sigma_i  (FlowMatchEulerDiscreteScheduler.step,
 *             D/schedulers/scheduling_flow_match_euler_discrete.py:319-330)
 *      AMO:   coef[3*step..] = {t_over - t, a, b}, noise fp32 [rows, C]  (StochasticRFOvershotDiscreteScheduler.step,
 *             D/schedulers/scheduling_stochastic_rf_discrete_overshot.py:306-361 with attn_map None). */
int tfx_euler_step(const void* v, void* x, void* xin, int64_t ldxin, int32_t C, int64_t rows, const float* coef,
                   const int32_t* step_ptr, int32_t step, tfx_stream stream);
int tfx_amo_step(const void* v, void* x, void* xin, int64_t ldxin, int32_t C, int64_t rows, const float* coef,
                 const int32_t* step_ptr, int32_t step, const float* noise, tfx_stream stream);

/* ---- small ops of the conditioning path (CombinedTimestepGuidanceTextProjEmbeddings, D/models/embeddings.py:1327-1339) */
int tfx_timestep_embedding(const float* t, void* out /* [n,256] = cos|sin */, int32_t n, tfx_stream stream);
int tfx_silu(const void* a, void* out, int64_t n, tfx_stream stream);
int tfx_add(const void* a, const void* b, void* out, int64_t n, tfx_stream stream);
/* dst[r, col0 : col0+C] = src[r, :]  (initial fill of the x_embedder input; C, col0, ld multiples of 8) */
int tfx_scatter_cols(const void* src, void* dst, int64_t rows, int32_t C, int64_t ld, int32_t col0, tfx_stream stream);
/* generic strided 2-D copy dst[b][r, 0:cols] = src[b][r, 0:cols]  (cols % 8 == 0) */
int tfx_copy_rows(const void* src, int64_t src_ld, int64_t src_bstride, void* dst, int64_t dst_ld, int64_t dst_bstride,
                  int32_t rows, int32_t cols, int32_t batch, tfx_stream stream);
/* device-side step cursor for single-graph replay: cur[:] = table[*step_ptr][:]; *step_ptr += 1 */
int tfx_select_step(const void* table, void* cur, int64_t per_step_elems, int32_t* step_ptr, tfx_stream stream);
int tfx_advance_step(int32_t* step_ptr, tfx_stream stream);

/* ---- one full FluxTransformer2DModel.forward (D/models/transformers/transformer_flux.py:1028-1212) ------------------
 * Weight layout (all bf16, nn.Linear [out,in]); fusions are pure row concatenations done at load time:
 *   double block: qkv_img = [to_k; to_v; to_q] (3D x D), qkv_txt = [add_k_proj; add_v_proj; add_q_proj],
 *                 out_img = to_out.0, out_txt = to_add_out, ff1/ff2 = ff.net.0.proj / ff.net.2 (and ff_context)
 *   single block: qkv_mlp = [to_k; to_v; to_q; proj_mlp] (7D x D), proj_out (D x 5D: attn | mlp columns)
 * Modulation: mod [B][mod_len] for THIS step = Linear(SiLU(temb)) of every norm*.linear stacked in the order
 *   double i: [img: shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp | txt: same six]   (12D each)
 *   single j: [shift scale gate] (3D each), then norm_out: [scale shift] (2D).
 * Workspace (caller-owned, bf16): hid [B][N,D], xn [B][N,D], y [B][N,7D], with N = T + S (text rows first). */
/* w8 / w8_scale (optional, may be NULL): the same weight as e4m3 bytes [out,in] with one fp32 scale per output channel
 * (tfx_quantize_rows_fp8 of w); used when tfx_dit_desc.flags bit 2 is set and in % 256 == 0, see below. */
typedef struct tfx_linear { const void* w; const void* b; const void* w8; const float* w8_scale; } tfx_linear;
/* attn_score_bound (ABI 6, optional): tfx_attn_args.score_bound of THIS block's attention launch, from this block's own q / k
 * RMSNorm weights -- 128 * max(max|norm_q|, max|norm_added_q|) * max(max|norm_k|, max|norm_added_k|) * 128^-0.5 (+ rounding margin).
 * 0 = unknown: the launch falls back to tfx_dit_desc.attn_score_bound (0 there too = no promise).  Per block, so that one block
 * with large norm scales only changes the form of its own attention launch. */
typedef struct tfx_double_block {
  tfx_linear qkv_img, qkv_txt, out_img, out_txt, ff1_img, ff2_img, ff1_txt, ff2_txt;
  const void *norm_q, *norm_k, *norm_added_q, *norm_added_k;
  float attn_score_bound;
} tfx_double_block;
typedef struct tfx_single_block {
  tfx_linear qkv_mlp, proj_out;
  const void *norm_q, *norm_k;
  float attn_score_bound;
} tfx_single_block;
typedef struct tfx_dit_desc {
  int32_t D, H, in_channels, out_channels, n_double, n_single;
  int32_t B, S, T;
  tfx_linear x_embedder, proj_out;
  const tfx_double_block* dbl;   /* host array [n_double] */
  const tfx_single_block* sgl;   /* host array [n_single] */
  const void* xin;               /* [B][S, in_channels] */
  const void* ctx0;              /* [B][T, D] context_embedder output (step invariant) */
  const void* mod; int64_t mod_bstride;
  const float* cos_tab; const float* sin_tab;  /* fp32 [N,128] */
  void* hid; void* xn; void* y;
  void* out;                     /* [B][S, out_channels] */
  /* Partial runs (block-level parity tests): blocks [first_block, last_block) of the n_double + n_single sequence
   * (last_block < 0 = all); flags bit 0: skip x_embedder / ctx0 copy (hid is preloaded with [text | image] rows),
   * bit 1: skip norm_out + proj_out.  A full forward is first_block = 0, last_block = -1, flags = 0. */
  int32_t first_block, last_block, flags;
  /* flags bit 2 (BASELINE config 5, fp8 linears): every block Linear that carries w8 runs as
   * tfx_quantize_rows_fp8(activations) -> tfx_gemm_fp8; q8 [B][N, 5D] bytes and q8_scale [B][N] fp32 are the
   * caller-owned workspace of the quantised activations.  Embedders, modulation and proj_out stay bf16. */
  void* q8; float* q8_scale;
  void* gemm_workspace; int64_t gemm_workspace_bytes;   /* optional, passed to every block Linear (tfx_gemm_args.workspace) */
  /* optional: the rotary table as (cos, sin) pairs, fp32 [N, 64, 2] (cos_tab / sin_tab hold every value twice: pair i =
   * columns 2i, 2i + 1).  When set, the q | k | v (| mlp) projections with at least as many 256 x 256 tiles as the device
   * has CUs apply the per-head RMSNorm + RoPE in their GEMM epilogue (bf16 mode, and since round 5 the fp8 mode of flags bit 2);
   * the others -- and every projection when this is NULL -- are followed by tfx_rmsnorm_rope as a separate pass.  Same rounding
   * points either way. */
  const float* rope_cs;
  /* optional: the flow-matching Euler update fused into proj_out's epilogue (scheduling_flow_match_euler_discrete.py:319-330:
   * x' = x + (sigma_next - sigma) * v).  euler_gate: bf16 [B][out_channels] rows (row stride euler_gate_bstride elements), every
   * element = the step's bf16-rounded dsigma.  When set, the final projection runs with the gated-residual epilogue --
   * xin[:, :, :out_channels] <- xin[:, :, :out_channels] + bf16(dsigma * bf16(proj_out(...))), the reference's rounding points,
   * bit-identical to tfx_euler_step on the stored model output -- i.e. the latent state lives IN the x_embedder input, `out` is
   * not written, and there is neither a scheduler launch nor a copy of the new latents into the next step's input. */
  const void* euler_gate; int64_t euler_gate_bstride;
  /* optional: tfx_attn_args.score_bound for the attention launches of blocks whose own attn_score_bound is 0 (0 = unknown).  The
   * caller derives it from the q / k RMSNorm weights: 128 * max|w_q| * max|w_k| * 128^-0.5 (text-stream norms included), see
   * tfx_attn_args; since ABI 6 the per-block fields are the ones the engine fills. */
  float attn_score_bound;
} tfx_dit_desc;
int tfx_dit_forward(const tfx_dit_desc* desc, tfx_stream stream);

/* Caller-owned workspace of tfx_dit_forward for one problem size, as ONE allocation: total bytes, and the byte offsets
 * (256-byte aligned) of its parts in off[0..5] = hid, xn, y, q8, q8_scale, gemm_workspace (q8 / q8_scale only when
 * flags bit 2 (fp8 linears) is set, else -1); *gemm_workspace_bytes = size of the split-K / stream-K scratch (128 MiB).  Host-side
 * arithmetic only, no GPU needed.  ("never allocate persistent memory except through an explicit
 * workspace the Python side owns".) */
int64_t tfx_workspace_bytes(int32_t B, int32_t S, int32_t T, int32_t D, int32_t flags);
int tfx_workspace_layout(int32_t B, int32_t S, int32_t T, int32_t D, int32_t flags, int64_t* off, int64_t* gemm_workspace_bytes);

/* ---- one denoising step = modulation rows of step *step_ptr -> tfx_dit_forward -> scheduler update -> ++*step_ptr, and
 *      its capture into a hipGraph (D/pipelines/flux/pipeline_flux_fill.py:2077-2116: the loop body; the reference issues
 *      ~4.5 k launches per step from Python, here a step is ONE hipGraphLaunch).  The step index lives on the device, so the
 *      same graph serves every step: mod_table [n_steps][B][mod_len] (tfx_dit_desc.mod must point at mod_cur, which the step
 *      refreshes from the table), coef as for tfx_euler_step / tfx_amo_step (sampler 0 / 1), noise fp32 [B*S, out_channels]
 *      (AMO only; rewritten by the caller between replays), latents [B*S, out_channels] updated in place and mirrored into
 *      columns [0, out_channels) of dit.xin.
 *      tfx_dit_step_run: the eager form (call it at least once before capturing: first launches size LDS limits).
 *      tfx_dit_step_capture: records the same launches on `stream` (must not be the NULL stream) in thread-local capture
 *      mode and instantiates them; tfx_dit_step_replay launches the instantiated graph; tfx_graph_destroy frees it.
 *      All pointers of the descriptor are baked into the graph: they must stay valid (and unmoved) while it is replayed. */
typedef struct tfx_step_desc {
  tfx_dit_desc dit;
  const void* mod_table; void* mod_cur; int64_t mod_step_elems;   /* elements per step = B * mod_len */
  int32_t* step_ptr;
  void* latents;
  const float* coef; const float* noise;
  int32_t sampler;                                                 /* 0 Euler, 1 AMO, 2 Euler fused into proj_out's epilogue:
                                                                      dit.euler_gate must point into mod_cur (the step's dsigma row
                                                                      travels with its modulation rows: mod_step_elems covers it), the
                                                                      latents live in dit.xin[:, :, :out_channels]; `latents` / `coef` are
                                                                      not used -- copy the result out with tfx_copy_rows after the loop */
} tfx_step_desc;
typedef void* tfx_graph;
int tfx_dit_step_run(const tfx_step_desc* step, tfx_stream stream);
int tfx_dit_step_capture(const tfx_step_desc* step, tfx_stream stream, tfx_graph* out);
int tfx_dit_step_replay(tfx_graph graph, tfx_stream stream);
int tfx_graph_destroy(tfx_graph graph);

/* ---- VAE ends (AutoencoderKL, D/models/autoencoders/autoencoder_kl.py:263-332) on NHWC bf16 activations -------------
 * 3x3 convolution as an implicit GEMM on the MFMA kernel (ResnetBlock2D convs D/models/resnet.py:327-366, Upsample2D
 * nearest-2x + conv D/models/upsampling.py:142-192 with up = 2, Downsample2D pad (0,1,0,1) stride 2
 * D/models/downsampling.py:132-150 with stride = 2, pad_lo = 0).  x [B, inH, inW, Cin] (Cin % 64 == 0),
 * w [Cout, 3, 3, Cin] (KRSC) -- or the narrow form of the two conv_in layers: Cin in {8, 16, 32} with w [Cout, Kp],
 * Kp = 9 Cin rounded up to a multiple of 64, columns (tap, ci) then zeros --, out [B, H, W, Cout] = conv(x) + bias (+ res when res != NULL; may alias out),
 * zero_page: >= 128 bytes of zeros (source of out-of-image taps).  variant: -1 auto, 0 generic, 1 MFMA. */
int tfx_conv3x3_nhwc(const void* x, int32_t B, int32_t inH, int32_t inW, int32_t Cin, const void* w, const void* bias,
                     void* out, int32_t H, int32_t W, int32_t Cout, int32_t stride, int32_t up, int32_t pad_lo,
                     const void* res, const void* zero_page, int variant, tfx_stream stream);
/* The same convolution (stride 1, pad 1, no upsample) in PIXEL-PAIR form, for layers with few output channels: one GEMM row is two
 * horizontally adjacent output pixels (W even), N = 2 * Cout -- their channels are neighbours in NHWC, so `out` / `res` are the
 * ordinary [B, H, W, Cout] tensors -- and K runs over the 3 x 4 input taps the pair touches: w_pair [2 * Cout][3][4][Cin] holds,
 * for pixel o of the pair, the kernel's column dx at position dx + o and zeros elsewhere; bias_pair [2 * Cout] = the bias twice.
 * 4/3 of the FLOPs, but a 128-channel layer then fills the MFMA kernel's 256-column tile instead of half of it (VAE: the
 * full-resolution blocks of D/models/autoencoders/vae.py:60-360; same result up to fp32 summation order). */
int tfx_conv3x3_pair_nhwc(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w_pair, const void* bias_pair,
                          void* out, int32_t Cout, const void* res, const void* zero_page, tfx_stream stream);
/* GroupNorm(groups, eps, affine) followed by SiLU when silu != 0, x/out [B, HW, C] NHWC.  workspace: fp32
 * [B * (ceil(HW/1024) + 1) * groups * 2]. */
int tfx_groupnorm_nhwc(const void* x, void* out, const void* gamma, const void* beta, float* workspace, int32_t B,
                       int64_t HW, int32_t C, int32_t groups, float eps, int32_t silu, tfx_stream stream);

/* ---- the pixel / latent layout steps around the VAE (FluxFillPipeline.__call__, D/pipelines/flux/pipeline_flux_fill.py
 *      "P:", and VaeImageProcessor, D/image_processor.py "IP:"), so that no NCHW tensor and no torch op sits between the
 *      caller's image and the encoder, or between the decoder and the caller's image.  dtype codes: 0 f32, 1 bf16, 2 u8. */
/* *flag |= 1 when any element of x is negative (IP:700-707: tensors already in [-1, 1] are not re-normalised). */
int tfx_any_negative(const void* x, int32_t dtype, int64_t n, int32_t* flag, tfx_stream stream);
/* out [B, H, W, 8] bf16 NHWC (channels >= C zero) = bf16( norm(img) * (1 - mask) ): the encoder input.  img: f32 / bf16
 * planes [B, C, H, W], or u8 interleaved [B, H, W, C] read as value / 255 (PIL branch, IP:133-154 inverse).  mask: NULL or
 * f32 / u8 [mask_batch, H, W] (mask_batch 1 or B), binarised at 0.5 when binarize != 0 (IP:535-536).  norm_mode: 0 none,
 * 1 x -> 2x - 1 (IP:221-225), 2 the same unless *neg_flag != 0.   (IP:587-716 + P:2030-2031) */
int tfx_prep_image(const void* img, int32_t img_dtype, const void* mask, int32_t mask_dtype, void* out, int32_t B, int32_t C,
                   int32_t H, int32_t W, int32_t mask_batch, int32_t norm_mode, int32_t binarize, const int32_t* neg_flag,
                   tfx_stream stream);
/* The callers' composition step on the device (reference: /root/reference/run_inference.py:409-467 and
 * scripts/run_eval.py:143-198 stack the rendered glyph image and the scene -- glyph first -- and a black mask for the glyph part
 * over the scene's mask; IP's do_convert_grayscale then takes PIL's "L" of the RGB mask): canvas [B, H, W, 3] u8 and
 * cmask [B, H, W] u8 from glyph [B, gh, gw, 3], scene [B, sh, sw, 3] and the scene's RGB mask [B, sh, sw, 3] (u8,
 * interleaved).  direction 0: vertical, H = gh + sh, W = gw = sw; 1: horizontal, W = gw + sw, H = gh = sh.  Grey value =
 * Pillow's (19595 R + 38470 G + 7471 B + 0x8000) >> 16.  mask_rgb != 0: cmask is [B, H, W, 3] and keeps the RGB values (a
 * resize follows: the reference resizes the RGB mask, then takes "L").  The outputs feed tfx_prep_image (dtype 2) /
 * tfx_pack_mask, directly or through tfx_resample_u8 + tfx_rgb_to_grey_u8. */
int tfx_compose_canvas(const void* glyph, const void* scene, const void* scene_mask_rgb, void* canvas, void* cmask, int32_t B,
                       int32_t gh, int32_t gw, int32_t sh, int32_t sw, int32_t direction, int32_t mask_rgb, tfx_stream stream);
/* out [pixels] u8 = PIL convert("L") of interleaved RGB u8 [pixels][3]. */
int tfx_rgb_to_grey_u8(const void* rgb, void* out, int64_t pixels, tfx_stream stream);
/* One pass of Pillow's 8-bit convolution resampler (PIL.Image.resize of "RGB" / "L" images; the callers' resize to a multiple
 * of 32, /root/reference/run_inference.py:65-69) along the middle axis of in [outer][in_len][inner] u8 -> out
 * [outer][out_len][inner]: horizontal pass of [B, H, W, C]: outer = B H, len = W, inner = C; vertical pass: outer = B, len = H,
 * inner = W C (Pillow runs the horizontal pass first; both round to u8).  bounds [out_len][2] = (first input index, count),
 * coeffs [out_len][ksize] = the filter weights in fixed point with 22 fractional bits, computed on the host exactly as
 * libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc do (textflux_amd/image_processor.py::pil_resample_tables);
 * out = clip8((2^21 + sum in * k) >> 22): integer arithmetic, bit-identical to Pillow. */
int tfx_resample_u8(const void* in, void* out, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int64_t outer,
                    int32_t in_len, int32_t out_len, int32_t inner, tfx_stream stream);
/* out[b, t, col0 + (i*8+j)*4 + py*2+px] = mask[(2ty+py)*8 + i, (2tx+px)*8 + j]  (P:1563-1580: 8x8 pixel blocks -> channels,
 * then _pack_latents), t = ty * (W/16) + tx, row stride ld. */
int tfx_pack_mask(const void* mask, int32_t mask_dtype, void* out, int32_t B, int32_t H, int32_t W, int32_t mask_batch,
                  int32_t binarize, int64_t ld, int32_t col0, tfx_stream stream);
/* moments [B, h, w, 2L] NHWC bf16 (mean | logvar: the encoder's conv_out), eps [B, L, h, w] f32 / bf16 or NULL (mode) ->
 * out[b, t, col0 + c*4 + py*2+px] = ((mean + exp(0.5 clamp(logvar, -30, 20)) * eps) - shift) * scale, every intermediate
 * rounded to bf16 as the reference's tensor ops do (D/models/autoencoders/vae.py:781-802, P:1528-1530, 1743-1748). */
int tfx_vae_sample_pack(const void* moments, const void* eps, int32_t eps_dtype, void* out, int32_t B, int32_t h, int32_t w,
                        int32_t L, float shift, float scale, int64_t ld, int32_t col0, tfx_stream stream);
/* latents [B, (h/2)(w/2), >= 4L] (row stride ld) -> z [B, h, w, L] NHWC bf16 = latents / scale + shift, un-patchified
 * (P:1752-1765, 2126-2127): the decoder input. */
int tfx_unpack_latents(const void* latents, int64_t ld, void* out, int32_t B, int32_t h, int32_t w, int32_t L, float shift,
                       float scale, tfx_stream stream);
/* x [B, H, W, Cs] NHWC bf16 (first C channels), window rows [y0, y0 + Hc) x columns [x0, x0 + Wc) -> mode 0 NCHW bf16 |
 * 1 NHWC f32 | 2 NHWC u8 = round(255 v) | 3 NCHW f32; denorm != 0: v = clamp(x / 2 + 0.5, 0, 1) in bf16 steps (IP:227-239,
 * 196-209, 133-154).  The window is the callers' result crop (run_inference.py:460-465), applied on the device. */
int tfx_postprocess(const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t Cs, int32_t C, int32_t mode, int32_t denorm,
                    int32_t y0, int32_t x0, int32_t Hc, int32_t Wc, tfx_stream stream);
/* helpers of the VAE mid-block attention (one head of dim C over h*w tokens, AttnProcessor2_0,
 * D/models/attention_processor.py:2799-2881): out[b][c, n] = in[b][n, c]; p[r, :N] = bf16(softmax(scale * s[r, :N])) with
 * s fp32 (row stride lds) from tfx_gemm_bf16_f32 and p bf16 (row stride ldp), fp32 statistics -- what a flash kernel
 * keeps in registers, here through HBM because one head of dim 512 does not fit the 128-wide attention kernel. */
int tfx_transpose(const void* in, int64_t ldi, int64_t in_bstride, void* out, int64_t ldo, int64_t out_bstride, int32_t N,
                  int32_t C, int32_t batch, tfx_stream stream);
int tfx_row_softmax(const float* s, int64_t lds, void* p, int64_t ldp, int32_t rows, int32_t N, float scale, tfx_stream stream);

/* ---- text encoders of encode_prompt (P:1411-1503: CLIP-L pooled output of `prompt`, T5-XXL sequence of `prompt_2`).  The
 *      reference calls third-party `transformers` (pinned 4.43.3: models/t5/modeling_t5.py T5Stack / T5Block / T5Attention /
 *      T5LayerNorm / T5DenseGatedActDense; models/clip/modeling_clip.py CLIPTextTransformer / CLIPEncoderLayer / CLIPAttention /
 *      CLIPMLP); the Linear layers run on tfx_gemm_bf16 (residual adds in its epilogue: the bf16 residual stream of a
 *      torch_dtype = bfloat16 load), and: */
/* nn.LayerNorm(D, eps) with elementwise affine over bf16 rows: out[r, :] = bf16((x[r, :] - mean) * rstd * gamma + beta), fp32
 * arithmetic and ONE rounding (what F.layer_norm does on bf16 tensors; CLIPEncoderLayer.layer_norm1 / 2, final_layer_norm).
 * Any gamma (no (1 + scale) re-parameterisation).  D % 8 == 0, D <= 3072, ldx / ldo in elements, multiples of 8. */
int tfx_layernorm(const void* x, int64_t ldx, void* out, int64_t ldo, const void* gamma, const void* beta, int64_t rows,
                  int32_t D, float eps, tfx_stream stream);
/* softmax(scale * q k^T + bias) v for heads of dim 64 and N <= 512 keys (tfx_attn_args with 64-wide heads: element (b, n, h, d)
 * at base + b*bstride + n*ld + h*64 + d).  rel_bias: NULL or fp32 [H, 2N-1], bias(h, query, key) = rel_bias[h][key - query +
 * N - 1] (T5Attention.compute_bias: a function of key - query only); causal != 0 masks key > query (CLIP). */
int tfx_attention64(const tfx_attn_args* args, const float* rel_bias, int32_t causal, tfx_stream stream);
/* T5LayerNorm: out[r, :] = bf16(w * bf16(x[r, :] * rsqrt(mean(x[r, :]^2) + eps))); x f32 (x_dtype 0) or bf16 (1). */
int tfx_rmsnorm(const void* x, int32_t x_dtype, int64_t ldx, const void* w, void* out, int64_t ldo, int64_t rows, int32_t D,
                float eps, tfx_stream stream);
/* out[i, :] = table[ids[i], :] (bf16 rows of D elements, ids int64 clamped to [0, vocab)): nn.Embedding. */
int tfx_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int32_t D, int64_t vocab, tfx_stream stream);
/* fp32 accumulation buffer helper (T5 under torch_dtype = float16 keeps `wo` in fp32 and its residual stream promotes to fp32;
 * under bfloat16 -- what the reference loads -- the stream is bf16 and this is not on the path): mode 0 x += y (bf16),
 * 1 x += y (f32), 2 x = y (bf16 -> f32). */
int tfx_add_into_f32(float* x, const void* y, int64_t n, int32_t mode, tfx_stream stream);
/* mode 0: out = a * b (T5DenseGatedActDense: gelu(wi_0 x) * wi_1 x), mode 1: out = a * sigmoid(1.702 a) (CLIP quick_gelu);
 * bf16 [rows, cols] with row strides. */
int tfx_mul_act(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int32_t cols,
                int32_t mode, tfx_stream stream);

/* ---- tuning knobs (no reference counterpart).  "attention_waves" selects the attention kernel: 30 (default) = one
 *      256-thread workgroup per CU -- one wave per SIMD, 64 query rows per wave, the tile loop pipelined per (32-key block,
 *      32-row q-block) unit (attention_w4.hip); it stores whole output rows in 16-byte pieces, so output views whose row or
 *      batch stride is not a multiple of 8 elements are served by 10.  10 = one 512-thread workgroup of 256 query rows per
 *      CU, 32 rows per wave; both keep the softmax bookkeeping on the matrix pipe (pre-scaled Q, lazy reference maximum
 *      subtracted by an extra MFMA k-step, row sums from a ones-block of the PV MFMA).  8 = the 512-thread schedule with the
 *      textbook exact online maximum (the independent implementation the tests compare with); 20 = 10 pipelined per
 *      half-tile (attention_hp.hip).  Bench builds only (-DTFX_BENCH): 12 / 4 = two independent 256-thread workgroups per
 *      CU of 10 / 8; 9 = 8 with 128 keys per barrier; 16 = softmax / MFMA ping-pong between the wave groups.  All compute
 *      the same softmax; 10, 12, 20 and 30 differ from the others in rounding only (one extra bf16 rounding of q * scale,
 *      row sums of the bf16 weights); 40 = 30 on v_mfma_f32_16x16x32_bf16 (attention_w16.hip); 0 restores the default.
 *      "attention_streamk": 1 (default) lets kernel 30 deal the (b, h, 256-query) items of a batch sample's last, partly filled
 *                         round of CUs as (item, 64-key tile) units -- every CU of the sample's group gets the same number of key
 *                         tiles, an item cut by a CU boundary is finished by a merge pass -- when tfx_attn_args.workspace is given
 *                         and the library's estimate says it pays (P1024 batch 8: -1.1 % per launch, 2048 x 1024 batch 1: -12 %,
 *                         1024 x 672 batch 1: -16 %); 2 = whenever admissible; 0 = whole items only: a sample's bits then do not
 *                         depend on how many samples share its batch (identical samples of ONE batch agree either way).
 *      "attention_tail_split": 1 lets kernel 30 cut the q-tiles of a partly filled last round of workgroups into two key ranges
 *      (+ a merge kernel; faster at small batches, but a sample's bits then depend on the batch size: default 0).
 *      "gemm_group_m": row tiles per group of the GEMM tile order (default 0 = by shape: 1 for N <= 3072, else 4).
 *      "gemm_place": slot assignment of the persistent GEMM's operand requests, 1 or 2 (default 2; gemm.hip).
 *      "gemm_splitk": 0 disables the split-K path of few-tile GEMMs (default 1).
 *      "fp8_fuse_qkn": 0 = in fp8 mode the q | k | v (| mlp) projections are followed by the separate q / k norm + RoPE pass (default 1:
 *      fused into the e4m3 GEMM's epilogue like the bf16 mode's).
 *      "ln_joint": 0 = the LayerNorm + modulation of a double block's text and image rows as two launches (default 1: one launch over the
 *      joint rows, the text rows taking the second modulation; bit-identical, round 6).
 *      "ln_prefetch": the LayerNorm + modulation kernel requests the modulation rows up front: 0 never, 1 always, 2 (default) for <= 16384
 *      rows (latency-bound launches: batch 1); bit-identical either way. */
int tfx_set_option(const char* name, int value);
/* The only device memory the library ever allocates itself is behind an opt-in knob: the attention tail-split partials
 * ("attention_tail_split" 1; 138 MB per (device, stream) that launched with it, at most 8).  tfx_release_scratch synchronises the
 * device and frees them all; streams seen afterwards allocate afresh.  Graphs captured while the knob was on hold the old pointers
 * and must be destroyed first (tfx_graph_destroy).  With the defaults this is a no-op.  Returns 0. */
int tfx_release_scratch(void);

/* ---- measurement hooks (no reference counterpart: the reference has no profiling) --------------------
 * When enabled, every MFMA-GEMM (kind 0) / attention (kind 1) launch is bracketed by hipEvents on its own stream and
 * its algorithmic FLOPs (2*M*N*K*batch, 4*B*H*N^2*128) are recorded; tfx_prof_collect waits for those launches and
 * returns the summed kernel time, FLOPs and launch count (kind 2: fp8 GEMM launches), then clears the records.  Not capturable into a graph. */
int tfx_prof_enable(int on);
int tfx_prof_collect(int kind, double* total_ms, double* total_flops, int* launches);
/* Round 6 (ABI 7).  The matrix-pipe rate the BOARD sustains at its power cap, as the denominator of a roofline fraction that does not
 * depend on which box runs the bench: one launch of a kernel that issues nothing but the GEMM kernels' own MFMA sections
 * (v_mfma_f32_16x16x32_bf16; fp8 != 0: v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales) on fragments read ONCE from `operands`
 * (operand_bytes >= 16 KiB of the caller's data -- random N(0, 1) values for the worst-case figure; power, and with it the clock,
 * depends on operand entropy), one 8-wave workgroup per CU, `ktiles` (a multiple of 48) K-tiles of 64 (fp8: 32) MFMAs per wave.
 * Writes the launch's FLOPs to *flops; the caller times it (events on `stream`) over 2-3 s: shorter runs finish inside the power
 * controller's averaging window and report the uncapped clock.  No reference counterpart. */
int tfx_mfma_peak_probe(const void* operands, int64_t operand_bytes, int32_t fp8, int32_t ktiles, double* flops, tfx_stream stream);
/* Which form of the attention kernel the launches took (host-side counters, also counted at graph capture, not at replay):
 * counts[0..7] = launches since the last reset of { 0: kernel 30 with its own overflow guard, 1..3: its option-31..33 variants,
 * 4: kernel 30 reference-free (score_bound accepted), 5: attention_hp (20), 6: the 16 x 16 kernel (40), 7: any other schedule };
 * counts[8] (ABI 8) = how many of those launches dealt the items of their last, partly filled round as (item, 64-key tile) units
 * ("attention_streamk").  Copies min(n, 9) counters, then clears them when reset != 0.  Returns the number copied. */
int tfx_attention_mode_counts(int64_t* counts, int32_t n, int32_t reset);
/* Phase timing of the attention kernels (tools/attn_timing.py): buf = device uint64 [blocks][8 waves][4 phases] that the
 * instrumented kernel variants fill with cycle counts, NULL switches back to the plain kernels. */
int tfx_debug_attention_timing(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* TEXTFLUX_HIP_H */
