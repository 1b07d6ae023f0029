// Internal (C++) launch API shared between the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfx {

// Error plumbing: every launcher returns 0 on success; on failure the message is kept per thread and is
// readable through tfx_last_error().
int fail(const char* fmt, ...);
int check_launch(const char* what);
const char* last_error();

// Optional per-launch timing of the dominant kernels (bench.py's live roofline measurement): when enabled, the
// GEMM / attention launchers bracket each launch with hipEvents on the launch stream and account its algorithmic
// FLOPs.  kind: 0 = gemm8p, 1 = attention.
void prof_enable(int on);
bool prof_on();
bool prof_on(hipStream_t st);  // false while `st` is being captured into a graph
void prof_begin(int kind, double flops, hipStream_t st);
void prof_end(int kind, hipStream_t st);
int prof_collect(int kind, double* total_ms, double* total_flops, int* launches);

// GEMM epilogues (C = epi(A @ W^T + bias)).
enum Epilogue : int {
  EPI_BIAS = 0,           // C = acc + bias
  EPI_BIAS_GELU = 1,      // C = gelu_tanh(acc + bias) for columns >= gelu_from_col, plain bias below it
  EPI_BIAS_GATE_RES = 2,  // C = res + gate[b, col] * (acc + bias)
  EPI_BIAS_RES = 3,       // C = res + (acc + bias)   (VAE residual blocks)
};

struct GemmArgs {
  const void* A; int64_t lda; int64_t a_bstride;     // activations  [batch][M, K] bf16, row stride lda
  const void* W;                                      // weights      [N, K] bf16 (nn.Linear layout), row stride ldw
  int64_t ldw;
  int64_t w_bstride = 0;                              // round 6: per-batch weights [batch][N, K] (elements; 0 = one W for every batch sample): the VAE mid-block attention's k / v^T
  const void* bias;                                   // [N] bf16 or null
  void* C; int64_t ldc; int64_t c_bstride;           // output       [batch][M, N] bf16
  int M, N, K, batch;
  int epilogue;
  int gelu_from_col;                                  // EPI_BIAS_GELU: first column that gets GELU
  const void* gate; int64_t gate_bstride;             // EPI_BIAS_GATE_RES: gate [batch][N] bf16
  const void* res; int64_t ldr; int64_t r_bstride;    // residual [batch][M, N] bf16 (may alias C)
  // implicit-GEMM 3x3 convolution on NHWC activations (conv_cin > 0): A = input [B, inH, inW, Cin], row m = output pixel
  // (b, y, x) of an H x W grid, K = 9 * Cin ordered (tap, ci); the input is read at ((y*stride - pad_lo + dy) >> up_shift,
  // (x*stride - pad_lo + dx) >> up_shift) -- up_shift = 1 folds a nearest 2x upsample into the gather -- and taps that
  // fall outside the (upsampled) input read `zero_page` (>= 128 B of zeros) instead.
  int conv_cin = 0, conv_inH = 0, conv_inW = 0, conv_H = 0, conv_W = 0, conv_stride = 1, conv_up_shift = 0, conv_pad_lo = 1;
  // pixel-pair form (conv_kw = 4, conv_stride_x = 2): one GEMM row = TWO horizontally adjacent output pixels, N = 2 * Cout (their
  // channels are neighbours in NHWC), K = 3 x 4 taps x Cin over the 4 input columns the pair touches, weights [2 Cout][3][4][Cin]
  // with zeros where a pixel does not use a column.  A third more FLOPs, but a 128-channel layer fills the 256-column tile.
  int conv_kw = 3, conv_stride_x = 0;
  const void* zero_page = nullptr;
  // fp8 (e4m3) operands, gemm_fp8 only: A and W hold one byte per element, C = (A.W^T) * a_scale[b][m] * w_scale[n] + bias
  const float* a_scale = nullptr; int64_t a_scale_bstride = 0;   // per activation row
  const float* w_scale = nullptr;                                 // per output channel
  // fused epilogue of the DiT's q | k | v (| mlp) projections (gemm_qkn_ok() shapes only): per-head (128 columns) RMSNorm
  // with weights qkn_wq / qkn_wk on the column ranges [qkn_q0, qkn_q1) / [qkn_k0, qkn_k1), then interleaved-pair RoPE with
  // (cos, sin) pairs qkn_rope_cs [tokens][64][2] fp32 at token qkn_pos0 + row; == rmsnorm_rope() applied afterwards
  const void* qkn_wq = nullptr; const void* qkn_wk = nullptr; const float* qkn_rope_cs = nullptr;
  int qkn_pos0 = 0, qkn_q0 = 0, qkn_q1 = 0, qkn_k0 = 0, qkn_k1 = 0; float qkn_eps = 1e-6f;
  // optional scratch for the K-sliced units (fp32 partials, 256 KiB per unit); without it few-tile GEMMs run unsplit
  void* workspace = nullptr; int64_t workspace_bytes = 0;
  // row-split weights (persistent MFMA kernel only; split_row a multiple of 256, 0 = off): tiles whose first row inside the batch
  // sample is below split_row use W2 / bias2 / gate2 / qkn_wq2 / qkn_wk2 instead of W / bias / gate / qkn_wq / qkn_wk -- the text and
  // image projections of a double block in ONE launch over the joint [text | image] rows.  W2 must lie behind W within 4 GiB
  // (one buffer descriptor): the engine allocates the pair as one tensor.
  int split_row = 0;
  const void* W2 = nullptr; const void* bias2 = nullptr; const void* gate2 = nullptr; const void* qkn_wq2 = nullptr; const void* qkn_wk2 = nullptr;
};
bool gemm_rowsplit_ok(const GemmArgs& a);                       // can this (split_row > 0) GEMM run as one launch?
void set_gemm_group_m(int gm);
void set_gemm_place(int v);
void set_gemm_waves(int v);      // 8 (default): gemm8pp_kernel; 4: gemm4w_kernel (one wave per SIMD) for unsliced bf16 launches
void set_gemm_splitk(int v);
int gemm_bf16(const GemmArgs& a, hipStream_t st);            // dispatches fast MFMA kernel or generic fallback
bool gemm_qkn_ok(const GemmArgs& a);                          // can this GEMM carry the fused q/k norm + RoPE epilogue?
bool gemm_fp8_qkn_ok(const GemmArgs& a);   // ... for the e4m3 path (gemm_fp8): an unsliced launch, same column conditions
int gemm_bf16_variant(const GemmArgs& a, int variant, hipStream_t st);  // 0 = generic, 1 = MFMA 8-phase
int gemm_bf16_f32out(const GemmArgs& a, hipStream_t st);     // C = fp32 raw accumulators [batch][M, N] (ldc, c_bstride in floats)
int mfma_peak_probe(const void* operands, int64_t operand_bytes, int fp8, int ktiles, double* flops, hipStream_t st);   // tfx_mfma_peak_probe
int gemm_fp8(const GemmArgs& a, hipStream_t st);             // persistent MFMA kernel on e4m3 operands (K % 256 == 0)
// per-row absmax quantisation bf16 -> e4m3: out = x / scale, scale = absmax / 448 (1 for an all-zero row)
// LayerNorm + modulation whose output is written as the e4m3 quantisation of the bf16 row (fp8 mode; == ln_modulate followed
// by quantize_rows_fp8, bit for bit)
int ln_modulate_fp8(const void* x, void* q8, float* q8_scale, const void* shift, const void* scale, int64_t mod_bstride,
                    int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldq, int64_t q_bstride,
                    int64_t s_bstride, float eps, hipStream_t st);
int quantize_rows_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride, float* scale,
                      int64_t s_bstride, int rows, int batch, int K, hipStream_t st);

struct AttnArgs {
  const void* q; const void* k; const void* v; void* o;   // bf16, element (b, n, h, d) at base + b*bstride + n*ld + h*128 + d
  int64_t ldq, ldk, ldv, ldo;
  int64_t q_bstride, k_bstride, v_bstride, o_bstride;
  int B, H, N;
  float scale;
  float score_bound = 0.f;   // caller's promise |scale * q . k| <= score_bound (0 = unknown): see tfx_attn_args
  void* workspace = nullptr; // optional scratch (tfx_attn_args.workspace): lets the head-dim-128 kernel deal (item, key tile) units (stream-K)
  int64_t workspace_bytes = 0;
};
int joint_attention(const AttnArgs& a, hipStream_t st);
int joint_attention_hp(const AttnArgs& a, hipStream_t st);   // half-tile software-pipelined kernel (attention_hp.hip)
int joint_attention_w16(const AttnArgs& a, hipStream_t st);  // one wave per SIMD on v_mfma_f32_16x16x32_bf16 (attention_w16.hip)
int attention_w4_release();                                    // frees every tail-split scratch buffer (tfx_release_scratch)
int attention_w4_prepare(hipStream_t st);                      // allocates the tail-split scratch of `st` (call outside stream capture)
void set_attention_tail_split(int v);                          // 0 = never split the last round's q-tiles by keys (bench knob)
int joint_attention_w4(const AttnArgs& a, hipStream_t st, int mode);   // one wave per SIMD, 64 query rows per wave (attention_w4.hip); mode (attn_w4_kernel<MODE>): 0 bookkeeping on the matrix pipe, 1 row sums on the VALU, 2 = 1 + lazy reference offset, 3 = 0 + lazy reference offset, 4 no reference at all (only when AttnArgs::score_bound is admissible, attention.hip)
void set_attention_ablation(int a);
void set_attention_use_bound(int v);
int attention_mode_counts(int64_t* counts, int n, int reset);   // tfx_attention_mode_counts
void attention_note_streamk();                                   // counts[8]: a w4 launch whose last round ran as the stream-K tail
int blend_edge(const void* a, int64_t a_bs, int64_t a_ts, int64_t a_us, void* b, int64_t b_bs, int64_t b_ts, int64_t b_us, int batch,
               int extent, int len, int C, hipStream_t st);
int gate_residual(const void* x, int64_t ldx, int64_t x_bs, const void* gate, int64_t gate_bs, const void* res, int64_t ldr, int64_t r_bs,
                  void* out, int64_t ldo, int64_t o_bs, int rows, int batch, int D, hipStream_t st);   // 0: ignore AttnArgs::score_bound
void set_attention_streamk(int v);     // 0: whole (b, h, q-tile) items only; 1 (default): stream-K dealing when it pays and a workspace was passed; 2: whenever admissible
void set_attention_persistent(int v);  // 0: one workgroup per (b, h, q-tile) item instead of one per CU
void set_attention_debug(void* p);
void set_attention_waves(int nw);  // 8 (one 512-thread workgroup per CU) or 4 (two independent 256-thread workgroups)

int groupnorm_silu_nhwc(const void* x, void* out, const void* gamma, const void* beta, float* stats_ws, int B,
                        int64_t HW, int C, int groups, float eps, bool silu, hipStream_t st);
void set_ln_prefetch(int v);   // tfx_set_option ln_prefetch
int ln_modulate_split(const void* x, void* out, const void* shift, const void* scale, const void* shift2, const void* scale2,
                      int split_row, int64_t mod_bstride, int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldo,
                      int64_t o_bstride, float eps, hipStream_t st);   // rows < split_row of a sample: (shift2, scale2)
int ln_modulate(const void* x, void* out, const void* shift, const void* scale, int64_t mod_bstride,
                int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldo,
                int64_t o_bstride, float eps, hipStream_t st);
// nn.LayerNorm with elementwise affine on [rows, D] bf16 rows: ONE bf16 rounding, as F.layer_norm (CLIP text model)
int layernorm_affine(const void* x, void* out, const void* gamma, const void* beta, int64_t rows, int D, int64_t ldx, int64_t ldo,
                     float eps, hipStream_t st);
int rmsnorm_rope(void* buf, int64_t ld, int64_t bstride, int q_off, int k_off, int H, int Ntok, int T, int B,
                 const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt, const float* cosT,
                 const float* sinT, float eps, hipStream_t st);
int sched_step(bool amo, const void* v, void* x, void* xin, int64_t ldxin, int C, int64_t rows, const float* coef,
               const int* step_ptr, int step, const float* noise, hipStream_t st);
int timestep_embedding(const float* t, void* out, int n, hipStream_t st);
int silu_bf16(const void* a, void* out, int64_t n, hipStream_t st);
int add_bf16(const void* a, const void* b, void* out, int64_t n, hipStream_t st);
int scatter_cols(const void* src, void* dst, int64_t rows, int C, int64_t ld, int col0, hipStream_t st);
int copy_rows(const void* src, int64_t sld, int64_t sbs, void* dst, int64_t dld, int64_t dbs, int rows, int cols,
              int batch, hipStream_t st);
// image / latent layout kernels around the VAE and the helpers of its mid-block attention (imageops.hip)
int any_negative(const void* x, int dtype, int64_t n, int* flag, hipStream_t st);
int prep_image(const void* img, int img_dtype, const void* mask, int mask_dtype, void* out, int B, int C, int H, int W,
               int mask_b, int norm_mode, int binarize, const int* neg_flag, hipStream_t st);
int compose_canvas(const void* glyph, const void* scene, const void* smask, void* canvas, void* cmask, int B, int gh, int gw, int sh,
                   int sw, int dir, int mask_rgb, hipStream_t st);
int rgb_to_grey(const void* rgb, void* out, int64_t n, hipStream_t st);
int resample_u8(const void* in, void* out, const int* bounds, const int* coeffs, int ksize, int64_t outer, int in_len, int out_len,
                int inner, hipStream_t st);
int pack_mask(const void* mask, int mask_dtype, void* out, int B, int H, int W, int mask_b, int binarize, int64_t ld, int col0,
              hipStream_t st);
int sample_pack(const void* moments, const void* eps, int eps_dtype, void* out, int B, int h, int w, int L, float shift,
                float scale, int64_t ld, int col0, hipStream_t st);
int unpack_latents(const void* lat, int64_t ld, void* out, int B, int h, int w, int L, float shift, float scale, hipStream_t st);
int postprocess(const void* x, void* out, int B, int H, int W, int Cs, int C, int mode, int denorm, int y0, int x0, int Hc, int Wc,
                hipStream_t st);
int transpose_bf16(const void* in, int64_t ldi, int64_t ibs, void* out, int64_t ldo, int64_t obs, int N, int C, int batch,
                   hipStream_t st);
int row_softmax(const float* s, int64_t lds, void* p, int64_t ldp, int rows, int N, float scale, hipStream_t st);
// text-encoder kernels (textenc.hip)
int attention64(const AttnArgs& a, const float* rel_bias, int causal, hipStream_t st);
int rmsnorm(const void* x, int x_dtype, int64_t ldx, const void* w, void* out, int64_t ldo, int64_t rows, int D, float eps,
            hipStream_t st);
int gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int D, int64_t vocab, hipStream_t st);
int add_into_f32(float* x, const void* y, int64_t n, int mode, hipStream_t st);
int mul_act(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int cols, int mode,
            hipStream_t st);
int select_step(const void* table, void* cur, int64_t per_step_elems, int* step_ptr, hipStream_t st);
int advance_step(int* step_ptr, hipStream_t st);

}  // namespace tfx
