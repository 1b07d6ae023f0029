// extern "C" surface of libtextflux_hip.so (declared in include/textflux_hip.h) and the host-side DiT forward
// that strings the kernels together (one FluxTransformer2DModel.forward, reference:
// diffusers/src/diffusers/models/transformers/transformer_flux.py:1028-1212).
#include "../../include/textflux_hip.h"

#include <cstring>

#include "launch.h"

using namespace tfx;

namespace {

inline hipStream_t S(tfx_stream s) { return (hipStream_t)s; }
inline const uint16_t* bf(const void* p) { return (const uint16_t*)p; }
inline uint16_t* bf(void* p) { return (uint16_t*)p; }

struct Gemm {
  GemmArgs a;
  const tfx_linear* lin;
  Gemm(const void* A, int64_t lda, int64_t abs_, const tfx_linear& l, int64_t ldw, void* C, int64_t ldc, int64_t cbs,
       int M, int N, int K, int batch) : lin(&l) {
    a = GemmArgs();
    a.A = A; a.lda = lda; a.a_bstride = abs_;
    a.W = l.w; a.ldw = ldw; a.bias = l.b;
    a.C = C; a.ldc = ldc; a.c_bstride = cbs;
    a.M = M; a.N = N; a.K = K; a.batch = batch;
    a.epilogue = EPI_BIAS;
  }
  Gemm& gelu(int from_col) { a.epilogue = EPI_BIAS_GELU; a.gelu_from_col = from_col; return *this; }
  Gemm& gate_res(const void* gate, int64_t gbs, const void* res, int64_t ldr, int64_t rbs) {
    a.epilogue = EPI_BIAS_GATE_RES; a.gate = gate; a.gate_bstride = gbs; a.res = res; a.ldr = ldr; a.r_bstride = rbs;
    return *this;
  }
  Gemm& scratch(void* ws, int64_t bytes) { a.workspace = ws; a.workspace_bytes = bytes; return *this; }
  // row-split weights: rows [0, split_row) of every sample (the text rows of the joint stream) take l2 / gate2 (/ norm weights wq2, wk2)
  Gemm& rowsplit(int split_row, const tfx_linear& l2, const void* gate2 = nullptr) {
    a.split_row = split_row; a.W2 = l2.w; a.bias2 = l2.b; a.gate2 = gate2;
    return *this;
  }
  Gemm& qknorm2(const void* wq2, const void* wk2) { a.qkn_wq2 = wq2; a.qkn_wk2 = wk2; return *this; }
  bool rowsplit_ok() const { return gemm_rowsplit_ok(a); }
  // fused per-head RMSNorm + RoPE on the k / q column ranges [0, D) / [2D, 3D) of a [k | v | q | ...] projection
  Gemm& qknorm(const void* wq, const void* wk, const float* cs, int pos0, int D, float eps) {
    a.qkn_wq = wq; a.qkn_wk = wk; a.qkn_rope_cs = cs; a.qkn_pos0 = pos0;
    a.qkn_k0 = 0; a.qkn_k1 = D; a.qkn_q0 = 2 * D; a.qkn_q1 = 3 * D; a.qkn_eps = eps;
    if (a.epilogue == EPI_BIAS) { a.epilogue = EPI_BIAS_GELU; a.gelu_from_col = 1 << 30; }   // bias only: GELU never starts
    return *this;
  }
  bool qknorm_ok(const void* wq, const void* wk, const float* cs, int pos0, int D, float eps, bool fp8 = false) const {
    Gemm t = *this;
    t.qknorm(wq, wk, cs, pos0, D, eps);
    if (t.a.split_row > 0 && !(t.a.qkn_wq2 && t.a.qkn_wk2)) return false;
    return fp8 ? gemm_fp8_qkn_ok(t.a) : gemm_qkn_ok(t.a);
  }
  int run(hipStream_t st) const { return gemm_bf16(a, st); }
  bool fp8_ready() const { return lin->w8 && lin->w8_scale && a.K % 256 == 0; }
  // fp8 linear whose activation rows were already quantised by the producer (ln_modulate_fp8)
  int run_pre(hipStream_t st, const void* q, int64_t qld, int64_t qbs, const float* qs, int64_t qs_bs) const {
    GemmArgs f = a;
    f.A = q; f.lda = qld; f.a_bstride = qbs;
    f.W = lin->w8;
    f.a_scale = qs; f.a_scale_bstride = qs_bs; f.w_scale = lin->w8_scale;
    return gemm_fp8(f, st);
  }
  // fp8 linears (desc.flags bit 2): quantise the activation rows into the q8 workspace, then the e4m3 GEMM
  int run(hipStream_t st, void* q8, float* q8_scale) const {
    if (!q8 || !lin->w8 || !lin->w8_scale || a.K % 256) return gemm_bf16(a, st);
    if (int e = quantize_rows_fp8(a.A, a.lda, a.a_bstride, q8, a.K, (int64_t)a.M * a.K, q8_scale, a.M, a.M, a.batch, a.K, st))
      return e;
    GemmArgs f = a;
    f.A = q8; f.lda = a.K; f.a_bstride = (int64_t)a.M * a.K;
    f.W = lin->w8; f.ldw = a.ldw;
    f.a_scale = q8_scale; f.a_scale_bstride = a.M; f.w_scale = lin->w8_scale;
    return gemm_fp8(f, st);
  }
};

#define TRY(x)            \
  do {                    \
    if (int _e = (x)) return _e; \
  } while (0)

static int g_fp8_fuse_qkn = 1;    // tfx_set_option fp8_fuse_qkn: 0 = fp8 projections followed by the separate q / k norm + RoPE pass (round 4; A/B knob)
static int g_ln_joint = 1;        // tfx_set_option ln_joint: 0 = the LayerNorm + modulation of a double block's text and image rows as two launches (A/B knob)
static int g_group_streams = 1;   // tfx_set_option gemm_group_streams: 0 = the text and image GEMMs of a double block as separate launches (A/B knob)

int dit_forward(const tfx_dit_desc& d, hipStream_t st) {
  const int D = d.D, H = d.H, B = d.B, Sn = d.S, T = d.T, N = Sn + T;
  if (D != H * 128) return fail("dit_forward: inner dim %d != heads %d * 128", D, H);
  if (B <= 0 || Sn <= 0 || T < 0) return fail("dit_forward: bad B/S/T");
  const int64_t D7 = 7ll * D;
  const int64_t hid_bs = (int64_t)N * D, y_bs = (int64_t)N * D7;
  uint16_t* hid = bf(d.hid);
  uint16_t* xn = bf(d.xn);
  uint16_t* y = bf(d.y);
  uint16_t* hid_img = hid + (int64_t)T * D;
  uint16_t* xn_img = xn + (int64_t)T * D;
  uint16_t* y_img = y + (int64_t)T * D7;
  const uint16_t* mod = bf(d.mod);
  const int64_t mbs = d.mod_bstride;
  const float eps = 1e-6f;
  const float att_scale = 0.08838834764831845f;  // 128^-0.5
  const int nblk = d.n_double + d.n_single;
  const int first = d.first_block < 0 ? 0 : d.first_block;
  const int last = (d.last_block < 0 || d.last_block > nblk) ? nblk : d.last_block;
  void* q8 = (d.flags & 4) ? d.q8 : nullptr;
  float* q8s = (d.flags & 4) ? d.q8_scale : nullptr;
  if ((d.flags & 4) && (!d.q8 || !d.q8_scale)) return fail("dit_forward: fp8 flag set but the q8 workspace is null");

  if (!(d.flags & 1)) {
    // x_embedder (transformer_flux.py:1086) straight into the image rows of the joint stream; text rows <- ctx0
    TRY(Gemm(d.xin, d.in_channels, (int64_t)Sn * d.in_channels, d.x_embedder, d.in_channels, hid_img, D, hid_bs, Sn, D,
             d.in_channels, B).run(st));
    if (T > 0) TRY(copy_rows(d.ctx0, D, (int64_t)T * D, hid, D, hid_bs, T, D, B, st));
  }

  // y = [k | v | q | ...]; RMSNorm + RoPE of q, k on the row range [row0, row0 + rows) as a separate pass (projections that
  // did not carry it in their epilogue); attention output overwrites q
  auto norm_rope_rows = [&](int row0, int rows, const void* nq, const void* nk) -> int {
    if (rows <= 0) return 0;
    return rmsnorm_rope(y + (int64_t)row0 * D7, D7, y_bs, 2 * D, 0, H, rows, 0, B, nq, nk, nq, nk, d.cos_tab + (int64_t)row0 * 128,
                        d.sin_tab + (int64_t)row0 * 128, eps, st);
  };
  auto attention = [&](float block_bound) -> int {   // the block's own score bound (ABI 6), else the forward-wide one
    AttnArgs a;
    a.q = y + 2 * D; a.k = y; a.v = y + D; a.o = y + 2 * D;
    a.ldq = a.ldk = a.ldv = a.ldo = D7;
    a.q_bstride = a.k_bstride = a.v_bstride = a.o_bstride = y_bs;
    a.B = B; a.H = H; a.N = N; a.scale = att_scale; a.score_bound = block_bound > 0.f ? block_bound : d.attn_score_bound;
    a.workspace = d.gemm_workspace; a.workspace_bytes = d.gemm_workspace_bytes;   // the split-K scratch is idle between GEMMs: stream-K partials
    return joint_attention(a, st);
  };

  // LayerNorm + modulation of rows [row0, row0 + rows) of every batch's joint stream, feeding ONE Linear.  bf16 mode:
  // xn (bf16) then the GEMM.  fp8 mode: the norm writes the e4m3 rows + scales straight into the q8 workspace
  // (layout [B][N][D] bytes / [B][N]) and the GEMM consumes them -- no bf16 round trip, no separate quantisation pass.
  // projection of rows [row0, row0 + rows) into y with q/k norm + RoPE: fused into the GEMM epilogue when the shape allows
  // (persistent kernel, enough tiles to fill the chip unsplit; since round 5 in fp8 mode too), else the GEMM followed by the separate pass
  const bool may_fuse = d.rope_cs != nullptr;
  auto fused_here = [&](const Gemm& gm, int row0, const void* nq, const void* nk) -> bool {
    Gemm t = gm;      // with the scratch the launch will have: a GEMM the auto path K-slices cannot carry the fused epilogue
    t.scratch(d.gemm_workspace, d.gemm_workspace_bytes);
    const bool f8 = q8 && gm.fp8_ready();
    if (f8 && !g_fp8_fuse_qkn) return false;
    return may_fuse && t.qknorm_ok(nq, nk, d.rope_cs, row0, D, eps, f8);
  };
  auto norm_gemm = [&](const uint16_t* src, int row0, int rows, const uint16_t* shift, const uint16_t* scale, Gemm gm) -> int {
    gm.scratch(d.gemm_workspace, d.gemm_workspace_bytes);
    if (q8 && gm.fp8_ready()) {
      uint8_t* q = (uint8_t*)q8 + (int64_t)row0 * D;
      float* qs = q8s + row0;
      TRY(ln_modulate_fp8(src, q, qs, shift, scale, mbs, rows, B, D, D, hid_bs, D, hid_bs, N, eps, st));
      return gm.run_pre(st, q, D, hid_bs, qs, N);
    }
    TRY(ln_modulate(src, xn + (int64_t)row0 * D, shift, scale, mbs, rows, B, D, D, hid_bs, D, hid_bs, eps, st));
    return gm.run(st, q8, q8s);
  };

  for (int blk = first; blk < last; ++blk) {
    if (blk < d.n_double) {
      // ---- FluxTransformerBlock.forward (transformer_flux.py:794-841)
      const tfx_double_block& w = d.dbl[blk];
      const uint16_t* mi = mod + (int64_t)blk * 12 * D;  // img: shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp
      const uint16_t* mt = mi + 6 * D;                   // txt: same six
      // The text and image projections of the block as ONE launch each over the joint [text | image] rows (row-split weights) when
      // the text length is a whole number of tiles and the two weight matrices sit in one allocation (the engine's loader puts
      // them there); else two launches.  bf16 mode only.
      Gemm jq(xn, D, hid_bs, w.qkv_img, D, y, D7, y_bs, N, 3 * D, D, B);
      jq.rowsplit(T, w.qkv_txt).qknorm2(w.norm_added_q, w.norm_added_k).scratch(d.gemm_workspace, d.gemm_workspace_bytes);
      Gemm jo(y + 2 * D, D7, y_bs, w.out_img, D, hid, D, hid_bs, N, D, D, B);
      jo.gate_res(mi + 2 * D, mbs, hid, D, hid_bs).rowsplit(T, w.out_txt, mt + 2 * D).scratch(d.gemm_workspace, d.gemm_workspace_bytes);
      Gemm j1(xn, D, hid_bs, w.ff1_img, D, y + 3 * D, D7, y_bs, N, 4 * D, D, B);
      j1.gelu(0).rowsplit(T, w.ff1_txt).scratch(d.gemm_workspace, d.gemm_workspace_bytes);
      Gemm j2(y + 3 * D, D7, y_bs, w.ff2_img, 4 * D, hid, D, hid_bs, N, D, 4 * D, B);
      j2.gate_res(mi + 5 * D, mbs, hid, D, hid_bs).rowsplit(T, w.ff2_txt, mt + 5 * D).scratch(d.gemm_workspace, d.gemm_workspace_bytes);
      const bool joint = g_group_streams && T > 0 && !q8 && jq.rowsplit_ok() && jo.rowsplit_ok() && j1.rowsplit_ok() && j2.rowsplit_ok();
      if (joint) {
        if (g_ln_joint) {   // round 6: both streams' LayerNorm + modulation as ONE launch over the joint rows (text rows: the second modulation)
          TRY(ln_modulate_split(hid, xn, mi, mi + D, mt, mt + D, T, mbs, N, B, D, D, hid_bs, D, hid_bs, eps, st));
        } else {
          TRY(ln_modulate(hid_img, xn_img, mi, mi + D, mbs, Sn, B, D, D, hid_bs, D, hid_bs, eps, st));
          TRY(ln_modulate(hid, xn, mt, mt + D, mbs, T, B, D, D, hid_bs, D, hid_bs, eps, st));
        }
        const bool fj = may_fuse && jq.qknorm_ok(w.norm_q, w.norm_k, d.rope_cs, 0, D, eps);
        if (fj) jq.qknorm(w.norm_q, w.norm_k, d.rope_cs, 0, D, eps);
        TRY(jq.run(st));
        if (!fj)
          TRY(rmsnorm_rope(y, D7, y_bs, 2 * D, 0, H, N, T, B, w.norm_q, w.norm_k, w.norm_added_q, w.norm_added_k, d.cos_tab, d.sin_tab, eps, st));
        TRY(attention(w.attn_score_bound));
        TRY(jo.run(st));                                                      // hidden += gate_msa * to_out(attn), both streams
        if (g_ln_joint) {
          TRY(ln_modulate_split(hid, xn, mi + 3 * D, mi + 4 * D, mt + 3 * D, mt + 4 * D, T, mbs, N, B, D, D, hid_bs, D, hid_bs, eps, st));
        } else {
          TRY(ln_modulate(hid_img, xn_img, mi + 3 * D, mi + 4 * D, mbs, Sn, B, D, D, hid_bs, D, hid_bs, eps, st));
          TRY(ln_modulate(hid, xn, mt + 3 * D, mt + 4 * D, mbs, T, B, D, D, hid_bs, D, hid_bs, eps, st));
        }
        TRY(j1.run(st));
        TRY(j2.run(st));
        continue;
      }
      {
        Gemm gi(xn_img, D, hid_bs, w.qkv_img, D, y_img, D7, y_bs, Sn, 3 * D, D, B);
        const bool fi = fused_here(gi, T, w.norm_q, w.norm_k);
        if (fi) gi.qknorm(w.norm_q, w.norm_k, d.rope_cs, T, D, eps);
        TRY(norm_gemm(hid_img, T, Sn, mi, mi + D, gi));
        if (!fi) TRY(norm_rope_rows(T, Sn, w.norm_q, w.norm_k));
        if (T > 0) {
          Gemm gt(xn, D, hid_bs, w.qkv_txt, D, y, D7, y_bs, T, 3 * D, D, B);
          const bool ft = fused_here(gt, 0, w.norm_added_q, w.norm_added_k);
          if (ft) gt.qknorm(w.norm_added_q, w.norm_added_k, d.rope_cs, 0, D, eps);
          TRY(norm_gemm(hid, 0, T, mt, mt + D, gt));
          if (!ft) TRY(norm_rope_rows(0, T, w.norm_added_q, w.norm_added_k));
        }
      }
      TRY(attention(w.attn_score_bound));
      // hidden += gate_msa * to_out(attn)   (:817-818, 830-831)
      TRY(Gemm(y_img + 2 * D, D7, y_bs, w.out_img, D, hid_img, D, hid_bs, Sn, D, D, B)
              .gate_res(mi + 2 * D, mbs, hid_img, D, hid_bs).scratch(d.gemm_workspace, d.gemm_workspace_bytes).run(st, q8, q8s));
      if (T > 0)
        TRY(Gemm(y + 2 * D, D7, y_bs, w.out_txt, D, hid, D, hid_bs, T, D, D, B)
                .gate_res(mt + 2 * D, mbs, hid, D, hid_bs).scratch(d.gemm_workspace, d.gemm_workspace_bytes).run(st, q8, q8s));
      // MLP: norm2 * (1 + scale_mlp) + shift_mlp -> ff -> gated residual (:820-826, 833-837)
      TRY(norm_gemm(hid_img, T, Sn, mi + 3 * D, mi + 4 * D,
                    Gemm(xn_img, D, hid_bs, w.ff1_img, D, y_img + 3 * D, D7, y_bs, Sn, 4 * D, D, B).gelu(0)));
      if (T > 0)
        TRY(norm_gemm(hid, 0, T, mt + 3 * D, mt + 4 * D, Gemm(xn, D, hid_bs, w.ff1_txt, D, y + 3 * D, D7, y_bs, T, 4 * D, D, B).gelu(0)));
      TRY(Gemm(y_img + 3 * D, D7, y_bs, w.ff2_img, 4 * D, hid_img, D, hid_bs, Sn, D, 4 * D, B)
              .gate_res(mi + 5 * D, mbs, hid_img, D, hid_bs).scratch(d.gemm_workspace, d.gemm_workspace_bytes).run(st, q8, q8s));
      if (T > 0)
        TRY(Gemm(y + 3 * D, D7, y_bs, w.ff2_txt, 4 * D, hid, D, hid_bs, T, D, 4 * D, B)
                .gate_res(mt + 5 * D, mbs, hid, D, hid_bs).scratch(d.gemm_workspace, d.gemm_workspace_bytes).run(st, q8, q8s));
    } else {
      // ---- FluxSingleTransformerBlock.forward (transformer_flux.py:715-739) on the joint [text | image] sequence
      const int j = blk - d.n_double;
      const tfx_single_block& w = d.sgl[j];
      const uint16_t* ms = mod + (int64_t)d.n_double * 12 * D + (int64_t)j * 3 * D;  // shift scale gate
      {
        Gemm gs = Gemm(xn, D, hid_bs, w.qkv_mlp, D, y, D7, y_bs, N, 7 * D, D, B).gelu(3 * D);
        const bool fs = fused_here(gs, 0, w.norm_q, w.norm_k);
        if (fs) gs.qknorm(w.norm_q, w.norm_k, d.rope_cs, 0, D, eps);
        TRY(norm_gemm(hid, 0, N, ms, ms + D, gs));
        if (!fs) TRY(norm_rope_rows(0, N, w.norm_q, w.norm_k));
      }
      TRY(attention(w.attn_score_bound));
      TRY(Gemm(y + 2 * D, D7, y_bs, w.proj_out, 5 * D, hid, D, hid_bs, N, D, 5 * D, B)
              .gate_res(ms + 2 * D, mbs, hid, D, hid_bs).scratch(d.gemm_workspace, d.gemm_workspace_bytes).run(st, q8, q8s));
    }
  }

  if (!(d.flags & 2)) {
    // norm_out (AdaLayerNormContinuous: chunk order scale, shift) + proj_out on the image rows (:1200-1203)
    const uint16_t* mo = mod + (int64_t)d.n_double * 12 * D + (int64_t)d.n_single * 3 * D;
    TRY(ln_modulate(hid_img, xn_img, mo + D, mo, mbs, Sn, B, D, D, hid_bs, D, hid_bs, eps, st));
    if (d.euler_gate) {
      // flow-matching Euler step in the epilogue: x' = x + bf16(dsigma * bf16(v)), in place on the latent columns of xin
      // (gate = the step's dsigma in every column, residual = output = xin[:, :, :out_channels])
      void* lat = const_cast<void*>(d.xin);
      const int64_t xbs = (int64_t)Sn * d.in_channels;
      TRY(Gemm(xn_img, D, hid_bs, d.proj_out, D, lat, d.in_channels, xbs, Sn, d.out_channels, D, B)
              .gate_res(d.euler_gate, d.euler_gate_bstride, lat, d.in_channels, xbs).run(st));
    } else {
      TRY(Gemm(xn_img, D, hid_bs, d.proj_out, D, d.out, d.out_channels, (int64_t)Sn * d.out_channels, Sn, d.out_channels,
               D, B).run(st));
    }
  }
  return 0;
}

}  // namespace

extern "C" {

const char* tfx_version(void) { return "textflux_hip 0.1 (gfx950)"; }
const char* tfx_last_error(void) { return last_error(); }
int tfx_abi_info(int32_t* out, int n) {
  const int32_t v[7] = {TFX_ABI_VERSION, (int32_t)sizeof(tfx_gemm_args), (int32_t)sizeof(tfx_attn_args), (int32_t)sizeof(tfx_dit_desc),
                        (int32_t)sizeof(tfx_step_desc), (int32_t)sizeof(tfx_double_block), (int32_t)sizeof(tfx_single_block)};
  for (int i = 0; i < 7 && i < n; ++i) out[i] = v[i];
  return 7;
}

int tfx_query_arch(char* buf, int buflen) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail("tfx_query_arch: no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("tfx_query_arch: hipGetDeviceProperties failed");
  if (buf && buflen > 0) {
    std::strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
  }
  return 0;
}

#ifdef TFX_BENCH
// bench library only (tools/gemm_phase_timers.py, gemm_shapes_power.py): the next tfx_gemm_bf16 calls carry the DiT's fused q / k
// RMSNorm + RoPE epilogue on the [k | v | q | ...] column layout, as tfx_dit_forward attaches it internally
static struct { const void* wq; const void* wk; const float* cs; int D; } g_bench_qkn = {nullptr, nullptr, nullptr, 0};
extern "C" void tfx_bench_gemm_qkn(const void* wq, const void* wk, const float* cs, int D) { g_bench_qkn = {wq, wk, cs, D}; }
#endif

int tfx_gemm_bf16(const tfx_gemm_args* g, int variant, tfx_stream stream) {
  if (!g) return fail("tfx_gemm_bf16: null args");
  GemmArgs a;
  a.A = g->A; a.lda = g->lda; a.a_bstride = g->a_bstride;
  a.W = g->W; a.ldw = g->ldw; a.bias = g->bias; a.w_bstride = g->w_bstride;
  a.C = g->C; a.ldc = g->ldc; a.c_bstride = g->c_bstride;
  a.M = g->M; a.N = g->N; a.K = g->K; a.batch = g->batch;
  a.epilogue = g->epilogue; a.gelu_from_col = g->gelu_from_col;
  a.gate = g->gate; a.gate_bstride = g->gate_bstride;
  a.res = g->res; a.ldr = g->ldr; a.r_bstride = g->r_bstride;
  a.workspace = g->workspace; a.workspace_bytes = g->workspace_bytes;
  if (!a.A || !a.W || !a.C) return fail("tfx_gemm_bf16: null matrix pointer");
#ifdef TFX_BENCH
  if (g_bench_qkn.cs) {
    const int D = g_bench_qkn.D;
    a.qkn_wq = g_bench_qkn.wq; a.qkn_wk = g_bench_qkn.wk; a.qkn_rope_cs = g_bench_qkn.cs; a.qkn_pos0 = 0;
    a.qkn_k0 = 0; a.qkn_k1 = D; a.qkn_q0 = 2 * D; a.qkn_q1 = 3 * D; a.qkn_eps = 1e-6f;
    if (!gemm_qkn_ok(a)) return fail("tfx_bench_gemm_qkn: shape cannot carry the fused epilogue");
  }
#endif
  return variant < 0 ? gemm_bf16(a, S(stream)) : gemm_bf16_variant(a, variant, S(stream));
}

int tfx_gemm_bf16_qkn(const tfx_gemm_args* g, const tfx_qkn_args* q, tfx_stream stream) {
  if (!g || !q) return fail("tfx_gemm_bf16_qkn: null args");
  if (!g->A || !g->W || !g->C) return fail("tfx_gemm_bf16_qkn: null matrix pointer");
  if (!q->norm_q || !q->norm_k || !q->rope_cs) return fail("tfx_gemm_bf16_qkn: norm weights and the rotary table are required");
  if ((uintptr_t)q->norm_q % 16 || (uintptr_t)q->norm_k % 16 || (uintptr_t)q->rope_cs % 16)
    return fail("tfx_gemm_bf16_qkn: norm weights and the rotary table must be 16-byte aligned");
  if (q->q0 < 0 || q->q1 < q->q0 || q->q1 > g->N || q->k0 < 0 || q->k1 < q->k0 || q->k1 > g->N || q->pos0 < 0)
    return fail("tfx_gemm_bf16_qkn: column ranges outside [0, N)");
  GemmArgs a;
  a.A = g->A; a.lda = g->lda; a.a_bstride = g->a_bstride;
  a.W = g->W; a.ldw = g->ldw; a.bias = g->bias; a.w_bstride = g->w_bstride;
  a.C = g->C; a.ldc = g->ldc; a.c_bstride = g->c_bstride;
  a.M = g->M; a.N = g->N; a.K = g->K; a.batch = g->batch;
  if (g->epilogue != EPI_BIAS && g->epilogue != EPI_BIAS_GELU) return fail("tfx_gemm_bf16_qkn: epilogue must be 0 (bias) or 1 (bias + GELU from a column)");
  // the fused epilogue is an instantiation of the bias + GELU kernel (norm tiles and GELU tiles are separate straight-line paths): plain
  // bias = GELU from a column beyond N, as tfx_dit_forward passes it
  a.epilogue = EPI_BIAS_GELU;
  a.gelu_from_col = g->epilogue == EPI_BIAS_GELU ? g->gelu_from_col : (g->N + 255) / 256 * 256;
  a.gate = nullptr; a.gate_bstride = 0; a.res = nullptr; a.ldr = 0; a.r_bstride = 0;
  a.workspace = g->workspace; a.workspace_bytes = g->workspace_bytes;
  a.qkn_wq = q->norm_q; a.qkn_wk = q->norm_k; a.qkn_rope_cs = q->rope_cs; a.qkn_pos0 = q->pos0;
  a.qkn_q0 = q->q0; a.qkn_q1 = q->q1; a.qkn_k0 = q->k0; a.qkn_k1 = q->k1; a.qkn_eps = q->eps;
  return gemm_bf16(a, S(stream));     // refuses shapes the fused epilogue cannot take (gemm_qkn_ok)
}

int tfx_gemm_bf16_f32(const tfx_gemm_args* g, tfx_stream stream) {
  if (!g) return fail("tfx_gemm_bf16_f32: null args");
  if (!g->A || !g->W || !g->C) return fail("tfx_gemm_bf16_f32: null matrix pointer");
  GemmArgs a;
  a.A = g->A; a.lda = g->lda; a.a_bstride = g->a_bstride;
  a.W = g->W; a.ldw = g->ldw; a.bias = g->bias; a.w_bstride = g->w_bstride;
  a.C = g->C; a.ldc = g->ldc; a.c_bstride = g->c_bstride;
  a.M = g->M; a.N = g->N; a.K = g->K; a.batch = g->batch;
  a.epilogue = g->epilogue; a.gelu_from_col = 0;
  a.gate = nullptr; a.gate_bstride = 0; a.res = nullptr; a.ldr = 0; a.r_bstride = 0;
  return gemm_bf16_f32out(a, S(stream));
}

int tfx_gemm_fp8(const tfx_gemm_args* g, const float* a_scale, int64_t a_scale_bstride, const float* w_scale,
                 tfx_stream stream) {
  if (!g) return fail("tfx_gemm_fp8: null args");
  GemmArgs a;
  a.A = g->A; a.lda = g->lda; a.a_bstride = g->a_bstride;
  a.W = g->W; a.ldw = g->ldw; a.bias = g->bias; a.w_bstride = g->w_bstride;
  a.C = g->C; a.ldc = g->ldc; a.c_bstride = g->c_bstride;
  a.M = g->M; a.N = g->N; a.K = g->K; a.batch = g->batch;
  a.epilogue = g->epilogue; a.gelu_from_col = g->gelu_from_col;
  a.gate = g->gate; a.gate_bstride = g->gate_bstride;
  a.res = g->res; a.ldr = g->ldr; a.r_bstride = g->r_bstride;
  a.a_scale = a_scale; a.a_scale_bstride = a_scale_bstride; a.w_scale = w_scale;
  a.workspace = g->workspace; a.workspace_bytes = g->workspace_bytes;
  if (!a.A || !a.W || !a.C) return fail("tfx_gemm_fp8: null matrix pointer");
  return gemm_fp8(a, S(stream));
}

int tfx_quantize_rows_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride,
                          float* scale, int64_t s_bstride, int32_t rows, int32_t batch, int32_t K, tfx_stream stream) {
  if (!x || !out || !scale) return fail("tfx_quantize_rows_fp8: null pointer");
  return quantize_rows_fp8(x, ldx, x_bstride, out, ldo, o_bstride, scale, s_bstride, rows, batch, K, S(stream));
}

int tfx_ln_modulate_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* q8, int64_t ldq, int64_t q_bstride,
                        float* q8_scale, int64_t s_bstride, const void* shift, const void* scale, int64_t mod_bstride,
                        int32_t rows_per_batch, int32_t batch, int32_t D, float eps, tfx_stream stream) {
  if (!x || !q8 || !q8_scale || !shift || !scale) return fail("tfx_ln_modulate_fp8: null pointer");
  return ln_modulate_fp8(x, q8, q8_scale, shift, scale, mod_bstride, rows_per_batch, batch, D, ldx, x_bstride, ldq,
                         q_bstride, s_bstride, eps, S(stream));
}

int tfx_ln_modulate(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride,
                    const void* shift, const void* scale, int64_t mod_bstride, int32_t rows_per_batch, int32_t batch,
                    int32_t D, float eps, tfx_stream stream) {
  if (!x || !out || !shift || !scale) return fail("tfx_ln_modulate: null pointer");
  return ln_modulate(x, out, shift, scale, mod_bstride, rows_per_batch, batch, D, ldx, x_bstride, ldo, o_bstride, eps,
                     S(stream));
}

int tfx_layernorm(const void* x, int64_t ldx, void* out, int64_t ldo, const void* gamma, const void* beta, int64_t rows,
                  int32_t D, float eps, tfx_stream stream) {
  if (!x || !out || !gamma || !beta) return fail("tfx_layernorm: null pointer");
  if (ldx % 8 || ldo % 8) return fail("tfx_layernorm: row strides must be multiples of 8 elements");
  return layernorm_affine(x, out, gamma, beta, rows, D, ldx, ldo, eps, S(stream));
}

int tfx_rmsnorm_rope(void* buf, int64_t ld, int64_t bstride, int32_t q_off, int32_t k_off, int32_t H, int32_t Ntok,
                     int32_t T, int32_t B, const void* wq_img, const void* wk_img, const void* wq_txt,
                     const void* wk_txt, const float* cos_tab, const float* sin_tab, float eps, tfx_stream stream) {
  if (!buf || !wq_img || !wk_img || !wq_txt || !wk_txt || !cos_tab || !sin_tab) return fail("tfx_rmsnorm_rope: null pointer");
  if (ld % 8 || bstride % 8 || q_off % 8 || k_off % 8) return fail("tfx_rmsnorm_rope: offsets/strides must be multiples of 8");
  return rmsnorm_rope(buf, ld, bstride, q_off, k_off, H, Ntok, T, B, wq_img, wk_img, wq_txt, wk_txt, cos_tab, sin_tab,
                      eps, S(stream));
}

int tfx_rmsnorm_rope_qk(void* buf, int64_t ld, int64_t bstride, int32_t q_off, int32_t k_off, int32_t H, int32_t Ntok,
                        int32_t T, int32_t B, const void* wq_img, const void* wk_img, const void* wq_txt,
                        const void* wk_txt, const float* cos_tab, const float* sin_tab, float eps, tfx_stream stream) {
  return tfx_rmsnorm_rope(buf, ld, bstride, q_off, k_off, H, Ntok, T, B, wq_img, wk_img, wq_txt, wk_txt, cos_tab, sin_tab, eps, stream);
}

int tfx_gate_residual(const void* x, int64_t ldx, int64_t x_bstride, const void* gate, int64_t gate_bstride, const void* res,
                      int64_t ldr, int64_t r_bstride, void* out, int64_t ldo, int64_t o_bstride, int32_t rows_per_batch,
                      int32_t batch, int32_t D, tfx_stream stream) {
  if (!x || !gate || !res || !out) return fail("tfx_gate_residual: null pointer");
  if (D <= 0 || D % 8 || (ldx | x_bstride | ldr | r_bstride | ldo | o_bstride | gate_bstride) % 8)
    return fail("tfx_gate_residual: D and every stride must be multiples of 8 elements");
  if (((uintptr_t)x | (uintptr_t)gate | (uintptr_t)res | (uintptr_t)out) % 16) return fail("tfx_gate_residual: pointers must be 16-byte aligned");
  if (rows_per_batch < 0 || batch < 0) return fail("tfx_gate_residual: negative extent");
  return gate_residual(x, ldx, x_bstride, gate, gate_bstride, res, ldr, r_bstride, out, ldo, o_bstride, rows_per_batch, batch, D, S(stream));
}

int tfx_blend_edge_nhwc(const void* a, int64_t a_bstride, int64_t a_tstride, int64_t a_ustride, void* b, int64_t b_bstride,
                        int64_t b_tstride, int64_t b_ustride, int32_t batch, int32_t extent, int32_t len, int32_t C, tfx_stream stream) {
  if (!a || !b) return fail("tfx_blend_edge_nhwc: null pointer");
  if (C <= 0 || C % 8 || (a_bstride | a_tstride | a_ustride | b_bstride | b_tstride | b_ustride) % 8)
    return fail("tfx_blend_edge_nhwc: C and every stride must be multiples of 8 elements");
  if (((uintptr_t)a | (uintptr_t)b) % 16) return fail("tfx_blend_edge_nhwc: pointers must be 16-byte aligned");
  if (batch < 0 || extent < 0 || len < 0) return fail("tfx_blend_edge_nhwc: negative extent");
  return blend_edge(a, a_bstride, a_tstride, a_ustride, b, b_bstride, b_tstride, b_ustride, batch, extent, len, C, S(stream));
}

int tfx_release_scratch(void) { return attention_w4_release(); }

int tfx_attention_mode_counts(int64_t* counts, int32_t n, int32_t reset) {
  if (!counts && n > 0) return fail("tfx_attention_mode_counts: null pointer");
  return attention_mode_counts(counts, n, reset);
}

int tfx_joint_attention(const tfx_attn_args* g, tfx_stream stream) {
  if (!g || !g->q || !g->k || !g->v || !g->o) return fail("tfx_joint_attention: null pointer");
  AttnArgs a;
  a.q = g->q; a.k = g->k; a.v = g->v; a.o = g->o;
  a.ldq = g->ldq; a.ldk = g->ldk; a.ldv = g->ldv; a.ldo = g->ldo;
  a.q_bstride = g->q_bstride; a.k_bstride = g->k_bstride; a.v_bstride = g->v_bstride; a.o_bstride = g->o_bstride;
  a.B = g->B; a.H = g->H; a.N = g->N; a.scale = g->scale; a.score_bound = g->score_bound;
  a.workspace = g->workspace; a.workspace_bytes = g->workspace ? g->workspace_bytes : 0;
  return joint_attention(a, S(stream));
}

int tfx_euler_step(const void* v, void* x, void* xin, int64_t ldxin, int32_t C, int64_t rows, const float* coef,
                   const int32_t* step_ptr, int32_t step, tfx_stream stream) {
  if (!v || !x || !coef) return fail("tfx_euler_step: null pointer");
  return sched_step(false, v, x, xin, ldxin, C, rows, coef, step_ptr, step, nullptr, S(stream));
}
int tfx_amo_step(const void* v, void* x, void* xin, int64_t ldxin, int32_t C, int64_t rows, const float* coef,
                 const int32_t* step_ptr, int32_t step, const float* noise, tfx_stream stream) {
  if (!v || !x || !coef) return fail("tfx_amo_step: null pointer");
  return sched_step(true, v, x, xin, ldxin, C, rows, coef, step_ptr, step, noise, S(stream));
}

int tfx_timestep_embedding(const float* t, void* out, int32_t n, tfx_stream stream) {
  if (!t || !out) return fail("tfx_timestep_embedding: null pointer");
  return timestep_embedding(t, out, n, S(stream));
}
int tfx_silu(const void* a, void* out, int64_t n, tfx_stream stream) {
  if (!a || !out) return fail("tfx_silu: null pointer");
  return silu_bf16(a, out, n, S(stream));
}
int tfx_add(const void* a, const void* b, void* out, int64_t n, tfx_stream stream) {
  if (!a || !b || !out) return fail("tfx_add: null pointer");
  return add_bf16(a, b, out, n, S(stream));
}
int tfx_scatter_cols(const void* src, void* dst, int64_t rows, int32_t C, int64_t ld, int32_t col0, tfx_stream stream) {
  if (!src || !dst) return fail("tfx_scatter_cols: null pointer");
  return scatter_cols(src, dst, rows, C, ld, col0, S(stream));
}
int tfx_copy_rows(const void* src, int64_t src_ld, int64_t src_bstride, void* dst, int64_t dst_ld, int64_t dst_bstride,
                  int32_t rows, int32_t cols, int32_t batch, tfx_stream stream) {
  if (!src || !dst) return fail("tfx_copy_rows: null pointer");
  return copy_rows(src, src_ld, src_bstride, dst, dst_ld, dst_bstride, rows, cols, batch, S(stream));
}
int tfx_select_step(const void* table, void* cur, int64_t per_step_elems, int32_t* step_ptr, tfx_stream stream) {
  if (!table || !cur || !step_ptr) return fail("tfx_select_step: null pointer");
  return select_step(table, cur, per_step_elems, step_ptr, S(stream));
}
int tfx_advance_step(int32_t* step_ptr, tfx_stream stream) {
  if (!step_ptr) return fail("tfx_advance_step: null pointer");
  return advance_step(step_ptr, S(stream));
}

int tfx_conv3x3_nhwc(const void* x, int32_t B, int32_t inH, int32_t inW, int32_t Cin, const void* w, const void* bias,
                     void* out, int32_t H, int32_t W, int32_t Cout, int32_t stride, int32_t up, int32_t pad_lo,
                     const void* res, const void* zero_page, int variant, tfx_stream stream) {
  if (!x || !w || !out || !zero_page) return fail("tfx_conv3x3_nhwc: null pointer");
  if (up != 1 && up != 2) return fail("tfx_conv3x3_nhwc: up must be 1 or 2");
  GemmArgs a;
  std::memset(&a, 0, sizeof(a));
  // narrow inputs (Cin = 8 / 16 / 32): K = 9 * Cin padded with zero weights to a multiple of 64
  const int K = Cin < 64 ? (9 * Cin + 63) / 64 * 64 : 9 * Cin;
  a.A = x; a.lda = Cin; a.W = w; a.ldw = K; a.bias = bias;
  a.C = out; a.ldc = Cout; a.M = B * H * W; a.N = Cout; a.K = K; a.batch = 1;
  a.epilogue = res ? EPI_BIAS_RES : EPI_BIAS;
  a.res = res; a.ldr = Cout;
  a.conv_cin = Cin; a.conv_inH = inH; a.conv_inW = inW; a.conv_H = H; a.conv_W = W;
  a.conv_stride = stride; a.conv_up_shift = up == 2 ? 1 : 0; a.conv_pad_lo = pad_lo; a.zero_page = zero_page;
  a.conv_kw = 3; a.conv_stride_x = 0; a.qkn_eps = 1e-6f;
  return variant < 0 ? gemm_bf16(a, S(stream)) : gemm_bf16_variant(a, variant, S(stream));
}

int tfx_conv3x3_pair_nhwc(const void* x, int32_t B, int32_t H, int32_t W, int32_t Cin, const void* w_pair, const void* bias_pair,
                          void* out, int32_t Cout, const void* res, const void* zero_page, tfx_stream stream) {
  if (!x || !w_pair || !out || !zero_page) return fail("tfx_conv3x3_pair_nhwc: null pointer");
  if (W % 2 || Cin % 64 || Cout % 4) return fail("tfx_conv3x3_pair_nhwc: W must be even, Cin a multiple of 64, Cout of 4");
  GemmArgs a;
  std::memset(&a, 0, sizeof(a));
  const int K = 12 * Cin;
  a.A = x; a.lda = Cin; a.W = w_pair; a.ldw = K; a.bias = bias_pair;
  a.C = out; a.ldc = 2 * Cout; a.M = B * H * (W / 2); a.N = 2 * Cout; a.K = K; a.batch = 1;
  a.epilogue = res ? EPI_BIAS_RES : EPI_BIAS;
  a.res = res; a.ldr = 2 * Cout;
  a.conv_cin = Cin; a.conv_inH = H; a.conv_inW = W; a.conv_H = H; a.conv_W = W / 2;
  a.conv_stride = 1; a.conv_up_shift = 0; a.conv_pad_lo = 1; a.zero_page = zero_page;
  a.conv_kw = 4; a.conv_stride_x = 2;
  return gemm_bf16(a, S(stream));
}

int tfx_groupnorm_nhwc(const void* x, void* out, const void* gamma, const void* beta, float* workspace, int32_t B,
                       int64_t HW, int32_t C, int32_t groups, float eps, int32_t silu, tfx_stream stream) {
  if (!x || !out || !gamma || !beta || !workspace) return fail("tfx_groupnorm_nhwc: null pointer");
  return groupnorm_silu_nhwc(x, out, gamma, beta, workspace, B, HW, C, groups, eps, silu != 0, S(stream));
}

int tfx_any_negative(const void* x, int32_t dtype, int64_t n, int32_t* flag, tfx_stream stream) {
  if (!x || !flag) return fail("tfx_any_negative: null pointer");
  return any_negative(x, dtype, n, flag, S(stream));
}
int tfx_prep_image(const void* img, int32_t img_dtype, const void* mask, int32_t mask_dtype, void* out, int32_t B, int32_t C,
                   int32_t H, int32_t W, int32_t mask_batch, int32_t norm_mode, int32_t binarize, const int32_t* neg_flag,
                   tfx_stream stream) {
  if (!img || !out) return fail("tfx_prep_image: null pointer");
  return prep_image(img, img_dtype, mask, mask_dtype, out, B, C, H, W, mask_batch, norm_mode, binarize, neg_flag, S(stream));
}
int tfx_compose_canvas(const void* glyph, const void* scene, const void* scene_mask_rgb, void* canvas, void* cmask, int32_t B,
                       int32_t gh, int32_t gw, int32_t sh, int32_t sw, int32_t direction, int32_t mask_rgb, tfx_stream stream) {
  if (!glyph || !scene || !scene_mask_rgb || !canvas || !cmask) return fail("tfx_compose_canvas: null pointer");
  return compose_canvas(glyph, scene, scene_mask_rgb, canvas, cmask, B, gh, gw, sh, sw, direction, mask_rgb, S(stream));
}
int tfx_rgb_to_grey_u8(const void* rgb, void* out, int64_t pixels, tfx_stream stream) {
  if (!rgb || !out) return fail("tfx_rgb_to_grey_u8: null pointer");
  return rgb_to_grey(rgb, out, pixels, S(stream));
}
int tfx_resample_u8(const void* in, void* out, const int32_t* bounds, const int32_t* coeffs, int32_t ksize, int64_t outer,
                    int32_t in_len, int32_t out_len, int32_t inner, tfx_stream stream) {
  if (!in || !out || !bounds || !coeffs) return fail("tfx_resample_u8: null pointer");
  return resample_u8(in, out, bounds, coeffs, ksize, outer, in_len, out_len, inner, S(stream));
}
int tfx_pack_mask(const void* mask, int32_t mask_dtype, void* out, int32_t B, int32_t H, int32_t W, int32_t mask_batch,
                  int32_t binarize, int64_t ld, int32_t col0, tfx_stream stream) {
  if (!mask || !out) return fail("tfx_pack_mask: null pointer");
  return pack_mask(mask, mask_dtype, out, B, H, W, mask_batch, binarize, ld, col0, S(stream));
}
int tfx_vae_sample_pack(const void* moments, const void* eps, int32_t eps_dtype, void* out, int32_t B, int32_t h, int32_t w,
                        int32_t L, float shift, float scale, int64_t ld, int32_t col0, tfx_stream stream) {
  if (!moments || !out) return fail("tfx_vae_sample_pack: null pointer");
  return sample_pack(moments, eps, eps_dtype, out, B, h, w, L, shift, scale, ld, col0, S(stream));
}
int tfx_unpack_latents(const void* latents, int64_t ld, void* out, int32_t B, int32_t h, int32_t w, int32_t L, float shift,
                       float scale, tfx_stream stream) {
  if (!latents || !out) return fail("tfx_unpack_latents: null pointer");
  return unpack_latents(latents, ld, out, B, h, w, L, shift, scale, S(stream));
}
int tfx_postprocess(const void* x, void* out, int32_t B, int32_t H, int32_t W, int32_t Cs, int32_t C, int32_t mode, int32_t denorm,
                    int32_t y0, int32_t x0, int32_t Hc, int32_t Wc, tfx_stream stream) {
  if (!x || !out) return fail("tfx_postprocess: null pointer");
  return postprocess(x, out, B, H, W, Cs, C, mode, denorm, y0, x0, Hc, Wc, S(stream));
}
int tfx_transpose(const void* in, int64_t ldi, int64_t in_bstride, void* out, int64_t ldo, int64_t out_bstride, int32_t N,
                  int32_t C, int32_t batch, tfx_stream stream) {
  if (!in || !out) return fail("tfx_transpose: null pointer");
  return transpose_bf16(in, ldi, in_bstride, out, ldo, out_bstride, N, C, batch, S(stream));
}
int tfx_row_softmax(const float* s, int64_t lds, void* p, int64_t ldp, int32_t rows, int32_t N, float scale, tfx_stream stream) {
  if (!s || !p) return fail("tfx_row_softmax: null pointer");
  return row_softmax(s, lds, p, ldp, rows, N, scale, S(stream));
}

int tfx_attention64(const tfx_attn_args* g, const float* rel_bias, int32_t causal, tfx_stream stream) {
  if (!g || !g->q || !g->k || !g->v || !g->o) return fail("tfx_attention64: null pointer");
  AttnArgs a;
  a.q = g->q; a.k = g->k; a.v = g->v; a.o = g->o;
  a.ldq = g->ldq; a.ldk = g->ldk; a.ldv = g->ldv; a.ldo = g->ldo;
  a.q_bstride = g->q_bstride; a.k_bstride = g->k_bstride; a.v_bstride = g->v_bstride; a.o_bstride = g->o_bstride;
  a.B = g->B; a.H = g->H; a.N = g->N; a.scale = g->scale;
  return attention64(a, rel_bias, causal, S(stream));
}
int tfx_rmsnorm(const void* x, int32_t x_dtype, int64_t ldx, const void* w, void* out, int64_t ldo, int64_t rows, int32_t D,
                float eps, tfx_stream stream) {
  if (!x || !w || !out) return fail("tfx_rmsnorm: null pointer");
  return rmsnorm(x, x_dtype, ldx, w, out, ldo, rows, D, eps, S(stream));
}
int tfx_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int32_t D, int64_t vocab, tfx_stream stream) {
  if (!table || !ids || !out) return fail("tfx_gather_rows: null pointer");
  return gather_rows(table, ids, out, n, D, vocab, S(stream));
}
int tfx_add_into_f32(float* x, const void* y, int64_t n, int32_t mode, tfx_stream stream) {
  if (!x || !y) return fail("tfx_add_into_f32: null pointer");
  return add_into_f32(x, y, n, mode, S(stream));
}
int tfx_mul_act(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int32_t cols,
                int32_t mode, tfx_stream stream) {
  if (!a || !out) return fail("tfx_mul_act: null pointer");
  return mul_act(a, lda, b, ldb, out, ldo, rows, cols, mode, S(stream));
}

int tfx_set_option(const char* name, int value) {
  if (!name) return fail("tfx_set_option: null name");
  if (!std::strcmp(name, "attention_waves")) {
    if (value == 0) { set_attention_waves(0); return 0; }     // back to the library default and its size heuristic
    if (value != 4 && value != 8 && value != 9 && value != 10 && value != 12 && value != 16 && value != 20 && !(value >= 30 && value <= 34) && value != 40)
      return fail("tfx_set_option: attention_waves must be 4, 8, 9 (128 keys per barrier), 10 / 12 (matrix-pipe softmax, 8 / 4 waves), 16 (ping-pong), 20 (half-tile pipelined), 30 .. 34 (one wave per SIMD, 32x32x16 MFMA: 30 bookkeeping on the matrix pipe -- or no reference at all when the call carries an admissible score bound -- / 31 row sums on the VALU / 32 = 31 + lazy reference offset / 33 = 30 + lazy reference offset / 34 = no reference with a score bound, else 33) or 40 (one wave per SIMD, 16x16x32 MFMA)");
#ifndef TFX_BENCH
    if (value != 30 && value != 34)
      return fail("tfx_set_option: attention_waves %d is a bench-only kernel (libtextflux_hip_bench.so, `make bench`); the product library "
                  "carries 30 (the default) and 34", value);
#endif
    set_attention_waves(value);
    return 0;
  }
  if (!std::strcmp(name, "gemm_group_m")) { set_gemm_group_m(value); return 0; }
  if (!std::strcmp(name, "attention_tail_split")) { set_attention_tail_split(value); return 0; }
  if (!std::strcmp(name, "attention_use_bound")) { set_attention_use_bound(value); return 0; }
  if (!std::strcmp(name, "attention_persistent")) { set_attention_persistent(value); return 0; }
  if (!std::strcmp(name, "attention_streamk")) { set_attention_streamk(value); return 0; }
  if (!std::strcmp(name, "gemm_place")) { set_gemm_place(value); return 0; }
  if (!std::strcmp(name, "gemm_waves")) {
#ifndef TFX_BENCH
    if (value != 8 && value != 0) return fail("tfx_set_option: gemm_waves 4 (gemm4w_kernel) is bench-only (libtextflux_hip_bench.so, `make bench`)");
#endif
    set_gemm_waves(value);
    return 0;
  }
  if (!std::strcmp(name, "gemm_group_streams")) { g_group_streams = value; return 0; }
  if (!std::strcmp(name, "ln_joint")) { g_ln_joint = value; return 0; }
  if (!std::strcmp(name, "ln_prefetch")) { set_ln_prefetch(value); return 0; }
  if (!std::strcmp(name, "fp8_fuse_qkn")) { g_fp8_fuse_qkn = value; return 0; }
  if (!std::strcmp(name, "gemm_splitk")) { set_gemm_splitk(value); return 0; }
  if (!std::strcmp(name, "attention_ablation")) { set_attention_ablation(value); return 0; }  // bench-only
  return fail("tfx_set_option: unknown option '%s'", name);
}

int tfx_debug_attention_timing(void* buf) { set_attention_debug(buf); return 0; }

int tfx_mfma_peak_probe(const void* operands, int64_t operand_bytes, int32_t fp8, int32_t ktiles, double* flops, tfx_stream stream) {
  return mfma_peak_probe(operands, operand_bytes, fp8, ktiles, flops, S(stream));
}

int tfx_prof_enable(int on) { prof_enable(on); return 0; }
int tfx_prof_collect(int kind, double* total_ms, double* total_flops, int* launches) {
  if (kind < 0 || kind > 2) return fail("tfx_prof_collect: kind must be 0 (gemm), 1 (attention) or 2 (fp8 gemm)");
  return prof_collect(kind, total_ms, total_flops, launches);
}

static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }
int tfx_workspace_layout(int32_t B, int32_t Sn, int32_t T, int32_t D, int32_t flags, int64_t* off, int64_t* gemm_ws_bytes) {
  if (B <= 0 || Sn <= 0 || T < 0 || D <= 0 || !off) return fail("tfx_workspace_layout: bad arguments");
  const int64_t N = (int64_t)Sn + T, hid = align256(B * N * D * 2), y = align256(B * N * 7 * (int64_t)D * 2);
  const bool fp8 = (flags & 4) != 0;
  const int64_t q8 = fp8 ? align256(B * N * 5 * (int64_t)D) : 0, q8s = fp8 ? align256(B * N * 4) : 0, gws = 128ll << 20;
  int64_t o = 0;
  off[0] = o; o += hid;
  off[1] = o; o += hid;
  off[2] = o; o += y;
  off[3] = fp8 ? o : -1; o += q8;
  off[4] = fp8 ? o : -1; o += q8s;
  off[5] = o;
  if (gemm_ws_bytes) *gemm_ws_bytes = gws;
  return 0;
}
int64_t tfx_workspace_bytes(int32_t B, int32_t Sn, int32_t T, int32_t D, int32_t flags) {
  int64_t off[6], gws = 0;
  if (tfx_workspace_layout(B, Sn, T, D, flags, off, &gws)) return -1;
  return off[5] + gws;
}

namespace {
struct StepGraph { hipGraph_t graph; hipGraphExec_t exec; };

int step_check(const tfx_step_desc* s) {
  if (!s) return fail("tfx_dit_step: null descriptor");
  if (!s->mod_table || !s->mod_cur || !s->step_ptr) return fail("tfx_dit_step: null pointer in descriptor");
  if (s->dit.mod != s->mod_cur) return fail("tfx_dit_step: dit.mod must point at mod_cur (the rows the step selects)");
  if (s->sampler < 0 || s->sampler > 2) return fail("tfx_dit_step: sampler must be 0 (Euler), 1 (AMO) or 2 (Euler fused into proj_out)");
  if (s->sampler == 2) {
    const char* g = (const char*)s->dit.euler_gate;
    if (!g || g < (const char*)s->mod_cur || g >= (const char*)s->mod_cur + s->mod_step_elems * 2)
      return fail("tfx_dit_step: sampler 2 needs dit.euler_gate inside mod_cur (the step's dsigma row is selected with its modulation rows)");
    return 0;
  }
  if (s->dit.euler_gate) return fail("tfx_dit_step: dit.euler_gate is set but sampler is not 2");
  if (!s->latents || !s->coef) return fail("tfx_dit_step: null pointer in descriptor");
  if (s->sampler == 1 && !s->noise) return fail("tfx_dit_step: the AMO sampler needs a noise buffer");
  if (!s->dit.out) return fail("tfx_dit_step: dit.out is null");
  return 0;
}

int step_enqueue(const tfx_step_desc& s, hipStream_t st) {
  TRY(select_step(s.mod_table, s.mod_cur, s.mod_step_elems, s.step_ptr, st));
  TRY(tfx_dit_forward(&s.dit, (tfx_stream)st));
  if (s.sampler == 2) return advance_step(s.step_ptr, st);     // the update happened in proj_out's epilogue
  const int64_t rows = (int64_t)s.dit.B * s.dit.S;
  TRY(sched_step(s.sampler == 1, s.dit.out, s.latents, const_cast<void*>(s.dit.xin), s.dit.in_channels, s.dit.out_channels, rows,
                 s.coef, s.step_ptr, 0, s.noise, st));
  return advance_step(s.step_ptr, st);
}
}  // namespace

int tfx_dit_step_run(const tfx_step_desc* s, tfx_stream stream) {
  TRY(step_check(s));
  return step_enqueue(*s, S(stream));
}

int tfx_dit_step_capture(const tfx_step_desc* s, tfx_stream stream, tfx_graph* out) {
  TRY(step_check(s));
  if (!out) return fail("tfx_dit_step_capture: null output handle");
  if (!stream) return fail("tfx_dit_step_capture: the NULL stream cannot be captured; pass a created stream");
  hipStream_t st = S(stream);
  (void)attention_w4_prepare(st);   // the attention kernel's scratch (of this stream) cannot be allocated inside a capture
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return fail("tfx_dit_step_capture: hipStreamBeginCapture: %s", hipGetErrorString(e));
  const int rc = step_enqueue(*s, st);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }    // last error already set by the failing launcher
  if (e != hipSuccess || !g) return fail("tfx_dit_step_capture: hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return fail("tfx_dit_step_capture: hipGraphInstantiate: %s", hipGetErrorString(e)); }
  *out = new StepGraph{g, x};
  return 0;
}

int tfx_dit_step_replay(tfx_graph graph, tfx_stream stream) {
  if (!graph) return fail("tfx_dit_step_replay: null graph");
  const hipError_t e = hipGraphLaunch(((StepGraph*)graph)->exec, S(stream));
  if (e != hipSuccess) return fail("tfx_dit_step_replay: hipGraphLaunch: %s", hipGetErrorString(e));
  return 0;
}

int tfx_graph_destroy(tfx_graph graph) {
  if (!graph) return 0;
  StepGraph* g = (StepGraph*)graph;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

int tfx_dit_forward(const tfx_dit_desc* d, tfx_stream stream) {
  if (!d) return fail("tfx_dit_forward: null descriptor");
  if (!d->xin || !d->mod || !d->hid || !d->xn || !d->y || !d->out || !d->cos_tab || !d->sin_tab)
    return fail("tfx_dit_forward: null buffer in descriptor");
  if ((d->n_double > 0 && !d->dbl) || (d->n_single > 0 && !d->sgl)) return fail("tfx_dit_forward: null block table");
  if (d->T > 0 && !d->ctx0) return fail("tfx_dit_forward: ctx0 is null");
  return dit_forward(*d, S(stream));
}

}  // extern "C"
