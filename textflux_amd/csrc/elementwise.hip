// HBM-bound kernels of the DiT step: fused LayerNorm+AdaLN modulation, fused per-head RMSNorm+RoPE,
// flow-matching Euler / AMO scheduler steps, sinusoidal timestep embedding, SiLU, bf16 add.
// All loads/stores are 16 B per lane (bf16x8); math is fp32; every bf16 rounding point of the
// reference's unfused op chain is reproduced so results track the bf16 reference.
#include <algorithm>

#include "common.h"
#include "launch.h"

namespace tfx {

// ---------------------------------------------------------------------------------------------
// LayerNorm(no affine, eps) * (1 + scale[b]) + shift[b]     (reference: AdaLayerNormZero /
// AdaLayerNormZeroSingle / AdaLayerNormContinuous, D/models/normalization.py:170, :202, :365 and the
// norm2 path of FluxTransformerBlock, transformer_flux.py:820-821).
// One wave per token row, row held in registers (<= NCH*8*64 elements), two-pass statistics.
// F8: instead of the bf16 row, write its per-row absmax e4m3 quantisation (quant_rows_fp8_kernel of the SAME bf16 values,
// bit for bit) to q8 / q8_scale -- the fp8 mode's fused LayerNorm -> GEMM operand path.
// AFFINE: nn.LayerNorm with elementwise affine instead -- out = bf16((x - mean) * rstd * gamma + beta), fp32 throughout and
// ONE rounding, which is what F.layer_norm does on bf16 tensors (scale = gamma, shift = beta, mod_bstride = 0; the CLIP text
// model's LayerNorms, transformers models/clip/modeling_clip.py CLIPEncoderLayer / final_layer_norm).
// PREF (round 6): the row's chunks are requested together (rounds 1-5 loaded each under its own `if (chunk < nchunk)`: hipcc waited for every
// load before the next -- six dependent memory round trips per row, invisible at batch 8 behind 20 waves per CU and the whole 8.9 us of
// the kernel at batch 1), and with PREF the modulation rows, which do not depend on the statistics, right behind them.
template <int NCH, bool F8 = false, bool AFFINE = false, bool PREF = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ shift,
                                                          const bf16_t* __restrict__ scale, int64_t mod_bstride,
                                                          int rows_per_batch, int64_t rows, int D, int64_t ldx,
                                                          int64_t x_bstride, int64_t ldo, int64_t o_bstride, float eps,
                                                          uint8_t* __restrict__ q8 = nullptr, float* __restrict__ q8_scale = nullptr,
                                                          int64_t s_bstride = 0, int split_row = 0,
                                                          const bf16_t* __restrict__ shift2 = nullptr, const bf16_t* __restrict__ scale2 = nullptr) {
  // split_row > 0 (round 6): rows below split_row of every batch sample -- the TEXT rows of the joint [text | image] stream -- take the
  // second modulation (shift2 / scale2): the two LayerNorm + modulation launches of a double block's streams as one (a wave-uniform select)
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = (int)(row / rows_per_batch);
  const int r = (int)(row - (int64_t)b * rows_per_batch);
  const bf16_t* xr = x + b * x_bstride + r * ldx;
  bf16_t* orow = out + b * o_bstride + r * ldo;
  const int nchunk = D >> 3;
  float v[NCH][8];
  float sum = 0.f;
  const bool second = r < split_row;
  const bf16_t* sh = (second ? shift2 : shift) + b * mod_bstride;
  const bf16_t* sc = (second ? scale2 : scale) + b * mod_bstride;
  u32x4 raw[NCH], scr[PREF ? NCH : 1], shr[PREF ? NCH : 1];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {     // clamped, unconditional: all requests of the row in flight at once (a chunk beyond D re-reads the last one)
    const int chc = min(lane + c * 64, nchunk - 1);
    raw[c] = *reinterpret_cast<const u32x4*>(xr + chc * 8);
  }
  if constexpr (PREF) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int chc = min(lane + c * 64, nchunk - 1);
      scr[c] = *reinterpret_cast<const u32x4*>(sc + chc * 8);
      shr[c] = *reinterpret_cast<const u32x4*>(sh + chc * 8);
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      unpack8(raw[c], v[c]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[c][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (lane + c * 64 < nchunk) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[c][i] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lane + c * 64;
    if (ch < nchunk) {
      float s8[8], h8[8], o8[8];
      if constexpr (PREF) {
        unpack8(scr[c], s8);
        unpack8(shr[c], h8);
      } else {
        unpack8(*reinterpret_cast<const u32x4*>(sc + ch * 8), s8);
        unpack8(*reinterpret_cast<const u32x4*>(sh + ch * 8), h8);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (AFFINE) {
          o8[i] = (v[c][i] - mean) * rstd * s8[i] + h8[i];    // rounded once, by pack8
          continue;
        }
        const float xn = round_bf((v[c][i] - mean) * rstd);   // F.layer_norm output, bf16
        const float t = round_bf(1.0f + s8[i]);               // (1 + scale), bf16
        o8[i] = round_bf(xn * t) + h8[i];                     // product rounded, sum rounded by pack8
      }
      if (F8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[c][i] = round_bf(o8[i]);   // the bf16 row the unfused path would have stored
      } else {
        *reinterpret_cast<u32x4*>(orow + ch * 8) = pack8(o8);
      }
    }
  }
  if (F8) {
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (lane + c * 64 < nchunk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[c][i]));
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    const float qs = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
    if (lane == 0) q8_scale[b * s_bstride + r] = qs;
    uint8_t* qrow = q8 + b * o_bstride + r * ldo;   // ldo / o_bstride count bytes of the e4m3 rows here
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fminf(fmaxf(__fdiv_rn(v[c][i], qs), -448.f), 448.f);
        uint32_t w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        *reinterpret_cast<u32x2*>(qrow + ch * 8) = u32x2{w0, w1};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Per-head RMSNorm (fp32 variance, eps) * weight, then interleaved-pair RoPE, in place on the q and k
// column ranges of a fused projection buffer [B, Ntok, ld].  Rows < T (text tokens) use the
// norm_added_{q,k} weights, rows >= T the norm_{q,k} weights (FluxAttnProcessor2_0,
// D/models/attention_processor.py:2001-2037; RMSNorm normalization.py:534-549; apply_rotary_emb
// embeddings.py:899-918).  16 lanes per 128-wide head row, 8 elements (one 16-B chunk) per lane.
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16_t* __restrict__ buf, int64_t ld, int64_t bstride,
                                                           int q_off, int k_off, int H, int Ntok, int T, int B,
                                                           const bf16_t* __restrict__ wq_img,
                                                           const bf16_t* __restrict__ wk_img,
                                                           const bf16_t* __restrict__ wq_txt,
                                                           const bf16_t* __restrict__ wk_txt,
                                                           const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, float eps) {
  // 16 lanes per token, HSPLIT lane groups share a token's heads: each lane keeps the token's cos / sin for its 8
  // positions and both norm weights in registers and walks over its share of the 2H head rows (q heads, then k heads),
  // so the rotary table is read once per token instead of once per head row (it was 2/3 of this kernel's L2 traffic).
  constexpr int HSPLIT = 4;
  const int sub = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int64_t tokrow = grp / HSPLIT;
  const int part = (int)(grp - tokrow * HSPLIT);
  if (tokrow >= (int64_t)B * Ntok) return;
  const int b = (int)(tokrow / Ntok);
  const int n = (int)(tokrow - (int64_t)b * Ntok);
  const float4* c4 = reinterpret_cast<const float4*>(cosT + (int64_t)n * 128 + sub * 8);
  const float4* s4 = reinterpret_cast<const float4*>(sinT + (int64_t)n * 128 + sub * 8);
  const float4 ca = c4[0], cb = c4[1], sa = s4[0], sb = s4[1];
  const float cs[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
  const float sn[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
  float wq[8], wk[8];
  unpack8(*reinterpret_cast<const u32x4*>(((n < T) ? wq_txt : wq_img) + sub * 8), wq);
  unpack8(*reinterpret_cast<const u32x4*>(((n < T) ? wk_txt : wk_img) + sub * 8), wk);
  bf16_t* row = buf + b * bstride + (int64_t)n * ld + sub * 8;
  const int per = (2 * H + HSPLIT - 1) / HSPLIT;
  const int hr0 = part * per, hr_end = min(2 * H, (part + 1) * per);
  constexpr int CH = 6;   // head rows requested together: in place, so hipcc will not move a load above an earlier store
  for (int base = hr0; base < hr_end; base += CH) {
    u32x4 raw[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int hr = min(base + c, hr_end - 1);
      raw[c] = *reinterpret_cast<const u32x4*>(row + (hr >= H ? k_off + (hr - H) * 128 : q_off + hr * 128));
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int hr = base + c;
      if (hr >= hr_end) break;
      const bool is_k = hr >= H;
      float x[8];
      unpack8(raw[c], x);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
      float y[8], o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = round_bf(round_bf(x[i] * r) * (is_k ? wk[i] : wq[i]));  // .to(bf16) then * weight (bf16)
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        o[i] = y[i] * cs[i] + (-y[i + 1]) * sn[i];
        o[i + 1] = y[i + 1] * cs[i + 1] + y[i] * sn[i + 1];
      }
      *reinterpret_cast<u32x4*>(row + (is_k ? k_off + (hr - H) * 128 : q_off + hr * 128)) = pack8(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Scheduler steps on packed latents [rows, C] (C = 64): results are written both to the bf16 latent
// state and, when xin != nullptr, into columns [0, C) of the next step's x_embedder input [rows, ldxin]
// (this replaces the per-step torch.cat of pipeline_flux_fill.py:2085).
//   Euler  (scheduling_flow_match_euler_discrete.py:322-330):  x' = bf16( f32(x) + bf16(dsigma * v) )
//   AMO    (scheduling_stochastic_rf_discrete_overshot.py:340-361, attn_map None):
//            x' = bf16( (f32(x) + bf16(dt * (-v))) * a + eps * b )
// coef points at {dsigma} (Euler) or {dt_over, a, b} (AMO) per step; step index read from *step_ptr when
// given (device-side, so that one captured graph serves every step), else `step`.
template <bool AMO>
__global__ __launch_bounds__(256) void sched_step_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ x,
                                                         bf16_t* __restrict__ xin, int64_t ldxin, int C,
                                                         int64_t nchunks, const float* __restrict__ coef,
                                                         const int* __restrict__ step_ptr, int step,
                                                         const float* __restrict__ noise) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int s = step_ptr ? *step_ptr : step;
  float vv[8], xx[8], o[8];
  unpack8(*reinterpret_cast<const u32x4*>(v + i * 8), vv);
  unpack8(*reinterpret_cast<const u32x4*>(x + i * 8), xx);
  if (AMO) {
    const float dt = coef[3 * s], a = coef[3 * s + 1], bcoef = coef[3 * s + 2];
    const float4 n0 = reinterpret_cast<const float4*>(noise + i * 8)[0];
    const float4 n1 = reinterpret_cast<const float4*>(noise + i * 8)[1];
    const float e[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j)  // explicit _rn ops: no FMA contraction, bit-identical to the unfused torch chain
      o[j] = __fadd_rn(__fmul_rn(__fadd_rn(xx[j], round_bf(__fmul_rn(dt, -vv[j]))), a), __fmul_rn(e[j], bcoef));
  } else {
    const float ds = coef[s];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = __fadd_rn(xx[j], round_bf(__fmul_rn(ds, vv[j])));
  }
  const u32x4 packed = pack8(o);
  *reinterpret_cast<u32x4*>(x + i * 8) = packed;
  if (xin) {
    const int64_t e0 = i * 8;
    const int64_t row = e0 / C;
    const int col = (int)(e0 - row * C);
    *reinterpret_cast<u32x4*>(xin + row * ldxin + col) = packed;
  }
}

// Sinusoidal timestep embedding, flip_sin_to_cos=True, downscale_freq_shift=0, dim 256, fp32 -> bf16
// (get_timestep_embedding, D/models/embeddings.py:27-78 as configured at :1322).  out[n, 256] = cos | sin.
__global__ void timestep_embedding_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int n) {
  const int i = blockIdx.x;
  const int j = threadIdx.x;  // 0..127
  if (i >= n) return;
  const float freq = expf(-9.210340371976184f * (float)j / 128.0f);  // -ln(10000) * j / half_dim
  const float ang = t[i] * freq;
  out[i * 256 + j] = f2bf(cosf(ang));
  out[i * 256 + 128 + j] = f2bf(sinf(ang));
}

// mode 0: out = silu(a);  mode 1: out = a + b.   bf16 in/out, fp32 math.
__global__ __launch_bounds__(256) void unary_binary_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                           bf16_t* __restrict__ out, int64_t n, int mode) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = bf2f(a[i]);
  out[i] = f2bf(mode == 0 ? x / (1.0f + expf(-x)) : x + bf2f(b[i]));
}

// Copies rows of a [rows, C] bf16 matrix into columns [col0, col0+C) of a wider [rows, ld] matrix
// (initial fill of the x_embedder input: latents -> cols 0..63, masked_image_latents -> cols 64..383).
__global__ __launch_bounds__(256) void scatter_cols_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                           int64_t rows, int C, int64_t ld, int col0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cpr = C >> 3;
  if (i >= rows * cpr) return;
  const int64_t row = i / cpr;
  const int c = (int)(i - row * cpr);
  *reinterpret_cast<u32x4*>(dst + row * ld + col0 + c * 8) = *reinterpret_cast<const u32x4*>(src + row * C + c * 8);
}

// dst[b][r, 0:cols] = src[b][r, 0:cols], 16 bytes per thread.
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, int64_t sld, int64_t sbs,
                                                        bf16_t* __restrict__ dst, int64_t dld, int64_t dbs, int rows,
                                                        int cpr, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cpr);
  const int64_t rr = i / cpr;
  const int r = (int)(rr % rows);
  const int b = (int)(rr / rows);
  *reinterpret_cast<u32x4*>(dst + b * dbs + (int64_t)r * dld + c * 8) =
      *reinterpret_cast<const u32x4*>(src + b * sbs + (int64_t)r * sld + c * 8);
}

// out[b][r, :] = res[b][r, :] + bf16(gate[b][:] * x[b][r, :])  (tfx_gate_residual: `hidden + gate.unsqueeze(1) * attn_output`,
// transformer_flux.py:733-735, 817-818, 824-826; two bf16 ops in the reference, so the product is rounded before the add -- the
// arithmetic of the GEMM's EPI_BIAS_GATE_RES epilogue on an already rounded Linear output).  HBM-bound: 16 bytes per lane and
// stream, 3 * rows * D * 2 bytes of traffic (the gate row stays in L2).
__global__ __launch_bounds__(256) void gate_residual_kernel(const bf16_t* x, int64_t ldx, int64_t xbs,
                                                            const bf16_t* __restrict__ gate, int64_t gbs, const bf16_t* res, int64_t ldr,
                                                            int64_t rbs, bf16_t* out, int64_t ldo, int64_t obs, int rows, int cpr,
                                                            int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cpr);
  const int64_t rr = i / cpr;
  const int r = (int)(rr % rows);
  const int b = (int)(rr / rows);
  const u32x4 xv = *reinterpret_cast<const u32x4*>(x + b * xbs + (int64_t)r * ldx + c * 8);
  const u32x4 gv = *reinterpret_cast<const u32x4*>(gate + b * gbs + c * 8);
  const u32x4 rv = *reinterpret_cast<const u32x4*>(res + b * rbs + (int64_t)r * ldr + c * 8);
  u32x4 ov;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t xw = xv[k], gw = gv[k], rw = rv[k];
    const float lo = bf2f((bf16_t)(rw & 0xffff)) + round_bf(bf2f((bf16_t)(gw & 0xffff)) * bf2f((bf16_t)(xw & 0xffff)));
    const float hi = bf2f((bf16_t)(rw >> 16)) + round_bf(bf2f((bf16_t)(gw >> 16)) * bf2f((bf16_t)(xw >> 16)));
    ov[k] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
  }
  *reinterpret_cast<u32x4*>(out + b * obs + (int64_t)r * ldo + c * 8) = ov;
}

// Tile seam blend of the tiled VAE (AutoencoderKL.blend_v / blend_h, D/models/autoencoders/autoencoder_kl.py:334-344): for t in
// [0, extent), u in [0, len):  b[., t, u, :] = a[., t, u, :] * (1 - t / extent) + b[., t, u, :] * (t / extent), IN PLACE on b.  (t, u) =
// (row, column) for the vertical blend, (column, row) for the horizontal one: the caller passes the strides and points `a` at the
// first of its last `extent` rows / columns.  Rounding as the reference's bf16 tensor ops: python-float weights in fp32, each
// product rounded to bf16, the sum rounded.  NHWC, C % 8 == 0, 16 bytes per lane; the seams are a sliver of the image.
__global__ __launch_bounds__(256) void blend_edge_kernel(const bf16_t* __restrict__ a, int64_t a_bs, int64_t a_ts, int64_t a_us,
                                                         bf16_t* __restrict__ b, int64_t b_bs, int64_t b_ts, int64_t b_us, int extent,
                                                         int len, int cpr, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cpr);
  int64_t r = i / cpr;
  const int u = (int)(r % len);
  r /= len;
  const int t = (int)(r % extent);
  const int bi = (int)(r / extent);
  const float wb = (float)((double)t / (double)extent), wa = (float)(1.0 - (double)t / (double)extent);
  const bf16_t* ap = a + bi * a_bs + (int64_t)t * a_ts + (int64_t)u * a_us + c * 8;
  bf16_t* bp = b + bi * b_bs + (int64_t)t * b_ts + (int64_t)u * b_us + c * 8;
  float fa[8], fb[8], o[8];
  unpack8(*reinterpret_cast<const u32x4*>(ap), fa);
  unpack8(*reinterpret_cast<const u32x4*>(bp), fb);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = round_bf(fa[e] * wa) + round_bf(fb[e] * wb);
  *reinterpret_cast<u32x4*>(bp) = pack8(o);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(groups, eps, affine) [+ SiLU] on NHWC activations x [B, HW, C] (VAE blocks: ResnetBlock2D norm1/norm2,
// D/models/resnet.py:327-366; conv_norm_out, D/models/autoencoders/vae.py:191-193, 352-354; mid-block attention
// group_norm, D/models/attention_processor.py:2824).  Three kernels: per-chunk partial sums (deterministic, no
// atomics across blocks) -> per-(batch, group) mean / rstd -> normalise (+SiLU), 16 B per lane.
constexpr int GN_CHUNK_PIX = 1024;

__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part,
                                                         int64_t HW, int C, int groups, int nchunk) {
  // per-thread partials -> LDS -> one thread per (group, statistic) adds them in thread order: bit-reproducible (shared
  // atomics would add in arrival order and make the whole VAE vary from run to run in the last bits)
  __shared__ float tp[256][4];
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
  const int cpr = C >> 3;                       // 16-byte chunks per pixel
  const int cpg = C / groups;                   // channels per group: 4, 8 or 16
  const int64_t p0 = (int64_t)ch * GN_CHUNK_PIX;
  const int64_t p1 = min(HW, p0 + GN_CHUNK_PIX);
  const int c8 = tid % cpr;
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;  // two 4-channel halves of this lane's chunk
  for (int64_t p = p0 + tid / cpr; p < p1; p += 256 / cpr) {
    float v[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + ((int64_t)b * HW + p) * C + c8 * 8), v);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s0 += v[i]; q0 += v[i] * v[i]; s1 += v[4 + i]; q1 += v[4 + i] * v[4 + i]; }
  }
  tp[tid][0] = s0; tp[tid][1] = q0; tp[tid][2] = s1; tp[tid][3] = q1;
  __syncthreads();
  if (tid < 2 * groups) {
    const int g = tid >> 1, which = tid & 1;
    float acc = 0.f;
    for (int t = 0; t < 256; ++t) {
      const int cc = (t % cpr) * 8;
      if (cc / cpg == g) acc += tp[t][which];
      if ((cc + 4) / cpg == g) acc += tp[t][2 + which];
    }
    part[(((int64_t)b * nchunk + ch) * groups + g) * 2 + which] = acc;
  }
}

// One block of 256 threads per sample: thread t sums the chunks t / 32, t / 32 + 8, ... of group-slot t % 32 (and t % 32 + 32 when there are
// more than 32 groups) in double, the eight partial sums of a group are added in a fixed order.  (Until round 6: one thread per group walking
// all nchunk partials one dependent, strided load at a time -- 104 us on average, 410 us at 1024 x 1024, for a few KiB of input.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stat, int nchunk, int groups,
                                                          float inv_count, float eps) {
  __shared__ double ps[8][64][2];
  const int b = blockIdx.x, slot = threadIdx.x & 31, lane8 = threadIdx.x >> 5;
  for (int g = slot; g < 64; g += 32) {
    double s = 0.0, q = 0.0;
    if (g < groups)
      for (int c = lane8; c < nchunk; c += 8) {
        const float2 v = *reinterpret_cast<const float2*>(part + (((int64_t)b * nchunk + c) * groups + g) * 2);
        s += v.x;
        q += v.y;
      }
    ps[lane8][g][0] = s;
    ps[lane8][g][1] = q;
  }
  __syncthreads();
  const int g = threadIdx.x;
  if (g >= groups) return;
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s += ps[k][g][0]; q += ps[k][g][1]; }
  const double mean = s * inv_count, var = q * inv_count - mean * mean;
  stat[(b * groups + g) * 2] = (float)mean;
  stat[(b * groups + g) * 2 + 1] = rsqrtf((float)(var > 0 ? var : 0) + eps);
}

// grid (ceil(chunks per sample / (256 * GN_APPLY_ITER)), B).  C / 8 divides 256 (groupnorm_silu_nhwc checks it), so a thread keeps ONE 16-byte
// channel chunk for all its pixels: gamma, beta and the statistics of the (at most two: channels per group is a multiple of 4) groups its
// eight channels lie in are loaded once, and there is no per-element division.  (Until round 6 every lane derived pixel, sample and group
// from a flat 64-bit index with runtime divisions and loaded the statistics per element: 3.2 TB/s at 1024 x 1024 x 256.)  Same arithmetic
// per element, bit-identical output.
constexpr int GN_APPLY_ITER = 4;
template <bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       const float* __restrict__ stat, int64_t chunks_per_sample, int C, int groups) {
  const int b = blockIdx.y, cpr = C >> 3, cpg = C / groups;
  const int c8 = threadIdx.x & (cpr - 1);
  float ga[8], be[8];
  unpack8(*reinterpret_cast<const u32x4*>(gamma + c8 * 8), ga);
  unpack8(*reinterpret_cast<const u32x4*>(beta + c8 * 8), be);
  const int g0 = (c8 * 8) / cpg, g1 = (c8 * 8 + 4) / cpg;
  const float2 st0 = *reinterpret_cast<const float2*>(stat + (b * groups + g0) * 2);
  const float2 st1 = *reinterpret_cast<const float2*>(stat + (b * groups + g1) * 2);
  const int64_t base = (int64_t)b * chunks_per_sample;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * GN_APPLY_ITER) + threadIdx.x;
  u32x4 raw[GN_APPLY_ITER];
#pragma unroll
  for (int k = 0; k < GN_APPLY_ITER; ++k) {
    const int64_t i = i0 + k * 256;
    raw[k] = *reinterpret_cast<const u32x4*>(x + (base + (i < chunks_per_sample ? i : chunks_per_sample - 1)) * 8);
  }
#pragma unroll
  for (int k = 0; k < GN_APPLY_ITER; ++k) {
    const int64_t i = i0 + k * 256;
    if (i >= chunks_per_sample) break;
    float v[8], o[8];
    unpack8(raw[k], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mean = e < 4 ? st0.x : st1.x, rstd = e < 4 ? st0.y : st1.y;
      float y = round_bf((v[e] - mean) * rstd * ga[e] + be[e]);   // F.group_norm output in bf16
      if (SILU) y = y / (1.0f + __expf(-y));
      o[e] = y;
    }
    *reinterpret_cast<u32x4*>(out + (base + i) * 8) = pack8(o);
  }
}

// Device-side step cursor for graph replay: cur_mod[b, :] = mod_table[*step, b, :]; optionally ++*step.
__global__ __launch_bounds__(256) void select_step_kernel(const bf16_t* __restrict__ table, bf16_t* __restrict__ cur,
                                                          int64_t per_step_chunks, int* step_ptr, int advance) {
  const int s = *step_ptr;
  const u32x4* src = reinterpret_cast<const u32x4*>(table) + (int64_t)s * per_step_chunks;
  u32x4* dst = reinterpret_cast<u32x4*>(cur);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_step_chunks; i += (int64_t)gridDim.x * 256)
    dst[i] = src[i];
  (void)advance;
}
__global__ void advance_step_kernel(int* step_ptr) { *step_ptr += 1; }

// ---------------------------------------------------------------------------------------------
// tfx_set_option ln_prefetch: 0 never, 1 always, 2 (default) by size -- the modulation rows requested up front cost 48 more registers per
// lane (occupancy 6 -> 3 waves per SIMD): free where the kernel is latency-bound (few rows), measured per size below
static int g_ln_prefetch = 2;
void set_ln_prefetch(int v) { g_ln_prefetch = v; }
static bool ln_prefetch_for(int64_t rows) { return g_ln_prefetch == 1 || (g_ln_prefetch == 2 && rows <= 16384); }
int ln_modulate(const void* x, void* out, const void* shift, const void* scale, int64_t mod_bstride,
                int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldo,
                int64_t o_bstride, float eps, hipStream_t st) {
  if (D % 8 || D > 6 * 512) return fail("ln_modulate: D must be a multiple of 8 and <= 3072");
  const int64_t rows = (int64_t)rows_per_batch * batch;
  if (rows == 0) return 0;
  if (ln_prefetch_for(rows))
    ln_modulate_kernel<6, false, false, true><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
        (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)shift, (const bf16_t*)scale, mod_bstride, rows_per_batch, rows, D,
        ldx, x_bstride, ldo, o_bstride, eps);
  else
    ln_modulate_kernel<6><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
        (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)shift, (const bf16_t*)scale, mod_bstride, rows_per_batch, rows, D,
        ldx, x_bstride, ldo, o_bstride, eps);
  return check_launch("ln_modulate");
}

// The LayerNorm + modulation of BOTH streams of a double block in one launch over the joint [text | image] rows: rows < split_row of a
// sample take (shift2, scale2) -- norm1_context / the text stream's norm2 --, the others (shift, scale).  Same kernel, same arithmetic per
// row: bit-identical to the two launches it replaces; at batch 1 it removes a 6 us latency-bound launch per pair (38 per step).
int ln_modulate_split(const void* x, void* out, const void* shift, const void* scale, const void* shift2, const void* scale2,
                      int split_row, int64_t mod_bstride, int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldo,
                      int64_t o_bstride, float eps, hipStream_t st) {
  if (D % 8 || D > 6 * 512) return fail("ln_modulate: D must be a multiple of 8 and <= 3072");
  if (split_row < 0 || split_row > rows_per_batch || (split_row > 0 && (!shift2 || !scale2))) return fail("ln_modulate_split: split_row / second modulation");
  const int64_t rows = (int64_t)rows_per_batch * batch;
  if (rows == 0) return 0;
  if (ln_prefetch_for(rows))
    ln_modulate_kernel<6, false, false, true><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
        (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)shift, (const bf16_t*)scale, mod_bstride, rows_per_batch, rows, D,
        ldx, x_bstride, ldo, o_bstride, eps, nullptr, nullptr, 0, split_row, (const bf16_t*)shift2, (const bf16_t*)scale2);
  else
    ln_modulate_kernel<6><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
        (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)shift, (const bf16_t*)scale, mod_bstride, rows_per_batch, rows, D,
        ldx, x_bstride, ldo, o_bstride, eps, nullptr, nullptr, 0, split_row, (const bf16_t*)shift2, (const bf16_t*)scale2);
  return check_launch("ln_modulate_split");
}

int layernorm_affine(const void* x, void* out, const void* gamma, const void* beta, int64_t rows, int D, int64_t ldx, int64_t ldo,
                     float eps, hipStream_t st) {
  if (D % 8 || D > 6 * 512) return fail("layernorm: D must be a multiple of 8 and <= 3072");
  if (rows <= 0) return 0;
  ln_modulate_kernel<6, false, true><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
      (const bf16_t*)x, (bf16_t*)out, (const bf16_t*)beta, (const bf16_t*)gamma, 0, (int)std::min<int64_t>(rows, 1 << 30), rows, D,
      ldx, 0, ldo, 0, eps);
  return check_launch("layernorm");
}

int ln_modulate_fp8(const void* x, void* q8, float* q8_scale, const void* shift, const void* scale, int64_t mod_bstride,
                    int rows_per_batch, int batch, int D, int64_t ldx, int64_t x_bstride, int64_t ldq, int64_t q_bstride,
                    int64_t s_bstride, float eps, hipStream_t st) {
  if (D % 8 || D > 6 * 512) return fail("ln_modulate_fp8: D must be a multiple of 8 and <= 3072");
  if (ldq % 8 || q_bstride % 8 || (uintptr_t)q8 % 8) return fail("ln_modulate_fp8: e4m3 rows must be 8-byte aligned");
  const int64_t rows = (int64_t)rows_per_batch * batch;
  if (rows == 0) return 0;
  ln_modulate_kernel<6, true><<<dim3((unsigned)((rows + 3) / 4)), 256, 0, st>>>(
      (const bf16_t*)x, nullptr, (const bf16_t*)shift, (const bf16_t*)scale, mod_bstride, rows_per_batch, rows, D, ldx,
      x_bstride, ldq, q_bstride, eps, (uint8_t*)q8, q8_scale, s_bstride);
  return check_launch("ln_modulate_fp8");
}

int rmsnorm_rope(void* buf, int64_t ld, int64_t bstride, int q_off, int k_off, int H, int Ntok, int T, int B,
                 const void* wq_img, const void* wk_img, const void* wq_txt, const void* wk_txt, const float* cosT,
                 const float* sinT, float eps, hipStream_t st) {
  const int64_t groups = (int64_t)B * Ntok * 4;   // 4 lane groups of 16 per token (HSPLIT in the kernel)
  if (groups == 0 || H == 0) return 0;
  rmsnorm_rope_kernel<<<dim3((unsigned)((groups + 15) / 16)), 256, 0, st>>>(
      (bf16_t*)buf, ld, bstride, q_off, k_off, H, Ntok, T, B, (const bf16_t*)wq_img, (const bf16_t*)wk_img,
      (const bf16_t*)wq_txt, (const bf16_t*)wk_txt, cosT, sinT, eps);
  return check_launch("rmsnorm_rope");
}

int sched_step(bool amo, const void* v, void* x, void* xin, int64_t ldxin, int C, int64_t rows, const float* coef,
               const int* step_ptr, int step, const float* noise, hipStream_t st) {
  if (C % 8) return fail("sched_step: C must be a multiple of 8");
  const int64_t nch = rows * C / 8;
  if (nch == 0) return 0;
  dim3 grid((unsigned)((nch + 255) / 256));
  if (amo) {
    if (!noise) return fail("amo_step: noise pointer is null");
    sched_step_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)v, (bf16_t*)x, (bf16_t*)xin, ldxin, C, nch, coef,
                                                  step_ptr, step, noise);
  } else {
    sched_step_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)v, (bf16_t*)x, (bf16_t*)xin, ldxin, C, nch, coef,
                                                   step_ptr, step, nullptr);
  }
  return check_launch("sched_step");
}

int timestep_embedding(const float* t, void* out, int n, hipStream_t st) {
  if (n == 0) return 0;
  timestep_embedding_kernel<<<n, 128, 0, st>>>(t, (bf16_t*)out, n);
  return check_launch("timestep_embedding");
}

int silu_bf16(const void* a, void* out, int64_t n, hipStream_t st) {
  if (n == 0) return 0;
  unary_binary_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>((const bf16_t*)a, nullptr, (bf16_t*)out, n, 0);
  return check_launch("silu");
}
int add_bf16(const void* a, const void* b, void* out, int64_t n, hipStream_t st) {
  if (n == 0) return 0;
  unary_binary_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>((const bf16_t*)a, (const bf16_t*)b,
                                                                        (bf16_t*)out, n, 1);
  return check_launch("add");
}
int scatter_cols(const void* src, void* dst, int64_t rows, int C, int64_t ld, int col0, hipStream_t st) {
  if (C % 8 || col0 % 8 || ld % 8) return fail("scatter_cols: C, col0, ld must be multiples of 8");
  const int64_t n = rows * (C / 8);
  if (n == 0) return 0;
  scatter_cols_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, rows, C, ld,
                                                                         col0);
  return check_launch("scatter_cols");
}
int copy_rows(const void* src, int64_t sld, int64_t sbs, void* dst, int64_t dld, int64_t dbs, int rows, int cols,
              int batch, hipStream_t st) {
  if (cols % 8 || sld % 8 || dld % 8 || sbs % 8 || dbs % 8) return fail("copy_rows: cols/strides must be multiples of 8");
  const int64_t total = (int64_t)batch * rows * (cols / 8);
  if (total == 0) return 0;
  copy_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>((const bf16_t*)src, sld, sbs, (bf16_t*)dst, dld,
                                                                          dbs, rows, cols / 8, total);
  return check_launch("copy_rows");
}
int gate_residual(const void* x, int64_t ldx, int64_t x_bs, const void* gate, int64_t gate_bs, const void* res, int64_t ldr, int64_t r_bs,
                  void* out, int64_t ldo, int64_t o_bs, int rows, int batch, int D, hipStream_t st) {
  const int64_t total = (int64_t)batch * rows * (D / 8);
  if (total == 0) return 0;
  gate_residual_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>((const bf16_t*)x, ldx, x_bs, (const bf16_t*)gate, gate_bs,
                                                                             (const bf16_t*)res, ldr, r_bs, (bf16_t*)out, ldo, o_bs, rows,
                                                                             D / 8, total);
  return check_launch("gate_residual");
}
int blend_edge(const void* a, int64_t a_bs, int64_t a_ts, int64_t a_us, void* b, int64_t b_bs, int64_t b_ts, int64_t b_us, int batch,
               int extent, int len, int C, hipStream_t st) {
  const int64_t total = (int64_t)batch * extent * len * (C / 8);
  if (total <= 0) return 0;
  blend_edge_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>((const bf16_t*)a, a_bs, a_ts, a_us, (bf16_t*)b, b_bs, b_ts, b_us,
                                                                          extent, len, C / 8, total);
  return check_launch("blend_edge");
}
int groupnorm_silu_nhwc(const void* x, void* out, const void* gamma, const void* beta, float* ws, int B, int64_t HW,
                        int C, int groups, float eps, bool silu, hipStream_t st) {
  if (C % 8 || groups > 64 || C % groups || (C / groups) % 4 || 256 % (C / 8)) return fail("groupnorm: unsupported C / groups");
  const int nchunk = (int)((HW + GN_CHUNK_PIX - 1) / GN_CHUNK_PIX);
  float* part = ws;                                   // [B, nchunk, groups, 2]
  float* stat = ws + (int64_t)B * nchunk * groups * 2; // [B, groups, 2]
  gn_partial_kernel<<<dim3(nchunk, B), 256, 0, st>>>((const bf16_t*)x, part, HW, C, groups, nchunk);
  gn_finalize_kernel<<<B, 256, 0, st>>>(part, stat, nchunk, groups, 1.0f / ((float)HW * (C / groups)), eps);
  const int64_t per_sample = HW * (C / 8);
  const dim3 grid((unsigned)((per_sample + 256 * GN_APPLY_ITER - 1) / (256 * GN_APPLY_ITER)), (unsigned)B);
  if (silu)
    gn_apply_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)out, (const bf16_t*)gamma, (const bf16_t*)beta,
                                                stat, per_sample, C, groups);
  else
    gn_apply_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)x, (bf16_t*)out, (const bf16_t*)gamma, (const bf16_t*)beta,
                                                 stat, per_sample, C, groups);
  return check_launch("groupnorm_silu_nhwc");
}
int select_step(const void* table, void* cur, int64_t per_step_elems, int* step_ptr, hipStream_t st) {
  if (per_step_elems % 8) return fail("select_step: per-step size must be a multiple of 8 elements");
  const int64_t ch = per_step_elems / 8;
  unsigned grid = (unsigned)((ch + 255) / 256);
  if (grid > 2048) grid = 2048;
  select_step_kernel<<<grid, 256, 0, st>>>((const bf16_t*)table, (bf16_t*)cur, ch, step_ptr, 0);
  return check_launch("select_step");
}
int advance_step(int* step_ptr, hipStream_t st) {
  advance_step_kernel<<<1, 1, 0, st>>>(step_ptr);
  return check_launch("advance_step");
}

// ---- per-row absmax quantisation bf16 -> fp8 e4m3 (OCP): one 256-thread block per row, row cached in registers.
// scale = absmax / 448, out = round_to_e4m3(x / scale); v_cvt_pk_fp8_f32 does not saturate (NaN above 448), hence the clamp.
template <int VPT>   // 16-byte vectors (8 elements) per thread; K <= 256 * 8 * VPT
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, int64_t x_bs, uint8_t* __restrict__ out,
                                                            int64_t ldo, int64_t o_bs, float* __restrict__ scale, int64_t s_bs, int rows, int K) {
  const int row = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const bf16_t* xr = x + b * x_bs + (int64_t)row * ldx;
  u32x4 v[VPT];
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (tid + i * 256) * 8;
    v[i] = u32x4{0, 0, 0, 0};
    if (c < K) v[i] = *reinterpret_cast<const u32x4*>(xr + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      amax = fmaxf(amax, fabsf(__uint_as_float(v[i][e] << 16)));
      amax = fmaxf(amax, fabsf(__uint_as_float(v[i][e] & 0xffff0000u)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  __shared__ float red[4];
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;   // the correctly rounded quotient, as torch's amax / 448
  if (tid == 0) scale[b * s_bs + row] = sc;
  uint8_t* orow = out + b * o_bs + (int64_t)row * ldo;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = (tid + i * 256) * 8;
    if (c >= K) continue;
    float f[8];
    unpack8(v[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(__fdiv_rn(f[e], sc), -448.f), 448.f);  // exact quotient: bf16 data sits on e4m3 ties
    uint32_t w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    *reinterpret_cast<u32x2*>(orow + c) = u32x2{w0, w1};
  }
}

int quantize_rows_fp8(const void* x, int64_t ldx, int64_t x_bstride, void* out, int64_t ldo, int64_t o_bstride, float* scale,
                      int64_t s_bstride, int rows, int batch, int K, hipStream_t st) {
  if (rows <= 0 || batch <= 0) return 0;
  if (K <= 0 || K % 8 || ldx % 8 || x_bstride % 8 || ldo % 8 || o_bstride % 8 || (uintptr_t)x % 16 || (uintptr_t)out % 8)
    return fail("quantize_rows_fp8: K, strides must be multiples of 8 and pointers 16 / 8-byte aligned");
  if (K > 256 * 8 * 8) return fail("quantize_rows_fp8: K = %d exceeds 16384", K);
  const dim3 grid(rows, batch);
  const bf16_t* xp = (const bf16_t*)x;
  uint8_t* op = (uint8_t*)out;
  if (K <= 256 * 8 * 2) quant_rows_fp8_kernel<2><<<grid, 256, 0, st>>>(xp, ldx, x_bstride, op, ldo, o_bstride, scale, s_bstride, rows, K);
  else if (K <= 256 * 8 * 6) quant_rows_fp8_kernel<6><<<grid, 256, 0, st>>>(xp, ldx, x_bstride, op, ldo, o_bstride, scale, s_bstride, rows, K);
  else quant_rows_fp8_kernel<8><<<grid, 256, 0, st>>>(xp, ldx, x_bstride, op, ldo, o_bstride, scale, s_bstride, rows, K);
  return check_launch("quantize_rows_fp8");
}

}  // namespace tfx
