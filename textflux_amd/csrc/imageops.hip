// The two ends of FluxFillPipeline.__call__ around the VAE, as HBM-bound layout kernels (reference sites in
// D/pipelines/flux/pipeline_flux_fill.py and D/image_processor.py; "P:" / "IP:" below):
//   prep_image      IP:587-716 (tensor / PIL branches: /255, 2x-1) + P:2030 `image * (1 - mask)` + the bf16 cast of
//                   P:2031, written straight as the NHWC, 8-channel-padded input of the encoder's conv_in
//   pack_mask       P:1563-1580  mask [B,1,H,W] -> 8x8 pixel blocks -> 2x2 patchify -> [B,S,256] (binarised, IP:535-536)
//   sample_pack     P:1528-1530, 1554-1560  posterior sample (mean + std * eps), (z - shift) * scale, 2x2 patchify
//   unpack_latents  P:1752-1765, 2126-2127  un-patchify + z / scale + shift, written NHWC for the decoder's conv_in
//   postprocess     IP:718-771   denormalise, clamp, -> NCHW bf16 ("pt") / NHWC fp32 ("np") / NHWC uint8 ("pil")
// plus two helpers of the VAE mid-block attention (single head of dim C, D/models/attention_processor.py:2799-2881):
//   transpose       v [N, C] -> v^T [C, N] so that P @ v runs on the MFMA GEMM (C = A @ W^T)
//   row_softmax     bf16 softmax(scale * s) over rows of the fp32 score matrix (tfx_gemm_bf16_f32), fp32 statistics
// Every bf16 rounding point of the reference's op chain is kept (each torch op on bf16 tensors rounds its result).
// Python-scalar operands (shift_factor, scaling_factor) follow the semantics of the reference's DEVICE kernels: the scalar
// stays fp32 (opmath) and `tensor / scalar` is a multiply by the fp32 reciprocal -- torch's CPU kernels round the scalar to
// bf16 first and divide, so a bf16 CPU run of the reference differs from its GPU run in exactly these two places.
#include "common.h"
#include "launch.h"

namespace tfx {

// flag |= 1 if any element is negative (VaeImageProcessor.preprocess skips the 2x-1 normalisation for tensors that are
// already in [-1, 1]: `if do_normalize and image.min() < 0: do_normalize = False`, IP:700-707) -- decided on the device
template <typename T>
__global__ __launch_bounds__(256) void any_negative_kernel(const T* __restrict__ x, int64_t n, int* __restrict__ flag) {
  bool neg = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v;
    if constexpr (sizeof(T) == 2) v = bf2f(x[i]); else v = x[i];
    neg |= v < 0.f;
  }
  if (__any(neg) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__device__ __forceinline__ float ld_f(const float* p, int64_t i) { return p[i]; }
__device__ __forceinline__ float ld_f(const bf16_t* p, int64_t i) { return bf2f(p[i]); }
__device__ __forceinline__ float ld_f(const uint8_t* p, int64_t i) { return __fdiv_rn((float)p[i], 255.0f); }

// out[b, y, x, 0..7] = bf16( img * (1 - mask) ), channels C..7 zero.  TI: float / bf16 planes [B, C, H, W], or uint8
// interleaved [B, H, W, C] (values / 255).  mask (optional): float / uint8 [Bm, H, W], Bm in {1, B}.
// norm_mode: 0 none, 1 always 2x-1, 2 2x-1 unless *neg_flag != 0.
template <typename TI, typename TM>
__global__ __launch_bounds__(256) void prep_image_kernel(const TI* __restrict__ img, const TM* __restrict__ mask,
                                                         bf16_t* __restrict__ out, int B, int C, int H, int W, int mask_b,
                                                         int norm_mode, int binarize, const int* __restrict__ neg_flag) {
  const int64_t hw = (int64_t)H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * hw) return;
  const int b = (int)(i / hw);
  const int64_t pix = i - b * hw;
  const bool norm = norm_mode == 1 || (norm_mode == 2 && *neg_flag == 0);
  float keep = 1.0f;
  if (mask) {
    float m = ld_f(mask, (mask_b == 1 ? 0 : b) * hw + pix);
    if (binarize) m = m < 0.5f ? 0.f : 1.f;
    keep = 1.0f - m;
  }
  float v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    v[c] = 0.f;
    if (c < C) {
      float x;
      if constexpr (sizeof(TI) == 1) x = ld_f(img, (b * hw + pix) * C + c);
      else x = ld_f(img, ((int64_t)b * C + c) * hw + pix);
      if (norm) x = __fsub_rn(__fmul_rn(2.0f, x), 1.0f);
      v[c] = __fmul_rn(x, keep);
    }
  }
  *reinterpret_cast<u32x4*>(out + i * 8) = pack8(v);
}

// mask [Bm, H, W] -> out[b, t, col0 + (i*8+j)*4 + py*2+px] = mask[(2ty+py)*8 + i, (2tx+px)*8 + j], h = H/8, w = W/8,
// t = ty * (w/2) + tx.  One thread = one token x 8 consecutive packed columns (i, j0..j0+1, all four (py, px)).
template <typename TM>
__global__ __launch_bounds__(256) void pack_mask_kernel(const TM* __restrict__ mask, bf16_t* __restrict__ out, int B, int H,
                                                        int W, int mask_b, int binarize, int64_t ld, int col0) {
  const int h2 = H / 16, w2 = W / 16;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t S = (int64_t)h2 * w2;
  if (i >= B * S * 32) return;
  const int c8 = (int)(i & 31);
  const int64_t bt = i >> 5;
  const int b = (int)(bt / S);
  const int t = (int)(bt - b * S);
  const int ty = t / w2, tx = t - ty * w2;
  const int ii = c8 >> 2, j0 = (c8 & 3) * 2;
  const TM* mb = mask + (int64_t)(mask_b == 1 ? 0 : b) * H * W;
  float v[8];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int y = (2 * ty + (pp >> 1)) * 8 + ii, x = (2 * tx + (pp & 1)) * 8 + j0 + jj;
      float m = ld_f(mb, (int64_t)y * W + x);
      if (binarize) m = m < 0.5f ? 0.f : 1.f;
      v[jj * 4 + pp] = m;
    }
  *reinterpret_cast<u32x4*>(out + bt * ld + col0 + c8 * 8) = pack8(v);
}

// moments NHWC [B, h, w, 2L] (mean | logvar) bf16, eps [B, L, h, w] bf16 or fp32 (null: the mode) ->
// out[b, t, col0 + c*4 + py*2+px] = bf16chain( (mean + exp(0.5 * clamp(logvar)) * eps - shift) * scale ) at pixel
// (2ty+py, 2tx+px); every intermediate rounded to bf16 as the reference's bf16 tensor ops round them
// (DiagonalGaussianDistribution, D/models/autoencoders/vae.py:781-802; P:1530).
template <typename TE>
__global__ __launch_bounds__(256) void sample_pack_kernel(const bf16_t* __restrict__ mom, const TE* __restrict__ eps,
                                                          bf16_t* __restrict__ out, int B, int h, int w, int L, float shift,
                                                          float scale, int64_t ld, int col0) {
  const int h2 = h / 2, w2 = w / 2;
  const int64_t S = (int64_t)h2 * w2;
  const int cpt = L / 2;                       // 8-column groups per token (2 channels x 4 positions each)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * S * cpt) return;
  const int g = (int)(i % cpt);
  const int64_t bt = i / cpt;
  const int b = (int)(bt / S);
  const int t = (int)(bt - b * S);
  const int ty = t / w2, tx = t - ty * w2;
  float v[8];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int c = g * 2 + cc, y = 2 * ty + (pp >> 1), x = 2 * tx + (pp & 1);
      const bf16_t* m = mom + (((int64_t)b * h + y) * w + x) * (2 * L);
      const float mean = bf2f(m[c]);
      float z = mean;
      if (eps) {
        const float lv = fminf(fmaxf(bf2f(m[L + c]), -30.f), 20.f);
        const float sd = round_bf(expf(round_bf(0.5f * lv)));
        const float e = ld_f(eps, (((int64_t)b * L + c) * h + y) * w + x);
        z = round_bf(mean + round_bf(sd * round_bf(e)));
      }
      v[cc * 4 + pp] = round_bf(z - shift) * scale;
    }
  *reinterpret_cast<u32x4*>(out + bt * ld + col0 + g * 8) = pack8(v);
}

// latents [B, S, 4L] -> z NHWC [B, h, w, L] = bf16(bf16(lat * (1 / scale)) + shift); z[b, y, x, c] <- col c*4 + (y&1)*2 + (x&1)
// of token (y>>1, x>>1).  One thread = one pixel x 8 channels.
__global__ __launch_bounds__(256) void unpack_latents_kernel(const bf16_t* __restrict__ lat, int64_t ld, bf16_t* __restrict__ out,
                                                             int B, int h, int w, int L, float shift, float scale) {
  const int cg = L / 8;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * h * w * cg) return;
  const int g = (int)(i % cg);
  const int64_t p = i / cg;
  const int x = (int)(p % w);
  const int y = (int)((p / w) % h);
  const int b = (int)(p / ((int64_t)w * h));
  const bf16_t* row = lat + ((int64_t)b * (h / 2) * (w / 2) + (int64_t)(y >> 1) * (w / 2) + (x >> 1)) * ld + (y & 1) * 2 + (x & 1);
  const float inv_scale = __fdiv_rn(1.0f, scale);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = round_bf(__fmul_rn(bf2f(row[(g * 8 + e) * 4]), inv_scale)) + shift;
  *reinterpret_cast<u32x4*>(out + i * 8) = pack8(v);
}

// x NHWC [B, H, W, Cs] bf16 (first C channels used), cropped to the window (y0, x0, Hc, Wc) -> mode 0: NCHW bf16,
// 1: NHWC fp32, 2: NHWC uint8 (round(255 v)), 3: NCHW fp32.  denorm: v = clamp(bf16(bf16(x * 0.5) + 0.5), 0, 1)
// (VaeImageProcessor.denormalize, IP:227-239).  The window is the callers' result crop (run_inference.py:460-465: only the
// scene part of the concatenated image is kept), applied before the image leaves the device.
__global__ __launch_bounds__(256) void postprocess_kernel(const bf16_t* __restrict__ x, void* __restrict__ out, int B, int H, int W,
                                                          int Cs, int C, int mode, int denorm, int y0, int x0, int Hc, int Wc) {
  const int64_t HWc = (int64_t)Hc * Wc;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * HWc) return;
  const int b = (int)(i / HWc);
  const int64_t pix = i - b * HWc;
  const int yy = (int)(pix / Wc), xx = (int)(pix - (int64_t)yy * Wc);
  const bf16_t* src = x + (((int64_t)b * H + y0 + yy) * W + x0 + xx) * Cs;
  for (int c = 0; c < C; ++c) {
    float v = bf2f(src[c]);
    if (denorm) v = fminf(fmaxf(round_bf(round_bf(v * 0.5f) + 0.5f), 0.f), 1.f);
    if (mode == 0) ((bf16_t*)out)[((int64_t)b * C + c) * HWc + pix] = f2bf(v);
    else if (mode == 1) ((float*)out)[i * C + c] = v;
    else if (mode == 2) ((uint8_t*)out)[i * C + c] = (uint8_t)rintf(__fmul_rn(v, 255.0f));
    else ((float*)out)[((int64_t)b * C + c) * HWc + pix] = v;
  }
}

// out[b][c, n] = in[b][n, c]  (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int64_t ldi, int64_t ibs,
                                                        bf16_t* __restrict__ out, int64_t ldo, int64_t obs, int N, int C) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    if (n0 + r < N && c0 + tx < C) tile[r][tx] = in[b * ibs + (int64_t)(n0 + r) * ldi + c0 + tx];
  __syncthreads();
  for (int r = ty; r < 64; r += 4)
    if (c0 + r < C && n0 + tx < N) out[b * obs + (int64_t)(c0 + r) * ldo + n0 + tx] = tile[tx][r];
}

// p[r, :N] = bf16( softmax(scale * s[r, :N]) ): fp32 scores in (row stride lds), bf16 weights out (row stride ldp), fp32
// statistics, one 256-thread block per row (the row stays in L2 between the three passes).
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ s, int64_t lds, bf16_t* __restrict__ pout,
                                                          int64_t ldp, int N, float scale_log2e) {
  __shared__ float red[4];
  const float* row = s + (int64_t)blockIdx.x * lds;
  bf16_t* prow = pout + (int64_t)blockIdx.x * ldp;
  const int tid = threadIdx.x;
  const bool vec = (N % 8 == 0) && (lds % 4 == 0) && (ldp % 8 == 0) && ((uintptr_t)s % 16 == 0) && ((uintptr_t)pout % 16 == 0);
  auto block_reduce = [&](float v, bool is_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float u = __shfl_xor(v, o, 64);
      v = is_max ? fmaxf(v, u) : v + u;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
  };
  auto ld8 = [&](int c, float* f) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(row + c * 8), b = *reinterpret_cast<const f32x4*>(row + c * 8 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
  };
  float mx = -INFINITY;
  if (vec) {
    for (int c = tid; c < N / 8; c += 256) {
      float f[8];
      ld8(c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
  } else {
    for (int c = tid; c < N; c += 256) mx = fmaxf(mx, row[c]);
  }
  mx = block_reduce(mx, true);
  const float mc = mx * scale_log2e;
  float sum = 0.f;
  if (vec) {
    for (int c = tid; c < N / 8; c += 256) {
      float f[8];
      ld8(c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f(f[e] * scale_log2e - mc);
    }
  } else {
    for (int c = tid; c < N; c += 256) sum += __builtin_amdgcn_exp2f(row[c] * scale_log2e - mc);
  }
  sum = block_reduce(sum, false);
  const float inv = 1.0f / sum;
  if (vec) {
    for (int c = tid; c < N / 8; c += 256) {
      float f[8];
      ld8(c, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = __builtin_amdgcn_exp2f(f[e] * scale_log2e - mc) * inv;
      *reinterpret_cast<u32x4*>(prow + c * 8) = pack8(f);
    }
  } else {
    for (int c = tid; c < N; c += 256) prow[c] = f2bf(__builtin_amdgcn_exp2f(row[c] * scale_log2e - mc) * inv);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

int any_negative(const void* x, int dtype, int64_t n, int* flag, hipStream_t st) {
  if (n <= 0) return 0;
  unsigned grid = blocks_for(n);
  if (grid > 4096) grid = 4096;
  if (dtype == 0) any_negative_kernel<float><<<grid, 256, 0, st>>>((const float*)x, n, flag);
  else if (dtype == 1) any_negative_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, n, flag);
  else return fail("any_negative: dtype must be 0 (f32) or 1 (bf16)");
  return check_launch("any_negative");
}

int prep_image(const void* img, int img_dtype, const void* mask, int mask_dtype, void* out, int B, int C, int H, int W,
               int mask_b, int norm_mode, int binarize, const int* neg_flag, hipStream_t st) {
  if (C < 1 || C > 8) return fail("prep_image: 1..8 channels");
  if (norm_mode == 2 && !neg_flag) return fail("prep_image: norm_mode 2 needs the negative-values flag");
  if (mask && mask_b != 1 && mask_b != B) return fail("prep_image: mask batch must be 1 or B");
  const int64_t n = (int64_t)B * H * W;
  if (n <= 0) return 0;
  const unsigned grid = blocks_for(n);
  bf16_t* o = (bf16_t*)out;
#define TFX_PREP(TI, TM) prep_image_kernel<TI, TM><<<grid, 256, 0, st>>>((const TI*)img, (const TM*)mask, o, B, C, H, W, mask_b, norm_mode, binarize, neg_flag)
  const int md = mask ? mask_dtype : 0;
  if (md != 0 && md != 2) return fail("prep_image: mask dtype must be 0 (f32) or 2 (u8)");
  switch (img_dtype * 4 + md) {
    case 0: TFX_PREP(float, float); break;
    case 2: TFX_PREP(float, uint8_t); break;
    case 4: TFX_PREP(bf16_t, float); break;
    case 6: TFX_PREP(bf16_t, uint8_t); break;
    case 8: TFX_PREP(uint8_t, float); break;
    case 10: TFX_PREP(uint8_t, uint8_t); break;
    default: return fail("prep_image: image dtype must be 0 (f32 planes), 1 (bf16 planes) or 2 (u8 interleaved)");
  }
#undef TFX_PREP
  return check_launch("prep_image");
}

// Composition of the pipeline's input canvas out of its parts (reference: run_inference.py:409-467 -- the rendered glyph
// image and the scene are stacked, glyph first, the glyph part of the mask is black -- and PIL's convert("L") of the RGB mask
// in VaeImageProcessor.preprocess): canvas [B, H, W, 3] u8, cmask [B, H, W] u8 from glyph [B, gh, gw, 3], scene [B, sh, sw, 3]
// and the scene's RGB mask [B, sh, sw, 3], all u8 interleaved.  dir 0: vertical (H = gh + sh, W = gw = sw), 1: horizontal.
// The grey value is Pillow's integer formula (L24 = 19595 R + 38470 G + 7471 B + 0x8000) >> 16, bit for bit.
__global__ __launch_bounds__(256) void compose_canvas_kernel(const uint8_t* __restrict__ glyph, const uint8_t* __restrict__ scene,
                                                             const uint8_t* __restrict__ smask, uint8_t* __restrict__ canvas,
                                                             uint8_t* __restrict__ cmask, int B, int gh, int gw, int sh, int sw,
                                                             int dir, int mask_rgb) {
  const int H = dir ? sh : gh + sh, W = dir ? gw + sw : sw;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * H * W) return;
  const int x = (int)(i % W);
  const int64_t by = i / W;
  const int y = (int)(by % H), b = (int)(by / H);
  const bool in_glyph = dir ? x < gw : y < gh;
  uint8_t r, g, bl, m0 = 0, m1 = 0, m2 = 0;
  if (in_glyph) {
    const uint8_t* p = glyph + (((int64_t)b * gh + y) * gw + x) * 3;
    r = p[0]; g = p[1]; bl = p[2];
  } else {
    const int sy = dir ? y : y - gh, sx = dir ? x - gw : x;
    const int64_t o = (((int64_t)b * sh + sy) * sw + sx) * 3;
    r = scene[o]; g = scene[o + 1]; bl = scene[o + 2];
    m0 = smask[o]; m1 = smask[o + 1]; m2 = smask[o + 2];
  }
  uint8_t* c = canvas + i * 3;
  c[0] = r; c[1] = g; c[2] = bl;
  if (mask_rgb) {            // the mask stays RGB when a resize follows (the reference resizes the RGB mask, then takes "L")
    uint8_t* mm = cmask + i * 3;
    mm[0] = m0; mm[1] = m1; mm[2] = m2;
  } else {
    cmask[i] = (uint8_t)((19595u * m0 + 38470u * m1 + 7471u * m2 + 0x8000u) >> 16);
  }
}

// PIL's convert("L") of interleaved RGB u8: (19595 R + 38470 G + 7471 B + 0x8000) >> 16.
__global__ __launch_bounds__(256) void rgb_to_grey_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = rgb + i * 3;
  out[i] = (uint8_t)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
}

// One pass of Pillow's 8-bit convolution resampler (libImaging/Resample.c: ImagingResampleHorizontal_8bpc / Vertical_8bpc)
// along the middle axis of in [outer][in_len][inner] -> out [outer][out_len][inner]: per output position xx the window
// [bounds[2xx], + bounds[2xx+1]) and its ksize fixed-point coefficients (22 fractional bits, precomputed on the host exactly as
// precompute_coeffs + normalize_coeffs_8bpc do), ss = 2^21 + sum in * k, out = clip8(ss >> 22).  Integer arithmetic: bit-exact.
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize,
                                                          int64_t outer, int in_len, int out_len, int inner) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= outer * out_len * inner) return;
  const int e = (int)(i % inner);
  const int64_t ox = i / inner;
  const int xx = (int)(ox % out_len);
  const int64_t o = ox / out_len;
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = coeffs + (int64_t)xx * ksize;
  const uint8_t* src = in + ((o * in_len + xmin) * inner + e);
  int ss = 1 << 21;
  for (int x = 0; x < xmax; ++x) ss += (int)src[(int64_t)x * inner] * k[x];
  ss >>= 22;
  out[i] = (uint8_t)(ss < 0 ? 0 : ss > 255 ? 255 : ss);
}

int compose_canvas(const void* glyph, const void* scene, const void* smask, void* canvas, void* cmask, int B, int gh, int gw, int sh,
                   int sw, int dir, int mask_rgb, hipStream_t st) {
  if (dir != 0 && dir != 1) return fail("compose_canvas: direction 0 (vertical) or 1 (horizontal)");
  if (dir == 0 ? gw != sw : gh != sh) return fail("compose_canvas: glyph and scene must share the side they are stacked along");
  const int64_t n = (int64_t)B * (dir ? sh : gh + sh) * (dir ? gw + sw : sw);
  if (n <= 0) return 0;
  compose_canvas_kernel<<<blocks_for(n), 256, 0, st>>>((const uint8_t*)glyph, (const uint8_t*)scene, (const uint8_t*)smask,
                                                       (uint8_t*)canvas, (uint8_t*)cmask, B, gh, gw, sh, sw, dir, mask_rgb);
  return check_launch("compose_canvas");
}

int rgb_to_grey(const void* rgb, void* out, int64_t n, hipStream_t st) {
  if (n <= 0) return 0;
  rgb_to_grey_kernel<<<blocks_for(n), 256, 0, st>>>((const uint8_t*)rgb, (uint8_t*)out, n);
  return check_launch("rgb_to_grey");
}

int resample_u8(const void* in, void* out, const int* bounds, const int* coeffs, int ksize, int64_t outer, int in_len, int out_len,
                int inner, hipStream_t st) {
  if (ksize < 1 || in_len < 1 || out_len < 1 || inner < 1) return fail("resample_u8: bad geometry");
  const int64_t n = outer * out_len * inner;
  if (n <= 0) return 0;
  resample_u8_kernel<<<blocks_for(n), 256, 0, st>>>((const uint8_t*)in, (uint8_t*)out, bounds, coeffs, ksize, outer, in_len, out_len, inner);
  return check_launch("resample_u8");
}

int pack_mask(const void* mask, int mask_dtype, void* out, int B, int H, int W, int mask_b, int binarize, int64_t ld, int col0,
              hipStream_t st) {
  if (H % 16 || W % 16) return fail("pack_mask: H and W must be multiples of 16");
  if (ld % 8 || col0 % 8) return fail("pack_mask: ld / col0 must be multiples of 8");
  if (mask_b != 1 && mask_b != B) return fail("pack_mask: mask batch must be 1 or B");
  const int64_t n = (int64_t)B * (H / 16) * (W / 16) * 32;
  if (n <= 0) return 0;
  if (mask_dtype == 0) pack_mask_kernel<float><<<blocks_for(n), 256, 0, st>>>((const float*)mask, (bf16_t*)out, B, H, W, mask_b, binarize, ld, col0);
  else if (mask_dtype == 2) pack_mask_kernel<uint8_t><<<blocks_for(n), 256, 0, st>>>((const uint8_t*)mask, (bf16_t*)out, B, H, W, mask_b, binarize, ld, col0);
  else return fail("pack_mask: mask dtype must be 0 (f32) or 2 (u8)");
  return check_launch("pack_mask");
}

int sample_pack(const void* moments, const void* eps, int eps_dtype, void* out, int B, int h, int w, int L, float shift,
                float scale, int64_t ld, int col0, hipStream_t st) {
  if (h % 2 || w % 2 || L % 2 || ld % 8 || col0 % 8) return fail("sample_pack: h, w, L even; ld / col0 multiples of 8");
  const int64_t n = (int64_t)B * (h / 2) * (w / 2) * (L / 2);
  if (n <= 0) return 0;
  if (!eps || eps_dtype == 1)
    sample_pack_kernel<bf16_t><<<blocks_for(n), 256, 0, st>>>((const bf16_t*)moments, (const bf16_t*)eps, (bf16_t*)out, B, h, w, L, shift, scale, ld, col0);
  else if (eps_dtype == 0)
    sample_pack_kernel<float><<<blocks_for(n), 256, 0, st>>>((const bf16_t*)moments, (const float*)eps, (bf16_t*)out, B, h, w, L, shift, scale, ld, col0);
  else return fail("sample_pack: eps dtype must be 0 (f32) or 1 (bf16)");
  return check_launch("sample_pack");
}

int unpack_latents(const void* lat, int64_t ld, void* out, int B, int h, int w, int L, float shift, float scale, hipStream_t st) {
  if (h % 2 || w % 2 || L % 8) return fail("unpack_latents: h, w even; L a multiple of 8");
  const int64_t n = (int64_t)B * h * w * (L / 8);
  if (n <= 0) return 0;
  unpack_latents_kernel<<<blocks_for(n), 256, 0, st>>>((const bf16_t*)lat, ld, (bf16_t*)out, B, h, w, L, shift, scale);
  return check_launch("unpack_latents");
}

int postprocess(const void* x, void* out, int B, int H, int W, int Cs, int C, int mode, int denorm, int y0, int x0, int Hc, int Wc,
                hipStream_t st) {
  if (mode < 0 || mode > 3 || C > Cs) return fail("postprocess: mode 0..3, C <= Cs");
  if (y0 < 0 || x0 < 0 || Hc <= 0 || Wc <= 0 || y0 + Hc > H || x0 + Wc > W) return fail("postprocess: crop window outside the image");
  const int64_t n = (int64_t)B * Hc * Wc;
  if (n <= 0) return 0;
  postprocess_kernel<<<blocks_for(n), 256, 0, st>>>((const bf16_t*)x, out, B, H, W, Cs, C, mode, denorm, y0, x0, Hc, Wc);
  return check_launch("postprocess");
}

int transpose_bf16(const void* in, int64_t ldi, int64_t ibs, void* out, int64_t ldo, int64_t obs, int N, int C, int batch,
                   hipStream_t st) {
  if (N <= 0 || C <= 0 || batch <= 0) return 0;
  transpose_kernel<<<dim3((N + 63) / 64, (C + 63) / 64, batch), 256, 0, st>>>((const bf16_t*)in, ldi, ibs, (bf16_t*)out, ldo, obs, N, C);
  return check_launch("transpose");
}

int row_softmax(const float* s, int64_t lds, void* p, int64_t ldp, int rows, int N, float scale, hipStream_t st) {
  if (rows <= 0 || N <= 0) return 0;
  row_softmax_kernel<<<rows, 256, 0, st>>>(s, lds, (bf16_t*)p, ldp, N, scale * 1.4426950408889634f);
  return check_launch("row_softmax");
}

}  // namespace tfx
