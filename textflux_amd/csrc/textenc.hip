// Kernels of the two text encoders of FluxFillPipeline.encode_prompt (D/pipelines/flux/pipeline_flux_fill.py:1411-1503):
// T5 v1.1 encoder (T5EncoderModel, 24 layers at d_model 4096 / 64 heads x 64 / d_ff 10240 for T5-XXL) and the CLIP-L text
// model (12 layers, 768 wide, 12 heads x 64).  Their arithmetic lives in third-party `transformers` (pinned 4.43.3 by the
// reference's requirements.txt; models/t5/modeling_t5.py, models/clip/modeling_clip.py), not under /root/reference; the
// published algorithms are restated in oracle/text_oracle.py and pinned against the installed `transformers` on tiny
// random configurations (tests/golden/g10_text.safetensors).  The big matrix products run on the DiT's MFMA GEMM; here:
//   attn64         softmax(scale * q k^T + bias) v for heads of dim 64, N <= 512 keys: T5's bucketed relative-position
//                  bias (looked up from a [H, 2N-1] table by key - query, no scaling, no mask) or CLIP's causal mask;
//                  one workgroup = 128 queries of one (batch, head), the head's K and V^T resident in LDS, MFMA for both
//                  products, exact online softmax in fp32
//   rmsnorm        T5LayerNorm: bf16( w * bf16(x * rsqrt(mean(x^2) + eps)) ), x fp32 (the residual stream) or bf16
//   gather_rows    token / position embedding lookup
//   add_into_f32   x32 += y: T5's residual stream is fp32 under torch_dtype = bf16 because `wo` is kept in fp32
//                  (T5PreTrainedModel._keep_in_fp32_modules = ["wo"]) and bf16 + fp32 promotes
//   mul / quick_gelu  gated-GELU product (T5DenseGatedActDense) and CLIP's x * sigmoid(1.702 x)
#include "common.h"
#include "launch.h"

namespace tfx {

namespace {
constexpr int A64_HD = 64, A64_KROW = 144;   // K rows in LDS: 128 B of data + 16 B pad
}

__global__ __launch_bounds__(256) void attn64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                     const bf16_t* __restrict__ Vp, bf16_t* __restrict__ O, int64_t ldq,
                                                     int64_t ldk, int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs,
                                                     int64_t v_bs, int64_t o_bs, int H, int N, int NP, float scale,
                                                     const float* __restrict__ rel_bias, int causal) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const bf16_t* Qb = Q + b * q_bs + h * A64_HD;
  const bf16_t* Kb = Kp + b * k_bs + h * A64_HD;
  const bf16_t* Vb = Vp + b * v_bs + h * A64_HD;
  bf16_t* Ob = O + b * o_bs + h * A64_HD;
  char* Ks = smem;                                   // [NP][144 B]
  const int vt_stride = 2 * (NP + 8);                // V^T rows: NP keys + 8 pad, bytes
  char* Vt = smem + (size_t)NP * A64_KROW;           // [64][vt_stride]

  // ---- stage the head's K (row-major) and V (transposed) once per workgroup; keys >= N are zero
  for (int c = tid; c < NP * 8; c += 256) {
    const int key = c >> 3, ch = c & 7;
    u32x4 kv = u32x4{0, 0, 0, 0}, vv = u32x4{0, 0, 0, 0};
    if (key < N) {
      kv = *reinterpret_cast<const u32x4*>(Kb + (int64_t)key * ldk + ch * 8);
      vv = *reinterpret_cast<const u32x4*>(Vb + (int64_t)key * ldv + ch * 8);
    }
    *reinterpret_cast<u32x4*>(Ks + key * A64_KROW + ch * 16) = kv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint16_t val = (uint16_t)((vv[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
      *reinterpret_cast<uint16_t*>(Vt + (ch * 8 + e) * vt_stride + key * 2) = val;
    }
  }
  __syncthreads();

  const int qrow = blockIdx.y * 128 + wave * 32 + l31;
  const int qrow_c = qrow < N ? qrow : N - 1;
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qrow_c * ldq + s * 16 + hi * 8);

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;
  const float* bias_row = rel_bias ? rel_bias + (int64_t)h * (2 * N - 1) + (N - 1) - qrow_c : nullptr;   // + key

  for (int kb = 0; kb < NP / 32; ++kb) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kb * 32 + l31) * A64_KROW + (2 * st + hi) * 16);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
    }
    // scores in natural units: scale * q.k + bias; masked keys -> -inf
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = s[r] * scale;
      if (bias_row) v += bias_row[key < N ? key : N - 1];
      if (key >= N || (causal && key > qrow_c)) v = -INFINITY;
      s[r] = v * LOG2E;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;     // a fully masked block leaves everything at zero weight
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_use);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    bf16x8 pf[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) { pf[0][e] = (__bf16)s[e]; pf[1][e] = (__bf16)s[8 + e]; }
    // O^T[d][q] += V^T[d][keys] P^T: k-slot j of lane half hi <-> key 16 ks + (j & 3) + 8 (j >> 2) + 4 hi (the
    // accumulator's own row order, so P feeds the MFMA straight from the score registers)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const char* vr = Vt + (db * 32 + l31) * vt_stride + (kb * 32 + 16 * ks + 4 * hi) * 2;
        const u32x2 lo = *reinterpret_cast<const u32x2*>(vr), hi2 = *reinterpret_cast<const u32x2*>(vr + 16);
        const u32x4 both = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, both), pf[ks], o[db], 0, 0, 0);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qrow < N) {
    bf16_t* orow = Ob + (int64_t)qrow * ldo + 4 * hi;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2 w;
        w[0] = pack_bf2(o[db][qd * 4 + 0] * inv, o[db][qd * 4 + 1] * inv);
        w[1] = pack_bf2(o[db][qd * 4 + 2] * inv, o[db][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + db * 32 + qd * 8) = w;
      }
  }
}

// T5LayerNorm (no mean subtraction, no bias): out = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) ); one wave per row.
template <typename TX>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const TX* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ out, int64_t ldo, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TX* xr = x + row * ldx;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    float v;
    if constexpr (sizeof(TX) == 2) v = bf2f(xr[c]); else v = xr[c];
    ss += v * v;
  }
  const float r = rsqrtf(wave_sum(ss) / (float)D + eps);
  for (int c = lane; c < D; c += 64) {
    float v;
    if constexpr (sizeof(TX) == 2) v = bf2f(xr[c]); else v = xr[c];
    out[row * ldo + c] = f2bf(bf2f(w[c]) * round_bf(v * r));
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                          bf16_t* __restrict__ out, int64_t n, int D, int64_t vocab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int cpr = D >> 3;
  if (i >= n * cpr) return;
  const int64_t r = i / cpr;
  const int c = (int)(i - r * cpr);
  int64_t id = ids[r];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  *reinterpret_cast<u32x4*>(out + r * D + c * 8) = *reinterpret_cast<const u32x4*>(table + id * D + c * 8);
}

// mode 0: x32 += y (bf16);  1: x32 += y (f32);  2: x32 = y (bf16 -> f32)
__global__ __launch_bounds__(256) void add_into_f32_kernel(float* __restrict__ x, const void* __restrict__ y, int64_t n, int mode) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) x[i] += bf2f(((const bf16_t*)y)[i]);
  else if (mode == 1) x[i] += ((const float*)y)[i];
  else x[i] = bf2f(((const bf16_t*)y)[i]);
}

// mode 0: out = a * b (bf16, row-strided operands: a[r * lda + c], b[r * ldb + c]);  mode 1: out = a * sigmoid(1.702 a)
__global__ __launch_bounds__(256) void mul_act_kernel(const bf16_t* __restrict__ a, int64_t lda, const bf16_t* __restrict__ b,
                                                      int64_t ldb, bf16_t* __restrict__ out, int64_t ldo, int64_t rows, int cols,
                                                      int mode) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols;
  const int c = (int)(i - r * cols);
  const float x = bf2f(a[r * lda + c]);
  float v;
  if (mode == 0) v = x * bf2f(b[r * ldb + c]);
  else v = x * round_bf(1.0f / (1.0f + __expf(-round_bf(1.702f * x))));   // x * sigmoid(1.702 * x), bf16 op by op
  out[r * ldo + c] = f2bf(v);
}

// ---------------------------------------------------------------------------------------------------------------------
int attention64(const AttnArgs& a, const float* rel_bias, int causal, hipStream_t st) {
  if (a.B <= 0 || a.H <= 0 || a.N <= 0) return 0;
  if (a.N > 512) return fail("attention64: at most 512 keys (the head's K and V live in LDS)");
  if ((a.ldq | a.ldk | a.ldv | a.q_bstride | a.k_bstride | a.v_bstride) % 8 || (a.ldo | a.o_bstride) % 4)
    return fail("attention64: strides must be multiples of 8 elements (q, k, v) / 4 (o)");
  if (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) % 16 || (uintptr_t)a.o % 8) return fail("attention64: alignment");
  const int NP = (a.N + 63) / 64 * 64;
  const int lds = NP * A64_KROW + 64 * 2 * (NP + 8);
  static int lds_set = 0;
  if (lds > lds_set) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)attn64_kernel);
    (void)hipGetLastError();
    if (hipFuncSetAttribute((const void*)attn64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess)
      return fail("attention64: cannot raise the dynamic LDS limit");
    lds_set = 160 * 1024;
  }
  attn64_kernel<<<dim3(a.B * a.H, (a.N + 127) / 128), 256, lds, st>>>(
      (const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, (bf16_t*)a.o, a.ldq, a.ldk, a.ldv, a.ldo, a.q_bstride,
      a.k_bstride, a.v_bstride, a.o_bstride, a.H, a.N, NP, a.scale, rel_bias, causal);
  return check_launch("attention64");
}

int rmsnorm(const void* x, int x_dtype, int64_t ldx, const void* w, void* out, int64_t ldo, int64_t rows, int D, float eps,
            hipStream_t st) {
  if (rows <= 0 || D <= 0) return 0;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (x_dtype == 0) rmsnorm_kernel<float><<<grid, 256, 0, st>>>((const float*)x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, rows, D, eps);
  else if (x_dtype == 1) rmsnorm_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, rows, D, eps);
  else return fail("rmsnorm: x dtype must be 0 (f32) or 1 (bf16)");
  return check_launch("rmsnorm");
}

int gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int D, int64_t vocab, hipStream_t st) {
  if (D % 8) return fail("gather_rows: D must be a multiple of 8");
  if (n <= 0) return 0;
  const int64_t total = n * (D / 8);
  gather_rows_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const bf16_t*)table, ids, (bf16_t*)out, n, D, vocab);
  return check_launch("gather_rows");
}

int add_into_f32(float* x, const void* y, int64_t n, int mode, hipStream_t st) {
  if (n <= 0) return 0;
  if (mode < 0 || mode > 2) return fail("add_into_f32: mode 0..2");
  add_into_f32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, y, n, mode);
  return check_launch("add_into_f32");
}

int mul_act(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int64_t rows, int cols, int mode,
            hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (mode == 0 && !b) return fail("mul_act: second operand is null");
  mul_act_kernel<<<(unsigned)((rows * cols + 255) / 256), 256, 0, st>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out,
                                                                        ldo, rows, cols, mode);
  return check_launch("mul_act");
}

}  // namespace tfx
