// Shared device helpers for the TextFlux gfx950 kernels (bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tfx {

typedef uint16_t bf16_t;  // raw bf16 bits in global / LDS memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// bf16 <-> fp32.  to_bf16 is round-to-nearest-even (hardware v_cvt_pk_bf16_f32 on gfx950), the same
// rounding torch applies on `.to(torch.bfloat16)`.
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }
// round two values to bf16 precision: one conversion + two unpacks (round_bf twice: two conversions + two shifts)
__device__ __forceinline__ void round_bf2(float& a, float& b);
// two values -> one v_cvt_pk_bf16_f32 (the scalar form above costs a conversion per value plus an OR to combine them)
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ void round_bf2(float& a, float& b) {
  const uint32_t p = pack_bf2(a, b);
  a = __uint_as_float(p << 16);
  b = __uint_as_float(p & 0xffff0000u);
}
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return v;
}

// tanh-approximated GELU, F.gelu(x, approximate="tanh") (reference: D/models/activations.py:83 via
// nn.GELU(approximate="tanh")):  0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3).
// Evaluated as x * sigmoid(2u) = x / (1 + 2^(-2u log2 e)) with the hardware v_exp_f32 / v_rcp_f32 (1 ulp each, far
// below the bf16 output rounding); ocml tanhf costs ~4x more VALU in the GEMM epilogue.
__device__ __forceinline__ float gelu_tanh(float x) {
  // -2 u log2(e) = x * (c0 + c1 x^2): 3 multiply-adds in front of the two transcendental ops (each costs two VALU slots, tools/ubench/valu_rate)
  const float c0 = -2.3022081981443252f, c1 = -0.10294323958002349f;
  const float z = x * __builtin_fmaf(x * x, c1, c0);
  const float e = __builtin_amdgcn_exp2f(z);  // exp(-2u)
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace tfx
