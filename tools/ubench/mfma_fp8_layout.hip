// Operand layout check of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit scales) and of
// v_cvt_pk_fp8_f32 on gfx950: D[i][j] = sum_k A[i][k] * B[j][k], lane l supplies row (l & 31), bytes k = (l >> 5) * 32 .. +32.
// hipcc --offload-arch=gfx950 tools/ubench/mfma_fp8_layout.hip -o gpurun_out/mfma_fp8_layout && gpurun_out/mfma_fp8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static float e4m3(uint8_t v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m / 8.f, -6);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.f + (float)m / 8.f, e - 7);
  return s ? -r : r;
}

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, const float* xs, uint32_t* q) {
  const int l = threadIdx.x;
  i32x8 a, b;
  const int* ap = reinterpret_cast<const int*>(A + (l & 31) * 64 + (l >> 5) * 32);
  const int* bp = reinterpret_cast<const int*>(B + (l & 31) * 64 + (l >> 5) * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // C/D layout of the 32x32 MFMAs: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
  // fp8 conversion: 4 floats -> one dword
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(xs[l * 4 + 0], xs[l * 4 + 1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(xs[l * 4 + 2], xs[l * 4 + 3], w, true);
  q[l] = w;
}

int main() {
  std::vector<uint8_t> A(32 * 64), B(32 * 64);
  srand(1);
  for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
  for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
  std::vector<float> xs(256);
  for (int i = 0; i < 256; ++i) xs[i] = (i % 7 == 0 ? 1000.f : 1.f) * ((rand() % 2001) - 1000) / 37.f;
  uint8_t *dA, *dB; float *dD, *dx; uint32_t* dq;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 1024 * 4); hipMalloc(&dx, 1024); hipMalloc(&dq, 256);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipMemcpy(dx, xs.data(), 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD, dx, dq);
  std::vector<float> D(1024); std::vector<uint32_t> q(64);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, 256, hipMemcpyDeviceToHost);
  // hypothesis: D[i][j] with i = W-operand (first) row, j = second operand row
  double worst = 0, worstT = 0, mag = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0;
      for (int kk = 0; kk < 64; ++kk) ref += (double)e4m3(A[i * 64 + kk]) * e4m3(B[j * 64 + kk]);
      worst = fmax(worst, fabs(ref - D[i * 32 + j]));
      worstT = fmax(worstT, fabs(ref - D[j * 32 + i]));
      mag = fmax(mag, fabs(ref));
    }
  printf("mfma 32x32x64 fp8: max|err| D[i][j] (A row = D row) %.4g ; transposed %.4g ; max|ref| %.4g\n", worst, worstT, mag);
  int bad = 0; double worst_rel = 0;
  for (int i = 0; i < 256; ++i) {
    const uint8_t b = (q[i / 4] >> (8 * (i % 4))) & 0xff;
    const float got = e4m3(b), x = xs[i];
    const float sat = fminf(fmaxf(x, -448.f), 448.f);
    const double rel = fabs(got - sat) / fmax(fabs(sat), 1e-3);
    if (!(rel <= 0.0626)) { if (bad < 5) printf("cvt: x %.4f -> 0x%02x = %.4f\n", x, b, got); ++bad; }
    worst_rel = fmax(worst_rel, rel);
  }
  printf("cvt_pk_fp8_f32: %d of 256 outside 2^-4 relative (saturating at 448), worst rel %.4f\n", bad, worst_rel);
  return 0;
}
