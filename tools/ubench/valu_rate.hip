// Issue rate of VALU instruction classes on gfx950 (cycles per wave instruction at 1 and 2 waves per SIMD): decides how the GEMM
// epilogue's GELU (v_exp_f32 + v_rcp_f32 per element) is priced.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 1) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 2) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 3) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));) }
    if (OP == 4) { REP16(asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %4, %4, %4\n v_exp_f32 %1, %1\n v_mul_f32 %5, %5, %5\n v_exp_f32 %2, %2\n v_mul_f32 %6, %6, %6\n v_exp_f32 %3, %3\n v_mul_f32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 5) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %6, %6, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP>
static void run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
  for (int threads : {256, 512}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<OP><<<256, threads>>>(out, 100, cyc);
    hipEventRecord(e0);
    k<OP><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_wave_instr\": %.3f, \"memtime_ticks_per_instr\": %.3f}\n", name, threads / 256, ms * 1e6 / n, (double)c / n);
  }
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0>("v_mul_f32", 128, out, cyc); run<1>("v_exp_f32", 128, out, cyc); run<2>("v_rcp_f32", 128, out, cyc);
  run<3>("v_pk_mul_f32", 64, out, cyc); run<4>("v_exp_f32+v_mul_f32 interleaved", 128, out, cyc); run<5>("v_cvt_pk_bf16_f32", 64, out, cyc);
  return 0;
}
