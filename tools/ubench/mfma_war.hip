// Does an issued v_mfma read its A / B operands at issue, or when it starts executing?  A test MFMA (ones x ones, k = 16 ->
// every output 16) is issued behind NPRE independent MFMAs that keep the pipe busy and is IMMEDIATELY followed by an
// instruction that overwrites one of its source registers with twos (-> outputs > 16 if the overwritten value was used).
// hipcc --offload-arch=gfx950 -O3 mfma_war.hip -o mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>

#define PRE "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n" \
            "v_mfma_f32_32x32x16_bf16 a[16:31], v[20:23], v[24:27], a[16:31]\n"
#define ONES "0x3f803f80"
#define TWOS "0x40004000"

template <int MODE>
__global__ void k(float* out) {
  float r0, r1;
  // v[0:3] = A (ones), v[4:7] = B (ones), a[40:43] = A' (ones, AGPR copy), v[20:27] = zeros for the filler MFMAs
  asm volatile(
      "v_mov_b32 v0, " ONES "\n v_mov_b32 v1, " ONES "\n v_mov_b32 v2, " ONES "\n v_mov_b32 v3, " ONES "\n"
      "v_mov_b32 v4, " ONES "\n v_mov_b32 v5, " ONES "\n v_mov_b32 v6, " ONES "\n v_mov_b32 v7, " ONES "\n"
      "v_mov_b32 v8, " TWOS "\n"
      "v_accvgpr_write_b32 a40, v0\n v_accvgpr_write_b32 a41, v0\n v_accvgpr_write_b32 a42, v0\n v_accvgpr_write_b32 a43, v0\n"
      "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
      "s_nop 15\n s_nop 15\n"
      PRE PRE PRE PRE
      ".if %2 == 0\n"   // VGPR source, VALU overwrite right behind
      "v_mfma_f32_32x32x16_bf16 v[40:55], v[0:3], v[4:7], 0\n"
      "v_mov_b32 v0, v8\n v_mov_b32 v1, v8\n v_mov_b32 v2, v8\n v_mov_b32 v3, v8\n"
      ".elseif %2 == 1\n" // AGPR source (srcA), v_accvgpr_write right behind
      "v_mfma_f32_32x32x16_bf16 v[40:55], a[40:43], v[4:7], 0\n"
      "v_accvgpr_write_b32 a40, v8\n v_accvgpr_write_b32 a41, v8\n v_accvgpr_write_b32 a42, v8\n v_accvgpr_write_b32 a43, v8\n"
      ".elseif %2 == 2\n" // VGPR srcB, VALU overwrite
      "v_mfma_f32_32x32x16_bf16 v[40:55], v[0:3], v[4:7], 0\n"
      "v_mov_b32 v4, v8\n v_mov_b32 v5, v8\n v_mov_b32 v6, v8\n v_mov_b32 v7, v8\n"
      ".elseif %2 == 3\n" // VGPR source, overwrite after one more MFMA
      "v_mfma_f32_32x32x16_bf16 v[40:55], v[0:3], v[4:7], 0\n"
      "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n"
      "v_mov_b32 v0, v8\n v_mov_b32 v1, v8\n v_mov_b32 v2, v8\n v_mov_b32 v3, v8\n"
      ".elseif %2 == 4\n" // AGPR source, overwrite after one more MFMA
      "v_mfma_f32_32x32x16_bf16 v[40:55], a[40:43], v[4:7], 0\n"
      "v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]\n"
      "v_accvgpr_write_b32 a40, v8\n v_accvgpr_write_b32 a41, v8\n v_accvgpr_write_b32 a42, v8\n v_accvgpr_write_b32 a43, v8\n"
      ".elseif %2 == 5\n" // AGPR srcB, accvgpr_write right behind
      "v_mfma_f32_32x32x16_bf16 v[40:55], v[0:3], a[40:43], 0\n"
      "v_accvgpr_write_b32 a40, v8\n v_accvgpr_write_b32 a41, v8\n v_accvgpr_write_b32 a42, v8\n v_accvgpr_write_b32 a43, v8\n"
      ".endif\n"
      "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
      "v_mov_b32 %0, v40\n v_mov_b32 %1, v55\n"
      : "=v"(r0), "=v"(r1) : "n"(MODE)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v20","v21","v22","v23","v24","v25","v26","v27",
        "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55",
        "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
        "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a40","a41","a42","a43");
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = r0;
  out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = r1;
}

template <int MODE>
void run(const char* name) {
  const int blocks = 256, threads = 256, n = blocks * threads * 2;
  float* d; hipMalloc(&d, n * 4);
  k<MODE><<<blocks, threads>>>(d);
  static float h[256 * 256 * 2];
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; float mn = 1e30f, mxv = -1e30f;
  for (int i = 0; i < n; ++i) { if (h[i] != 16.f) ++bad; mn = h[i] < mn ? h[i] : mn; mxv = h[i] > mxv ? h[i] : mxv; }
  printf("%-70s: %d of %d outputs != 16 (min %.1f max %.1f)\n", name, bad, n, mn, mxv);
  hipFree(d);
}
int main() {
  run<0>("srcA in VGPRs, v_mov overwrites it right behind the MFMA");
  run<1>("srcA in AGPRs, v_accvgpr_write overwrites it right behind the MFMA");
  run<2>("srcB in VGPRs, v_mov overwrites it right behind the MFMA");
  run<3>("srcA in VGPRs, overwritten one MFMA later");
  run<4>("srcA in AGPRs, overwritten one MFMA later");
  run<5>("srcB in AGPRs, v_accvgpr_write overwrites it right behind the MFMA");
  return 0;
}
