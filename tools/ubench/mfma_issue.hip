// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 with accumulators in ArchVGPRs vs AccVGPRs, chained vs
// interleaved, 1 or 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_issue.hip -o mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8_V(A0, A1) \
  "v_mfma_f32_32x32x16_bf16 v[" A0 "], v[0:3], v[4:7], v[" A0 "]\n" \
  "v_mfma_f32_32x32x16_bf16 v[" A1 "], v[0:3], v[8:11], v[" A1 "]\n"
#define REP8_A(A0, A1) \
  "v_mfma_f32_32x32x16_bf16 a[" A0 "], v[0:3], v[4:7], a[" A0 "]\n" \
  "v_mfma_f32_32x32x16_bf16 a[" A1 "], v[0:3], v[8:11], a[" A1 "]\n"

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  // MODE 0: VGPR acc, 8 independent accumulators round-robin; 1: AGPR acc, same; 2: VGPR acc, 2 accumulators (chains)
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)
      asm volatile(REP8_V("64:79", "80:95") REP8_V("96:111", "112:127") REP8_V("128:143", "144:159") REP8_V("160:175", "176:191")
                   ::: "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",
                       "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95",
                       "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111",
                       "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127",
                       "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143",
                       "v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159",
                       "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175",
                       "v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191",
                       "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11");
    else if (MODE == 1)
      asm volatile(REP8_A("0:15", "16:31") REP8_A("32:47", "48:63") REP8_A("64:79", "80:95") REP8_A("96:111", "112:127")
                   ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
                       "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31",
                       "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47",
                       "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63",
                       "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79",
                       "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95",
                       "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111",
                       "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127",
                       "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11");
    else
      asm volatile(REP8_V("64:79", "80:95") REP8_V("64:79", "80:95") REP8_V("64:79", "80:95") REP8_V("64:79", "80:95")
                   ::: "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",
                       "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95",
                       "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11");
  }
  if (out && iters < 0) out[threadIdx.x] = 1.f;
}

template <int MODE>
void run(const char* name, int threads) {
  const int iters = 20000, blocks = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, threads>>>(nullptr, 100);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<blocks, threads>>>(nullptr, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double mfma = (double)iters * 8 * (threads / 64) * blocks;
  const double tf = mfma * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-44s threads/block %3d: %.3f ms  %.0f TFLOP/s (zero operands)\n", name, threads, ms, tf);
}

int main() {
  run<0>("VGPR acc, 8 independent accumulators", 256);
  run<1>("AGPR acc, 8 independent accumulators", 256);
  run<2>("VGPR acc, 2 accumulators (dependent pairs)", 256);
  run<0>("VGPR acc, 8 independent accumulators", 512);
  run<1>("AGPR acc, 8 independent accumulators", 512);
  run<2>("VGPR acc, 2 accumulators (dependent pairs)", 512);
  return 0;
}
