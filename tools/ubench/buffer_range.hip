// Does the MUBUF range check of a raw buffer (stride 0) include the SGPR offset?  num_records = 1024 bytes over a 64 KiB
// allocation filled with 0x11111111; loads at (voffset, soffset) pairs inside / outside the 1024 bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ void k(const unsigned* p, unsigned* out) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 1024, 0x00020000);
  const int cases[6][2] = {{0, 0}, {2048, 0}, {0, 2048}, {512, 512}, {1008, 0}, {0, 1008}};
  for (int c = 0; c < 6; ++c) {
    u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, cases[c][0] + (int)threadIdx.x * 0, cases[c][1], 0));
    if (threadIdx.x == 0) out[c] = v[0];
  }
}
int main() {
  unsigned *d, *o, h[16384], r[6];
  for (auto& x : h) x = 0x11111111u;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  const char* names[6] = {"voffset 0    soffset 0    (inside)", "voffset 2048 soffset 0    (outside via VGPR)", "voffset 0    soffset 2048 (outside via SGPR)",
                          "voffset 512  soffset 512  (sum = num_records)", "voffset 1008 soffset 0    (last 16 bytes)", "voffset 0    soffset 1008 (last 16 bytes via SGPR)"};
  for (int c = 0; c < 6; ++c) printf("%-50s -> %08x %s\n", names[c], r[c], r[c] ? "(memory)" : "(zero: out of range)");
  return 0;
}
