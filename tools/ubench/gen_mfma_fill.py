"""Generates mfma_fill.hip: how many single-issue instructions hide in the gap between two v_mfma_f32_32x32x16_bf16 of ONE
wave (1 or 2 waves per SIMD), for three MFMA orders:
  rr    8 independent accumulators round-robin
  alt   one dependent chain (S) alternating with 4 independent accumulators (O0..O3)  -- attention-like
  chain 8 MFMAs on one accumulator, then 8 on independent ones, fillers in every gap
usage: python gen_mfma_fill.py > mfma_fill.hip && hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill"""
FILL = ["v_exp_f32 v{d}, v{s}", "v_fma_f32 v{d}, v{s}, v{s}, v{s2}", "v_cvt_pk_bf16_f32 v{d}, v{s}, v{s2}", "v_exp_f32 v{d}, v{s}",
        "v_max3_f32 v{d}, v{s}, v{s2}, v{s}", "v_fma_f32 v{d}, v{s}, v{s}, v{s2}", "v_exp_f32 v{d}, v{s}", "v_add_f32 v{d}, v{s}, v{s2}"]
def acc(i, agpr):
    return f"a[{16*i}:{16*i+15}]" if agpr else f"v[{64+16*i}:{64+16*i+15}]"
def body(order, nfill, kind):
    seq = {"rr": [0, 1, 2, 3, 4, 5, 6, 7] * 2, "alt": [0, 1, 0, 2, 0, 3, 0, 4] * 2, "chain": [0] * 8 + [1, 2, 3, 4, 5, 6, 7, 1]}[order]
    out, fi = [], 0
    for a in seq:
        agpr = a != 0          # accumulator 0 (the S chain) lives in arch VGPRs, the others in AGPRs
        out.append(f"v_mfma_f32_32x32x16_bf16 {acc(a, agpr)}, v[0:3], v[4:7], {acc(a, agpr)}")
        for _ in range(nfill):
            if kind == "mix":
                t = FILL[fi % len(FILL)]
            elif kind == "exp":
                t = FILL[0]
            elif kind == "lds":
                t = "ds_read_b128 v[{d4}:{d4e}], v12" if fi % 2 == 0 else FILL[1]
            else:
                t = FILL[1]
            d = 20 + fi % 8
            out.append(t.format(d=d, s=30 + fi % 6, s2=36 + fi % 4, d4=40 + 4 * (fi % 4), d4e=43 + 4 * (fi % 4)))
            fi += 1
    return out
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n")
cases = []
for order in ("rr", "alt", "chain"):
    for kind in ("mix", "exp", "fma", "lds"):
        for nf in (0, 2, 3, 4, 5, 6, 8):
            if nf == 0 and kind != "mix":
                continue
            name = f"k_{order}_{kind}_{nf}"
            cases.append((name, order, kind, nf))
            lines = body(order, nf, kind)
            asm = "\n".join(f'      "{l}\\n"' for l in lines)
            clob = ",".join([f'"v{i}"' for i in range(0, 80)] + [f'"a{i}"' for i in range(16, 128)])
            print(f"""__global__ __launch_bounds__(512) void {name}(long long* out, int iters) {{
  extern __shared__ char smem[];
  asm volatile("v_mov_b32 v12, 0\\n" ::: "v12");
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {{
    asm volatile(
{asm}
      "s_waitcnt lgkmcnt(0)\\n" ::: {clob}, "memory");
  }}
  long long t1 = __builtin_readcyclecounter();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}}
""")
print("""template <class K> void run(K kern, const char* name, int threads, int nf) {
  const int iters = 4000, blocks = 256;
  long long* d; hipMalloc(&d, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  kern<<<blocks, threads, 65536>>>(d, 50);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kern<<<blocks, threads, 65536>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 16;
  const double wps = threads / 256.0;
  printf("%-22s waves/SIMD %.0f  fillers/gap %d: %6.1f clk/MFMA/wave  %6.1f clk/MFMA/SIMD  wall %.0f TF/s\\n", name, wps, nf,
         cyc / mf, cyc / mf / wps, mf * (threads / 64) * blocks * 32768.0 / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {""")
for name, order, kind, nf in cases:
    print(f'  run({name}, "{order}/{kind}", 256, {nf});')
for name, order, kind, nf in cases:
    if kind == "mix":
        print(f'  run({name}, "{order}/{kind}", 512, {nf});')
print("  return 0;\n}")
