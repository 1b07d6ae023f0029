"""Replays the instruction stream of one attention tile (4 pipeline steps of attn_w4_kernel, taken from hipcc's assembly)
in a loop with no memory traffic behind it: how many cycles does the ISSUE of that stream take?
usage: python gen_stream_replay.py <kernel.s> [drop-regex ...] > stream_replay.hip ; hipcc --offload-arch=gfx950 -O3 stream_replay.hip -o stream_replay
Lines matching a drop-regex are removed (ablations: 'v_exp', 'ds_read', 'v_mfma', ...)."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
drops = [re.compile(d) for d in sys.argv[2:]]
start = next(i for i, l in enumerate(src) if "Inner Loop Header" in l)
end = next(i for i in range(start, len(src)) if "s_barrier" in src[i])
body = []
for l in src[start + 1:end]:
    t = l.split(";")[0].strip()
    if not t or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("buffer_") or op.startswith("s_barrier") or op.startswith("scratch_"):
        continue
    if op == "s_waitcnt" and "vmcnt" in t:
        t = re.sub(r"vmcnt\(\d+\)\s*", "", t).strip()
        if t == "s_waitcnt":
            continue
    if op.startswith("ds_write"):
        continue            # staging writes left out: compute stream only
    if any(d.search(t) for d in drops):
        continue
    body.append(t)
n_mfma = sum(1 for t in body if t.startswith("v_mfma"))
asm = "\n".join(f'      "{t}\\n"' for t in body)
clob = ",".join([f'"v{i}"' for i in range(256)] + [f'"a{i}"' for i in range(256)] + ['"vcc"', '"scc"', '"memory"'] + [f'"s{i}"' for i in range(16, 100)])
print(f"""#include <hip/hip_runtime.h>
#include <cstdio>
// {len(body)} instructions, {n_mfma} MFMAs per iteration
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters) {{
  extern __shared__ char smem[];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {{
    asm volatile(
{asm}
      ::: {clob});
  }}
  long long t1 = __builtin_readcyclecounter();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}}
int main() {{
  long long* d; hipMalloc(&d, 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 101376);
  k<<<256, 256, 101376>>>(d, 20);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000;
  hipEventRecord(a);
  k<<<256, 256, 101376>>>(d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long cyc; hipMemcpy(&cyc, d, 8, hipMemcpyDeviceToHost);
  printf("%d instructions, %d MFMAs per tile: %.0f cycles per tile (MFMA-bound floor %d), wall %.3f us per tile\\n", {len(body)}, {n_mfma}, (double)cyc / iters, {n_mfma} * 32, ms * 1e3 / iters);
  return 0;
}}""")
