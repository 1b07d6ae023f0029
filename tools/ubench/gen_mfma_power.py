#!/usr/bin/env python3
"""Generates tools/ubench/mfma_power.hip: register-only MFMA loops with EXPLICIT register allocation (inline asm), to measure
what a bf16 MFMA costs in energy by instruction shape, wave tile, issue order and accumulator register file.

Why: GEMM and attention sit at the board's power cap on random data, so time = dynamic energy / (cap - idle) and
the instruction mix that does the same FLOPs for fewer joules is the faster one.  hipcc's own allocation of such loops moves
accumulators between the register files every iteration, hence the asm.

Each variant: one 4- or 8-wave workgroup per CU; every wave loads its operand fragments once (random N(0,1) bf16, or zeros),
then runs `tiles` x 48 K-tiles of MFMAs (accumulators re-zeroed per tile, like a K = 3072 GEMM tile).
tools/mfma_power.py runs the binary per variant and samples rocm-smi alongside."""
import os

VARIANTS = [
    # name, shape, MI, NJ, order, acc file, waves per workgroup
    ("32x32x16 wave128x64  chain4  accV 8w (= gemm8pp)", 32, 4, 2, "chain", "v", 8),
    ("32x32x16 wave128x64  chain4  accV 4w", 32, 4, 2, "chain", "v", 4),
    ("32x32x16 wave128x64  chain4  accA 4w", 32, 4, 2, "chain", "a", 4),
    ("32x32x16 wave128x64  Astat   accV 4w", 32, 4, 2, "astat", "v", 4),
    ("32x32x16 wave128x64  Bstat   accV 4w", 32, 4, 2, "bstat", "v", 4),
    ("32x32x16 wave128x128 chain4  accA 4w", 32, 4, 4, "chain", "a", 4),
    ("32x32x16 wave128x128 Astat   accA 4w", 32, 4, 4, "astat", "a", 4),
    ("16x16x32 wave128x64  chain2  accV 4w", 16, 8, 4, "chain", "v", 4),
    ("16x16x32 wave128x64  Astat   accV 4w", 16, 8, 4, "astat", "v", 4),
    ("16x16x32 wave128x64  Astat   accA 4w", 16, 8, 4, "astat", "a", 4),
    ("16x16x32 wave128x128 Astat   accA 4w", 16, 8, 8, "astat", "a", 4),
    # the same 128-MFMA K-tile with the memory instructions a one-wave-per-SIMD 256x256x64 GEMM main loop would carry beside it
    # (per wave and K-tile: 32 fragment reads, 16 LDS-DMA requests of 1 KiB, one barrier): does the matrix pipe stay fed?
    ("16x16x32 wave128x128 + 32 ds_read_b128 / K-tile", 16, 8, 8, "astat", "a", 4, 32, 0, 0),
    ("16x16x32 wave128x128 + 16 LDS-DMA / K-tile", 16, 8, 8, "astat", "a", 4, 0, 16, 0),
    ("16x16x32 wave128x128 + 32 ds_read + 16 LDS-DMA + barrier / K-tile", 16, 8, 8, "astat", "a", 4, 32, 16, 1),
    ("16x16x32 wave128x128 + 32 ds_read + 16 LDS-DMA (all in 2nd half) + barrier", 16, 8, 8, "astat", "a", 4, 32, 16, 2),
    ("16x16x32 wave128x128 + 32 ds_read + 16 buffer_load->VGPR + 16 ds_write_b128 + barrier", 16, 8, 8, "astat", "a", 4, 32, 16, 3),
    # the shipped structure (gemm8pp): 8 waves = 2 row groups ping-ponging one barrier apart, wave tile 128x64; per wave and K-tile 4 x
    # (loads section: 6 fragment reads + 2 LDS-DMA requests | barrier | 16 MFMAs | barrier)
    ("16x16x32 wave128x64 8w ping-pong: 4 x (6 ds_read + 2 LDS-DMA | barrier | 16 MFMA | barrier)", 16, 8, 4, "astat", "v", 8, 24, 8, 4),
]


def kernel(idx, shape, MI, NJ, order, accf, waves, n_ds=0, n_dma=0, mode=0):
    KS = 4 if shape == 32 else 2                 # k-steps per 64-deep K-tile
    accw = 16 if shape == 32 else 4
    mn = "v_mfma_f32_32x32x16_bf16" if shape == 32 else "v_mfma_f32_16x16x32_bf16"
    # operand fragments: 4 VGPRs each.  v0..v1 = address; A at v[8..), B after
    a0 = 8
    A = lambda j, kk: a0 + 4 * (j * KS + kk)
    b0 = a0 + 4 * NJ * KS
    B = lambda i, kk: b0 + 4 * (i * KS + kk)
    op_end = b0 + 4 * MI * KS
    nacc = MI * NJ * accw
    acc0 = 0 if accf == "a" else (op_end + 7) // 8 * 8
    C = lambda i, j: acc0 + accw * (i * NJ + j)
    assert (accf == "a" and nacc <= 256 and op_end <= 256) or (accf == "v" and acc0 + nacc <= (256 if waves == 8 else 512)), (idx, acc0, nacc)
    rng = lambda f, r, n: f"{f}[{r}:{r + n - 1}]"
    L = []
    nfrag = (NJ + MI) * KS
    for f in range(nfrag):
        L.append(f"global_load_dwordx4 v[{a0 + 4 * f}:{a0 + 4 * f + 3}], %0, off offset:{16 * f}")
    L.append("s_waitcnt vmcnt(0)")
    if n_ds or n_dma:
        L.append("v_mov_b32 v2, %2")             # per-lane LDS byte address (lane * 16)
        L.append("v_mov_b32 v4, %3")             # global source of the DMA / staging loads (this wave's operand block)
        L.append("v_mov_b32 v5, %4")
        for k in range(8):
            L.append(f"s_add_u32 s{22 + k}, %5, {k * 2048}")    # M0 values: 8 x 2 KiB slots of this wave's LDS slice
        L.append("s_nop 4")
    if mode == 4:
        L.append("s_cmp_lt_u32 %6, 4")           # waves 4..7: the second row group starts one barrier late
        L.append("s_cbranch_scc1 9f")
        L.append("s_barrier")
        L.append("9:")
    L.append("s_mov_b32 s20, %1")                # tiles
    L.append("1:")
    for r in range(nacc):
        L.append(f"v_accvgpr_write_b32 a{acc0 + r}, 0" if accf == "a" else f"v_mov_b32 v{acc0 + r}, 0")
    L.append("s_mov_b32 s21, 48")
    L.append("s_nop 7")
    L.append("2:")
    seq = []
    if order == "chain":
        if shape == 32:
            for j in range(NJ):
                for i in range(MI):
                    for kk in range(KS):
                        seq.append((i, j, kk))
        else:        # two accumulators interleaved so that a dependent pair is never back to back on the 4-pass MFMA
            for j in range(NJ):
                for i in range(0, MI, 2):
                    for kk in range(KS):
                        seq.append((i, j, kk))
                        seq.append((i + 1, j, kk))
    elif order == "astat":
        for kk in range(KS):
            for j in range(NJ):
                for i in range(MI):
                    seq.append((i, j, kk))
    else:
        for kk in range(KS):
            for i in range(MI):
                for j in range(NJ):
                    seq.append((i, j, kk))
    # fillers: ds_read_b128 into scratch VGPRs v[240:255] (rotating), LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction
    # into this wave's 16 KiB LDS slice, source = the wave's operand block, L2-resident), spread evenly over the MFMAs
    nm = len(seq)
    fill = {}
    if n_ds or n_dma:
        assert op_end <= 224 and (mode != 4 or acc0 + nacc <= 232)
        ds_at = [int((k + 0.5) * nm / n_ds) for k in range(n_ds)] if n_ds else []
        if mode == 2:
            dma_at = [nm // 2 + int((k + 0.5) * (nm // 2) / n_dma) for k in range(n_dma)] if n_dma else []
        else:
            dma_at = [int((k + 0.25) * nm / n_dma) for k in range(n_dma)] if n_dma else []
        for k, pos in enumerate(ds_at):
            fill.setdefault(pos, []).append(f"ds_read_b128 v[{240 + 4 * (k % 4)}:{243 + 4 * (k % 4)}], v2 offset:{(k % 16) * 1024}")
        for k, pos in enumerate(dma_at):
            if mode == 3:     # register staging: load to VGPRs now, write the piece loaded 8 requests ago to LDS
                r = 224 + 4 * (k % 4)
                fill.setdefault(pos, []).append(f"ds_write_b128 v2, v[{r}:{r + 3}] offset:{16384 + (k % 16) * 1024}")
                fill.setdefault(pos, []).append(f"global_load_dwordx4 v[{r}:{r + 3}], v[4:5], off offset:{(k % 4) * 1024}")
            else:
                fill.setdefault(pos, []).append(f"s_mov_b32 m0, s{22 + (k % 8)}")
                fill.setdefault(pos, []).append(f"global_load_lds_dwordx4 v[4:5], off offset:{(k % 4) * 1024}")
    if mode == 4:
        # group 1 (waves 4..7) runs one barrier behind group 0: it executes one extra barrier before the loop (s30 = wave >> 2)
        sec = nm // 4
        for q in range(4):
            for k in range(n_ds // 4):
                L.append(f"ds_read_b128 v[{232 + 4 * (k % 4)}:{235 + 4 * (k % 4)}], v2 offset:{((q * 6 + k) % 16) * 1024}")
            for k in range(n_dma // 4):
                L.append(f"s_mov_b32 m0, s{22 + ((q * 2 + k) % 8)}")
                L.append(f"global_load_lds_dwordx4 v[4:5], off offset:{((q * 2 + k) % 4) * 1024}")
            L.append("s_waitcnt vmcnt(6)")
            L.append("s_barrier")
            L.append("s_setprio 1")
            L.append("s_waitcnt lgkmcnt(0)")
            for (i, j, kk) in seq[q * sec:(q + 1) * sec]:
                c = rng(accf, C(i, j), accw)
                L.append(f"{mn} {c}, {rng('v', A(j, kk), 4)}, {rng('v', B(i, kk), 4)}, {c}")
            L.append("s_setprio 0")
            L.append("s_barrier")
        seq_emit = []
    else:
        seq_emit = seq
    for n_, (i, j, kk) in enumerate(seq_emit):
        if mode in (1, 2, 3) and n_ == nm // 2:
            L.append("s_waitcnt vmcnt(%d)" % (4 if mode == 3 else 0))
            L.append("s_waitcnt lgkmcnt(0)")
            L.append("s_barrier")
        for f in fill.get(n_, []):
            L.append(f)
        c = rng(accf, C(i, j), accw)
        L.append(f"{mn} {c}, {rng('v', A(j, kk), 4)}, {rng('v', B(i, kk), 4)}, {c}")
    if n_ds or n_dma:
        L.append("s_waitcnt lgkmcnt(0)")
    L.append("s_sub_u32 s21, s21, 1")
    L.append("s_cmp_lg_u32 s21, 0")
    L.append("s_cbranch_scc1 2b")
    L.append("s_nop 15")
    L.append("s_nop 15")
    L.append("s_sub_u32 s20, s20, 1")
    L.append("s_cmp_lg_u32 s20, 0")
    L.append("s_cbranch_scc1 1b")
    if mode == 4:
        L.append("s_cmp_lt_u32 %6, 4")
        L.append("s_cbranch_scc0 8f")
        L.append("s_barrier")
        L.append("8:")
    clob = [f"v{r}" for r in range(a0, op_end)] + [f"{accf}{acc0 + r}" for r in range(nacc)] + ["s20", "s21", "scc"]
    mix = bool(n_ds or n_dma)
    if mix:
        clob += ["v2", "v4", "v5"] + [f"v{r}" for r in (range(232, 248) if mode == 4 else range(224, 256))] + [f"s{22 + k}" for k in range(8)]
    body = "\\n\\t".join(L)
    clobs = ", ".join(f'"{c}"' for c in clob)
    flops_per_ktile = len(seq) * (32 * 32 * 16 * 2 if shape == 32 else 16 * 16 * 32 * 2)
    if mix:
        src = f"""
__global__ __launch_bounds__({64 * waves}) void k{idx}(const char* src, int tiles) {{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const char* p = src + (size_t)(blockIdx.x * {64 * waves} + threadIdx.x) * {16 * nfrag};
  const unsigned lane16 = (threadIdx.x & 63) * 16;
  // the DMA / staging source: 1 KiB contiguous per wave instruction (lane * 16 B), as a real operand stream (8 lanes per 128-B line)
  const unsigned long long gp = (unsigned long long)(src + (size_t)(blockIdx.x * {waves} + (threadIdx.x >> 6)) * 8192 + lane16);
  const unsigned lo = (unsigned)gp, hi = (unsigned)(gp >> 32);
  const unsigned wbase = __builtin_amdgcn_readfirstlane(32768u + (threadIdx.x >> 6) * 16384u + (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
  const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  asm volatile("{body}" :: "v"(p), "s"(tiles), "v"(lane16), "v"(lo), "v"(hi), "s"(wbase), "s"(wv) : {clobs}, "memory");
}}
"""
    else:
        src = f"""
__global__ __launch_bounds__({64 * waves}) void k{idx}(const char* src, int tiles) {{
  const char* p = src + (size_t)(blockIdx.x * {64 * waves} + threadIdx.x) * {16 * nfrag};
  asm volatile("{body}" :: "v"(p), "s"(tiles) : {clobs}, "memory");
}}
"""
    return src, flops_per_ktile, nfrag


def main():
    out = ['// GENERATED by tools/ubench/gen_mfma_power.py -- do not edit.  hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power',
           '#include <hip/hip_runtime.h>', '#include <chrono>', '#include <cstdio>', '#include <cstdlib>', '#include <cstring>',
           '#include <random>', '#include <vector>']
    table = []
    for idx, var in enumerate(VARIANTS):
        name, shape, MI, NJ, order, accf, waves = var[:7]
        src, fl, nfrag = kernel(idx, shape, MI, NJ, order, accf, waves, *var[7:])
        out.append(src)
        table.append((name, idx, fl, nfrag, waves, 131072 if len(var) > 7 else 0))
    out.append("struct V { const char* name; void (*fn)(const char*, int); double flops_ktile; int nfrag, waves, lds; };")
    out.append("static const V vs[] = {" + ", ".join(f'{{"{n}", k{i}, {fl}.0, {nf}, {w}, {l}}}' for n, i, fl, nf, w, l in table) + "};")
    out.append(r'''
int main(int argc, char** argv) {
  const int nv = sizeof(vs) / sizeof(vs[0]);
  if (argc < 2) { for (int i = 0; i < nv; ++i) printf("%d %s\n", i, vs[i].name); return 0; }
  const int v = atoi(argv[1]);
  const double secs = argc > 2 ? atof(argv[2]) : 2.0;
  const bool zero = argc > 3 && argv[3][0] == 'z';
  if (v < 0 || v >= nv) return 1;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t n = (size_t)cus * 64 * vs[v].waves * vs[v].nfrag * 8;     // bf16 elements
  std::vector<uint16_t> h(n);
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < n; ++i) {
    const float f = zero ? 0.f : nd(rng);
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    h[i] = (uint16_t)(u >> 16);
  }
  char* d;
  (void)hipMalloc(&d, n * 2 + 65536);     // the DMA / staging fillers read up to 4 KiB past a lane's operand block
  (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  const int tiles = 100;
  if (vs[v].lds) (void)hipFuncSetAttribute((const void*)vs[v].fn, hipFuncAttributeMaxDynamicSharedMemorySize, vs[v].lds);
  vs[v].fn<<<cus, 64 * vs[v].waves, vs[v].lds>>>(d, 2);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double tot_ms = 0, launches = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; ++r) vs[v].fn<<<cus, 64 * vs[v].waves, vs[v].lds>>>(d, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    tot_ms += ms; launches += 4;
  }
  const double flops = vs[v].flops_ktile * 48.0 * tiles * vs[v].waves * cus * launches;
  printf("{\"variant\": %d, \"name\": \"%s\", \"data\": \"%s\", \"tflops\": %.1f, \"kernel_s\": %.3f, \"flops\": %.6e}\n", v, vs[v].name,
         zero ? "zero" : "N(0,1)", flops / (tot_ms * 1e-3) / 1e12, tot_ms * 1e-3, flops);
  return 0;
}
''')
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "mfma_power.hip"), "w") as f:
        f.write("\n".join(out))
    print("wrote mfma_power.hip with", len(VARIANTS), "variants")


if __name__ == "__main__":
    main()
