// bf16 (and e4m3) GEMM for the DiT linear layers:  C[b][M,N] = epi( A[b][M,K] @ W[N,K]^T + bias ).
//
// Three kernels:
//   gemm8pp_kernel  persistent MFMA kernel, what the DiT runs on (K % 128 == 0): see the comment above it;
//                   template flags select the e4m3 operand type (FP8) and the split-K work unit (SPLIT);
//   gemm8p_kernel   one tile per block: same tile / fragments / ping-pong, K % 64 == 0, and the implicit-GEMM 3x3
//                   convolution mode of the VAE (CONV);
//   gemm_generic_kernel  fp32 FMA fallback for any shape (unit-test configs) and on-device cross-check.
//
// Common to the MFMA kernels: 256x256x64 block tile, 8 waves (512 threads) as 2 row-groups x 4 column stripes (a wave owns
// 128 token rows x 64 output columns = 8 x 4 accumulators of 16 x 16), v_mfma_f32_16x16x32_bf16 with the operands swapped (MFMA
// "A" = weight rows, "B" = token rows) so that every lane ends up holding 4 consecutive output columns of one token row.
// Round 3 moved both kernels from v_mfma_f32_32x32x16_bf16 to the 16 x 16 shape: on random data every MFMA-dense kernel runs
// at the board's power limit, and the 16 x 16 MFMA does the same FLOPs for ~10 % fewer joules (register-only streams: 2.04-2.06
// PFLOP/s at 2.19 GHz against 1.77-1.82 at 1.92 GHz, tools/ubench/mfma_power, profiles/r03_mfma_power.json; per FLOP it moves
// 16 registers through the matrix pipe where 32 x 32 x 16 moves 20: the accumulator traffic halves) -- same LDS image, same
// fragment bytes, same schedule, +5..7 % sustained on every GEMM shape of the workload (profiles/r03_gemm_shapes.jsonl).
// Operand tiles go L2 -> LDS with LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR round trip); the 16-byte chunks of each
// 128-byte LDS row are XOR-swizzled with ((row>>1)&7) (applied on the per-lane *source* address, the LDS image stays
// lane-linear) so every ds_read_b128 lane group is bank-conflict free.  The two row-groups run a ping-pong schedule offset by
// one s_barrier: while one group's 4 waves (one per SIMD) issue 16 MFMAs, the other group reads its next fragments from LDS;
// waits on the operand requests are counted (s_waitcnt vmcnt(n)), never drained, inside the main loop.
// (tools/ubench/mfma_power variants 13 / 16 replay this main loop's instruction mix on registers only, next to the
// one-wave-per-SIMD 128 x 128-per-wave alternative VERDICT round 2 asked for: 1.81 vs 1.80 PFLOP/s -- the structures tie, so the
// ping-pong stayed.  hipBLASLt's 4-wave assembly kernel of the same tile was 3-5 % ahead on bias-only shapes until the epilogue /
// request-path work of round 3's second half -- branch-free buffer-descriptor accesses, tile-independent request offsets (no
// spills), the epilogue's drain as a builtin -- after which the 36864 x 9216 x 3072 bias shape measures 1491 vs 1467 TFLOP/s,
// 1.93 vs 1.97 J per launch, profiles/r03_power.json; DESIGN.md "Where a tile's time goes".)
//
// One-tile kernel schedule (slots = half phases; tile t, phase q in 0..3; group g runs loads(p) at slot 2p-1+g and
// MFMA(p) at slot 2p+g, p = 4t+q).  Each wave keeps X fragments for 64 rows and both 32-column W fragments in registers:
//   q0: read X_lo, W_lo | q1: read W_hi | q2: read X_hi | q3: no reads,
// so the last LDS read of tile t is X_lo, W_lo @q0, W_hi @q1, X_hi @q2.  With 2 buffer sets tile t+2 overwrites tile t,
// one 64-row quarter per slot, each in the first slot where every read of the overwritten bytes has been waited for
// (own lgkmcnt) and fenced (>= 1 barrier):
//   slot 8t+1 G0:X0_lo  +2 G1:W_lo_a  +3 G0:W_lo_b  +4 G1:X1_lo  +5 G0:W_hi_a  +6 G1:W_hi_b  +7 G0:X0_hi  +8 G1:X1_hi
// first readers: X0_lo/W_lo 8t+15, X1_lo 8t+16, W_hi 8t+17, X0_hi 8t+19, X1_hi 8t+20 -> >= 11 slots in flight; a request
// is waited for by its issuer (vmcnt(8): the 4 newest sections may still be in flight), then one barrier -> readable.
// See DESIGN.md "GEMM".
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "launch.h"

namespace tfx {

struct GemmParams {
  const bf16_t* A; int64_t lda, a_bs;
  const bf16_t* W; int64_t ldw, w_bs;
  const bf16_t* bias;
  bf16_t* C; int64_t ldc, c_bs;
  int M, N, K, batch, tm, tn, gm;
  int gelu_from;
#ifdef TFX_BENCH
  unsigned long long* timers;                       // bench only: s_memtime sums of wave 0 / wave 4 {K loop, epilogue wait, epilogue, tiles}
#endif
  const bf16_t* gate; int64_t gate_bs;
  const bf16_t* res; int64_t ldr, r_bs;
  int cin, inH, inW, oH, oW, cstride, cup, cpad;   // implicit 3x3 convolution (cin > 0), see GemmArgs
  int ckw, cstride_x;                               // taps per kernel row (3; 4 = the pixel-pair form) and the x stride (= cstride; 2 = pixel-pair form)
  int csh;                                          // log2(cin) when cin < 64 (narrow-input mode), else 0
  const bf16_t* zero;
  const float* a_scale; int64_t as_bs; const float* w_scale;   // fp8 kernel only
  // K-sliced work units (SPLIT kernels).  Units [0, u_full) are whole tiles with the normal epilogue; the last tail_r tiles of every
  // batch sample's tile order follow as (tile, slice) units, slice-major: sk slices of nt / sk K-tiles each, raw fp32 accumulators
  // to ws [slice][tail tile][256][256] (compact), finished by tail_reduce_kernel<EPI> (slices summed in order, then the epilogue).
  // tail_r = tiles per sample, u_full = 0: the whole GEMM is sliced (fewer tiles than CUs).  fp32-OUTPUT mode (ws_ld != 0, sk = 1,
  // every tile a "slice" unit): the raw accumulators go straight to ws = C with row stride ws_ld and batch stride ws_bs.
  float* ws; int sk, u_full, tail_r;
  int64_t ws_ld, ws_bs;
  // row-split weights (split_row > 0): tiles whose first row (inside the batch sample) lies below split_row -- the TEXT rows of the
  // joint [text | image] stream -- take the second weight set: W at byte offset w2_off from W (same descriptor), bias2 / gate2 /
  // nq_w2 / nk_w2 instead of bias / gate / nq_w / nk_w.  One launch for the text and image projections of a double block.
  int split_row; uint32_t w2_off;
  const bf16_t* bias2; const bf16_t* gate2; const bf16_t* nq_w2; const bf16_t* nk_w2;
  // fused per-head RMSNorm + RoPE of the q / k column ranges (QKN kernel instantiation; see GemmArgs)
  const bf16_t* nq_w; const bf16_t* nk_w; const float* rope_cs; int rope_pos0, nq0, nq1, nk0, nk1; float n_eps;
};

// ------------------------------------------------------------------------------------------------
// Generic fallback: any M, N, K (K % 8 == 0 not even required), fp32 FMA on 64x64x16 LDS tiles.
// Only used for shapes the MFMA kernel does not take (K % 64 != 0: unit-test configs) and as the
// on-device cross-check of the fast kernel.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmParams p) {
  __shared__ float As[16][65];
  __shared__ float Ws[16][65];
  const int b = blockIdx.z;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const bf16_t* A = p.A + b * p.a_bs;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int r = i >> 4, c = i & 15;
      const int m = m0 + r, n = n0 + r, k = k0 + c;
      float av = 0.f;
      if (m < p.M && k < p.K) {
        if (p.cin > 0) {
          const int tap = k / p.cin, ci = k - tap * p.cin, dy = tap / p.ckw, dx = tap - p.ckw * dy;
          const int pix = m % (p.oH * p.oW), bb = m / (p.oH * p.oW), y = pix / p.oW, x = pix - y * p.oW;
          const int yy = y * p.cstride - p.cpad + dy, xx = x * p.cstride_x - p.cpad + dx;
          if (tap < 3 * p.ckw && yy >= 0 && xx >= 0 && yy < (p.inH << p.cup) && xx < (p.inW << p.cup))
            av = bf2f(A[((int64_t)(bb * p.inH + (yy >> p.cup)) * p.inW + (xx >> p.cup)) * p.cin + ci]);
        } else {
          av = bf2f(A[(int64_t)m * p.lda + k]);
        }
      }
      As[c][r] = av;
      Ws[c][r] = (n < p.N && k < p.K) ? bf2f(p.W[b * p.w_bs + (int64_t)n * p.ldw + k]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; w[i] = Ws[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * w[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      if (p.ws) { p.ws[b * p.ws_bs + (int64_t)m * p.ws_ld + n] = acc[i][j]; continue; }   // fp32-output mode (raw accumulators)
      float v = acc[i][j] + (p.bias ? bf2f(p.bias[n]) : 0.f);
      if (EPI == EPI_BIAS_GELU) { if (n >= p.gelu_from) v = gelu_tanh(round_bf(v)); }   // the Linear's bf16 output is what nn.GELU sees (attention.py:1209-1212)
      if (EPI == EPI_BIAS_GATE_RES) {
        const float g = bf2f(p.gate[b * p.gate_bs + n]);
        const float r = bf2f(p.res[b * p.r_bs + (int64_t)m * p.ldr + n]);
        v = r + round_bf(g * round_bf(v));
      }
      if (EPI == EPI_BIAS_RES) v = bf2f(p.res[b * p.r_bs + (int64_t)m * p.ldr + n]) + round_bf(v);
      p.C[b * p.c_bs + (int64_t)m * p.ldc + n] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Cache-policy bits of the kernel's three global streams (buffer instructions' aux operand: bit 0 sc0, bit 1 nt, bit 4 sc1); the macros exist
// for A/B builds (tools/run_r04_cachepolicy.sh, profiles/r04_cachepolicy.log).  Round 4: the OUTPUT stores are non-temporal -- a tile's 128 KiB of
// results are written once and not read by this kernel, and without the hint each round of epilogues (32 CUs x 128 KiB = one XCD's whole 4 MiB L2)
// pushes the operand panels the next tiles are about to share out of L2: 444.6 -> 439.6 ms per DiT forward on one box, 432.1 -> 429.9 on another.
// nt on the operand REQUESTS is a disaster (618 ms: the panels ARE the reuse); nt on the residual reads, sc0 / sc1 on the stores: no difference.
#ifndef TFX_GLDS_AUX
#define TFX_GLDS_AUX 0
#endif
#ifndef TFX_GSTORE_AUX
#define TFX_GSTORE_AUX 2
#endif
constexpr int GLDS_AUX = TFX_GLDS_AUX;      // operand prefetch (LDS-DMA requests)
constexpr int GSTORE_AUX = TFX_GSTORE_AUX;  // epilogue's output stores
#ifndef TFX_GRES_AUX
#define TFX_GRES_AUX 0
#endif
constexpr int GRES_AUX = TFX_GRES_AUX;      // epilogue's residual reads (read once)
#ifndef TFX_FP8_HEAD
#define TFX_FP8_HEAD 2                      // which fp8 instantiations unroll a tile's first two K-tiles (gemm8pp_kernel)
#endif
constexpr int LDS_X = 0;            // X_g set s at g*32768 + s*16384   (128 rows x 128 B)
constexpr int LDS_W = 65536;        // W   set s at 65536 + s*32768     (256 rows x 128 B)
constexpr int LDS_DUMMY = 131072;   // 8 x 1 KiB sink for out-of-range prefetches (keeps vmcnt counts uniform)
constexpr int LDS_TOTAL = LDS_DUMMY + 8192;          // one-tile kernel: operand sets + sink (its epilogue stages inside the sets)
constexpr int STG_WAVE = 4096;      // epilogue staging per wave: 32 rows x 128 B, XOR-swizzled 16-byte chunks

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// AUX = cache-policy bits of the load (gfx940+: 1 = sc0, 2 = nt, 16 = sc1).
template <int AUX>
__device__ __forceinline__ void glds16(const bf16_t* g, char* smem, uint32_t lds_off) {
  __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)(smem + lds_off), 16, 0, AUX);
}

#define TFX_BARRIER()                          \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

#define TFX_CAT(lo_, hi_) __builtin_shufflevector(__builtin_bit_cast(i32x4, lo_), __builtin_bit_cast(i32x4, hi_), 0, 1, 2, 3, 4, 5, 6, 7)

// One MFMA section of a wave: the 4 x 2 block of 16 x 16 accumulators (64 token rows x 32 output columns) advanced by one 64-deep
// K-tile.  bf16: 2 k-steps of v_mfma_f32_16x16x32_bf16 = 16 MFMAs of 4 passes (the two MFMAs of one accumulator are 8 apart:
// no dependent pair is ever back to back); e4m3: one v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per accumulator =
// 8 MFMAs of 8 passes.  256 matrix-pipe cycles either way.  WF[j][s] / XF[i][s]: fragment of column block j / row block i, 16-byte
// piece s of the lane's K-tile bytes.
template <bool FP8, bool HALF = false, int S0 = 0, bool ZERO = false>
__device__ __forceinline__ void mfma_section(f32x4 (&acc)[8][4], const bf16x8 (&WF)[2][2], const bf16x8 (&XF)[4][2], const int MIB,
                                             const int NJB) {
  // ZERO: the section's first MFMA into each accumulator takes a zero C operand (the first K-tile of a tile: no 128 v_mov per tile)
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FP8) {
#pragma unroll
    for (int j = (HALF ? S0 : 0); j < (HALF ? S0 + 1 : 2); ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[MIB + i][NJB + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(TFX_CAT(WF[j][0], WF[j][1]), TFX_CAT(XF[i][0], XF[i][1]),
                                                                                 ZERO ? z : acc[MIB + i][NJB + j], 0, 0, 0, 0x7f7f7f7f, 0,
                                                                                 0x7f7f7f7f);
  } else {
#pragma unroll
    for (int s = (HALF ? S0 : 0); s < (HALF ? S0 + 1 : 2); ++s)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[MIB + i][NJB + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[j][s], XF[i][s], (ZERO && s == 0) ? z : acc[MIB + i][NJB + j], 0, 0, 0);
  }
}
// the intrinsics are pure: nothing but this keeps hipcc from moving a section's MFMAs across the barriers around it
#define TFX_PIN_SECTION(MIB, NJB)                                                                                        \
  do {                                                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                    \
        asm volatile("" : "+v"(acc[(MIB) + i_][(NJB) + j_]));                                                            \
  } while (0)

// bench-only phase timers of the persistent kernel (tools/gemm_phase_timers.py): s_memtime stamps in SGPRs
#ifdef TFX_BENCH
#define TFX_STAMP(i) do { if (tfx_stamp) tfx_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TFX_STAMP(i) ((void)0)
#endif
// ------------------------------------------------------------------------------------------------
// Epilogue of one 256 x 256 tile, shared by the two MFMA kernels.  With the operands swapped (MFMA "A" = weight rows, "B" =
// token rows) lane l of a wave holds, per (mi, nj), FOUR CONSECUTIVE OUTPUT COLUMNS of one token row:
//     m = m0 + g*128 + mi*16 + (l & 15),    n = n0 + wc*64 + nj*16 + (l >> 4)*4 + e,   e = 0..3,  mi = 0..7, nj = 0..3.
// bias / GELU / gate are applied in fp32 on those; the bf16 results cross a wave-private LDS tile (32 rows x 128 B per
// 32-row block, 16-byte chunks XOR-swizzled with (row >> 1) & 7: conflict-free 8-byte writes and 16-byte reads) and leave as
// 16-byte stores in which 8 lanes cover one full 128-byte line of a row.  Everything the epilogue needs from memory (bias,
// gate, the first residual rows) is requested up front and waited for once; residual rows of later blocks RES_DEPTH blocks
// ahead of their use.  FP8: the accumulators are first dequantised in place (row scale x channel scale).  QKN: tiles inside
// the q / k column ranges get the per-head RMSNorm + RoPE (see gemm8pp_kernel).  `stg`: this wave's 4 KiB staging tile,
// `stg_partner`: the staging tile of the wave that owns the other 64 columns of this wave's heads (QKN only).
template <int EPI, bool FP8, bool QKN, int RES_DEPTH>
__device__ __forceinline__ void tile_epilogue(f32x4 (&acc)[8][4], const GemmParams& p, const int m0, const int n0, const int b,
                                              const int g, const int wc, const int lane, char* stg, const char* stg_partner,
                                              unsigned long long* tfx_stamp = nullptr, const bool second = false) {
  // row-split weights: this tile's bias / gate / norm weights (block-uniform pointer selects)
  const bf16_t* const p_bias = second ? p.bias2 : p.bias;
  const bf16_t* const p_gate = second ? p.gate2 : p.gate;
  const bf16_t* const p_nq = second ? p.nq_w2 : p.nq_w;
  const bf16_t* const p_nk = second ? p.nk_w2 : p.nk_w;
  // lane-derived offsets are rebuilt from an opaque copy of the lane id: derived from `lane` they are loop invariants that
  // hipcc keeps in ~20 VGPRs across the K loop, which is what pushes the kernel into spilling
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int q_e = lane_e >> 4, r16 = lane_e & 15;
  const int ncol = n0 + wc * 64 + q_e * 4;
  const bool do_gelu = (EPI == EPI_BIAS_GELU) && (n0 >= p.gelu_from);
  const int crow = lane_e >> 3, cchunk = lane_e & 7;
  const int nst = n0 + wc * 64 + cchunk * 8;
  constexpr bool HAS_RES = (EPI == EPI_BIAS_GATE_RES || EPI == EPI_BIAS_RES);
  // C, the residual, bias and gate are addressed through buffer descriptors whose range check does the edge handling: a lane
  // whose row is beyond M or whose 8 columns start beyond N carries an out-of-range offset, its loads return zero and its stores
  // are dropped -- no branch around any access, so hipcc batches a block's LDS reads and stores instead of serialising
  // {branch, ds_read, wait, store} per row.  Descriptors start at this wave's first row (wave-uniform 64-bit address).
  const int rows_ok = min(max(p.M - m0 - g * 128, 0), 128);               // valid rows of this wave's 128
  constexpr uint32_t OOR = 0x80000000u;                                     // >= every num_records below
  auto uniform_rsrc = [](const void* base, int bytes) {   // scalar (SGPR) descriptor: otherwise every access sits in a waterfall loop
    const uint64_t a = (uint64_t)base;
    const uint64_t u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |   // (the builtin returns a signed int)
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
    return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
  };
  const auto rsrcC = uniform_rsrc(p.C + b * p.c_bs + (int64_t)(m0 + g * 128) * p.ldc, rows_ok ? (int)(((int64_t)(rows_ok - 1) * p.ldc + p.N) * 2) : 0);
  const auto rsrcR = uniform_rsrc(HAS_RES ? p.res + b * p.r_bs + (int64_t)(m0 + g * 128) * p.ldr : p.C,
                                  HAS_RES && rows_ok ? (int)(((int64_t)(rows_ok - 1) * p.ldr + p.N) * 2) : 0);
  const uint32_t col_off = nst < p.N ? (uint32_t)nst * 2u : OOR;          // N % 8 == 0: a lane's 8 columns are all in or all out
  auto load_res = [&](int blk, u32x4 (&rr)[4]) {
#pragma unroll
    for (int itr = 0; itr < 4; ++itr)
      rr[itr] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)((uint32_t)(blk * 32 + itr * 8 + crow) * (uint32_t)(p.ldr * 2) + col_off), 0, GRES_AUX);
  };
  if constexpr (FP8) {
    // dequantise in place first (row scale x channel scale): the scale registers are dead before bias / gate / residual are
    // requested.  Costs a second memory round trip per tile, saves the spills of holding both sets.
    float sa[8];
    f32x4 sw[4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) sa[mi] = p.a_scale[b * p.as_bs + min(m0 + g * 128 + mi * 16 + r16, p.M - 1)];
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      const int n = ncol + nj * 16;
      sw[nj] = *reinterpret_cast<const f32x4*>(p.w_scale + (n < p.N ? n : 0));
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mi][nj][e] = (acc[mi][nj][e] * sa[mi]) * sw[nj][e];
    __builtin_amdgcn_sched_barrier(0);
  }
  u32x2 bsr[4], gtr[4];   // bias / gate stay packed (bf16 pairs) until they are used
  u32x4 rr[RES_DEPTH][4];
  const auto rsrcB = uniform_rsrc(p_bias ? p_bias : p.C, p_bias ? p.N * 2 : 0);
  const auto rsrcG = uniform_rsrc(EPI == EPI_BIAS_GATE_RES ? p_gate + b * p.gate_bs : p.C, EPI == EPI_BIAS_GATE_RES ? p.N * 2 : 0);
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {   // no bias / columns beyond N: zeros (computed, never stored)
    bsr[nj] = __builtin_amdgcn_raw_buffer_load_b64(rsrcB, (ncol + nj * 16) * 2, 0, 0);
    if (EPI == EPI_BIAS_GATE_RES) gtr[nj] = __builtin_amdgcn_raw_buffer_load_b64(rsrcG, (ncol + nj * 16) * 2, 0, 0);
  }
  if (HAS_RES) {
#pragma unroll
    for (int d = 0; d < RES_DEPTH; ++d) load_res(d, rr[d]);
  }
  const bool in_q = QKN && n0 >= p.nq0 && n0 < p.nq1;
  const bool norm_tile = QKN && (in_q || (n0 >= p.nk0 && n0 < p.nk1));      // block-uniform
  // every operand request of this tile's K loop (incl. the next tile's first K-tiles) is older than the stores below;
  // loads and stores retire out of order with respect to each other, so the counted waits of the next K loop are only
  // meaningful once these have landed
  // (the builtin, not inline asm: hipcc's own wait-count pass must see that the bias / gate / residual registers have landed, or
  // it re-waits for them -- vmcnt(0) -- at their next redefinition, which is the first fragment read of the K loop)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  __builtin_amdgcn_sched_barrier(0);
  TFX_STAMP(1);
  float rinv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  u32x4 nwlo = u32x4{0u, 0u, 0u, 0u}, nwhi = u32x4{0u, 0u, 0u, 0u};   // norm weights of the 8 columns this lane stores: even / odd column of each pair, the other half zero
  if (norm_tile) {
    {
      const u32x4 nw8 = *reinterpret_cast<const u32x4*>((in_q ? p_nq : p_nk) + (wc & 1) * 64 + cchunk * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) { nwlo[i] = nw8[i] & 0xffffu; nwhi[i] = nw8[i] & 0xffff0000u; }
    }
    // Round 6: the Linear's output as PACKED bf16 pairs (what the reference's RMSNorm sees), kept in the first two registers of each
    // accumulator quad -- one v_cvt_pk_bf16_f32 per pair, no unpack -- and the sum of squares by v_dot2c_f32_bf16 on the packed pair
    // (p.lo^2 + p.hi^2 + ss: one instruction where round 2-5 spent two unpacks and two FMAs -- the BUILTIN, not inline asm: the sums feed
    // v_permlane*_swap, which needs two wait states behind a VALU write of its operand, and hipcc pads only behind writes it can see;
    // the first version's asm left the swap reading a stale sum now and then: graph replay != eager in tests/test_configs_gpu.py);
    // then the row's 64 columns of this wave
    // (the four lanes l, l ^ 16, l ^ 32, l ^ 48 hold one row)
    float ss[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      ss[mi] = 0.f;
#pragma unroll
      for (int nj = 0; nj < 4; ++nj) {
        const u32x2 br = bsr[nj];
        const float bs[4] = {__uint_as_float(br[0] << 16), __uint_as_float(br[0] & 0xffff0000u),
                             __uint_as_float(br[1] << 16), __uint_as_float(br[1] & 0xffff0000u)};
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const uint32_t pk = pack_bf2(acc[mi][nj][2 * h2] + bs[2 * h2], acc[mi][nj][2 * h2 + 1] + bs[2 * h2 + 1]);
          ss[mi] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, pk), __builtin_bit_cast(bf16x2_t, pk), ss[mi], false);
          acc[mi][nj][h2] = __uint_as_float(pk);
        }
      }
      // lanes l ^ 16, l ^ 32 by v_permlane16_swap / v_permlane32_swap (VALU: rounds 2-5 used two ds_bpermute round trips per row block)
      {
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ss[mi]), __float_as_uint(ss[mi]), false, false);
        ss[mi] = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss[mi]), __float_as_uint(ss[mi]), false, false);
        ss[mi] = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
      }
    }
    if (q_e == 0) {
      *reinterpret_cast<f32x4*>(stg + r16 * 32) = f32x4{ss[0], ss[1], ss[2], ss[3]};
      *reinterpret_cast<f32x4*>(stg + r16 * 32 + 16) = f32x4{ss[4], ss[5], ss[6], ss[7]};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TFX_BARRIER();                                 // both stripes of every head have published their sums
    const f32x4 o0 = *reinterpret_cast<const f32x4*>(stg_partner + r16 * 32);
    const f32x4 o1 = *reinterpret_cast<const f32x4*>(stg_partner + r16 * 32 + 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TFX_BARRIER();                                 // ... and read the partner's, before the staging areas are reused below
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) rinv[mi] = rsqrtf((ss[mi] + (mi < 4 ? o0[mi & 3] : o1[mi & 3])) * (1.0f / 128.0f) + p.n_eps);
  }
  // NORM / GELU are per-tile (block-uniform) properties: one straight-line instantiation each instead of a branch per element group
  auto store_blocks = [&](auto NORM_T, auto GELU_T) __attribute__((always_inline)) {
    constexpr bool NORM = decltype(NORM_T)::value;
    constexpr bool GELU = decltype(GELU_T)::value;
    // (cos, sin) pairs of the four rows this lane stores from a row block.  The requests of block k + 1 are issued BEFORE the
    // stores of block k: vmcnt retires in issue order, so a table load behind a store would only return after that store's
    // acknowledgement from L2 (~1 us), four times per tile.
    f32x4 csr[4][2];
    // round 6: through a buffer descriptor that starts at this wave's first table row (rows beyond M read zeros and are never stored):
    // one per-lane offset for the whole tile, the row block in the SCALAR offset -- no 64-bit address arithmetic per request
    const auto rsrcT = uniform_rsrc(NORM ? p.rope_cs + (int64_t)(p.rope_pos0 + m0 + g * 128) * 128 : (const float*)p.C, NORM ? rows_ok * 512 : 0);
    const int cs_off = crow * 512 + (wc & 1) * 256 + cchunk * 32;
    auto load_cs = [&](int blk) {
#pragma unroll
      for (int itr = 0; itr < 4; ++itr) {
        csr[itr][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcT, cs_off, (blk * 32 + itr * 8) * 512, 0));
        csr[itr][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcT, cs_off + 16, (blk * 32 + itr * 8) * 512, 0));
      }
    };
    if (NORM) load_cs(0);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mi = blk * 2 + h;
        const int R = h * 16 + r16;                 // row inside the 32-row staging tile
#pragma unroll
        for (int nj = 0; nj < 4; ++nj) {
          const u32x2 br = bsr[nj];
          const float bs[4] = {__uint_as_float(br[0] << 16), __uint_as_float(br[0] & 0xffff0000u),
                               __uint_as_float(br[1] << 16), __uint_as_float(br[1] & 0xffff0000u)};
          float v[4];
          if (NORM) {      // the pre-pass left the bias-added, bf16-rounded pairs packed in acc[mi][nj][0..1]: x * (1 / rms) here, where the
                           // row's factor is lane-local (round 6; rounds 2-5 shuffled it to the transposed rows), rounded by the pack below
            const uint32_t p0 = __float_as_uint(acc[mi][nj][0]), p1 = __float_as_uint(acc[mi][nj][1]);
            v[0] = __uint_as_float(p0 << 16) * rinv[mi];
            v[1] = __uint_as_float(p0 & 0xffff0000u) * rinv[mi];
            v[2] = __uint_as_float(p1 << 16) * rinv[mi];
            v[3] = __uint_as_float(p1 & 0xffff0000u) * rinv[mi];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[mi][nj][e] + bs[e];
          }
          if (GELU) {   // nn.GELU acts on the Linear's bf16 OUTPUT (activations.py:85-88 after the bf16 nn.Linear): round first
            round_bf2(v[0], v[1]);
            round_bf2(v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
          }
          if (EPI == EPI_BIAS_GATE_RES) {
            const u32x2 gr = gtr[nj];
            const float gt[4] = {__uint_as_float(gr[0] << 16), __uint_as_float(gr[0] & 0xffff0000u),
                                 __uint_as_float(gr[1] << 16), __uint_as_float(gr[1] & 0xffff0000u)};
            round_bf2(v[0], v[1]);
            round_bf2(v[2], v[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gt[e] * v[e];
          }
          u32x2 o;
          o[0] = pack_bf2(v[0], v[1]);
          o[1] = pack_bf2(v[2], v[3]);
          // columns nj*16 + q*4 .. +4 of row R = 8-byte half (q & 1) of 16-byte chunk nj*2 + (q >> 1), stored at chunk ^ ((R >> 1) & 7):
          // the 16 lanes of a ds_write_b64 group (one q, rows 0..15) touch 16 different 16-byte slots of the 256-byte bank line
          *reinterpret_cast<u32x2*>(stg + R * 128 + (((nj * 2 + (q_e >> 1)) ^ ((R >> 1) & 7)) << 4) + ((q_e & 1) << 3)) = o;
        }
      }
      // wave-private region + in-order LDS pipe: no barrier between the writes above and the reads below
      if (QKN) __builtin_amdgcn_sched_barrier(0);   // keeps the table loads of later row blocks from being hoisted over live accumulators
      u32x4 vals[4];
#pragma unroll
      for (int itr = 0; itr < 4; ++itr) {
        const int row = itr * 8 + crow;
        u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((cchunk ^ ((row >> 1) & 7)) << 4));
        if (NORM) {
          // this lane now holds 8 consecutive columns (4 rotation pairs) of row `row`, normalised and rounded: a * weight is the EXACT fp32
          // product of two bf16 values -- v_dot2_f32_bf16 against the weight pair with its other half zeroed does unpack + multiply in one
          // instruction --, rounded (the reference's second rounding point, normalization.py:543), then the rotation with the row's
          // (cos, sin) pairs, 32 contiguous bytes of the table, rounded once
          const f32x4 c0 = csr[itr][0], c1 = csr[itr][1];
          float y[8], o8[8];
          // The eight products of a row's eight columns in ONE statement: VOP3P form with a literal zero addend (the builtin lowers to
          // v_mov 0 + v_dot2c: one instruction more per product).  A DOT result is NOT interlocked on gfx940+ -- a VALU reading it needs 3
          // wait states (LLVM GCNHazardRecognizer, DotWriteDifferentVALURead) and hipcc pads only behind DOTs it can see: the first version
          // (one statement per product, no s_nop) let the pack behind it read stale registers now and then, a forward was not
          // deterministic; tools/check_mfma_hazard.py knows the rule since.  Eight in a row + ONE s_nop 2: only the last results are
          // inside the window when the statement ends.  (A second trap on the way: __builtin_bit_cast applied directly to an ext-vector
          // ELEMENT reads element 0 whatever the index, hipcc 7.2 -- hence no fdot2 builtin on val[i] here.)
          asm("v_dot2_f32_bf16 %0, %8, %12, 0\n\tv_dot2_f32_bf16 %1, %8, %16, 0\n\t"
              "v_dot2_f32_bf16 %2, %9, %13, 0\n\tv_dot2_f32_bf16 %3, %9, %17, 0\n\t"
              "v_dot2_f32_bf16 %4, %10, %14, 0\n\tv_dot2_f32_bf16 %5, %10, %18, 0\n\t"
              "v_dot2_f32_bf16 %6, %11, %15, 0\n\tv_dot2_f32_bf16 %7, %11, %19, 0\n\ts_nop 2"
              : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3]), "=&v"(y[4]), "=&v"(y[5]), "=&v"(y[6]), "=&v"(y[7])
              : "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(nwlo[0]), "v"(nwlo[1]), "v"(nwlo[2]), "v"(nwlo[3]),
                "v"(nwhi[0]), "v"(nwhi[1]), "v"(nwhi[2]), "v"(nwhi[3]));
#pragma unroll
          for (int i = 0; i < 4; ++i) round_bf2(y[2 * i], y[2 * i + 1]);
          o8[0] = y[0] * c0[0] + (-y[1]) * c0[1];  o8[1] = y[1] * c0[0] + y[0] * c0[1];
          o8[2] = y[2] * c0[2] + (-y[3]) * c0[3];  o8[3] = y[3] * c0[2] + y[2] * c0[3];
          o8[4] = y[4] * c1[0] + (-y[5]) * c1[1];  o8[5] = y[5] * c1[0] + y[4] * c1[1];
          o8[6] = y[6] * c1[2] + (-y[7]) * c1[3];  o8[7] = y[7] * c1[2] + y[6] * c1[3];
          val = pack8(o8);
        }
        if (HAS_RES) {
          float fv[8], fr[8];
          unpack8(val, fv);
          unpack8(rr[blk % RES_DEPTH][itr], fr);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fr[e];
          val = pack8(fv);
        }
        vals[itr] = val;
      }
      // requests first, stores second (see load_cs)
      if ((NORM && blk + 1 < 4) || (HAS_RES && blk + RES_DEPTH < 4)) {
        __builtin_amdgcn_sched_barrier(0);
        if (NORM && blk + 1 < 4) load_cs(blk + 1);
        if (HAS_RES && blk + RES_DEPTH < 4) load_res(blk + RES_DEPTH, rr[blk % RES_DEPTH]);   // in flight while the next blocks are converted and staged
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int itr = 0; itr < 4; ++itr)
        __builtin_amdgcn_raw_buffer_store_b128(vals[itr], rsrcC,
                                               (int)((uint32_t)(blk * 32 + itr * 8 + crow) * (uint32_t)(p.ldc * 2) + col_off), 0, GSTORE_AUX);
      if (QKN) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (QKN && norm_tile) store_blocks(std::true_type{}, std::false_type{});   // q / k columns lie in front of gelu_from
  else if (EPI == EPI_BIAS_GELU && do_gelu) store_blocks(std::false_type{}, std::true_type{});
  else store_blocks(std::false_type{}, std::false_type{});
}

// ABL (bench-only ablations with WRONG results by construction, never dispatched by the product path; tools/
// bench_gemm_abl.py, gemm_abl2.py, energy_probe.py): bit 0 no operand requests in the main loop, bit 1 no LDS fragment
// reads, bit 2 no barriers, bit 3 every tile reads panel 0 (L2-resident), bit 4 no MFMAs, bit 15 half the request bytes,
// bit 16 every request reads the same 1 KiB (L1-resident).  ABL = 0 is the real kernel.
// CONV: 0 plain GEMM, 1 implicit 3x3 convolution with Cin % 64 == 0 (one tap per K-tile), 2 its narrow-input form
// (Cin = 8 / 16 / 32: the tap is a per-lane quantity; a separate instantiation so that the wide form keeps its registers).
template <int EPI, int ABL = 0, int CONV = 0>
__global__ __launch_bounds__(512) void gemm8p_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;   // row group: rows g*128 .. +128 of the block tile
  const int wc = wave & 3;   // column stripe: cols wc*64 .. +64

  // ---- tile coordinates: XCD-contiguous remap (bijective), then grouped (8 row tiles) ordering.
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int per_batch = p.tm * p.tn;
  const int b = bid / per_batch;
  int idx = bid - b * per_batch;
  const int GM = p.gm;
  const int grp = idx / (GM * p.tn);
  const int first_m = grp * GM;
  const int gsz = min(GM, p.tm - first_m);
  idx -= grp * GM * p.tn;
  const int tile_m = first_m + idx % gsz;
  const int tile_n = idx / gsz;
  const int m0 = tile_m * 256, n0 = tile_n * 256;

  const bf16_t* Xb = p.A + b * p.a_bs + (int64_t)((ABL & 8) ? 0 : m0) * p.lda;   // ABL bit3: every tile loads panel 0
  const bf16_t* Wb = p.W + b * p.w_bs + (int64_t)((ABL & 8) ? 0 : n0) * p.ldw;
  const int nt = p.K >> 6;

  // ---- prefetch bookkeeping.  Item table of this wave's group, in phase order q0..q3 (u = tile being computed):
  //   G0: X0_hi(u+1)  X0_lo(u+2)   W_lo_b(u+2)  W_hi_a(u+2)
  //   G1: X1_hi(u+1)  W_lo_a(u+2)  X1_lo(u+2)   W_hi_b(u+2)
  // i.e. every 64-row quarter of the tile-(u+2) operands is re-requested in the first slot in which all reads of
  // the bytes it overwrites are known complete, which leaves each request >= 11 slots (~2800 cycles at full MFMA
  // rate) before its first reader.  Every lane issues 2 x 16-byte loads per item; piece = 8 consecutive LDS rows
  // written by one wave instruction.
  const int lr = lane >> 3, cphys = lane & 7;
  int goff[4][2];        // per-lane element offset from Xb / Wb (without the k offset); CONV X items: chunk offset
  int cpix[4][2], cyx[4][2];  // CONV X items: first pixel of the row's image, packed (iy0 + 1) << 16 | (ix0 + 1)
  uint32_t ldst[4][2];   // wave-uniform LDS byte offset inside set 0 of the destination tile
  bool isx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool x_item = g == 0 ? (q == 0 || q == 1) : (q == 0 || q == 2);
    isx[q] = x_item;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = wc * 2 + j;  // piece 0..7 (wc doubles as the wave's index inside its group)
      int row0;                   // first tile-local row of the piece
      if (x_item) {
        row0 = (q == 0 ? 64 : 0) + pc * 8;
      } else {
        const int whi_item = (q == 3);                                   // q3 stages W_hi, q1/q2 stage W_lo
        const int half = g == 0 ? (q == 2 ? 128 : 0) : (q == 1 ? 0 : 128);  // "a" = stripes 0,1; "b" = stripes 2,3
        row0 = half + (pc >> 2) * 64 + whi_item * 32 + (pc & 3) * 8;
      }
      const int row = row0 + lr;
      const int key = (row >> 1) & 7;
      const int clog = cphys ^ key;
      cpix[q][j] = 0; cyx[q][j] = 0;
      if (x_item) {
        int grow = g * 128 + row;  // each group stages its own half of the X tile
        if (m0 + grow > p.M - 1) grow = p.M - 1 - m0;
        if (CONV) {
          const int m = m0 + grow, hw = p.oH * p.oW;
          const int bb = m / hw, pix = m - bb * hw, y = pix / p.oW, x = pix - y * p.oW;
          cpix[q][j] = bb * p.inH * p.inW;
          cyx[q][j] = ((y * p.cstride - p.cpad + 1) << 16) | (x * p.cstride_x - p.cpad + 1);
          goff[q][j] = clog * 8;
        } else {
          goff[q][j] = grow * (int)p.lda + clog * 8;
        }
        ldst[q][j] = LDS_X + g * 32768 + row0 * 128;
      } else {
        int grow = row;
        if (n0 + grow > p.N - 1) grow = p.N - 1 - n0;
        goff[q][j] = grow * (int)p.ldw + clog * 8;
        ldst[q][j] = LDS_W + row0 * 128;
      }
    }
  }
  const uint32_t dummy = LDS_DUMMY + wave * 1024;
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)Xb, 0, 0x7fffffff, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0x7fffffff, 0x00020000);

  auto stage = [&](int q, int tile) {
    const bool ok = tile < nt;
    const int kt = ok ? tile : nt - 1;
    const uint32_t setoff = (tile & 1) * (isx[q] ? 16384u : 32768u);
    constexpr int kAux = GLDS_AUX;
    if (CONV && isx[q]) {
      const int vh = p.inH << p.cup, vw = p.inW << p.cup;
      if (CONV == 2) {
        // narrow inputs (Cin = 8 / 16 / 32: conv_in of the VAE ends): a 64-wide K-tile spans 64 / Cin taps, so the tap
        // is a per-lane quantity of the lane's 16-byte chunk; K is padded to a multiple of 64 and taps >= 9 read zeros
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = kt * 64 + goff[q][j], tap = e >> p.csh, ci = e & (p.cin - 1);
          const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
          const int yy = (cyx[q][j] >> 16) - 1 + dy, xx = (cyx[q][j] & 0xffff) - 1 + dx;
          const bool inside = tap < 9 && (unsigned)yy < (unsigned)vh && (unsigned)xx < (unsigned)vw;
          const bf16_t* src = inside ? p.A + ((int64_t)(cpix[q][j] + (yy >> p.cup) * p.inW + (xx >> p.cup)) * p.cin + ci)
                                     : p.zero + goff[q][j];
          glds16<kAux>(src, smem, ok ? ldst[q][j] + setoff : dummy);
        }
        return;
      }
      // K index -> (tap, channel block): a 64-wide K-tile never straddles a tap because Cin % 64 == 0
      const int k0 = kt * 64, tap = k0 / p.cin, ci0 = k0 - tap * p.cin, dy = tap / p.ckw, dx = tap - p.ckw * dy;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int yy = (cyx[q][j] >> 16) - 1 + dy, xx = (cyx[q][j] & 0xffff) - 1 + dx;
        const bool inside = (unsigned)yy < (unsigned)vh && (unsigned)xx < (unsigned)vw;
        const bf16_t* src = inside ? p.A + ((int64_t)(cpix[q][j] + (yy >> p.cup) * p.inW + (xx >> p.cup)) * p.cin + ci0 + goff[q][j])
                                   : p.zero + goff[q][j];
        glds16<kAux>(src, smem, ok ? ldst[q][j] + setoff : dummy);
      }
      return;
    }
    // buffer form of the LDS DMA (everything except the gathered X rows of the conv mode above): per-lane 32-bit byte
    // offset + scalar K offset against a per-tile descriptor -- no 64-bit per-lane address arithmetic in the issue path
    const auto rs = isx[q] ? rsrcX : rsrcW;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (!((ABL & 32768) && j == 1))   // bench-only bit 15: half the prefetch bytes
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + (ok ? ldst[q][j] + setoff : dummy)), 16,
                                                 (ABL & 65536) ? lane * 16 : goff[q][j] * 2,  // bit 16: L1-hot source
                                                 (ABL & 65536) ? 0 : kt * 128, 0, kAux);
  };

  // ---- fragment read addresses (set 0).  v_mfma_f32_16x16x32_bf16: lane l supplies row (l & 15) of a 16-row block and the 8
  // k-values 8 * (l >> 4) .. of a 32-deep k-step, i.e. 16-byte chunk 4 s + (l >> 4) of the row's 128-byte K-tile (k-step s);
  // rows are 16-aligned block + (l & 15), so the swizzle key (row >> 1) & 7 is (l & 15) >> 1
  const int r16 = lane & 15, q4 = lane >> 4;
  uint32_t fx[2], fw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint32_t o = r16 * 128 + (((4 * s + q4) ^ (r16 >> 1)) << 4);
    fx[s] = LDS_X + g * 32768 + o;
    fw[s] = LDS_W + wc * 64 * 128 + o;
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: all of tile 0, plus the part of tile 1 that the steady state would have issued
  // during "tile -1" (the (u+2)-type items).
#pragma unroll
  for (int q = 0; q < 4; ++q) stage(q, 0);
#pragma unroll
  for (int q = 1; q < 4; ++q) stage(q, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFX_BARRIER();
  if (g == 1) TFX_BARRIER();  // stagger: G1 runs one barrier behind G0

  bf16x8 xf[4][2], wlo[2][2], whi[2][2];

#define LDS_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
  // One MFMA section = 16 MFMAs (k-step 0 of all eight accumulators, then k-step 1) with the section's operand request (MID)
  // between the two k-steps; sched_barriers keep hipcc from re-interleaving them.
#define MFMA16(WF, MIB, NJB, MID)                                                                                         \
  do {                                                                                                                    \
    if (ABL & 16) { /* bench-only: no MFMAs, fragment reads kept live */                                                  \
      _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) asm volatile("" ::"v"(xf[i_][s_]));                              \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) asm volatile("" ::"v"(WF[j_][s_]));                              \
      }                                                                                                                   \
      break;                                                                                                              \
    }                                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                        \
    mfma_section<false, true, 0>(acc, WF, xf, MIB, NJB);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    MID;                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    mfma_section<false, true, 1>(acc, WF, xf, MIB, NJB);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                                        \
  } while (0)
  // waits sit in the loads section: the 4 newest sections (8 requests) may still be in flight
#define WAIT_PREFETCH() do { if (ABL & 32768) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } while (0)
  // The request of a phase is issued in the middle of the wave's own MFMA section: +3 % over issuing it in the loads
  // section of the same phase.  (The persistent kernel below moves all requests into the two light loads sections
  // instead, which is better still.)
#define MID_STAGE(q, t) do { if (!(ABL & 1)) stage(q, t); } while (0)
#define BODY_BARRIER() do { if (!(ABL & 4)) TFX_BARRIER(); } while (0)
  if (ABL & 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i][s] = LDS_FRAG(fx[s] + i * 2048);
#pragma unroll
      for (int j = 0; j < 2; ++j) { wlo[j][s] = LDS_FRAG(fw[s] + j * 2048); whi[j][s] = LDS_FRAG(fw[s] + 4096 + j * 2048); }
    }
  }
  auto tile_body = [&](int u, const uint32_t xs, const uint32_t ws) {
    // xs / ws: byte offset of this tile's set inside the X / W regions (0 or 16384 / 32768)
    // ---- q0: X_lo (rows 0..63 of the group's half), W_lo (cols 0..31 of the stripe)
    if (!(ABL & 2)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i][s] = LDS_FRAG(fx[s] + xs + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) wlo[j][s] = LDS_FRAG(fw[s] + ws + j * 2048);
      }
    }
    WAIT_PREFETCH();
    BODY_BARRIER();
    MFMA16(wlo, 0, 0, MID_STAGE(0, u + 1));
    BODY_BARRIER();
    // ---- q1: W_hi (cols 32..63)
    if (!(ABL & 2)) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) whi[j][s] = LDS_FRAG(fw[s] + ws + 4096 + j * 2048);
    }
    WAIT_PREFETCH();
    BODY_BARRIER();
    MFMA16(whi, 0, 2, MID_STAGE(1, u + 2));
    BODY_BARRIER();
    // ---- q2: X_hi (rows 64..127)
    if (!(ABL & 2)) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i][s] = LDS_FRAG(fx[s] + xs + 8192 + i * 2048);
    }
    WAIT_PREFETCH();
    BODY_BARRIER();
    MFMA16(whi, 4, 2, MID_STAGE(2, u + 2));
    BODY_BARRIER();
    // ---- q3: no reads
    WAIT_PREFETCH();
    BODY_BARRIER();
    MFMA16(wlo, 4, 0, MID_STAGE(3, u + 2));
    BODY_BARRIER();
  };

  for (int u = 0; u < nt; u += 2) {
    tile_body(u, 0u, 0u);
    if (u + 1 < nt) tile_body(u + 1, 16384u, 32768u);
  }
  if (g == 0) TFX_BARRIER();  // re-align the two groups
#undef LDS_FRAG
#undef MFMA16
#undef WAIT_PREFETCH
#undef MID_STAGE
#undef BODY_BARRIER

  // ---- epilogue (tile_epilogue): the operand sets are free now -- every wave's last fragment read lies before the barrier
  // above -- so each wave stages through 4 KiB of them; out-of-range prefetches may still target the sink region and are
  // drained by the epilogue's own vmcnt(0)
  tile_epilogue<EPI, false, false, 2>(acc, p, m0, n0, b, g, wc, lane, smem + wave * STG_WAVE, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Persistent form of gemm8p_kernel -- the kernel the DiT runs on (any M, N; K a multiple of 128; no conv): one block
// per CU walks its XCD's contiguous range of the tile order.  Same tile, fragments, MFMA order and barrier ping-pong
// as above; what changes:
//   * requests are issued from the LOADS sections.  An LDS-DMA instruction costs its wave ~60 issue cycles; issued
//     between the MFMAs of a section (above) that is a hole in the matrix pipe, issued while the partner group owns
//     the pipe it is free as long as the loads section stays under the 256 cycles of the partner's 16 MFMAs.  Each
//     group stages its own X half and the W stripes {2g, 2g+1}, everything for K-tile u+2 (same buffer set as u):
//         item 0  X_lo  reads done after L0            item 1  W_lo  after L0 (+1 slot: the partner's read)
//         item 2  W_hi  after L1 (+1)                  item 3  X_hi  after L2
//     PLACE 1:  L1 {0,1}  L3 {2,3}          PLACE 2 (default):  L1 {0}  L2 {1}  L3 {2,3}
//     Deadlines (issuer's own counted wait, one barrier before the first reader): items 0,1 at L3 of K-tile u+1,
//     item 2 at L0 of u+2, item 3 at L1 of u+2 -> vmcnt 12 / 10 / 12 (PLACE 1) or 12 / 10 / 10 (PLACE 2); every
//     request has >= 10 slots (~2.5k cycles) in flight, the MFMA sections carry nothing but MFMAs;
//   * the requests of a tile's last two K-tiles fetch K-tiles 0 and 1 of the block's NEXT tile, so a tile starts
//     with its operands in LDS: no per-tile prologue latency, no sink traffic;
//   * the epilogue stages through its own 32 KiB (8 waves x 32 rows x 128 B, XOR-swizzled 16-byte chunks) behind
//     the two sets, one 32-row block of the accumulators at a time;
//   * the two row groups stay one barrier apart across tiles: one group's epilogue overlaps the other's MFMAs.
// Operands are addressed through ONE buffer descriptor per matrix, the tile origin folded into the scalar offset
// (32 bits, checked by persist_ok); rows beyond M / N are clamped on the way in and never stored.
constexpr int PP_STG = 131072;
constexpr int PP_STG_WAVE = STG_WAVE;
constexpr int PP_LDS_TOTAL = PP_STG + 8 * PP_STG_WAVE;  // 163840 = all of the CU's LDS

// FP8: A and W hold e4m3 bytes.  The byte geometry is the bf16 kernel's -- a K-tile is 128 bytes per row, now 128 k --
// so staging, swizzle, buffer sets, slots and waits are unchanged; a 16x16 accumulator takes one
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; 8 passes) per K-tile instead of two 16x16x32 bf16 (4 passes each),
// i.e. the same 256 cycles per section for twice the k, and the epilogue applies the per-row / per-channel scales.
// Lane l supplies row (l & 15), k bytes (l >> 4) * 32 .. +32 of the 128-k step = 16-byte chunks 2 (l >> 4), 2 (l >> 4) + 1.
// SPLIT: some or all work units are (tile, K slice) pairs (GemmParams::u_full / tail_r / sk): whole tiles first, with the normal
// epilogue, then the last tail_r tiles of every sample's tile order cut into sk slices of nt / sk K-tiles each, whose raw fp32
// accumulators go to p.ws and are finished by tail_reduce_kernel<EPI> (slices summed in order, bias, activation, gate,
// residual).  Used when a GEMM has fewer tiles than the chip has CUs (every tile sliced), and -- round 4 -- when a SMALL GEMM's
// last round of tiles would leave most CUs idle (the tiles of that round sliced so that they fill one short round instead).
// QKN (the fused q | k | v (| mlp) projections of the DiT blocks): tiles inside the q / k column ranges -- a 256-column tile
// is exactly two heads -- get the per-head RMSNorm (fp32 sum of squares over the 128 columns of the bf16-rounded Linear
// output, * weight) and the interleaved-pair RoPE applied in the epilogue, rounding for rounding as rmsnorm_rope_kernel
// (elementwise.hip) does it after the fact (reference: D/models/attention_processor.py:2001-2037).  A head's 128 columns
// belong to two waves (column stripes wc, wc ^ 1): their partial sums meet in LDS between two extra barriers per such tile.
template <int EPI, int PLACE = 2, bool FP8 = false, bool SPLIT = false, bool QKN = false>
__global__ __launch_bounds__(512) void gemm8pp_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;
  const int wc = wave & 3;
  constexpr int ESZ = FP8 ? 1 : 2;       // bytes per operand element
  const int nt_full = (p.K * ESZ) >> 7;                    // K-tiles of 128 bytes per row
  const int nt_slice = SPLIT ? nt_full / p.sk : nt_full;   // ... of a (tile, slice) unit

  // ---- this block's tiles: XCD x owns a contiguous range of the (grouped) tile order; its blocks stride through it
  const int per_batch = p.tm * p.tn;
  const int T1 = p.batch * per_batch;               // tiles
  const int T = SPLIT ? p.u_full + p.batch * p.tail_r * p.sk : T1;             // work units
  // Whole tiles and (tile, slice) units cost differently, so the two kinds are dealt out separately: XCD x owns a contiguous range
  // of EACH, its blocks stride through the whole tiles first and through the slice units after (every block ends up with its
  // share of both: 588 tiles = 512 whole + 76 x 3 slices gives each block 2 whole tiles and most of them one short unit).
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, bi = blockIdx.x >> 3;
  const int len0 = SPLIT ? p.u_full : T, len1 = T - len0;
  auto xrange = [&](int len, int& start, int& cnt) {
    const int q8 = len >> 3, r8 = len & 7;
    start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    cnt = q8 + (xcd < r8 ? 1 : 0);
  };
  int start0, cnt0, start1 = 0, cnt1 = 0;
  xrange(len0, start0, cnt0);
  if (SPLIT) xrange(len1, start1, cnt1);
  const int n0u = bi < cnt0 ? (cnt0 - bi + per_xcd - 1) / per_xcd : 0;
  const int n1u = (SPLIT && bi < cnt1) ? (cnt1 - bi + per_xcd - 1) / per_xcd : 0;
  const int nunits = n0u + n1u;
  if (nunits == 0) return;
  auto unit_id = [&](int k) { return k < n0u ? start0 + bi + k * per_xcd : len0 + start1 + bi + (k - n0u) * per_xcd; };
  int it = 0;        // index into this block's unit list

  struct Tile { int b, m0, n0, unit, nt; bool second; uint32_t xoff, woff; };   // unit: -1 = whole tile, else index of the (tile, slice) unit
  auto coords = [&](int id) {
    Tile t;
    int idx, slice = 0;
    t.unit = -1;
    t.nt = nt_full;
    if (SPLIT && id >= p.u_full) {
      t.unit = id - p.u_full;                       // slice-major: neighbours in the unit order are neighbouring tiles of one K range
      const int TT = p.batch * p.tail_r;
      slice = t.unit / TT;
      const int tt = t.unit - slice * TT;           // tail tile
      t.b = tt / p.tail_r;
      idx = per_batch - p.tail_r + (tt - t.b * p.tail_r);
      t.nt = nt_slice;
    } else {
      const int full_per_batch = SPLIT ? per_batch - p.tail_r : per_batch;
      t.b = id / full_per_batch;
      idx = id - t.b * full_per_batch;
    }
    const int GM = p.gm;
    const int grp = idx / (GM * p.tn);
    const int first_m = grp * GM;
    const int gsz = min(GM, p.tm - first_m);
    idx -= grp * GM * p.tn;
    t.m0 = (first_m + idx % gsz) * 256;
    t.n0 = (idx / gsz) * 256;
    t.second = t.m0 < p.split_row;
    t.xoff = (uint32_t)((t.b * p.a_bs + (int64_t)t.m0 * p.lda) * ESZ) + (uint32_t)(slice * nt_slice) * 128u;
    t.woff = (uint32_t)((t.b * p.w_bs + (int64_t)t.n0 * p.ldw) * ESZ) + (uint32_t)(slice * nt_slice) * 128u + (t.second ? p.w2_off : 0u);
    return t;
  };

  // ---- request items: piece = 8 consecutive LDS rows written by one wave instruction, 2 pieces per item and wave;
  // 16-byte chunks XOR-swizzled with ((row >> 1) & 7) on the per-lane SOURCE address (LDS image stays lane-linear)
  const int lr = lane >> 3, cphys = lane & 7;
  uint32_t ldst[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = wc * 2 + j;
      if (q == 0 || q == 3) ldst[q][j] = LDS_X + g * 32768 + ((q == 3 ? 64 : 0) + pc * 8) * 128;
      else ldst[q][j] = LDS_W + (g * 128 + (pc >> 2) * 64 + (q == 2 ? 32 : 0) + (pc & 3) * 8) * 128;
    }
  // A request's byte offset = per-lane part + per-piece scalar part + tile origin + K-tile.  The lane part -- row (lane >> 3) of the
  // piece, swizzled chunk -- does not depend on the tile and, because every piece starts on a multiple of 8 rows whose (row / 8)
  // parity is j, only on j:  (row >> 1) & 7 = 4 j + (lr >> 1).  Four VGPRs for the life of the block; the piece's first row
  // is a scalar added per request.  Edge tiles rely on the descriptors' range check (num_records = the operand's extent: rows beyond
  // M of the last batch / beyond N read as zeros into LDS and are never stored; rows beyond M of an inner batch read the next batch).
  int vx[2], vw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int clog = cphys ^ ((j << 2) | (lr >> 1));
    vx[j] = lr * (int)p.lda * ESZ + clog * 16;
    vw[j] = lr * (int)p.ldw * ESZ + clog * 16;
  }
  uint32_t srow[4][2];   // scalar byte offset of the piece's first row
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pc = wc * 2 + j;
      if (q == 0 || q == 3) srow[q][j] = (uint32_t)(g * 128 + (q == 3 ? 64 : 0) + pc * 8) * (uint32_t)(p.lda * ESZ);
      else srow[q][j] = (uint32_t)(g * 128 + (pc >> 2) * 64 + (q == 2 ? 32 : 0) + (pc & 3) * 8) * (uint32_t)(p.ldw * ESZ);
    }
  // num_records: the operand's extent in bytes (< 2^32, persist_ok)
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)(uint32_t)((((int64_t)(p.batch - 1) * p.a_bs + (int64_t)(p.M - 1) * p.lda + p.K) * ESZ)), 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.W, 0, (int)((uint32_t)((((int64_t)(p.batch - 1) * p.w_bs + (int64_t)(p.N - 1) * p.ldw + p.K) * ESZ)) + (p.split_row > 0 ? p.w2_off : 0u)), 0x00020000);

  // request item q of K-tile kt of the tile with origins (xo, wo) into buffer set `set`
  auto stage = [&](int q, uint32_t xo, uint32_t wo, int kt, int set) {
    const bool x_item = (q == 0 || q == 3);
    const uint32_t so = (x_item ? xo : wo) + (uint32_t)kt * 128u;
    const uint32_t setoff = set * (x_item ? 16384u : 32768u);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      // (the piece's row offset is added to the LANE offset, one v_add per request: carried in the scalar offset instead, the
      // K = 12288 shapes -- a 24 KiB row pitch -- lose 10 %, 1.33 against 1.48 PFLOP/s; every other pitch is indifferent)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_item ? rsrcX : rsrcW, (lds_void*)(smem + ldst[q][j] + setoff), 16,
                                               (x_item ? vx[j] : vw[j]) + (int)srow[q][j], so, 0, GLDS_AUX);
  };

  // fragment read addresses (see gemm8p_kernel): bf16 chunk 4 s + (l >> 4) = k-step s; e4m3 chunks 2 (l >> 4) + s = the two
  // halves of the lane's 32 k bytes
  const int r16 = lane & 15, q4 = lane >> 4;
  uint32_t fx[2], fw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int chunk = FP8 ? 2 * q4 + s : 4 * s + q4;
    const uint32_t o = r16 * 128 + ((chunk ^ (r16 >> 1)) << 4);
    fx[s] = LDS_X + g * 32768 + o;
    fw[s] = LDS_W + wc * 64 * 128 + o;
  }

  Tile cur = coords(unit_id(0));
  // ---- prologue (first tile of the block only): K-tiles 0 and 1 complete
#pragma unroll
  for (int q = 0; q < 4; ++q) stage(q, cur.xoff, cur.woff, 0, 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) stage(q, cur.xoff, cur.woff, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFX_BARRIER();
  if (g == 1) TFX_BARRIER();  // stagger: G1 runs one barrier behind G0, for the whole life of the block

  f32x4 acc[8][4];
  bf16x8 xf[4][2], wlo[2][2], whi[2][2];
#define LDS_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PP_MFMA16(WF, MIB, NJB, Z)                                                                           \
  do {                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    mfma_section<FP8, false, 0, Z>(acc, WF, xf, MIB, NJB);                                                   \
    TFX_PIN_SECTION(MIB, NJB);                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)
  // one K-tile out of buffer set SET; (XO, WO, KT) = origins / K-tile of the operands requested meanwhile (two K-tiles ahead)
#define PP_TILE_W(SET, XO, WO, KT, W0, W1, W3, Z)                                                            \
  do {                                                                                                       \
    constexpr uint32_t xs = (SET) * 16384u, ws = (SET) * 32768u;                                             \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {               /* L0: X_lo, W_lo */                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) xf[i][s] = LDS_FRAG(fx[s] + xs + i * 2048);              \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) wlo[j][s] = LDS_FRAG(fw[s] + ws + j * 2048);             \
    }                                                                                                        \
    if (W0) PP_VMCNT(10);                                                                                    \
    TFX_BARRIER();                                                                                           \
    PP_MFMA16(wlo, 0, 0, Z);                                                                                 \
    TFX_BARRIER();                                                                                           \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                 /* L1: W_hi */                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) whi[j][s] = LDS_FRAG(fw[s] + ws + 4096 + j * 2048);      \
    stage(0, XO, WO, KT, SET);                                                                               \
    if (PLACE == 1) { stage(1, XO, WO, KT, SET);     if (W1) PP_VMCNT(12); } else { if (W1) PP_VMCNT(10); }  \
    TFX_BARRIER();                                                                                           \
    PP_MFMA16(whi, 0, 2, Z);                                                                                 \
    TFX_BARRIER();                                                                                           \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                 /* L2: X_hi */                             \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) xf[i][s] = LDS_FRAG(fx[s] + xs + 8192 + i * 2048);       \
    if (PLACE != 1) stage(1, XO, WO, KT, SET);                                                               \
    TFX_BARRIER();                                                                                           \
    PP_MFMA16(whi, 4, 2, Z);                                                                                 \
    TFX_BARRIER();                                                                                           \
    stage(2, XO, WO, KT, SET);                                    /* L3: no reads */                         \
    stage(3, XO, WO, KT, SET);                                                                               \
    if (W3) PP_VMCNT(12);                                                                                    \
    TFX_BARRIER();                                                                                           \
    PP_MFMA16(wlo, 4, 0, Z);                                                                                 \
    TFX_BARRIER();                                                                                           \
  } while (0)
#define PP_TILE(SET, XO, WO, KT) PP_TILE_W(SET, XO, WO, KT, 1, 1, 1, false)

  char* stg = smem + PP_STG + wave * PP_STG_WAVE;
#ifdef TFX_BENCH
  unsigned long long tfx_stamp[4] = {0, 0, 0, 0}, tfx_sum[4] = {0, 0, 0, 0};
#else
  unsigned long long* tfx_stamp = nullptr;
#endif
  for (;;) {
    TFX_STAMP(3);
    const int nit = it + 1;
    const bool has_next = nit < nunits;
    // the last tile of the block re-requests its own first K-tiles: harmless (nobody reads them) and keeps the
    // load counts of the waits uniform
    const Tile nxt = coords(unit_id(has_next ? nit : it));
    const uint32_t cx = cur.xoff, cw = cur.woff, nx = nxt.xoff, nw = nxt.woff;
    // Everything K-tiles 0 and 1 read was waited for before this tile started (prologue / the epilogue's vmcnt(0)), and
    // the first requests of THIS tile have their deadline at L3 of K-tile 1: the earlier counted waits could only stall
    // on the previous epilogue's stores, which retire in issue order with the requests.  The first K-tile's MFMAs start from
    // a zero C operand instead of zeroed accumulators.
    int u0 = 0;
    const int nt = cur.nt;
    // (not in the fp8 gated-residual instantiation: there the two extra loop bodies tip hipcc's allocation into spills.  Round 6: nor in
    // the fp8 q / k-norm instantiation -- round 5 added it with the two bodies and hipcc spilled the request offsets, reloaded them in
    // front of the K loop and then protected the reloaded registers INSIDE the loop with its own s_waitcnt vmcnt(4): three near-drains of
    // the operand prefetch per two K-tiles in the kernel that carries 47 % of an fp8 forward's GEMM FLOPs; tests/test_isa_hazards.py now
    // rejects ANY compiler-inserted vmcnt wait in a K loop.  TFX_FP8_HEAD: A/B builds, 1 = round 5's rule, 0 = no fp8 kernel has the bodies)
    constexpr bool kHead = !FP8 || (TFX_FP8_HEAD == 1 ? EPI != EPI_BIAS_GATE_RES : TFX_FP8_HEAD == 2 ? (EPI != EPI_BIAS_GATE_RES && !QKN) : false);
    if (nt >= 4 && kHead) {
      PP_TILE_W(0, cx, cw, 2, 0, 0, 0, true);
      PP_TILE_W(1, cx, cw, 3, 0, 0, 1, false);
      u0 = 2;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    }
    for (int u = u0; u < nt - 2; u += 2) {
      PP_TILE(0, cx, cw, u + 2);
      PP_TILE(1, cx, cw, u + 3);
    }
    PP_TILE(0, nx, nw, 0);  // K-tiles nt-2, nt-1: their requests are the next tile's K-tiles 0 and 1
    PP_TILE(1, nx, nw, 1);
    TFX_STAMP(0);

    // ---- epilogue (tile_epilogue), one 32-row block of the accumulators at a time
    if (SPLIT && cur.unit >= 0) {
      // raw fp32 partials: lane = row (l & 15) of each 16-row block, 4 consecutive columns per nj -> 16-byte stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int le = lane;
      asm volatile("" : "+v"(le));
      if (p.ws_ld == 0) {
        // compact: this unit's own 256 x 256 fp32 tile (rows beyond M / columns beyond N hold zeros-times-garbage: never read)
        float* wsp = p.ws + (int64_t)cur.unit * 65536 + (g * 128 + (le & 15)) * 256 + wc * 64 + (le >> 4) * 4;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int nj = 0; nj < 4; ++nj) *reinterpret_cast<f32x4*>(wsp + mi * 16 * 256 + nj * 16) = acc[mi][nj];
      } else {
        float* wsp = p.ws + (int64_t)cur.b * p.ws_bs;     // fp32-output mode
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int m = cur.m0 + g * 128 + mi * 16 + (le & 15);
#pragma unroll
          for (int nj = 0; nj < 4; ++nj) {
            const int n = cur.n0 + wc * 64 + nj * 16 + (le >> 4) * 4;
            if (m < p.M && n + 3 < p.N) {
              *reinterpret_cast<f32x4*>(wsp + (int64_t)m * p.ws_ld + n) = acc[mi][nj];
            } else if (m < p.M) {   // ragged right edge
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) wsp[(int64_t)m * p.ws_ld + n + e] = acc[mi][nj][e];
            }
          }
        }
      }
    } else {
      tile_epilogue<EPI, FP8, QKN, (FP8 ? 1 : 2)>(acc, p, cur.m0, cur.n0, cur.b, g, wc, lane, stg,
                                                  smem + PP_STG + (wave ^ 1) * PP_STG_WAVE, tfx_stamp, cur.second);
    }
#ifdef TFX_BENCH
    TFX_STAMP(2);
    tfx_sum[0] += tfx_stamp[0] - tfx_stamp[3];   // K loop (incl. next-tile bookkeeping)
    tfx_sum[1] += tfx_stamp[1] - tfx_stamp[0];   // epilogue: operand / bias requests drained
    tfx_sum[2] += tfx_stamp[2] - tfx_stamp[1];   // epilogue: convert, stage, store issue
    tfx_sum[3] += 1;
#endif
    if (!has_next) break;
    cur = nxt;
    it = nit;
  }
  if (g == 0) TFX_BARRIER();  // pairs with G1's extra barrier of the prologue
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the closing requests must not land in a successor's LDS
#ifdef TFX_BENCH
  if (p.timers && wc == 0 && lane == 0)
    for (int i = 0; i < 4; ++i) atomicAdd(p.timers + g * 4 + i, tfx_sum[i]);
#endif
#undef LDS_FRAG
#undef PP_VMCNT
#undef PP_MFMA16
#undef PP_TILE
#undef PP_TILE_W
}

#ifdef TFX_BENCH   // round 6: measured 11-14 % slower than the ping-pong kernel (profiles/r04_gemm4w_ab.json): bench library only
// ------------------------------------------------------------------------------------------------
// Round 4: the ONE-WAVE-PER-SIMD form of the persistent kernel (tfx_set_option gemm_waves 4).  Same
// 256 x 256 block tile, same tile order, same accumulation order per output element (bit-identical to the other two MFMA kernels),
// but 4 waves of 128 x 128 (8 x 8 accumulators of 16 x 16 = 256 accumulator registers in the AccVGPRs, the whole 512-register file
// per wave): a third fewer LDS fragment reads per MFMA (16 b128 reads feed 64 MFMAs, against 24 for 64 in the 8-wave kernel) and
// no ping-pong -- each wave hides its own reads and requests behind its own MFMAs.
//   * K is walked in 32-deep SUB-tiles (one v_mfma_f32_16x16x32_bf16 k-step): 64-byte rows, 4 LDS sets of 32 KiB (X 256 rows |
//     W 256 rows), one barrier per sub-tile (64 MFMAs per wave).  In iteration t a wave waits for its requests of sub-tile t + 1
//     (vmcnt(16): the two youngest sub-tiles may still be in flight), for its fragment reads of sub-tile t, and meets the others;
//     then it reads the 16 fragments of sub-tile t + 1 into the other fragment set, requests sub-tile t + 4 into the LDS set the
//     fragments of t just left, and issues the 64 MFMAs of t -- every request has three iterations (~3 k cycles) in flight.
//   * requests: one wave instruction = 16 rows x 64 B; the 16-byte chunks of a row are XOR-swizzled with (row >> 2) & 3 on the
//     SOURCE address (the LDS image stays lane-linear, as LDS-DMA requires): conflict-free ds_read_b128 of 16 rows x one chunk.
//   * the requests of a tile's last four sub-tiles fetch the first four of the block's next tile.
//   * epilogue (not yet hidden under the next tile's MFMAs in this version): four 32-row blocks through an 8 KiB wave-private
//     staging tile (256-byte rows, chunks XOR-swizzled with row & 15), whole 256-byte row segments stored by 16 lanes each.  A
//     128-column head lies inside ONE wave: the fused q / k RMSNorm needs neither the partner exchange nor its two barriers.
constexpr int W4G_SET = 32768, W4G_WOFF = 16384, W4G_STG = 131072, W4G_STG_WAVE = 8192, W4G_LDS_TOTAL = W4G_STG + 4 * W4G_STG_WAVE;

template <int EPI, bool QKN>
__global__ __launch_bounds__(256) void gemm4w_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wcn = wave & 1;          // rows wr * 128 .. + 128, columns wcn * 128 .. + 128 of the block tile
  const int nsub = p.K >> 5;                          // 32-deep sub-tiles (a multiple of 4: K % 128 == 0)

  // ---- this block's tiles (as gemm8pp_kernel without K-sliced units)
  const int per_batch = p.tm * p.tn;
  const int T = p.batch * per_batch;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3, bi = blockIdx.x >> 3;
  const int q8 = T >> 3, r8 = T & 7;
  const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  if (bi >= xcnt) return;
  struct Tile { int b, m0, n0; bool second; uint32_t xoff, woff; };
  auto coords = [&](int id) {
    Tile t;
    t.b = id / per_batch;
    int idx = id - t.b * per_batch;
    const int GM = p.gm;
    const int grp = idx / (GM * p.tn);
    const int first_m = grp * GM;
    const int gsz = min(GM, p.tm - first_m);
    idx -= grp * GM * p.tn;
    t.m0 = (first_m + idx % gsz) * 256;
    t.n0 = (idx / gsz) * 256;
    t.second = t.m0 < p.split_row;
    t.xoff = (uint32_t)((t.b * p.a_bs + (int64_t)t.m0 * p.lda) * 2);
    t.woff = (uint32_t)((t.b * p.w_bs + (int64_t)t.n0 * p.ldw) * 2) + (t.second ? p.w2_off : 0u);
    return t;
  };

  // ---- requests: wave w stages rows [64 w, 64 w + 64) of X and of W, four 16-row pieces each
  const int lr = lane >> 2, cphys = lane & 3;
  const int clog = cphys ^ ((lr >> 2) & 3);
  const int vx = lr * (int)p.lda * 2 + clog * 16, vw = lr * (int)p.ldw * 2 + clog * 16;
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)(uint32_t)((((int64_t)(p.batch - 1) * p.a_bs + (int64_t)(p.M - 1) * p.lda + p.K) * 2)), 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.W, 0, (int)((uint32_t)((((int64_t)(p.batch - 1) * p.w_bs + (int64_t)(p.N - 1) * p.ldw + p.K) * 2)) + (p.split_row > 0 ? p.w2_off : 0u)), 0x00020000);
  const int rowx = wave * 64 * (int)p.lda * 2, roww = wave * 64 * (int)p.ldw * 2;       // byte offset of the wave's first row
  const int px = 16 * (int)p.lda * 2, pw = 16 * (int)p.ldw * 2;                          // ... of a piece
  // piece pc (0..3 X, 4..7 W) of sub-tile `sub` of the tile with origins (xo, wo) into LDS set `set`
  auto request = [&](int pc, uint32_t xo, uint32_t wo, int sub, int set) __attribute__((always_inline)) {
    const bool x_item = pc < 4;
    const int k = pc & 3;
    const uint32_t dst = set * W4G_SET + (x_item ? 0 : W4G_WOFF) + (wave * 64 + k * 16) * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_item ? rsrcX : rsrcW, (lds_void*)(smem + dst), 16,
                                             x_item ? vx + rowx + k * px : vw + roww + k * pw, (x_item ? xo : wo) + (uint32_t)sub * 64u, 0,
                                             GLDS_AUX);
  };

  // ---- fragment reads: lane l = row l & 15 of a 16-row block, logical chunk l >> 4 (its 8 k values of the 32-deep step)
  const int r16 = lane & 15, q4 = lane >> 4;
  const uint32_t fl = r16 * 64 + ((q4 ^ ((r16 >> 2) & 3)) << 4);
  const uint32_t fa = wr * 128 * 64 + fl, fb = W4G_WOFF + wcn * 128 * 64 + fl;
#define LDS_FRAG(off) (*reinterpret_cast<const bf16x8*>(smem + (off)))

  f32x4 acc[8][8];
  bf16x8 FA[2][8], FB[2][8];
  char* stg = smem + W4G_STG + wave * W4G_STG_WAVE;

  int it = bi;
  Tile cur = coords(xstart + it);
  // ---- prologue: sub-tiles 0 .. 3 of the first tile
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) request(pc, cur.xoff, cur.woff, q, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TFX_BARRIER();
#pragma unroll
  for (int i = 0; i < 8; ++i) { FA[0][i] = LDS_FRAG(fa + i * 1024); FB[0][i] = LDS_FRAG(fb + i * 1024); }

  // one sub-tile out of LDS set Q (fragments in F[Q & 1]); meanwhile the fragments of the next sub-tile (set Q + 1) are read and sub-tile
  // (RT, origins RX / RW) is requested into set Q.  WAIT: the counted wait (not in a tile's first three sub-tiles: what they read was
  // waited for by the previous epilogue / the prologue, and a counted wait there would only stall on the epilogue's stores).  Z: the
  // tile's first sub-tile starts its accumulators from a zero C operand.
#define W4G_SUB(Q, RX, RW, RT, WAIT, Z)                                                                                     \
  do {                                                                                                                      \
    if (WAIT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                      \
    TFX_BARRIER();                                                                                                          \
    constexpr int cur_ = (Q) & 1, nxt_ = cur_ ^ 1;                                                                          \
    constexpr uint32_t nset_ = (((Q) + 1) & 3) * W4G_SET;                                                                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                         \
      /* per row block: two fragment reads of the next sub-tile and one request in front of its eight MFMAs.  (Measured alternative: */ \
      /* the MFMAs first, the sixteen reads bunched into the first four row blocks and the requests into the last four -- 5 % slower: */ \
      /* what one wave can hide behind a 16-cycle MFMA is one short instruction, a burst of four reads or two requests is not hidden)  */ \
      FA[nxt_][i] = LDS_FRAG(fa + nset_ + i * 1024);                                                                        \
      FB[nxt_][i] = LDS_FRAG(fb + nset_ + i * 1024);                                                                        \
      request(i, RX, RW, RT, Q);                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                         \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[cur_][j], FA[cur_][i], (Z) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                                    \
    }                                                                                                                       \
  } while (0)

  for (;;) {
    const int nit = it + per_xcd;
    const bool has_next = nit < xcnt;
    const Tile nxt = coords(xstart + (has_next ? nit : it));
    const uint32_t cx = cur.xoff, cw = cur.woff, nx = nxt.xoff, nw = nxt.woff;
    W4G_SUB(0, cx, cw, 4, false, true);
    W4G_SUB(1, cx, cw, 5, false, false);
    W4G_SUB(2, cx, cw, 6, false, false);
    W4G_SUB(3, cx, cw, 7, true, false);
    for (int t = 4; t < nsub - 4; t += 4) {
      W4G_SUB(0, cx, cw, t + 4, true, false);
      W4G_SUB(1, cx, cw, t + 5, true, false);
      W4G_SUB(2, cx, cw, t + 6, true, false);
      W4G_SUB(3, cx, cw, t + 7, true, false);
    }
    W4G_SUB(0, nx, nw, 0, true, false);     // the last four sub-tiles request the first four of the next tile
    W4G_SUB(1, nx, nw, 1, true, false);
    W4G_SUB(2, nx, nw, 2, true, false);
    W4G_SUB(3, nx, nw, 3, true, false);

    // ---- epilogue
    {
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int q_e = lane_e >> 4, r_e = lane_e & 15;
      const int m0w = cur.m0 + wr * 128, n0w = cur.n0 + wcn * 128;
      const bf16_t* const p_bias = cur.second ? p.bias2 : p.bias;
      const bf16_t* const p_gate = cur.second ? p.gate2 : p.gate;
      constexpr bool HAS_RES = (EPI == EPI_BIAS_GATE_RES || EPI == EPI_BIAS_RES);
      const bool do_gelu = (EPI == EPI_BIAS_GELU) && (cur.n0 >= p.gelu_from);
      const bool in_q = QKN && cur.n0 >= p.nq0 && cur.n0 < p.nq1;
      const bool norm_tile = QKN && (in_q || (cur.n0 >= p.nk0 && cur.n0 < p.nk1));
      const int rows_ok = min(max(p.M - m0w, 0), 128);
      constexpr uint32_t OOR = 0x80000000u;
      auto uniform_rsrc = [](const void* base, int bytes) {
        const uint64_t a = (uint64_t)base;
        const uint64_t u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
        return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
      };
      const auto rsrcC = uniform_rsrc(p.C + cur.b * p.c_bs + (int64_t)m0w * p.ldc, rows_ok ? (int)(((int64_t)(rows_ok - 1) * p.ldc + p.N) * 2) : 0);
      const auto rsrcR = uniform_rsrc(HAS_RES ? p.res + cur.b * p.r_bs + (int64_t)m0w * p.ldr : p.C,
                                      HAS_RES && rows_ok ? (int)(((int64_t)(rows_ok - 1) * p.ldr + p.N) * 2) : 0);
      const auto rsrcB = uniform_rsrc(p_bias ? p_bias : p.C, p_bias ? p.N * 2 : 0);
      const auto rsrcG = uniform_rsrc(EPI == EPI_BIAS_GATE_RES ? p_gate + cur.b * p.gate_bs : p.C, EPI == EPI_BIAS_GATE_RES ? p.N * 2 : 0);
      const int ncol = n0w + q_e * 4;                          // + j * 16: the lane's 4 accumulator columns of column block j
      const int crow = lane_e >> 4, cchunk = lane_e & 15;      // read-back: row crow + 4 it of the 32-row block, 16-byte chunk cchunk
      const int nst = n0w + cchunk * 8;
      const uint32_t col_off = nst < p.N ? (uint32_t)nst * 2u : OOR;
      u32x2 bsr[8], gtr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bsr[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcB, (ncol + j * 16) * 2, 0, 0);
        if (EPI == EPI_BIAS_GATE_RES) gtr[j] = __builtin_amdgcn_raw_buffer_load_b64(rsrcG, (ncol + j * 16) * 2, 0, 0);
      }
      u32x4 rr[8];
      auto load_res = [&](int blk) {
#pragma unroll
        for (int itr = 0; itr < 8; ++itr)
          rr[itr] = __builtin_amdgcn_raw_buffer_load_b128(rsrcR, (int)((uint32_t)(blk * 32 + itr * 4 + crow) * (uint32_t)(p.ldr * 2) + col_off), 0, GRES_AUX);
      };
      if (HAS_RES) load_res(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): bias / gate / residual, and every operand request up to the next tile's first four sub-tiles
      __builtin_amdgcn_sched_barrier(0);
      float rinv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      u32x4 nw8 = u32x4{0u, 0u, 0u, 0u};
      if (norm_tile) {
        nw8 = *reinterpret_cast<const u32x4*>((in_q ? (cur.second ? p.nq_w2 : p.nq_w) : (cur.second ? p.nk_w2 : p.nk_w)) + cchunk * 8);
        // Linear output in bf16 (what the reference's RMSNorm sees), in place; sum of squares of the lane's 32 columns, then of the
        // row's 128 columns: the four lanes l, l ^ 16, l ^ 32, l ^ 48 hold one row of this wave's head
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float ss = 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const u32x2 br = bsr[j];
            const float bs[4] = {__uint_as_float(br[0] << 16), __uint_as_float(br[0] & 0xffff0000u),
                                 __uint_as_float(br[1] << 16), __uint_as_float(br[1] & 0xffff0000u)};
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              float x0 = acc[i][j][e] + bs[e], x1 = acc[i][j][e + 1] + bs[e + 1];
              round_bf2(x0, x1);
              acc[i][j][e] = x0;
              acc[i][j][e + 1] = x1;
              ss += x0 * x0;
              ss += x1 * x1;
            }
          }
          ss += __shfl_xor(ss, 16, 64);
          ss += __shfl_xor(ss, 32, 64);
          rinv[i] = rsqrtf(ss * (1.0f / 128.0f) + p.n_eps);
        }
      }
      auto store_blocks = [&](auto NORM_T, auto GELU_T) __attribute__((always_inline)) {
        constexpr bool NORM = decltype(NORM_T)::value;
        constexpr bool GELU = decltype(GELU_T)::value;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int i = blk * 2 + h;
            const int R = h * 16 + r_e;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const u32x2 br = bsr[j];
              const float bs[4] = {__uint_as_float(br[0] << 16), __uint_as_float(br[0] & 0xffff0000u),
                                   __uint_as_float(br[1] << 16), __uint_as_float(br[1] & 0xffff0000u)};
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = NORM ? acc[i][j][e] : acc[i][j][e] + bs[e];
              if (GELU) {   // nn.GELU acts on the Linear's bf16 OUTPUT: the reference's rounding point (round 5; it was skipped before)
                round_bf2(v[0], v[1]);
                round_bf2(v[2], v[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
              }
              if (EPI == EPI_BIAS_GATE_RES) {
                const u32x2 gr = gtr[j];
                const float gt[4] = {__uint_as_float(gr[0] << 16), __uint_as_float(gr[0] & 0xffff0000u),
                                     __uint_as_float(gr[1] << 16), __uint_as_float(gr[1] & 0xffff0000u)};
                round_bf2(v[0], v[1]);
                round_bf2(v[2], v[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gt[e] * v[e];
              }
              u32x2 o;
              o[0] = pack_bf2(v[0], v[1]);
              o[1] = pack_bf2(v[2], v[3]);
              // columns j * 16 + q * 4 .. + 4 of row R = 8-byte half (q & 1) of 16-byte chunk j * 2 + (q >> 1), stored at chunk ^ (R & 15)
              *reinterpret_cast<u32x2*>(stg + R * 256 + (((j * 2 + (q_e >> 1)) ^ (R & 15)) << 4) + ((q_e & 1) << 3)) = o;
            }
          }
          u32x4 vals[8];
          f32x4 csr[2];
#pragma unroll
          for (int itr = 0; itr < 8; ++itr) {
            const int row = itr * 4 + crow;
            u32x4 val = *reinterpret_cast<const u32x4*>(stg + row * 256 + ((cchunk ^ (row & 15)) << 4));
            if (NORM) {
              const int mrow = min(m0w + blk * 32 + row, p.M - 1);
              const float* cs = p.rope_cs + (int64_t)(p.rope_pos0 + mrow) * 128 + cchunk * 8;
              csr[0] = *reinterpret_cast<const f32x4*>(cs);
              csr[1] = *reinterpret_cast<const f32x4*>(cs + 4);
              const float rr_row = __shfl(rinv[blk * 2 + (itr >> 2)], row & 15, 64);
              float x[8], wv[8], y[8], o8[8];
              unpack8(val, x);
              unpack8(nw8, wv);
#pragma unroll
              for (int e = 0; e < 8; e += 2) {
                float a0 = x[e] * rr_row, a1 = x[e + 1] * rr_row;
                round_bf2(a0, a1);
                a0 *= wv[e];
                a1 *= wv[e + 1];
                round_bf2(a0, a1);
                y[e] = a0;
                y[e + 1] = a1;
              }
              const f32x4 c0 = csr[0], c1 = csr[1];
              o8[0] = y[0] * c0[0] + (-y[1]) * c0[1];  o8[1] = y[1] * c0[0] + y[0] * c0[1];
              o8[2] = y[2] * c0[2] + (-y[3]) * c0[3];  o8[3] = y[3] * c0[2] + y[2] * c0[3];
              o8[4] = y[4] * c1[0] + (-y[5]) * c1[1];  o8[5] = y[5] * c1[0] + y[4] * c1[1];
              o8[6] = y[6] * c1[2] + (-y[7]) * c1[3];  o8[7] = y[7] * c1[2] + y[6] * c1[3];
              val = pack8(o8);
            }
            if (HAS_RES) {
              float fv[8], fr[8];
              unpack8(val, fv);
              unpack8(rr[itr], fr);
#pragma unroll
              for (int e = 0; e < 8; ++e) fv[e] += fr[e];
              val = pack8(fv);
            }
            vals[itr] = val;
          }
          if (HAS_RES && blk + 1 < 4) {
            __builtin_amdgcn_sched_barrier(0);
            load_res(blk + 1);          // requested before this block's stores (vmcnt retires in issue order)
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int itr = 0; itr < 8; ++itr)
            __builtin_amdgcn_raw_buffer_store_b128(vals[itr], rsrcC,
                                                   (int)((uint32_t)(blk * 32 + itr * 4 + crow) * (uint32_t)(p.ldc * 2) + col_off), 0, GSTORE_AUX);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if (QKN && norm_tile) store_blocks(std::true_type{}, std::false_type{});
      else if (EPI == EPI_BIAS_GELU && do_gelu) store_blocks(std::false_type{}, std::true_type{});
      else store_blocks(std::false_type{}, std::false_type{});
    }
    if (!has_next) break;
    cur = nxt;
    it = nit;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the closing requests must not land in a successor's LDS
#undef LDS_FRAG
#undef W4G_SUB
}
#endif  // TFX_BENCH (gemm4w_kernel)

// Second pass of the K-sliced units: C = epi(sum_s P[s] + bias) over the tail tiles, 8 columns per thread, slices summed in order.
// Thread i -> (tail tile i / 8192, row (i / 32) % 256, columns 8 (i % 32) .. + 8) of ws [slice][tail tile][256][256].
template <int EPI, bool FP8 = false>
__global__ __launch_bounds__(256) void tail_reduce_kernel(GemmParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int TT = p.batch * p.tail_r;
  const int tt = (int)(i >> 13);
  if (tt >= TT) return;
  const int row = (int)(i >> 5) & 255, c8 = (int)i & 31;
  // the tile's coordinates: the last tail_r positions of the sample's (grouped) tile order, as gemm8pp_kernel::coords
  const int per_batch = p.tm * p.tn;
  const int b = tt / p.tail_r;
  int idx = per_batch - p.tail_r + (tt - b * p.tail_r);
  const int GM = p.gm;
  const int grp = idx / (GM * p.tn);
  const int first_m = grp * GM;
  const int gsz = min(GM, p.tm - first_m);
  idx -= grp * GM * p.tn;
  const int m = (first_m + idx % gsz) * 256 + row;
  const int n = (idx / gsz) * 256 + c8 * 8;
  if (m >= p.M || n >= p.N) return;
  const bool second = (m - row) < p.split_row;
  const int64_t slice_stride = (int64_t)TT * 65536;
  const float* src = p.ws + (int64_t)tt * 65536 + row * 256 + c8 * 8;
  float v[8];
  // Round 6: everything this thread needs from memory is requested up front -- bias, gate, residual and ALL slices (the slice loop used to
  // be a runtime loop of load / wait / add: sk dependent round trips in a kernel that is 11-15 us of pure latency at batch 1); the slices
  // are still ADDED in order (deterministic, the same bits as before)
  const bf16_t* bias = second ? p.bias2 : p.bias;
  u32x4 braw = u32x4{0u, 0u, 0u, 0u}, graw = braw, rraw = braw;
  if (bias) braw = *reinterpret_cast<const u32x4*>(bias + n);
  if (EPI == EPI_BIAS_GATE_RES) graw = *reinterpret_cast<const u32x4*>((second ? p.gate2 : p.gate) + b * p.gate_bs + n);
  if (EPI == EPI_BIAS_GATE_RES || EPI == EPI_BIAS_RES) rraw = *reinterpret_cast<const u32x4*>(p.res + b * p.r_bs + (int64_t)m * p.ldr + n);
  auto sum_slices = [&](auto SKc) __attribute__((always_inline)) {
    constexpr int SK = decltype(SKc)::value;
    f32x4 a[SK][2];
#pragma unroll
    for (int s = 0; s < SK; ++s) {
      a[s][0] = *reinterpret_cast<const f32x4*>(src + s * slice_stride);
      a[s][1] = *reinterpret_cast<const f32x4*>(src + s * slice_stride + 4);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[0][0][e]; v[4 + e] = a[0][1][e]; }
#pragma unroll
    for (int s = 1; s < SK; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[s][0][e]; v[4 + e] += a[s][1][e]; }
  };
  switch (p.sk) {      // plan_slices: 2, 3, 4, 6 or 8 (1 = the fp32-output mode never comes here)
    case 2: sum_slices(std::integral_constant<int, 2>{}); break;
    case 3: sum_slices(std::integral_constant<int, 3>{}); break;
    case 4: sum_slices(std::integral_constant<int, 4>{}); break;
    case 6: sum_slices(std::integral_constant<int, 6>{}); break;
    case 8: sum_slices(std::integral_constant<int, 8>{}); break;
    default: {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
      for (int s = 1; s < p.sk; ++s) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(src + s * slice_stride);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(src + s * slice_stride + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
    }
  }
  if (FP8) {  // dequantise: row scale, then channel scale (the order of the unsplit epilogue)
    const float sa = p.a_scale[b * p.as_bs + m];
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(p.w_scale + n), w1 = *reinterpret_cast<const f32x4*>(p.w_scale + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (v[e] * sa) * w0[e]; v[4 + e] = (v[4 + e] * sa) * w1[e]; }
  }
  if (bias) {
    float bs[8];
    unpack8(braw, bs);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bs[e];
  }
  if (EPI == EPI_BIAS_GELU && n >= p.gelu_from) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh(round_bf(v[e]));
  }
  if (EPI == EPI_BIAS_GATE_RES) {
    float gt[8];
    unpack8(graw, gt);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gt[e] * round_bf(v[e]);
  }
  u32x4 val = pack8(v);
  if (EPI == EPI_BIAS_GATE_RES || EPI == EPI_BIAS_RES) {
    float fv[8], fr[8];
    unpack8(val, fv);
    unpack8(rraw, fr);
#pragma unroll
    for (int e = 0; e < 8; ++e) fv[e] += fr[e];
    val = pack8(fv);
  }
  *reinterpret_cast<u32x4*>(p.C + b * p.c_bs + (int64_t)m * p.ldc + n) = val;
}

// ------------------------------------------------------------------------------------------------
// MFMA-only probe (tfx_mfma_peak_probe; round 6): the matrix-pipe rate this BOARD sustains on the caller's
// operand data at its power cap, measured by the product library under whatever clock runs the bench -- the denominator of
// roofline.frac_of_capped.  One workgroup per CU, 8 waves as in the GEMMs (two per SIMD); every wave loads its A / B fragments ONCE
// from the caller's buffer (random N(0, 1) bf16 / e4m3 values: power depends on operand entropy, zeros run 35 % faster) and then
// issues nothing but the GEMM kernels' own MFMA sections on registers -- v_mfma_f32_16x16x32_bf16 (fp8: v_mfma_scale_f32_16x16x128_f8f6f4,
// unit scales) into 8 x 4 accumulators, no dependent pair back to back, accumulators restarted from a zero C operand every 48 K-tiles
// like a K = 3072 tile -- no LDS, no global traffic, no barrier.  FLOPs per wave and K-tile: 64 MFMAs x 16384 (fp8: 32 x 65536).
template <bool FP8>
__global__ __launch_bounds__(512) void mfma_probe_kernel(const u32x4* __restrict__ src, int nfrag, int ktiles, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 xf[4][2], wlo[2][2], whi[2][2];
  // 16 fragments per lane, different for every lane / wave / workgroup (mod the buffer)
  unsigned f0 = ((unsigned)blockIdx.x * 8u + (unsigned)wave) * 64u * 16u + (unsigned)lane;
  auto ld = [&](int i) { return __builtin_bit_cast(bf16x8, src[(f0 + (unsigned)i * 64u) % (unsigned)nfrag]); };
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) xf[i][s] = ld(i * 2 + s);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) { wlo[j][s] = ld(8 + j * 2 + s); whi[j][s] = ld(12 + j * 2 + s); }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define PROBE_SECTION(WF, MIB, NJB, Z)                      \
  do {                                                     \
    mfma_section<FP8, false, 0, Z>(acc, WF, xf, MIB, NJB); \
    TFX_PIN_SECTION(MIB, NJB);                             \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
  for (int t = 0; t < ktiles; t += 48) {
    // (the fragments are loop invariants, the two row halves share them, and an MFMA from a zero C operand is a pure function of its
    // operands: without the opaque copies the compiler computes the first K-tile's products once and copies them in)
#define PROBE_OPAQUE()                                                                      \
  do {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xf[i][0]));        \
  } while (0)
    PROBE_OPAQUE(); PROBE_SECTION(wlo, 0, 0, true);
    PROBE_OPAQUE(); PROBE_SECTION(whi, 0, 2, true);
    PROBE_OPAQUE(); PROBE_SECTION(whi, 4, 2, true);
    PROBE_OPAQUE(); PROBE_SECTION(wlo, 4, 0, true);
#undef PROBE_OPAQUE
    for (int u = 1; u < 48; ++u) {
      PROBE_SECTION(wlo, 0, 0, false); PROBE_SECTION(whi, 0, 2, false); PROBE_SECTION(whi, 4, 2, false); PROBE_SECTION(wlo, 4, 0, false);
    }
  }
#undef PROBE_SECTION
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (r == 123456.789f) sink[0] = r;     // keeps the accumulators live; (practically) never taken
}

int mfma_peak_probe(const void* operands, int64_t operand_bytes, int fp8, int ktiles, double* flops, hipStream_t st) {
  if (!operands || (uintptr_t)operands % 16 || operand_bytes < 16 * 64 * 16) return fail("mfma_peak_probe: operands must be a 16-byte aligned buffer of at least 16 KiB");
  if (ktiles < 48 || ktiles % 48) return fail("mfma_peak_probe: ktiles must be a positive multiple of 48");
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  static float* sink = nullptr;      // 4 bytes, never written in practice (the kernel needs a live side effect)
  if (!sink && hipMalloc((void**)&sink, 256) != hipSuccess) return fail("mfma_peak_probe: hipMalloc failed");
  const int nfrag = (int)std::min<int64_t>(operand_bytes / 16, 1 << 30);
  if (fp8) mfma_probe_kernel<true><<<cus, 512, 0, st>>>((const u32x4*)operands, nfrag, ktiles, sink);
  else mfma_probe_kernel<false><<<cus, 512, 0, st>>>((const u32x4*)operands, nfrag, ktiles, sink);
  if (flops) *flops = (double)cus * 8.0 * (double)ktiles * (fp8 ? 32.0 * 65536.0 : 64.0 * 16384.0);
  return check_launch("mfma_peak_probe");
}

// ------------------------------------------------------------------------------------------------
static int g_gemm_splitk = 2;     // K-sliced work units: 0 never (A/B knob), 1 only GEMMs with fewer tiles than CUs (round 3), 2 also the last round of small GEMMs
void set_gemm_splitk(int v) { g_gemm_splitk = v; }
static int g_gemm_waves = 8;      // 8: the ping-pong kernel (gemm8pp_kernel); 4: the one-wave-per-SIMD kernel (gemm4w_kernel) where eligible
void set_gemm_waves(int v) { g_gemm_waves = v == 4 ? 4 : 8; }
static int g_gemm_place = 2;      // request placement of the persistent kernel (bench knob, see gemm8pp_kernel)
void set_gemm_place(int v) { g_gemm_place = v; }
#ifdef TFX_BENCH
static unsigned long long* g_gemm_timers = nullptr;   // device buffer of 8 counters (tools/gemm_phase_timers.py)
extern "C" void tfx_bench_gemm_timers(unsigned long long* dev) { g_gemm_timers = dev; }
#endif
// row tiles per group of the tile order (L2 locality).  0 = by shape: GEMMs with at most 12 column tiles (N <= 3072: the
// gated-residual projections) run 1.4-1.9 % faster with 1, everything wider prefers 4 (tools/gemm_group_sweep.py)
static int g_gemm_group_m = 0;
void set_gemm_group_m(int gm) { g_gemm_group_m = gm < 0 ? 0 : gm; }

static GemmParams make_params(const GemmArgs& a) {
  GemmParams p;
  p.A = (const bf16_t*)a.A; p.lda = a.lda; p.a_bs = a.a_bstride;
  p.W = (const bf16_t*)a.W; p.ldw = a.ldw; p.w_bs = a.w_bstride;
  p.bias = (const bf16_t*)a.bias;
  p.C = (bf16_t*)a.C; p.ldc = a.ldc; p.c_bs = a.c_bstride;
  p.M = a.M; p.N = a.N; p.K = a.K; p.batch = a.batch;
  p.tm = (a.M + 255) / 256; p.tn = (a.N + 255) / 256;
  p.gm = g_gemm_group_m ? g_gemm_group_m : (p.tn <= 12 ? 1 : 4);
  p.gelu_from = a.gelu_from_col;
#ifdef TFX_BENCH
  p.timers = g_gemm_timers;
#endif
  p.gate = (const bf16_t*)a.gate; p.gate_bs = a.gate_bstride;
  p.res = (const bf16_t*)a.res; p.ldr = a.ldr; p.r_bs = a.r_bstride;
  p.cin = a.conv_cin; p.inH = a.conv_inH; p.inW = a.conv_inW; p.oH = a.conv_H; p.oW = a.conv_W;
  p.cstride = a.conv_stride; p.cup = a.conv_up_shift; p.cpad = a.conv_pad_lo; p.zero = (const bf16_t*)a.zero_page;
  p.ckw = a.conv_kw; p.cstride_x = a.conv_stride_x > 0 ? a.conv_stride_x : a.conv_stride;
  p.csh = a.conv_cin == 8 ? 3 : a.conv_cin == 16 ? 4 : a.conv_cin == 32 ? 5 : 0;
  p.a_scale = a.a_scale; p.as_bs = a.a_scale_bstride; p.w_scale = a.w_scale;
  p.ws = nullptr; p.sk = 1; p.u_full = 0; p.tail_r = 0; p.ws_ld = 0; p.ws_bs = 0;
  p.split_row = a.split_row; p.w2_off = 0;
  p.bias2 = (const bf16_t*)a.bias2; p.gate2 = (const bf16_t*)a.gate2; p.nq_w2 = (const bf16_t*)a.qkn_wq2; p.nk_w2 = (const bf16_t*)a.qkn_wk2;
  if (a.split_row > 0) p.w2_off = (uint32_t)((const char*)a.W2 - (const char*)a.W);
  p.nq_w = (const bf16_t*)a.qkn_wq; p.nk_w = (const bf16_t*)a.qkn_wk; p.rope_cs = a.qkn_rope_cs; p.rope_pos0 = a.qkn_pos0;
  p.nq0 = a.qkn_q0; p.nq1 = a.qkn_q1; p.nk0 = a.qkn_k0; p.nk1 = a.qkn_k1; p.n_eps = a.qkn_eps;
  return p;
}

static bool conv_ok(const GemmArgs& a) {
  const bool wide = a.conv_cin % 64 == 0 && (a.conv_kw == 3 || a.conv_kw == 4) && a.K == 3 * a.conv_kw * a.conv_cin;
  const bool narrow = a.conv_kw == 3 && (a.conv_cin == 8 || a.conv_cin == 16 || a.conv_cin == 32) && a.K == (9 * a.conv_cin + 63) / 64 * 64;
  return (wide || narrow) && a.batch == 1 && a.zero_page && (uintptr_t)a.zero_page % 16 == 0 &&
         a.conv_H > 0 && a.conv_W > 0 && (a.conv_inH << a.conv_up_shift) < 32768 && (a.conv_inW << a.conv_up_shift) < 32768 &&
         (int64_t)a.M * 1 == (int64_t)a.conv_H * a.conv_W * (a.M / (a.conv_H * a.conv_W)) && a.M % (a.conv_H * a.conv_W) == 0;
}

static bool fast_ok(const GemmArgs& a) {
  if (a.conv_cin > 0 && !conv_ok(a)) return false;
  const bool al16 = ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.C % 16 == 0) &&
                    ((uintptr_t)a.bias % 8 == 0);
  return a.K % 64 == 0 && a.K >= 64 && a.N % 8 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 8 == 0 && a.w_bstride % 8 == 0 && a.w_bstride >= 0 &&
         a.a_bstride % 8 == 0 && a.c_bstride % 8 == 0 && al16 && (int64_t)255 * a.lda < (1ll << 31) &&
         (int64_t)255 * a.ldw < (1ll << 31) &&
         ((a.epilogue != EPI_BIAS_GATE_RES && a.epilogue != EPI_BIAS_RES) || (a.ldr % 8 == 0 && a.r_bstride % 8 == 0 && a.gate_bstride % 4 == 0 &&
                                              (uintptr_t)a.res % 16 == 0 && (uintptr_t)a.gate % 8 == 0)) &&
         (a.epilogue != EPI_BIAS_GELU || a.gelu_from_col % 256 == 0);
}

// Shapes the persistent kernel takes: an even number of K-tiles and every operand byte offset inside 32 bits (the
// tile origin travels in the scalar offset of the buffer load).  Everything else the MFMA path accepts goes to the
// one-tile kernel.
static bool persist_ok(const GemmParams& p) {
  return p.cin == 0 && p.K % 128 == 0 &&
         ((int64_t)(p.batch - 1) * p.a_bs + (int64_t)p.tm * 256 * p.lda) * 2 < (1ll << 32) - 65536 &&
         ((int64_t)(p.batch - 1) * p.w_bs + (int64_t)p.tn * 256 * p.ldw) * 2 < (1ll << 32) - 65536;
}

// Which work units of a persistent-kernel GEMM are K-sliced (GemmParams::sk / u_full / tail_r).  T < grid: every tile, so that
// (tile, slice) units fill the chip.  Otherwise, for SMALL GEMMs only (fewer than 8 rounds of the chip -- beyond that the partly
// filled round is under a ninth of the time and the fp32 round trip of its tiles costs what the slicing saves): the R = T mod grid
// tiles of the last round when one short round of slices can take them (R c <= grid), the same tile positions in every batch
// sample (identical samples keep identical bits).  Slices keep an even number >= 8 of K-tiles; ws: 256 KiB per unit.
// (Round 6 measured the generalisation -- the last round of a LONG-K GEMM with >= 8 rounds or R c > grid as several short rounds of slices:
// 1728 tiles, R = 192 as 768 quarter-K units -- and it LOSES: +40 us / +67 us per launch at K = 15360 / 12288, +1.1 % per forward; the
// partials of every CU leave at the same moment and come back in the second pass.  profiles/r06_splitk_long_k_ab.log.)
struct SlicePlan { int sk, u_full, tail_r; };
static SlicePlan plan_slices(const GemmParams& p, int grid, int nt, void* ws, int64_t ws_bytes) {
  SlicePlan pl{1, 0, 0};
  if (!g_gemm_splitk || !ws || p.N % 8) return pl;
  const int per_batch = p.tm * p.tn, T = p.batch * per_batch;
  auto ok = [&](int units, int c) { return units * c <= grid && nt % (2 * c) == 0 && nt / c >= 8 && (int64_t)units * c * 262144 <= ws_bytes; };
  if (T < grid) {
    for (int c : {8, 6, 4, 3, 2})
      if (ok(T, c)) return SlicePlan{c, 0, per_batch};
    return pl;
  }
  const int R = T % grid;
  if (R == 0 || T / grid >= 8 || R % p.batch || g_gemm_splitk < 2) return pl;
  for (int c : {4, 3, 2})
    if (ok(R, c)) return SlicePlan{c, T - R, R / p.batch};
  return pl;
}

#ifdef TFX_BENCH
template <int ABL>
static int launch_ablation(const GemmParams& p, hipStream_t st) {
  (void)hipFuncSetAttribute((const void*)gemm8p_kernel<EPI_BIAS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  gemm8p_kernel<EPI_BIAS, ABL><<<(unsigned)(p.batch * p.tm * p.tn), 512, LDS_TOTAL, st>>>(p);
  return check_launch("gemm_ablation");
}
#endif  // TFX_BENCH

template <int EPI>
static int launch_variant(const GemmParams& p, int variant, void* ws, int64_t ws_bytes, hipStream_t st) {
  if (variant >= 10) {  // tools/bench_kernels.py only: timing ablations with WRONG results by construction
#ifndef TFX_BENCH
    return fail("gemm: ablation variants are bench-only (build with -DTFX_BENCH)");
#else
    switch (variant - 10) {
      case 1: return launch_ablation<1>(p, st);
      case 2: return launch_ablation<2>(p, st);
      case 3: return launch_ablation<3>(p, st);
      case 4: return launch_ablation<4>(p, st);
      case 7: return launch_ablation<7>(p, st);
      case 8: return launch_ablation<8>(p, st);
      case 16: return launch_ablation<16>(p, st);
      case 17: return launch_ablation<17>(p, st);
      case 18: return launch_ablation<18>(p, st);
      case 32768: return launch_ablation<32768>(p, st);
      case 65536: return launch_ablation<65536>(p, st);
      case 98304: return launch_ablation<98304>(p, st);
    }
    return fail("gemm: unknown ablation");
#endif
  }
  if ((variant == 1 || variant == 3) && persist_ok(p)) {
    static int grid = 0;
    if (!grid) {
      int dev = 0, cus = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
      const void* fns[2] = {(const void*)gemm8pp_kernel<EPI, 1>, (const void*)gemm8pp_kernel<EPI, 2>};
      for (const void* fn : fns) {
        hipFuncAttributes fa;  // forces the (lazily loaded) code object in before the attribute is set
        (void)hipFuncGetAttributes(&fa, fn);
        (void)hipGetLastError();
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL);
        if (e != hipSuccess)
          return fail("gemm: cannot raise dynamic LDS limit to %d bytes: %s", PP_LDS_TOTAL, hipGetErrorString(e));
      }
      grid = cus & ~7;  // one block per CU (160 KiB of LDS each), a whole number per XCD
    }
    const bool prof = prof_on(st);
    if (prof) prof_begin(0, 2.0 * p.M * (double)p.N * p.K * p.batch, st);
    // K-sliced units (variant 1 = auto only; the caller must have passed a workspace for the fp32 partials): plan_slices
    const int nt = p.K >> 6;
    const SlicePlan pl = variant == 1 ? plan_slices(p, grid, nt, ws, ws_bytes) : SlicePlan{1, 0, 0};
    const int sk = pl.sk;
#ifdef TFX_BENCH
    if (g_gemm_waves == 4 && sk == 1 && p.K >= 256) {   // one wave per SIMD (tfx_set_option gemm_waves 4): every unsliced bf16 launch
      if (p.rope_cs && EPI != EPI_BIAS_GELU) return fail("gemm: the q/k norm + RoPE epilogue rides on the bias(+GELU) epilogue");
      static bool attr4[2] = {false, false};
      const int qi = p.rope_cs ? 1 : 0;
      const void* fn = qi ? (const void*)gemm4w_kernel<EPI_BIAS_GELU, true> : (const void*)gemm4w_kernel<EPI, false>;
      if (!attr4[qi]) {
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, fn);
        (void)hipGetLastError();
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W4G_LDS_TOTAL) != hipSuccess)
          return fail("gemm: cannot raise dynamic LDS limit for the 4-wave kernel");
        attr4[qi] = true;
      }
      if (qi) gemm4w_kernel<EPI_BIAS_GELU, true><<<grid, 256, W4G_LDS_TOTAL, st>>>(p);
      else gemm4w_kernel<EPI, false><<<grid, 256, W4G_LDS_TOTAL, st>>>(p);
    } else
#endif
    if (p.rope_cs) {   // fused q / k RMSNorm + RoPE epilogue: EPI_BIAS_GELU instantiation only (plain bias = gelu_from >= N)
      // (round 6 tried to let it ride on launches whose last round is K-sliced -- the batch-1 geometries, where the sliced tiles are mlp
      // columns -- through a <.., SPLIT, QKN> instantiation: 256 VGPRs + 9-12 spilled, and 0.3 % SLOWER per 576 x 512 image than the
      // unfused launch + the separate 12 us pass it replaces, profiles/r06_batch1_experiments.md; not kept)
      if (sk > 1) return fail("gemm: the q/k norm + RoPE epilogue cannot ride on a K-sliced launch (gemm_qkn_ok)");
      if (EPI != EPI_BIAS_GELU) return fail("gemm: the q/k norm + RoPE epilogue rides on the bias(+GELU) epilogue");
      static bool attrq = false;
      const void* fn = (const void*)gemm8pp_kernel<EPI_BIAS_GELU, 2, false, false, true>;
      if (!attrq) {
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, fn);
        (void)hipGetLastError();
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL) != hipSuccess)
          return fail("gemm: cannot raise dynamic LDS limit for the q/k-norm kernel");
        attrq = true;
      }
      gemm8pp_kernel<EPI_BIAS_GELU, 2, false, false, true><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
    } else if (sk > 1) {
      static bool attr = false;
      if (!attr) {
        const void* fns[2] = {(const void*)gemm8pp_kernel<EPI_BIAS, 2, false, true>, (const void*)gemm8pp_kernel<EPI, 2, false, true>};
        for (const void* fn : fns) {
          hipFuncAttributes fa;
          (void)hipFuncGetAttributes(&fa, fn);
          (void)hipGetLastError();
          if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL) != hipSuccess)
            return fail("gemm: cannot raise dynamic LDS limit for the K-sliced kernel");
        }
        attr = true;
      }
      GemmParams ps = p;
      ps.ws = (float*)ws; ps.sk = sk; ps.u_full = pl.u_full; ps.tail_r = pl.tail_r;
      if (pl.u_full == 0)     // every tile sliced: no whole-tile epilogue in the stream, the bias-only instantiation serves every EPI
        gemm8pp_kernel<EPI_BIAS, 2, false, true><<<grid, 512, PP_LDS_TOTAL, st>>>(ps);
      else
        gemm8pp_kernel<EPI, 2, false, true><<<grid, 512, PP_LDS_TOTAL, st>>>(ps);
      tail_reduce_kernel<EPI><<<(unsigned)(p.batch * pl.tail_r * 32), 256, 0, st>>>(ps);
    } else if (g_gemm_place == 1) {
      gemm8pp_kernel<EPI, 1><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
    } else {
      gemm8pp_kernel<EPI, 2><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
    }
    if (prof) prof_end(0, st);
    return check_launch("gemm_bf16");
  }
  if (variant == 3) return fail("gemm: shape not eligible for the persistent kernel");
  if (variant == 1 || variant == 2) {
    const int conv = p.cin > 0 ? (p.csh ? 2 : 1) : 0;
    static bool attr_set[3] = {false, false, false};
    if (!attr_set[conv]) {
      const void* fn = conv == 2 ? (const void*)gemm8p_kernel<EPI, 0, 2> : conv == 1 ? (const void*)gemm8p_kernel<EPI, 0, 1>
                                                                                  : (const void*)gemm8p_kernel<EPI, 0, 0>;
      hipFuncAttributes fa;  // forces the (lazily loaded) code object in before the attribute is set
      (void)hipFuncGetAttributes(&fa, fn);
      (void)hipGetLastError();
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
      if (e != hipSuccess)
        return fail("gemm: cannot raise dynamic LDS limit to %d bytes: %s", LDS_TOTAL, hipGetErrorString(e));
      attr_set[conv] = true;
    }
    const unsigned grid = (unsigned)(p.batch * p.tm * p.tn);
    const bool prof = prof_on(st) && !conv;
    if (prof) prof_begin(0, 2.0 * p.M * (double)p.N * p.K * p.batch, st);
    if (conv == 2)
      gemm8p_kernel<EPI, 0, 2><<<grid, 512, LDS_TOTAL, st>>>(p);
    else if (conv == 1)
      gemm8p_kernel<EPI, 0, 1><<<grid, 512, LDS_TOTAL, st>>>(p);
    else
      gemm8p_kernel<EPI, 0, 0><<<grid, 512, LDS_TOTAL, st>>>(p);
    if (prof) prof_end(0, st);
  } else {
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.batch);
    gemm_generic_kernel<EPI><<<grid, 256, 0, st>>>(p);
  }
  return check_launch("gemm_bf16");
}

int gemm_bf16_variant(const GemmArgs& a, int variant, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return 0;
  if (a.K <= 0) return fail("gemm: K must be positive");
  if (variant >= 1 && !fast_ok(a)) return fail("gemm: shape/alignment not supported by the MFMA kernel");
  if (a.epilogue == EPI_BIAS_GATE_RES && (!a.gate || !a.res)) return fail("gemm: gate/res pointers required");
  if (a.epilogue == EPI_BIAS_RES && !a.res) return fail("gemm: res pointer required");
  const GemmParams p = make_params(a);
  if (a.split_row > 0 && !((variant == 1 || variant == 3) && gemm_rowsplit_ok(a))) return fail("gemm: row-split weights need the persistent MFMA kernel");
  switch (a.epilogue) {
    case EPI_BIAS: return launch_variant<EPI_BIAS>(p, variant, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_GELU: return launch_variant<EPI_BIAS_GELU>(p, variant, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_GATE_RES: return launch_variant<EPI_BIAS_GATE_RES>(p, variant, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_RES: return launch_variant<EPI_BIAS_RES>(p, variant, a.workspace, a.workspace_bytes, st);
  }
  return fail("gemm: unknown epilogue %d", a.epilogue);
}

// Row-split weights need the persistent kernel (the tile origin decides the weight set), split_row on a tile boundary, the second
// weight matrix behind the first inside one 32-bit descriptor range, and both bias / gate pointers (or neither).
bool gemm_rowsplit_ok(const GemmArgs& a) {
  if (a.split_row <= 0 || a.split_row % 256 || a.split_row >= a.M || !fast_ok(a) || a.conv_cin > 0 || !a.W2 || a.w_bstride) return false;
  const GemmParams p = make_params(a);
  if (!persist_ok(p)) return false;
  const int64_t d = (const char*)a.W2 - (const char*)a.W;
  if (d <= 0 || d % 16 || d + (int64_t)p.tn * 256 * p.ldw * 2 >= (1ll << 32) - 65536) return false;
  if ((a.bias == nullptr) != (a.bias2 == nullptr) || (uintptr_t)a.bias2 % 8) return false;
  if (a.epilogue == EPI_BIAS_GATE_RES && (!a.gate2 || (uintptr_t)a.gate2 % 8)) return false;
  if (a.qkn_rope_cs && (!a.qkn_wq2 || !a.qkn_wk2)) return false;
  return true;
}

int gemm_bf16(const GemmArgs& a, hipStream_t st) {
  if (a.split_row > 0 && !gemm_rowsplit_ok(a)) return fail("gemm: shape / layout not eligible for row-split weights (gemm_rowsplit_ok)");
  if (a.qkn_rope_cs && !gemm_qkn_ok(a)) return fail("gemm: shape not eligible for the fused q/k norm + RoPE epilogue");
  return gemm_bf16_variant(a, fast_ok(a) ? 1 : 0, st);
}

// The fused q/k norm + RoPE epilogue needs the persistent kernel, whole heads per tile pair (column ranges on 256-column
// tile boundaries), the bias(+GELU) epilogue, and enough tiles that the auto path would not split K (the split-K reduce
// pass has no such epilogue; few-tile GEMMs -- the text stream at small batch -- keep the separate rmsnorm_rope kernel).
bool gemm_qkn_ok(const GemmArgs& a) {
  if (!fast_ok(a) || a.conv_cin > 0 || (a.epilogue != EPI_BIAS_GELU && a.epilogue != EPI_BIAS)) return false;
  const GemmParams p = make_params(a);
  if (!persist_ok(p)) return false;
  if ((a.qkn_q0 | a.qkn_q1 | a.qkn_k0 | a.qkn_k1) % 256) return false;
  // the normalised columns carry no activation (the epilogue instantiates norm and GELU tiles separately)
  if (a.epilogue == EPI_BIAS_GELU && a.gelu_from_col < (a.qkn_q1 > a.qkn_k1 ? a.qkn_q1 : a.qkn_k1)) return false;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  // ... and no K-sliced units in the launch the auto path would make with this workspace (tail_reduce_kernel has no such epilogue)
  return plan_slices(p, cus & ~7, p.K >> 6, a.workspace, a.workspace_bytes).sk == 1;
}

// The same question for the e4m3 path (gemm_fp8): it slices K only when the whole GEMM has fewer tiles than CUs, so the fused epilogue
// is available exactly when it does not (a: the bf16-side description of the Linear, as tfx_dit_forward builds it).
bool gemm_fp8_qkn_ok(const GemmArgs& a) {
  if (a.conv_cin > 0 || (a.epilogue != EPI_BIAS_GELU && a.epilogue != EPI_BIAS) || a.split_row > 0) return false;
  if (a.K <= 0 || a.K % 256 || a.N % 8 || (a.qkn_q0 | a.qkn_q1 | a.qkn_k0 | a.qkn_k1) % 256) return false;
  if (a.epilogue == EPI_BIAS_GELU && (a.gelu_from_col % 256 || a.gelu_from_col < (a.qkn_q1 > a.qkn_k1 ? a.qkn_q1 : a.qkn_k1))) return false;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
  const int64_t T = (int64_t)a.batch * ((a.M + 255) / 256) * ((a.N + 255) / 256);
  return T >= (cus & ~7);
}

// ---- fp32 output (raw accumulators, no epilogue): C [batch][M, N] floats with row stride ldc.  The score GEMM of the
// VAE mid-block attention: q k^T must reach the softmax unrounded.  The persistent kernel in its (tile, slice) form with
// ONE slice writes exactly that; shapes it does not take (K % 128 != 0) go to the generic kernel.
int gemm_bf16_f32out(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return 0;
  if (a.K <= 0) return fail("gemm_f32out: K must be positive");
  if (a.epilogue != EPI_BIAS || a.bias) return fail("gemm_f32out: raw accumulators only (no bias / epilogue)");
  if (a.conv_cin > 0) return fail("gemm_f32out: no convolution mode");
  if ((uintptr_t)a.C % 16 || a.ldc % 4 || a.c_bstride % 4) return fail("gemm_f32out: C must be 16-byte aligned, ldc / c_bstride multiples of 4");
  GemmParams p = make_params(a);
  p.ws = (float*)a.C; p.sk = 1; p.u_full = 0; p.tail_r = p.tm * p.tn; p.ws_ld = a.ldc; p.ws_bs = a.c_bstride;   // every tile a raw-store unit
  GemmArgs chk = a;
  chk.C = const_cast<void*>(a.A);   // alignment of the fp32 C was checked above; the remaining fast-path conditions are operand-side
  chk.ldc = 8; chk.c_bstride = 8;
  if (fast_ok(chk) && persist_ok(p)) {
    static int grid = 0;
    if (!grid) {
      int dev = 0, cus = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
      const void* fn = (const void*)gemm8pp_kernel<EPI_BIAS, 2, false, true>;
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, fn);
      (void)hipGetLastError();
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL) != hipSuccess)
        return fail("gemm_f32out: cannot raise dynamic LDS limit");
      grid = cus & ~7;
    }
    const bool prof = prof_on(st);
    if (prof) prof_begin(0, 2.0 * p.M * (double)p.N * p.K * p.batch, st);
    gemm8pp_kernel<EPI_BIAS, 2, false, true><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
    if (prof) prof_end(0, st);
  } else {
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.batch);
    gemm_generic_kernel<EPI_BIAS><<<grid, 256, 0, st>>>(p);
  }
  return check_launch("gemm_f32out");
}

// ---- fp8 (e4m3 x e4m3 -> fp32) path: the persistent kernel only
template <int EPI>
static int launch_fp8(const GemmParams& p, void* ws, int64_t ws_bytes, hipStream_t st) {
  static int grid = 0;
  if (!grid) {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    const void* fns[2] = {(const void*)gemm8pp_kernel<EPI, 2, true>, (const void*)gemm8pp_kernel<EPI_BIAS, 2, true, true>};
    for (const void* fn : fns) {
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, fn);
      (void)hipGetLastError();
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL);
      if (e != hipSuccess) return fail("gemm_fp8: cannot raise dynamic LDS limit to %d bytes: %s", PP_LDS_TOTAL, hipGetErrorString(e));
    }
    grid = cus & ~7;
  }
  const bool prof = prof_on(st);
  if (prof) prof_begin(2, 2.0 * p.M * (double)p.N * p.K * p.batch, st);
  const int T = p.batch * p.tm * p.tn, nt = p.K >> 7;   // 128-byte K-tiles of e4m3
  SlicePlan pl = T < grid ? plan_slices(p, grid, nt, ws, ws_bytes) : SlicePlan{1, 0, 0};   // fp8: whole-GEMM slicing only
  if (p.rope_cs) {   // fused q / k RMSNorm + RoPE epilogue (round 5: also in fp8 mode; gemm_fp8_qkn_ok)
    if (pl.sk > 1 || EPI != EPI_BIAS_GELU) {
      if (prof) prof_end(2, st);
      return fail("gemm_fp8: the q/k norm + RoPE epilogue needs an unsliced launch with the bias(+GELU) epilogue (gemm_fp8_qkn_ok)");
    }
    static bool attrq = false;
    const void* fn = (const void*)gemm8pp_kernel<EPI_BIAS_GELU, 2, true, false, true>;
    if (!attrq) {
      hipFuncAttributes fa;
      (void)hipFuncGetAttributes(&fa, fn);
      (void)hipGetLastError();
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL) != hipSuccess) {
        if (prof) prof_end(2, st);
        return fail("gemm_fp8: cannot raise dynamic LDS limit for the q/k-norm kernel");
      }
      attrq = true;
    }
    gemm8pp_kernel<EPI_BIAS_GELU, 2, true, false, true><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
  } else if (pl.sk > 1) {
    GemmParams ps = p;
    ps.ws = (float*)ws; ps.sk = pl.sk; ps.u_full = 0; ps.tail_r = pl.tail_r;
    gemm8pp_kernel<EPI_BIAS, 2, true, true><<<grid, 512, PP_LDS_TOTAL, st>>>(ps);
    tail_reduce_kernel<EPI, true><<<(unsigned)(p.batch * pl.tail_r * 32), 256, 0, st>>>(ps);
  } else {
    gemm8pp_kernel<EPI, 2, true><<<grid, 512, PP_LDS_TOTAL, st>>>(p);
  }
  if (prof) prof_end(2, st);
  return check_launch("gemm_fp8");
}

int gemm_fp8(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return 0;
  if (!a.a_scale || !a.w_scale) return fail("gemm_fp8: scale pointers required");
  if (a.w_bstride) return fail("gemm_fp8: per-batch weights (w_bstride) are a bf16-path feature");
  if (a.conv_cin > 0) return fail("gemm_fp8: no convolution mode");
  const bool al = ((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.C % 16 == 0) &&
                  ((uintptr_t)a.bias % 8 == 0) && ((uintptr_t)a.w_scale % 16 == 0);
  const int tm = (a.M + 255) / 256, tn = (a.N + 255) / 256;
  if (!(a.K > 0 && a.K % 256 == 0 && a.N % 8 == 0 && a.lda % 16 == 0 && a.ldw % 16 == 0 && a.a_bstride % 16 == 0 &&
        a.ldc % 8 == 0 && a.c_bstride % 8 == 0 && al && (int64_t)(a.batch - 1) * a.a_bstride + (int64_t)tm * 256 * a.lda < (1ll << 32) &&
        (int64_t)tn * 256 * a.ldw < (1ll << 32) && (int64_t)255 * a.lda < (1ll << 31) && (int64_t)255 * a.ldw < (1ll << 31)))
    return fail("gemm_fp8: shape/alignment not supported (K %% 256 == 0, N %% 8 == 0, 16-byte aligned rows)");
  if ((a.epilogue == EPI_BIAS_GATE_RES || a.epilogue == EPI_BIAS_RES) &&
      !(a.res && a.ldr % 8 == 0 && a.r_bstride % 8 == 0 && (uintptr_t)a.res % 16 == 0))
    return fail("gemm_fp8: residual pointer / alignment");
  if (a.epilogue == EPI_BIAS_GATE_RES && !(a.gate && a.gate_bstride % 4 == 0 && (uintptr_t)a.gate % 8 == 0))
    return fail("gemm_fp8: gate pointer / alignment");
  if (a.epilogue == EPI_BIAS_GELU && a.gelu_from_col % 256) return fail("gemm_fp8: gelu_from_col must be a multiple of 256");
  // the fused q / k norm + RoPE epilogue: every condition of gemm_fp8_qkn_ok is re-checked HERE -- a public caller
  // that sets rope_cs with unaligned column ranges, a GELU range overlapping q / k, row-split weights or null norm weights would
  // otherwise reach the kernel and fault on the device
  if (a.qkn_rope_cs) {
    if (!a.qkn_wq || !a.qkn_wk) return fail("gemm_fp8: the fused q/k norm + RoPE epilogue needs both norm weight vectors");
    if (!gemm_fp8_qkn_ok(a)) return fail("gemm_fp8: shape / column ranges not eligible for the fused q/k norm + RoPE epilogue (gemm_fp8_qkn_ok)");
  }
  const GemmParams p = make_params(a);
  switch (a.epilogue) {
    case EPI_BIAS: return launch_fp8<EPI_BIAS>(p, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_GELU: return launch_fp8<EPI_BIAS_GELU>(p, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_GATE_RES: return launch_fp8<EPI_BIAS_GATE_RES>(p, a.workspace, a.workspace_bytes, st);
    case EPI_BIAS_RES: return launch_fp8<EPI_BIAS_RES>(p, a.workspace, a.workspace_bytes, st);
  }
  return fail("gemm_fp8: unknown epilogue %d", a.epilogue);
}

}  // namespace tfx
