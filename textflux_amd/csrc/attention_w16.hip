// Joint attention on v_mfma_f32_16x16x32_bf16 (attention_waves = 40): the one-wave-per-SIMD schedule of attention_w4.hip rebuilt
// around the 16 x 16 MFMA.
//
// Why another kernel.  On random data every MFMA-dense kernel of this library runs at the board's power limit, and
// tools/ubench/mfma_power (profiles/r03_mfma_power.json) measures what the two bf16 shapes cost there: a register-only stream of
// v_mfma_f32_16x16x32_bf16 sustains 2.04-2.06 PFLOP/s at 2.19 GHz, the same stream of v_mfma_f32_32x32x16_bf16 1.77-1.82 at
// 1.92 GHz -- 10 % fewer joules per FLOP, 15 % more throughput.  The 16 x 16 shape also makes the softmax bookkeeping cheaper:
//   * the reference maximum is subtracted by the MFMA's C operand: the first MFMA of a score chain reads a 4-register tuple
//     holding -m_ref[q] instead of zeros (attention_w4 spends a ninth k-step per 32 x 32 block on that: +12.5 % score MFMAs);
//   * the row sums ride on a ones block of 16 rows instead of 32: +1 MFMA per 8 instead of +1 per 4 on the P.V side.
// Per 64-key tile and wave: 64 + 72 = 136 MFMAs of 16 cycles = 2176 matrix cycles, against 76 of 32 = 2432 in attention_w4.
//
// Layout.  Workgroup = 4 waves (one per SIMD, 512 registers each), 256 query rows of one (batch, head); a wave owns 64 rows =
// four 16-row q-blocks.  S^T = K Q^T: MFMA A = K fragment (16 keys x 32 d: lane l holds key l & 15, d = 32 s + 8 (l >> 4) ..),
// B = Q fragment (16 rows x 32 d), D[key][q]: lane l holds query l & 15, keys 4 (l >> 4) + e of the 16-key block -- a query's
// 16 scores of a block sit in the four lanes l, l ^ 16, l ^ 32, l ^ 48 (row maximum: in-lane max, v_permlane16_swap,
// v_permlane32_swap).  P feeds the P.V MFMA straight from the score registers: the k-slot order of a 32-key step is DEFINED as
// [keys 4g .. 4g+3 of block 2t, keys 4g .. 4g+3 of block 2t+1] (g = l >> 4), and the V^T fragments are read in that order with
// two ds_read_b64_tr_b16 (rows 32 t + 4 g + .., and + 16) -- no cross-lane movement of P.  O^T[d][q] += V^T P^T: lane holds query
// l & 15, d = 16 db + 4 g + e.
// LDS: ring of 3 tiles, K and V rows at a 288-byte pitch (conflict-free ds_read_b128 of 16 rows x 4 chunks, conflict-free
// transpose reads of 8 rows x 32 B), one barrier per tile; staging global -> registers -> LDS as a write / refill stream in
// step 1 of every tile, descriptors sized to the valid rows (rows past N read as zeros), as attention_w4.
// Pipeline: unit u = (tile, q-block); step u runs the VALU softmax of unit u in the shadow of the MFMA stream
// S(u + 1) [16] interleaved with the pending P(u - 1).V [18]; K fragments (the whole tile: 16) are reloaded for the next tile in
// step 2, right behind their last use, V fragments (16) in step 0.  One MFMA per scheduling region.
// Hazards: scores are read by inline-asm VALU instructions; W16_TOUCH (a compiler-visible read) in front makes hipcc pad the
// MFMA -> VALU wait states by construction (tests/test_isa_hazards.py checks the emitted ISA).
#include <type_traits>

#include "common.h"
#include "launch.h"

namespace tfx {

namespace {
constexpr int X_KV = 64, X_HD = 128;
constexpr int X_PITCH = 288;                   // row pitch of the K and V tiles
constexpr int X_TILE = X_KV * X_PITCH;         // 18432 B
constexpr int X_NBUF = 3;
constexpr int X_KBASE = X_NBUF * X_TILE;       // V tiles first, then K tiles
constexpr float X_THR = 4.0f;                  // lazy-reference threshold (log2 units)
typedef __attribute__((address_space(3))) s16x4 xlds_s16x4;
typedef __attribute__((ext_vector_type(8))) short xs16x8;
template <int V>
using XC = std::integral_constant<int, V>;
template <int I, int N, class F>
__device__ __forceinline__ void x_static_for(F&& f) {
  if constexpr (I < N) {
    f(XC<I>{});
    x_static_for<I + 1, N>(f);
  }
}
}  // namespace

constexpr int ATT_LDS_W16 = 2 * X_NBUF * X_TILE;   // 108 KiB

#define X_GAP() __builtin_amdgcn_sched_barrier(0)
// wait until every issued MFMA has written its result (prologue and output only)
#define X_DRAIN_MFMA()                                                                                                  \
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"      \
               "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
#ifndef W16_NO_TOUCH
#define W16_TOUCH(acc)                                                                 \
  do {                                                                                 \
    const int t_ = __builtin_amdgcn_readfirstlane(__float_as_int((acc)[0]));           \
    asm volatile("" ::"s"(t_));                                                        \
  } while (0)
#else
#define W16_TOUCH(acc) do { } while (0)
#endif

__device__ __forceinline__ float x_max7(float a, float b, float c, float d, float e, float f, float g) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7"
      : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g));
  return r;
}
__device__ __forceinline__ float x_max4(float a, float b, float c, float d) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}
__device__ __forceinline__ float x_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t x_cvt_pk(float lo, float hi) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__global__ __launch_bounds__(256, 1) void attn_w16_kernel(const bf16_t* Q, const bf16_t* __restrict__ Kp,
                                                          const bf16_t* __restrict__ Vp, bf16_t* O, int64_t ldq, int64_t ldk,
                                                          int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                                                          int64_t o_bs, int H, int N, int nqb, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, r16 = lane & 15;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int qblk = bid % nqb;
  bid /= nqb;
  const int h = bid % H;
  const int b = bid / H;
  const bf16_t* Qb = Q + b * q_bs + h * X_HD;
  const bf16_t* Kb = Kp + b * k_bs + h * X_HD;
  const bf16_t* Vb = Vp + b * v_bs + h * X_HD;
  bf16_t* Ob = O + b * o_bs + h * X_HD;

  // ---- Q fragments of the wave's four q-blocks (row l & 15, d = 32 s + 8 g ..), pre-scaled into the exp2 domain (one extra bf16
  // rounding of q); all sixteen requests go out before the first conversion
  bf16x8 qf[4][4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int row = qblk * 256 + wave * 64 + qb * 16 + r16;
    const int rc = row < N ? row : N - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)rc * ldq + s * 32 + g * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  auto convert_q = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int qb = 0; qb < 4; ++qb)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qb][s][e] = (__bf16)((float)qf[qb][s][e] * scale_log2e);
        asm volatile("" : "+a"(qf[qb][s]));   // home of the Q fragments: the AccVGPRs (srcB of the score MFMAs reads them there)
      }
  };

  const int nkv = (N + X_KV - 1) / X_KV;
  // ---- staging: thread t moves the 16-byte chunks (row t / 16 + 16 i, chunk t % 16), i = 0..3, of a tile's K and V
  u32x4 kreg[4], vreg[4];
  const int ldk2 = (int)ldk * 2, ldv2 = (int)ldv * 2;
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)((uint32_t)(N - 1) * (uint32_t)ldk2 + 256u), 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)((uint32_t)(N - 1) * (uint32_t)ldv2 + 256u), 0x00020000);
  auto load_piece = [&](int j, int ko, int vo, auto Ic, bool k_side) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    if (k_side)
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, ko, (j * X_KV + 16 * i) * ldk2, 0));
    else
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, (j * X_KV + 16 * i) * ldv2, 0));
  };
  auto stage_offsets = [&](int& ko, int& vo) __attribute__((always_inline)) {
    int te = tid;
    asm volatile("" : "+v"(te));     // rebuilt where needed: as loop invariants the offsets would pin two registers
    ko = (int)__umul24(te >> 4, ldk2) + (te & 15) * 16;
    vo = (int)__umul24(te >> 4, ldv2) + (te & 15) * 16;
  };
  auto load_tile = [&](int j) __attribute__((always_inline)) {
    int ko, vo;
    stage_offsets(ko, vo);
    load_piece(j, ko, vo, XC<0>{}, true); load_piece(j, ko, vo, XC<0>{}, false); load_piece(j, ko, vo, XC<1>{}, true); load_piece(j, ko, vo, XC<1>{}, false);
    load_piece(j, ko, vo, XC<2>{}, true); load_piece(j, ko, vo, XC<2>{}, false); load_piece(j, ko, vo, XC<3>{}, true); load_piece(j, ko, vo, XC<3>{}, false);
  };
  auto write_piece = [&](int buf, auto Ic, bool k_side) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    const int kr = tid >> 4, ch = tid & 15;
    if (k_side)
      *reinterpret_cast<u32x4*>(smem + X_KBASE + buf * X_TILE + (kr + 16 * i) * X_PITCH + ch * 16) = kreg[i];
    else
      *reinterpret_cast<u32x4*>(smem + buf * X_TILE + (kr + 16 * i) * X_PITCH + ch * 16) = vreg[i];
  };
  auto write_tile = [&](int buf) __attribute__((always_inline)) {
    write_piece(buf, XC<0>{}, true); write_piece(buf, XC<0>{}, false); write_piece(buf, XC<1>{}, true); write_piece(buf, XC<1>{}, false);
    write_piece(buf, XC<2>{}, true); write_piece(buf, XC<2>{}, false); write_piece(buf, XC<3>{}, true); write_piece(buf, XC<3>{}, false);
  };
  // ---- fragment read addresses: one per-lane base each for K and V, + immediates
  const char* rK = smem + X_KBASE + r16 * X_PITCH + g * 16;
  const char* rV = smem + (4 * g + (r16 >> 2)) * X_PITCH + (r16 & 3) * 8;
  // off: byte offset of the ring buffer inside the K / V region
  auto kread = [&](int off, int kb, int s) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(rK + off + kb * 16 * X_PITCH + s * 64);
  };
  auto vread = [&](int off, int t, int db) __attribute__((always_inline)) -> bf16x8 {
    const char* va = rV + off + t * 32 * X_PITCH + db * 32;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xlds_s16x4*)(va));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((xlds_s16x4*)(va + 16 * X_PITCH));
    return __builtin_bit_cast(bf16x8, (xs16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x4 o[4][8], ol[4], sc[2][4], mneg[4];
#pragma unroll
  for (int qb = 0; qb < 4; ++qb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ol[qb][r] = 0.f;
      mneg[qb][r] = 0.f;
#pragma unroll
      for (int db = 0; db < 8; ++db) o[qb][db][r] = 0.f;
    }
  bf16x8 vone, kf[4][4], vf[2][8];
  u32x4 pf[2][2];                // bf16 weights of the pending / the current unit, per 32-key step
#pragma unroll
  for (int e = 0; e < 8; ++e) vone[e] = (__bf16)1.0f;
  asm volatile("" : "+a"(vone));
  float m_ref[4] = {0.f, 0.f, 0.f, 0.f};   // lazy reference maximum per q-block row (exp2 domain); -m_ref sits in mneg[.]

  // ---- prologue: tiles 0 and 1 in LDS, tile 2 requested; K fragments of tile 0; S(tile 0, q-block 0)
  load_tile(0);
  u32x4 k1[4], v1[4];
  {
    int ko, vo;
    stage_offsets(ko, vo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      k1[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, ko, (X_KV + 16 * i) * ldk2, 0));
      v1[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, (X_KV + 16 * i) * ldv2, 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  convert_q();
  __builtin_amdgcn_sched_barrier(0);
  write_tile(0);
  load_tile(2);
  {
    const int kr = tid >> 4, ch = tid & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(smem + X_KBASE + X_TILE + (kr + 16 * i) * X_PITCH + ch * 16) = k1[i];
      *reinterpret_cast<u32x4*>(smem + X_TILE + (kr + 16 * i) * X_PITCH + ch * 16) = v1[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[kb][s] = kread(0, kb, s);
  X_GAP();
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      sc[0][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][s], qf[0][s], s == 0 ? mneg[0] : sc[0][kb], 0, 0, 0);
  X_DRAIN_MFMA();
  X_GAP();

  // One pipeline step = unit u = (tile, q-block QB).
  //   VALU : softmax of S(u) (sc[QB & 1]) -> P(u) (pf[QB & 1]);  rarely: move q-block QB's reference
  //   MFMA : S(u + 1) for q-block (QB + 1) & 3 (of the next tile when QB == 3) into sc[(QB + 1) & 1];  pending P(u - 1).V of
  //          q-block (QB + 3) & 3 (pf[(QB + 1) & 1], vf)
  //   QB == 2: last user of kf -> each K fragment is reloaded from ring offset KN (next tile) right behind its use
  //   QB == 0: last user of vf (previous tile's V) -> each V fragment is reloaded from ring offset VN (this tile)
  //   PV   : 0 on the very first step (nothing pending);  FIRST: the q-block's first unit pins the reference to the true maximum
  //   rag  : this tile reaches past N (last tile only): keys >= N are masked; kb_abs = its first key
  //   STG  : 1 + 4 * buf = move tile jst - 1 from the staging registers into ring buffer buf and request tile jst (QB == 1)
  auto step = [&](auto QBc, auto PVc, auto FIRSTc, auto KNc, auto VNc, auto STGc, int jst, bool rag, int kb_abs) __attribute__((always_inline)) {
    constexpr int QB = decltype(QBc)::value, NQ = (QB + 1) & 3, PQ = (QB + 3) & 3;
    constexpr bool PV = decltype(PVc)::value != 0, FIRST = decltype(FIRSTc)::value != 0;
    constexpr int KN = decltype(KNc)::value, VN = decltype(VNc)::value, STG = decltype(STGc)::value;
    f32x4(&cur)[4] = sc[QB & 1];
    f32x4(&nxt)[4] = sc[(QB + 1) & 1];
    u32x4(&pcur)[2] = pf[QB & 1];
    u32x4(&pprev)[2] = pf[(QB + 1) & 1];
    if (__builtin_expect(rag, 0)) {
      int kbase = kb_abs + 4 * g;
      asm volatile("" : "+v"(kbase));     // keep the index arithmetic inside the (last-tile-only) branch
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + 16 * kb + e >= N) cur[kb][e] = -INFINITY;
    }
    // MFMA i of the score stream (k-step i / 4 of key block i % 4: dependent MFMAs are 4 score MFMAs apart)
    auto S = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value, s = i / 4, kb = i % 4;
      if constexpr (s == 0) nxt[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][0], qf[NQ][0], mneg[NQ], 0, 0, 0);
      else nxt[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb][s], qf[NQ][s], nxt[kb], 0, 0, 0);
    };
    // MFMA i of the pending P.V (32-key step i / 9, block i % 9: 0..7 = d blocks, 8 = the ones block that sums the row)
    auto P = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value, t = i / 9, db = i % 9;
      if constexpr (PV) {
        if constexpr (db == 8) ol[PQ] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vone, __builtin_bit_cast(bf16x8, pprev[t]), ol[PQ], 0, 0, 0);
        else o[PQ][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][db], __builtin_bit_cast(bf16x8, pprev[t]), o[PQ][db], 0, 0, 0);
      }
    };
    // VALU stream of the exponentials, one instruction per call:  e0 e1 e2 e3 | c0 e4 e5 | c1 e6 e7 | ... | c5 e14 e15 | c6 c7.
    // Pack c_m reads e_2m, e_2m+1 and follows BOTH by at least two stream positions, so whatever hipcc does with a filler and
    // the MFMA of its own region, a transcendental result is never read by the very next instruction (it needs one instruction
    // in between, and the asm pack is invisible to hipcc's hazard pass; tools/check_mfma_hazard.py verifies the emitted ISA)
    auto F = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int k = decltype(Ic)::value;
      constexpr bool is_c = k >= 22 || (k >= 4 && (k - 4) % 3 == 0);
      if constexpr (is_c) {
        constexpr int c = k >= 22 ? k - 16 : (k - 4) / 3;     // pack c: scores 2c, 2c + 1 (flat index kb * 4 + e)
        pcur[c >> 2][c & 3] = x_cvt_pk(cur[(2 * c) >> 2][(2 * c) & 3], cur[(2 * c + 1) >> 2][(2 * c + 1) & 3]);
      } else {
        constexpr int e = k < 4 ? k : 4 + 2 * ((k - 4) / 3) + ((k - 4) % 3 - 1);
        cur[e >> 2][e & 3] = __builtin_amdgcn_exp2f(cur[e >> 2][e & 3]);
      }
    };
    int stg_ko = 0, stg_vo = 0;
    if constexpr ((STG & 3) == 1) stage_offsets(stg_ko, stg_vo);
    auto G = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int n = decltype(Ic)::value;            // 0..15:  W0 W1 | L0 W2 L1 W3 ... L5 W7 | L6 L7  (piece = K / V chunk, K first)
      if constexpr ((STG & 3) == 1) {
        constexpr bool is_w = n == 0 || n == 1 || (n < 15 && (n & 1));
        constexpr int gg = n < 2 ? n : is_w ? (n + 1) / 2 : n == 15 ? 7 : n / 2 - 1;
        if constexpr (is_w) write_piece(STG >> 2, XC<gg / 2>{}, (gg & 1) == 0);
        else load_piece(jst, stg_ko, stg_vo, XC<gg / 2>{}, (gg & 1) == 0);
      }
    };
    // ---- the 34 MFMAs of the step, one per scheduling region: even regions 0..30 = S 0..15, odd regions 1..31 = P 0..15,
    // regions 32, 33 = P 16, 17.  Fillers: regions 0..4 the row maximum, the (rare, out-of-line) reference move behind region
    // 5, regions 6..29 the 24 exponentials / packs, staging in regions 8..23 (step 1), K reloads two regions behind the S MFMA
    // that last read the fragment (step 2), V reloads two regions behind the P MFMA that last read it (step 0)
    float mx = 0.f;
    x_static_for<0, 34>([&](auto Rc) __attribute__((always_inline)) {
      constexpr int r = decltype(Rc)::value;
      if constexpr (r < 32 && (r & 1) == 0) S(XC<r / 2>{});
      else if constexpr (r < 32) P(XC<r / 2>{});
      else P(XC<r - 16>{});
      if constexpr (r == 0) {
        W16_TOUCH(cur[3]);   // written by the LAST MFMA of the score stream: every other score register is at least as far from its writer
        mx = x_max7(cur[0][0], cur[0][1], cur[0][2], cur[0][3], cur[1][0], cur[1][1], cur[1][2]);
      }
      if constexpr (r == 1) mx = x_max7(mx, cur[1][3], cur[2][0], cur[2][1], cur[2][2], cur[2][3], cur[3][0]);
      if constexpr (r == 2) mx = x_max4(mx, cur[3][1], cur[3][2], cur[3][3]);
      if constexpr (r == 3) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = x_max(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
      if constexpr (r == 4) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = x_max(__uint_as_float(sw[0]), __uint_as_float(sw[1]));   // all 64 keys of the row
      }
      if constexpr (r == 5) {
        X_GAP();
        // out of line: with one wave per SIMD nothing hides the instruction-fetch bubble of a TAKEN branch
        if (FIRST || __builtin_expect(!__all(mx <= X_THR), 0)) {
          // move the reference: everything q-block QB accumulated against the old one is rescaled exactly once (no MFMA on
          // o[QB] / ol[QB] is in this step's stream), the scores of this unit are shifted before they are exponentiated
          const float m_new = m_ref[QB] + (FIRST ? mx : fmaxf(mx, 0.f));
          const float d = m_new - m_ref[QB];
          m_ref[QB] = m_new;
#pragma unroll
          for (int e = 0; e < 4; ++e) mneg[QB][e] = -m_new;
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[kb][e] -= d;
          if constexpr (!FIRST) {
            const float f = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ol[QB][e] *= f;
#pragma unroll
              for (int db = 0; db < 8; ++db) o[QB][db][e] *= f;
            }
          }
        }
      }
      if constexpr (r >= 6 && r < 30) F(XC<r - 6>{});
      if constexpr (r >= 8 && r < 24) G(XC<r - 8>{});
      // K reload (step 2): S MFMA i = region 2 i read kf[i % 4][i / 4] for the last time; reload it two regions later
      if constexpr (QB == 2 && r >= 2 && r <= 32 && (r & 1) == 0) {
        constexpr int i = r / 2 - 1;
        kf[i % 4][i / 4] = kread(KN, i % 4, i / 4);
      }
      // V reload (step 0): P MFMA i (0..17) sits in region 2 i + 1 (i < 16) or 16 + i; fragment (t, db) = (i / 9, i % 9), db < 8
      if constexpr (QB == 0) {
        constexpr int i = r < 32 ? ((r & 1) ? (r - 3) / 2 : -1) : -1;    // fragment whose reader ran two regions ago
        if constexpr (i >= 0 && i < 16 && (i % 9) < 8) vf[i / 9][i % 9] = vread(VN, i / 9, i % 9);
        if constexpr (r == 33) {                                           // readers in regions 31 (i = 15), 32 (16), 33 (17 = ones block)
          vf[1][6] = vread(VN, 1, 6);
          vf[1][7] = vread(VN, 1, 7);
        }
      }
      X_GAP();
    });
  };

  // one 64-key tile j out of ring buffer B (compile-time: every fragment address is a per-lane base plus an immediate)
  auto tile = [&](int j, auto Bc, auto FIRSTc) __attribute__((always_inline)) {
    constexpr int B = decltype(Bc)::value, NB = (B + 1) % 3, WB = (B + 2) % 3;
    constexpr int FIRST = decltype(FIRSTc)::value;
    const bool rag = (j == nkv - 1) && (N & (X_KV - 1));
    // q0: S(j, q1);  pending (tile j - 1, q3);  vf <- V(j)
    step(XC<0>{}, XC<!FIRST>{}, XC<FIRST>{}, XC<0>{}, XC<B * X_TILE>{}, XC<0>{}, 0, rag, j * X_KV);
    // q1: S(j, q2);  pending (j, q0);  + staging: tile j + 2 (requested one tile ago) goes into the buffer tile j - 1 left before the
    // last barrier, each register refilled with its piece of tile j + 3 right behind its write
    step(XC<1>{}, XC<1>{}, XC<FIRST>{}, XC<0>{}, XC<0>{}, XC<1 + 4 * WB>{}, j + 3, rag, j * X_KV);
    // q2: S(j, q3);  pending (j, q1);  kf <- K(j + 1)
    step(XC<2>{}, XC<1>{}, XC<FIRST>{}, XC<NB * X_TILE>{}, XC<0>{}, XC<0>{}, 0, rag, j * X_KV);
    // q3: S(j + 1, q0);  pending (j, q2)
    step(XC<3>{}, XC<1>{}, XC<FIRST>{}, XC<0>{}, XC<0>{}, XC<0>{}, 0, rag, j * X_KV);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  tile(0, XC<0>{}, XC<1>{});
  for (int j = 1; j < nkv; j += 3) {
    tile(j, XC<1>{}, XC<0>{});
    if (j + 1 < nkv) tile(j + 1, XC<2>{}, XC<0>{});
    if (j + 2 < nkv) tile(j + 2, XC<0>{}, XC<0>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests past the end of the sequence (zeros) still write their registers
  // ---- drain: the pending P.V of the very last unit (last tile, q3); its V fragments are in registers.  (The S stream of the
  // last step computed scores of a tile that does not exist: never read.)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int db = 0; db < 8; ++db)
      o[3][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][db], __builtin_bit_cast(bf16x8, pf[1][t]), o[3][db], 0, 0, 0);
    ol[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vone, __builtin_bit_cast(bf16x8, pf[1][t]), ol[3], 0, 0, 0);
  }
  X_DRAIN_MFMA();

  // ---- finish: every row of the ones block holds the full row sum.  The normalised bf16 rows go through a wave-private LDS
  // tile (the ring is free: every wave's last fragment read lies before the last barrier) and leave as whole 256-byte rows,
  // 16 lanes x 16 bytes each.  Lane holds, per (qb, db), d = 16 db + 4 g .. + 4 of query row qb * 16 + (l & 15).
  constexpr int OROW = 272;
  char* ot = smem + wave * (64 * OROW);
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const float inv = 1.0f / ol[qb][0];
    char* orow = ot + (qb * 16 + r16) * OROW + 8 * g;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      u32x2 w;
      w[0] = pack_bf2(o[qb][db][0] * inv, o[qb][db][1] * inv);
      w[1] = pack_bf2(o[qb][db][2] * inv, o[qb][db][3] * inv);
      *reinterpret_cast<u32x2*>(orow + db * 32) = w;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int row0 = qblk * 256 + wave * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i, row = c >> 4, ch = c & 15;
    const u32x4 v = *reinterpret_cast<const u32x4*>(ot + row * OROW + ch * 16);
    if (row0 + row < N) *reinterpret_cast<u32x4*>(Ob + (int64_t)(row0 + row) * ldo + ch * 8) = v;
  }
}

int joint_attention_w16(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void*)attn_w16_kernel) != hipSuccess) return fail("attention: no attn_w16_kernel in this build");
    (void)hipGetLastError();
    if (fa.localSizeBytes != 0)
      return fail("attention: attn_w16_kernel spills %zu bytes per lane -- attention_w16.hip must be compiled with "
                  "-mllvm -amdgpu-mfma-vgpr-form", (size_t)fa.localSizeBytes);
    if (hipFuncSetAttribute((const void*)attn_w16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_W16) != hipSuccess)
      return fail("attention: cannot raise dynamic LDS limit to %d bytes", ATT_LDS_W16);
    attr_set = true;
  }
  const int nqb = (a.N + 255) / 256;
  const unsigned grid = (unsigned)(a.B * a.H * nqb);
  attn_w16_kernel<<<grid, 256, ATT_LDS_W16, st>>>((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, (bf16_t*)a.o, a.ldq,
                                                  a.ldk, a.ldv, a.ldo, a.q_bstride, a.k_bstride, a.v_bstride, a.o_bstride, a.H,
                                                  a.N, nqb, a.scale * 1.4426950408889634f);
  return 0;
}

}  // namespace tfx
