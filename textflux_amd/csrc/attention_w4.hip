// One-wave-per-SIMD joint attention (attention_waves = 30, the default): same math and matrix-pipe softmax bookkeeping as
// attn_mx_kernel (attention.hip), rebuilt around the SIMD's ISSUE budget.
//
// attn_mx_kernel gives every wave 32 query rows; per 64-key tile a wave issues 38 MFMAs (304 issue cycles), 32 v_exp_f32
// (8 cycles each), ~60 other VALU instructions and 48 LDS fragment reads -- about as many issue cycles as the 1216
// matrix-pipe cycles its MFMAs occupy, and with two waves per SIMD (served oldest-first, not interleaved) the two streams
// add up instead of overlapping: the pipe is 53-57 % busy whatever the schedule.  What does overlap is work issued by the
// SAME wave behind its own MFMA (tools/ubench/mfma_fill: one wave hides ~24 issue cycles behind each 32-cycle MFMA: four
// simple VALU instructions, or two v_exp_f32 and one more, or two LDS reads and two VALU).  The K / V fragment reads and
// the staging traffic are per WAVE, not per row, so here
//   * a workgroup is 4 waves (one per SIMD, the whole 512-register file each), a wave owns 64 query rows = two 32-row
//     q-blocks: every K / V fragment read from LDS feeds two MFMAs, staging per row halves;
//   * the tile loop is software-pipelined at the granularity of a UNIT = (32-key block, q-block): in step u the MFMA stream
//     is S(u+1) = K Q^T - m_ref (1 + 8 MFMAs, a dependent chain) interleaved with O^T += V^T P(u-1)^T, l += 1^T P(u-1)^T
//     (2 x (4 + 1) MFMAs, independent accumulators), and the VALU stream in their shadow is the softmax of unit u
//     (row maximum -> [rarely] move the lazy reference -> 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32).  Consecutive units
//     alternate between the two q-blocks, so the unit whose reference may move (u) never has MFMAs in flight on its own
//     accumulators: moving the reference rescales O / l of q-block u & 1 only;
//   * units run (kb0,q0) (kb0,q1) (kb1,q0) (kb1,q1): the K fragments of a key block serve two consecutive S chains and
//     the V fragments two consecutive P.V groups; each fragment register is reloaded shortly after its last use, one whole
//     step (~600 cycles) before its next, so no LDS latency is ever waited for and no fragment is double-buffered;
//   * ONE MFMA per scheduling region (sched_barrier), each followed by <= ~24 issue cycles of other work: an MFMA that finds
//     the pipe busy blocks its wave's issue until the pipe takes it, so work placed behind two adjacent MFMAs is not hidden
//     by the first.  Common path = fall-through (the reference move and the ragged mask are out of line): with one wave
//     per SIMD nothing hides the instruction-fetch bubble of a taken branch.
// Register files: the MFMAs are hipcc builtins and this file is compiled with -mllvm -amdgpu-mfma-vgpr-form (Makefile): the
// scores must come out in ArchVGPRs (the VALU reads them; the AGPR form costs a v_accvgpr_read per score), hipcc then keeps
// C / D of every MFMA in ArchVGPRs (O + l = 160, scores 32, weights 16) and most K / V fragments, the staging registers and
// -- pinned by an empty asm -- the 64 Q registers in the AccVGPRs, which srcA / srcB read directly (256 + ~165 registers).
// A hazard hipcc cannot see: the row maxima and the bf16 packs are inline asm (so that they stay where the schedule puts
// them); an MFMA result needs ~11 issued instructions before a VALU read and the hazard pass does not know those asm
// statements are VALU.  The score chain therefore runs one MFMA AHEAD of the P.V group: its last MFMA is followed by three
// MFMAs of its own step and the first of the next before the scores are read (reading earlier returned stale rows,
// sporadically: tests/test_kernels_gpu.py::test_attention_default_kernel_is_deterministic_and_nan_free); the only read
// right behind a chain (prologue) and the output sit behind an explicit drain.  (An MFMA's A / B operands, on the other
// hand, are safe as soon as it has issued: tools/ubench/mfma_war.)
// LDS: ring of 3 tiles (K rows padded to 272 B -> conflict-free b128 reads from ONE per-lane base register + immediates;
// V rows 256 B with their 64-byte segments XOR-swizzled for the transpose reads; PMC: 2.6 M conflict cycles in 199 M LDS-array
// cycles = 1.3 %, profiles/r03_attention_pmc.json), one barrier per
// 64-key tile.  Staging global -> registers -> LDS runs as a stream in the MFMA gaps of step 1 of every tile: piece g of tile
// j + 2 goes to LDS and its register is refilled with the same piece of tile j + 3 right behind, four steps (> 1 us) before
// it is needed -- with one wave per SIMD nothing else runs while a wave waits for memory.  The buffer descriptors are sized
// to the N valid rows, so the rows of a ragged last tile and whole tiles requested past the end read as zeros without any
// clamping code (their scores are masked / their weights are zero).
// Measured (MI355X, B = 8, H = 24, N = 4608, random data): 1.12-1.14 PFLOP/s vs 1.04-1.07 for attn_mx_kernel on the same
// box, 480 vs 495 ms per 57-block DiT forward; on zero data (no power cap) 1.44 PFLOP/s.  Both are power-limited on random
// data: 1.9 GHz at 1300 W.
#include <mutex>
#include <type_traits>

#include "common.h"
#include "launch.h"

namespace tfx {

namespace {
constexpr int W4_KV = 64, W4_HD = 128;
constexpr int W4_VT = W4_KV * 256;             // V tile bytes
constexpr int W4_KROW = 272;                   // K row pitch
constexpr int W4_KT = W4_KV * W4_KROW;         // K tile bytes
constexpr int W4_NBUF = 3;
constexpr int W4_KBASE = W4_NBUF * W4_VT;      // V tiles first, then K tiles
constexpr int W4_PROW = 132;                   // floats per row of a partial (size only: 128 d + row sum + reference maximum + pad; 256 rows per slot)
// Layout of one partial slot (256 x 132 floats; round 6 -- until then row-major, which made every store instruction of the producing kernel
// touch 64 different lines): the un-normalised O in the PRODUCER's register order -- 16-byte element ((wq, db, qd), lane), wq = wave * 2 +
// q-block, lane = hi * 32 + row-in-block, holds columns db * 32 + qd * 8 + hi * 4 .. + 4 of row wq * 32 + (lane & 31): one store instruction
// = 1 KiB contiguous -- followed by (row sum, reference maximum) pairs per row.
__device__ __forceinline__ int w4_part_off(int r, int col) {   // float offset of columns col .. col + 3 (col % 4 == 0) of row r (0 .. 255)
  return ((((r >> 5) * 4 + (col >> 5)) * 4 + ((col >> 3) & 3)) * 64 + ((col >> 2) & 1) * 32 + (r & 31)) * 4;
}
constexpr int W4_PART_LM = 256 * 128;          // float offset of the (l, m) pairs
constexpr float W4_THR = 4.0f;                 // lazy-reference threshold (log2 units), as attn_mx_kernel (MODE 0 / 1)
constexpr float W4_BIG = 64.0f;                // MODE 2: a row's reference stays 0 while its scores stay inside +-W4_BIG
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
template <int V>
using IC = std::integral_constant<int, V>;
}  // namespace

constexpr int ATT_LDS_W4 = W4_NBUF * (W4_VT + W4_KT);   // 99 KiB

#define W4_GAP() __builtin_amdgcn_sched_barrier(0)
#ifndef W4_ABL
#define W4_ABL 0   // timing ablations (wrong results): 1 no exp / pack, 2 no fragment reloads, 4 no staging, 8 no row maximum / branch, 16 no barrier, 64 no staging writes (requests kept), 128 no staging requests (writes kept), 256 requests as LDS-DMA into a scratch region (use with 64), 512 V fragments by one ds_read_b128 (a transposed V tile), 1024 the Q-side RMSNorm + RoPE in the Q prologue (what moving it out of the projection GEMM would cost here)
#endif
// wait until every issued MFMA has written its result (there is no counter for the matrix pipe): 24 x 16 idle issue slots,
// used twice per workgroup (before the first softmax, before the output)
#define W4_DRAIN_MFMA()                                                                                                 \
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"      \
               "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"      \
               "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

// The scores are read by inline-asm VALU instructions (below), which hipcc's hazard recogniser does not see as VALU: it would
// not insert the wait states an MFMA result needs before a VALU read (8-pass MFMA: 11).  W4_TOUCH(acc) is a COMPILER-VISIBLE
// VALU read of the accumulator tuple (v_readfirstlane_b32 of one element; the hazard is tracked per destination tuple) placed
// in program order before the first asm read: the recogniser pads in front of IT, whatever a future compiler or a schedule
// change does to the distance, and every later read is at least as far from the MFMA.  tools/check_mfma_hazard.py verifies
// the emitted ISA (tests/test_isa_hazards.py; -DW4_NO_TOUCH removes the touch and shortens the distance: the self-test).
#ifndef W4_NO_TOUCH
#define W4_TOUCH(acc)                                                                  \
  do {                                                                                 \
    const int t_ = __builtin_amdgcn_readfirstlane(__float_as_int((acc)[0]));           \
    asm volatile("" ::"s"(t_));                                                        \
  } while (0)
#else
#define W4_TOUCH(acc) do { } while (0)
#endif

__device__ __forceinline__ void w4_mfma_s0(f32x16& d, const bf16x8& k, const bf16x8& q) {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q, z, 0, 0, 0);
}
__device__ __forceinline__ void w4_mfma_s(f32x16& d, const bf16x8& k, const bf16x8& q) {
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k, q, d, 0, 0, 0);
}
__device__ __forceinline__ void w4_mfma_o(f32x16& d, const bf16x8& v, const u32x4& p) {
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, __builtin_bit_cast(bf16x8, p), d, 0, 0, 0);
}
__device__ __forceinline__ void w4_mfma_l(f32x16& d, const bf16x8& ones, const u32x4& p) {
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, __builtin_bit_cast(bf16x8, p), d, 0, 0, 0);
}
// single-instruction VALU helpers: hipcc would canonicalise MFMA outputs (v_max x, x) in front of fmaxf and sink the
// exponentials / conversions to their first use in the NEXT step -- these stay where the schedule puts them
__device__ __forceinline__ float w4_max7(float a, float b, float c, float d, float e, float f, float g) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7"
      : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g));
  return r;
}
__device__ __forceinline__ float w4_max4(float a, float b, float c, float d) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max_f32 %0, %0, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d));
  return r;
}
__device__ __forceinline__ float w4_max(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// l += p.lo + p.hi (the two bf16 weights of a packed register, times 1.0 each), fp32 accumulate
__device__ __forceinline__ void w4_dot2c(float& l, uint32_t p) {
  asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(l) : "v"(p));
}
__device__ __forceinline__ uint32_t w4_cvt_pk(float lo, float hi) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// MODE 0: the round-2 bookkeeping (row sums and the reference offset on the matrix pipe: 76 MFMAs per 64-key tile, 64 of them
// Q.K^T / P.V).  MODE 1: the row sums leave the matrix pipe -- with the swapped Q.K^T a query row is lane-local, so l += sum p is
// eight v_dot2c_f32_bf16 (p0 * 1 + p1 * 1 + l: exactly the bf16 weights P.V multiplies) per unit, issued one step later in the
// MFMA shadow of the step's first regions; per-lane partial sums (each lane holds 16 of a unit's 32 keys), the two halves of a
// row meet once, at the end: 68 MFMAs per tile.  MODE 2: as 1, and the reference offset (one MFMA per unit whose only job is
// to subtract m_ref) is issued only while some row of the wave HAS a non-zero reference: the reference stays 0 as long as a row's
// scores stay inside +-W4_BIG (exp2 domain; softmax does not depend on the reference, and fp32 / bf16 share an exponent range
// that holds 2^+-64 weights and their sums over 2^13 keys with room to spare), is pinned to the row maximum by the first tile
// only when that lies outside, and moves later only when a row's maximum exceeds it by more than W4_BIG: 64 MFMAs per tile on
// ordinary data, the MODE-1 stream otherwise (a wave-uniform, not-taken branch in front of the chain).  MODE 3: MODE 0's stream
// (row sums on the matrix pipe) with MODE 2's lazy reference offset: 72 MFMAs per tile, no VALU instruction more than MODE 0.
// MODE 4: no reference at all (no row maximum, no branch, no offset MFMA) -- only launched when the caller's score bound is ADMISSIBLE
// (attention.hip::attn_bound_admissible, round 5): with |scale q.k| <= score_bound the exp2-domain scores lie in +-b, b = bound log2 e,
// every weight in [2^-b, 2^b], a row sum <= N 2^b and an un-normalised output <= N 2^b max|v|; nothing leaves the exponent range fp32 and
// bf16 share while b + log2 N + 24 <= 126, which GRANTS |v| <= 2^24 (a property of the caller's V that the library cannot verify from the
// norm weights; the guarded modes assume the same kind of thing about N max|v|).  N = 4608: b <= 89.8, score_bound <= 62.2 -- more than
// W4_BIG, which only governs the lazy reference of modes 2 / 3.  The DiT derives the bound from the q / k RMSNorm weights: after the
// norm |q| <= sqrt(128) max|w_q|, RoPE preserves the norm, so |q . k| scale <= 128 max|w_q| max|w_k| 128^-1/2.
#ifdef TFX_BENCH
// bench library only (tools/attn_item_timers.py): s_memtime sums of wave 0 per workgroup {prologue, tile loop, output, items}
__device__ unsigned long long* g_w4_timers = nullptr;
#define W4_STAMP(i) do { if (wave == 0) w4_stamp[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_STAMP(i) ((void)0)
#endif
// Stream-K tail of the persistent form (round 6, tfx_set_option attention_streamk).  Every batch sample has its own group of C / B CUs.  The
// sample's items run as whole items, round by round (CU c takes items c, c + group, ...: the CUs of a group work on neighbouring items at any
// time, so a head's K / V stay L2-resident among them), as long as a round is full; the R < group items of the last, partly filled round are
// dealt as (item, 64-key tile) units instead: their R nkv units, item-major, are cut into `group` contiguous ranges of `share` units, one per
// CU, so that every CU ends with the same number of key tiles (13 items + half an item each at P1024 batch 8 instead of 14 items on half of
// the CUs and 13 on the others).  Boundary c of the tail = c * share, snapped to the item boundary when it would leave a piece shorter than
// W4_SK_MIN tiles.  A tail item cut by a boundary leaves un-normalised partials (the tail split's format) that attn_w4_merge_sk_kernel adds up.
// Every sample is dealt by the same rule: identical samples of a batch produce identical bits; a sample's bits do depend on the batch
// size (as with the K-sliced GEMMs).  (First version of the round: ALL units of a sample dealt contiguously -- 15 % SLOWER at P1024 batch 8:
// the CUs of an XCD then sit 13 items apart and nothing they load is shared in L2.)
constexpr int W4_SK_MIN = 4;
__host__ __device__ __forceinline__ int w4_sk_bound(int c, int group, int share, int nkv, int total) {
  if (c >= group) return total;
  int p = c * share;
  if (p >= total) return total;
  const int r = p % nkv;
  if (r < W4_SK_MIN) p -= r;
  else if (r > nkv - W4_SK_MIN) p += nkv - r;
  return p < total ? p : total;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void attn_w4_kernel(const bf16_t* Q, const bf16_t* __restrict__ Kp,
                                                         const bf16_t* __restrict__ Vp, bf16_t* O, int64_t ldq, int64_t ldk,
                                                         int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                                                         int64_t o_bs, int H, int N, int nqb, float scale_log2e, int nfull,
                                                         int nparts, int nsplit, int xsplit, float* part, int T_items, int sk_group,
                                                         int sk_share) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool LV = MODE == 1 || MODE == 2, LZ = MODE >= 2, NOREF = MODE == 4, FT = LV || NOREF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  // Tail split (nsplit > 1, joint_attention_w4): the LAST xsplit (head, q-tile) pairs of every batch sample are each cut into
  // nsplit KEY ranges and dispatched after the nfull ordinary workgroups -- nfull a whole number of rounds of the chip, so the last
  // round is made of short workgroups instead of being partly empty; such a workgroup leaves its un-normalised O, row sums and
  // reference maxima in `part`, attn_w4_merge_kernel finishes.  Which tiles are split depends on (head, q-tile) only, never on
  // the batch index: identical samples of a batch still produce identical bits; and the parts of one (batch, head) are neighbours
  // in the dispatch order, so its K / V stay L2-resident among them.  (Measured alternatives: the last q-tile of EVERY head --
  // re-streams all K / V from HBM at the end, no gain; whole heads -- nfull is no longer a multiple of the CU count, some CUs run
  // three halves in a row, 1.4 % slower than no split.)
  // PERSISTENT form (T_items > 0; round 4): one workgroup per CU walks its XCD's contiguous range of the T_items (b, h, q-tile)
  // items.  A fresh workgroup spends 12 k cycles (plus its dispatch) in front of its first tile -- two dependent memory round trips
  // for Q and the first K / V tiles; here the next item's Q rows and its first K / V tile are requested behind the current item's
  // tile loop and fly while its output is normalised, staged and stored: 6 k cycles (tools/attn_item_timers.py).
  int kpart = -1, ptile = 0, qblk, h, b;
  int item = 0, item_step = 0, item_end = 0;
  // stream-K: this CU's range [sk_lo, sk_hi) of its sample's TAIL units, the sample's first item, the CU's two partial slots, its unit list
  int sk_lo = 0, sk_hi = 0, sk_base = 0, sk_slot0 = 0, sk_u = 0, sk_nunits = 0, sk_nwhole = 0, sk_full = 0, sk_cu = 0, sk_tfirst = 0;
  const int nkv_all = (N + W4_KV - 1) / W4_KV;
  // unit u of this CU: (item, first tile, end tile) -- whole items first (stride = the group), then the pieces of its tail range
  auto sk_unit = [&](int u, int& it, int& ts, int& te) __attribute__((always_inline)) {
    if (u < sk_nwhole) {
      it = sk_base + sk_cu + u * sk_group; ts = 0; te = nkv_all;
    } else {
      const int ti = sk_tfirst + (u - sk_nwhole);
      it = sk_base + sk_full + ti; ts = max(sk_lo - ti * nkv_all, 0); te = min(sk_hi - ti * nkv_all, nkv_all);
    }
  };
  int ts = 0, te = nkv_all;      // this pass's key tiles [ts, te) of its item (stream-K; everything otherwise)
  if (sk_share > 0) {            // T_items = items of ONE sample, nfull = number of samples
    const int L = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);            // CUs of one XCD are neighbours in a sample's group
    const int sample = L / sk_group;
    sk_cu = L - sample * sk_group;
    if (sample >= nfull) return;
    sk_full = (T_items / sk_group) * sk_group;
    sk_nwhole = sk_full / sk_group;
    const int total = (T_items - sk_full) * nkv_all;
    sk_lo = w4_sk_bound(sk_cu, sk_group, sk_share, nkv_all, total);
    sk_hi = w4_sk_bound(sk_cu + 1, sk_group, sk_share, nkv_all, total);
    sk_tfirst = sk_lo / nkv_all;
    sk_nunits = sk_nwhole + (sk_hi > sk_lo ? (sk_hi - 1) / nkv_all - sk_tfirst + 1 : 0);
    if (sk_nunits == 0) return;
    sk_base = sample * T_items;
    sk_slot0 = 2 * L;
    sk_unit(0, item, ts, te);
    qblk = item % nqb;
    h = (item / nqb) % H;
    b = item / (nqb * H);
  } else if (T_items > 0) {
    const int nwg = gridDim.x, q = T_items >> 3, r = T_items & 7, xcd = bid & 7, slot = bid >> 3;
    const int xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, xc = q + (xcd < r ? 1 : 0);
    if (slot >= xc) return;
    item = xs + slot;
    item_step = nwg >> 3;
    item_end = xs + xc;
    qblk = item % nqb;
    h = (item / nqb) % H;
    b = item / (nqb * H);
  } else if (nsplit > 1) {
    const int fpad = (nfull + 7) & ~7, pf = H * nqb - xsplit;      // pf: unsplit pairs per sample
    int pair;
    if (bid < fpad) {
      const int q = nfull >> 3, r = nfull & 7, xcd = bid & 7, k = bid >> 3;
      if (k >= q + (xcd < r ? 1 : 0)) return;
      const int fid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
      b = fid / pf;
      pair = fid % pf;
    } else {
      const int pi = bid - fpad;
      if (pi >= nparts) return;
      kpart = pi % nsplit;
      ptile = pi / nsplit;
      b = ptile / xsplit;
      pair = pf + ptile % xsplit;
    }
    h = pair / nqb;
    qblk = pair % nqb;
  } else {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    qblk = bid % nqb;
    bid /= nqb;
    h = bid % H;
    b = bid / H;
  }
  bf16x8 qn[2][8];               // persistent form: the NEXT item's Q rows (raw), requested behind this item's tile loop
  bool have_pref = false;        // ... and whether qn / kreg / vreg hold this item's Q rows / first K, V tile already
  u32x4 kreg[4], vreg[4];        // staging registers: thread t moves the 16-byte chunks (row t / 16 + 16 i, chunk t % 16), i = 0..3, of a tile's K and V
#ifdef TFX_BENCH
  unsigned long long w4_stamp[4] = {0, 0, 0, 0}, w4_sum[4] = {0, 0, 0, 0};
#endif
  for (;;) {                     // one pass per item (exactly one in the non-persistent form)
  W4_STAMP(0);
  bool has_next = item + item_step < item_end;            // (false in the non-persistent form: all three are 0)
  int item2 = item + item_step, ts2 = 0;
  if (sk_share > 0) {
    int te2;
    has_next = sk_u + 1 < sk_nunits;
    if (has_next) sk_unit(sk_u + 1, item2, ts2, te2);
  }
  const int qblk2 = item2 % nqb, h2 = (item2 / nqb) % H, b2 = item2 / (nqb * H);
  const bf16_t* Qb = Q + b * q_bs + h * W4_HD;
  const bf16_t* Kb = Kp + b * k_bs + h * W4_HD;   // (advanced to the first key of a split range below)
  const bf16_t* Vb = Vp + b * v_bs + h * W4_HD;
  bf16_t* Ob = O + b * o_bs + h * W4_HD;
  // keys of this workgroup: all N, or (tail split) the 64-key tiles [kpart, kpart + 1) * nkv / nsplit
  int Nk = N;
  if (kpart >= 0) {
    const int nkv_all = (N + W4_KV - 1) / W4_KV;
    const int t0 = kpart * nkv_all / nsplit, t1 = (kpart + 1) * nkv_all / nsplit;
    Nk = min(N, t1 * W4_KV) - t0 * W4_KV;
    Kb += (int64_t)t0 * W4_KV * ldk;
    Vb += (int64_t)t0 * W4_KV * ldv;
  }
  int pslot = -1;                // stream-K: >= 0 when this pass covers only a part of its item's keys (its partial slot)
  if (sk_share > 0) {
    if (ts > 0) pslot = sk_slot0;
    else if (te < nkv_all) pslot = sk_slot0 + 1;
    Nk = min(N, te * W4_KV) - ts * W4_KV;
    Kb += (int64_t)ts * W4_KV * ldk;
    Vb += (int64_t)ts * W4_KV * ldv;
  }

  // ---- Q fragments of the wave's two q-blocks, pre-scaled into the exp2 domain (one extra bf16 rounding of q).  All sixteen
  // requests go out before the first conversion: with one wave per SIMD a request-wait-convert loop would pay sixteen memory
  // round trips per workgroup
  int qrow[2];
  bf16x8 qf[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = qblk * 256 + wave * 64 + qb * 32 + l31;
    const int rc = qrow[qb] < N ? qrow[qb] : N - 1;
    if (have_pref) {
#pragma unroll
      for (int s = 0; s < 8; ++s) qf[qb][s] = qn[qb][s];
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)rc * ldq + s * 16 + hi * 8);
    }
  }
  // the next item's Q rows -> qn (requested behind the tile loop)
  auto load_qn = [&]() __attribute__((always_inline)) {
    const bf16_t* Qb2 = Q + b2 * q_bs + h2 * W4_HD;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int r2 = qblk2 * 256 + wave * 64 + qb * 32 + l31;
      const int rc = r2 < N ? r2 : N - 1;
#pragma unroll
      for (int s = 0; s < 8; ++s) qn[qb][s] = *reinterpret_cast<const bf16x8*>(Qb2 + (int64_t)rc * ldq + s * 16 + hi * 8);
    }
  };
  __builtin_amdgcn_sched_barrier(0);
  auto convert_q = [&]() __attribute__((always_inline)) {
    if constexpr (W4_ABL & 1024) {
      // timing ablation (round 6): what the Q-side per-head RMSNorm + RoPE would cost HERE, in the Q prologue,
      // if the projection GEMM's epilogue left q un-normalised -- the real instruction mix on the real registers (sum of squares by
      // v_dot2c on the packed pairs, the two halves of a row meeting by v_permlane32_swap, x / rms -> bf16 -> * weight by v_dot2 -> bf16 ->
      // rotation -> bf16, then the existing pre-scale), with opaque stand-ins for the weight / table VALUES: the 32 float4 table loads per
      // lane and item are NOT issued, so the price measured is a lower bound
      float cc = scale_log2e, sn = scale_log2e * 0.5f;
      uint32_t wlo = 0x3f80u, whi = 0x3f800000u;
      asm volatile("" : "+v"(cc), "+v"(sn), "+v"(wlo), "+v"(whi));
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const u32x4 pk = __builtin_bit_cast(u32x4, qf[qb][s]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t pi = pk[i];
            ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, pi), __builtin_bit_cast(bf2_t, pi), ss, false);
          }
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
        const float rinv = rsqrtf((__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * (1.0f / 128.0f) + 1e-6f);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          u32x4 pk = __builtin_bit_cast(u32x4, qf[qb][s]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t a = pack_bf2(__uint_as_float(pk[i] << 16) * rinv, __uint_as_float(pk[i] & 0xffff0000u) * rinv);
            float y0, y1;
            y0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, wlo), 0.0f, false);
            y1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, whi), 0.0f, false);
            const uint32_t yp = pack_bf2(y0, y1);
            y0 = __uint_as_float(yp << 16);
            y1 = __uint_as_float(yp & 0xffff0000u);
            pk[i] = pack_bf2(y0 * cc + (-y1) * sn, y1 * cc + y0 * sn);
          }
          qf[qb][s] = __builtin_bit_cast(bf16x8, pk);
        }
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qb][s][e] = (__bf16)((float)qf[qb][s][e] * scale_log2e);
        asm volatile("" : "+a"(qf[qb][s]));   // home of the Q fragments: the AccVGPRs (srcB of the score MFMAs reads them there)
      }
  };

  const int nkv = (Nk + W4_KV - 1) / W4_KV;
  // ---- staging (kreg / vreg, declared in front of the item loop)
  // buffer descriptors sized to this head's N valid rows: a request past them (the rows of a ragged last tile, whole tiles
  // requested past the end of the sequence) returns zeros and moves nothing -- the scores of such keys are masked anyway,
  // and their zero V rows meet zero weights.  (The range check covers VGPR + SGPR offset: tools/ubench/buffer_range.hip.)
  const int ldk2 = (int)ldk * 2, ldv2 = (int)ldv * 2;
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)((uint32_t)(Nk - 1) * (uint32_t)ldk2 + 256u), 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)((uint32_t)(Nk - 1) * (uint32_t)ldv2 + 256u), 0x00020000);
  // the next item's K / V (persistent form; its first tile is requested after this item's output stores -- load_tile(0) below -- and flies under the next prologue)
  // (stream-K: the next pass may start at tile ts2 of its item -- the first piece of this CU's tail range)
  const auto rsK2 = __builtin_amdgcn_make_buffer_rsrc((void*)(Kp + b2 * k_bs + h2 * W4_HD + (int64_t)ts2 * W4_KV * ldk), 0,
                                                      (int)((uint32_t)(N - 1 - ts2 * W4_KV) * (uint32_t)ldk2 + 256u), 0x00020000);
  const auto rsV2 = __builtin_amdgcn_make_buffer_rsrc((void*)(Vp + b2 * v_bs + h2 * W4_HD + (int64_t)ts2 * W4_KV * ldv), 0,
                                                      (int)((uint32_t)(N - 1 - ts2 * W4_KV) * (uint32_t)ldv2 + 256u), 0x00020000);
  // piece i (rows t / 16 + 16 i) of tile j: global -> registers.  Rows are clamped to the last valid key (a no-op on full
  // tiles), tiles to the last tile (the pipeline requests up to two tiles past the end; nobody reads those buffers)
  // tile j: global -> registers, piece i = rows t / 16 + 16 i.  Full tiles: one per-lane offset (rebuilt per burst: as a loop
  // invariant it would pin two registers) and the piece in the scalar offset; the last tile of a ragged N clamps its rows
  // piece i (rows t / 16 + 16 i) of tile j, K or V side: global -> registers.  ko / vo: per-lane byte offset of (row t / 16,
  // chunk t % 16); the tile and the piece go into the scalar offset
  // requests go through rsKs / rsVs: this item's descriptors, except in the last tile of the persistent form, whose staging step
  // requests the NEXT item's first tile (set once per tile: a select per request costs 32 scalar instructions per tile, 3.5 %)
  auto rsKs = rsK, rsVs = rsV;
  auto load_piece = [&](int j, int ko, int vo, auto Ic, bool k_side) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    if (W4_ABL & 32) j = 0;          // timing ablation: every request hits the same (cache-resident) tile
    if constexpr (W4_ABL & 256) {    // timing ablation: the request as LDS-DMA into a scratch region behind the ring (what would a DMA-staged kernel pay?)
      typedef __attribute__((address_space(3))) void lds_void_;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_side ? rsKs : rsVs, (lds_void_*)(smem + ATT_LDS_W4 + (tid >> 6) * 1024), 16, k_side ? ko : vo,
                                               (j * W4_KV + 16 * i) * (k_side ? ldk2 : ldv2), 0, 0);
      return;
    }
    if (k_side)
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsKs, ko, (j * W4_KV + 16 * i) * ldk2, 0));
    else
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsVs, vo, (j * W4_KV + 16 * i) * ldv2, 0));
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
    load_piece(j, ko, vo, IC<0>{}, true); load_piece(j, ko, vo, IC<0>{}, false); load_piece(j, ko, vo, IC<1>{}, true); load_piece(j, ko, vo, IC<1>{}, false);
    load_piece(j, ko, vo, IC<2>{}, true); load_piece(j, ko, vo, IC<2>{}, false); load_piece(j, ko, vo, IC<3>{}, true); load_piece(j, ko, vo, IC<3>{}, false);
  };
  auto write_piece = [&](int buf, auto Ic, bool k_side) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    const int kr = tid >> 4, ch = tid & 15;
    if (k_side)
      *reinterpret_cast<u32x4*>(smem + W4_KBASE + buf * W4_KT + kr * W4_KROW + ch * 16 + i * 16 * W4_KROW) = kreg[i];
    else
      *reinterpret_cast<u32x4*>(smem + buf * W4_VT + kr * 256 + ((((ch >> 2) ^ (kr & 3)) << 6) | ((ch & 3) << 4)) + i * 16 * 256) = vreg[i];
  };
  auto write_tile = [&](int buf) __attribute__((always_inline)) {
    write_piece(buf, IC<0>{}, true); write_piece(buf, IC<0>{}, false); write_piece(buf, IC<1>{}, true); write_piece(buf, IC<1>{}, false);
    write_piece(buf, IC<2>{}, true); write_piece(buf, IC<2>{}, false); write_piece(buf, IC<3>{}, true); write_piece(buf, IC<3>{}, false);
  };
  // ---- fragment read addresses: one per-lane base for K, four (d blocks) for the V transpose reads, + immediates
  const char* rK = smem + W4_KBASE + l31 * W4_KROW + hi * 16;
  const int vi = lane & 15;
  const char* rV[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    rV[db] = smem + (4 * hi + (vi >> 2)) * 256 + 32 * ((lane >> 4) & 1) + (vi & 3) * 8 + ((db ^ ((vi >> 2) & 3)) << 6);
  // koff: byte offset of (buffer, key block) inside the K region; voff: of (buffer, first 16-key step) inside the V region
  auto kread = [&](int koff, int s) __attribute__((always_inline)) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(rK + koff + s * 32);
  };
  auto vread = [&](int voff, int ks, int db) __attribute__((always_inline)) -> bf16x8 {
    const char* va = rV[db] + voff + ks * 16 * 256;
    if constexpr (W4_ABL & 512) return *reinterpret_cast<const bf16x8*>(rK + voff + ks * 16 * 256 + db * 64);   // timing ablation: what a V^T tile would cost to read (one b128, K's conflict-free pattern)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va + 8 * 256));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 o[2][4], ol[2], sc[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ol[qb][r] = 0.f;
#pragma unroll
      for (int db = 0; db < 4; ++db) o[qb][db][r] = 0.f;
    }
  bf16x8 kone, vone, qm[2], kf[8], vf[2][4];
  u32x4 pf[2][2];                // bf16 weights of the pending / the current unit, per q-block and 16-key step
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    kone[e] = (__bf16)0.f;
    vone[e] = (__bf16)1.0f;
    qm[0][e] = qm[1][e] = (__bf16)0.f;
  }
  if (hi == 0) kone[0] = (__bf16)1.0f;
  asm volatile("" : "+a"(kone), "+a"(vone));   // constants of the bookkeeping MFMAs: AccVGPR residents, not re-materialised
  float m_ref[2] = {0.f, 0.f};   // bf16-exact lazy reference maximum per q-block row (exp2 domain); -m_ref sits in qm[.][0]
  float lsum[2] = {0.f, 0.f};    // LV: this lane's share (16 of every 32 keys) of the row sums; the halves meet at the end
  bool any_ref = false;          // LZ: some row of this wave has a non-zero reference (wave-uniform): the offset MFMA is needed

#ifdef W4_DEBUG_CLEAR_LDS
  for (int i = tid * 16; i < ATT_LDS_W4; i += 256 * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0, 0, 0, 0};
  __syncthreads();
#endif
  // ---- prologue: tiles 0 and 1 in LDS, tile 2 requested; K fragments of (tile 0, key block 0); S(0)
  // Tiles 0 AND 1 are requested together (tile 1 into the sixteen registers of the K / V fragments, idle until the first
  // fragment read) and the Q conversion runs under their flight: one memory round trip in front of the first MFMA, not three
  if (!have_pref) load_tile(0);                   // (persistent form, later items: requested inside the previous item's last tile)
  u32x4 k1[4], v1[4];
  {
    int ko, vo;
    stage_offsets(ko, vo);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      k1[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, ko, (W4_KV + 16 * i) * ldk2, 0));
      v1[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, (W4_KV + 16 * i) * ldv2, 0));
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
      *reinterpret_cast<u32x4*>(smem + W4_KBASE + W4_KT + kr * W4_KROW + ch * 16 + i * 16 * W4_KROW) = k1[i];
      *reinterpret_cast<u32x4*>(smem + W4_VT + kr * 256 + ((((ch >> 2) ^ (kr & 3)) << 6) | ((ch & 3) << 4)) + i * 16 * 256) = v1[i];
    }
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 8; ++s) kf[s] = kread(0, s);
  W4_GAP();
  if constexpr (LZ) {
    w4_mfma_s0(sc[0], kf[0], qf[0][0]);           // every reference is 0 at the start: the chain starts from a zero C operand
#pragma unroll
    for (int s = 1; s < 8; ++s) w4_mfma_s(sc[0], kf[s], qf[0][s]);
  } else {
    w4_mfma_s0(sc[0], kone, qm[0]);               // reference 0: a zero product, starts the accumulate chain
#pragma unroll
    for (int s = 0; s < 8; ++s) w4_mfma_s(sc[0], kf[s], qf[0][s]);
  }
  // The first read of these scores is the first step's W4_TOUCH (compiler-visible), in front of which hipcc pads for the chain's last
  // MFMA; dependent MFMAs are interlocked among themselves.  Rounds 2-3 put 24 x `s_nop 15` here and before the output (from the time
  // when every read was inline asm with no visible one in front): dead time twice per item, removed in round 4 (-DW4_KEEP_DRAINS restores
  // them; tools/check_mfma_hazard.py and the determinism tests are the net).
#ifdef W4_KEEP_DRAINS
  W4_DRAIN_MFMA();
#endif
  W4_GAP();

  // One pipeline step = unit u of q-block QB (u & 1).
  //   VALU : softmax of S(u) (sc[QB]) -> P(u) (pf[QB]);  rarely: move q-block QB's reference
  //   MFMA : S(u+1) of the OTHER q-block (sc[OQ]) from the K fragments in kf;  pending P(u-1).V of the other q-block (pf[OQ], vf)
  //   EVEN : (QB == 0) last user of kf / vf -> each register is reloaded right after its use: kf <- K rows at KN, vf <- V rows at VN
  //   FIRST: the q-block's first unit pins the reference to the true row maximum
  //   rag  : this unit's key block reaches past N (last tile only): keys >= N are masked; kb_abs = its first key
  //   STG  : staging spread over this step's MFMA gaps: 1 = tile jst -> LDS buffer (STG >> 2) from the registers, 2 = request tile jst
  auto step = [&](auto QBc, auto PVc, auto FIRSTc, auto KNc, auto VNc, auto STGc, int jst, bool rag, int kb_abs) __attribute__((always_inline)) {
    constexpr int QB = decltype(QBc)::value, OQ = 1 - QB;
    constexpr bool EVEN = QB == 0, PV = decltype(PVc)::value != 0, FIRST = decltype(FIRSTc)::value != 0;
    constexpr int KN = decltype(KNc)::value, VN = decltype(VNc)::value, STG = decltype(STGc)::value;
    f32x16& cur = sc[QB];
    f32x16& nxt = sc[OQ];
    if (__builtin_expect_with_probability(rag, 0, 1.0)) {
      int kbase = kb_abs + 4 * hi;
      asm volatile("" : "+v"(kbase));     // keep the index arithmetic inside the (last-tile-only) branch
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) >= Nk) cur[r] = -INFINITY;
    }
    // MFMA i of the S chain (0: the reference offset, 1..8: the head-dim steps) and of the pending P.V (ks = i / 5, block i % 5)
    auto S = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
      if constexpr (i == 0) {
        w4_mfma_s0(nxt, kone, qm[OQ]);
      } else if constexpr (i == 1 && LZ) {
        // the chain starts here; the reference offset joins it (one more MFMA, out of line) only while some row of this wave has one
        w4_mfma_s0(nxt, kf[0], qf[OQ][0]);
        if constexpr (!NOREF) {
          if (__builtin_expect_with_probability(any_ref, 0, 1.0)) w4_mfma_s(nxt, kone, qm[OQ]);
        }
      } else {
        w4_mfma_s(nxt, kf[i - 1], qf[OQ][i - 1]);
        // reloaded two MFMAs after its last reader: spreads the LDS reads over the regions
        if constexpr (EVEN && i >= 2 && !(W4_ABL & 2)) kf[i - 2] = kread(KN, i - 2);
      }
    };
    // LV: eight P.V MFMAs (ks = i / 4, d block i % 4), no row-sum MFMA; each V fragment is reloaded one MFMA after its use, the last
    // one and kf[7] by RL() behind the step's last MFMA
    auto PL = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value, ks = i / 4, db = i % 4;
      if constexpr (PV) w4_mfma_o(o[OQ][db], vf[ks][db], pf[OQ][ks]);
      if constexpr (EVEN && i >= 1 && !(W4_ABL & 2)) vf[(i - 1) / 4][(i - 1) % 4] = vread(VN, (i - 1) / 4, (i - 1) % 4);
    };
    auto RL = [&]() __attribute__((always_inline)) {
      if constexpr (EVEN && !(W4_ABL & 2)) {
        vf[1][3] = vread(VN, 1, 3);
        kf[7] = kread(KN, 7);
      }
    };
    // LV: row sums of the PENDING unit (other q-block, weights pf[OQ] finished last step) -- one v_dot2c_f32_bf16 per packed pair
    auto DL = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int c = decltype(Ic)::value;
      if constexpr (PV && !(W4_ABL & 1)) w4_dot2c(lsum[OQ], pf[OQ][c >> 2][c & 3]);
    };
    auto P = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value, ks = i / 5, db = i % 5;
      if constexpr (PV) {
        if constexpr (db == 4) {
          w4_mfma_l(ol[OQ], vone, pf[OQ][ks]);
        } else {
          w4_mfma_o(o[OQ][db], vf[ks][db], pf[OQ][ks]);
        }
      }
      if constexpr (EVEN && i >= 1 && (i - 1) % 5 < 4 && !(W4_ABL & 2)) vf[(i - 1) / 5][(i - 1) % 5] = vread(VN, (i - 1) / 5, (i - 1) % 5);
      if constexpr (EVEN && i == 9 && !(W4_ABL & 2)) kf[7] = kread(KN, 7);
    };
    // VALU stream of the exponentials, one instruction per call, in an order in which every bf16 pack follows its two
    // v_exp_f32 by at least two instructions (a transcendental result needs one instruction before a VALU reads it, and the asm
    // pack is invisible to hipcc's hazard pass):  e0 e1 e2 c0 e3 e4 c1 e5 e6 c2 ... e13 e14 c6 e15 c7
    auto F = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int k = decltype(Ic)::value;
      if constexpr (W4_ABL & 1) return;
      // LV modes and MODE 4 end  e13 e14 e15 c6 c7  (no MFMA region separates the last pack from its exponentials there)
      constexpr bool is_c = FT ? (k >= 22 || (k >= 3 && k <= 18 && k % 3 == 0)) : ((k == 23) || (k >= 3 && k < 22 && k % 3 == 0));
      if constexpr (is_c) {
        constexpr int c = FT ? (k >= 22 ? k - 16 : k / 3 - 1) : (k == 23 ? 7 : k / 3 - 1);
        pf[QB][c >> 2][c & 3] = w4_cvt_pk(cur[2 * c], cur[2 * c + 1]);
      } else {
        constexpr int e = FT ? (k < 3 ? k : k >= 19 ? k - 6 : k - (k / 3))
                             : (k < 3 ? k : k == 22 ? 15 : k - (k / 3));      // exponentials seen so far = index minus packs before it
        cur[e] = __builtin_amdgcn_exp2f(cur[e]);
      }
    };
    // staging stream of the step that moves tile jst - 1 from the registers into LDS buffer STG >> 2 and refills every register
    // with its piece of tile jst right behind:  W0 W1 L0 W2 L1 W3 L2 ... W7 L6 L7  (piece g = K / V chunk g / 2, K first)
    int stg_ko = 0, stg_vo = 0;
    if constexpr ((STG & 3) == 1 && !(W4_ABL & 4)) stage_offsets(stg_ko, stg_vo);
    auto G = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int n = decltype(Ic)::value;            // 0..15
      if constexpr ((STG & 3) == 1 && !(W4_ABL & 4)) {
        constexpr bool is_w = n == 0 || n == 1 || (n < 15 && (n & 1));   // W0 W1 | L0 W2 L1 W3 ... L5 W7 | L6 L7
        constexpr int g = n < 2 ? n : is_w ? (n + 1) / 2 : n == 15 ? 7 : n / 2 - 1;
        if constexpr (is_w) { if constexpr (!(W4_ABL & 64)) write_piece(STG >> 2, IC<g / 2>{}, (g & 1) == 0); }
        else { if constexpr (!(W4_ABL & 128)) load_piece(jst, stg_ko, stg_vo, IC<g / 2>{}, (g & 1) == 0); }
      }
    };
    // One MFMA per scheduling region, each with <= ~24 issue cycles of other work behind it (an MFMA that finds the pipe busy
    // blocks the wave's issue until the pipe takes it, so work behind TWO adjacent MFMAs is not hidden by the first).
    // The score chain runs one MFMA ahead of the P.V group: its last MFMA S(8) is followed by three more MFMAs of this step
    // and the first of the next before anything reads the scores -- an MFMA result needs ~11 issued instructions before a
    // VALU read, and hipcc cannot insert that wait in front of the inline-asm maxima (it does not know they are VALU).
    float mx = 0.f;
    // the four pieces of the row maximum (VALU, inline asm) and the rare reference move, shared by the three region lists below
    auto MX = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
      if constexpr (W4_ABL & 8) return;
      if constexpr (i == 0) mx = w4_max7(cur[0], cur[1], cur[2], cur[3], cur[4], cur[5], cur[6]);
      if constexpr (i == 1) mx = w4_max7(mx, cur[7], cur[8], cur[9], cur[10], cur[11], cur[12]);
      if constexpr (i == 2) mx = w4_max4(mx, cur[13], cur[14], cur[15]);
      if constexpr (i == 3) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = w4_max(__uint_as_float(sw[0]), __uint_as_float(sw[1]));   // both 32-key halves of the row
      }
    };
    auto MOVE = [&]() __attribute__((always_inline)) {
      // out of line: with one wave per SIMD nothing hides the instruction-fetch bubble of a TAKEN branch, so the common
      // path must be the fall-through
      bool take;
      if constexpr (LZ) take = !(W4_ABL & 8) && !__all(FIRST ? fabsf(mx) <= W4_BIG : mx <= W4_BIG);
      else take = FIRST || (!(W4_ABL & 8) && !__all(mx <= W4_THR));
      if (__builtin_expect_with_probability(take, FIRST && !LZ, 1.0)) {
        // move the reference: everything q-block QB accumulated against the old one is rescaled exactly once (no MFMA on
        // o[QB] / ol[QB] is in this step's stream), the scores of this unit are shifted before they are exponentiated
        float m_new;
        if constexpr (LZ) {   // only the rows that need it: a first tile's maximum outside +-W4_BIG pins, a later excess beyond W4_BIG moves
          const bool mv = FIRST ? fabsf(mx) > W4_BIG : mx > W4_BIG;
          m_new = mv ? round_bf(m_ref[QB] + mx) : m_ref[QB];
        } else {
          m_new = round_bf(m_ref[QB] + (FIRST ? mx : fmaxf(mx, 0.f)));
        }
        const float d = m_new - m_ref[QB];
        m_ref[QB] = m_new;
        qm[QB][0] = (__bf16)(hi == 0 ? -m_new : 0.f);
        if constexpr (LZ) any_ref = __any(m_ref[0] != 0.f || m_ref[1] != 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] -= d;
        if constexpr (!FIRST) {
          const float f = __builtin_amdgcn_exp2f(-d);
          if constexpr (LV) lsum[QB] *= f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (!LV) ol[QB][r] *= f;
#pragma unroll
            for (int db = 0; db < 4; ++db) o[QB][db][r] *= f;
          }
        }
      }
    };
    if constexpr (MODE == 0) {
    S(IC<0>{});
    W4_TOUCH(cur);
    MX(IC<0>{});
    W4_GAP();
    S(IC<1>{});
    MX(IC<1>{});
    W4_GAP();
    P(IC<0>{});
    MX(IC<2>{});
    W4_GAP();
    S(IC<2>{});
    MX(IC<3>{});
    W4_GAP();
    P(IC<1>{});
    W4_GAP();
    MOVE();
    W4_GAP();
    S(IC<3>{}); F(IC<0>{}); F(IC<1>{}); G(IC<0>{});
    W4_GAP();
    P(IC<2>{}); F(IC<2>{}); F(IC<3>{}); G(IC<1>{});
    W4_GAP();
    S(IC<4>{}); F(IC<4>{}); F(IC<5>{}); G(IC<2>{});
    W4_GAP();
    P(IC<3>{}); F(IC<6>{}); F(IC<7>{}); G(IC<3>{});
    W4_GAP();
    S(IC<5>{}); F(IC<8>{}); F(IC<9>{}); G(IC<4>{});
    W4_GAP();
    P(IC<4>{}); F(IC<10>{}); F(IC<11>{}); G(IC<5>{});
    W4_GAP();
    S(IC<6>{}); F(IC<12>{}); F(IC<13>{}); G(IC<6>{});
    W4_GAP();
    P(IC<5>{}); F(IC<14>{}); F(IC<15>{}); G(IC<7>{});
    W4_GAP();
    S(IC<7>{}); F(IC<16>{}); F(IC<17>{}); G(IC<8>{});
    W4_GAP();
    P(IC<6>{}); F(IC<18>{}); F(IC<19>{}); G(IC<9>{});
    W4_GAP();
    S(IC<8>{}); F(IC<20>{}); F(IC<21>{}); G(IC<10>{});
#ifdef W4_HAZARD_SELFTEST   // a score read from inline asm two instructions behind its chain's last MFMA: what check_mfma_hazard.py must catch
    { const float t_ = w4_max(nxt[0], nxt[1]); asm volatile("" ::"v"(t_)); }
#endif
    W4_GAP();
    P(IC<7>{}); F(IC<22>{}); G(IC<11>{});
    W4_GAP();
    P(IC<8>{}); G(IC<12>{}); G(IC<13>{});
    W4_GAP();
    P(IC<9>{}); G(IC<14>{}); G(IC<15>{});
    W4_GAP();
    F(IC<23>{});
    W4_GAP();
    } else if constexpr (MODE == 1) {
    // 17 MFMAs: S0..S8 interleaved with the eight P.V; the pending unit's row sums (DL) ride behind the first five, the staging
    // stream (G, one step in four) one piece per region
    S(IC<0>{}); W4_TOUCH(cur); MX(IC<0>{}); G(IC<0>{});
    W4_GAP();
    S(IC<1>{}); MX(IC<1>{}); DL(IC<0>{}); G(IC<1>{});
    W4_GAP();
    PL(IC<0>{}); MX(IC<2>{}); DL(IC<1>{}); DL(IC<2>{}); G(IC<2>{});
    W4_GAP();
    S(IC<2>{}); MX(IC<3>{}); DL(IC<3>{}); DL(IC<4>{}); G(IC<3>{});
    W4_GAP();
    PL(IC<1>{}); DL(IC<5>{}); DL(IC<6>{}); DL(IC<7>{}); G(IC<4>{});
    W4_GAP();
    MOVE();
    W4_GAP();
    S(IC<3>{}); F(IC<0>{}); F(IC<1>{}); G(IC<5>{});
    W4_GAP();
    PL(IC<2>{}); F(IC<2>{}); F(IC<3>{}); G(IC<6>{});
    W4_GAP();
    S(IC<4>{}); F(IC<4>{}); F(IC<5>{}); G(IC<7>{});
    W4_GAP();
    PL(IC<3>{}); F(IC<6>{}); F(IC<7>{}); G(IC<8>{});
    W4_GAP();
    S(IC<5>{}); F(IC<8>{}); F(IC<9>{}); G(IC<9>{});
    W4_GAP();
    PL(IC<4>{}); F(IC<10>{}); F(IC<11>{}); G(IC<10>{});
    W4_GAP();
    S(IC<6>{}); F(IC<12>{}); F(IC<13>{}); G(IC<11>{});
    W4_GAP();
    PL(IC<5>{}); F(IC<14>{}); F(IC<15>{}); G(IC<12>{});
    W4_GAP();
    S(IC<7>{}); F(IC<16>{}); F(IC<17>{}); G(IC<13>{});
    W4_GAP();
    PL(IC<6>{}); F(IC<18>{}); F(IC<19>{}); G(IC<14>{});
    W4_GAP();
    S(IC<8>{}); F(IC<20>{}); F(IC<21>{}); G(IC<15>{});
#ifdef W4_HAZARD_SELFTEST
    { const float t_ = w4_max(nxt[0], nxt[1]); asm volatile("" ::"v"(t_)); }
#endif
    W4_GAP();
    PL(IC<7>{}); F(IC<22>{}); RL(); F(IC<23>{});
    W4_GAP();
    } else if constexpr (MODE == 3) {
    // MODE 0's stream without the reference-offset MFMA (it joins S(1), out of line, only while a row of the wave has a reference):
    // 18 MFMAs, the VALU stream unchanged
    S(IC<1>{}); W4_TOUCH(cur); MX(IC<0>{});
    W4_GAP();
    P(IC<0>{}); MX(IC<1>{});
    W4_GAP();
    S(IC<2>{}); MX(IC<2>{});
    W4_GAP();
    P(IC<1>{}); MX(IC<3>{});
    W4_GAP();
    MOVE();
    W4_GAP();
    S(IC<3>{}); F(IC<0>{}); F(IC<1>{}); G(IC<0>{});
    W4_GAP();
    P(IC<2>{}); F(IC<2>{}); F(IC<3>{}); G(IC<1>{});
    W4_GAP();
    S(IC<4>{}); F(IC<4>{}); F(IC<5>{}); G(IC<2>{});
    W4_GAP();
    P(IC<3>{}); F(IC<6>{}); F(IC<7>{}); G(IC<3>{});
    W4_GAP();
    S(IC<5>{}); F(IC<8>{}); F(IC<9>{}); G(IC<4>{});
    W4_GAP();
    P(IC<4>{}); F(IC<10>{}); F(IC<11>{}); G(IC<5>{});
    W4_GAP();
    S(IC<6>{}); F(IC<12>{}); F(IC<13>{}); G(IC<6>{});
    W4_GAP();
    P(IC<5>{}); F(IC<14>{}); F(IC<15>{}); G(IC<7>{});
    W4_GAP();
    S(IC<7>{}); F(IC<16>{}); F(IC<17>{}); G(IC<8>{});
    W4_GAP();
    P(IC<6>{}); F(IC<18>{}); F(IC<19>{}); G(IC<9>{});
    W4_GAP();
    S(IC<8>{}); F(IC<20>{}); F(IC<21>{}); G(IC<10>{});
#ifdef W4_HAZARD_SELFTEST
    { const float t_ = w4_max(nxt[0], nxt[1]); asm volatile("" ::"v"(t_)); }
#endif
    W4_GAP();
    P(IC<7>{}); F(IC<22>{}); G(IC<11>{});
    W4_GAP();
    P(IC<8>{}); G(IC<12>{}); G(IC<13>{});
    W4_GAP();
    P(IC<9>{}); G(IC<14>{}); G(IC<15>{});
    W4_GAP();
    F(IC<23>{});
    W4_GAP();
    } else if constexpr (MODE == 4) {
    // the caller's score bound is admissible (attn_bound_admissible: bound log2 e + log2 N + 24 <= 126, |v| <= 2^24 granted): no reference at all -- no row maximum, no branch, no offset
    // MFMA; the exponentials start with the step and spread over all 18 regions (one staging piece each in 16 of them).  Every pack
    // reads exponentials that are at least TWO regions old (e0 e1 | e2 | e3 | c0 e4 | e5 | c1 e6 | ...), so whatever order hipcc gives
    // the instructions inside a region -- it is free to sink the compiler-visible v_exp_f32 behind the region's asm pack, or to hoist
    // the next region's pack over its MFMA -- an MFMA lies between a transcendental result and its (invisible) reader; the first
    // version kept one region of distance and needed an s_nop in front of every pack (tools/check_mfma_hazard.py found the P.V-less
    // first step's violations)
    auto E = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int e = decltype(Ic)::value;
      if constexpr (!(W4_ABL & 1)) cur[e] = __builtin_amdgcn_exp2f(cur[e]);
    };
    auto C2 = [&](auto Ic) __attribute__((always_inline)) {
      constexpr int c = decltype(Ic)::value;
      if constexpr (!(W4_ABL & 1)) pf[QB][c >> 2][c & 3] = w4_cvt_pk(cur[2 * c], cur[2 * c + 1]);
    };
    S(IC<1>{}); W4_TOUCH(cur); E(IC<0>{}); E(IC<1>{}); G(IC<0>{});
    W4_GAP();
    P(IC<0>{}); E(IC<2>{}); G(IC<1>{});
    W4_GAP();
    S(IC<2>{}); E(IC<3>{}); G(IC<2>{});
    W4_GAP();
    P(IC<1>{}); C2(IC<0>{}); E(IC<4>{}); G(IC<3>{});
    W4_GAP();
    S(IC<3>{}); E(IC<5>{}); G(IC<4>{});
    W4_GAP();
    P(IC<2>{}); C2(IC<1>{}); E(IC<6>{}); G(IC<5>{});
    W4_GAP();
    S(IC<4>{}); E(IC<7>{}); G(IC<6>{});
    W4_GAP();
    P(IC<3>{}); C2(IC<2>{}); E(IC<8>{}); G(IC<7>{});
    W4_GAP();
    S(IC<5>{}); E(IC<9>{}); G(IC<8>{});
    W4_GAP();
    P(IC<4>{}); C2(IC<3>{}); E(IC<10>{}); G(IC<9>{});
    W4_GAP();
    S(IC<6>{}); E(IC<11>{}); G(IC<10>{});
    W4_GAP();
    P(IC<5>{}); C2(IC<4>{}); E(IC<12>{}); G(IC<11>{});
    W4_GAP();
    S(IC<7>{}); E(IC<13>{}); G(IC<12>{});
    W4_GAP();
    P(IC<6>{}); C2(IC<5>{}); E(IC<14>{}); G(IC<13>{});
    W4_GAP();
    S(IC<8>{}); E(IC<15>{}); G(IC<14>{});
#ifdef W4_HAZARD_SELFTEST
    { const float t_ = w4_max(nxt[0], nxt[1]); asm volatile("" ::"v"(t_)); }
#endif
    W4_GAP();
    P(IC<7>{}); C2(IC<6>{}); G(IC<15>{});
    W4_GAP();
    P(IC<8>{});
    W4_GAP();
    P(IC<9>{}); C2(IC<7>{});
    W4_GAP();
    } else {
    // 16 MFMAs on ordinary data (S(1) starts the chain; the reference offset joins it, out of line, only while a row has one)
    S(IC<1>{}); W4_TOUCH(cur); MX(IC<0>{}); G(IC<0>{});
    W4_GAP();
    PL(IC<0>{}); MX(IC<1>{}); DL(IC<0>{}); G(IC<1>{});
    W4_GAP();
    S(IC<2>{}); MX(IC<2>{}); DL(IC<1>{}); DL(IC<2>{}); G(IC<2>{});
    W4_GAP();
    PL(IC<1>{}); MX(IC<3>{}); DL(IC<3>{}); DL(IC<4>{}); G(IC<3>{});
    W4_GAP();
    S(IC<3>{}); DL(IC<5>{}); DL(IC<6>{}); DL(IC<7>{}); G(IC<4>{});
    W4_GAP();
    MOVE();
    W4_GAP();
    PL(IC<2>{}); F(IC<0>{}); F(IC<1>{}); G(IC<5>{});
    W4_GAP();
    S(IC<4>{}); F(IC<2>{}); F(IC<3>{}); G(IC<6>{});
    W4_GAP();
    PL(IC<3>{}); F(IC<4>{}); F(IC<5>{}); G(IC<7>{});
    W4_GAP();
    S(IC<5>{}); F(IC<6>{}); F(IC<7>{}); G(IC<8>{});
    W4_GAP();
    PL(IC<4>{}); F(IC<8>{}); F(IC<9>{}); G(IC<9>{});
    W4_GAP();
    S(IC<6>{}); F(IC<10>{}); F(IC<11>{}); G(IC<10>{});
    W4_GAP();
    PL(IC<5>{}); F(IC<12>{}); F(IC<13>{}); G(IC<11>{});
    W4_GAP();
    S(IC<7>{}); F(IC<14>{}); F(IC<15>{}); G(IC<12>{});
    W4_GAP();
    PL(IC<6>{}); F(IC<16>{}); F(IC<17>{}); G(IC<13>{});
    W4_GAP();
    S(IC<8>{}); F(IC<18>{}); F(IC<19>{}); G(IC<14>{});
#ifdef W4_HAZARD_SELFTEST
    { const float t_ = w4_max(nxt[0], nxt[1]); asm volatile("" ::"v"(t_)); }
#endif
    W4_GAP();
    PL(IC<7>{}); F(IC<20>{}); F(IC<21>{}); G(IC<15>{}); RL();
    W4_GAP();
    F(IC<22>{}); F(IC<23>{});
    W4_GAP();
    }
  };

  const int j_rag = (Nk & (W4_KV - 1)) ? nkv - 1 : -1;          // the tile whose key blocks reach past N
  // one 64-key tile j out of ring buffer B (compile-time: every fragment address is a per-lane base plus an immediate)
  auto tile = [&](int j, auto Bc, auto FIRSTc) __attribute__((always_inline)) {
    constexpr int B = decltype(Bc)::value, NB = (B + 1) % 3, WB = (B + 2) % 3;
    constexpr int FIRST = decltype(FIRSTc)::value;
    // (one scalar compare each against tile indices fixed in front of the loop: with one wave per SIMD every scalar instruction of
    // the tile loop is issue time)
    const bool rag = j == j_rag;
    // (kb0, q0): S(kb0, q1);  pending (tile j-1: kb1, q1);  reload kf <- K(j) kb1, vf <- V(j) kb0
    step(IC<0>{}, IC<!FIRST>{}, IC<FIRST>{}, IC<B * W4_KT + 32 * W4_KROW>{}, IC<B * W4_VT>{}, IC<0>{}, 0, rag, j * W4_KV);
    // (kb0, q1): S(kb1, q0);  pending (kb0, q0);  + staging: tile j + 2 (requested one tile ago) goes into the buffer tile j - 1
    // left before the last barrier, and each register is refilled with its piece of tile j + 3 right behind its write -- four
    // steps (> 1 us) before it is needed: with one wave per SIMD nothing else runs while a wave waits for memory
    step(IC<1>{}, IC<1>{}, IC<FIRST>{}, IC<0>{}, IC<0>{}, IC<1 + 4 * WB>{}, j + 3, rag, j * W4_KV);
    // (kb1, q0): S(kb1, q1);  pending (kb0, q1);  reload kf <- K(j+1) kb0, vf <- V(j) kb1
    step(IC<0>{}, IC<1>{}, IC<0>{}, IC<NB * W4_KT>{}, IC<B * W4_VT + 2 * 16 * 256>{}, IC<0>{}, 0, rag, j * W4_KV + 32);
    // (kb1, q1): S(tile j+1: kb0, q0);  pending (kb1, q0)
    step(IC<1>{}, IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, 0, rag, j * W4_KV + 32);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(W4_ABL & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  W4_STAMP(1);
  tile(0, IC<0>{}, IC<1>{});
  for (int j = 1; j < nkv; j += 3) {
    tile(j, IC<1>{}, IC<0>{});
    if (j + 1 < nkv) tile(j + 1, IC<2>{}, IC<0>{});
    if (j + 2 < nkv) tile(j + 2, IC<0>{}, IC<0>{});
  }
  W4_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // requests past the end of the sequence (zeros) still write their registers
  // persistent form: the NEXT item's Q rows are requested here, behind the tile loop, and fly while this item's output is normalised,
  // staged and stored; its first K / V tile is requested behind the output's stores and flies under the next prologue's Q conversion.
  // (Measured on the way, tools/attn_item_timers.py: requests from inside the last tiles cost EVERY tile ~60 cycles -- even a single
  // compare + branch per tile: the next-Q registers live across the loop --, 4.2 k per item; all 24 requests in one burst here make the
  // output's stores queue behind them, +4.6 k.)
  if (has_next) load_qn();
  // ---- drain: the pending P.V of the very last unit (last tile: kb1, q1), V fragments already in registers
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int db = 0; db < 4; ++db) w4_mfma_o(o[1][db], vf[ks][db], pf[1][ks]);
    if constexpr (!LV) w4_mfma_l(ol[1], vone, pf[1][ks]);
  }
  float ltot[2];                                 // full row sums (LV: the two 16-key halves of every unit live in lanes l and l ^ 32)
  if constexpr (LV) {
#pragma unroll
    for (int c = 0; c < 8; ++c) w4_dot2c(lsum[1], pf[1][c >> 2][c & 3]);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) ltot[qb] = lsum[qb] + __shfl_xor(lsum[qb], 32, 64);
  }
#ifdef W4_KEEP_DRAINS
  W4_DRAIN_MFMA();                               // the last MFMA results before the VALU reads them
#endif
  if constexpr (!LV) {
    ltot[0] = ol[0][0];                          // every row of the ones-block holds the full row sum
    ltot[1] = ol[1][0];
  }

  // ---- tail split: un-normalised O (fp32), row sum and reference maximum of this key range -> part [tile][range][row][132]
  if (kpart >= 0 || pslot >= 0) {
    float* pp = part + (kpart >= 0 ? (int64_t)ptile * nsplit + kpart : (int64_t)pslot) * (256 * W4_PROW);
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float* pr = pp + (wave * 2 + qb) * (16 * 256) + lane * 4;        // w4_part_off: 1 KiB contiguous per store instruction
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<f32x4*>(pr + (db * 4 + qd) * 256) =
              f32x4{o[qb][db][qd * 4 + 0], o[qb][db][qd * 4 + 1], o[qb][db][qd * 4 + 2], o[qb][db][qd * 4 + 3]};
      if (hi == 0) *reinterpret_cast<float2*>(pp + W4_PART_LM + (wave * 64 + qb * 32 + l31) * 2) = float2{ltot[qb], m_ref[qb]};
    }
    if (kpart >= 0) return;
  } else {
  // ---- finish.  The normalised bf16 rows go through a wave-private LDS
  // tile (the K / V ring is free: every wave's last fragment read lies before the last barrier) and leave as whole 256-byte
  // rows, 16 lanes x 16 bytes each (row-per-lane 8-byte stores touch 32 lines per instruction and queue up at the end of
  // the block, when every wave of the workgroup stores at once)
  constexpr int OROW = 272;
  char* ot = smem + wave * (64 * OROW);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float inv = 1.0f / ltot[qb];
    char* orow = ot + (qb * 32 + l31) * OROW + 8 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2 w;
        w[0] = pack_bf2(o[qb][db][qd * 4 + 0] * inv, o[qb][db][qd * 4 + 1] * inv);
        w[1] = pack_bf2(o[qb][db][qd * 4 + 2] * inv, o[qb][db][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + db * 64 + qd * 16) = w;
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
  }    // (whole item)
#ifdef TFX_BENCH
  W4_STAMP(3);
  w4_sum[0] += w4_stamp[1] - w4_stamp[0]; w4_sum[1] += w4_stamp[2] - w4_stamp[1]; w4_sum[2] += w4_stamp[3] - w4_stamp[2]; w4_sum[3] += 1;
  if (!has_next && g_w4_timers && tid == 0)
    for (int i = 0; i < 4; ++i) atomicAdd(g_w4_timers + i, w4_sum[i]);
#endif
  if (!has_next) break;
  rsKs = rsK2;
  rsVs = rsV2;
  load_tile(0);                                   // the next item's first K / V tile (its descriptors; kreg / vreg are idle until the next prologue writes them)
  item = item2; b = b2; h = h2; qblk = qblk2;
  if (sk_share > 0) { ++sk_u; int it_; sk_unit(sk_u, it_, ts, te); }
  have_pref = true;
  // every wave has read its output tile out of LDS: the ring buffers are free for the next item.  A bare barrier behind an LDS-only
  // wait -- __syncthreads() would also wait for the next item's requests that were just issued (4.6 k cycles per item, measured)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  }
}

// Finishes the q-tiles of a tail split: O = sum_r o_r 2^(m_r - m) / sum_r l_r 2^(m_r - m), m = max_r m_r (exp2 domain, the
// kernel's own bookkeeping).  One thread per (row, 4 head-dim columns).
__global__ __launch_bounds__(256) void attn_w4_merge_kernel(const float* part, bf16_t* O, int64_t ldo, int64_t o_bs, int H, int N,
                                                            int nqb, int xsplit, int nsplit) {
  const int rg = blockIdx.x * 8 + (threadIdx.x >> 5), c = (threadIdx.x & 31) * 4;
  const int ptile = rg >> 8, r = rg & 255;
  const int pair = H * nqb - xsplit + ptile % xsplit, b = ptile / xsplit, h = pair / nqb, qblk = pair % nqb;
  const int row = qblk * 256 + r;
  if (row >= N) return;
  const float* pr = part + (int64_t)ptile * nsplit * (256 * W4_PROW);
  const int po = w4_part_off(r, c), lo = W4_PART_LM + r * 2;
  float m = -INFINITY;
  for (int k = 0; k < nsplit; ++k) m = fmaxf(m, pr[(int64_t)k * 256 * W4_PROW + lo + 1]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int k = 0; k < nsplit; ++k) {
    const float* pk = pr + (int64_t)k * 256 * W4_PROW;
    const float w = __builtin_amdgcn_exp2f(pk[lo + 1] - m);
    const f32x4 v = *reinterpret_cast<const f32x4*>(pk + po);
    acc += v * w;
    l += pk[lo] * w;
  }
  const float inv = 1.0f / l;
  u32x2 o2;
  o2[0] = pack_bf2(acc[0] * inv, acc[1] * inv);
  o2[1] = pack_bf2(acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<u32x2*>(O + b * o_bs + (int64_t)row * ldo + h * W4_HD + c) = o2;
}

// Stream-K: adds up the partials of every tail item that a CU boundary cuts.  8 blocks per (sample, boundary), one per 32-row block of the item;
// the blocks of a boundary that cuts nothing, or that is not the FIRST boundary inside its item, leave at once.  Parts of tail item i, in key
// order: slot B (2 L + 1) of the CU whose range contains the item's first tile, then slot A (2 L) of every CU whose non-empty range starts
// inside the item.  Threads read in the producer's register order (w4_part_off: wave w = column block db, 1 KiB contiguous per load
// instruction; the first version read row-major positions out of that layout and ran at 1.6 TB/s -- 36 us per launch, more than the
// dealing saves), the bf16 rows leave through an LDS tile as whole 256-byte rows.  Same arithmetic as attn_w4_merge_kernel (the reference
// maxima are all 0 on the reference-free stream: the weights are 1, the merge a plain sum).
__global__ __launch_bounds__(256) void attn_w4_merge_sk_kernel(const float* part, bf16_t* O, int64_t ldo, int64_t o_bs, int H, int N, int nqb,
                                                               int group, int share, int Tp) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[32][128 + 8];
  const int bi = blockIdx.x >> 3, wq = blockIdx.x & 7;
  const int sample = bi / group, c = bi - sample * group;
  if (c == 0) return;
  const int nkv = (N + W4_KV - 1) / W4_KV, full = (Tp / group) * group, total = (Tp - full) * nkv;
  const int p = w4_sk_bound(c, group, share, nkv, total);
  if (p >= total || p % nkv == 0) return;
  const int il = p / nkv, istart = il * nkv, iend = istart + nkv;
  if (w4_sk_bound(c - 1, group, share, nkv, total) > istart) return;
  const int gi = full + il, h = gi / nqb, qblk = gi - h * nqb;
  if (qblk * 256 + wq * 32 >= N) return;
  int slots[8], np = 0;
  slots[np++] = 2 * (sample * group + c - 1) + 1;
  for (int j = c; j < group && np < 8; ++j) {
    const int lo = w4_sk_bound(j, group, share, nkv, total), hi = w4_sk_bound(j + 1, group, share, nkv, total);
    if (lo >= iend || lo >= total) break;
    if (hi > lo) slots[np++] = 2 * (sample * group + j);
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int lmo = W4_PART_LM + (wq * 32 + l31) * 2, vo = ((wq * 4 + w) * 4 * 64 + lane) * 4;
  float2 lm[8];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    lm[k] = float2{0.f, -INFINITY};
    if (k < np) lm[k] = *reinterpret_cast<const float2*>(part + (int64_t)slots[k] * (256 * W4_PROW) + lmo);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) m = fmaxf(m, lm[k].y);
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float l = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < np) {            // block-uniform
      const float* pk = part + (int64_t)slots[k] * (256 * W4_PROW) + vo;
      f32x4 v[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) v[qd] = *reinterpret_cast<const f32x4*>(pk + qd * 256);
      const float wgt = __builtin_amdgcn_exp2f(lm[k].y - m);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) acc[qd] += v[qd] * wgt;
      l += lm[k].x * wgt;
    }
  const float inv = 1.0f / l;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    u32x2 o2;
    o2[0] = pack_bf2(acc[qd][0] * inv, acc[qd][1] * inv);
    o2[1] = pack_bf2(acc[qd][2] * inv, acc[qd][3] * inv);
    *reinterpret_cast<u32x2*>(&sm[l31][w * 32 + qd * 8 + hi * 4]) = o2;
  }
  __syncthreads();
  const int orow = tid >> 3, och = tid & 7, row = qblk * 256 + wq * 32 + orow;
  if (row < N) {
    bf16_t* dst = O + sample * o_bs + (int64_t)row * ldo + h * W4_HD + och * 16;
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(&sm[orow][och * 16]);
    *reinterpret_cast<u32x4*>(dst + 8) = *reinterpret_cast<const u32x4*>(&sm[orow][och * 16 + 8]);
  }
}

// scratch of the tail split: one per (device, stream) that ever ran a split launch -- two launches in flight on different streams
// (a graph replay on a side stream next to an eager call) must not share partials.  Allocated on first use outside a stream
// capture, freed only by tfx_release_scratch (captured graphs keep the pointer); a launch on a stream without scratch (first seen
// during capture, or the table is full) simply runs unsplit.
static constexpr int W4_PART_TILES = 1024;   // (q-tile, key range) slots: 2 rounds of a 512-CU chip, 138 MB
static constexpr int W4_PART_SLOTS = 8;
struct W4Scratch { int dev; hipStream_t st; float* part; int cus; };
static W4Scratch g_w4_scr[W4_PART_SLOTS];
static int g_w4_nscr = 0;
static std::mutex g_w4_mu;
// tfx_set_option attention_tail_split: OFF by default -- a sample's attention output must not depend on how many samples share its
// batch (tests/test_fullsize_gpu.py asserts it bit for bit), and which tiles fall into the last round does; 1 = split when it pays
static int g_w4_split = 0;
void set_attention_tail_split(int v) { g_w4_split = v; }

// the scratch of (current device, st); allocates it when `may_alloc` (never inside a stream capture)
static const W4Scratch* w4_scratch(hipStream_t st, bool may_alloc) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_w4_mu);
  for (int i = 0; i < g_w4_nscr; ++i)
    if (g_w4_scr[i].dev == dev && g_w4_scr[i].st == st) return &g_w4_scr[i];
  if (!may_alloc || g_w4_nscr == W4_PART_SLOTS) return nullptr;
  W4Scratch s{dev, st, nullptr, 256};
  if (hipDeviceGetAttribute(&s.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || s.cus < 8) s.cus = 256;
  if (hipMalloc((void**)&s.part, (size_t)W4_PART_TILES * 256 * W4_PROW * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_w4_scr[g_w4_nscr] = s;
  return &g_w4_scr[g_w4_nscr++];
}

int attention_w4_release() {
  std::lock_guard<std::mutex> lk(g_w4_mu);
  if (g_w4_nscr == 0) return 0;
  (void)hipDeviceSynchronize();
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int i = 0; i < g_w4_nscr; ++i) {
    (void)hipSetDevice(g_w4_scr[i].dev);
    (void)hipDeviceSynchronize();
    (void)hipFree(g_w4_scr[i].part);
  }
  (void)hipSetDevice(cur);
  (void)hipGetLastError();
  g_w4_nscr = 0;
  return 0;
}

int attention_w4_prepare(hipStream_t st) {
  if (!g_w4_split) return 0;      // nothing to allocate while the tail split is off (the default)
  return w4_scratch(st, true) ? 0 : fail("attention: cannot allocate the tail-split scratch");
}

// tfx_set_option attention_streamk: 1 (default) = deal (item, key tile) units instead of whole items when the estimate below says it pays and
// the caller passed a workspace (AttnArgs::workspace); 0 = whole items only (a sample's bits then do not depend on the batch size); 2 = always
// when admissible (tests).
static int g_w4_streamk = 1;
void set_attention_streamk(int v) { g_w4_streamk = v; }
static int g_w4_persist = 1;     // tfx_set_option attention_persistent: 0 = one workgroup per (b, h, q-tile) item (round 3)
void set_attention_persistent(int v) { g_w4_persist = v; }
static int g_w4_grid() {         // workgroups of the persistent form: one per CU, a whole number per XCD
  static int grid = 0;
  if (!grid) {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    grid = cus & ~7;
  }
  return grid;
}

#ifdef TFX_BENCH
extern "C" void tfx_bench_attn_timers(unsigned long long* dev) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_timers), &dev, sizeof(dev));
}
#endif

template <int MODE>
static int w4_launch(const AttnArgs& a, hipStream_t st, unsigned grid, int nqb, int nfull, int nparts, int nsplit, int xsplit, float* part,
                     int sk_group, int sk_share) {
  static bool attr_set = false;
  const void* fn = (const void*)attn_w4_kernel<MODE>;
  if (!attr_set) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return fail("attention: no attn_w4_kernel<%d> in this build", MODE);
    (void)hipGetLastError();
    // built without -mllvm -amdgpu-mfma-vgpr-form (see Makefile) the accumulators land in the AccVGPRs and ~500 registers spill
    if (fa.localSizeBytes != 0)
      return fail("attention: attn_w4_kernel<%d> spills %zu bytes per lane -- attention_w4.hip must be compiled with "
                  "-mllvm -amdgpu-mfma-vgpr-form", MODE, (size_t)fa.localSizeBytes);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_W4 + ((W4_ABL & 256) ? 4096 : 0)) != hipSuccess)
      return fail("attention: cannot raise dynamic LDS limit to %d bytes", ATT_LDS_W4);
    attr_set = true;
  }
  // persistent form: one workgroup per CU over all (b, h, q-tile) items, when there are more items than CUs and no tail split
  int T_items = 0;
  if (sk_share > 0) {            // stream-K: one workgroup per CU, T_items = items of one sample, nfull = samples
    T_items = a.H * nqb;
    nfull = a.B;
    grid = (unsigned)g_w4_grid();
  } else if (g_w4_persist && nsplit == 1 && (int)grid > g_w4_grid()) {
    T_items = (int)grid;
    grid = (unsigned)g_w4_grid();
  }
  attn_w4_kernel<MODE><<<grid, 256, ATT_LDS_W4 + ((W4_ABL & 256) ? 4096 : 0), st>>>((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, (bf16_t*)a.o, a.ldq,
                                                       a.ldk, a.ldv, a.ldo, a.q_bstride, a.k_bstride, a.v_bstride, a.o_bstride, a.H,
                                                       a.N, nqb, a.scale * 1.4426950408889634f, nfull, nparts, nsplit, xsplit, part, T_items, sk_group, sk_share);
  return 0;
}

// mode: 0 .. 4 = attn_w4_kernel<MODE> (tfx_set_option attention_waves 30 .. 34; 34 only when AttnArgs::score_bound allows it)
int joint_attention_w4(const AttnArgs& a, hipStream_t st, int mode) {
  const int nqb = (a.N + 255) / 256;
  const int T = a.B * a.H * nqb;
  // Tail split: T workgroups of equal length on C CUs take ceil(T / C) rounds, and when the last one is partly filled the chip
  // idles for the rest of it.  Cut the tiles of that last round -- the last m (head, q-tile) pairs of every sample, B m = the
  // tail rounded up to a multiple of B -- into two key ranges: the ordinary workgroups then fill whole rounds and the halves one
  // short one.  P1024 batch 8: 3456 workgroups = 13.5 rounds -> 3328 ordinary (13 rounds) + 256 halves: 1.861 -> 1.851 ms (a half
  // costs ~0.6 of a whole, and the rounds of a long kernel are not in step any more: the ideal 3.6 % shrinks to 0.5 %); batch 2:
  // 864 = 3.375 rounds -> 768 + 192 halves, 0.508 -> 0.481 ms.  Not
  // taken when a half would be shorter than 24 key tiles (its prologue, Q load and partial store cost more than the round
  // gains), without a full round in front, or when the last round is more than half full (the halves would need two rounds).
  int nfull = T, nsplit = 1, nparts = 0, xsplit = 0;
  float* part = nullptr;
  if (g_w4_split) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = !(hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone);
    (void)hipGetLastError();
    if (const W4Scratch* sc = w4_scratch(st, !capturing)) {
      const int C = sc->cus, tail = T % C, nkv = (a.N + W4_KV - 1) / W4_KV;
      const int m = (tail + a.B - 1) / a.B;
      if (tail && T >= C && m * a.B * 2 <= C + 8 && nkv >= 48 && m < a.H * nqb && m * a.B * 2 <= W4_PART_TILES) {
        xsplit = m; nsplit = 2; nfull = T - m * a.B; nparts = m * a.B * 2; part = sc->part;
      }
    }
  }
  // Stream-K tail (see w4_sk_bound): per sample, the R items of the partly filled last round are dealt as (item, tile) units to the sample's
  // group of C / B CUs.  Estimate in 64-key tile times on the slowest CU: whole items cost ceil(Tp / group) (nkv + F); the dealt form
  // floor(Tp / group) (nkv + F) + share + (share / nkv + 1) F + M -- F = an item's prologue + output (tools/attn_item_timers.py: 6 - 9 tile
  // times), M = the partial stores, the merge pass (10 - 12 us) and the gap in front of it.  A piece count per item <= 8 (attn_w4_merge_sk_kernel) needs share >= nkv / 6.
  int sk_group = 0, sk_share = 0;
  if (g_w4_streamk && g_w4_persist && nsplit == 1 && a.workspace && a.B <= g_w4_grid()) {
    const int C = g_w4_grid(), nkv = (a.N + W4_KV - 1) / W4_KV, Tp = a.H * nqb, group = C / a.B;
    const int full = (Tp / group) * group, R = Tp - full;
    const int share = R ? (R * nkv + group - 1) / group : 0;
    const int64_t need = (int64_t)2 * a.B * group * 256 * W4_PROW * (int64_t)sizeof(float);
    const float F = 7.f, M = 20.f;      // fitted to nine shapes (tools/attn_streamk_time.py, profiles/r06_attn_streamk.log): 0.65 F + M = 25 +- 10
    const float t_whole = (float)(full / group) * ((float)nkv + F);
    const float t_now = t_whole + (float)nkv + F, t_sk = t_whole + (float)share + ((float)share / (float)nkv + 1.f) * F + M;
    if (R > 0 && nkv >= 16 && share * 6 >= nkv && share >= 8 && need <= a.workspace_bytes && ((uintptr_t)a.workspace & 15) == 0 &&
        (g_w4_streamk >= 2 || t_sk < 0.99f * t_now)) {
      sk_group = group; sk_share = share; part = (float*)a.workspace;
    }
  }
  const unsigned grid = nsplit > 1 ? (unsigned)(((nfull + 7) & ~7) + nparts) : (unsigned)T;
#ifdef TFX_BENCH
  const int rc = mode == 4 ? w4_launch<4>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share)
               : mode == 3 ? w4_launch<3>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share)
               : mode == 2 ? w4_launch<2>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share)
               : mode == 1 ? w4_launch<1>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share)
                           : w4_launch<0>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share);
#else   // product library: the reference-free stream and the guarded form; modes 1 .. 3 are A/B builds (round 4), bench library only
  if (mode != 4 && mode != 0) return fail("attention: attn_w4_kernel<%d> is bench-only (libtextflux_hip_bench.so)", mode);
  const int rc = mode == 4 ? w4_launch<4>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share)
                           : w4_launch<0>(a, st, grid, nqb, nfull, nparts, nsplit, xsplit, part, sk_group, sk_share);
#endif
  if (rc) return rc;
  if (sk_share > 0) attention_note_streamk();
  if (sk_share > 0)
    attn_w4_merge_sk_kernel<<<(unsigned)(a.B * sk_group * 8), 256, 0, st>>>(part, (bf16_t*)a.o, a.ldo, a.o_bstride, a.H, a.N, nqb, sk_group,
                                                                             sk_share, a.H * nqb);
  if (nsplit > 1)
    attn_w4_merge_kernel<<<(unsigned)(xsplit * a.B * 32), 256, 0, st>>>(part, (bf16_t*)a.o, a.ldo, a.o_bstride, a.H, a.N, nqb,
                                                                              xsplit, nsplit);
  return 0;
}

}  // namespace tfx
