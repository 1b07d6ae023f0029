// Half-tile software-pipelined joint attention (attention_waves = 20): same math, data layouts and matrix-pipe softmax
// bookkeeping as attn_mx_kernel (attention.hip), different instruction stream.
//
// attn_mx_kernel runs three phases per 64-key tile -- QK^T (MFMA only), softmax (VALU only), PV (MFMA only) -- and its
// time is the SUM of the two pipes' work: the two waves of a SIMD do not overlap one's VALU phase with the other's MFMA
// phase (DESIGN.md "Attention": matrix pipe busy 53 %).  What does overlap is VALU work placed in the shadow of the SAME
// wave's MFMAs (<= 5 single-issue instructions per 32-cycle MFMA, MI355X_MICROARCH.md).  This kernel therefore pipelines
// at the granularity of a 32-key half-tile so that every MFMA block has independent VALU work of the neighbouring
// half-tile beside it, with NO extra registers (the two 32-key score blocks s0 / s1 of the tile kernel become the
// "current" and "next" stage of the pipeline):
//
//   MFMA stream of step h :  S(h+1) = K(h+1) Q^T - m_ref (1 offset MFMA + 8)  alternating with  O^T += V(h-1)^T P(h-1)^T
//                            (8 MFMAs) -- consecutive MFMAs never share an accumulator, so the VALU
//                            fillers between them do not break an accumulate chain's forwarding
//   VALU stream of step h :  row maximum of S(h) relative to the reference (v_max3, one cross-half shuffle, vote)
//                            [rare] move the reference: O, l, the pending P(h-1) registers are scaled, S(h) and the
//                            partial S(h+1) shifted -- every contribution accumulated against the old reference is
//                            rescaled exactly once
//                            P(h) = bf16(exp2(S(h)))  (16 v_exp, 8 v_cvt_pk)
//
// QK(h+1) of an odd h reads the NEXT tile's K rows and the pending P.V of an even h reads the PREVIOUS tile's V rows, so
// tiles j - 1, j, j + 1 are live in iteration j while tile j + 2 is being written: a ring of four LDS buffers (128 KiB);
// one barrier per 64-key tile as before.
#include <type_traits>

#include "common.h"
#include "launch.h"

namespace tfx {

namespace {
constexpr int HP_KV = 64, HP_HD = 128;
constexpr int HP_TILE = HP_KV * HP_HD * 2;      // 16 KiB per K (or V) tile
constexpr float HP_THR = 4.0f;                   // lazy-reference threshold (log2 units), as attn_mx_kernel
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
}  // namespace

constexpr int ATT_LDS_HP = 4 * 2 * HP_TILE;      // K, V x 4 buffers = 128 KiB

__global__ __launch_bounds__(512, 2) void attn_hp_kernel(const bf16_t* Q, const bf16_t* __restrict__ Kp,
                                                         const bf16_t* __restrict__ Vp, bf16_t* O, int64_t ldq, int64_t ldk,
                                                         int64_t ldv, int64_t ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                                                         int64_t o_bs, int H, int N, int nqb, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int qb = bid % nqb;
  bid /= nqb;
  const int h = bid % H;
  const int b = bid / H;
  const bf16_t* Qb = Q + b * q_bs + h * HP_HD;
  const bf16_t* Kb = Kp + b * k_bs + h * HP_HD;
  const bf16_t* Vb = Vp + b * v_bs + h * HP_HD;
  bf16_t* Ob = O + b * o_bs + h * HP_HD;

  // ---- Q fragments, pre-scaled into the exp2 domain (one extra bf16 rounding of q, as attn_mx_kernel)
  const int qrow = qb * 256 + wave * 32 + l31;
  const int qrow_c = qrow < N ? qrow : N - 1;
  bf16x8 qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = *reinterpret_cast<const bf16x8*>(Qb + (int64_t)qrow_c * ldq + s * 16 + hi * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qf[s][e] * scale_log2e);
  }

  const int nkv = (N + HP_KV - 1) / HP_KV;
  // ---- staging: thread t moves 16-byte chunks t and t + 512 of a tile's K and V (global -> registers -> LDS).  Buffer
  // loads: per-thread 32-bit byte offsets inside a tile + the tile's origin in the scalar offset (no 64-bit per-lane
  // pointers live across the loop).
  u32x4 kreg[2], vreg[2];
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)0xffffffffu, 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (int)0xffffffffu, 0x00020000);
  auto load_tile = [&](int j) __attribute__((always_inline)) {
    int te = tid;
    asm volatile("" : "+v"(te));     // offsets rebuilt per tile: as loop invariants they would pin 4 more VGPRs
    const int jj = j < nkv ? j : nkv - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = te + i * 512;
      int key = c >> 4;
      if (jj == nkv - 1) key = min(key, N - 1 - jj * HP_KV);      // last (possibly ragged) tile: clamped rows
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, key * (int)ldk * 2 + (c & 15) * 16,
                                                                                  jj * HP_KV * (int)ldk * 2, 0));
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, key * (int)ldv * 2 + (c & 15) * 16,
                                                                                  jj * HP_KV * (int)ldv * 2, 0));
    }
  };
  auto write_tile = [&](int buf) __attribute__((always_inline)) {
    char* kd = smem + buf * 2 * HP_TILE;
    char* vd = kd + HP_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 512;
      const int key = c >> 4, ch = c & 15;
      *reinterpret_cast<u32x4*>(kd + key * 256 + ((ch ^ (key & 15)) << 4)) = kreg[i];
      *reinterpret_cast<u32x4*>(vd + key * 256 + ((((ch >> 2) ^ (key & 3)) << 6) | ((ch & 3) << 4))) = vreg[i];
    }
  };
  uint32_t koff[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) koff[s] = l31 * 256 + (((2 * s + hi) ^ (lane & 15)) << 4);
  const int vi = lane & 15;
  const uint32_t vrow = (4 * hi + (vi >> 2)) * 256 + 32 * ((lane >> 4) & 1) + (vi & 3) * 8;
  uint32_t voff[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) voff[db] = vrow + ((db ^ ((vi >> 2) & 3)) << 6);

  f32x16 o[4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db][r] = 0.f;
  }
  float l_run = 0.f;   // row sum of this lane's keys (fp32 weights, summed on the VALU in the MFMAs' shadow)
  bf16x8 kone, qm;
#pragma unroll
  for (int e = 0; e < 8; ++e) { kone[e] = (__bf16)0.f; qm[e] = (__bf16)0.f; }
  if (hi == 0) kone[0] = (__bf16)1.0f;
  float m_ref = 0.f;

  // ---- prologue: tiles 0 and 1 in LDS, tile 2 requested
  load_tile(0);
  write_tile(0);
  load_tile(1);
  write_tile(1);
  load_tile(2);
  __syncthreads();

  // scores of key block (32 keys) `blk` (0 / 1) of the tile at kt, against the CURRENT reference (prologue only)
  auto qk = [&](const char* kt, int blk, f32x16& s) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qm, s, 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt + koff[st] + blk * 32 * 256);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s, 0, 0, 0);
    }
  };
  auto mask_ragged = [&](int j, int blk, f32x16& s) __attribute__((always_inline)) {
    if (j == nkv - 1 && (N & (HP_KV - 1))) {
      const int kbase = j * HP_KV + blk * 32 + 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kbase + (r & 3) + 8 * (r >> 2) >= N) s[r] = -INFINITY;
    }
  };
  auto vfrag = [&](const char* vt, int ks, int db) __attribute__((always_inline)) -> bf16x8 {
    const char* va = vt + voff[db] + ks * 16 * 256;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(va + 8 * 256));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  bf16x8 pf[2];   // P: the pending half-tile's weights are consumed by its P.V MFMAs just before the next half-tile's overwrite them

  // One pipeline phase = one half-tile step h.  MFMA stream: the 9 MFMAs of S(h+1) = K(h+1) Q^T - m_ref alternate with the
  // 8 MFMAs of the pending O^T += V(h-1)^T P(h-1)^T (never two consecutive MFMAs on one accumulator with VALU between
  // them); VALU stream in their shadow: row maximum of S(h) -> [rare: move the reference] -> P(h) = bf16(exp2(S(h))).
  //   cur / nxt   scores S(h) (finished here) / S(h+1) (produced here)
  //   pf / pf  P(h-1) (consumed by the pending P.V) / P(h) (produced)
  //   vt_prev, ks0  V tile and first k-step of the pending P.V;  kt_next, blk_next, j_next  K rows of S(h+1)
  auto phase = [&](f32x16& cur, f32x16& nxt, const char* vt_prev, int ks0, const char* kt_next, int blk_next,
                   int j_next) __attribute__((always_inline)) {
    const char* kb = kt_next + blk_next * 32 * 256;
    auto kfrag = [&](int st) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(kb + koff[st]); };
    // P.V MFMA i of 10: (kk = i / 5; db = i % 5, db == 4 is the row-sum block of ones)
    bf16x8 kf0 = kfrag(0), kf1 = kfrag(1), vf0 = vfrag(vt_prev, ks0, 0), vf1 = vfrag(vt_prev, ks0, 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) nxt[r] = 0.f;
    float mx = cur[0];
    // ---- block 1: 3 + 3 MFMAs beside the row maximum
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kone, qm, nxt, 0, 0, 0);
    bf16x8 vf2 = vfrag(vt_prev, ks0, 2);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0, pf[0], o[0], 0, 0, 0);
    mx = fmaxf(fmaxf(mx, cur[1]), cur[2]);
    mx = fmaxf(fmaxf(mx, cur[3]), cur[4]);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kf2 = kfrag(2);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0, qf[0], nxt, 0, 0, 0);
    bf16x8 vf3 = vfrag(vt_prev, ks0, 3);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1, pf[0], o[1], 0, 0, 0);
    mx = fmaxf(fmaxf(mx, cur[5]), cur[6]);
    mx = fmaxf(fmaxf(mx, cur[7]), cur[8]);
    mx = fmaxf(fmaxf(mx, cur[9]), cur[10]);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kf3 = kfrag(3);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1, qf[1], nxt, 0, 0, 0);
    o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf2, pf[0], o[2], 0, 0, 0);
    mx = fmaxf(fmaxf(mx, cur[11]), cur[12]);
    mx = fmaxf(fmaxf(mx, cur[13]), cur[14]);
    mx = fmaxf(mx, cur[15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    __builtin_amdgcn_sched_barrier(0);
    if (!__all(mx <= HP_THR)) {
      // move the reference.  At this point O / l hold every finished P.V plus the first three MFMAs of the pending one,
      // whose remaining MFMAs follow below: O, l AND the pending P registers are scaled by f (each contribution exactly
      // once); the current scores and the partial S(h+1) (its offset step used the old reference) are shifted.
      const float m_new = round_bf(m_ref + fmaxf(mx, 0.f));
      const float d = m_new - m_ref;
      const float f = __builtin_amdgcn_exp2f(-d);
      m_ref = m_new;
      qm[0] = (__bf16)(hi == 0 ? -m_new : 0.f);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cur[r] -= d;
        nxt[r] -= d;
#pragma unroll
        for (int db = 0; db < 4; ++db) o[db][r] *= f;
      }
      l_run *= f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) pf[kk][e] = (__bf16)((float)pf[kk][e] * f);
    }
    // ---- block 2: 6 + 7 MFMAs beside the exponentials (2 v_exp + 1 v_cvt_pk per pair of scores)
#define HP_EXP2(i)                                                         \
    cur[2 * (i)] = __builtin_amdgcn_exp2f(cur[2 * (i)]);                   \
    cur[2 * (i) + 1] = __builtin_amdgcn_exp2f(cur[2 * (i) + 1]);           \
    l_run += cur[2 * (i)] + cur[2 * (i) + 1];                              \
    pf[(i) >> 2][((i) & 3) * 2] = (__bf16)cur[2 * (i)];                  \
    pf[(i) >> 2][((i) & 3) * 2 + 1] = (__bf16)cur[2 * (i) + 1];
    bf16x8 kf4 = kfrag(4);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf2, qf[2], nxt, 0, 0, 0);
    bf16x8 vf4 = vfrag(vt_prev, ks0 + 1, 0);
    o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf3, pf[0], o[3], 0, 0, 0);
    HP_EXP2(0)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kf5 = kfrag(5);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf3, qf[3], nxt, 0, 0, 0);
    bf16x8 vf5 = vfrag(vt_prev, ks0 + 1, 1);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf4, pf[1], o[0], 0, 0, 0);
    HP_EXP2(1)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kf6 = kfrag(6);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf4, qf[4], nxt, 0, 0, 0);
    bf16x8 vf6 = vfrag(vt_prev, ks0 + 1, 2);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf5, pf[1], o[1], 0, 0, 0);
    HP_EXP2(2)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 kf7 = kfrag(7);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf5, qf[5], nxt, 0, 0, 0);
    bf16x8 vf7 = vfrag(vt_prev, ks0 + 1, 3);
    o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf6, pf[1], o[2], 0, 0, 0);
    HP_EXP2(3)
    __builtin_amdgcn_sched_barrier(0);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf6, qf[6], nxt, 0, 0, 0);
    o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf7, pf[1], o[3], 0, 0, 0);   // last reader of the pending pf[1]
    __builtin_amdgcn_sched_barrier(0);
    HP_EXP2(4)
    __builtin_amdgcn_sched_barrier(0);
    nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf7, qf[7], nxt, 0, 0, 0);
    HP_EXP2(5)
    __builtin_amdgcn_sched_barrier(0);
    HP_EXP2(6)
    HP_EXP2(7)
    __builtin_amdgcn_sched_barrier(0);
#undef HP_EXP2
    mask_ragged(j_next, blk_next, nxt);
  };

  f32x16 sa, sb;          // pipeline stages: scores of the current / the next half-tile
  // ---- tile 0 (peeled: its first half-tile has no pending P.V and pins the reference to the true row maximum)
  {
    if (nkv > 2) write_tile(2);
    if (nkv > 3) load_tile(3);
    qk(smem, 0, sa);
    mask_ragged(0, 0, sa);
    float mx = sa[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sa[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    m_ref = round_bf(mx);
    qm[0] = (__bf16)(hi == 0 ? -m_ref : 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = __builtin_amdgcn_exp2f(sa[r] - m_ref); l_run += sa[r]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { pf[0][e] = (__bf16)sa[e]; pf[1][e] = (__bf16)sa[8 + e]; }
    qk(smem, 1, sb);
    mask_ragged(0, 1, sb);
    phase(sb, sa, smem + HP_TILE, 0, smem + 2 * HP_TILE, 0, 1);   // S(2) from tile 1 (stale rows if nkv == 1: unused)
    __syncthreads();
  }
  // one 64-key tile j >= 1 out of ring buffer BUF (compile-time: fragment addresses are a per-lane base plus an immediate)
  auto tile = [&](int j, auto BUF) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
    constexpr int nbuf = (buf + 1) % 4, wbuf = (buf + 2) % 4, pbuf = (buf + 3) % 4;   // next / staged-into / previous tile
    // staging: tile j + 2 (requested one iteration ago) goes into the buffer that held tile j - 2 (last read before the
    // previous barrier); tile j + 1, needed by this iteration's second phase, was written one iteration ago
    if (j + 2 < nkv) write_tile(wbuf);
    if (j + 3 < nkv) load_tile(j + 3);
    const char* kt = smem + buf * 2 * HP_TILE;
    const char* vt = kt + HP_TILE;
    const char* kn = smem + nbuf * 2 * HP_TILE;
    const char* vprev = smem + pbuf * 2 * HP_TILE + HP_TILE;
    phase(sa, sb, vprev, 2, kt, 1, j);        // finish S(2j); pending = 2nd half of tile j - 1; produce S(2j + 1)
    phase(sb, sa, vt, 0, kn, 0, j + 1);       // finish S(2j + 1); pending = 1st half of tile j; produce S(2j + 2)
    __syncthreads();
  };
  for (int j = 1; j < nkv; j += 4) {
    tile(j, std::integral_constant<int, 1>{});
    if (j + 1 < nkv) tile(j + 1, std::integral_constant<int, 2>{});
    if (j + 2 < nkv) tile(j + 2, std::integral_constant<int, 3>{});
    if (j + 3 < nkv) tile(j + 3, std::integral_constant<int, 0>{});
  }
  // ---- drain: P.V of the very last half-tile (second half of tile nkv - 1)
  {
    const char* vt = smem + ((nkv - 1) & 3) * 2 * HP_TILE + HP_TILE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(vt, 2 + kk, db), pf[kk], o[db], 0, 0, 0);
    }
  }

  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));   // this lane's keys + the partner half's
  if (qrow < N) {
    bf16_t* orow = Ob + (int64_t)qrow * ldo + 4 * hi;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        u32x2 w;
        w[0] = pack_bf2(o[db][qd * 4 + 0] * inv, o[db][qd * 4 + 1] * inv);
        w[1] = pack_bf2(o[db][qd * 4 + 2] * inv, o[db][qd * 4 + 3] * inv);
        *reinterpret_cast<u32x2*>(orow + db * 32 + qd * 8) = w;
      }
  }
}

int joint_attention_hp(const AttnArgs& a, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)attn_hp_kernel);
    (void)hipGetLastError();
    if (hipFuncSetAttribute((const void*)attn_hp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_LDS_HP) != hipSuccess)
      return fail("attention: cannot raise dynamic LDS limit to %d bytes", ATT_LDS_HP);
    attr_set = true;
  }
  const int nqb = (a.N + 255) / 256;
  const unsigned grid = (unsigned)(a.B * a.H * nqb);
  attn_hp_kernel<<<grid, 512, ATT_LDS_HP, st>>>((const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, (bf16_t*)a.o, a.ldq,
                                                 a.ldk, a.ldv, a.ldo, a.q_bstride, a.k_bstride, a.v_bstride, a.o_bstride, a.H,
                                                 a.N, nqb, a.scale * 1.4426950408889634f);
  return 0;
}

}  // namespace tfx
