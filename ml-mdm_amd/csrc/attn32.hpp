// Attention on v_mfma_f32_32x32x16_bf16 (included by attention.hip; reference ml_mdm/models/unet.py:276-313): the backward
// kernels of the product path, and a forward on the same tiles that does not beat attn_fwd_kernel in the train step and sits
// behind a development switch (see the comment in front of f_step64 and attention.hip's attn_fwd_launch).
//
// Why another set of kernels.  The 16x16x32 kernels of attention.hip give a wave 16 keys (or queries): at the 16x16 level
// of the U-Net (L = 256, d = 96; 26 of 31 attention layers) the one-block-per-head backward walks the query tiles TWICE
// for its 256 keys and the key tiles twice for its 256 queries, every MFMA reads a full 1 KB operand fragment from LDS for
// 16 K FLOP, and the 128-byte-row tile image makes the transpose reads 2-way bank conflicted -- 160-190 TF (6-8 % of the
// matrix peak).  Here a wave owns 32 rows:
//   * 8 waves x 32 = all 256 keys (queries) of the level in ONE pass over the resident operand tiles;
//   * a 32x32x16 MFMA does 32 K FLOP per 1 KB LDS fragment -- half the LDS bytes per FLOP;
//   * the accumulator layout of S = Q K^T (lane <-> key, register r <-> query 8 (r >> 2) + 4 (lane >> 5) + (r & 3)) makes
//     registers 8 s .. 8 s + 7 of P / dS, packed to bf16, the operand of reduction step s of dV^T += dO^T P, dK^T += Q^T dS
//     as they are (no cross-lane movement); the matching operand is two ds_read_b64_tr_b16 per lane from the NATURAL
//     [row][d] image (rows 16 s + 4 hi + i and 16 s + 8 + 4 hi + i).  Same for dQ^T += K^T dS^T with lane <-> query;
//   * lse and delta enter as the INITIAL accumulators (C operand) of the S and dP products: S' = Q K^T - lse / scale,
//     dP' = dO V^T - delta, so P = 2^(c2 S') and dS = P dP' -- one multiply and one v_exp per score, no subtracts;
//   * the tile image is a plain [row][d] array (pitch 2 d bytes) whose 16-byte chunk index is XOR-ed with a function of the
//     row chosen so that BOTH access patterns are bank-conflict free: ds_read_b128 of one chunk column by 16 rows of all
//     residues mod 16, and the transpose read's 4 rows x 64 bytes per 32 lanes (A32::swz).
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (D = A B + C; cdna guide section 3, and csrc/gemm_x.hpp which runs on them):
//   A: lane l holds A[row = l & 31][k = 8 (l >> 5) + j], j = 0..7      B: lane l holds B[k = 8 (l >> 5) + j][col = l & 31]
//   C / D: lane l, register r holds D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
#pragma once

namespace mdm {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// natural [row][D] bf16 image with swizzled 16-byte chunks
template <int D> struct A32 {
  static_assert(D == 64 || D == 96, "A32: head dims 64 and 96");
  static constexpr int PITCH = 2 * D;      // bytes per row
  static constexpr int CPR = D / 8;        // 16-byte chunks per row
  static constexpr int KS = D / 16;        // reduction steps of a product over d
  static constexpr int NB = D / 32;        // 32-wide blocks across d
  // chunk XOR of a row.  Conflict-free by construction for (a) 16 rows of all residues mod 16 reading the same logical
  // chunk (ds_read_b128 lane groups), (b) rows 4 q .. 4 q + 3 x logical chunks 4 m .. 4 m + 3 (one half of a transpose read):
  //   D = 96 (pitch 192 B: row r starts at 16-byte slot 12 r mod 16 = 4 (-r & 3)): XOR the low two chunk bits with
  //           (r >> 2) & 3 -- rows of equal r & 3 differ there, rows of different r & 3 start in different 64-byte quarters;
  //   D = 64 (pitch 128 B: two rows per 256 bytes): f = bit 1 of r -> chunk bit 2, bits 2-3 of r -> chunk bits 0-1.
  static __device__ __forceinline__ int swz(int row) {
    return D == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
  }
  static __device__ __forceinline__ int off(int row, int chunk) { return row * PITCH + ((chunk ^ swz(row)) << 4); }
};

// Per-lane LDS byte offsets of the two fragment kinds inside an image, relative to a 32-row tile (tile t adds
// 32 t PITCH: the swizzle only looks at row bits 1-3).
//   b[s]      ds_read_b128 of row (lane & 31), elements 16 s + 8 hi .. + 7          (A or B operand of a product over d)
//   tr[t][blk] ds_read_b64_tr_b16: rows 8 t + 4 hi + (i >> 2), columns 32 blk + 16 ((lane >> 4) & 1) + 4 (i & 3), i = lane & 15;
//              reduction step s2 adds 16 s2 PITCH.  The lane receives column 32 blk + (lane & 31) of those four rows.
template <int D> struct Frag32Off {
  using G = A32<D>;
  unsigned b[G::KS];
  unsigned tr[2][G::NB];
  __device__ __forceinline__ Frag32Off(int lane) {
    const int n = lane & 31, hi = lane >> 5, i = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) b[s] = (unsigned)G::off(n, 2 * s + hi);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int blk = 0; blk < G::NB; ++blk) {
        const int row = 8 * t + 4 * hi + (i >> 2);
        const int col = 32 * blk + 16 * cb + 4 * (i & 3);
        tr[t][blk] = (unsigned)(G::off(row, col >> 3) + ((col >> 2) & 1) * 8);
      }
  }
};

__device__ __forceinline__ bf16x8 lds_b128(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 lds_tr_pair(const char* p0, const char* p1) {
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p1));
  s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return *reinterpret_cast<bf16x8*>(&v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s2) {
  return bf16x8{(bf16)v[8 * s2 + 0], (bf16)v[8 * s2 + 1], (bf16)v[8 * s2 + 2], (bf16)v[8 * s2 + 3],
                (bf16)v[8 * s2 + 4], (bf16)v[8 * s2 + 5], (bf16)v[8 * s2 + 6], (bf16)v[8 * s2 + 7]};
}
__device__ __forceinline__ f32x16 splat16(float v) {
  return f32x16{v, v, v, v, v, v, v, v, v, v, v, v, v, v, v, v};
}
__device__ __forceinline__ f32x16 mma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// "These fragments have arrived": an empty asm that reads them.  Placed in front of a loop whose body also has global loads in
// flight, it keeps hipcc from waiting for THOSE inside the loop (its wait-count pass merges "the fragments loaded before the
// loop may still be pending" into every iteration and then covers them with vmcnt(n) waits that, from the second iteration
// on, can only be waiting for the loop's own prefetch).
template <int N> __device__ __forceinline__ void frags_arrived(const bf16x8 (&f)[N]) {
#pragma unroll
  for (int s = 0; s < N; ++s) asm volatile("" ::"v"(f[s]));
}

// rows [0, nrows) of a strided head view behind a buffer descriptor: rows past the end read as zeros
struct RowSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned row_bytes;
  __device__ __forceinline__ RowSrc(const bf16* src, int rs, int nrows) {
    row_bytes = (unsigned)rs * 2u;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(src), 0, (unsigned)nrows * row_bytes, 0x00020000);
  }
  __device__ __forceinline__ uint4 chunk(int row, int byte_in_row) const {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)row * row_bytes + (unsigned)byte_in_row, 0, 0);
    return uint4{(unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]};
  }
  // the operand fragment of row `row`: elements 16 s + 8 hi .. + 7
  __device__ __forceinline__ bf16x8 frag(int row, int s, int hi) const {
    uint4 v = chunk(row, (16 * s + 8 * hi) * 2);
    return *reinterpret_cast<bf16x8*>(&v);
  }
};

// ---- the two tile steps -------------------------------------------------------------------------------------------------
// Left to itself hipcc issues the LDS reads of a step two at a time, right in front of the MFMA pair that consumes them: a
// wave then pays one LDS round trip per MFMA pair, then runs the softmax with nothing in flight, then pays the round trips
// of the transpose reads -- measured 28 % of the matrix peak at L = 1024 with the arithmetic of a step at 12-24 MFMAs.  The
// steps below are written as BATCHES separated by scheduling fences (sched_barrier): every read of a batch is issued
// before the batch's consumers, and the batches are placed so that a read's latency lies under independent work:
//     S / dP products (operands read one step earlier)  |  transpose reads of this tile issued  |  softmax (VALU)
//     |  the NEXT tile's S / dP operands issued  |  dQ (dV, dK) products.
// LDS returns in order, so the waits hipcc inserts are counted (lgkmcnt(n)), not drains.
#define MDM_FENCE() __builtin_amdgcn_sched_barrier(0)

// Issue priority of a wave for one tile step.  The two waves of a SIMD (waves w and w + 4 of a block) are arbitrated by
// priority, then age: left alone, the second-dispatched half runs its tile loops 40 % longer (measured: 17.5 K against 12.3 K
// cycles for the nine tiles of phase Q) and every barrier waits for it; a static s_setprio for that half only swaps the roles
// (12.1 K / 17.2 K).  Alternating the priority tile by tile lets the pair take turns, so both halves finish together.
#ifndef MDM_ATT_SR
#define MDM_ATT_SR 64
#endif
#ifndef MDM_ATT_PRIO
#define MDM_ATT_PRIO 0
#endif
// timing-only ablations of the streaming kernels (development: tools/build_variant.sh -DMDM_ATT_ABL=<bits>; results are wrong):
// 1 no v_exp, 2 no transpose reads, 4 no barrier in the stage loops, 8 no fetch / commit in the stage loops, 16 no S / dP
// products, 32 no dQ / dV / dK products
#ifndef MDM_ATT_ABL
#define MDM_ATT_ABL 0
#endif

#define MDM_EXP2(x) ((MDM_ATT_ABL & 1) ? (x) : __builtin_amdgcn_exp2f(x))
__device__ __forceinline__ void prio_flip(int step_plus_half) {
#if MDM_ATT_PRIO == 2
  if (__builtin_amdgcn_readfirstlane(step_plus_half) & 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}
#ifndef MDM_QPF96
#define MDM_QPF96 3
#endif
#ifndef MDM_KPF96
#define MDM_KPF96 3
#endif

// A operands (K, V rows: phase Q; Q, dO rows: phase K) of the S / dP products of one tile, read one step ahead
template <int D, int PF> struct Ops32 { bf16x8 x[PF], y[PF]; };
template <int D, int PF>
__device__ __forceinline__ void load_ops32(Ops32<D, PF>& o, const char* Xt, const char* Yt, const Frag32Off<D>& fo) {
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    o.x[s] = lds_b128(Xt + fo.b[s]);
    o.y[s] = lds_b128(Yt + fo.b[s]);
  }
}

// One 32-key tile for a wave that owns 32 queries (lane <-> query).  Kt: the tile's K rows in a swizzled image; o: its K / V
// operand fragments (read during the previous step); Kn / Vn: the NEXT tile's rows (any valid tile if there is none); qf /
// gf: the wave's Q / dO operand fragments; nl = -lse / scale, nd = -delta of the lane's query; `live`: bit k = key k of the
// tile takes part (wave-uniform); dq: dQ^T accumulators (register r of block blk <-> channel 32 blk + 8 (r >> 2) + 4 hi + (r & 3)).
template <int D> struct QPf { static constexpr int value = D <= 64 ? A32<D>::KS : MDM_QPF96; };
template <int D>
__device__ __forceinline__ void q_step32(Ops32<D, QPf<D>::value>& o, const char* Kt, const char* Vt, const char* Kn, const char* Vn,
                                         const bf16x8 (&qf)[A32<D>::KS], const bf16x8 (&gf)[A32<D>::KS],
                                         const float nl, const float nd, const unsigned live, const float c2, const int hi,
                                         const Frag32Off<D>& fo, f32x16 (&dq)[A32<D>::NB]) {
  using G = A32<D>;
  constexpr int PF = QPf<D>::value;
  bf16x8 xr[G::KS - PF + 1], yr[G::KS - PF + 1];   // the reduction steps not read ahead (+ 1: no zero-length arrays)
#pragma unroll
  for (int s = PF; s < G::KS; ++s) {
    xr[s - PF] = lds_b128(Kt + fo.b[s]);
    yr[s - PF] = lds_b128(Vt + fo.b[s]);
  }
  // (lse and delta are per-lane scalars here: as initial accumulators they would cost 32 v_mov per tile, so the products start
  // from the inline constant 0 and the scalars enter in the softmax: 2^(c2 S + c2 nl), P (dP + nd))
  const f32x16 zero = splat16(0.f);
  f32x16 sc, dp;
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    if (MDM_ATT_ABL & 16) { sc = zero; dp = zero; continue; }
    sc = mma32(o.x[s], qf[s], s == 0 ? zero : sc);
    dp = mma32(o.y[s], gf[s], s == 0 ? zero : dp);
  }
#pragma unroll
  for (int s = PF; s < G::KS; ++s) {
    if (MDM_ATT_ABL & 16) continue;
    sc = mma32(xr[s - PF], qf[s], sc);
    dp = mma32(yr[s - PF], gf[s], dp);
  }
  MDM_FENCE();
  bf16x8 tk[2][G::NB];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int blk = 0; blk < G::NB; ++blk)
      tk[s2][blk] = (MDM_ATT_ABL & 2) ? o.x[blk] : lds_tr_pair(Kt + s2 * 16 * G::PITCH + fo.tr[0][blk], Kt + s2 * 16 * G::PITCH + fo.tr[1][blk]);
  MDM_FENCE();
  const float nl2 = nl * c2;
  if (live == 0xffffffffu) {
    // two scores per instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32; only v_exp_f32 is per element): the loops of
    // these kernels sit on their VALU floor -- with every MFMA, barrier and fetch removed the L = 1024 pair still takes 38 %
    // of its time (profiles/r05_attn32_ablations.txt)
    const f32x2 c2v = {c2, c2}, nlv = {nl2, nl2}, ndv = {nd, nd};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 e = f32x2{sc[r], sc[r + 1]} * c2v + nlv;
      const f32x2 pr = {MDM_EXP2(e[0]), MDM_EXP2(e[1])};
      const f32x2 ds = pr * (f32x2{dp[r], dp[r + 1]} + ndv);
      sc[r] = ds[0]; sc[r + 1] = ds[1];
    }
  } else {
    const unsigned lv = live >> (4 * hi);    // this lane's registers hold keys (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = ((lv >> ((r & 3) + 8 * (r >> 2))) & 1u) ? MDM_EXP2(fmaf(sc[r], c2, nl2)) : 0.f;
      sc[r] = pr * (dp[r] + nd);
    }
  }
  const bf16x8 ds0 = pack8(sc, 0), ds1 = pack8(sc, 1);
  MDM_FENCE();
  load_ops32<D, PF>(o, Kn, Vn, fo);
  MDM_FENCE();
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dq[blk] = mma32(tk[0][blk], ds0, dq[blk]); else dq[blk][0] += (float)ds0[0] + (float)tk[0][blk][0];
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dq[blk] = mma32(tk[1][blk], ds1, dq[blk]); else dq[blk][1] += (float)ds1[0] + (float)tk[1][blk][0];
}

// One 32-query tile for a wave that owns 32 keys (lane <-> key).  Qt / Gt: the tile's rows of Q / dO in swizzled images; o:
// the first PF reduction steps of their operand fragments (read during the previous step); Qn / Gn: the NEXT tile (any valid
// tile if none); nl / nd: -lse / scale and -delta of the tile's 32 queries (LDS) -- they ARE the initial accumulators;
// kf / vf: the wave's K / V operand fragments; dk / dv: dK^T / dV^T accumulators (register <-> channel as in q_step32).
// PF: how many of the KS reduction steps are read one tile ahead (registers: all of them at d = 64, half at d = 96).
template <int D> struct KPf { static constexpr int value = D <= 64 ? A32<D>::KS : MDM_KPF96; };
template <int D>
__device__ __forceinline__ void k_step32(Ops32<D, KPf<D>::value>& o, const char* Qt, const char* Gt, const char* Qn, const char* Gn,
                                         const float* nl, const float* nd,
                                         const bf16x8 (&kf)[A32<D>::KS], const bf16x8 (&vf)[A32<D>::KS], const bool key_live,
                                         const float c2, const int hi, const Frag32Off<D>& fo,
                                         f32x16 (&dk)[A32<D>::NB], f32x16 (&dv)[A32<D>::NB]) {
  using G = A32<D>;
  constexpr int PF = KPf<D>::value;
  f32x16 sc, dp;
  bf16x8 xr[G::KS - PF + 1], yr[G::KS - PF + 1];   // (+ 1: no zero-length arrays)
#pragma unroll
  for (int s = PF; s < G::KS; ++s) {
    xr[s - PF] = lds_b128(Qt + fo.b[s]);
    yr[s - PF] = lds_b128(Gt + fo.b[s]);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {              // register 4 g + e <-> query 8 g + 4 hi + e
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(nl + 8 * g + 4 * hi);
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(nd + 8 * g + 4 * hi);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[4 * g + e] = l4[e]; dp[4 * g + e] = d4[e]; }
  }
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    if (MDM_ATT_ABL & 16) continue;
    sc = mma32(o.x[s], kf[s], sc);
    dp = mma32(o.y[s], vf[s], dp);
  }
#pragma unroll
  for (int s = PF; s < G::KS; ++s) {
    if (MDM_ATT_ABL & 16) continue;
    sc = mma32(xr[s - PF], kf[s], sc);
    dp = mma32(yr[s - PF], vf[s], dp);
  }
  MDM_FENCE();
  bf16x8 tg[2][G::NB];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int blk = 0; blk < G::NB; ++blk)
      tg[s2][blk] = (MDM_ATT_ABL & 2) ? o.y[blk] : lds_tr_pair(Gt + s2 * 16 * G::PITCH + fo.tr[0][blk], Gt + s2 * 16 * G::PITCH + fo.tr[1][blk]);
  MDM_FENCE();
  if (__builtin_amdgcn_readfirstlane(__all(key_live))) {   // (the common case: no select per score; packed as in q_step32)
    const f32x2 c2v = {c2, c2};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 e = f32x2{sc[r], sc[r + 1]} * c2v;
      const f32x2 pr = {MDM_EXP2(e[0]), MDM_EXP2(e[1])};
      const f32x2 ds = pr * f32x2{dp[r], dp[r + 1]};
      sc[r] = pr[0]; sc[r + 1] = pr[1];
      dp[r] = ds[0]; dp[r + 1] = ds[1];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = key_live ? MDM_EXP2(sc[r] * c2) : 0.f;
      sc[r] = pr;
      dp[r] = pr * dp[r];
    }
  }
  const bf16x8 p0 = pack8(sc, 0), p1 = pack8(sc, 1), s0 = pack8(dp, 0), s1 = pack8(dp, 1);
  MDM_FENCE();
  bf16x8 tq[2][G::NB];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int blk = 0; blk < G::NB; ++blk)
      tq[s2][blk] = (MDM_ATT_ABL & 2) ? kf[blk] : lds_tr_pair(Qt + s2 * 16 * G::PITCH + fo.tr[0][blk], Qt + s2 * 16 * G::PITCH + fo.tr[1][blk]);
  MDM_FENCE();
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dv[blk] = mma32(tg[0][blk], p0, dv[blk]); else dv[blk][0] += (float)p0[0] + (float)tg[0][blk][0];
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dv[blk] = mma32(tg[1][blk], p1, dv[blk]); else dv[blk][1] += (float)p1[0] + (float)tg[1][blk][0];
  MDM_FENCE();
  load_ops32<D, PF>(o, Qn, Gn, fo);
  MDM_FENCE();
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dk[blk] = mma32(tq[0][blk], s0, dk[blk]); else dk[blk][0] += (float)s0[0] + (float)tq[0][blk][0];
#pragma unroll
  for (int blk = 0; blk < G::NB; ++blk) if (!(MDM_ATT_ABL & 32)) dk[blk] = mma32(tq[1][blk], s1, dk[blk]); else dk[blk][1] += (float)s1[0] + (float)tq[1][blk][0];
}

// a wave's [d][32] accumulator block (lane <-> row of the tensor, register <-> channel) -> bf16 rows in global memory.
// Lane (n, hi) holds channels 8 g + 4 hi + 0..3 of row n (g = r >> 2): 8-byte pieces.  One v_permlane32_swap per packed
// register pairs the pieces of the two half-waves (hi = 0 takes channels 16 j .. 16 j + 7 of its row, hi = 1 channels
// 16 j + 8 .. + 15), so a row leaves as 16-byte stores -- half the store instructions (the store tail of these kernels is
// issue-bound, not bandwidth-bound).
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi_) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  bf16x2_t v = {(bf16)lo, (bf16)hi_};
  return *reinterpret_cast<unsigned*>(&v);
}
template <int D>
__device__ __forceinline__ void store_rows32(bf16* row_ptr, const f32x16 (&acc)[A32<D>::NB], const float mul, const int hi, const bool ok) {
#pragma unroll
  for (int blk = 0; blk < A32<D>::NB; ++blk)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // a = group g = 2 j (registers 8 j .. 8 j + 3), b = group 2 j + 1 (registers 8 j + 4 .. 8 j + 7), two packed words each
      unsigned a0 = pack_bf16x2(acc[blk][8 * j + 0] * mul, acc[blk][8 * j + 1] * mul);
      unsigned a1 = pack_bf16x2(acc[blk][8 * j + 2] * mul, acc[blk][8 * j + 3] * mul);
      unsigned b0 = pack_bf16x2(acc[blk][8 * j + 4] * mul, acc[blk][8 * j + 5] * mul);
      unsigned b1 = pack_bf16x2(acc[blk][8 * j + 6] * mul, acc[blk][8 * j + 7] * mul);
      // swap a's upper half-wave with b's lower half-wave: hi = 0 lanes then hold (own a, partner's a) = channels 16 j .. + 7,
      // hi = 1 lanes (partner's b, own b) = channels 16 j + 8 .. + 15
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      const uint4 v = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
      if (ok) *reinterpret_cast<uint4*>(row_ptr + 32 * blk + 16 * j + 8 * hi) = v;
    }
}

// The text keys' partial dK_c^T / dV_c^T blocks of a wave -> its own fp32 slot [2][D][32] in LDS, with PLAIN stores: LDS float
// atomics (ds_add_f32) into one shared buffer were measured at ~190 LDS cycles per wave-instruction -- 74 of the kernel's 84 M
// LDS-busy cycles at L = 256, d = 96, and every other wave's fragment reads queued behind them (SQ_WAIT_INST_LDS 84 M
// against 8 M without).
template <int D>
__device__ __forceinline__ void store_partial32(float* slot, const f32x16 (&dk)[A32<D>::NB], const f32x16 (&dv)[A32<D>::NB],
                                                const int n, const int hi) {
#pragma unroll
  for (int blk = 0; blk < A32<D>::NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = 32 * blk + (r & 3) + 8 * (r >> 2) + 4 * hi;
      slot[ch * 32 + n] = dk[blk][r];
      slot[(D + ch) * 32 + n] = dv[blk][r];
    }
}
// ... and the sum of `nslot` slots -> dK_c / dV_c rows (whole block; `nthreads` threads)
template <int D>
__device__ __forceinline__ void store_text32(const float* slots, const int nslot, const AttnArgs& p, const int b, const int h,
                                             const int S, const int tid, const int nthreads) {
  constexpr int CPR = A32<D>::CPR;
  for (int i = tid; i < 2 * 32 * CPR; i += nthreads) {
    const int t = i / (32 * CPR), rem_ = i - t * (32 * CPR);
    const int cc = rem_ / 32, key = rem_ - cc * 32;        // consecutive threads <-> consecutive keys: conflict-free LDS reads
    if (key >= S) continue;
    const float mul = t ? 1.f : p.scale;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int w = 0; w < nslot; ++w)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += slots[w * (2 * D * 32) + (t * D + cc * 8 + e) * 32 + key];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)(acc[e] * mul);
    bf16* dst = reinterpret_cast<bf16*>(t ? p.dvc : p.dkc) + (size_t)b * p.dkc_bs + (size_t)h * D + (size_t)key * p.dkc_rs + cc * 8;
    *reinterpret_cast<bf16x8*>(dst) = o;
  }
}

// live text keys of batch element b as a bit mask (bit k: k < S and mask[b][k] != 0); wave-uniform, S <= 32
__device__ __forceinline__ unsigned text_mask32(const AttnArgs& p, const int b, const int S, const int lane) {
  bool live = lane < S;
  if (live && p.mask) live = p.mask[(size_t)b * p.S + lane] != 0.f;
  return (unsigned)__ballot(live);
}

// delta_self = rowsum(dO o (O - O_c)), delta_cross = rowsum(dO o O_c) of query row qi from the dO operand fragments a lane
// already holds (elements 16 s + 8 hi .. + 7 of the row; the two halves of a row sit in lanes n and n + 32)
template <int D>
__device__ __forceinline__ void delta32(const bf16x8 (&gf)[A32<D>::KS], const RowSrc& osrc, const bf16* Ocp, const int o_rs,
                                        const int qi, const bool qok, const int hi, float& a, float& c) {
  a = 0.f; c = 0.f;
#pragma unroll
  for (int s = 0; s < A32<D>::KS; ++s) {
    const bf16x8 of = osrc.frag(qi, s, hi);
    bf16x8 cf = bf16x8{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (Ocp && qok) cf = *reinterpret_cast<const bf16x8*>(Ocp + (size_t)qi * o_rs + 16 * s + 8 * hi);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gv = (float)gf[s][e], ov = (float)of[e], cv = (float)cf[e];
      a += gv * (ov - cv);
      c += gv * cv;
    }
  }
  a += __shfl_xor(a, 32, 64);
  c += __shfl_xor(c, 32, 64);
}

// ---------------------------------------------------------------------------------------------------------------------
// Short sequences (L <= 256 queries, S <= 32 text keys, d = 64 / 96, bf16): the whole backward of one (batch, head) in
// one block of 8 waves, every operand LDS-resident once.
//   stage    K, V, K_c, V_c, Q -> LDS images (all global loads of the stage in flight before the first LDS store)
//   phase Q  a wave owns 32 queries (Q, dO operand fragments and delta = rowsum(dO o O) in registers): for every 32-key
//            tile S^T' = K Q^T - lse / scale, dP^T' = V dO^T - delta (lane <-> query: lse and delta are per-lane scalars),
//            dS^T = 2^(c2 S^T') dP^T', dQ^T += K^T dS^T.  lse' and delta rows of the head go to LDS for phase K.
//   switch   dO -> the LDS region K leaves (Q is already resident)
//   phase K  a wave owns 32 keys (K, V operand fragments in registers): for every 32-query tile S' = Q K^T + lse' (C operand),
//            dP' = dO V^T + (-delta), P = 2^(c2 S'), dS = P dP', dV^T += dO^T P, dK^T += Q^T dS.
//            The text keys are one more 32-key tile: wave w takes it against ITS query tile w, the eight partial
//            [d][32] blocks are summed in LDS (ds_add_f32) and written by the whole block.
// 7 matmuls (S and dP in both phases): keeping dS for the other phase would take 128 KB next to 144 KB of operands.
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512, 1) void attn_bwd_small32_kernel(AttnArgs p) {
  using T = bf16;
  using G = A32<D>;
  constexpr int KS = G::KS, NB = G::NB, CPR = G::CPR, PITCH = G::PITCH;
  constexpr int TILE = 256 * PITCH;          // a 256-row image
  constexpr int CT = 32 * PITCH;             // a text image (32 rows)
  constexpr int T32 = 32 * PITCH;            // one 32-row tile inside an image
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const R0 = smem;                     // K, then dO
  char* const R1 = smem + TILE;              // V, then the text keys' reduction buffer
  char* const R2 = smem + 2 * TILE;          // Q
  char* const KC = smem + 3 * TILE;
  char* const VC = KC + CT;
  float* const fl = reinterpret_cast<float*>(VC + CT);
  float* const nlse_self = fl;               // [256] each: -lse / scale (queries past L: -1e30), -delta
  float* const nlse_cross = fl + 256;
  float* const ndel_self = fl + 512;
  float* const ndel_cross = fl + 768;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hi = lane >> 5;
  const Frag32Off<D> fo(lane);
  const int bh = xcd_remap((int)blockIdx.x, (int)gridDim.x);   // consecutive (batch, head) on one XCD: their rows share lines
  const int b = bh / p.H, h = bh - b * p.H;
  const bool has_c = p.kc != nullptr;
  const int L = p.L, S = has_c ? p.S : 0;
  const int nt = (L + 31) >> 5;              // 32-row tiles (queries = self keys), <= 8
  const float c2 = p.scale * LOG2E;
  const float inv_scale = 1.0f / p.scale;

  const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * D;
  const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)b * p.k_bs + (size_t)h * D;
  const T* DOp = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  const T* Op = reinterpret_cast<const T*>(p.out) + (size_t)b * p.o_bs + (size_t)h * D;
  const T* Kcp = has_c ? reinterpret_cast<const T*>(p.kc) + (size_t)b * p.c_bs + (size_t)h * D : Kp;
  const T* Vcp = has_c ? reinterpret_cast<const T*>(p.vc) + (size_t)b * p.c_bs + (size_t)h * D : Vp;
  const T* Ocp = (has_c && p.out_cross) ? reinterpret_cast<const T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
  const RowSrc qsrc(Qp, p.q_rs, L), ksrc(Kp, p.k_rs, L), vsrc(Vp, p.k_rs, L), gsrc(DOp, p.o_rs, L), osrc(Op, p.o_rs, L);
  const RowSrc kcsrc(Kcp, has_c ? p.c_rs : p.k_rs, S), vcsrc(Vcp, has_c ? p.c_rs : p.k_rs, S);

  const unsigned tmask = has_c ? text_mask32(p, b, S, lane) : 0u;   // live text keys (wave-uniform bit mask)
  // development aid (mdm_dev_set_attn_dbg): shader-clock stamps of the phases, [block][wave][16]
#define ATT_STAMP(i)                                                                                                     \
  if (p.dbg && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = (unsigned long long)__builtin_amdgcn_s_memtime();
  ATT_STAMP(0);

  // `rows` rows of a head view -> a swizzled image; chunk c of the call = row c / CPR, chunk c % CPR
  constexpr int NV = (256 * CPR + 511) / 512;
  auto stage_fetch = [&](uint4 (&v)[NV], const RowSrc& src, int rows) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = tid + i * 512;
      const int row = c / CPR, cc = c - row * CPR;
      v[i] = uint4{0u, 0u, 0u, 0u};
      if (c < rows * CPR) v[i] = src.chunk(row, cc * 16);
    }
  };
  auto stage_commit = [&](char* dst, const uint4 (&v)[NV], int rows) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = tid + i * 512;
      const int row = c / CPR, cc = c - row * CPR;
      if (c < rows * CPR) *reinterpret_cast<uint4*>(dst + G::off(row, cc)) = v[i];
    }
  };

  // ---- stage ------------------------------------------------------------------------------------------------------
  {
    uint4 kv[NV], vv[NV], qv[NV];
    stage_fetch(kv, ksrc, nt * 32);
    stage_fetch(vv, vsrc, nt * 32);
    stage_fetch(qv, qsrc, nt * 32);
    uint4 tc = uint4{0u, 0u, 0u, 0u};
    const int trow = (tid & 255) / CPR, tcc = (tid & 255) - trow * CPR;    // 32 x CPR <= 384 chunks per text image
    uint4 tc2 = uint4{0u, 0u, 0u, 0u};
    if (has_c) {
      // threads 0-255: chunks 0-255 of K_c, threads 256-511: of V_c; the remaining 32 CPR - 256 chunks in a second round
      const RowSrc& ts = __builtin_amdgcn_readfirstlane(tid) < 256 ? kcsrc : vcsrc;   // (wave-uniform: no waterfall loop)
      tc = ts.chunk(trow, tcc * 16);
      if (32 * CPR > 256) {
        const int c = 256 + (tid & 255);
        const int r2 = c / CPR, c2_ = c - r2 * CPR;
        if (c < 32 * CPR) tc2 = ts.chunk(r2, c2_ * 16);
      }
    }
    stage_commit(R0, kv, nt * 32);
    ATT_STAMP(1);
    stage_commit(R1, vv, nt * 32);
    stage_commit(R2, qv, nt * 32);
    if (has_c) {
      char* timg = __builtin_amdgcn_readfirstlane(tid) < 256 ? KC : VC;
      *reinterpret_cast<uint4*>(timg + G::off(trow, tcc)) = tc;
      if (32 * CPR > 256) {
        const int c = 256 + (tid & 255);
        const int r2 = c / CPR, c2_ = c - r2 * CPR;
        if (c < 32 * CPR) *reinterpret_cast<uint4*>(timg + G::off(r2, c2_)) = tc2;
      }
    }
  }

  // ---- phase Q ----------------------------------------------------------------------------------------------------
  const int q0 = wave * 32;
  const bool q_active = __builtin_amdgcn_readfirstlane(q0) < L;
  if (q_active) {
    const int qi = q0 + n;
    bf16x8 qf[KS], gf[KS];                   // (dO from global memory -- its image is staged at the switch; Q from its image)
#pragma unroll
    for (int s = 0; s < KS; ++s) gf[s] = gsrc.frag(qi, s, hi);
    float a, c;
    delta32<D>(gf, osrc, Ocp, p.o_rs, qi, qi < L, hi, a, c);
    const size_t lo = ((size_t)b * p.H + h) * L + (qi < L ? qi : 0);
    const bool qok = qi < L;
    const float nls = qok ? -p.lse_self[lo] * inv_scale : -1e30f;
    const float nlc = (qok && has_c) ? -p.lse_cross[lo] * inv_scale : -1e30f;
    const float nds = qok ? -a : 0.f, ndc = qok ? -c : 0.f;
    if (hi == 0) {
      nlse_self[qi] = nls; nlse_cross[qi] = nlc; ndel_self[qi] = nds; ndel_cross[qi] = ndc;
    }
    ATT_STAMP(2);
    __syncthreads();                         // (all 512 threads reach one of the two barriers of this if / else)
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = lds_b128(R2 + q0 * PITCH + fo.b[s]);

    f32x16 dq[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) dq[blk] = splat16(0.f);
    const int ntile = nt + (has_c ? 1 : 0);
    Ops32<D, QPf<D>::value> ko;
    load_ops32<D, QPf<D>::value>(ko, R0, R1, fo);       // tile 0 (the next tile's operands are read inside each step)
    ATT_STAMP(3);
    for (int kt = 0; kt < ntile; ++kt) {
      const bool cross = kt >= nt;
      const char* Kt = cross ? KC : R0 + kt * T32;
      const int kn = kt + 1 < ntile ? kt + 1 : kt;
      const char* Vt = cross ? VC : R1 + kt * T32;
      const char* Kn = kn >= nt ? KC : R0 + kn * T32;
      const char* Vn = kn >= nt ? VC : R1 + kn * T32;
      const int rem = L - kt * 32;
      const unsigned live = cross ? tmask : (rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u));
      prio_flip(kt + (wave >> 2));
      q_step32<D>(ko, Kt, Vt, Kn, Vn, qf, gf, cross ? nlc : nls, cross ? ndc : nds, live, c2, hi, fo, dq);
    }
    ATT_STAMP(4);
    // dQ^T: lane <-> query, register r <-> channel 32 blk + 8 (r >> 2) + 4 hi + (r & 3)
    store_rows32<D>(reinterpret_cast<T*>(p.dq) + (size_t)b * p.q_bs + (size_t)h * D + (size_t)(qok ? qi : 0) * p.q_rs, dq, p.scale, hi, qok);
  } else {
    __syncthreads();
  }

  // ---- switch: dO -> R0 ---------------------------------------------------------------------------------------------------
  ATT_STAMP(5);
  // the operand fragments of phase K (this wave's 32 keys) come out of the K / V images while they are still in place
  const int k0 = wave * 32;
  const bool k_active = __builtin_amdgcn_readfirstlane(k0) < L;
  bf16x8 kf[KS], vf[KS];
  if (k_active) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[s] = lds_b128(R0 + k0 * PITCH + fo.b[s]);
      vf[s] = lds_b128(R1 + k0 * PITCH + fo.b[s]);
    }
  }
  __syncthreads();                           // every wave is done with K, V, K_c, V_c
  ATT_STAMP(6);
  {
    uint4 gv[NV];
    stage_fetch(gv, gsrc, nt * 32);
    stage_commit(R0, gv, nt * 32);
  }
  __syncthreads();
  ATT_STAMP(7);

  // ---- phase K ----------------------------------------------------------------------------------------------------
  if (k_active) {
    const int key = k0 + n;
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) { dk[blk] = splat16(0.f); dv[blk] = splat16(0.f); }
    Ops32<D, KPf<D>::value> qo;
    load_ops32<D, KPf<D>::value>(qo, R2, R0, fo);
    for (int qt = 0; qt < nt; ++qt) {
      const int qn = qt + 1 < nt ? qt + 1 : qt;
      prio_flip(qt + (wave >> 2));
      k_step32<D>(qo, R2 + qt * T32, R0 + qt * T32, R2 + qn * T32, R0 + qn * T32, nlse_self + qt * 32, ndel_self + qt * 32,
                  kf, vf, key < L, c2, hi, fo, dk, dv);
    }
    ATT_STAMP(8);
    {
      const size_t ro = (size_t)b * p.dk_bs + (size_t)h * D + (size_t)(key < L ? key : 0) * p.dk_rs;
      store_rows32<D>(reinterpret_cast<T*>(p.dk) + ro, dk, p.scale, hi, key < L);
      store_rows32<D>(reinterpret_cast<T*>(p.dv) + ro, dv, 1.f, hi, key < L);
    }
    ATT_STAMP(9);
  }
  if (has_c) {
    // The text keys: waves 0-3 take the query tiles w, w + 4 against them; their four partial [2][d][32] blocks go to LDS
    // slots (the operand images are dead by then) and the whole block adds them up on the way out.
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) { dk[blk] = splat16(0.f); dv[blk] = splat16(0.f); }
    if (wave < 4 && __builtin_amdgcn_readfirstlane(wave) < nt) {
      bf16x8 kf[KS], vf[KS];                 // the text keys' operand fragments: their LDS images are still in place
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        kf[s] = lds_b128(KC + fo.b[s]);
        vf[s] = lds_b128(VC + fo.b[s]);
      }
      Ops32<D, KPf<D>::value> qo;
      load_ops32<D, KPf<D>::value>(qo, R2 + wave * T32, R0 + wave * T32, fo);
      for (int qt = wave; qt < nt; qt += 4) {
        const int qn = qt + 4 < nt ? qt + 4 : qt;
        k_step32<D>(qo, R2 + qt * T32, R0 + qt * T32, R2 + qn * T32, R0 + qn * T32, nlse_cross + qt * 32, ndel_cross + qt * 32,
                    kf, vf, ((tmask >> n) & 1u) != 0u, c2, hi, fo, dk, dv);
      }
    }
    ATT_STAMP(10);
    __syncthreads();                         // nobody reads Q / dO any more
    float* slots = reinterpret_cast<float*>(smem);   // [4][2][D][32] over R0, R1 (4 x 24 KB at d = 96)
    if (wave < 4) store_partial32<D>(slots + wave * (2 * D * 32), dk, dv, n, hi);
    ATT_STAMP(11);
    __syncthreads();
    store_text32<D>(slots, 4, p, b, h, S, tid, 512);
    ATT_STAMP(12);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long sequences (L > 256: the 32x32 level of the U-Net, L = 1024, d = 64): two streaming kernels on the same tile steps.
//
// attn_bwd_dq32_kernel   block = 256 queries of one (batch, head), 8 waves x 32 queries (Q, dO operand fragments, lse, delta in
//   registers; delta = rowsum(dO o O) is computed here and stored for the dK / dV kernel).  The keys stream through LDS in
//   stages of 128 (K and V rows; three LDS buffers: the stage after next is fetched during this stage's MFMAs and committed
//   after them, so a tile can read its successor's operands ahead even across a stage boundary; one barrier per stage);
//   the text keys are one more stage.
// attn_bwd_dkv32_kernel  block = 256 keys, 8 waves x 32 keys (K, V operand fragments in registers); the queries stream
//   through LDS in stages of 128 (Q and dO rows + their -lse / scale and -delta).  The text keys of a (batch, head) are ONE
//   more block whose waves each take every 8th 32-query tile (staged privately per wave: no block barrier in that loop)
//   and sum their partial dK_c^T / dV_c^T in LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NTHR = 512> struct Stream32 {
  using G = A32<D>;
  static constexpr int SR = MDM_ATT_SR;                      // rows (keys / queries) of a stage: SR / 32 tiles per barrier
  static constexpr int NSUB = SR / 32;
  static constexpr int HALF = SR * G::PITCH;                 // one tensor's rows
  static constexpr int NVS = 2 * SR * G::CPR / NTHR;         // 16-byte chunks per thread and stage (two tensors)
  static_assert(2 * SR * G::CPR % NTHR == 0, "stage chunks divide over the block's threads");
  // chunk i of thread tid: tensor (c / (SR CPR)), row, chunk.  The tensor of a chunk is the same for a whole wave (SR CPR is a
  // multiple of 64); saying so (readfirstlane) keeps the buffer descriptor in SGPRs -- a per-lane choice between two
  // descriptors makes every load a waterfall loop
  static __device__ __forceinline__ void fetch(uint4 (&v)[NVS], const RowSrc& a, const RowSrc& b_, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < NVS; ++i) {
      const int c = tid + i * NTHR;
      const int which = c / (SR * G::CPR), c1 = c - which * (SR * G::CPR);
      const int row = c1 / G::CPR, cc = c1 - row * G::CPR;
      if (__builtin_amdgcn_readfirstlane(which)) v[i] = b_.chunk(row0 + row, cc * 16);
      else v[i] = a.chunk(row0 + row, cc * 16);
    }
  }
  static __device__ __forceinline__ void commit(char* stage, const uint4 (&v)[NVS], int tid) {
#pragma unroll
    for (int i = 0; i < NVS; ++i) {
      const int c = tid + i * NTHR;
      const int which = c / (SR * G::CPR), c1 = c - which * (SR * G::CPR);
      const int row = c1 / G::CPR, cc = c1 - row * G::CPR;
      *reinterpret_cast<uint4*>(stage + which * HALF + G::off(row, cc)) = v[i];
    }
  }
};

template <int D>
__global__ __launch_bounds__(512, 1) void attn_bwd_dq32_kernel(AttnArgs p) {
  using T = bf16;
  using G = A32<D>;
  using ST = Stream32<D>;
  constexpr int KS = G::KS, NB = G::NB, PITCH = G::PITCH, T32 = 32 * PITCH;
  constexpr int STAGE = 2 * ST::HALF, SR = ST::SR, NSUB = ST::NSUB;   // K rows | V rows of SR keys; three stages in LDS
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hi = lane >> 5;
  const Frag32Off<D> fo(lane);
  const int bx_ = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = bx_ / (int)gridDim.x, bx = bx_ - by * (int)gridDim.x;
  const int b = by / p.H, h = by - b * p.H;
  const bool has_c = p.kc != nullptr;
  const int L = p.L, S = has_c ? p.S : 0;
  const float c2 = p.scale * LOG2E, inv_scale = 1.0f / p.scale;

  const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * D;
  const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)b * p.k_bs + (size_t)h * D;
  const T* DOp = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  const T* Op = reinterpret_cast<const T*>(p.out) + (size_t)b * p.o_bs + (size_t)h * D;
  const T* Kcp = has_c ? reinterpret_cast<const T*>(p.kc) + (size_t)b * p.c_bs + (size_t)h * D : Kp;
  const T* Vcp = has_c ? reinterpret_cast<const T*>(p.vc) + (size_t)b * p.c_bs + (size_t)h * D : Vp;
  const T* Ocp = (has_c && p.out_cross) ? reinterpret_cast<const T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
  const RowSrc qsrc(Qp, p.q_rs, L), ksrc(Kp, p.k_rs, L), vsrc(Vp, p.k_rs, L), gsrc(DOp, p.o_rs, L), osrc(Op, p.o_rs, L);
  const RowSrc kcsrc(Kcp, has_c ? p.c_rs : p.k_rs, S), vcsrc(Vcp, has_c ? p.c_rs : p.k_rs, S);
  const unsigned tmask = has_c ? text_mask32(p, b, S, lane) : 0u;

  const int q0 = bx * 256 + wave * 32, qi = q0 + n;
  const bool w_active = __builtin_amdgcn_readfirstlane(q0) < L, qok = qi < L;
  const int nself = (L + SR - 1) / SR, ntot = nself + (has_c ? 1 : 0);
  // stage t = self keys SR t .. SR t + SR - 1, or (t == nself) the text keys; its global loads
  uint4 stg[ST::NVS];
  auto fetch = [&](const int t) {
    if (t < nself) ST::fetch(stg, ksrc, vsrc, t * SR, tid);
    else ST::fetch(stg, kcsrc, vcsrc, 0, tid);
  };
  fetch(0);
  bf16x8 qf[KS], gf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qf[s] = qsrc.frag(qi, s, hi);
    gf[s] = gsrc.frag(qi, s, hi);
  }
  ST::commit(smem, stg, tid);
  if (ntot > 1) {
    fetch(1);
    ST::commit(smem + STAGE, stg, tid);
  }
  float a, c;
  delta32<D>(gf, osrc, Ocp, p.o_rs, qi, qok, hi, a, c);
  const size_t lo = ((size_t)b * p.H + h) * L + (qok ? qi : 0);
  if (qok && hi == 0) {
    p.delta_self_w[lo] = a;
    if (p.delta_cross_w) p.delta_cross_w[lo] = c;
  }
  const float nls = qok ? -p.lse_self[lo] * inv_scale : -1e30f;
  const float nlc = (qok && has_c) ? -p.lse_cross[lo] * inv_scale : -1e30f;
  const float nds = qok ? -a : 0.f, ndc = qok ? -c : 0.f;

  f32x16 dq[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) dq[blk] = splat16(0.f);
  frags_arrived(qf);
  frags_arrived(gf);
  __syncthreads();                                             // stages 0 and 1 are in LDS
  Ops32<D, QPf<D>::value> ko;
  load_ops32<D, QPf<D>::value>(ko, smem, smem + ST::HALF, fo);
  // Iteration t: stage t sits in buffer t % 3 and stage t + 1 in buffer (t + 1) % 3 (committed one iteration ago, visible
  // since this iteration's barrier); stage t + 2 is fetched now and committed into buffer (t + 2) % 3 -- which every wave
  // left before the barrier -- after the arithmetic.  So the last tile of a stage reads its successor's operands ahead
  // like any other.  CUR is compile-time: fragment addresses are per-lane offset + immediate.
  auto stage = [&](auto cur_c, const int t) {
    constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % 3, NN = (CUR + 2) % 3;
    const char* Ks = smem + CUR * STAGE;
    if (!(MDM_ATT_ABL & 4)) __syncthreads();
    const bool more = t + 2 < ntot && !(MDM_ATT_ABL & 8);
    if (more) fetch(t + 2);
    if (w_active) {
      const bool cross = t >= nself;
      const int nsub = cross ? 1 : min(NSUB, (L - t * SR + 31) >> 5);
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        if (sub >= nsub) break;
        prio_flip(t * NSUB + sub + (wave >> 2));
        const int rem = L - (t * SR + sub * 32);
        const unsigned live = cross ? tmask : (rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u));
        // the next tile: the next tile of this stage, else the first of the next stage, else (nothing left) itself
        const bool in_stage = sub + 1 < nsub, last = !in_stage && t + 1 >= ntot;
        const char* Kn = in_stage ? Ks + (sub + 1) * T32 : (last ? Ks + sub * T32 : smem + NXT * STAGE);
        q_step32<D>(ko, Ks + sub * T32, Ks + ST::HALF + sub * T32, Kn, Kn + ST::HALF, qf, gf, cross ? nlc : nls, cross ? ndc : nds,
                    live, c2, hi, fo, dq);
      }
    }
    if (more) ST::commit(smem + NN * STAGE, stg, tid);
  };
  for (int t = 0; t < ntot; t += 3) {
    stage(IntC<0>{}, t);
    if (t + 1 < ntot) stage(IntC<1>{}, t + 1);
    if (t + 2 < ntot) stage(IntC<2>{}, t + 2);
  }
  store_rows32<D>(reinterpret_cast<T*>(p.dq) + (size_t)b * p.q_bs + (size_t)h * D + (size_t)(qok ? qi : 0) * p.q_rs, dq, p.scale, hi, qok);
}

template <int D>
__global__ __launch_bounds__(512, 1) void attn_bwd_dkv32_kernel(AttnArgs p) {
  using T = bf16;
  using G = A32<D>;
  using ST = Stream32<D>;
  constexpr int KS = G::KS, NB = G::NB, CPR = G::CPR, PITCH = G::PITCH, T32 = 32 * PITCH, PF = KPf<D>::value;
  constexpr int SR = ST::SR, NSUB = ST::NSUB;
  constexpr int STAGE = 2 * ST::HALF + 8 * SR;                 // Q rows | dO rows | -lse / scale [SR] | -delta [SR]
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hi = lane >> 5;
  const Frag32Off<D> fo(lane);
  const int bx_ = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = bx_ / (int)gridDim.x, bx = bx_ - by * (int)gridDim.x;
  const int b = by / p.H, h = by - b * p.H;
  const bool has_c = p.kc != nullptr;
  const int L = p.L, S = has_c ? p.S : 0;
  const float c2 = p.scale * LOG2E, inv_scale = 1.0f / p.scale;
  const int nself_blocks = (L + 255) >> 8;

  const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* DOp = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  const RowSrc qsrc(Qp, p.q_rs, L), gsrc(DOp, p.o_rs, L);
  const size_t lrow = ((size_t)b * p.H + h) * L;

  if (bx < nself_blocks) {
    // ---- 256 self keys ------------------------------------------------------------------------------------------------
    const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * D;
    const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)b * p.k_bs + (size_t)h * D;
    const RowSrc ksrc(Kp, p.k_rs, L), vsrc(Vp, p.k_rs, L);
    const int k0 = bx * 256 + wave * 32, key = k0 + n;
    const bool w_active = __builtin_amdgcn_readfirstlane(k0) < L;
    const int nst = (L + SR - 1) / SR;
    uint4 stg[ST::NVS];
    // threads 0 .. SR-1: lse, SR .. 2 SR - 1: delta of query SR t + (tid % SR).  The RAW value is kept until the commit: arithmetic on it
    // here would put an s_waitcnt vmcnt(0) -- a wait for the whole stage's global loads -- in front of the stage's MFMAs
    float fl_r = 0.f;
    bool fl_ok = false;
    const float* const fl_src = (tid < SR ? p.lse_self : p.delta_self) + lrow;
    auto fetch = [&](const int t) {
      ST::fetch(stg, qsrc, gsrc, t * SR, tid);
      if (tid < 2 * SR) {
        const int q = t * SR + (tid & (SR - 1));
        fl_ok = q < L;
        fl_r = fl_src[fl_ok ? q : 0];
      }
    };
    auto commit = [&](char* buf) {
      ST::commit(buf, stg, tid);
      if (tid < 2 * SR) {
        const float v = tid < SR ? (fl_ok ? -fl_r * inv_scale : -1e30f) : (fl_ok ? -fl_r : 0.f);
        reinterpret_cast<float*>(buf + 2 * ST::HALF)[tid] = v;
      }
    };
    fetch(0);
    bf16x8 kf[KS], vf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      kf[s] = ksrc.frag(key, s, hi);
      vf[s] = vsrc.frag(key, s, hi);
    }
    commit(smem);
    if (nst > 1) {
      fetch(1);
      commit(smem + STAGE);
    }
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) { dk[blk] = splat16(0.f); dv[blk] = splat16(0.f); }
    frags_arrived(kf);
    frags_arrived(vf);
    __syncthreads();
    Ops32<D, PF> qo;
    load_ops32<D, PF>(qo, smem, smem + ST::HALF, fo);
    // (three LDS buffers, as in attn_bwd_dq32_kernel)
    auto stage = [&](auto cur_c, const int t) {
      constexpr int CUR = decltype(cur_c)::value, NXT = (CUR + 1) % 3, NN = (CUR + 2) % 3;
      const char* Qs = smem + CUR * STAGE;
      const float* nl = reinterpret_cast<const float*>(Qs + 2 * ST::HALF);
      if (!(MDM_ATT_ABL & 4)) __syncthreads();
      const bool more = t + 2 < nst && !(MDM_ATT_ABL & 8);
      if (more) fetch(t + 2);
      if (w_active) {
        const int nsub = min(NSUB, (L - t * SR + 31) >> 5);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
          if (sub >= nsub) break;
          prio_flip(t * NSUB + sub + (wave >> 2));
          const bool in_stage = sub + 1 < nsub, last = !in_stage && t + 1 >= nst;
          const char* Qn = in_stage ? Qs + (sub + 1) * T32 : (last ? Qs + sub * T32 : smem + NXT * STAGE);
          k_step32<D>(qo, Qs + sub * T32, Qs + ST::HALF + sub * T32, Qn, Qn + ST::HALF, nl + sub * 32, nl + SR + sub * 32,
                      kf, vf, key < L, c2, hi, fo, dk, dv);
        }
      }
      if (more) commit(smem + NN * STAGE);
    };
    for (int t = 0; t < nst; t += 3) {
      stage(IntC<0>{}, t);
      if (t + 1 < nst) stage(IntC<1>{}, t + 1);
      if (t + 2 < nst) stage(IntC<2>{}, t + 2);
    }
    {
      const size_t ro = (size_t)b * p.dk_bs + (size_t)h * D + (size_t)(key < L ? key : 0) * p.dk_rs;
      store_rows32<D>(reinterpret_cast<T*>(p.dk) + ro, dk, p.scale, hi, key < L);
      store_rows32<D>(reinterpret_cast<T*>(p.dv) + ro, dv, 1.f, hi, key < L);
    }
    return;
  }

  // ---- the text keys: wave w takes query tiles w, w + 8, ... (private LDS tile per wave) ------------------------------------
  const T* Kcp = reinterpret_cast<const T*>(p.kc) + (size_t)b * p.c_bs + (size_t)h * D;
  const T* Vcp = reinterpret_cast<const T*>(p.vc) + (size_t)b * p.c_bs + (size_t)h * D;
  const RowSrc kcsrc(Kcp, p.c_rs, S), vcsrc(Vcp, p.c_rs, S);
  const unsigned tmask = text_mask32(p, b, S, lane);
  constexpr int PRIV = 2 * T32 + 256;                          // Q tile | dO tile | -lse / scale [32] | -delta [32]
  char* const mine = smem + wave * PRIV;
  bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = kcsrc.frag(n, s, hi);
    vf[s] = vcsrc.frag(n, s, hi);
  }
  f32x16 dk[NB], dv[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) { dk[blk] = splat16(0.f); dv[blk] = splat16(0.f); }
  const int nt = (L + 31) >> 5;
  constexpr int NVP = 32 * CPR / 64;                           // chunks per lane and tensor of a 32-row tile
  for (int qt = wave; qt < nt; qt += 8) {
    uint4 qv[NVP], gv[NVP];
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      const int c = lane + i * 64;
      const int row = c / CPR, cc = c - row * CPR;
      qv[i] = qsrc.chunk(qt * 32 + row, cc * 16);
      gv[i] = gsrc.chunk(qt * 32 + row, cc * 16);
    }
    const int q = qt * 32 + n;
    float f = 0.f;
    if (hi == 0) f = q < L ? -p.lse_cross[lrow + q] * inv_scale : -1e30f;
    else f = q < L ? -p.delta_cross[lrow + q] : 0.f;
    __builtin_amdgcn_wave_barrier();                           // (the previous tile's LDS reads are issued before these writes)
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
      const int c = lane + i * 64;
      const int row = c / CPR, cc = c - row * CPR;
      *reinterpret_cast<uint4*>(mine + G::off(row, cc)) = qv[i];
      *reinterpret_cast<uint4*>(mine + T32 + G::off(row, cc)) = gv[i];
    }
    reinterpret_cast<float*>(mine + 2 * T32)[lane] = f;        // lanes 0-31: -lse / scale, 32-63: -delta
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* nl = reinterpret_cast<const float*>(mine + 2 * T32);
    Ops32<D, PF> qo;
    load_ops32<D, PF>(qo, mine, mine + T32, fo);
    k_step32<D>(qo, mine, mine + T32, mine, mine + T32, nl, nl + 32, kf, vf, ((tmask >> n) & 1u) != 0u, c2, hi, fo, dk, dv);
  }
  __syncthreads();                                             // the private tiles are dead: the slots overlay them
  float* const slots = reinterpret_cast<float*>(smem);         // [8][2][D][32]
  store_partial32<D>(slots + wave * (2 * D * 32), dk, dv, n, hi);
  __syncthreads();
  store_text32<D>(slots, 8, p, b, h, S, tid, 512);
}


template <int D> constexpr int attn_bwd_dq32_lds() { return 3 * 2 * Stream32<D>::SR * 2 * D; }
template <int D> constexpr int attn_bwd_dkv32_lds() {
  constexpr int stages = 3 * (2 * Stream32<D>::SR * 2 * D + 8 * Stream32<D>::SR), priv = 8 * (2 * 32 * 2 * D + 256), slots = 8 * 2 * D * 32 * 4;
  return stages > priv ? (stages > slots ? stages : slots) : (priv > slots ? priv : slots);
}

#undef ATT_STAMP
template <int D> constexpr int attn_bwd_small32_lds() { return 3 * 256 * 2 * D + 2 * 32 * 2 * D + 4096; }

}  // namespace mdm
