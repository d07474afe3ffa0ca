// Fused (flash-style) self + text-cross attention, forward and backward, gfx950 MFMA.
//
// Reference semantics (ml_mdm/models/unet.py:276-313, SelfAttention):
//   q, k, v = split(qkv(norm(x)))                     heads are channel-major: head h = channels [h*d, (h+1)*d)
//   h_self  = softmax((q*s)^T (k*s)) v ,  s = d^-1/4  (softmax in fp32, unet.py:292)
//   h_cross = softmax((q*s)^T (k_c*s) masked) v_c     with k_c, v_c = split(kv_cond(norm_cond(cond)))
//   h = h_self + h_cross                              two *separate* softmaxes (unet.py:302-307)
// Nothing of size L x L is ever written to HBM; the backward recomputes the
// probabilities from the saved log-sum-exp of each softmax.
//
// Layout: activations are NHWC, so the 1x1-conv output qkv is [B, L, 3C] and a
// head's rows are strided views (row stride 3C) -- no head split/merge copies.
//
// MFMA operand plan (16x16 tiles, fragment = 8 reduction elements per lane,
// common.hpp).  Forward, per 64-key tile held in LDS:
//   S^T = K Q^T     A = K rows (LDS, natural [key][d]),   B = Q rows (registers)
//   O^T = V^T P^T   A = V^T rows (LDS, transposed [d][key]), B = P (registers)
// With S^T the softmax row of a query lives in one lane column (lane & 15) so
// the running max / sum need only two cross-quad shuffles, and the P registers
// are already a valid B fragment for the PV product.  The K rows of key tile kt
// are read in the permuted order key(kt, r) = (kt>>1)*32 + (r>>2)*8 + (kt&1)*4 + (r&3)
// so that a lane's 8 probabilities of one PV step are 8 *contiguous* keys of the
// transposed V tile.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace mdm {

struct AttnArgs {
  // forward operands; row (b, i) of tensor X, head h, channel c: X[b * X_bs + i * X_rs + h * d + c]
  const void* q; size_t q_bs; int q_rs;
  const void* k; const void* v; size_t k_bs; int k_rs;      // self keys / values (same strides)
  const void* kc; const void* vc; size_t c_bs; int c_rs;    // cross keys / values, null if absent
  const float* mask;                                         // [B, S] 0/1 or null
  void* out; void* out_cross; size_t o_bs; int o_rs;         // out_cross may be null
  float* lse_self; float* lse_cross;                         // [B, H, L]
  // backward only
  const void* dout;                                          // same strides as out
  const float* delta_self; const float* delta_cross;         // [B, H, L]
  void* dq;                                                  // strides as q
  void* dk; void* dv; size_t dk_bs; int dk_rs;               // gradient of the self (k, v)
  void* dkc; void* dvc; size_t dkc_bs; int dkc_rs;           // gradient of the cross (k_c, v_c); null if absent
  float* delta_self_w; float* delta_cross_w;                 // written by the dQ kernel, read by the dK/dV kernel
  int B, H, L, S;
  float scale;                                               // 1/sqrt(d)
  unsigned long long* dbg;                                   // development aid (mdm_dev_set_attn_dbg): phase time stamps, or null
};

template <typename T> __device__ __forceinline__ void frag_from_global(Frag<T>& f, const T* p, bool valid);
template <> __device__ __forceinline__ void frag_from_global<bf16>(Frag<bf16>& f, const bf16* p, bool valid) {
  uint4 z = {0u, 0u, 0u, 0u};
  uint4 r = valid ? *reinterpret_cast<const uint4*>(p) : z;
  f.v = *reinterpret_cast<bf16x8*>(&r);
}
template <> __device__ __forceinline__ void frag_from_global<float>(Frag<float>& f, const float* p, bool valid) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f.lo = valid ? *reinterpret_cast<const f32x4*>(p) : z;
  f.hi = valid ? *reinterpret_cast<const f32x4*>(p + 4) : z;
}

// the 8 elements of a fragment as fp32
template <typename T> __device__ __forceinline__ void frag_to_f32(const Frag<T>& f, float (&o)[8]);
template <> __device__ __forceinline__ void frag_to_f32<bf16>(const Frag<bf16>& f, float (&o)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)f.v[e];
}
template <> __device__ __forceinline__ void frag_to_f32<float>(const Frag<float>& f, float (&o)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = f.lo[e]; o[4 + e] = f.hi[e]; }
}

// two f32x4 score tiles -> one B fragment (8 consecutive reduction slots)
template <typename T> __device__ __forceinline__ void frag_from_acc(Frag<T>& f, const f32x4& a, const f32x4& b);
template <> __device__ __forceinline__ void frag_from_acc<bf16>(Frag<bf16>& f, const f32x4& a, const f32x4& b) {
  f.v = bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
template <> __device__ __forceinline__ void frag_from_acc<float>(Frag<float>& f, const f32x4& a, const f32x4& b) {
  f.lo = a; f.hi = b;
}
// the same three for any fragment type (overloads; FragSplit = fp32 values as bf16 hi + lo, common.hpp)
__device__ __forceinline__ void frag_from_global_x(Frag<bf16>& f, const bf16* p, bool valid) { frag_from_global<bf16>(f, p, valid); }
__device__ __forceinline__ void frag_from_global_x(Frag<float>& f, const float* p, bool valid) { frag_from_global<float>(f, p, valid); }
__device__ __forceinline__ void frag_from_global_x(FragSplit& f, const float* p, bool valid) {
  Frag<float> t;
  frag_from_global<float>(t, p, valid);
  f.from_f32(t.lo, t.hi);
}
__device__ __forceinline__ void frag_from_acc_x(Frag<bf16>& f, const f32x4& a, const f32x4& b) { frag_from_acc<bf16>(f, a, b); }
__device__ __forceinline__ void frag_from_acc_x(Frag<float>& f, const f32x4& a, const f32x4& b) { frag_from_acc<float>(f, a, b); }
__device__ __forceinline__ void frag_from_acc_x(FragSplit& f, const f32x4& a, const f32x4& b) { f.from_f32(a, b); }

template <typename T, int D> struct AttnGeom {
  static constexpr int EPV = Tr<T>::EPV, KSTEPS = Tr<T>::KSTEPS;
  static constexpr int DS = D / 32;                        // 32-deep mma steps across d
  static constexpr int DT = D / 16;                        // 16-wide tiles across d
  static constexpr int NPAN = (DS + KSTEPS - 1) / KSTEPS;  // 128-byte panels of a natural [64][d] tile
  static constexpr int TPAN = 2 / KSTEPS;                  // panels of a transposed [d][64] tile
  static constexpr int NAT_BYTES = NPAN * 64 * 128;
  static constexpr int TR_BYTES = TPAN * D * 128;
  static constexpr int CPR = D / EPV;                      // chunks per row
};

// rows [row0, row0+64) of a [rows][d] head view -> LDS natural tile (rows >= nrows zero-filled)
template <typename T, int D>
__device__ __forceinline__ void load_nat_tile(char* dst, const T* src, int rs, int row0, int nrows, int tid) {
  using G = AttnGeom<T, D>;
  for (int c = tid; c < 64 * G::CPR; c += 256) {
    const int row = c / G::CPR, cc = c - row * G::CPR;
    const bool ok = row0 + row < nrows;
    uint4 val = *reinterpret_cast<const uint4*>(src + (ok ? (size_t)(row0 + row) * rs + cc * G::EPV : (size_t)0));
    val.x = ok ? val.x : 0u; val.y = ok ? val.y : 0u; val.z = ok ? val.z : 0u; val.w = ok ? val.w : 0u;
    *reinterpret_cast<uint4*>(dst + (cc >> 3) * (64 * 128) + lds_chunk_off(row, cc & 7)) = val;
  }
}
// same rows, written transposed: LDS [d][64 rows]
template <typename T, int D>
__device__ __forceinline__ void load_tr_tile(char* dst, const T* src, int rs, int row0, int nrows, int tid) {
  using G = AttnGeom<T, D>;
  constexpr int NBLK = (64 / G::EPV) * G::CPR;
  for (int bi = tid; bi < NBLK; bi += 256) {
    // 8 consecutive lanes = the row blocks of one channel chunk: conflict-free transposed stores (see wgrad)
    constexpr int KB = 64 / G::EPV;
    const int cc = bi / KB, kb = bi - cc * KB;
    Blk<T> blk;
#pragma unroll
    for (int e = 0; e < G::EPV; ++e) {
      const int row = row0 + kb * G::EPV + e;
      const bool ok = row < nrows;
      blk.load_row_sel(e, src + (ok ? (size_t)row * rs + cc * G::EPV : (size_t)0), ok);
    }
#pragma unroll
    for (int c = 0; c < G::EPV; ++c)
      *reinterpret_cast<uint4*>(dst + (kb >> 3) * (D * 128) + lds_chunk_off(cc * G::EPV + c, kb & 7)) = blk.col(c);
  }
}

// The same natural tile in two steps, so the global loads of the NEXT tile can be in flight while the MFMAs of the
// current one run: fetch (HBM -> registers) ... compute ... commit (registers -> LDS).
template <typename T, int D> struct NatRegs { uint4 v[(64 * AttnGeom<T, D>::CPR + 255) / 256]; };
// The global side goes through a buffer descriptor over rows [0, nrows) of the head view: per-lane byte offsets are
// computed once, a tile is one 32-bit add per load, and rows past the end read as zero by the range check (no 64-bit
// address arithmetic or per-element selects inside the key loop).
template <typename T, int D> struct NatSrc {
  static constexpr int NV = (64 * AttnGeom<T, D>::CPR + 255) / 256;
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned vo[NV];
  unsigned row_bytes;
  __device__ __forceinline__ NatSrc(const T* src, int rs, int nrows, int tid) {
    using G = AttnGeom<T, D>;
    row_bytes = (unsigned)rs * (unsigned)sizeof(T);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(src), 0, (unsigned)nrows * row_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = tid + 256 * i;
      const int row = c / G::CPR, cc = c - row * G::CPR;
      vo[i] = c < 64 * G::CPR ? (unsigned)row * row_bytes + (unsigned)(cc * 16) : 0x7F000000u;
    }
  }
};
template <typename T, int D>
__device__ __forceinline__ void fetch_nat(NatRegs<T, D>& r, const NatSrc<T, D>& s, int row0) {
  const unsigned base = (unsigned)row0 * s.row_bytes;
#pragma unroll
  for (int i = 0; i < NatSrc<T, D>::NV; ++i) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, s.vo[i] + base, 0, 0);
    r.v[i] = uint4{(unsigned)v[0], (unsigned)v[1], (unsigned)v[2], (unsigned)v[3]};
  }
}
template <typename T, int D>
__device__ __forceinline__ void commit_nat(char* dst, const NatRegs<T, D>& r, int tid) {
  using G = AttnGeom<T, D>;
  constexpr int NV = (64 * G::CPR + 255) / 256;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = tid + 256 * i;
    const int row = c / G::CPR, cc = c - row * G::CPR;
    if (c < 64 * G::CPR) *reinterpret_cast<uint4*>(dst + (cc >> 3) * (64 * 128) + lds_chunk_off(row, cc & 7)) = r.v[i];
  }
}

// Fragment whose rows are CHANNELS (d index c16*16 + l16) and whose 8 reduction slots are tile ROWS
// hh*32 + quad*8 + [0, 8) -- i.e. a fragment of the transposed tile.
//   bf16: read from the NATURAL [row][d] image with the LDS transpose read (lane i of a 16-lane group supplies
//         row i>>2, 4-channel segment i&3 and receives channel i; tools/probes/ds_read_tr_probe.hip), so no second,
//         transposed copy of the tile is ever staged;
//   fp32: read from the separately staged transposed image (ds_read_tr is a 16-bit instruction).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
template <typename T, int D>
__device__ __forceinline__ void load_frag_T(Frag<T>& f, const char* nat, const char* tr, int c16, int hh, int quad, int l16);
template <int D>
struct FragT {
  static __device__ __forceinline__ void bf(Frag<bf16>& f, const char* nat, int c16, int hh, int quad, int l16) {
    const int d = c16 * 16 + (l16 & 3) * 4;
    const int r = hh * 32 + quad * 8 + (l16 >> 2);
    const char* base = nat + (d >> 6) * (64 * 128) + (((d >> 2) & 1) << 3);
    const int chunk = (d & 63) >> 3;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + lds_chunk_off(r, chunk)));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + lds_chunk_off(r + 4, chunk)));
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    f.v = *reinterpret_cast<bf16x8*>(&v);
  }
};
// The per-lane byte offsets of FragT::bf inside a natural tile, computed once per kernel: the read of channel tile c16,
// row half hh is then `tile + lo[c16] + hh * 4096` (rows + 32 keep their swizzle), an immediate when `tile` is a
// compile-time LDS offset.
template <int D> struct TrOff {
  unsigned lo[D / 16], hi[D / 16];
  __device__ __forceinline__ TrOff(int quad, int l16) {
#pragma unroll
    for (int c16 = 0; c16 < D / 16; ++c16) {
      const int d = c16 * 16 + (l16 & 3) * 4;
      const int r = quad * 8 + (l16 >> 2);
      const unsigned base = (unsigned)((d >> 6) * (64 * 128) + (((d >> 2) & 1) << 3));
      lo[c16] = base + (unsigned)lds_chunk_off(r, (d & 63) >> 3);
      hi[c16] = base + (unsigned)lds_chunk_off(r + 4, (d & 63) >> 3);
    }
  }
  __device__ __forceinline__ void read(Frag<bf16>& f, const char* nat, int c16, int hh) const {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(nat + lo[c16] + hh * 4096));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(nat + hi[c16] + hh * 4096));
    s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    f.v = *reinterpret_cast<bf16x8*>(&v);
  }
};
template <int N> struct IntC { static constexpr int value = N; };

template <typename T, int D>
__device__ __forceinline__ void load_frag_T(Frag<T>& f, const char* nat, const char* tr, int c16, int hh, int quad, int l16) {
  if constexpr (sizeof(T) == 2) {
    FragT<D>::bf(f, nat, c16, hh, quad, l16);
  } else {
    constexpr int KSTEPS = Tr<T>::KSTEPS;
    load_frag<T>(f, tr + (hh / KSTEPS) * (D * 128), c16 * 16 + l16, hh % KSTEPS, quad);
  }
}

__device__ __forceinline__ int perm_row(int kt, int r) { return (kt >> 1) * 32 + (r >> 2) * 8 + (kt & 1) * 4 + (r & 3); }

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// OCM (out_cross in memory): the cross softmax runs FIRST and its normalised result goes to p.out_cross; the final
// store re-reads it (same lane, L2-warm) and adds the self part -- so no second accumulator set lives across the
// long self loop (48 registers at d = 96: one more resident wave per SIMD).  Without an out_cross buffer the sum
// is kept in registers.
template <typename T, int D, int QT, bool OCM, bool SPLIT = false>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_fwd_kernel(AttnArgs p) {
  using G = AttnGeom<T, D>;
  using F = typename FragOf<T, SPLIT>::type;   // SPLIT (float only): both matmuls as three bf16 MFMAs on hi / lo halves
  constexpr int KSTEPS = G::KSTEPS, DS = G::DS, DT = G::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + G::NAT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = lane >> 4, l16 = lane & 15;
  const TrOff<D> troff(quad, l16);
  // XCD-aware block order: the hardware deals consecutive workgroups round-robin over the 8 XCDs, so with the natural
  // (tile, head) order the tiles of one (batch, head) -- which all stream the same K / V (or Q / dO) rows -- land on
  // 8 different L2s and fetch them from HBM 8 times (PMC: 1.38 GB per launch at L = 1024 against ~0.4 GB algorithmic).
  // The remap keeps them on one XCD.
  const int bx_ = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = bx_ / (int)gridDim.x, bx = bx_ - by * (int)gridDim.x;
  const int b = by / p.H, h = by - b * p.H;
  const int q0 = bx * (64 * QT) + wave * (16 * QT);

  const T* Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  F qf[QT][DS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = q0 + qt * 16 + l16;
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) frag_from_global_x(qf[qt][ks], Q + (size_t)qi * p.q_rs + ks * 32 + quad * 8, qi < p.L);
  }

  f32x4 res[OCM ? 1 : QT][OCM ? 1 : DT];
  if constexpr (!OCM) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) res[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const float c2 = p.scale * 1.4426950408889634f;   // scores -> base-2 exponent
  const int npass = p.kc ? 2 : 1;
  for (int ipass = 0; ipass < npass; ++ipass) {
    const int pass = npass - 1 - ipass;   // cross keys first, self keys last
    const T* Kp = reinterpret_cast<const T*>(pass ? p.kc : p.k) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const T* Vp = reinterpret_cast<const T*>(pass ? p.vc : p.v) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const int rs = pass ? p.c_rs : p.k_rs;
    const int nk = pass ? p.S : p.L;
    const float* mrow = (pass && p.mask) ? p.mask + (size_t)b * p.S : nullptr;

    f32x4 o[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      m_run[qt] = -1e30f; l_run[qt] = 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // bf16: double-buffered NATURAL K and V tiles (V^T fragments come from the LDS transpose read), the next tile's
    // global loads issued before this tile's MFMAs and committed to the other buffer after them; one barrier / tile.
    // fp32: K natural + V transposed, staged synchronously (ds_read_tr is a 16-bit instruction).
    constexpr bool PF = sizeof(T) == 2;
    NatRegs<T, D> kr, vr;
    const NatSrc<T, D> ksrc(Kp, rs, nk, tid), vsrc(Vp, rs, nk, tid);
    if constexpr (PF) {
      __syncthreads();   // the previous pass may still be reading buffer 0
      fetch_nat<T, D>(kr, ksrc, 0);
      fetch_nat<T, D>(vr, vsrc, 0);
      commit_nat<T, D>(smem, kr, tid);
      commit_nat<T, D>(smem + G::NAT_BYTES, vr, tid);
    }
    // one 64-key tile out of LDS stage CUR (a compile-time constant, so that every fragment address of the tile is
    // `per-lane offset + immediate`: the loop below alternates two instantiations instead of computing addresses)
    auto tile = [&](auto cur_c, const int k0) {
      constexpr int CUR = decltype(cur_c)::value;
      __syncthreads();
      if constexpr (PF) {
        Ks = smem + CUR * (2 * G::NAT_BYTES);
        Vs = Ks + G::NAT_BYTES;
        if (k0 + 64 < nk) {
          fetch_nat<T, D>(kr, ksrc, k0 + 64);
          fetch_nat<T, D>(vr, vsrc, k0 + 64);
        }
      } else {
        load_nat_tile<T, D>(Ks, Kp, rs, k0, nk, tid);
        load_tr_tile<T, D>(Vs, Vp, rs, k0, nk, tid);
        __syncthreads();
      }

      f32x4 s[QT][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        F kf[DS];
        const int row = perm_row(kt, l16);
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) load_frag_x(kf[ks], Ks + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < DS; ++ks) mma16(s[qt][kt], kf[ks], qf[qt][ks]);
        }
      }
      // Softmax in the base-2 domain: t = s * (scale * log2 e), running max m_run and sum l_run per query column;
      // p = 2^(t - m) is one fma + one v_exp_f32.  A full, unmasked tile (the common case) needs no per-key predicate.
      // lane holds key positions (kt>>1)*32 + quad*8 + (kt&1)*4 + i
      const bool full = (k0 + 64 <= nk) && !mrow;   // block-uniform
      if (!full) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int key = k0 + (kt >> 1) * 32 + quad * 8 + (kt & 1) * 4 + i;
            bool ok = key < nk;
            if (ok && mrow) ok = mrow[key] != 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[qt][kt][i] = ok ? s[qt][kt][i] : -3.0e38f;
          }
      }
      F pf[QT][2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[qt][kt][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[qt], fmaxf(mx * c2, -1e30f));   // a fully masked tile leaves m_run alone
        const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
        m_run[qt] = m_new;
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            // masked scores are -3e38: the fma saturates to -inf and 2^-inf = 0
            const float e = __builtin_amdgcn_exp2f(fmaf(s[qt][kt][i], c2, -m_new));
            s[qt][kt][i] = e; ls += e;
          }
        l_run[qt] = l_run[qt] * alpha + ls;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[qt][dt] *= alpha;
        frag_from_acc_x(pf[qt][0], s[qt][0], s[qt][1]);
        frag_from_acc_x(pf[qt][1], s[qt][2], s[qt][3]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          F vf;
          if constexpr (PF) troff.read(vf, Vs, dt, hh);
          else load_frag_x(vf, Vs + (hh / KSTEPS) * (D * 128), dt * 16 + l16, hh % KSTEPS, quad);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) mma16(o[qt][dt], vf, pf[qt][hh]);
        }
      if constexpr (PF) {
        if (k0 + 64 < nk) {
          char* nb = smem + (1 - CUR) * (2 * G::NAT_BYTES);
          commit_nat<T, D>(nb, kr, tid);
          commit_nat<T, D>(nb + G::NAT_BYTES, vr, tid);
        }
      }
    };
    for (int k0 = 0; k0 < nk; k0 += 128) {
      tile(IntC<0>{}, k0);
      if (k0 + 64 < nk) tile(IntC<1>{}, k0 + 64);
    }

    // finalise this softmax
    T* OC = (pass && p.out_cross) ? reinterpret_cast<T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
    float* LSE = pass ? p.lse_cross : p.lse_self;
    T* O = reinterpret_cast<T*>(p.out) + (size_t)b * p.o_bs + (size_t)h * D;
    const T* OCR = (OCM && p.kc) ? reinterpret_cast<const T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
    // OCM, last pass: the cross part of every output piece is requested HERE, before the first store.  Read inside the
    // store loop (rounds 2-5) each piece was load -> wait -> add -> store, and gfx950's single in-order vmcnt makes the wait
    // for a load that follows a store a wait for that store's acknowledgement: QT x DT dependent round trips at the end of
    // every block (12 at d = 96; tools/store_wait_scan.py).
    // (bf16 only: 2 registers per piece.  The fp32 kernels -- parity mode and the bf16x3 sampling mode, 4 registers per
    //  piece -- are at their register limit and keep the read in the store loop.)
    constexpr bool OCB = OCM && sizeof(T) == 2;
    using OcV = typename std::conditional<sizeof(T) == 2, bf16x4, f32x4>::type;
    OcV ocv[OCB ? QT : 1][OCB ? DT : 1];
    if constexpr (OCB) {
      if (pass == 0 && OCR) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int qi = min(q0 + qt * 16 + l16, p.L - 1);     // (rows past the end are not stored; clamped, not predicated)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            ocv[qt][dt] = *reinterpret_cast<const OcV*>(OCR + (size_t)qi * p.o_rs + dt * 16 + quad * 4);
        }
      }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float l = l_run[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = l > 0.f ? 1.f / l : 0.f;
      const int qi = q0 + qt * 16 + l16;
      if (LSE && quad == 0 && qi < p.L) LSE[((size_t)b * p.H + h) * p.L + qi] = m_run[qt] * 0.6931471805599453f + __logf(l);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        f32x4 val = o[qt][dt] * inv;
        if (OC && qi < p.L) {
          T* dst = OC + (size_t)qi * p.o_rs + dt * 16 + quad * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[i] = from_f32<T>(val[i]);
        }
        if constexpr (!OCM) {
          res[qt][dt] += val;
          val = res[qt][dt];
        }
        if (pass == 0 && qi < p.L) {   // last pass: the output row
          T* dst = O + (size_t)qi * p.o_rs + dt * 16 + quad * 4;
          if (OCM && OCR) {
            if constexpr (OCB) {
#pragma unroll
              for (int i = 0; i < 4; ++i) val[i] += (float)ocv[OCB ? qt : 0][OCB ? dt : 0][i];
            } else {
              const T* src = OCR + (size_t)qi * p.o_rs + dt * 16 + quad * 4;
#pragma unroll
              for (int i = 0; i < 4; ++i) val[i] += to_f32(src[i]);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[i] = from_f32<T>(val[i]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward, dQ:   dS^T = P^T o (V dO^T - delta),  dQ^T = K^T dS^T  (both softmaxes accumulate)
// ---------------------------------------------------------------------------------------
template <typename T, int D, int QT>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_bwd_dq_kernel(AttnArgs p) {
  using G = AttnGeom<T, D>;
  constexpr int KSTEPS = G::KSTEPS, DS = G::DS, DT = G::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + G::NAT_BYTES;
  char* KTs = smem + 2 * G::NAT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = lane >> 4, l16 = lane & 15;
  const TrOff<D> troff(quad, l16);
  // XCD-aware block order: the hardware deals consecutive workgroups round-robin over the 8 XCDs, so with the natural
  // (tile, head) order the tiles of one (batch, head) -- which all stream the same K / V (or Q / dO) rows -- land on
  // 8 different L2s and fetch them from HBM 8 times (PMC: 1.38 GB per launch at L = 1024 against ~0.4 GB algorithmic).
  // The remap keeps them on one XCD.
  const int bx_ = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = bx_ / (int)gridDim.x, bx = bx_ - by * (int)gridDim.x;
  const int b = by / p.H, h = by - b * p.H;
  const int q0 = bx * (64 * QT) + wave * (16 * QT);

  const T* Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* DO = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  Frag<T> qf[QT][DS], gf[QT][DS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = q0 + qt * 16 + l16;
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) {
      frag_from_global<T>(qf[qt][ks], Q + (size_t)qi * p.q_rs + ks * 32 + quad * 8, qi < p.L);
      frag_from_global<T>(gf[qt][ks], DO + (size_t)qi * p.o_rs + ks * 32 + quad * 8, qi < p.L);
    }
  }
  f32x4 dq[QT][DT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float c2 = p.scale * 1.4426950408889634f;

  // delta_self = rowsum(dO * (O - O_cross)), delta_cross = rowsum(dO * O_cross) of this block's query rows, from the dO
  // fragments already in registers (a lane holds 8 channels of row l16 per 32-channel step; the four quads of a row
  // are summed with two shuffles).  Kept in registers for this kernel and stored for the dK / dV kernel that follows
  // -- this used to be a separate streaming kernel per layer.
  float del_self[QT], del_cross[QT];
  {
    const T* Op = reinterpret_cast<const T*>(p.out) + (size_t)b * p.o_bs + (size_t)h * D;
    const T* Ocp = p.out_cross ? reinterpret_cast<const T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int qi = q0 + qt * 16 + l16;
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int ks = 0; ks < DS; ++ks) {
        Frag<T> of, cf;
        frag_from_global<T>(of, Op + (size_t)qi * p.o_rs + ks * 32 + quad * 8, qi < p.L);
        frag_from_global<T>(cf, Ocp ? Ocp + (size_t)qi * p.o_rs + ks * 32 + quad * 8 : Op, Ocp != nullptr && qi < p.L);
        float gv[8], ov[8], cv[8];
        frag_to_f32<T>(gf[qt][ks], gv);
        frag_to_f32<T>(of, ov);
        frag_to_f32<T>(cf, cv);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a += gv[e] * (ov[e] - cv[e]); c += gv[e] * cv[e]; }
      }
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
      del_self[qt] = a; del_cross[qt] = c;
      if (quad == 0 && qi < p.L) {
        const size_t o = ((size_t)b * p.H + h) * p.L + qi;
        p.delta_self_w[o] = a;
        if (p.delta_cross_w) p.delta_cross_w[o] = c;
      }
    }
  }

  const int npass = p.kc ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const T* Kp = reinterpret_cast<const T*>(pass ? p.kc : p.k) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const T* Vp = reinterpret_cast<const T*>(pass ? p.vc : p.v) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const int rs = pass ? p.c_rs : p.k_rs;
    const int nk = pass ? p.S : p.L;
    const float* mrow = (pass && p.mask) ? p.mask + (size_t)b * p.S : nullptr;
    const float* LSE = (pass ? p.lse_cross : p.lse_self) + ((size_t)b * p.H + h) * p.L;
    float lse[QT], del[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int qi = q0 + qt * 16 + l16;
      lse[qt] = (qi < p.L ? LSE[qi] : 1e30f) * 1.4426950408889634f;   // base-2 domain: p = 2^(s * c2 - lse2)
      del[qt] = pass ? del_cross[qt] : del_self[qt];
    }
    // bf16: double-buffered K / V tiles with the next tile's global loads in flight during the MFMAs (see forward)
    constexpr bool PF = sizeof(T) == 2;
    NatRegs<T, D> kr, vr;
    const NatSrc<T, D> ksrc(Kp, rs, nk, tid), vsrc(Vp, rs, nk, tid);
    if constexpr (PF) {
      __syncthreads();
      fetch_nat<T, D>(kr, ksrc, 0);
      fetch_nat<T, D>(vr, vsrc, 0);
      commit_nat<T, D>(smem, kr, tid);
      commit_nat<T, D>(smem + G::NAT_BYTES, vr, tid);
    }
    // one 64-key tile out of LDS stage CUR (compile-time: fragment addresses = per-lane offset + immediate; see forward)
    auto tile = [&](auto cur_c, const int k0) {
      constexpr int CUR = decltype(cur_c)::value;
      __syncthreads();
      if constexpr (PF) {
        Ks = smem + CUR * (2 * G::NAT_BYTES);
        Vs = Ks + G::NAT_BYTES;
        if (k0 + 64 < nk) {
          fetch_nat<T, D>(kr, ksrc, k0 + 64);
          fetch_nat<T, D>(vr, vsrc, k0 + 64);
        }
      } else {
        load_nat_tile<T, D>(Ks, Kp, rs, k0, nk, tid);
        load_nat_tile<T, D>(Vs, Vp, rs, k0, nk, tid);
        load_tr_tile<T, D>(KTs, Kp, rs, k0, nk, tid);
        __syncthreads();
      }
      f32x4 s[QT][4], dp[QT][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int row = perm_row(kt, l16);
        Frag<T> kf[DS], vf[DS];
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
          load_frag<T>(kf[ks], Ks + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
          load_frag<T>(vf[ks], Vs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          s[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < DS; ++ks) {
            mma16(s[qt][kt], kf[ks], qf[qt][ks]);
            mma16(dp[qt][kt], vf[ks], gf[qt][ks]);
          }
        }
      }
      if ((k0 + 64 <= nk) && !mrow) {   // full, unmasked tile (block-uniform): no per-key predicate
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
              const float pr = __builtin_amdgcn_exp2f(fmaf(s[qt][kt][i], c2, -lse[qt]));
              s[qt][kt][i] = pr * (dp[qt][kt][i] - del[qt]);
            }
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int key = k0 + (kt >> 1) * 32 + quad * 8 + (kt & 1) * 4 + i;
            bool ok = key < nk;
            if (ok && mrow) ok = mrow[key] != 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
              const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(s[qt][kt][i], c2, -lse[qt])) : 0.f;
              s[qt][kt][i] = pr * (dp[qt][kt][i] - del[qt]);
            }
          }
      }
      Frag<T> dsf[QT][2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        frag_from_acc<T>(dsf[qt][0], s[qt][0], s[qt][1]);
        frag_from_acc<T>(dsf[qt][1], s[qt][2], s[qt][3]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          Frag<T> ktf;
          if constexpr (PF) troff.read(ktf, Ks, dt, hh);
          else load_frag_T<T, D>(ktf, Ks, KTs, dt, hh, quad, l16);
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) mma16(dq[qt][dt], ktf, dsf[qt][hh]);
        }
      if constexpr (PF) {
        if (k0 + 64 < nk) {
          char* nb = smem + (1 - CUR) * (2 * G::NAT_BYTES);
          commit_nat<T, D>(nb, kr, tid);
          commit_nat<T, D>(nb + G::NAT_BYTES, vr, tid);
        }
      }
    };
    for (int k0 = 0; k0 < nk; k0 += 128) {
      tile(IntC<0>{}, k0);
      if (k0 + 64 < nk) tile(IntC<1>{}, k0 + 64);
    }
  }
  T* DQ = reinterpret_cast<T*>(p.dq) + (size_t)b * p.q_bs + (size_t)h * D;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = q0 + qt * 16 + l16;
    if (qi >= p.L) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      T* dst = DQ + (size_t)qi * p.q_rs + dt * 16 + quad * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = from_f32<T>(dq[qt][dt][i] * p.scale);
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward, dK / dV of one key set (self or cross).  Block = 64 keys (16 per wave), loops over
// 64-query tiles staged in LDS (natural + transposed images of Q and dO).
//   S = Q K^T (rows = queries, permuted), dP = dO V^T, dS = P o (dP - delta)
//   dV^T = dO^T P,  dK^T = Q^T dS
// ---------------------------------------------------------------------------------------
template <typename T, int D, int KT>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnArgs p) {
  // KT = 16-key tiles per wave: the block owns 64 * KT keys; the Q / dO fragments of a query tile are read from LDS
  // once and reused for every key tile of the wave (KT = 2 halves the tile loads and LDS reads per key)
  using G = AttnGeom<T, D>;
  constexpr int KSTEPS = G::KSTEPS, DS = G::DS, DT = G::DT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Gs = smem + G::NAT_BYTES;
  char* QTs = smem + 2 * G::NAT_BYTES;
  char* GTs = QTs + G::TR_BYTES;
  float* lse_s = reinterpret_cast<float*>(GTs + G::TR_BYTES);
  float* del_s = lse_s + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = lane >> 4, l16 = lane & 15;
  const TrOff<D> troff(quad, l16);
  // XCD-aware block order: the hardware deals consecutive workgroups round-robin over the 8 XCDs, so with the natural
  // (tile, head) order the tiles of one (batch, head) -- which all stream the same K / V (or Q / dO) rows -- land on
  // 8 different L2s and fetch them from HBM 8 times (PMC: 1.38 GB per launch at L = 1024 against ~0.4 GB algorithmic).
  // The remap keeps them on one XCD.
  const int bx_ = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  const int by = bx_ / (int)gridDim.x;
  int bx = bx_ - by * (int)gridDim.x;
  const int b = by / p.H, h = by - b * p.H;
  // one launch covers both key sets: the first ceil(L / (64 KT)) blocks of a (batch, head) own self keys, the rest
  // the text (cross) keys
  const int nself = (p.L + 64 * KT - 1) / (64 * KT);
  const int pass = bx >= nself ? 1 : 0;
  if (pass) bx -= nself;
  const int nk = pass ? p.S : p.L;

  const T* Kp = reinterpret_cast<const T*>(pass ? p.kc : p.k) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
  const T* Vp = reinterpret_cast<const T*>(pass ? p.vc : p.v) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
  const int rs = pass ? p.c_rs : p.k_rs;
  int key[KT];
  bool key_ok[KT], key_live[KT];
  bool tile_live[KT];   // wave-uniform: this 16-key tile holds at least one real key (a dead tile skips its MFMAs)
  Frag<T> kf[KT][DS], vf[KT][DS];
#pragma unroll
  for (int kk = 0; kk < KT; ++kk) {
    key[kk] = bx * (64 * KT) + wave * (16 * KT) + kk * 16 + l16;
    key_ok[kk] = key[kk] < nk;
    tile_live[kk] = __builtin_amdgcn_readfirstlane(bx * (64 * KT) + wave * (16 * KT) + kk * 16) < nk;
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) {
      frag_from_global<T>(kf[kk][ks], Kp + (size_t)key[kk] * rs + ks * 32 + quad * 8, key_ok[kk]);
      frag_from_global<T>(vf[kk][ks], Vp + (size_t)key[kk] * rs + ks * 32 + quad * 8, key_ok[kk]);
    }
    key_live[kk] = key_ok[kk];
    if (key_ok[kk] && pass && p.mask) key_live[kk] = p.mask[(size_t)b * p.S + key[kk]] != 0.f;
  }

  const T* Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* DO = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  const float* LSE = (pass ? p.lse_cross : p.lse_self) + ((size_t)b * p.H + h) * p.L;
  const float* DEL = (pass ? p.delta_cross : p.delta_self) + ((size_t)b * p.H + h) * p.L;

  const float c2 = p.scale * 1.4426950408889634f;
  f32x4 dk[KT][DT], dv[KT][DT];
#pragma unroll
  for (int kk = 0; kk < KT; ++kk)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // bf16: double-buffered Q / dO tiles (+ their lse / delta rows), next tile's global loads in flight during the MFMAs
  constexpr bool PF = sizeof(T) == 2;
  constexpr int BUF = 2 * G::NAT_BYTES + 512;   // one stage: Q, dO, lse[64], delta[64]
  NatRegs<T, D> qr, gr;
  const NatSrc<T, D> qsrc(Q, p.q_rs, p.L, tid), gsrc(DO, p.o_rs, p.L, tid);
  float lse_r = 0.f, del_r = 0.f;
  if constexpr (PF) {
    fetch_nat<T, D>(qr, qsrc, 0);
    fetch_nat<T, D>(gr, gsrc, 0);
    commit_nat<T, D>(smem, qr, tid);
    commit_nat<T, D>(smem + G::NAT_BYTES, gr, tid);
    if (tid < 64) {
      float* ls = reinterpret_cast<float*>(smem + 2 * G::NAT_BYTES);
      ls[tid] = (tid < p.L ? LSE[tid] : 1e30f) * 1.4426950408889634f;
      ls[64 + tid] = tid < p.L ? DEL[tid] : 0.f;
    }
  }
  // one 64-query tile out of LDS stage CUR (compile-time: fragment addresses = per-lane offset + immediate; see forward)
  auto tile = [&](auto cur_c, const int q0) {
    constexpr int CUR = decltype(cur_c)::value;
    __syncthreads();
    if constexpr (PF) {
      Qs = smem + CUR * BUF;
      Gs = Qs + G::NAT_BYTES;
      lse_s = reinterpret_cast<float*>(Qs + 2 * G::NAT_BYTES);
      del_s = lse_s + 64;
      if (q0 + 64 < p.L) {
        fetch_nat<T, D>(qr, qsrc, q0 + 64);
        fetch_nat<T, D>(gr, gsrc, q0 + 64);
        if (tid < 64) {
          const int qi = q0 + 64 + tid;
          lse_r = (qi < p.L ? LSE[qi] : 1e30f) * 1.4426950408889634f;
          del_r = qi < p.L ? DEL[qi] : 0.f;
        }
      }
    } else {
      load_nat_tile<T, D>(Qs, Q, p.q_rs, q0, p.L, tid);
      load_nat_tile<T, D>(Gs, DO, p.o_rs, q0, p.L, tid);
      load_tr_tile<T, D>(QTs, Q, p.q_rs, q0, p.L, tid);
      load_tr_tile<T, D>(GTs, DO, p.o_rs, q0, p.L, tid);
      if (tid < 64) {
        const int qi = q0 + tid;
        lse_s[tid] = (qi < p.L ? LSE[qi] : 1e30f) * 1.4426950408889634f;   // base-2 domain
        del_s[tid] = qi < p.L ? DEL[qi] : 0.f;
      }
      __syncthreads();
    }
    // The 64 staged queries go through in two halves of 32 (hh = the (kt >> 1) half of perm_row, which is also the
    // reduction half of the transposed fragments): only two score tiles per key tile are live at a time, which is
    // what lets a wave own KT = 2 key tiles -- every Q / dO fragment read from LDS then feeds two MFMAs.
    // A wave without a single real key skips the arithmetic of the tile (one branch around the whole body); a dead
    // SECOND tile is computed anyway -- its K / V fragments are zeros and its keys are masked, so it adds nothing.  A
    // branch per MFMA pair (round 3) cut the loop into ~60 basic blocks of two MFMAs each; as one region the body is
    // 64 MFMAs + 204 VALU + 56 LDS reads that the scheduler interleaves -- measured time unchanged (0.87 ms at
    // L = 1024, d = 64, batch 64): the kernel is bound by its LDS fragment traffic and barriers, not by issue order.
    if (tile_live[0]) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      f32x4 s[KT][2], dp[KT][2];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int row = perm_row(hh * 2 + k2, l16);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) { s[kk][k2] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kk][k2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < DS; ++ks) {
          Frag<T> a, g;
          load_frag<T>(a, Qs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
          load_frag<T>(g, Gs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
#pragma unroll
          for (int kk = 0; kk < KT; ++kk) {
            mma16(s[kk][k2], a, kf[kk][ks]);
            mma16(dp[kk][k2], g, vf[kk][ks]);
          }
        }
      }
      // lane: key = l16 (column), query positions hh*32 + quad*8 + k2*4 + i
      Frag<T> pf[KT], dsf[KT];
#pragma unroll
      for (int kk = 0; kk < KT; ++kk) {
        f32x4 pr[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + hh * 32 + quad * 8 + k2 * 4);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + hh * 32 + quad * 8 + k2 * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pv = key_live[kk] ? __builtin_amdgcn_exp2f(fmaf(s[kk][k2][i], c2, -l4[i])) : 0.f;
            pr[k2][i] = pv;
            s[kk][k2][i] = pv * (dp[kk][k2][i] - d4[i]);
          }
        }
        frag_from_acc<T>(pf[kk], pr[0], pr[1]);
        frag_from_acc<T>(dsf[kk], s[kk][0], s[kk][1]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        Frag<T> a, g;
        if constexpr (PF) {
          troff.read(g, Gs, dt, hh);
          troff.read(a, Qs, dt, hh);
        } else {
          load_frag_T<T, D>(g, Gs, GTs, dt, hh, quad, l16);
          load_frag_T<T, D>(a, Qs, QTs, dt, hh, quad, l16);
        }
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
          mma16(dv[kk][dt], g, pf[kk]);
          mma16(dk[kk][dt], a, dsf[kk]);
        }
      }
    }
    }
    if constexpr (PF) {
      if (q0 + 64 < p.L) {
        char* nb = smem + (1 - CUR) * BUF;
        commit_nat<T, D>(nb, qr, tid);
        commit_nat<T, D>(nb + G::NAT_BYTES, gr, tid);
        if (tid < 64) {
          float* ls = reinterpret_cast<float*>(nb + 2 * G::NAT_BYTES);
          ls[tid] = lse_r;
          ls[64 + tid] = del_r;
        }
      }
    }
  };
  for (int q0 = 0; q0 < p.L; q0 += 128) {
    tile(IntC<0>{}, q0);
    if (q0 + 64 < p.L) tile(IntC<1>{}, q0 + 64);
  }
#pragma unroll
  for (int kk = 0; kk < KT; ++kk) {
    if (!key_ok[kk]) continue;
    T* DK = reinterpret_cast<T*>(pass ? p.dkc : p.dk) + (size_t)b * (pass ? p.dkc_bs : p.dk_bs) + (size_t)h * D + (size_t)key[kk] * (pass ? p.dkc_rs : p.dk_rs);
    T* DV = reinterpret_cast<T*>(pass ? p.dvc : p.dv) + (size_t)b * (pass ? p.dkc_bs : p.dk_bs) + (size_t)h * D + (size_t)key[kk] * (pass ? p.dkc_rs : p.dk_rs);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        DK[dt * 16 + quad * 4 + i] = from_f32<T>(dk[kk][dt][i] * p.scale);
        DV[dt * 16 + quad * 4 + i] = from_f32<T>(dv[kk][dt][i]);
      }
  }
}


// ---------------------------------------------------------------------------------------
// backward for SHORT sequences (bf16, L <= 256 queries, S <= 64 text keys): one block per (batch, head).
// At the 16x16 level of the 64x64 U-Net (26 of its 31 attention layers: L = 256, d = 96) the two streaming kernels
// above run 4 key / query tiles per block: every tile costs a global fetch, an LDS commit and two barriers for ~36
// MFMAs per wave, and the K / V (Q / dO) rows of a head are re-streamed by each of its 4 + 5 blocks -- 0.13 PF.  Here a
// head's operands become LDS-resident once per phase and the tile loops run without a barrier:
//   phase 0  Q and dO tiles of the head -> LDS; lse (base 2) and delta = rowsum(dO * O) rows -> LDS
//   phase 1  dK / dV: a wave owns 16 keys (K, V fragments in registers), walks the resident query tiles
//            (the inner loop of attn_bwd_dkv_kernel, KT = 1); 128 keys per pass, text keys = one more pass
//   phase 2  dQ: K / V tiles -> LDS (<= 3 key tiles at a time, the text tile is one of them), a wave owns 16 queries
//            (Q, dO fragments in registers; the inner loop of attn_bwd_dq_kernel, QT = 1); 128 queries per pass
// Same arithmetic, operand layouts and rounding points as the two-kernel path (which stays for long sequences and fp32).
// ---------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512, 1) void attn_bwd_small_kernel(AttnArgs p) {
  using T = bf16;
  using G = AttnGeom<T, D>;
  constexpr int KSTEPS = G::KSTEPS, DS = G::DS, DT = G::DT, CPR = G::CPR, NAT = G::NAT_BYTES;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const lse_self = reinterpret_cast<float*>(smem + 8 * NAT);   // [256] each, base-2 domain
  float* const lse_cross = lse_self + 256;
  float* const del_self = lse_cross + 256;
  float* const del_cross = del_self + 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int quad = lane >> 4, l16 = lane & 15;
  const TrOff<D> troff(quad, l16);
  // The heads of one batch element read neighbouring 2 d-byte pieces of the same qkv / out rows: with d = 96 a head's
  // piece (192 B) shares its 128-byte lines with the next head's, so heads on different XCDs fetch those lines twice
  // (measured: phase 0 at 1.6 TB/s).  The remap puts consecutive (batch, head) pairs on one XCD, where the L2 merges them.
  const int bh = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int b = bh / p.H, h = bh - b * p.H;
  const T* Q = reinterpret_cast<const T*>(p.q) + (size_t)b * p.q_bs + (size_t)h * D;
  const T* DO = reinterpret_cast<const T*>(p.dout) + (size_t)b * p.o_bs + (size_t)h * D;
  const bool has_c = p.kc != nullptr;
  const int nq = (p.L + 63) >> 6;             // query tiles = self key tiles (<= 4)
  const float c2 = p.scale * LOG2E;

  // `ntile` 64-row tiles (from tile index `tile0`) of two head views -> natural LDS tiles in slots slotA.. / slotB..
  // (rows past `nrows` are zeros).  Every global load of the call is in flight before the first LDS store: one memory
  // round trip per call, not one per tile.
  constexpr int PER = 64 * CPR;                     // 16-byte chunks of one tile
  auto stage2 = [&](const T* A, int a_rs, const T* Bp, int b_rs, int tile0, int ntile, int nrows, int slotA, int slotB) {
    uint4 v[CPR];                                   // 2 x 4 x PER / 512 chunks per thread at most
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
      const int c = tid + i * 512;
      v[i] = uint4{0u, 0u, 0u, 0u};
      if (c < 2 * ntile * PER) {
        const int which = c >= ntile * PER ? 1 : 0;
        const int c2 = c - which * ntile * PER;
        const int t = c2 / PER, rem = c2 - t * PER;
        const int row = rem / CPR, cc = rem - row * CPR;
        const int r = (tile0 + t) * 64 + row;
        if (r < nrows) v[i] = *reinterpret_cast<const uint4*>((which ? Bp : A) + (size_t)r * (which ? b_rs : a_rs) + cc * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < CPR; ++i) {
      const int c = tid + i * 512;
      if (c < 2 * ntile * PER) {
        const int which = c >= ntile * PER ? 1 : 0;
        const int c2 = c - which * ntile * PER;
        const int t = c2 / PER, rem = c2 - t * PER;
        const int row = rem / CPR, cc = rem - row * CPR;
        *reinterpret_cast<uint4*>(smem + ((which ? slotB : slotA) + t) * NAT + (cc >> 3) * (64 * 128) + lds_chunk_off(row, cc & 7)) = v[i];
      }
    }
  };

  // ---- phase 0 ------------------------------------------------------------------------------------------------
  stage2(Q, p.q_rs, DO, p.o_rs, 0, nq, p.L, 0, 4);
  {
    // delta_self = rowsum(dO * (O - O_cross)), delta_cross = rowsum(dO * O_cross): two threads per query row; O and
    // O_cross come from global memory (issued before the barrier), dO from the tile just staged
    const T* Op = reinterpret_cast<const T*>(p.out) + (size_t)b * p.o_bs + (size_t)h * D;
    const T* Ocp = p.out_cross ? reinterpret_cast<const T*>(p.out_cross) + (size_t)b * p.o_bs + (size_t)h * D : nullptr;
    const int qi = tid >> 1, half = tid & 1;
    float a = 0.f, c = 0.f;
    Chunk<T> ov[CPR / 2], cv_[CPR / 2];
    if (qi < p.L) {
#pragma unroll
      for (int j = 0; j < CPR / 2; ++j) {
        const size_t off = (size_t)qi * p.o_rs + (half * (CPR / 2) + j) * 8;
        ov[j].load(Op + off);
        if (Ocp) cv_[j].load(Ocp + off);
      }
    }
    __syncthreads();   // the staged dO tiles are visible
    if (qi < p.L) {
      const char* Gt = smem + (4 + (qi >> 6)) * NAT;
#pragma unroll
      for (int j = 0; j < CPR / 2; ++j) {
        const int cc = half * (CPR / 2) + j;
        Chunk<T> g;
        g.load(reinterpret_cast<const T*>(Gt + (cc >> 3) * (64 * 128) + lds_chunk_off(qi & 63, cc & 7)));
        const Chunk<T>& o = ov[j];
        const Chunk<T>& oc = cv_[j];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float cv = Ocp ? oc.v[e] : 0.f;
          a += g.v[e] * (o.v[e] - cv);
          c += g.v[e] * cv;
        }
      }
    }
    a += __shfl_xor(a, 1, 64);
    c += __shfl_xor(c, 1, 64);
    if (half == 0) {
      const size_t lo = ((size_t)b * p.H + h) * p.L + qi;
      const bool ok = qi < p.L;
      lse_self[qi] = ok ? p.lse_self[lo] * LOG2E : 1e30f;
      lse_cross[qi] = (ok && has_c) ? p.lse_cross[lo] * LOG2E : 1e30f;
      del_self[qi] = ok ? a : 0.f;
      del_cross[qi] = ok ? c : 0.f;
    }
  }
  // ---- phase 1: dK / dV ---------------------------------------------------------------------------------------------
  // KT = 16-key tiles per wave (as in attn_bwd_dkv_kernel): with 2, one pass covers 256 self keys, every Q / dO fragment
  // read from LDS feeds two MFMAs and a wave has two independent MFMA chains in flight
  constexpr int KT = D <= 64 ? 2 : 1;          // (at d = 96 two tiles spill: 166 us against 158 at L = 256, batch 64)
  constexpr int GK = 128 * KT;                      // keys per pass over the query tiles
  const int nself_g = (p.L + GK - 1) / GK, ncross_g = has_c ? (p.S + GK - 1) / GK : 0;
  __syncthreads();
  for (int grp = 0; grp < nself_g + ncross_g; ++grp) {
    const int pass = grp >= nself_g ? 1 : 0;
    const int g0 = (pass ? grp - nself_g : grp) * GK + wave * (16 * KT);
    const int nk = pass ? p.S : p.L;
    if (__builtin_amdgcn_readfirstlane(g0) >= nk) continue;   // wave-uniform: no real key in this wave's tiles
    const T* Kp = reinterpret_cast<const T*>(pass ? p.kc : p.k) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const T* Vp = reinterpret_cast<const T*>(pass ? p.vc : p.v) + (size_t)b * (pass ? p.c_bs : p.k_bs) + (size_t)h * D;
    const int rs = pass ? p.c_rs : p.k_rs;
    int key[KT];
    bool key_ok[KT], key_live[KT];
    Frag<T> kf[KT][DS], vf[KT][DS];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      key[kk] = g0 + kk * 16 + l16;
      key_ok[kk] = key[kk] < nk;
#pragma unroll
      for (int ks = 0; ks < DS; ++ks) {
        frag_from_global<T>(kf[kk][ks], Kp + (size_t)(key_ok[kk] ? key[kk] : 0) * rs + ks * 32 + quad * 8, key_ok[kk]);
        frag_from_global<T>(vf[kk][ks], Vp + (size_t)(key_ok[kk] ? key[kk] : 0) * rs + ks * 32 + quad * 8, key_ok[kk]);
      }
      key_live[kk] = key_ok[kk];
      if (key_ok[kk] && pass && p.mask) key_live[kk] = p.mask[(size_t)b * p.S + key[kk]] != 0.f;
    }
    const float* lse_a = pass ? lse_cross : lse_self;
    const float* del_a = pass ? del_cross : del_self;
    f32x4 dk[KT][DT], dv[KT][DT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { dk[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int qt = 0; qt < nq; ++qt) {
      const char* Qs = smem + qt * NAT;
      const char* Gs = smem + (4 + qt) * NAT;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        f32x4 s[KT][2], dp[KT][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int row = perm_row(hh * 2 + k2, l16);
#pragma unroll
          for (int kk = 0; kk < KT; ++kk) { s[kk][k2] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kk][k2] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int ks = 0; ks < DS; ++ks) {
            Frag<T> a, g;
            load_frag<T>(a, Qs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
            load_frag<T>(g, Gs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {   // (a dead second tile holds zero fragments and masked keys: no branch)
              mma16(s[kk][k2], a, kf[kk][ks]);
              mma16(dp[kk][k2], g, vf[kk][ks]);
            }
          }
        }
        // lane: key = l16 (column), query positions qt*64 + hh*32 + quad*8 + k2*4 + i
        Frag<T> pf[KT], dsf[KT];
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
          f32x4 pr[2];
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_a + qt * 64 + hh * 32 + quad * 8 + k2 * 4);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_a + qt * 64 + hh * 32 + quad * 8 + k2 * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float pv = key_live[kk] ? __builtin_amdgcn_exp2f(fmaf(s[kk][k2][i], c2, -l4[i])) : 0.f;
              pr[k2][i] = pv;
              s[kk][k2][i] = pv * (dp[kk][k2][i] - d4[i]);
            }
          }
          frag_from_acc<T>(pf[kk], pr[0], pr[1]);
          frag_from_acc<T>(dsf[kk], s[kk][0], s[kk][1]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          Frag<T> a, g;
          troff.read(g, Gs, dt, hh);
          troff.read(a, Qs, dt, hh);
#pragma unroll
          for (int kk = 0; kk < KT; ++kk) {
            mma16(dv[kk][dt], g, pf[kk]);
            mma16(dk[kk][dt], a, dsf[kk]);
          }
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      if (!key_ok[kk]) continue;
      T* DK = reinterpret_cast<T*>(pass ? p.dkc : p.dk) + (size_t)b * (pass ? p.dkc_bs : p.dk_bs) + (size_t)h * D + (size_t)key[kk] * (pass ? p.dkc_rs : p.dk_rs);
      T* DV = reinterpret_cast<T*>(pass ? p.dvc : p.dv) + (size_t)b * (pass ? p.dkc_bs : p.dk_bs) + (size_t)h * D + (size_t)key[kk] * (pass ? p.dkc_rs : p.dk_rs);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          DK[dt * 16 + quad * 4 + i] = from_f32<T>(dk[kk][dt][i] * p.scale);
          DV[dt * 16 + quad * 4 + i] = from_f32<T>(dv[kk][dt][i]);
        }
    }
  }

  // ---- phase 2: dQ --------------------------------------------------------------------------------------------------
  const int nqp = (p.L + 127) >> 7;           // query passes of 128 (16 per wave)
  f32x4 dq[2][DT];
#pragma unroll
  for (int qp = 0; qp < 2; ++qp)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[qp][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Q / dO fragments of the wave's 16 queries of either pass: fetched once, before the K / V tiles are staged
  Frag<T> qf2[2][DS], gf2[2][DS];
#pragma unroll
  for (int qp = 0; qp < 2; ++qp) {
    const int qi = qp * 128 + wave * 16 + l16;
    const bool ok = qp < nqp && qi < p.L;
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) {
      frag_from_global<T>(qf2[qp][ks], ok ? Q + (size_t)qi * p.q_rs + ks * 32 + quad * 8 : Q, ok);
      frag_from_global<T>(gf2[qp][ks], ok ? DO + (size_t)qi * p.o_rs + ks * 32 + quad * 8 : DO, ok);
    }
  }
  // key tiles in two groups of at most three (K in slots 0-2, V in slots 3-5): {self 0, self 1, text}, {self 2, self 3}
  const T* Kself = reinterpret_cast<const T*>(p.k) + (size_t)b * p.k_bs + (size_t)h * D;
  const T* Vself = reinterpret_cast<const T*>(p.v) + (size_t)b * p.k_bs + (size_t)h * D;
  for (int grp = 0; grp < (nq > 2 ? 2 : 1); ++grp) {
    const int ts = grp ? nq - 2 : min(nq, 2);                 // self tiles of this group
    const int tn = ts + ((grp == 0 && has_c) ? 1 : 0);
    __syncthreads();   // every wave is done with what the LDS held (phase 1 / the previous group)
    stage2(Kself, p.k_rs, Vself, p.k_rs, grp * 2, ts, p.L, 0, 3);
    if (grp == 0 && has_c) {
      const T* Kc = reinterpret_cast<const T*>(p.kc) + (size_t)b * p.c_bs + (size_t)h * D;
      const T* Vc = reinterpret_cast<const T*>(p.vc) + (size_t)b * p.c_bs + (size_t)h * D;
      stage2(Kc, p.c_rs, Vc, p.c_rs, 0, 1, p.S, ts, 3 + ts);
    }
    __syncthreads();
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      const int q0 = qp * 128 + wave * 16;
      if (qp >= nqp || __builtin_amdgcn_readfirstlane(q0) >= p.L) continue;
      const int qi = q0 + l16;
      const Frag<T>(&qf)[DS] = qf2[qp];
      const Frag<T>(&gf)[DS] = gf2[qp];
      for (int j = 0; j < tn; ++j) {
        const bool cross = j >= ts;
        const int k0 = cross ? 0 : (grp * 2 + j) * 64, nk = cross ? p.S : p.L;
        const float* mrow = (cross && p.mask) ? p.mask + (size_t)b * p.S : nullptr;
        const float lse = (cross ? lse_cross : lse_self)[qi & 255];
        const float del = (cross ? del_cross : del_self)[qi & 255];
        const char* Ks = smem + j * NAT;
        const char* Vs = smem + (3 + j) * NAT;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int row = perm_row(kt, l16);
          Frag<T> kf[DS], vf[DS];
#pragma unroll
          for (int ks = 0; ks < DS; ++ks) {
            load_frag<T>(kf[ks], Ks + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
            load_frag<T>(vf[ks], Vs + (ks / KSTEPS) * (64 * 128), row, ks % KSTEPS, quad);
          }
          s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < DS; ++ks) {
            mma16(s[kt], kf[ks], qf[ks]);
            mma16(dp[kt], vf[ks], gf[ks]);
          }
        }
        // (the variant test stays between the score MFMAs and the softmax here: as one straight-line region per variant,
        // as in attn_bwd_dq_kernel, this phase spills 41 registers at d = 96)
        if ((k0 + 64 <= nk) && !mrow) {   // full, unmasked tile (block-uniform): no per-key predicate
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float pr = __builtin_amdgcn_exp2f(fmaf(s[kt][i], c2, -lse));
              s[kt][i] = pr * (dp[kt][i] - del);
            }
        } else {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int key = k0 + (kt >> 1) * 32 + quad * 8 + (kt & 1) * 4 + i;
              bool ok = key < nk;
              if (ok && mrow) ok = mrow[key] != 0.f;
              const float pr = ok ? __builtin_amdgcn_exp2f(fmaf(s[kt][i], c2, -lse)) : 0.f;
              s[kt][i] = pr * (dp[kt][i] - del);
            }
        }
        Frag<T> dsf[2];
        frag_from_acc<T>(dsf[0], s[0], s[1]);
        frag_from_acc<T>(dsf[1], s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            Frag<T> ktf;
            troff.read(ktf, Ks, dt, hh);
            mma16(dq[qp][dt], ktf, dsf[hh]);
          }
      }
    }
  }
  T* DQ = reinterpret_cast<T*>(p.dq) + (size_t)b * p.q_bs + (size_t)h * D;
#pragma unroll
  for (int qp = 0; qp < 2; ++qp) {
    const int qi = qp * 128 + wave * 16 + l16;
    if (qp >= nqp || qi >= p.L) continue;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      T* dst = DQ + (size_t)qi * p.q_rs + dt * 16 + quad * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = from_f32<T>(dq[qp][dt][i] * p.scale);
    }
  }
}

}  // namespace mdm

#include "attn32.hpp"   // the backward on v_mfma_f32_32x32x16_bf16 (a wave owns 32 keys / queries)

using namespace mdm;

template <typename T, int D, bool SPLIT = false>
static int attn_fwd_launch(const AttnArgs& a, hipStream_t st) {
  using G = AttnGeom<T, D>;
  constexpr int smem = sizeof(T) == 2 ? 4 * G::NAT_BYTES : G::NAT_BYTES + (G::NAT_BYTES > G::TR_BYTES ? G::NAT_BYTES : G::TR_BYTES);
  ensure_dynamic_lds(attn_fwd_kernel<T, D, 2, true, SPLIT>, smem);
  // (the entry point requires out_cross whenever there are text keys: the cross part is staged through it, so no second
  //  accumulator set lives across the self loop -- the register-resident form spilled at d = 96 and was deleted in round 6)
  // Small batch (sampling at batch 1-4: 64 blocks of 128 queries at L = 256, batch 4, on 256 CUs): 16 queries per wave
  // instead of 32 -- twice the blocks, half the serial work of each; the grid is what is short there, not the LDS reads
  // per MFMA that 32 queries per wave save.
  if constexpr (sizeof(T) == 2 && !SPLIT) {
    const long blocks2 = (long)((a.L + 127) / 128) * a.B * a.H;
    // (not for a training batch at L = 256: 58 against 47-52 us at d = 96, batch 64 -- three blocks per CU instead of two do
    //  not make up for one MFMA per K / V fragment read; profiles/r06_did_not_pay.md)
    if (blocks2 * 2 <= device_cus() && a.L > 64) {
      ensure_dynamic_lds(attn_fwd_kernel<T, D, 1, true, SPLIT>, smem);
      dim3 grid1((a.L + 63) / 64, a.B * a.H);
      hipLaunchKernelGGL((attn_fwd_kernel<T, D, 1, true, SPLIT>), grid1, dim3(256), smem, st, a);
      MDM_LAUNCH_STATUS();
    }
  }
  dim3 grid((a.L + 127) / 128, a.B * a.H);   // 32 queries per wave: every K / V fragment feeds two MFMAs
  hipLaunchKernelGGL((attn_fwd_kernel<T, D, 2, true, SPLIT>), grid, dim3(256), smem, st, a);
  MDM_LAUNCH_STATUS();
}

// development knob (mdm_hip_dev.h), read by mdm_attn_bwd's kernel choice; a plain int set from the host thread that
// drives the experiment -- no environment lookups inside the boundary's entry points
static int g_attn_bwd_mode = 0;
extern "C" int mdm_dev_set_attn_bwd(int mode) {
  if (mode < 0 || mode > 5) return -1;
  g_attn_bwd_mode = mode;
  return 0;
}

// development aid: a device buffer of [blocks][8 waves][16] 64-bit shader-clock stamps written by attn_bwd_small32_kernel
static unsigned long long* g_attn_dbg = nullptr;
extern "C" int mdm_dev_set_attn_dbg(void* buf) {
  g_attn_dbg = reinterpret_cast<unsigned long long*>(buf);
  return 0;
}

template <typename T, int D>
static int attn_bwd_launch(AttnArgs a, void* dkc, void* dvc, size_t dc_bs, int dc_rs, hipStream_t st) {
  using G = AttnGeom<T, D>;
  a.dbg = g_attn_dbg;
  constexpr int smem_q = sizeof(T) == 2 ? 4 * G::NAT_BYTES : 2 * G::NAT_BYTES + G::TR_BYTES;
  constexpr int smem_kv = sizeof(T) == 2 ? 2 * (2 * G::NAT_BYTES + 512) : 2 * G::NAT_BYTES + 2 * G::TR_BYTES + 512;
  // key tiles per wave of the dK / dV kernel: 2 where the registers allow it (the kernel is LDS-bandwidth bound: PMC
  // showed 2.6 LDS instructions per MFMA at KT = 1, and every Q / dO fragment read feeds KT MFMAs)
  constexpr int KV_KT = (sizeof(T) == 2 && D <= 64) ? 2 : 1;
  auto kkv = attn_bwd_dkv_kernel<T, D, KV_KT>;
  // queries per wave of the dQ kernel: 32 (QT = 2) reuses each K / V fragment twice; at d = 96 its two operand sets
  // (Q, dO) + two score tiles no longer fit 256 registers, so 16 (QT = 1, three waves per SIMD) wins there
  constexpr int DQ_QT = D >= 96 ? 1 : 2;
  ensure_dynamic_lds(attn_bwd_dq_kernel<T, D, DQ_QT>, smem_q);
  ensure_dynamic_lds(kkv, smem_kv);
  a.delta_self_w = const_cast<float*>(a.delta_self); a.delta_cross_w = const_cast<float*>(a.delta_cross);
  a.dkc = dkc; a.dvc = dvc; a.dkc_bs = dc_bs; a.dkc_rs = dc_rs;
  if constexpr (sizeof(T) == 2) {
    // short sequences: one block per (batch, head), operands LDS-resident (attn_bwd_small_kernel)
    // development knob mdm_dev_set_attn_bwd (include/mdm_hip_dev.h), 0 in the product: 1 = "split" (the two streaming
    // kernels on 16x16x32 MFMAs, always), 2 = "small" (one block per head whenever the shape allows; the tests),
    // 3 = "small16" (as 2, but the round-3 kernel on 16x16x32 MFMAs), 4 = "stream32" (the two streaming kernels of
    // attn32.hpp whenever the shape allows)
    const bool split_only = g_attn_bwd_mode == 1 || g_attn_bwd_mode == 4;
    const bool force_small = g_attn_bwd_mode == 2 || g_attn_bwd_mode == 3;
    if constexpr (D == 64 || D == 96) {
      // a wave owns 32 keys / queries (attn32.hpp): one pass over the resident tiles for all 256 rows of the level
      // (at every batch size: with fewer heads than CUs -- the nested model's inner U-Net at batch 16 -- one block per head
      // is still no slower than the 4 + 5 blocks per head of the streaming kernels: 65 against 73 us at B x H = 128)
      if (a.L <= 256 && (!a.kc || a.S <= 32) && !split_only && g_attn_bwd_mode != 3) {
        constexpr int smem32 = attn_bwd_small32_lds<D>();
        ensure_dynamic_lds(attn_bwd_small32_kernel<D>, smem32);
        hipLaunchKernelGGL((attn_bwd_small32_kernel<D>), dim3(a.B * a.H), dim3(512), smem32, st, a);
        MDM_LAUNCH_STATUS();
      }
    }
    if constexpr (D == 64) {
      // long sequences (the 32x32 level: L = 1024, d = 64): the same tile steps, keys / queries streamed through LDS
      // (d = 96 would spill: 256 registers hold the accumulators and operands of a step, not the stream's staging on top)
      // (5 = "auto, but long sequences on the 16x16x32 streaming kernels": A/B of the in-step effect)
      if (((a.L > 256 && g_attn_bwd_mode == 0) || g_attn_bwd_mode == 4) && (!a.kc || a.S <= 32)) {
        constexpr int smem_dq = attn_bwd_dq32_lds<D>(), smem_dkv = attn_bwd_dkv32_lds<D>();
        ensure_dynamic_lds(attn_bwd_dq32_kernel<D>, smem_dq);
        ensure_dynamic_lds(attn_bwd_dkv32_kernel<D>, smem_dkv);
        const int nb = (a.L + 255) / 256;
        hipLaunchKernelGGL((attn_bwd_dq32_kernel<D>), dim3(nb, a.B * a.H), dim3(512), smem_dq, st, a);
        hipLaunchKernelGGL((attn_bwd_dkv32_kernel<D>), dim3(nb + (a.kc ? 1 : 0), a.B * a.H), dim3(512), smem_dkv, st, a);
        MDM_LAUNCH_STATUS();
      }
    }
    // (one block per head: worth it once the heads fill the chip -- at batch 16 the 128 blocks of the 64x64 U-Net's
    // 16x16 level would leave half the CUs idle, and the streaming kernels' 4 + 5 blocks per head win)
    if (a.L <= 256 && (!a.kc || a.S <= 64) && (a.B * a.H >= device_cus() || force_small) && !split_only) {
      constexpr int smem_small = 8 * G::NAT_BYTES + 4096;
      ensure_dynamic_lds(attn_bwd_small_kernel<D>, smem_small);
      hipLaunchKernelGGL((attn_bwd_small_kernel<D>), dim3(a.B * a.H), dim3(512), smem_small, st, a);
      MDM_LAUNCH_STATUS();
    }
  }
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, DQ_QT>), dim3((a.L + 64 * DQ_QT - 1) / (64 * DQ_QT), a.B * a.H), dim3(256), smem_q, st, a);
  const int nself = (a.L + 64 * KV_KT - 1) / (64 * KV_KT), ncross = a.kc ? (a.S + 64 * KV_KT - 1) / (64 * KV_KT) : 0;
  hipLaunchKernelGGL(kkv, dim3(nself + ncross, a.B * a.H), dim3(256), smem_kv, st, a);
  MDM_LAUNCH_STATUS();
}

// qkv: [B, L, 3C] (q | k | v along channels), kvc: [B, S, 2C] (k_c | v_c) or null, mask [B, S] or null.
// out, out_cross: [B, L, C]; lse_*: [B, H, L] fp32.  C = H * d, d in {32, 64, 96, 128}.
extern "C" int mdm_attn_fwd(const void* qkv, const void* kvc, const float* mask, void* out, void* out_cross,
                            float* lse_self, float* lse_cross, int B, int L, int S, int H, int d, int dtype,
                            void* stream) {
  MDM_CHECK_ARG(qkv && out);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16 || dtype == DT_F32_SPLIT);
  const bool split_products = dtype == DT_F32_SPLIT;   // fp32 tensors, both matmuls as bf16x3 products
  if (split_products) dtype = DT_F32;
  MDM_CHECK_ARG(B > 0 && L > 0 && H > 0);
  MDM_CHECK_ARG(!kvc || S > 0);
  MDM_CHECK_ARG(!kvc || out_cross);   // the cross part is staged through out_cross (and the backward needs it)
  const int C = H * d;
  const size_t es = dtype == DT_F32 ? 4 : 2;
  AttnArgs a = {};
  a.q = qkv; a.q_bs = (size_t)L * 3 * C; a.q_rs = 3 * C;
  a.k = (const char*)qkv + (size_t)C * es; a.v = (const char*)qkv + (size_t)2 * C * es; a.k_bs = a.q_bs; a.k_rs = 3 * C;
  if (kvc) { a.kc = kvc; a.vc = (const char*)kvc + (size_t)C * es; a.c_bs = (size_t)S * 2 * C; a.c_rs = 2 * C; }
  a.mask = mask; a.out = out; a.out_cross = out_cross; a.o_bs = (size_t)L * C; a.o_rs = C;
  a.lse_self = lse_self; a.lse_cross = lse_cross;
  a.B = B; a.H = H; a.L = L; a.S = S; a.scale = 1.0f / sqrtf((float)d);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MDM_ATTN_FWD(DD)                                                                        \
  case DD: return split_products ? attn_fwd_launch<float, DD, true>(a, st)                       \
                                 : dtype == DT_F32 ? attn_fwd_launch<float, DD>(a, st) : attn_fwd_launch<bf16, DD>(a, st);
  switch (d) {
    MDM_ATTN_FWD(32) MDM_ATTN_FWD(64) MDM_ATTN_FWD(96) MDM_ATTN_FWD(128)
    default: MDM_CHECK_ARG(!"unsupported head dim");
  }
#undef MDM_ATTN_FWD
  return -1;
}

// dqkv [B, L, 3C], dkvc [B, S, 2C] are fully overwritten.  delta_* : fp32 workspaces [B, H, L].
extern "C" int mdm_attn_bwd(const void* qkv, const void* kvc, const float* mask, const void* out,
                            const void* out_cross, const void* dout, const float* lse_self, const float* lse_cross,
                            float* delta_self, float* delta_cross, void* dqkv, void* dkvc, int B, int L, int S,
                            int H, int d, int dtype, void* stream) {
  MDM_CHECK_ARG(qkv && out && dout && lse_self && delta_self && dqkv);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  MDM_CHECK_ARG(!kvc || (out_cross && lse_cross && delta_cross && dkvc && S > 0));
  const int C = H * d;
  const size_t es = dtype == DT_F32 ? 4 : 2;
  AttnArgs a = {};
  a.q = qkv; a.q_bs = (size_t)L * 3 * C; a.q_rs = 3 * C;
  a.k = (const char*)qkv + (size_t)C * es; a.v = (const char*)qkv + (size_t)2 * C * es; a.k_bs = a.q_bs; a.k_rs = 3 * C;
  if (kvc) { a.kc = kvc; a.vc = (const char*)kvc + (size_t)C * es; a.c_bs = (size_t)S * 2 * C; a.c_rs = 2 * C; }
  a.mask = mask; a.o_bs = (size_t)L * C; a.o_rs = C;
  a.lse_self = const_cast<float*>(lse_self); a.lse_cross = const_cast<float*>(lse_cross);
  a.dout = dout; a.delta_self = delta_self; a.delta_cross = delta_cross;
  a.out = const_cast<void*>(out); a.out_cross = const_cast<void*>(out_cross);
  a.dq = dqkv; a.dk = (char*)dqkv + (size_t)C * es; a.dv = (char*)dqkv + (size_t)2 * C * es;
  a.dk_bs = a.q_bs; a.dk_rs = 3 * C;
  a.B = B; a.H = H; a.L = L; a.S = S; a.scale = 1.0f / sqrtf((float)d);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  void* dkc = dkvc;
  void* dvc = kvc ? (char*)dkvc + (size_t)C * es : nullptr;
#define MDM_ATTN_BWD(DD)                                                                                    \
  case DD: return dtype == DT_F32 ? attn_bwd_launch<float, DD>(a, dkc, dvc, (size_t)S * 2 * C, 2 * C, st)  \
                                  : attn_bwd_launch<bf16, DD>(a, dkc, dvc, (size_t)S * 2 * C, 2 * C, st);
  switch (d) {
    MDM_ATTN_BWD(32) MDM_ATTN_BWD(64) MDM_ATTN_BWD(96) MDM_ATTN_BWD(128)
    default: MDM_CHECK_ARG(!"unsupported head dim");
  }
#undef MDM_ATTN_BWD
  return -1;
}
