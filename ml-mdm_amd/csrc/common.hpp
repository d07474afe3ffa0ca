// Shared device-side helpers for the gfx950 (CDNA4) kernels of libmdm_hip.
//
// Conventions used by every kernel in this directory:
//   * activations are NHWC ("pixel-major"): [N, H, W, C] with C contiguous
//   * T is the storage type of activations and packed weights: float or __bf16
//   * all reductions / accumulators are fp32
//   * a "chunk" is 16 bytes = EPV elements of T (8 bf16 or 4 fp32)
//   * wavefront = 64 lanes, hard-coded
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_set>

namespace mdm {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// DT_F32_SPLIT: fp32 tensors and accumulators, the products formed as three bf16 MFMAs (x = hi + lo, both bf16:
// x*y ~ hi*hi' + hi*lo' + lo*hi') -- the arithmetic of the "fp32 at a third of the bf16 rate" sampling mode.  Accepted
// where a kernel has the path (mdm_conv_fwd*, mdm_attn_fwd); everything else takes DT_F32 for the same tensors.
enum { DT_F32 = 0, DT_BF16 = 1, DT_F32_SPLIT = 2, DT_F32_SPLIT_W = 3 };   // _W: + the weight operand pre-split into planes

template <typename T> struct Tr;
template <> struct Tr<float> {
  static constexpr int EPV = 4;       // elements per 16-byte chunk
  static constexpr int BK = 32;       // GEMM k-tile in elements (128 B rows)
  static constexpr int KSTEPS = 1;    // generic 32-deep mma steps per k-tile
};
template <> struct Tr<bf16> {
  static constexpr int EPV = 8;
  static constexpr int BK = 64;
  static constexpr int KSTEPS = 2;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// ---------------------------------------------------------------------------
// 16-byte chunk <-> fp32 lanes
// ---------------------------------------------------------------------------
template <typename T> struct Chunk;  // EPV values of T held as fp32
template <> struct Chunk<float> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ void store(float* p) const {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct Chunk<bf16> {
  float v[8];
  __device__ __forceinline__ void load(const bf16* p) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  __device__ __forceinline__ void store(bf16* p) const {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// ---------------------------------------------------------------------------
// MFMA fragments. A "fragment" is 8 consecutive reduction-dim elements owned by
// a lane; lane l owns row/col (l & 15) and reduction slots (l >> 4) * 8 + j.
// bf16 : one v_mfma_f32_16x16x32_bf16 consumes a fragment pair.
// fp32 : eight v_mfma_f32_16x16x4_f32 consume it (slot j of every quad per
//        instruction) -- exact fp32, used by the parity ("fp32") mode.
// Output mapping (both): acc[i] = D[row = (l >> 4) * 4 + i][col = l & 15] for
// D = A * B with A the first operand.
// ---------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<bf16> {
  bf16x8 v;
  __device__ __forceinline__ void load_lds(const char* p0, const char* /*p1*/) {
    v = *reinterpret_cast<const bf16x8*>(p0);
  }
};
template <> struct Frag<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ void load_lds(const char* p0, const char* p1) {
    lo = *reinterpret_cast<const f32x4*>(p0);
    hi = *reinterpret_cast<const f32x4*>(p1);
  }
};

// fp32 operand fragment held as two bf16 fragments, hi = bf16(x), lo = bf16(x - hi): |x - hi - lo| <= 2^-17 |x|, so the
// three products hi*hi' + hi*lo' + lo*hi' carry x*y to ~2^-16 relative -- fp32-accumulated, 3 x v_mfma_f32_16x16x32_bf16
// (48 cycles) for what the exact path does in 8 x v_mfma_f32_16x16x4_f32 (256 cycles).  Same lane ownership as
// Frag<float> (row / column l & 15, reduction slots (l >> 4) * 8 + j), so it drops into the fp32 kernels' loops.
struct FragSplit {
  bf16x8 hi, lo;
  __device__ __forceinline__ void from_f32(const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16 h0 = (bf16)a[j], h1 = (bf16)b[j];
      hi[j] = h0; hi[4 + j] = h1;
      lo[j] = (bf16)(a[j] - (float)h0); lo[4 + j] = (bf16)(b[j] - (float)h1);
    }
  }
  __device__ __forceinline__ void load_lds(const char* p0, const char* p1) {
    from_f32(*reinterpret_cast<const f32x4*>(p0), *reinterpret_cast<const f32x4*>(p1));
  }
};
template <typename T, bool SPLIT> struct FragOf { using type = Frag<T>; };
template <> struct FragOf<float, true> { using type = FragSplit; };

__device__ __forceinline__ void mma16(f32x4& acc, const FragSplit& a, const FragSplit& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);   // small terms first
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<bf16>& a, const Frag<bf16>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
}

// LDS tile addressing shared by the GEMM-shaped kernels: rows of 128 bytes (8
// chunks), chunk index XOR-swizzled with (row & 7) so that the 16-lane groups
// of ds_read_b128 land on 16 distinct 16-byte slots.
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) {
  return row * 128 + ((chunk ^ (row & 7)) << 4);
}

// Fragment read for mma step `ks` (32 reduction elements per step) of row `row`.
template <typename T>
__device__ __forceinline__ void load_frag(Frag<T>& f, const char* tile, int row, int ks, int quad);
template <>
__device__ __forceinline__ void load_frag<bf16>(Frag<bf16>& f, const char* tile, int row, int ks, int quad) {
  f.load_lds(tile + lds_chunk_off(row, ks * 4 + quad), nullptr);
}
template <>
__device__ __forceinline__ void load_frag<float>(Frag<float>& f, const char* tile, int row, int /*ks*/, int quad) {
  f.load_lds(tile + lds_chunk_off(row, 2 * quad), tile + lds_chunk_off(row, 2 * quad + 1));
}

// the same reads for any fragment type (overloads: the fp32 kernels pick Frag<float> or FragSplit by template flag)
__device__ __forceinline__ void load_frag_x(Frag<bf16>& f, const char* tile, int row, int ks, int quad) { load_frag<bf16>(f, tile, row, ks, quad); }
__device__ __forceinline__ void load_frag_x(Frag<float>& f, const char* tile, int row, int ks, int quad) { load_frag<float>(f, tile, row, ks, quad); }
__device__ __forceinline__ void load_frag_x(FragSplit& f, const char* tile, int row, int /*ks*/, int quad) {
  f.load_lds(tile + lds_chunk_off(row, 2 * quad), tile + lds_chunk_off(row, 2 * quad + 1));
}

// EPV x EPV register block with an in-register transpose: rows are 16-byte
// chunks loaded from HBM (reduction index slow), col(c) is the 16-byte chunk of
// the transposed block.  Used wherever an MFMA operand is reduction-major in HBM.
template <typename T> struct Blk;  // EPV x EPV register block + transpose
template <> struct Blk<float> {
  f32x4 r[4];
  __device__ __forceinline__ void zero_row(int i) { r[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void load_row(int i, const float* p) { r[i] = *reinterpret_cast<const f32x4*>(p); }
  // branch-free guarded load: `p` must be a mapped address even when !valid
  __device__ __forceinline__ void load_row_sel(int i, const float* p, bool valid) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r[i] = f32x4{valid ? t[0] : 0.f, valid ? t[1] : 0.f, valid ? t[2] : 0.f, valid ? t[3] : 0.f};
  }
  // row c of the transposed block: elements (r[0][c], r[1][c], r[2][c], r[3][c])
  __device__ __forceinline__ uint4 col(int c) const {
    f32x4 o = {r[0][c], r[1][c], r[2][c], r[3][c]};
    return *reinterpret_cast<uint4*>(&o);
  }
};
template <> struct Blk<bf16> {
  uint4 r[8];
  __device__ __forceinline__ void zero_row(int i) { r[i] = uint4{0u, 0u, 0u, 0u}; }
  __device__ __forceinline__ void load_row(int i, const bf16* p) { r[i] = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void load_row_sel(int i, const bf16* p, bool valid) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    r[i] = uint4{valid ? t.x : 0u, valid ? t.y : 0u, valid ? t.z : 0u, valid ? t.w : 0u};
  }
  __device__ __forceinline__ uint32_t word(const uint4& v, int i) const {
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
  }
  __device__ __forceinline__ uint4 col(int c) const {
    uint32_t o[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const uint32_t a = word(r[2 * pp], c >> 1), b = word(r[2 * pp + 1], c >> 1);
      o[pp] = (c & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    return uint4{o[0], o[1], o[2], o[3]};
  }
};

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// v_rcp_f32 (1 ulp) instead of an IEEE division: "a / b" expands to an 11-instruction div_scale / div_fmas /
// div_fixup sequence, which made the activation the most expensive part of the epilogues that apply it
__device__ __forceinline__ float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }
// (the two transcendentals of SiLU are free where it is used: with silu / dsilu replaced by one multiply the GroupNorm kernels
// -- HBM-bound -- leave the train step where it is, 92.3-92.8 against 92.6-92.7 ms; profiles/r05_did_not_pay.md #13)
__device__ __forceinline__ float silu_f(float z) { return z * rcp_f(1.f + __expf(-z)); }
// d silu(z) / dz
__device__ __forceinline__ float dsilu_f(float z) {
  float s = rcp_f(1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}
// exact (erf) GELU, nn.GELU() default (unet.py:270).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e.
// fp32 round-off class) instead of libm's erff: 1 rcp + 1 exp + a 5-term Horner chain, branch-free -- the GEMM
// epilogues that apply it (FFN up-projection and its gradient) were spending ~40 % of their time in erff.
// Both the cdf and the pdf of the standard normal come from the same exponential e^{-z^2/2}.
__device__ __forceinline__ void gauss_cdf_pdf(float z, float& cdf, float& pdf) {
  const float az = fabsf(z) * 0.70710678118654752f;          // |z| / sqrt(2)
  const float e = __expf(-az * az);                            // e^{-z^2/2}
  const float t = rcp_f(1.f + 0.3275911f * az);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.f - poly * e;                        // erf(|z|/sqrt 2)
  cdf = fmaf(0.5f, copysignf(erf_abs, z), 0.5f);
  pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float gelu_f(float z) {
  float c, p;
  gauss_cdf_pdf(z, c, p);
  return z * c;
}
__device__ __forceinline__ float dgelu_f(float z) {
  float c, p;
  gauss_cdf_pdf(z, c, p);
  return c + z * p;
}
// The same two functions for tensors stored in bf16.  The erf form above costs ~18 VALU instructions per element, two of
// them transcendental (quarter rate), and the GELU epilogues of the FFN GEMMs pay for it: with the activation's arithmetic
// removed altogether the FFN-up launch (768 -> 3072, batch 64) goes 117.6 -> 98.9 us, its x gelu'(aux) gradient 122.9 ->
// 105.9, the train step 92.6 -> 90.9 ms (round 5, variant libraries, one call).  For a bf16 result nothing near 1e-7 is
// needed, and the cheapest form has NO transcendental:
//   Phi(z) ~ 0.5 + zc P(zc^2),  gelu'(z) ~ 0.5 + zc Q(zc^2),  zc = clamp(z, -4, 4),  P, Q of degree 7 in zc^2,
// fitted minimax with zc P = zc Q = 0.5 exactly at zc = 4, so both saturate continuously at 1 / 0.  Evaluated in fp32 on
// every bf16 input: max |error| 1.3e-4 for gelu, 5.1e-4 for gelu' -- relative L2 over N(0,1) inputs 3.4e-5 / 5.2e-4 against
// 1.7e-3 for the bf16 rounding of the result itself; 12 / 10 plain VALU instructions.  Measured (A/B in one call): FFN-up
// 113 -> 103-109 us, its gradient 120 -> 110, 512 -> 2048 at 32x32 251 -> 233 / 276 -> 253, train step -0.7 ... -0.9 ms.
// (A sigmoid form z / (1 + 2^(-w(z))) with a fitted quintic w -- 9 instructions, v_exp + v_rcp among them, error 3e-5 --
// bought only 0.2 ms: profiles/r05_did_not_pay.md #12.)  The fp32 kernels keep erf.
__device__ __forceinline__ float odd_poly8(float zc, const float (&c)[8]) {
  const float x2 = zc * zc;
  float p = c[7];
#pragma unroll
  for (int k = 6; k >= 0; --k) p = fmaf(p, x2, c[k]);
  return fmaf(zc, p, 0.5f);
}
__device__ __forceinline__ float gelu_poly(float z) {
  constexpr float P[8] = {3.988065672e-01f, -6.606670007e-02f, 9.583257106e-03f, -1.021626672e-03f,
                          7.628174443e-05f, -3.717267849e-06f, 1.047660447e-07f, -1.283420065e-09f};
  const float cdf = fmaxf(odd_poly8(__builtin_amdgcn_fmed3f(z, -4.f, 4.f), P), 0.f);   // (rounding leaves -1e-6 at -4)
  return z * cdf;                                  // (z, not the clamped value: exact identity above 4, NaN / inf stay)
}
__device__ __forceinline__ float dgelu_poly(float z) {
  constexpr float Q[8] = {7.989620567e-01f, -2.662556930e-01f, 5.843304141e-02f, -8.376759832e-03f,
                          7.867717586e-04f, -4.620522401e-05f, 1.525062443e-06f, -2.145770312e-08f};
  return odd_poly8(__builtin_amdgcn_fmed3f(z, -4.f, 4.f), Q);   // (fmed3 maps NaN to -4: see DGeluCode for where a NaN goes)
}
// gelu'(pre) of a bf16 FFN pre-activation as ONE BYTE -- what the FFN-up launch leaves for its backward (round 6).
// The backward needs the pre-activation only through gelu'(pre); stored as bf16 `pre` that is 2 bytes written by the
// forward epilogue and 2 bytes read by the x gelu'(aux) epilogue of the input gradient per hidden element -- 100 MB each
// way per 768 -> 3072 layer at batch 64, in launches whose store phase is additive (HISTORY.md section 4.1).  gelu' lies in
// [-0.129, 1.129]; the code is q = round(196 g) + 28 (0 <-> 28 and 1 <-> 224 exactly, so the saturated tails decode to
// exactly 0 and 1), step 1 / 196 = 5.1e-3, |error| <= 2.6e-3 + the polynomial's 5e-4 -- the size of the bf16 rounding of
// a value near 1 (3.9e-3 ulp), unbiased, and only in the gradient (the forward output is bit-identical).  fp32 tensors
// keep the fp32 pre-activation and the erf form.  A NaN pre-activation encodes as 0 (v_cvt_u32_f32) = gelu' -0.143: the
// NaN itself travels through the forward output (gelu(NaN) = NaN), the loss, and the step's NaN skip.
// The same two polynomials on PAIRS of values with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of
// fp32 FMAs per instruction; the clamp and the final max stay one v_med3 / v_max per element): 7 / 5.5 instructions per
// element instead of 12 / 10, bit-identical results (the same FMA chain per element).  The GEMM epilogues that apply them are
// VALU-bound, not store-bound, where they run (round 6: one VALU instruction per element of the 256 x 256 tile costs the
// 768 -> 3072 launch ~1 us -- a block's 8 waves own the CU and nothing else issues while they compute).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 odd_poly8_pk(f32x2 zc, const float (&c)[8]) {
  const f32x2 x2 = zc * zc;
  f32x2 p = {c[7], c[7]};
#pragma unroll
  for (int k = 6; k >= 0; --k) p = __builtin_elementwise_fma(p, x2, f32x2{c[k], c[k]});
  return __builtin_elementwise_fma(zc, p, f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 clamp4_pk(f32x2 z) {
  return f32x2{__builtin_amdgcn_fmed3f(z[0], -4.f, 4.f), __builtin_amdgcn_fmed3f(z[1], -4.f, 4.f)};
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 z) {
  constexpr float P[8] = {3.988065672e-01f, -6.606670007e-02f, 9.583257106e-03f, -1.021626672e-03f,
                          7.628174443e-05f, -3.717267849e-06f, 1.047660447e-07f, -1.283420065e-09f};
  const f32x2 c = odd_poly8_pk(clamp4_pk(z), P);
  return z * f32x2{fmaxf(c[0], 0.f), fmaxf(c[1], 0.f)};
}
__device__ __forceinline__ f32x2 dgelu_poly2(f32x2 z) {
  constexpr float Q[8] = {7.989620567e-01f, -2.662556930e-01f, 5.843304141e-02f, -8.376759832e-03f,
                          7.867717586e-04f, -4.620522401e-05f, 1.525062443e-06f, -2.145770312e-08f};
  return odd_poly8_pk(clamp4_pk(z), Q);
}

#if defined(MDM_FFN_AUX_BF16)   // development A/B (tools/build_variant_src.sh): the bf16 pre-activation as before round 6
constexpr bool kFfnAuxByte = false;
#else
constexpr bool kFfnAuxByte = true;
#endif
// v_cvt_pk_u8_f32 converts with round-to-nearest-even and saturation, NaN -> 0 (tools/probes/cvt_pk_u8_probe.hip), and
// drops the byte into place: one instruction per element for the conversion AND the packing.
struct DGeluCode {
  static constexpr float SCALE = 196.f, ZERO = 28.f;
  static __device__ __forceinline__ unsigned enc(float z) {
    return __builtin_amdgcn_cvt_pk_u8_f32(fmaf(dgelu_poly(z), SCALE, ZERO), 0, 0u);
  }
  static __device__ __forceinline__ float dec(unsigned q) { return fmaf((float)q, 1.f / SCALE, -ZERO / SCALE); }
  // 4 consecutive elements <-> 4 bytes (N = 4 or 8 values at z / g)
  static __device__ __forceinline__ unsigned enc4(const float* z) {
    const f32x2 k = {SCALE, SCALE}, o = {ZERO, ZERO};
    const f32x2 t0 = __builtin_elementwise_fma(dgelu_poly2(f32x2{z[0], z[1]}), k, o);
    const f32x2 t1 = __builtin_elementwise_fma(dgelu_poly2(f32x2{z[2], z[3]}), k, o);
    unsigned w = __builtin_amdgcn_cvt_pk_u8_f32(t0[0], 0, 0u);
    w = __builtin_amdgcn_cvt_pk_u8_f32(t0[1], 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(t1[0], 2, w);
    return __builtin_amdgcn_cvt_pk_u8_f32(t1[1], 3, w);
  }
  static __device__ __forceinline__ void dec4(unsigned w, float* g) {
    const f32x2 k = {1.f / SCALE, 1.f / SCALE}, o = {-ZERO / SCALE, -ZERO / SCALE};
    const f32x2 a = __builtin_elementwise_fma(f32x2{(float)(w & 0xffu), (float)((w >> 8) & 0xffu)}, k, o);
    const f32x2 b = __builtin_elementwise_fma(f32x2{(float)((w >> 16) & 0xffu), (float)(w >> 24)}, k, o);
    g[0] = a[0]; g[1] = a[1]; g[2] = b[0]; g[3] = b[1];
  }
};
#if defined(MDM_GELU_EXACT)   // development A/B (tools/build_variant_gemm.sh): the erf form for bf16 tensors too
template <typename T> __device__ __forceinline__ float gelu_t(float z) { return gelu_f(z); }
template <typename T> __device__ __forceinline__ float dgelu_t(float z) { return dgelu_f(z); }
#else
template <typename T> __device__ __forceinline__ float gelu_t(float z) { return sizeof(T) == 2 ? gelu_poly(z) : gelu_f(z); }
template <typename T> __device__ __forceinline__ float dgelu_t(float z) { return sizeof(T) == 2 ? dgelu_poly(z) : dgelu_f(z); }
#endif
// v[e] = gelu(v[e]) / v[e] *= gelu'(a[e]) over a chunk (N even): bf16 tensors take the packed polynomial forms
template <typename T, int N> __device__ __forceinline__ void gelu_vec(float (&v)[N]) {
#if !defined(MDM_GELU_EXACT) && !defined(MDM_GELU_SCALAR)   // (MDM_GELU_SCALAR: development A/B, the per-element forms)
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
      const f32x2 r = gelu_poly2(f32x2{v[e], v[e + 1]});
      v[e] = r[0]; v[e + 1] = r[1];
    }
    return;
  }
#endif
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] = gelu_t<T>(v[e]);
}
template <typename T, int N> __device__ __forceinline__ void mul_dgelu_vec(float (&v)[N], const float (&a)[N]) {
#if !defined(MDM_GELU_EXACT) && !defined(MDM_GELU_SCALAR)   // (MDM_GELU_SCALAR: development A/B, the per-element forms)
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < N; e += 2) {
      const f32x2 r = f32x2{v[e], v[e + 1]} * dgelu_poly2(f32x2{a[e], a[e + 1]});
      v[e] = r[0]; v[e + 1] = r[1];
    }
    return;
  }
#endif
#pragma unroll
  for (int e = 0; e < N; ++e) v[e] *= dgelu_t<T>(a[e]);
}

// XCD-aware, bijective remap of a 1-D block id: consecutive logical ids land on
// the same XCD (hardware places block b on XCD b % 8) so neighbouring tiles that
// share operand panels hit the same L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace mdm

// host-side status helpers --------------------------------------------------
#define MDM_CHECK_ARG(cond)                                                                      \
  do {                                                                                           \
    if (!(cond)) {                                                                               \
      mdm_set_error(__FILE__, __LINE__, #cond);                                                  \
      return -1;                                                                                 \
    }                                                                                            \
  } while (0)
#define MDM_LAUNCH_STATUS()                                                                      \
  do {                                                                                           \
    hipError_t e__ = hipGetLastError();                                                          \
    if (e__ != hipSuccess) {                                                                     \
      mdm_set_error(__FILE__, __LINE__, hipGetErrorString(e__));                                 \
      return (int)e__;                                                                           \
    }                                                                                            \
    return 0;                                                                                    \
  } while (0)

extern "C" void mdm_set_error(const char* file, int line, const char* what);

// ---- host-side per-device state ----------------------------------------------------------------------------------
// Kernel attributes (dynamic LDS above 64 KB) and the CU count belong to a DEVICE, and a process may touch several
// (tests on cuda:1, a single-process multi-GPU host): both are keyed by hipGetDevice(), and the table is guarded
// because forward (caller thread) and backward (autograd thread) launch concurrently.
namespace mdm {
inline int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
template <typename K>
inline void ensure_dynamic_lds(K kern, int bytes) {
  static std::mutex mu;
  static std::unordered_set<uint64_t> done;
  const uint64_t key = (uint64_t)reinterpret_cast<uintptr_t>(reinterpret_cast<const void*>(kern)) ^
                       ((uint64_t)(unsigned)current_device() << 56) ^ ((uint64_t)(unsigned)bytes << 40);
  std::lock_guard<std::mutex> lock(mu);
  if (done.count(key)) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done.insert(key);
}
inline int device_cus() {
  static std::mutex mu;
  static int cus[64] = {};
  const int d = current_device() & 63;
  std::lock_guard<std::mutex> lock(mu);
  if (!cus[d]) {
    hipDeviceProp_t pr;
    cus[d] = (hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return cus[d];
}
}  // namespace mdm
