// GroupNorm (+FiLM, +SiLU) and LayerNorm forward/backward for NHWC activations.
//
// Reference semantics:
//   ResNet.forward      h = silu(norm1(x)); h = silu(norm2(h) * (1 + ta) + tb)   ml_mdm/models/unet.py:223-233
//   SelfAttention       qkv(norm(x)), ffn[0] = GroupNorm(32, C)                   unet.py:259,268,300
//   UNet output         silu(norm_out(x))                                          unet.py:877-879
//   cross-attn cond     norm_cond = LayerNorm(cond_dim)                            unet.py:263,304
// nn.GroupNorm(G, C, eps=1e-5, affine), biased variance.
//
// All kernels here are HBM-bound streaming kernels: 16-byte loads, fp32 math,
// deterministic two-stage reductions (per-slab partials -> finalize), no atomics.
//
// Forward  = gn_partial (reads x) -> gn_finalize (tiny) -> gn_apply (reads x, writes y)
//   y = act(a[n,c] * x + b[n,c]),  a = gamma*f*rstd, b = (beta - mean*rstd*gamma)*f + tb, f = 1 + ta
// Backward = gn_bwd_partial (reads dy, x) -> gn_bwd_finalize -> gn_bwd_apply (reads dy, x, writes dx)
//   dz = dy * act'(a*x + b);  dx = a*dz + q[n,g]*x + r[n,g]
#include <stdlib.h>

#include "common.hpp"

namespace mdm {

// ---- stage 1: per (n, slab) per-channel shifted sums -------------------------
// part[n][slab][c] = (s1, s2) with s1 = sum(x - K_c), s2 = sum((x - K_c)^2), K_c = x[n, 0, c]
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                         int HW, int C, int slabs, int pix_per_slab) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[256 * 16];
  const int n = blockIdx.x / slabs, slab = blockIdx.x - n * slabs;
  const int tid = threadIdx.x;
  const int nchunks = C / EPV;
  const int p_begin = slab * pix_per_slab, p_end = min(HW, p_begin + pix_per_slab);
  const T* xn = x + (size_t)n * HW * C;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int cw = min(256, nchunks - c0);
    const int rows_par = 256 / cw;
    const int tc = tid % cw, tr = tid / cw;
    float s1[EPV], s2[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (tr < rows_par) {
      Chunk<T> k;
      k.load(xn + (size_t)(c0 + tc) * EPV);
      for (int p = p_begin + tr; p < p_end; p += rows_par) {
        Chunk<T> ch;
        ch.load(xn + (size_t)p * C + (size_t)(c0 + tc) * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const float d = ch.v[e] - k.v[e];
          s1[e] += d; s2[e] += d * d;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) { red[tid * 16 + e] = s1[e]; red[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    if (tr == 0) {
      for (int r = 1; r < rows_par; ++r)
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          s1[e] += red[(r * cw + tc) * 16 + e];
          s2[e] += red[(r * cw + tc) * 16 + 8 + e];
        }
      float* o = part + ((size_t)blockIdx.x * C + (size_t)(c0 + tc) * EPV) * 2;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { o[2 * e] = s1[e]; o[2 * e + 1] = s2[e]; }
    }
    __syncthreads();
  }
}

// ---- stage 2: one block per n: group statistics + per-channel coefficients ---
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const T* __restrict__ film,
                                                          float* __restrict__ stats, float* __restrict__ coef, int HW,
                                                          int C, int G, int slabs, float eps) {
  __shared__ float sh_s1[2048], sh_s2[2048], sh_k[2048];  // C <= 2048 (host-checked)
  __shared__ float sh_mean[256], sh_rstd[256];            // G <= 256
  const int n = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  const T* xn = x + (size_t)n * HW * C;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int s = 0; s < slabs; ++s) {
      const float* pp = part + (((size_t)n * slabs + s) * C + c) * 2;
      a += pp[0]; b += pp[1];
    }
    sh_s1[c] = a; sh_s2[c] = b; sh_k[c] = to_f32(xn[c]);
  }
  __syncthreads();
  const float cnt = (float)HW;
  for (int g = tid; g < G; g += 256) {
    float mu = 0.f;
    for (int j = 0; j < cpg; ++j) { const int c = g * cpg + j; mu += sh_k[c] + sh_s1[c] / cnt; }
    mu /= (float)cpg;
    float var = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int c = g * cpg + j;
      const float d = sh_k[c] - mu;
      var += sh_s2[c] + 2.f * d * sh_s1[c] + cnt * d * d;
    }
    var = fmaxf(var / (cnt * (float)cpg), 0.f);
    const float rstd = rsqrtf(var + eps);
    sh_mean[g] = mu; sh_rstd[g] = rstd;
    stats[((size_t)n * G + g) * 2] = mu;
    stats[((size_t)n * G + g) * 2 + 1] = rstd;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float mu = sh_mean[g], rstd = sh_rstd[g];
    float f = 1.f, tb = 0.f;
    if (film) { f = 1.f + to_f32(film[(size_t)n * 2 * C + c]); tb = to_f32(film[(size_t)n * 2 * C + C + c]); }
    const float ga = gamma[c], be = beta[c];
    coef[((size_t)n * C + c) * 2] = ga * f * rstd;
    coef[((size_t)n * C + c) * 2 + 1] = (be - mu * rstd * ga) * f + tb;
  }
}

// ---- stage 3: y = act(a * x + b) ---------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ coef,
                                                       T* __restrict__ y, int HW, int C, size_t total_chunks) {
  constexpr int EPV = Tr<T>::EPV;
  const int cchunks = C / EPV;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    const size_t pix = i / cchunks;
    const int n = (int)(pix / HW);
    Chunk<T> ch;
    ch.load(x + i * EPV);
    const float* cf = coef + ((size_t)n * C + (size_t)cc * EPV) * 2;
    float ab[2 * EPV];
#pragma unroll
    for (int e = 0; e < 2 * EPV; e += 4) *reinterpret_cast<f32x4*>(ab + e) = *reinterpret_cast<const f32x4*>(cf + e);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float z = ab[2 * e] * ch.v[e] + ab[2 * e + 1];
      ch.v[e] = ACT ? silu_f(z) : z;
    }
    ch.store(y + i * EPV);
  }
}

// ---- backward stage 1: per (n, slab) per-channel A1 = sum dz, A2 = sum dz*x ----
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ coef, float* __restrict__ part,
                                                             int HW, int C, int slabs, int pix_per_slab) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[256 * 16];
  const int n = blockIdx.x / slabs, slab = blockIdx.x - n * slabs;
  const int tid = threadIdx.x;
  const int nchunks = C / EPV;
  const int p_begin = slab * pix_per_slab, p_end = min(HW, p_begin + pix_per_slab);
  const size_t nbase = (size_t)n * HW * C;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int cw = min(256, nchunks - c0);
    const int rows_par = 256 / cw;
    const int tc = tid % cw, tr = tid / cw;
    float a1[EPV], a2[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (tr < rows_par) {
      const float* cf = coef + ((size_t)n * C + (size_t)(c0 + tc) * EPV) * 2;
      float ab[2 * EPV];
#pragma unroll
      for (int e = 0; e < 2 * EPV; ++e) ab[e] = cf[e];
      for (int p = p_begin + tr; p < p_end; p += rows_par) {
        const size_t off = nbase + (size_t)p * C + (size_t)(c0 + tc) * EPV;
        Chunk<T> cx, cd;
        cx.load(x + off);
        cd.load(dy + off);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float dz = cd.v[e];
          if (ACT) dz *= dsilu_f(ab[2 * e] * cx.v[e] + ab[2 * e + 1]);
          a1[e] += dz; a2[e] += dz * cx.v[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) { red[tid * 16 + e] = a1[e]; red[tid * 16 + 8 + e] = a2[e]; }
    __syncthreads();
    if (tr == 0) {
      for (int r = 1; r < rows_par; ++r)
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          a1[e] += red[(r * cw + tc) * 16 + e];
          a2[e] += red[(r * cw + tc) * 16 + 8 + e];
        }
      float* o = part + ((size_t)blockIdx.x * C + (size_t)(c0 + tc) * EPV) * 2;
#pragma unroll
      for (int e = 0; e < EPV; ++e) { o[2 * e] = a1[e]; o[2 * e + 1] = a2[e]; }
    }
    __syncthreads();
  }
}

// ---- backward stage 2: one block per n ------------------------------------------
// outputs: qr[n][g] = (q, r); dfilm[n][2C] (if film); pgrad[n][c] = (dgamma_n, dbeta_n)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const T* __restrict__ film,
                                                              float* __restrict__ qr, T* __restrict__ dfilm,
                                                              float* __restrict__ pgrad, int HW, int C, int G, int slabs) {
  __shared__ float sh_fgA1[2048], sh_fgXh[2048];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / G;
  for (int c = tid; c < C; c += 256) {
    float A1 = 0.f, A2 = 0.f;
    for (int s = 0; s < slabs; ++s) {
      const float* pp = part + (((size_t)n * slabs + s) * C + c) * 2;
      A1 += pp[0]; A2 += pp[1];
    }
    const int g = c / cpg;
    const float mu = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
    const float Xh = rstd * (A2 - mu * A1);
    float f = 1.f;
    if (film) f = 1.f + to_f32(film[(size_t)n * 2 * C + c]);
    const float ga = gamma[c], be = beta[c];
    pgrad[((size_t)n * C + c) * 2] = f * Xh;
    pgrad[((size_t)n * C + c) * 2 + 1] = f * A1;
    if (film) {
      dfilm[(size_t)n * 2 * C + c] = from_f32<T>(ga * Xh + be * A1);
      dfilm[(size_t)n * 2 * C + C + c] = from_f32<T>(A1);
    }
    sh_fgA1[c] = f * ga * A1;
    sh_fgXh[c] = f * ga * Xh;
  }
  __syncthreads();
  const float m = (float)cpg * (float)HW;
  for (int g = tid; g < G; g += 256) {
    float S1 = 0.f, S2 = 0.f;
    for (int j = 0; j < cpg; ++j) { S1 += sh_fgA1[g * cpg + j]; S2 += sh_fgXh[g * cpg + j]; }
    const float mu = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
    const float q = -rstd * rstd * S2 / m;
    qr[((size_t)n * G + g) * 2] = q;
    qr[((size_t)n * G + g) * 2 + 1] = -rstd * S1 / m - q * mu;
  }
}

// dgamma[c] = sum_n pgrad[n][c][0]; dbeta[c] = sum_n pgrad[n][c][1]
// block = 32 channels x 8 interleaved sample ranges (256 threads), fixed-order LDS reduction
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ pgrad, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int N, int C, int accumulate) {
  __shared__ float red[8][32][2];
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int n = part; n < N; n += 8) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(pgrad + ((size_t)n * C + c) * 2);
      a += v[0]; b += v[1];
    }
  red[part][cl][0] = a; red[part][cl][1] = b;
  __syncthreads();
  if (part == 0 && c < C) {
#pragma unroll
    for (int q = 1; q < 8; ++q) { a += red[q][cl][0]; b += red[q][cl][1]; }
    dgamma[c] = accumulate ? dgamma[c] + a : a;
    dbeta[c] = accumulate ? dbeta[c] + b : b;
  }
}

// ---- backward stage 3: dx = a*dz + q*x + r ------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ qr, T* __restrict__ dx,
                                                           const T* __restrict__ dres, int HW, int C, int G,
                                                           size_t total_chunks) {
  constexpr int EPV = Tr<T>::EPV;
  const int cchunks = C / EPV;
  const int cpg = C / G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_chunks;
       i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    const size_t pix = i / cchunks;
    const int n = (int)(pix / HW);
    Chunk<T> cx, cd;
    cx.load(x + i * EPV);
    cd.load(dy + i * EPV);
    const float* cf = coef + ((size_t)n * C + (size_t)cc * EPV) * 2;
    float ab[2 * EPV];
#pragma unroll
    for (int e = 0; e < 2 * EPV; e += 4) *reinterpret_cast<f32x4*>(ab + e) = *reinterpret_cast<const f32x4*>(cf + e);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const int g = (cc * EPV + e) / cpg;
      const float q = qr[((size_t)n * G + g) * 2], r = qr[((size_t)n * G + g) * 2 + 1];
      float dz = cd.v[e];
      if (ACT) dz *= dsilu_f(ab[2 * e] * cx.v[e] + ab[2 * e + 1]);
      cd.v[e] = ab[2 * e] * dz + q * cx.v[e] + r;
    }
    if (dres) {
      Chunk<T> cr;
      cr.load(dres + i * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) cd.v[e] += cr.v[e];
    }
    cd.store(dx + i * EPV);
  }
}

// ---------------------------------------------------------------------------------
// Single-kernel GroupNorm for small images (the 16x16 / 32x32 levels: 80 of the 97 GroupNorms of the 64x64 U-Net).
// One block (256 / 512 / 1024 threads by image size) owns (sample n, a slice of whole groups, <= 64 channels): thread (r, c) = (tid / LPR,
// tid % LPR) keeps the 16-byte chunks (pixel r + it * R, chunk c) of the slice in registers, so x (and dy) are read
// from HBM exactly once, the statistics are the exact two-pass ones, and the three-kernel sequence
// partial -> finalize -> apply (each ~15 us of launch / tail latency on a 25 MB tensor) becomes one launch.
// LPR = chunk lanes per pixel row (8 for bf16, 16 for fp32; lanes past the slice idle), R = threads / LPR rows per pass,
// NI = passes (host guarantees HW <= NI * R).  Reductions: shuffles across the rows of a wave, then LDS across waves
// in a fixed order -> deterministic.
// ---------------------------------------------------------------------------------
template <typename T> struct GnF;
template <> struct GnF<bf16> { static constexpr int LPR = 8; };
template <> struct GnF<float> { static constexpr int LPR = 16; };

template <int LPR>
__device__ __forceinline__ float rows_sum(float v) {   // sum over the lanes of a wave that share tid % LPR
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int ACT, int NI, int NTHR>
__global__ __launch_bounds__(NTHR) void gn_fused_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ film,
                                                            T* __restrict__ y, float* __restrict__ stats,
                                                            float* __restrict__ coef, int HW, int C, int G, int CB,
                                                            float eps) {
  constexpr int EPV = Tr<T>::EPV, LPR = GnF<T>::LPR, R = NTHR / LPR, NW = NTHR / 64;
  __shared__ float sh[NW][LPR];
  __shared__ float gmean[8], grstd[8];
  const int n = blockIdx.x, cb0 = blockIdx.y * CB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid % LPR, r = tid / LPR;
  const int cpg = C / G, cpc = cpg / EPV, gpb = CB / cpg;
  const bool active = c * EPV < CB;
  const int ch0 = cb0 + c * EPV;
  const int gl = active ? (c * EPV) / cpg : 0;
  const T* xn = x + (size_t)n * HW * C + ch0;
  uint4 raw[NI];   // the slice stays in registers in storage format (converted on use: 4 registers per chunk)
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int p = r + it * R;
    raw[it] = uint4{0u, 0u, 0u, 0u};
    if (active && p < HW) {
      raw[it] = *reinterpret_cast<const uint4*>(xn + (size_t)p * C);
      Chunk<T> v;
      v.load(reinterpret_cast<const T*>(&raw[it]));
#pragma unroll
      for (int e = 0; e < EPV; ++e) s += v.v[e];
    }
  }
  const float cnt = (float)cpg * (float)HW;
  s = rows_sum<LPR>(s);
  if (lane < LPR) sh[wave][lane] = s;
  __syncthreads();
  if (tid < gpb) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w)
      for (int cc = tid * cpc; cc < (tid + 1) * cpc; ++cc) t += sh[w][cc];
    gmean[tid] = t / cnt;
  }
  __syncthreads();
  const float mu = gmean[gl];
  float q = 0.f;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int p = r + it * R;
    if (active && p < HW) {
      Chunk<T> v;
      v.load(reinterpret_cast<const T*>(&raw[it]));
#pragma unroll
      for (int e = 0; e < EPV; ++e) { const float d = v.v[e] - mu; q += d * d; }
    }
  }
  q = rows_sum<LPR>(q);
  if (lane < LPR) sh[wave][lane] = q;
  __syncthreads();
  if (tid < gpb) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w)
      for (int cc = tid * cpc; cc < (tid + 1) * cpc; ++cc) t += sh[w][cc];
    const float rstd = rsqrtf(t / cnt + eps);
    grstd[tid] = rstd;
    const int g = cb0 / cpg + tid;
    stats[((size_t)n * G + g) * 2] = gmean[tid];
    stats[((size_t)n * G + g) * 2 + 1] = rstd;
  }
  __syncthreads();
  if (!active) return;
  const float rstd = grstd[gl];
  float a[EPV], b[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int ch = ch0 + e;
    float f = 1.f, tb = 0.f;
    if (film) { f = 1.f + to_f32(film[(size_t)n * 2 * C + ch]); tb = to_f32(film[(size_t)n * 2 * C + C + ch]); }
    const float ga = gamma[ch], be = beta[ch];
    a[e] = ga * f * rstd;
    b[e] = (be - mu * rstd * ga) * f + tb;
  }
  if (r == 0) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      coef[((size_t)n * C + ch0 + e) * 2] = a[e];
      coef[((size_t)n * C + ch0 + e) * 2 + 1] = b[e];
    }
  }
  T* yn = y + (size_t)n * HW * C + ch0;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int p = r + it * R;
    if (p < HW) {
      Chunk<T> v;
      v.load(reinterpret_cast<const T*>(&raw[it]));
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const float z = a[e] * v.v[e] + b[e];
        v.v[e] = ACT ? silu_f(z) : z;
      }
      v.store(yn + (size_t)p * C);
    }
  }
}

template <typename T, int ACT, int NI, int NTHR>
__global__ __launch_bounds__(NTHR) void gn_fused_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const T* __restrict__ film, const float* __restrict__ stats,
                                                            const float* __restrict__ coef, T* __restrict__ dx,
                                                            T* __restrict__ dfilm, float* __restrict__ pgrad,
                                                            const T* __restrict__ dres, int HW, int C, int G, int CB) {
  constexpr int EPV = Tr<T>::EPV, LPR = GnF<T>::LPR, R = NTHR / LPR, NW = NTHR / 64;
  __shared__ float sh[NW][LPR][2 * EPV];
  __shared__ float tot[LPR][2 * EPV];
  __shared__ float sg[LPR][2];
  const int n = blockIdx.x, cb0 = blockIdx.y * CB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid % LPR, r = tid / LPR;
  const int cpg = C / G, cpc = cpg / EPV;
  const bool active = c * EPV < CB;
  const int ch0 = active ? cb0 + c * EPV : cb0;
  const int gl = active ? (c * EPV) / cpg : 0;
  const size_t base = (size_t)n * HW * C + ch0;
  float a[EPV], b[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    a[e] = coef[((size_t)n * C + ch0 + e) * 2];
    b[e] = coef[((size_t)n * C + ch0 + e) * 2 + 1];
  }
  uint4 rx[NI], rd[NI];   // x and dy of the slice, storage format
  float A1[EPV], A2[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) { A1[e] = 0.f; A2[e] = 0.f; }
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int p = r + it * R;
    rx[it] = uint4{0u, 0u, 0u, 0u}; rd[it] = uint4{0u, 0u, 0u, 0u};
    if (active && p < HW) {
      rx[it] = *reinterpret_cast<const uint4*>(x + base + (size_t)p * C);
      rd[it] = *reinterpret_cast<const uint4*>(dy + base + (size_t)p * C);
      Chunk<T> vx, vd;
      vx.load(reinterpret_cast<const T*>(&rx[it]));
      vd.load(reinterpret_cast<const T*>(&rd[it]));
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float dz = vd.v[e];
        if (ACT) dz *= dsilu_f(a[e] * vx.v[e] + b[e]);
        A1[e] += dz; A2[e] += dz * vx.v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) { A1[e] = rows_sum<LPR>(A1[e]); A2[e] = rows_sum<LPR>(A2[e]); }
  if (lane < LPR) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) { sh[wave][lane][e] = A1[e]; sh[wave][lane][EPV + e] = A2[e]; }
  }
  __syncthreads();
  if (tid < LPR * 2 * EPV) {
    const int cc = tid / (2 * EPV), e2 = tid % (2 * EPV);
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += sh[w][cc][e2];
    tot[cc][e2] = t;
  }
  __syncthreads();
  const int g = ch0 / cpg;
  const float mu = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
  float p1 = 0.f, p2 = 0.f;   // this chunk's part of the group sums  sum f*gamma*A1,  sum f*gamma*Xh
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int ch = ch0 + e;
    const float a1 = tot[c][e], a2 = tot[c][EPV + e];
    const float Xh = rstd * (a2 - mu * a1);
    float f = 1.f;
    if (film) f = 1.f + to_f32(film[(size_t)n * 2 * C + ch]);
    const float ga = gamma[ch], be = beta[ch];
    if (r == 0 && active) {
      pgrad[((size_t)n * C + ch) * 2] = f * Xh;
      pgrad[((size_t)n * C + ch) * 2 + 1] = f * a1;
      if (film) {
        dfilm[(size_t)n * 2 * C + ch] = from_f32<T>(ga * Xh + be * a1);
        dfilm[(size_t)n * 2 * C + C + ch] = from_f32<T>(a1);
      }
    }
    p1 += f * ga * a1;
    p2 += f * ga * Xh;
  }
  if (r == 0) { sg[c][0] = active ? p1 : 0.f; sg[c][1] = active ? p2 : 0.f; }
  __syncthreads();
  if (!active) return;
  float S1 = 0.f, S2 = 0.f;
  for (int cc = gl * cpc; cc < (gl + 1) * cpc; ++cc) { S1 += sg[cc][0]; S2 += sg[cc][1]; }
  const float m = (float)cpg * (float)HW;
  const float qq = -rstd * rstd * S2 / m;
  const float rr = -rstd * S1 / m - qq * mu;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int p = r + it * R;
    if (p < HW) {
      Chunk<T> vx, vd;
      vx.load(reinterpret_cast<const T*>(&rx[it]));
      vd.load(reinterpret_cast<const T*>(&rd[it]));
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float dz = vd.v[e];
        if (ACT) dz *= dsilu_f(a[e] * vx.v[e] + b[e]);
        vd.v[e] = a[e] * dz + qq * vx.v[e] + rr;
      }
      if (dres) {   // gradient that reaches x through the residual branch: added here instead of by a separate kernel
        Chunk<T> vr;
        vr.load(dres + base + (size_t)p * C);
#pragma unroll
        for (int e = 0; e < EPV; ++e) vd.v[e] += vr.v[e];
      }
      vd.store(dx + base + (size_t)p * C);
    }
  }
}

// channels per block of the fused kernels (0 = not applicable -> three-kernel path)
static int gn_fused_cb(int HW, int C, int G, int dtype, int max_ni) {
  static int enabled = -1;
  if (enabled < 0) { const char* e = getenv("MDM_HIP_GN_FUSED"); enabled = e ? atoi(e) : 1; }
  if (!enabled) return 0;
  const int epv = dtype == DT_F32 ? 4 : 8, lpr = dtype == DT_F32 ? 16 : 8;
  const int cpg = C / G;
  if (cpg % epv != 0 || cpg > lpr * epv) return 0;
  if (HW > max_ni * (1024 / lpr)) return 0;
  int gpb = (lpr * epv) / cpg;
  while (gpb > 1 && (G % gpb != 0 || gpb > 8)) --gpb;
  return cpg * gpb;
}

// ---------------------------------------------------------------------------------
// LayerNorm over the last dim D of [R, D]; one 256-thread block per row.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ stats, int D, float eps) {
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  const T* xr = x + row * D;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += to_f32(xr[i]);
  const float mu = block_sum(s, sh) / (float)D;
  float v = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) { const float d = to_f32(xr[i]) - mu; v += d * d; }
  const float rstd = rsqrtf(block_sum(v, sh) / (float)D + eps);
  if (threadIdx.x == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = rstd; }
  for (int i = threadIdx.x; i < D; i += 256)
    y[row * D + i] = from_f32<T>((to_f32(xr[i]) - mu) * rstd * gamma[i] + beta[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     T* __restrict__ dx, int D) {
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  const float mu = stats[row * 2], rstd = stats[row * 2 + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float g = to_f32(dy[row * D + i]) * gamma[i];
    const float xh = (to_f32(x[row * D + i]) - mu) * rstd;
    s1 += g; s2 += g * xh;
  }
  s1 = block_sum(s1, sh) / (float)D;
  s2 = block_sum(s2, sh) / (float)D;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float g = to_f32(dy[row * D + i]) * gamma[i];
    const float xh = (to_f32(x[row * D + i]) - mu) * rstd;
    dx[row * D + i] = from_f32<T>(rstd * (g - s1 - xh * s2));
  }
}

// dgamma[i] = sum_r dy*xhat, dbeta[i] = sum_r dy : partial per row-slab, then final
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_param_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const float* __restrict__ stats,
                                                                   float* __restrict__ part, int R, int D,
                                                                   int rows_per_block) {
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float g = to_f32(dy[(size_t)r * D + i]);
    a += g * (to_f32(x[(size_t)r * D + i]) - stats[r * 2]) * stats[r * 2 + 1];
    b += g;
  }
  part[((size_t)blockIdx.y * D + i) * 2] = a;
  part[((size_t)blockIdx.y * D + i) * 2 + 1] = b;
}
__global__ void ln_bwd_param_final_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int nslabs, int D, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D) return;
  float a = 0.f, b = 0.f;
  for (int s = 0; s < nslabs; ++s) { a += part[((size_t)s * D + i) * 2]; b += part[((size_t)s * D + i) * 2 + 1]; }
  dgamma[i] = accumulate ? dgamma[i] + a : a;
  dbeta[i] = accumulate ? dbeta[i] + b : b;
}

}  // namespace mdm

using namespace mdm;

static inline int gn_slabs(int N, int HW) {
  int s = (1024 + N - 1) / N;
  const int max_s = (HW + 15) / 16;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

// workspace (bytes, fp32): part [N][slabs][C][2] + pgrad [N][C][2] + qr [N][G][2]
extern "C" int mdm_gn_plan(int N, int HW, int C, int G, size_t* ws_bytes) {
  MDM_CHECK_ARG(ws_bytes);
  const int slabs = gn_slabs(N, HW);
  *ws_bytes = ((size_t)N * slabs * C * 2 + (size_t)N * C * 2 + (size_t)N * G * 2) * sizeof(float);
  return 0;
}

// y = act(GN(x) * (1 + ta) + tb);  stats [N][G][2] (mean, rstd) and coef [N][C][2] are saved for backward.
extern "C" int mdm_gn_fwd(const void* x, const float* gamma, const float* beta, const void* film, void* y,
                          float* stats, float* coef, float* ws, int N, int HW, int C, int G, float eps, int act,
                          int dtype, void* stream) {
  MDM_CHECK_ARG(x && gamma && beta && y && stats && coef && ws);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0 && C % G == 0 && C <= 2048 && G <= 256);
  MDM_CHECK_ARG(act == 0 || act == 1);
  const int slabs = gn_slabs(N, HW);
  const int pps = (HW + slabs - 1) / slabs;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (const int cb = gn_fused_cb(HW, C, G, dtype, 8)) {
    const dim3 grid(N, C / cb);
    const int lpr = dtype == DT_F32 ? 16 : 8;
    const int nthr = HW <= 8 * (256 / lpr) ? 256 : HW <= 8 * (512 / lpr) ? 512 : 1024;   // smallest block with <= 8 passes
#define MDM_GN_FUSED_FWD(TT, ACT)                                                                                  \
    if (nthr == 256) hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, ACT, 8, 256>), grid, dim3(256), 0, st, (const TT*)x, gamma, beta, (const TT*)film, (TT*)y, stats, coef, HW, C, G, cb, eps); \
    else if (nthr == 512) hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, ACT, 8, 512>), grid, dim3(512), 0, st, (const TT*)x, gamma, beta, (const TT*)film, (TT*)y, stats, coef, HW, C, G, cb, eps); \
    else hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, ACT, 8, 1024>), grid, dim3(1024), 0, st, (const TT*)x, gamma, beta, (const TT*)film, (TT*)y, stats, coef, HW, C, G, cb, eps)
    if (dtype == DT_F32) { if (act) { MDM_GN_FUSED_FWD(float, 1); } else { MDM_GN_FUSED_FWD(float, 0); } }
    else { if (act) { MDM_GN_FUSED_FWD(bf16, 1); } else { MDM_GN_FUSED_FWD(bf16, 0); } }
#undef MDM_GN_FUSED_FWD
    MDM_LAUNCH_STATUS();
  }
  const size_t total_chunks = (size_t)N * HW * C / epv;
  const int ab = (int)((total_chunks + 255) / 256 > 16384 ? 16384 : (total_chunks + 255) / 256);
  if (dtype == DT_F32) {
    hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(N * slabs), dim3(256), 0, st, (const float*)x, ws, HW, C, slabs, pps);
    hipLaunchKernelGGL(gn_finalize_kernel<float>, dim3(N), dim3(256), 0, st, (const float*)x, ws, gamma, beta, (const float*)film, stats, coef, HW, C, G, slabs, eps);
    if (act) hipLaunchKernelGGL((gn_apply_kernel<float, 1>), dim3(ab), dim3(256), 0, st, (const float*)x, coef, (float*)y, HW, C, total_chunks);
    else hipLaunchKernelGGL((gn_apply_kernel<float, 0>), dim3(ab), dim3(256), 0, st, (const float*)x, coef, (float*)y, HW, C, total_chunks);
  } else {
    hipLaunchKernelGGL(gn_partial_kernel<bf16>, dim3(N * slabs), dim3(256), 0, st, (const bf16*)x, ws, HW, C, slabs, pps);
    hipLaunchKernelGGL(gn_finalize_kernel<bf16>, dim3(N), dim3(256), 0, st, (const bf16*)x, ws, gamma, beta, (const bf16*)film, stats, coef, HW, C, G, slabs, eps);
    if (act) hipLaunchKernelGGL((gn_apply_kernel<bf16, 1>), dim3(ab), dim3(256), 0, st, (const bf16*)x, coef, (bf16*)y, HW, C, total_chunks);
    else hipLaunchKernelGGL((gn_apply_kernel<bf16, 0>), dim3(ab), dim3(256), 0, st, (const bf16*)x, coef, (bf16*)y, HW, C, total_chunks);
  }
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_gn_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const void* film,
                          const float* stats, const float* coef, const void* dres, void* dx, float* dgamma,
                          float* dbeta, void* dfilm, float* ws, int N, int HW, int C, int G, int act, int accumulate,
                          int dtype, void* stream) {
  MDM_CHECK_ARG(dy && x && gamma && beta && stats && coef && dx && dgamma && dbeta && ws);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  MDM_CHECK_ARG((film == nullptr) == (dfilm == nullptr));
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0 && C % G == 0 && C <= 2048 && G <= 256);
  const int slabs = gn_slabs(N, HW);
  const int pps = (HW + slabs - 1) / slabs;
  float* part = ws;
  float* pgrad = ws + (size_t)N * slabs * C * 2;
  float* qr = pgrad + (size_t)N * C * 2;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // x AND dy stay in registers here: 4 passes at most (beyond that the register file spills; three-kernel path)
  if (const int cb = gn_fused_cb(HW, C, G, dtype, 4)) {
    const dim3 grid(N, C / cb);
    const int lpr = dtype == DT_F32 ? 16 : 8;
    const int nthr = HW <= 4 * (512 / lpr) ? 512 : 1024;
#define MDM_GN_FUSED_BWD(TT, ACT)                                                                                  \
    if (nthr == 512) hipLaunchKernelGGL((gn_fused_bwd_kernel<TT, ACT, 4, 512>), grid, dim3(512), 0, st, (const TT*)dy, (const TT*)x, gamma, beta, (const TT*)film, stats, coef, (TT*)dx, (TT*)dfilm, pgrad, (const TT*)dres, HW, C, G, cb); \
    else hipLaunchKernelGGL((gn_fused_bwd_kernel<TT, ACT, 4, 1024>), grid, dim3(1024), 0, st, (const TT*)dy, (const TT*)x, gamma, beta, (const TT*)film, stats, coef, (TT*)dx, (TT*)dfilm, pgrad, (const TT*)dres, HW, C, G, cb)
    if (dtype == DT_F32) { if (act) { MDM_GN_FUSED_BWD(float, 1); } else { MDM_GN_FUSED_BWD(float, 0); } }
    else { if (act) { MDM_GN_FUSED_BWD(bf16, 1); } else { MDM_GN_FUSED_BWD(bf16, 0); } }
#undef MDM_GN_FUSED_BWD
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 31) / 32), dim3(256), 0, st, pgrad, dgamma, dbeta, N, C, accumulate);
    MDM_LAUNCH_STATUS();
  }
  const size_t total_chunks = (size_t)N * HW * C / epv;
  const int ab = (int)((total_chunks + 255) / 256 > 16384 ? 16384 : (total_chunks + 255) / 256);
#define MDM_GN_BWD(TT, ACT)                                                                                          \
  hipLaunchKernelGGL((gn_bwd_partial_kernel<TT, ACT>), dim3(N * slabs), dim3(256), 0, st, (const TT*)dy,             \
                     (const TT*)x, coef, part, HW, C, slabs, pps);                                                   \
  hipLaunchKernelGGL(gn_bwd_finalize_kernel<TT>, dim3(N), dim3(256), 0, st, part, stats, gamma, beta,                \
                     (const TT*)film, qr, (TT*)dfilm, pgrad, HW, C, G, slabs);                                       \
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((C + 31) / 32), dim3(256), 0, st, pgrad, dgamma, dbeta, N, C,        \
                     accumulate);                                                                                    \
  hipLaunchKernelGGL((gn_bwd_apply_kernel<TT, ACT>), dim3(ab), dim3(256), 0, st, (const TT*)dy, (const TT*)x, coef, \
                     qr, (TT*)dx, (const TT*)dres, HW, C, G, total_chunks);
  if (dtype == DT_F32) { if (act) { MDM_GN_BWD(float, 1) } else { MDM_GN_BWD(float, 0) } }
  else { if (act) { MDM_GN_BWD(bf16, 1) } else { MDM_GN_BWD(bf16, 0) } }
#undef MDM_GN_BWD
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int R, int D,
                          float eps, int dtype, void* stream) {
  MDM_CHECK_ARG(x && gamma && beta && y && stats && R > 0 && D > 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y, stats, D, eps);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(R), dim3(256), 0, st, (const bf16*)x, gamma, beta, (bf16*)y, stats, D, eps);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

// ws: fp32 [ceil(R/64)][D][2]
extern "C" int mdm_ln_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                          float* dgamma, float* dbeta, float* ws, int R, int D, int accumulate, int dtype,
                          void* stream) {
  MDM_CHECK_ARG(dy && x && gamma && stats && dx && dgamma && dbeta && ws);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int rpb = 64, nsl = (R + rpb - 1) / rpb;
  if (dtype == DT_F32) {
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)dy, (const float*)x, gamma, stats, (float*)dx, D);
    hipLaunchKernelGGL(ln_bwd_param_partial_kernel<float>, dim3((D + 255) / 256, nsl), dim3(256), 0, st, (const float*)dy, (const float*)x, stats, ws, R, D, rpb);
  } else if (dtype == DT_BF16) {
    hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3(R), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, gamma, stats, (bf16*)dx, D);
    hipLaunchKernelGGL(ln_bwd_param_partial_kernel<bf16>, dim3((D + 255) / 256, nsl), dim3(256), 0, st, (const bf16*)dy, (const bf16*)x, stats, ws, R, D, rpb);
  } else MDM_CHECK_ARG(false);
  hipLaunchKernelGGL(ln_bwd_param_final_kernel, dim3((D + 255) / 256), dim3(256), 0, st, ws, dgamma, dbeta, nsl, D, accumulate);
  MDM_LAUNCH_STATUS();
}
