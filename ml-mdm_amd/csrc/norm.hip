// GroupNorm (+FiLM, +SiLU) and LayerNorm forward/backward for NHWC activations.
//
// Reference semantics:
//   ResNet.forward      h = silu(norm1(x)); h = silu(norm2(h) * (1 + ta) + tb)   ml_mdm/models/unet.py:223-233
//   SelfAttention       qkv(norm(x)), ffn[0] = GroupNorm(32, C)                   unet.py:259,268,300
//   UNet output         silu(norm_out(x))                                          unet.py:877-879
//   cross-attn cond     norm_cond = LayerNorm(cond_dim)                            unet.py:263,304
// nn.GroupNorm(G, C, eps=1e-5, affine), biased variance.
//
// All kernels here are HBM-bound streaming kernels: 16-byte loads, fp32 math.  Activation-sized reductions are
// deterministic (fixed-order shuffles / LDS / per-slab partials); only the per-channel parameter gradients
// (dgamma, dbeta: a sum over the batch of per-sample terms) are accumulated with fp32 hardware atomics.
//
// Small images (a (sample, <= 64-channel slice) fits one block's registers): ONE kernel each way.
// Large images: TWO kernels each way --
//   forward   gn_partial (reads x) -> gn_apply (per-block finalize of its channel slice, reads x, writes y)
//     y = act(a[n,c] * x + b[n,c]),  a = gamma*f*rstd, b = (beta - mean*rstd*gamma)*f + tb, f = 1 + ta
//   backward  gn_bwd_partial (reads dy, x) -> gn_bwd_apply (per-block finalize, reads dy, x, dres, writes dx)
//     dz = dy * act'(a*x + b);  dx = a*dz + q[n,g]*x + r[n,g]
// Round 1 ran the finalize steps (and the batch reduction of dgamma / dbeta) as separate tiny kernels: 0.4 MB of
// traffic each, but 50-115 us apiece in the train step, because a dependent 24-block launch on the critical stream
// queues behind the weight-gradient grids of the side stream.  Every block of the apply kernels now redoes the
// finalize arithmetic for its own 64-channel slice from the slab partials (8 KB, L2-resident) instead.
#include <stdlib.h>

#include "common.hpp"

namespace mdm {

template <int N> struct GnInt { static constexpr int value = N; };   // compile-time operand counts / flags of the store loops


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
      // four pixel rows per trip: the four loads are in flight together (a slab is a few thousand rows per thread
      // column at batch 1-16, and one 16-byte load per round trip left the memory system idle); same summation order
      for (int p = p_begin + tr; p < p_end; p += 4 * rows_par) {
        uint4 raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pu = p + u * rows_par;
          raw[u] = pu < p_end ? *reinterpret_cast<const uint4*>(xn + (size_t)pu * C + (size_t)(c0 + tc) * EPV) : uint4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (p + u * rows_par < p_end) {
            Chunk<T> ch;
            ch.load(reinterpret_cast<const T*>(&raw[u]));
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
              const float d = ch.v[e] - k.v[e];
              s1[e] += d; s2[e] += d * d;
            }
          }
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

// ---- stage 2: finalize (per block, for its channel slice) + y = act(a * x + b) ---------------------------
// grid (pixel splits, C / CB, N).  A block owns CB channels (whole groups; 128 bytes of every pixel row when the
// group width allows) of sample n and the pixels [px0, px1).  Prologue: per-channel slab sums -> group mean / rstd
// (the shifted-sum formulas of the partial kernel) -> coefficients a, b; the block with pixel split 0 also
// stores stats / coef for the backward pass.
constexpr int GN_MAXCB = 128;   // channels per block slice (upper bound; host picks CB <= this)

// (sum over slabs of part[n][slab][cb0 + c][0..1]) for the CB channels of a block, by all 256 threads: thread t sums the
// slabs t / CB, t / CB + 256 / CB, ... of channel t % CB four loads at a time, the 256 / CB partial sums are added in a
// fixed order.  One thread per channel walking the slabs serially (one dependent L2 round trip each) capped the slab
// count at 32 per sample, which left the partial stage with 128 blocks at batch 4.
__device__ __forceinline__ void gn_slab_sums(const float* __restrict__ part, int n, int slabs, int C, int cb0, int CB,
                                             float* __restrict__ red /* [256][2] */, float* __restrict__ o1,
                                             float* __restrict__ o2) {
  const int tid = threadIdx.x;
  const int nsub = 256 / CB;                      // CB <= 128 -> nsub >= 2
  const int c = tid % CB, sub = tid / CB;
  float a = 0.f, b = 0.f;
  if (sub < nsub) {
    const float* pp = part + ((size_t)n * slabs * C + cb0 + c) * 2;
    int s_ = sub;
    for (; s_ + 3 * nsub < slabs; s_ += 4 * nsub) {
      const float2 v0 = *reinterpret_cast<const float2*>(pp + (size_t)s_ * C * 2);
      const float2 v1 = *reinterpret_cast<const float2*>(pp + (size_t)(s_ + nsub) * C * 2);
      const float2 v2 = *reinterpret_cast<const float2*>(pp + (size_t)(s_ + 2 * nsub) * C * 2);
      const float2 v3 = *reinterpret_cast<const float2*>(pp + (size_t)(s_ + 3 * nsub) * C * 2);
      a += v0.x; b += v0.y; a += v1.x; b += v1.y; a += v2.x; b += v2.y; a += v3.x; b += v3.y;
    }
    for (; s_ < slabs; s_ += nsub) {
      const float2 v = *reinterpret_cast<const float2*>(pp + (size_t)s_ * C * 2);
      a += v.x; b += v.y;
    }
  }
  red[tid * 2] = a; red[tid * 2 + 1] = b;
  __syncthreads();
  if (tid < CB) {
    float ta = 0.f, tb = 0.f;
    for (int j = 0; j < nsub; ++j) { ta += red[(j * CB + tid) * 2]; tb += red[(j * CB + tid) * 2 + 1]; }
    o1[tid] = ta; o2[tid] = tb;
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const T* __restrict__ film, T* __restrict__ y,
                                                       float* __restrict__ stats, float* __restrict__ coef, int HW, int C,
                                                       int G, int CB, int slabs, int pix_per_block, float eps) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float sh_s1[GN_MAXCB], sh_s2[GN_MAXCB], sh_k[GN_MAXCB], sh_a[GN_MAXCB], sh_b[GN_MAXCB];
  __shared__ float sh_mean[GN_MAXCB], sh_rstd[GN_MAXCB];
  const int n = blockIdx.z, cb0 = blockIdx.y * CB, tid = threadIdx.x;
  const int cpg = C / G, gpb = CB / cpg;
  const T* xn = x + (size_t)n * HW * C;
  __shared__ float sh_red[512];
  gn_slab_sums(part, n, slabs, C, cb0, CB, sh_red, sh_s1, sh_s2);
  if (tid < CB) sh_k[tid] = to_f32(xn[cb0 + tid]);
  __syncthreads();
  const float cnt = (float)HW;
  if (tid < gpb) {
    float mu = 0.f;
    for (int j = 0; j < cpg; ++j) { const int c = tid * cpg + j; mu += sh_k[c] + sh_s1[c] / cnt; }
    mu /= (float)cpg;
    float var = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int c = tid * cpg + j;
      const float d = sh_k[c] - mu;
      var += sh_s2[c] + 2.f * d * sh_s1[c] + cnt * d * d;
    }
    var = fmaxf(var / (cnt * (float)cpg), 0.f);
    const float rstd = rsqrtf(var + eps);
    sh_mean[tid] = mu; sh_rstd[tid] = rstd;
    if (blockIdx.x == 0) {
      const int g = cb0 / cpg + tid;
      stats[((size_t)n * G + g) * 2] = mu;
      stats[((size_t)n * G + g) * 2 + 1] = rstd;
    }
  }
  __syncthreads();
  if (tid < CB) {
    const int c = cb0 + tid, gl = tid / cpg;
    const float mu = sh_mean[gl], rstd = sh_rstd[gl];
    float f = 1.f, tb = 0.f;
    if (film) { f = 1.f + to_f32(film[(size_t)n * 2 * C + c]); tb = to_f32(film[(size_t)n * 2 * C + C + c]); }
    const float ga = gamma[c], be = beta[c];
    const float a = ga * f * rstd, b = (be - mu * rstd * ga) * f + tb;
    sh_a[tid] = a; sh_b[tid] = b;
    if (blockIdx.x == 0) {
      coef[((size_t)n * C + c) * 2] = a;
      coef[((size_t)n * C + c) * 2 + 1] = b;
    }
  }
  __syncthreads();
  const int lanes = CB / EPV, rows_par = 256 / lanes;
  const int cl = tid % lanes, rl = tid / lanes;
  if (rl >= rows_par) return;
  float a[EPV], b[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) { a[e] = sh_a[cl * EPV + e]; b[e] = sh_b[cl * EPV + e]; }
  const int px0 = blockIdx.x * pix_per_block, px1 = min(HW, px0 + pix_per_block);
  const size_t base = (size_t)n * HW * C + cb0 + cl * EPV;
  for (int p = px0 + rl; p < px1; p += 4 * rows_par) {
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = p + u * rows_par;
      if (pu < px1) raw[u] = *reinterpret_cast<const uint4*>(x + base + (size_t)pu * C);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = p + u * rows_par;
      if (pu < px1) {
        Chunk<T> ch;
        ch.load(reinterpret_cast<const T*>(&raw[u]));
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const float z = a[e] * ch.v[e] + b[e];
          ch.v[e] = ACT ? silu_f(z) : z;
        }
        ch.store(y + base + (size_t)pu * C);
      }
    }
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
      for (int p = p_begin + tr; p < p_end; p += 4 * rows_par) {   // four rows (eight loads) in flight, same summation order
        uint4 rx[4], rd[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pu = p + u * rows_par;
          if (pu < p_end) {
            const size_t off = nbase + (size_t)pu * C + (size_t)(c0 + tc) * EPV;
            rx[u] = *reinterpret_cast<const uint4*>(x + off);
            rd[u] = *reinterpret_cast<const uint4*>(dy + off);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (p + u * rows_par < p_end) {
            Chunk<T> cx, cd;
            cx.load(reinterpret_cast<const T*>(&rx[u]));
            cd.load(reinterpret_cast<const T*>(&rd[u]));
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
              float dz = cd.v[e];
              if (ACT) dz *= dsilu_f(ab[2 * e] * cx.v[e] + ab[2 * e + 1]);
              a1[e] += dz; a2[e] += dz * cx.v[e];
            }
          }
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

// ---- backward stage 2: finalize (per block, for its channel slice) + dx = a*dz + q*x + r (+ dres) ---------
// Same decomposition as gn_apply_kernel.  The block with pixel split 0 also emits the parameter-side results of
// its slice: dfilm[n] (stores) and this sample's terms of dgamma / dbeta -- fp32 atomic adds into the parameter's
// gradient-arena slot or a zero-filled buffer, or (pstride != 0) plain stores into per-sample rows [N][pstride] that
// mdm_gn_param_reduce_multi sums later.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ part, const float* __restrict__ stats,
                                                           const float* __restrict__ coef, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ film,
                                                           const T* __restrict__ dres, const T* __restrict__ dres2,
                                                           T* __restrict__ dx,
                                                           T* __restrict__ dfilm, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int HW, int C, int G, int CB,
                                                           int slabs, int pix_per_block, int pstride) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float sh_fgA1[GN_MAXCB], sh_fgXh[GN_MAXCB], sh_a[GN_MAXCB], sh_b[GN_MAXCB], sh_q[GN_MAXCB], sh_r[GN_MAXCB];
  const int n = blockIdx.z, cb0 = blockIdx.y * CB, tid = threadIdx.x;
  const int cpg = C / G, gpb = CB / cpg;
  __shared__ float sh_red[512], sh_A1[GN_MAXCB], sh_A2[GN_MAXCB];
  gn_slab_sums(part, n, slabs, C, cb0, CB, sh_red, sh_A1, sh_A2);
  if (tid < CB) {
    const int c = cb0 + tid;
    const float A1 = sh_A1[tid], A2 = sh_A2[tid];   // written by this same thread
    const int g = c / cpg;
    const float mu = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
    const float Xh = rstd * (A2 - mu * A1);
    float f = 1.f;
    if (film) f = 1.f + to_f32(film[(size_t)n * 2 * C + c]);
    const float ga = gamma[c], be = beta[c];
    if (blockIdx.x == 0) {
      if (pstride) {   // per-sample terms, summed later by gn_param_reduce_multi_kernel
        dgamma[(size_t)n * pstride + c] = f * Xh;
        dbeta[(size_t)n * pstride + c] = f * A1;
      } else {
        unsafeAtomicAdd(dgamma + c, f * Xh);
        unsafeAtomicAdd(dbeta + c, f * A1);
      }
      if (film) {
        dfilm[(size_t)n * 2 * C + c] = from_f32<T>(ga * Xh + be * A1);
        dfilm[(size_t)n * 2 * C + C + c] = from_f32<T>(A1);
      }
    }
    sh_fgA1[tid] = f * ga * A1;
    sh_fgXh[tid] = f * ga * Xh;
    sh_a[tid] = coef[((size_t)n * C + c) * 2];
    sh_b[tid] = coef[((size_t)n * C + c) * 2 + 1];
  }
  __syncthreads();
  if (tid < gpb) {
    float S1 = 0.f, S2 = 0.f;
    for (int j = 0; j < cpg; ++j) { S1 += sh_fgA1[tid * cpg + j]; S2 += sh_fgXh[tid * cpg + j]; }
    const int g = cb0 / cpg + tid;
    const float mu = stats[((size_t)n * G + g) * 2], rstd = stats[((size_t)n * G + g) * 2 + 1];
    const float m = (float)cpg * (float)HW;
    const float q = -rstd * rstd * S2 / m;
    sh_q[tid] = q;
    sh_r[tid] = -rstd * S1 / m - q * mu;
  }
  __syncthreads();
  const int lanes = CB / EPV, rows_par = 256 / lanes;
  const int cl = tid % lanes, rl = tid / lanes;
  if (rl >= rows_par) return;
  float a[EPV], b[EPV], q[EPV], r[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int c = cl * EPV + e;
    a[e] = sh_a[c]; b[e] = sh_b[c]; q[e] = sh_q[c / cpg]; r[e] = sh_r[c / cpg];
  }
  const int px0 = blockIdx.x * pix_per_block, px1 = min(HW, px0 + pix_per_block);
  const size_t base = (size_t)n * HW * C + cb0 + cl * EPV;
  // Four pixel rows per trip, every operand of the four requested before the first is used (8-16 loads of 16 bytes in
  // flight per thread): one row per trip -- two dependent-latency loads, then the arithmetic, then the store -- left this
  // kernel at 2.3-3.1 TB/s of its own traffic where the forward apply kernel (same map, four rows deep) reaches 5.
  // (Round 6: software-pipelining the trips over two register sets, so that a trip's operands are requested before the
  // previous trip's stores -- gfx950's single in-order vmcnt makes a later load wait for them -- cost 70 more registers and
  // was 3-8 % SLOWER on cold buffers: with 8-16 waves per SIMD other waves already cover the acknowledgement round trip.)
  for (int p = px0 + rl; p < px1; p += 4 * rows_par) {
    uint4 rx[4], rd[4], rr[4], rs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = p + u * rows_par;
      if (pu < px1) {
        const size_t off = base + (size_t)pu * C;
        rx[u] = *reinterpret_cast<const uint4*>(x + off);
        rd[u] = *reinterpret_cast<const uint4*>(dy + off);
        if (dres) rr[u] = *reinterpret_cast<const uint4*>(dres + off);
        if (dres2) rs[u] = *reinterpret_cast<const uint4*>(dres2 + off);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = p + u * rows_par;
      if (pu < px1) {
        const size_t off = base + (size_t)pu * C;
        Chunk<T> cx, cd;
        cx.load(reinterpret_cast<const T*>(&rx[u]));
        cd.load(reinterpret_cast<const T*>(&rd[u]));
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float dz = cd.v[e];
          if (ACT) dz *= dsilu_f(a[e] * cx.v[e] + b[e]);
          cd.v[e] = a[e] * dz + q[e] * cx.v[e] + r[e];
        }
        if (dres) {
          Chunk<T> cr;
          cr.load(reinterpret_cast<const T*>(&rr[u]));
#pragma unroll
          for (int e = 0; e < EPV; ++e) cd.v[e] += cr.v[e];
        }
        if (dres2) {   // a second consumer of x outside the block (the U-Net's skip connection)
          Chunk<T> cr;
          cr.load(reinterpret_cast<const T*>(&rs[u]));
#pragma unroll
          for (int e = 0; e < EPV; ++e) cd.v[e] += cr.v[e];
        }
        cd.store(dx + off);
      }
    }
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

template <typename T, int ACT, int NI, int NTHR, int LPR = GnF<T>::LPR>
__global__ __launch_bounds__(NTHR) void gn_fused_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ film,
                                                            T* __restrict__ y, float* __restrict__ stats,
                                                            float* __restrict__ coef, int HW, int C, int G, int CB,
                                                            float eps) {
  constexpr int EPV = Tr<T>::EPV, R = NTHR / LPR, NW = NTHR / 64;
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

template <typename T, int ACT, int NI, int NTHR, int LPR = GnF<T>::LPR>
__global__ __launch_bounds__(NTHR) void gn_fused_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const T* __restrict__ film, const float* __restrict__ stats,
                                                            const float* __restrict__ coef, T* __restrict__ dx,
                                                            T* __restrict__ dfilm, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const T* __restrict__ dres,
                                                            const T* __restrict__ dres2,
                                                            int HW, int C, int G, int CB, int pstride) {
  constexpr int EPV = Tr<T>::EPV, R = NTHR / LPR, NW = NTHR / 64;
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
  // (the residual-branch gradients are read in the store loop, after the two block-wide reductions.  Requesting them up here
  // with x and dy -- 16 more registers, three waves per SIMD instead of four -- was slower on cold buffers: 43 / 83 us against
  // 36 / 67 at 16x16, C = 768 / 1536; r05)
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
  for (int i = tid; i < LPR * 2 * EPV; i += NTHR) {
    const int cc = i / (2 * EPV), e2 = i % (2 * EPV);
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
      // this sample's term of the parameter gradients: atomics into a gradient-arena slot / zero-filled buffer, or a
      // plain store into the per-sample rows (64 samples x 1536 addresses of atomics cost as much as the kernel's own
      // 75 MB of traffic: 45 us against 24 us at 16x16x768)
      if (pstride) {
        dgamma[(size_t)n * pstride + ch] = f * Xh;
        dbeta[(size_t)n * pstride + ch] = f * a1;
      } else {
        unsafeAtomicAdd(dgamma + ch, f * Xh);
        unsafeAtomicAdd(dbeta + ch, f * a1);
      }
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
  // The residual-branch gradients (dres: the block's own residual; dres2: a second consumer of x outside the block, the
  // U-Net's skip connection) are added here instead of by a separate kernel.  gfx950 counts loads and stores in ONE in-order
  // counter (vmcnt): a load issued after a store cannot be waited for without waiting for the store's acknowledgement as
  // well, so the pass-by-pass form (load, wait, add, store) of rounds 3-5 paid one full memory round trip per pass, one after
  // the other.  Now the operands are requested in BATCHES of NB passes, each batch before the stores of the batch ahead of
  // it, in straight-line code (a variant per operand count, chosen by block-uniform branches: across the per-pass branches of
  // a predicated loop hipcc falls back to vmcnt(0) at every use).  Registers: 2 x NB x 4 per operand beyond x and dy, live
  // only here (requesting everything up front with x and dy cost a wave per SIMD and was slower: r05).
  constexpr int NB = NI < 4 ? NI : (NI > 4 ? 2 : 4);
  auto tail = [&](auto nres_c, auto exact_c) {
    constexpr int NRES = decltype(nres_c)::value;
    constexpr bool EXACT = decltype(exact_c)::value;   // HW == NI * R: no pixel predicate anywhere
    uint4 q1[2][NB], q2[2][NB];
    auto fetch = [&](int it0, int buf) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int p = r + (it0 + j) * R;
        const size_t off = base + (size_t)(EXACT || p < HW ? p : 0) * C;   // (a row that is not stored reads row 0: no branch)
        if constexpr (NRES >= 1) q1[buf][j] = *reinterpret_cast<const uint4*>(dres + off);
        if constexpr (NRES >= 2) q2[buf][j] = *reinterpret_cast<const uint4*>(dres2 + off);
      }
    };
    if constexpr (NRES >= 1) fetch(0, 0);
#pragma unroll
    for (int it0 = 0; it0 < NI; it0 += NB) {
      const int cur = (it0 / NB) & 1;
      if constexpr (NRES >= 1) { if (it0 + NB < NI) fetch(it0 + NB, cur ^ 1); }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int it = it0 + j;
        const int p = r + it * R;
        Chunk<T> vx, vd;
        vx.load(reinterpret_cast<const T*>(&rx[it]));
        vd.load(reinterpret_cast<const T*>(&rd[it]));
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          float dz = vd.v[e];
          if (ACT) dz *= dsilu_f(a[e] * vx.v[e] + b[e]);
          vd.v[e] = a[e] * dz + qq * vx.v[e] + rr;
        }
        if constexpr (NRES >= 1) {
          Chunk<T> vr;
          vr.load(reinterpret_cast<const T*>(&q1[cur][j]));
#pragma unroll
          for (int e = 0; e < EPV; ++e) vd.v[e] += vr.v[e];
        }
        if constexpr (NRES >= 2) {
          Chunk<T> vr;
          vr.load(reinterpret_cast<const T*>(&q2[cur][j]));
#pragma unroll
          for (int e = 0; e < EPV; ++e) vd.v[e] += vr.v[e];
        }
        if (EXACT || p < HW) vd.store(dx + base + (size_t)p * C);
      }
    }
  };
  const bool exact = HW == NI * R;
  if (dres2) { if (exact) tail(GnInt<2>{}, GnInt<1>{}); else tail(GnInt<2>{}, GnInt<0>{}); }
  else if (dres) { if (exact) tail(GnInt<1>{}, GnInt<1>{}); else tail(GnInt<1>{}, GnInt<0>{}); }
  else { if (exact) tail(GnInt<0>{}, GnInt<1>{}); else tail(GnInt<0>{}, GnInt<0>{}); }
}

// Shape of a fused launch: channels per block (cb: whole groups, <= 8 of them), block size and passes.  cb = 0: not
// applicable -> three-kernel path.
// (Tried in round 5 for the 24 / 48-channel groups of C = 768 / 1536, whose 8 chunk lanes hold 48 channels = 96-byte row
// pieces with two lanes idle: blocks of 32 chunk lanes = 192 channels = 384-byte pieces on 128-byte boundaries.  Not
// faster on cold buffers -- backward 37 / 72 against 35 / 67 us at 16x16, forward 24 / 53 against 18 / 37: neighbouring
// blocks run together and share their lines in L2, and four times fewer blocks hide less latency.)
struct GnFusedCfg { int cb, nthr, ni; };
static GnFusedCfg gn_fused_cfg(int HW, int C, int G, int dtype, bool bwd) {
  const int epv = dtype == DT_F32 ? 4 : 8, lpr = dtype == DT_F32 ? 16 : 8;
  const int cpg = C / G;
  GnFusedCfg f = {0, 0, 0};
  if (cpg % epv != 0 || cpg > lpr * epv) return f;
  int gpb = (lpr * epv) / cpg;
  while (gpb > 1 && (G % gpb != 0 || gpb > 8)) --gpb;
  const int cb = cpg * gpb;
  // forward: x in registers (4 per pass); backward: x AND dy (8 per pass).  Smallest block that covers the image.
  // The backward's 16 passes of a 512-thread block (read once instead of the three-kernel path's twice) pay where the
  // row pieces are whole 128-byte lines: 32x32 level, cold buffers, C = 512 / 1024: 77 / 145 us against 85 / 170;
  // C = 768 / 1280 (96- / 80-byte pieces): 138 / 266 against 131 / 243.
  static const int fwd_opts[][2] = {{256, 8}, {512, 8}, {1024, 8}};
  static const int bwd_opts[][2] = {{512, 4}, {1024, 4}, {512, 16}};
  const int (*opts)[2] = bwd ? bwd_opts : fwd_opts;
  for (int i = 0; i < 3; ++i) {
    if (opts[i][1] == 16 && !(dtype == DT_BF16 && cb % 64 == 0)) continue;
    if (HW <= opts[i][1] * (opts[i][0] / lpr)) { f.cb = cb; f.nthr = opts[i][0]; f.ni = opts[i][1]; return f; }
  }
  return f;
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

// ---------------------------------------------------------------------------------
// One input, L LayerNorms (different gamma / beta): every attention layer normalises the SAME text states before its
// key / value projection (unet.py:263-264, 304).  Statistics once per row; L affine outputs; the backward sums the L
// input gradients and emits the L parameter gradients.
// ---------------------------------------------------------------------------------
constexpr int LN_MAXL = 32;
// N (4 or 8) consecutive fp32 parameters as 16-byte loads
template <int N> __device__ __forceinline__ void load4n(const float* p, float (&o)[N]) {
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
    o[4 * q] = t[0]; o[4 * q + 1] = t[1]; o[4 * q + 2] = t[2]; o[4 * q + 3] = t[3];
  }
}

struct LnGroup {
  const float* gamma[LN_MAXL];
  const float* beta[LN_MAXL];
  void* y[LN_MAXL];          // forward outputs / backward: upstream gradients
  float* dgamma[LN_MAXL];
  float* dbeta[LN_MAXL];
};

// Round 6: both kernels move 16-byte chunks (a thread owns NCK chunks of a row: D = 2048 is exactly one chunk per thread of
// a 256-thread block in bf16) instead of 2-byte elements -- 31 layers x 16 two-byte loads per thread and layer left the
// backward at 0.9 TB/s (288 us for 270 MB), the forward at 1.5 TB/s.  Host-checked: D % EPV == 0, D <= 256 * EPV * LN_NCK.
constexpr int LN_NCK = 4;
template <typename T>
__global__ __launch_bounds__(256) void ln_multi_fwd_kernel(const T* __restrict__ x, LnGroup g, int L, float* __restrict__ stats,
                                                           int D, float eps) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  const T* xr = x + row * D;
  const int nch = D / EPV;
  Chunk<T> xv[LN_NCK];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_NCK; ++k) {
    const int c = threadIdx.x + 256 * k;
#pragma unroll
    for (int e = 0; e < EPV; ++e) xv[k].v[e] = 0.f;
    if (c < nch) {
      xv[k].load(xr + (size_t)c * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) s += xv[k].v[e];
    }
  }
  const float mu = block_sum(s, sh) / (float)D;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < LN_NCK; ++k) {
    if (threadIdx.x + 256 * k < nch) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) { const float d = xv[k].v[e] - mu; v += d * d; }
    }
  }
  const float rstd = rsqrtf(block_sum(v, sh) / (float)D + eps);
  if (threadIdx.x == 0) { stats[row * 2] = mu; stats[row * 2 + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < LN_NCK; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c >= nch) continue;
#pragma unroll
    for (int e = 0; e < EPV; ++e) xv[k].v[e] = (xv[k].v[e] - mu) * rstd;
    for (int l = 0; l < L; ++l) {
      float ga[EPV], be[EPV];
      load4n<EPV>(g.gamma[l] + (size_t)c * EPV, ga);
      load4n<EPV>(g.beta[l] + (size_t)c * EPV, be);
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < EPV; ++e) o.v[e] = xv[k].v[e] * ga[e] + be[e];
      o.store(reinterpret_cast<T*>(g.y[l]) + row * D + (size_t)c * EPV);
    }
  }
}

// dx[row] = sum_l rstd * (g_l - mean(g_l) - xhat * mean(g_l * xhat)),  g_l = dy_l * gamma_l
template <typename T>
__global__ __launch_bounds__(256) void ln_multi_bwd_kernel(LnGroup g, int L, const T* __restrict__ x,
                                                           const float* __restrict__ stats, T* __restrict__ dx, int D) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float sh[4];
  const size_t row = blockIdx.x;
  const float mu = stats[row * 2], rstd = stats[row * 2 + 1];
  const int nch = D / EPV;
  float xh[LN_NCK][EPV], acc[LN_NCK][EPV];
#pragma unroll
  for (int k = 0; k < LN_NCK; ++k) {
    const int c = threadIdx.x + 256 * k;
    Chunk<T> xv;
#pragma unroll
    for (int e = 0; e < EPV; ++e) xv.v[e] = mu;
    if (c < nch) xv.load(x + row * D + (size_t)c * EPV);
#pragma unroll
    for (int e = 0; e < EPV; ++e) { xh[k][e] = (xv.v[e] - mu) * rstd; acc[k][e] = 0.f; }
  }
  for (int l = 0; l < L; ++l) {
    const T* dyr = reinterpret_cast<const T*>(g.y[l]) + row * D;
    float gl[LN_NCK][EPV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NCK; ++k) {
      const int c = threadIdx.x + 256 * k;
#pragma unroll
      for (int e = 0; e < EPV; ++e) gl[k][e] = 0.f;
      if (c < nch) {
        Chunk<T> dv;
        dv.load(dyr + (size_t)c * EPV);
        float ga[EPV];
        load4n<EPV>(g.gamma[l] + (size_t)c * EPV, ga);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { gl[k][e] = dv.v[e] * ga[e]; s1 += gl[k][e]; s2 += gl[k][e] * xh[k][e]; }
      }
    }
    s1 = block_sum(s1, sh) / (float)D;
    s2 = block_sum(s2, sh) / (float)D;
#pragma unroll
    for (int k = 0; k < LN_NCK; ++k)
#pragma unroll
      for (int e = 0; e < EPV; ++e) acc[k][e] += gl[k][e] - s1 - xh[k][e] * s2;
  }
#pragma unroll
  for (int k = 0; k < LN_NCK; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) {
      Chunk<T> o;
#pragma unroll
      for (int e = 0; e < EPV; ++e) o.v[e] = rstd * acc[k][e];
      o.store(dx + row * D + (size_t)c * EPV);
    }
  }
}

// dgamma_l[i] += sum_r dy_l[r, i] * xhat[r, i], dbeta_l[i] += sum_r dy_l[r, i]: grid (D / 256, row slabs, L); one fp32
// atomic pair per (slab, layer, column).  (Round 6: a 16-byte chunk of columns per thread instead -- an eighth of the threads
// -- was slower, 195 against 136 us: this kernel lives on its thread count.)
template <typename T>
__global__ __launch_bounds__(256) void ln_multi_param_kernel(LnGroup g, const T* __restrict__ x, const float* __restrict__ stats,
                                                             int R, int D, int rows_per_block) {
  const int l = blockIdx.z;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= D) return;
  const T* dy = reinterpret_cast<const T*>(g.y[l]);
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float gy = to_f32(dy[(size_t)r * D + i]);
    a += gy * (to_f32(x[(size_t)r * D + i]) - stats[r * 2]) * stats[r * 2 + 1];
    b += gy;
  }
  unsafeAtomicAdd(g.dgamma[l] + i, a);
  unsafeAtomicAdd(g.dbeta[l] + i, b);
}

}  // namespace mdm

using namespace mdm;

// bytes of x + dy above which the two-kernel backward walks the batch in chunks (mdm_dev_set_gn_chunk_mb; 0 = always chunk)
static size_t g_gn_chunk_bytes = (size_t)160 << 20;
extern "C" int mdm_dev_set_gn_chunk_mb(int mb) {
  g_gn_chunk_bytes = mb < 0 ? ~(size_t)0 : (size_t)mb << 20;
  return 0;
}

// pixel slabs of the partial-sum stage: about 1024 blocks in total, at least 64 pixels each, at most 256 per sample
// (round 2 capped them at 32 because every apply block walked the slabs of a channel serially -- 256 slabs at batch 4 made
// a 64 x 64 norm take 71 us; with gn_slab_sums the walk is 256 / CB-way parallel and four loads deep, and the cap that
// left a 1024 x 1024 x 32 norm at batch 4 with 128 partial blocks (1.5 TB/s) is gone)
static inline int gn_slabs(int N, int HW) {
  int s = (1024 + N - 1) / N;
  const int max_s = (HW + 63) / 64;
  if (s > max_s) s = max_s;
  if (s > 256) s = 256;   // the apply blocks add the slabs with all their threads (gn_slab_sums); 32 before
  if (s < 1) s = 1;
  return s;
}

// channel slice of an apply block: whole groups, whole 16-byte chunks, 128 bytes per pixel row when the group width
// allows it (64 bf16 / 32 fp32 channels), never more than GN_MAXCB; 0 = this (C, G) is not supported
static inline int gn_slice(int C, int G, int epv) {
  const int cpg = C / G, target = epv * 8;
  int best = 0;
  for (int j = 1; j <= G; ++j) {
    if (G % j != 0) continue;
    const int cb = cpg * j;
    if (cb % epv != 0 || cb > GN_MAXCB) continue;
    best = cb;                       // largest legal slice so far
    if (cb >= target) break;         // first one that fills a 128-byte line
  }
  return best;
}

// pixel splits of the apply grid: about 2048 blocks in total, at least 64 pixels per block
static inline int gn_pix_splits(int N, int HW, int cslices) {
  int sp = (2048 + N * cslices - 1) / (N * cslices);
  const int max_sp = (HW + 63) / 64;
  if (sp > max_sp) sp = max_sp;
  if (sp < 1) sp = 1;
  return sp;
}

// workspace (bytes, fp32): part [N][slabs][C][2]
extern "C" int mdm_gn_plan(int N, int HW, int C, int G, size_t* ws_bytes) {
  MDM_CHECK_ARG(ws_bytes);
  const int slabs = gn_slabs(N, HW);
  *ws_bytes = ((size_t)N * slabs * C * 2) * sizeof(float);
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
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (const GnFusedCfg f = gn_fused_cfg(HW, C, G, dtype, false); f.cb) {
    const dim3 grid(N, C / f.cb);
    const int cb = f.cb;
#define MDM_GN_FUSED_FWD_L(TT, ACT, NI_, NT_, LPR_)                                                                 \
    hipLaunchKernelGGL((gn_fused_fwd_kernel<TT, ACT, NI_, NT_, LPR_>), grid, dim3(NT_), 0, st, (const TT*)x, gamma, beta, (const TT*)film, (TT*)y, stats, coef, HW, C, G, cb, eps)
#define MDM_GN_FUSED_FWD(TT, ACT, LPR_)                                                                             \
    if (f.nthr == 256) MDM_GN_FUSED_FWD_L(TT, ACT, 8, 256, LPR_);                                                  \
    else if (f.nthr == 512) MDM_GN_FUSED_FWD_L(TT, ACT, 8, 512, LPR_);                                             \
    else MDM_GN_FUSED_FWD_L(TT, ACT, 8, 1024, LPR_)
    if (dtype == DT_F32) { if (act) { MDM_GN_FUSED_FWD(float, 1, 16); } else { MDM_GN_FUSED_FWD(float, 0, 16); } }
    else { if (act) { MDM_GN_FUSED_FWD(bf16, 1, 8); } else { MDM_GN_FUSED_FWD(bf16, 0, 8); } }
#undef MDM_GN_FUSED_FWD
#undef MDM_GN_FUSED_FWD_L
    MDM_LAUNCH_STATUS();
  }
  const int slabs = gn_slabs(N, HW);
  const int pps = (HW + slabs - 1) / slabs;
  const int cb = gn_slice(C, G, epv);
  MDM_CHECK_ARG(cb > 0);
  const int sp = gn_pix_splits(N, HW, C / cb);
  const int ppb = (HW + sp - 1) / sp;
  const dim3 agrid(sp, C / cb, N);
#define MDM_GN_FWD(TT, ACT)                                                                                          \
  hipLaunchKernelGGL(gn_partial_kernel<TT>, dim3(N * slabs), dim3(256), 0, st, (const TT*)x, ws, HW, C, slabs, pps); \
  hipLaunchKernelGGL((gn_apply_kernel<TT, ACT>), agrid, dim3(256), 0, st, (const TT*)x, ws, gamma, beta,             \
                     (const TT*)film, (TT*)y, stats, coef, HW, C, G, cb, slabs, ppb, eps);
  if (dtype == DT_F32) { if (act) { MDM_GN_FWD(float, 1) } else { MDM_GN_FWD(float, 0) } }
  else { if (act) { MDM_GN_FWD(bf16, 1) } else { MDM_GN_FWD(bf16, 0) } }
#undef MDM_GN_FWD
  MDM_LAUNCH_STATUS();
}

// dgamma / dbeta: accumulate = 0 zero-fills the destination first, 1 adds into it (a gradient-arena slot) -- the
// kernels add one term per sample with fp32 atomics; 2 = dgamma / dbeta are per-sample rows [N][C] that the kernels
// fill with plain stores (no atomics), to be summed by mdm_gn_param_reduce_multi.
extern "C" int mdm_gn_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const void* film,
                          const float* stats, const float* coef, const void* dres, const void* dres2, void* dx,
                          float* dgamma, float* dbeta, void* dfilm, float* ws, int N, int HW, int C, int G, int act,
                          int accumulate, int dtype, void* stream) {
  MDM_CHECK_ARG(dy && x && gamma && beta && stats && coef && dx && dgamma && dbeta && ws);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  MDM_CHECK_ARG((film == nullptr) == (dfilm == nullptr));
  if (!dres && dres2) { dres = dres2; dres2 = nullptr; }   // the kernels count operands: a second one implies the first
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0 && C % G == 0 && C <= 2048 && G <= 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  MDM_CHECK_ARG(accumulate >= 0 && accumulate <= 2);
  const int pstride = accumulate == 2 ? C : 0;
  if (!accumulate) {
    if (hipMemsetAsync(dgamma, 0, (size_t)C * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(dbeta, 0, (size_t)C * sizeof(float), st) != hipSuccess) {
      mdm_set_error(__FILE__, __LINE__, "hipMemsetAsync(dgamma / dbeta)");
      return (int)hipGetLastError();
    }
  }
  // x AND dy stay in registers here: 4 passes with 1024 threads, up to 16 with 512 (8 registers per pass)
  if (const GnFusedCfg f = gn_fused_cfg(HW, C, G, dtype, true); f.cb) {
    const dim3 grid(N, C / f.cb);
    const int cb = f.cb;
#define MDM_GN_FUSED_BWD_L(TT, ACT, NI_, NT_, LPR_)                                                                 \
    hipLaunchKernelGGL((gn_fused_bwd_kernel<TT, ACT, NI_, NT_, LPR_>), grid, dim3(NT_), 0, st, (const TT*)dy, (const TT*)x, gamma, beta, (const TT*)film, stats, coef, (TT*)dx, (TT*)dfilm, dgamma, dbeta, (const TT*)dres, (const TT*)dres2, HW, C, G, cb, pstride)
#define MDM_GN_FUSED_BWD(TT, ACT, LPR_)                                                                             \
    if (f.nthr == 512 && f.ni == 4) MDM_GN_FUSED_BWD_L(TT, ACT, 4, 512, LPR_);                                     \
    else if (f.nthr == 512) MDM_GN_FUSED_BWD_L(TT, ACT, 16, 512, LPR_);                                            \
    else MDM_GN_FUSED_BWD_L(TT, ACT, 4, 1024, LPR_)
    if (dtype == DT_F32) { if (act) { MDM_GN_FUSED_BWD(float, 1, 16); } else { MDM_GN_FUSED_BWD(float, 0, 16); } }
    else { if (act) { MDM_GN_FUSED_BWD(bf16, 1, 8); } else { MDM_GN_FUSED_BWD(bf16, 0, 8); } }
#undef MDM_GN_FUSED_BWD
#undef MDM_GN_FUSED_BWD_L
    MDM_LAUNCH_STATUS();
  }
  const int cb = gn_slice(C, G, epv);
  MDM_CHECK_ARG(cb > 0);
  // Two kernels read x and dy: the partial sums, then the apply.  When the pair is larger than the Infinity Cache keeps
  // (256 MB on MI355X; 268 MB at 64 x 64 x 256 channels, batch 64) the second read comes from HBM again, so the batch is
  // walked in chunks of samples whose x + dy fit: partial(chunk), apply(chunk), next chunk.
  const size_t esize = dtype == DT_F32 ? 4 : 2;
  int nchunk = 1;
  while (nchunk < 8 && N % (nchunk * 2) == 0 && N / (nchunk * 2) >= 4 &&
         2 * (size_t)N * HW * C * esize / nchunk > g_gn_chunk_bytes)
    nchunk *= 2;
  if ((size_t)(N / nchunk) * gn_slabs(N / nchunk, HW) > (size_t)N * gn_slabs(N, HW)) nchunk = 1;   // the workspace is sized for N
  const int nc = N / nchunk;
  const int slabs = gn_slabs(nc, HW);
  const int pps = (HW + slabs - 1) / slabs;
  const int sp = gn_pix_splits(nc, HW, C / cb);
  const int ppb = (HW + sp - 1) / sp;
  const dim3 agrid(sp, C / cb, nc);
#define MDM_GN_BWD(TT, ACT)                                                                                          \
  for (int ch = 0; ch < nchunk; ++ch) {                                                                              \
    const size_t n0 = (size_t)ch * nc, t0 = n0 * HW * C;                                                             \
    const TT* film_ = film ? (const TT*)film + n0 * 2 * C : nullptr;                                                 \
    TT* dfilm_ = dfilm ? (TT*)dfilm + n0 * 2 * C : nullptr;                                                          \
    const TT* dres_ = dres ? (const TT*)dres + t0 : nullptr;                                                         \
    const TT* dres2_ = dres2 ? (const TT*)dres2 + t0 : nullptr;                                                      \
    hipLaunchKernelGGL((gn_bwd_partial_kernel<TT, ACT>), dim3(nc * slabs), dim3(256), 0, st, (const TT*)dy + t0,     \
                       (const TT*)x + t0, coef + n0 * C * 2, ws, HW, C, slabs, pps);                                 \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<TT, ACT>), agrid, dim3(256), 0, st, (const TT*)dy + t0, (const TT*)x + t0, ws, \
                       stats + n0 * G * 2, coef + n0 * C * 2, gamma, beta, film_, dres_, dres2_, (TT*)dx + t0, dfilm_, \
                       dgamma + n0 * pstride, dbeta + n0 * pstride, HW, C, G, cb, slabs, ppb, pstride);              \
  }
  if (dtype == DT_F32) { if (act) { MDM_GN_BWD(float, 1) } else { MDM_GN_BWD(float, 0) } }
  else { if (act) { MDM_GN_BWD(bf16, 1) } else { MDM_GN_BWD(bf16, 0) } }
#undef MDM_GN_BWD
  MDM_LAUNCH_STATUS();
}

// ---- per-sample GroupNorm parameter-gradient rows -> gradient slots, many layers per launch ------------------------
// Entry l: dgamma_l[c] += sum_n pg_l[n][c], dbeta_l[c] += sum_n pb_l[n][c] (fixed order: deterministic).  One block per
// 256 channels of an entry; `first_block` = prefix sum of ceil(C / 256).
struct GnParamDesc {
  const float* pg; const float* pb;   // [N][C] each
  float* dgamma; float* dbeta;        // [C], accumulated into
  int N, C, first_block, pad_;
};

// A logical block of the table (256 channels) is FOUR launched blocks of 64 channels x 4 sample groups (round 6: one thread
// per channel walking all N samples one after the other was a 64-deep chain of dependent-latency loads on a grid of a few
// hundred blocks: 87 us per flush for 8 MB).
__global__ __launch_bounds__(256) void gn_param_reduce_multi_kernel(const GnParamDesc* __restrict__ table, int n) {
  __shared__ float red[2][4][64];
  const int lb = (int)blockIdx.x >> 2, sub = (int)blockIdx.x & 3;
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last entry whose first_block <= lb
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= lb) lo = mid; else hi = mid - 1;
  }
  const GnParamDesc d = table[lo];
  const int cl = (int)threadIdx.x & 63, q = (int)threadIdx.x >> 6;
  const int c = (lb - d.first_block) * 256 + sub * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < d.C) {
    int s_ = q;
    for (; s_ + 12 < d.N; s_ += 16) {   // four rows of each operand in flight
      const float a0 = d.pg[(size_t)s_ * d.C + c], a1 = d.pg[(size_t)(s_ + 4) * d.C + c];
      const float a2 = d.pg[(size_t)(s_ + 8) * d.C + c], a3 = d.pg[(size_t)(s_ + 12) * d.C + c];
      const float b0 = d.pb[(size_t)s_ * d.C + c], b1 = d.pb[(size_t)(s_ + 4) * d.C + c];
      const float b2 = d.pb[(size_t)(s_ + 8) * d.C + c], b3 = d.pb[(size_t)(s_ + 12) * d.C + c];
      a += (a0 + a1) + (a2 + a3);
      b += (b0 + b1) + (b2 + b3);
    }
    for (; s_ < d.N; s_ += 4) { a += d.pg[(size_t)s_ * d.C + c]; b += d.pb[(size_t)s_ * d.C + c]; }
  }
  red[0][q][cl] = a; red[1][q][cl] = b;
  __syncthreads();
  if (q == 0 && c < d.C) {
    d.dgamma[c] += (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    d.dbeta[c] += (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  }
}

// table: DEVICE array of n descriptors {const float* pg, pb; float* dgamma, dbeta; int N, C, first_block, pad} (48 bytes)
extern "C" int mdm_gn_param_reduce_multi(const void* table, int n, int total_blocks, void* stream) {
  MDM_CHECK_ARG(table && n > 0 && total_blocks > 0);
  static_assert(sizeof(GnParamDesc) == 48, "descriptor layout is part of the ABI");
  hipLaunchKernelGGL(gn_param_reduce_multi_kernel, dim3(4 * total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const GnParamDesc*)table, n);
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

// ---- one input, L LayerNorms ------------------------------------------------------------------------------------------
// gamma / beta / y / dy / dgamma / dbeta: HOST arrays of L (<= 32) device pointers.  stats [R][2] is shared.
extern "C" int mdm_ln_multi_fwd(const void* x, const float* const* gamma, const float* const* beta, void* const* y, int L,
                                float* stats, int R, int D, float eps, int dtype, void* stream) {
  MDM_CHECK_ARG(x && gamma && beta && y && stats && L >= 1 && L <= LN_MAXL && R > 0 && D > 0);
  MDM_CHECK_ARG(dtype == DT_F32 ? (D % 4 == 0 && D <= 256 * 4 * LN_NCK) : (D % 8 == 0 && D <= 256 * 8 * LN_NCK));   // whole 16-byte chunks
  LnGroup g = {};
  for (int l = 0; l < L; ++l) { MDM_CHECK_ARG(gamma[l] && beta[l] && y[l]); g.gamma[l] = gamma[l]; g.beta[l] = beta[l]; g.y[l] = y[l]; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32) hipLaunchKernelGGL(ln_multi_fwd_kernel<float>, dim3(R), dim3(256), 0, st, (const float*)x, g, L, stats, D, eps);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(ln_multi_fwd_kernel<bf16>, dim3(R), dim3(256), 0, st, (const bf16*)x, g, L, stats, D, eps);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

// dx = sum over the L norms of their input gradients; dgamma[l] / dbeta[l] (+)= the parameter gradients (`accumulate`
// == 0 zero-fills them first).  D <= 4096.
extern "C" int mdm_ln_multi_bwd(const void* const* dy, const void* x, const float* const* gamma, const float* stats,
                                void* dx, float* const* dgamma, float* const* dbeta, int L, int R, int D, int accumulate,
                                int dtype, void* stream) {
  MDM_CHECK_ARG(dy && x && gamma && stats && dx && dgamma && dbeta && L >= 1 && L <= LN_MAXL && R > 0 && D > 0 && D <= 4096);
  MDM_CHECK_ARG(D % (dtype == DT_F32 ? 4 : 8) == 0);
  LnGroup g = {};
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int l = 0; l < L; ++l) {
    MDM_CHECK_ARG(dy[l] && gamma[l] && dgamma[l] && dbeta[l]);
    g.y[l] = const_cast<void*>(dy[l]); g.gamma[l] = gamma[l]; g.dgamma[l] = dgamma[l]; g.dbeta[l] = dbeta[l];
    if (!accumulate) {
      (void)hipMemsetAsync(dgamma[l], 0, (size_t)D * sizeof(float), st);
      (void)hipMemsetAsync(dbeta[l], 0, (size_t)D * sizeof(float), st);
    }
  }
  const int rpb = 64, nsl = (R + rpb - 1) / rpb;
  if (dtype == DT_F32) {
    hipLaunchKernelGGL(ln_multi_bwd_kernel<float>, dim3(R), dim3(256), 0, st, g, L, (const float*)x, stats, (float*)dx, D);
    hipLaunchKernelGGL(ln_multi_param_kernel<float>, dim3((D + 255) / 256, nsl, L), dim3(256), 0, st, g, (const float*)x, stats, R, D, rpb);
  } else if (dtype == DT_BF16) {
    hipLaunchKernelGGL(ln_multi_bwd_kernel<bf16>, dim3(R), dim3(256), 0, st, g, L, (const bf16*)x, stats, (bf16*)dx, D);
    hipLaunchKernelGGL(ln_multi_param_kernel<bf16>, dim3((D + 255) / 256, nsl, L), dim3(256), 0, st, g, (const bf16*)x, stats, R, D, rpb);
  } else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}
