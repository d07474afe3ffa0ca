// Implicit-GEMM convolution / linear kernels for gfx950 (MFMA), NHWC.
//
// One kernel family serves every dense contraction of the U-Net hot path
// (reference: ml_mdm/models/unet.py:199-221 conv1/conv2/conv3, :260-271 qkv /
// proj_out / ffn 1x1 convs, :206,:605-609,:763 linears, :514-532 resample convs;
// nested_unet.py:109-128 adapters):
//
//   forward   y[m, n]  = sum_k  A(m, k) * Wf[n, k]            (+bias, +res, GELU)
//   dgrad     dx[m, c] = sum_k' A'(m, k') * Wd[c, k']          (same kernel, Wd = flipped/transposed pack)
//   wgrad     dW[n, k] = sum_m  dY[m, n] * A(m, k)             (split over m, fp32 slabs)
//
// where A(m, k) is the *virtual* im2col matrix of the NHWC activation: m = output
// pixel (n, oh, ow), k = tap * Cin + cin, tap = kh * 3 + kw.  It is never
// materialised: the loader computes the gather address per 16-byte chunk.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN, k-tile = 128 bytes per row
// (64 bf16 / 32 fp32), LDS rows XOR-swizzled (common.hpp), double-buffered,
// register-staged global->LDS with the next tile's loads in flight during the
// MFMA phase.  Operands are fed to MFMA swapped (D^T = W * A^T) so every lane
// ends up with 4 consecutive output channels of one pixel -> vector stores.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.hpp"
#include "conv_args.hpp"

namespace mdm {



// 4 consecutive elements (8 / 16 bytes, aligned) -> fp32
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
}
__device__ __forceinline__ void load4(const bf16* p, float (&o)[4]) {
  const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  o[0] = (float)t[0]; o[1] = (float)t[1]; o[2] = (float)t[2]; o[3] = (float)t[3];
}

// 16-byte-aligned zeros in device memory: the source of every out-of-range LDS-DMA chunk
__device__ __attribute__((aligned(16))) uint4 g_zero_page[4] = {};


// Epilogue shared by the GEMM kernels: bias / activation / residual in registers (fp32), then the finished tile is
// staged through LDS (free after the k-loop, LDS_BYTES of it) so that HBM sees whole 16-byte chunks of complete
// output rows instead of the 8-byte-per-lane fragments of the MFMA layout.
struct NoPrefetch { __device__ __forceinline__ void operator()() const {} };

// `after_lds` runs once every wave is done with the LDS (the staged tile is in registers by then): a persistent
// kernel issues the next tile's first LDS-DMA there, so those loads overlap this tile's stores.
//
// Vector-memory ordering is what this function is built around (round 6).  gfx950 has ONE counter (vmcnt) for loads AND
// stores and it retires in issue order: a wait for a load that was issued AFTER a store is a wait for the store's
// acknowledgement too.  Rounds 3-5 had such a load in every chunk of the store loop (the development knob read from a
// __device__ variable -- a vector load --, and the join of the residual paths), so the 16-32 stores of a thread went out one
// acknowledgement round trip at a time: the "additive store phase at ~4.6 TB/s" of HISTORY.md section 4.1 (rounds 3-5).  Now:
//   * every operand load (bias, residual / gelu' code) is issued before the staged tile is read back and is waited for,
//     explicitly, BEFORE the next tile's DMA is queued -- nothing in the store loop loads;
//   * built and NOT kept: letting a whole tile's stores drain under the next tile's first two k-tiles with counted waits
//     (their DMA is older than the stores) -- the bookkeeping cost 8-14 more spilled registers in the 256-wide kernels and
//     the train step nothing or worse (88.8 vs 86.4 ms, profiles/r06_did_not_pay.md); branch-free store loops per epilogue
//     kind: 82-129 spilled registers, half the speed (tools/mfma_hazard_scan.py's spill cap caught both before the GPU did).
template <typename T, int BM, int BN, int WM, int WN, int LDS_BYTES, typename AfterLds = NoPrefetch, bool GN = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem,
                                              int m0, int n0, AfterLds after_lds = AfterLds()) {
  constexpr int EPV = Tr<T>::EPV;
  constexpr int NT_ = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int quad = lane >> 4, l16 = lane & 15;
  // bias / activation / residual are applied in registers (fp32); the finished tile is then staged through
  // LDS (free after the k-loop) so that HBM sees whole 16-byte chunks of complete output rows instead of the
  // 8-byte-per-lane fragments of the MFMA layout (the output-heavy 1x1 convs -- qkv, FFN up -- were store-bound).
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  T* __restrict__ Ypre = reinterpret_cast<T*>(p.ypre);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ AUX = reinterpret_cast<const T*>(p.aux);
  // Staged rows are unpadded; the 16-byte chunk index is XOR-swizzled with the row instead, so the 16 rows a lane
  // group writes (row stride = a multiple of 256 B = the whole bank array) land on 16 different chunks.  (Unswizzled,
  // the 256x256 tile -- too large to pad -- wrote with 16-way bank conflicts and stored at 2 TB/s instead of 6-7.)
  // (a row length that is not a power of two -- the 192-wide tile -- is padded by one chunk instead)
  constexpr int OCH = BN / EPV;                                    // 16-byte chunks per staged row
  constexpr bool POW2 = (OCH & (OCH - 1)) == 0;
  constexpr int PITCH = BN * (int)sizeof(T) + (POW2 ? 0 : 16);
  constexpr int SWZ = POW2 ? (OCH >= 32 ? 32 : OCH) - 1 : 0;
  static_assert(BM * PITCH <= LDS_BYTES, "output tile must fit the k-loop LDS");
  // (x gelu'(aux) together with a residual would need two operands per chunk: the entry points reject that combination)
  if ((p.Cout % EPV) == 0) {
    // 1. accumulators + bias -> T -> LDS, branch-free (rows / columns past the edge carry garbage that is never stored)
    f32x4 bv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * TN + j * 16 + quad * 4;
      bv[j] = (p.bias && n < p.Cout) ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ml = wm * TM + i * 16 + l16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nl = wn * TN + j * 16 + quad * 4;
        const f32x4 v = acc[i][j] + bv[j];
        constexpr int CPL = 16 / (4 * (int)sizeof(T));   // lanes' 4-element groups per 16-byte chunk (2 bf16 / 1 fp32)
        const int cl = nl / EPV;
        char* dst = smem + ml * PITCH + ((cl ^ (ml & SWZ)) << 4) + ((nl / 4) % CPL) * 8;
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          *reinterpret_cast<bf16x4*>(dst) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        }
      }
    }
    // 1b. The operand of the fused elementwise tail (the pre-activation of `x gelu'(aux)`, else the residual) is requested
    //    NOW, all chunks of the thread at once, into the registers the accumulators just vacated: the loads fly during
    //    the barrier and the LDS read-back below and cost ONE memory round trip per tile.  Loaded inline in the store loop they
    //    cost one round trip per chunk (16 per tile): the x gelu'(aux) epilogue of the FFN-down input gradient ran at
    //    1.7 TB/s of aux reads (768 -> 3072 at batch 64: 146 us against 85 us without the aux operand).
    //    bf16 tensors: the operand of x gelu' is the one-byte code of gelu'(pre) (common.hpp DGeluCode) -- 8 bytes per chunk.
    constexpr int NCHP = BM * (BN / EPV) / NT_;
    constexpr bool AUX8 = sizeof(T) == 2 && kFfnAuxByte;
    const T* const PSRC = (p.act == 2 && AUX) ? AUX : R;
    uint4 pre[NCHP];
    if (AUX8 && p.act == 2 && AUX) {
      const unsigned char* const A8 = reinterpret_cast<const unsigned char*>(p.aux);
#pragma unroll
      for (int i = 0; i < NCHP; ++i) {
        const int idx = tid + i * NT_;
        const int row = idx / (BN / EPV), ch = idx - row * (BN / EPV);
        const int m = m0 + row, n = n0 + ch * EPV;
        pre[i] = uint4{0u, 0u, 0u, 0u};
        if (m < p.M && n < p.Cout) {
          const uint2 t = *reinterpret_cast<const uint2*>(A8 + (size_t)m * p.Cout + n);
          pre[i].x = t.x; pre[i].y = t.y;
        }
      }
    } else if (PSRC) {
#pragma unroll
      for (int i = 0; i < NCHP; ++i) {
        const int idx = tid + i * NT_;
        const int row = idx / (BN / EPV), ch = idx - row * (BN / EPV);
        const int m = m0 + row, n = n0 + ch * EPV;
        pre[i] = uint4{0u, 0u, 0u, 0u};
        if (m < p.M && n < p.Cout) {
          size_t o = (size_t)m * p.Cout + n;
          if (p.ps_cout > 0) {   // pixel-shuffled output: the residual has the OUTPUT's layout (same map as the store loop)
            const int ph = n / p.ps_cout, co = n - ph * p.ps_cout;
            const int bw = m % p.ps_W, t_ = m / p.ps_W;
            const int bh = t_ % p.ps_H, img = t_ / p.ps_H;
            o = (((size_t)img * (2 * p.ps_H) + 2 * bh + (ph >> 1)) * (2 * p.ps_W) + 2 * bw + (ph & 1)) * p.ps_cout + co;
          }
          pre[i] = *reinterpret_cast<const uint4*>(PSRC + o);
        }
      }
    }
    __syncthreads();
    // 2. whole 16-byte chunks of complete rows: (store the pre-activation) -> activation -> (+ residual) -> store.
    //    The activation sees the value already rounded to T -- what the reference's autocast graph does too
    //    (conv output in bf16, then GELU / add as separate bf16 ops); in fp32 mode nothing is rounded.
    const int act = p.act;
    constexpr int NCH = BM * OCH / NT_;   // staged chunks per thread
    static_assert(NCH * NT_ == BM * OCH, "staged tile must divide evenly over the block");
    uint4 raw[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * NT_;
      const int row = idx / OCH, ch = idx - row * OCH;
      raw[i] = *reinterpret_cast<const uint4*>(smem + row * PITCH + ((ch ^ (row & SWZ)) << 4));
    }
    __syncthreads();   // the LDS is free again
    // the operand chunks have had two barriers and the read-back to arrive; from here on the only vector-memory traffic of
    // this wave is the next tile's DMA (after_lds) and, younger than it, this tile's stores
    if (PSRC) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    after_lds();
    if constexpr (GN && BM == 256 && BN == 192 && sizeof(T) == 2) {
      {
        // ---- epilogue with the GroupNorm of the output (host-checked: whole tiles, no activation on y itself) ----
        constexpr int NG = 8, CPG = 24;                         // groups per column tile, channels per group
        float* const gsum = reinterpret_cast<float*>(smem + LDS_BYTES);   // scratch behind the k-loop stages:
        float* const gsq = gsum + NG * 16;                       //   [NG][16] partial sums, partial squares,
        float* const gmean = gsq + NG * 16;                      //   [NG] mean, [NG] rstd,
        float* const grstd = gmean + NG;                         //   [192][2] coefficients a, b
        float* const cab = grstd + NG;
        if (tid < 2 * NG * 16) gsum[tid] = 0.f;
        float part[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) part[g] = 0.f;
        // pass 1: finish y (residual), store it, keep the ROUNDED values (what a separate norm kernel would read)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int idx = tid + i * NT_;
          const int row = idx / OCH, ch = idx - row * OCH;
          const size_t o = (size_t)(m0 + row) * p.Cout + n0 + ch * EPV;
          Chunk<T> c;
          c.load(reinterpret_cast<const T*>(&raw[i]));
          if (R) {
            Chunk<T> rr;
            rr.load(reinterpret_cast<const T*>(&pre[i]));
#pragma unroll
            for (int e = 0; e < EPV; ++e) c.v[e] += rr.v[e];
            c.store(reinterpret_cast<T*>(&raw[i]));
            c.load(reinterpret_cast<const T*>(&raw[i]));
          }
          *reinterpret_cast<uint4*>(Y + o) = raw[i];
          float sv = 0.f;
#pragma unroll
          for (int e = 0; e < EPV; ++e) sv += c.v[e];
          const int g = ch / 3;
#pragma unroll
          for (int gg = 0; gg < NG; ++gg) part[gg] += g == gg ? sv : 0.f;
        }
        __syncthreads();   // scratch zeroed
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) atomicAdd(&gsum[gg * 16 + (lane & 15)], part[gg]);
        __syncthreads();
        if (tid < NG) {
          float t = 0.f;
          for (int k = 0; k < 16; ++k) t += gsum[tid * 16 + k];
          gmean[tid] = t / (float)(BM * CPG);
        }
        __syncthreads();
        // pass 2: centred squares
#pragma unroll
        for (int g = 0; g < NG; ++g) part[g] = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int ch = (tid + i * NT_) % OCH;
          Chunk<T> c;
          c.load(reinterpret_cast<const T*>(&raw[i]));
          const int g = ch / 3;
          const float mu = gmean[g];
          float sv = 0.f;
#pragma unroll
          for (int e = 0; e < EPV; ++e) { const float d = c.v[e] - mu; sv += d * d; }
#pragma unroll
          for (int gg = 0; gg < NG; ++gg) part[gg] += g == gg ? sv : 0.f;
        }
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) atomicAdd(&gsq[gg * 16 + (lane & 15)], part[gg]);
        __syncthreads();
        const int img = m0 / BM;                                // the sample this row tile is
        if (tid < NG) {
          float t = 0.f;
          for (int k = 0; k < 16; ++k) t += gsq[tid * 16 + k];
          const float rstd = rsqrtf(t / (float)(BM * CPG) + p.gn_eps);
          grstd[tid] = rstd;
          const int g = n0 / CPG + tid;
          p.gn_stats[((size_t)img * p.gn_groups + g) * 2] = gmean[tid];
          p.gn_stats[((size_t)img * p.gn_groups + g) * 2 + 1] = rstd;
        }
        __syncthreads();
        if (tid < BN) {
          const int cglob = n0 + tid, g = tid / CPG;
          const float ga = p.gn_gamma[cglob], be = p.gn_beta[cglob];
          const float a_ = ga * grstd[g], b_ = be - gmean[g] * grstd[g] * ga;
          cab[2 * tid] = a_; cab[2 * tid + 1] = b_;
          p.gn_coef[((size_t)img * p.Cout + cglob) * 2] = a_;
          p.gn_coef[((size_t)img * p.Cout + cglob) * 2 + 1] = b_;
        }
        __syncthreads();
        // pass 3: the normalised output
        T* __restrict__ Y2 = reinterpret_cast<T*>(p.gn_y);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int idx = tid + i * NT_;
          const int row = idx / OCH, ch = idx - row * OCH;
          const size_t o = (size_t)(m0 + row) * p.Cout + n0 + ch * EPV;
          Chunk<T> c;
          c.load(reinterpret_cast<const T*>(&raw[i]));
#pragma unroll
          for (int e = 0; e < EPV; ++e) {
            const float z = cab[2 * (ch * EPV + e)] * c.v[e] + cab[2 * (ch * EPV + e) + 1];
            c.v[e] = p.gn_act ? silu_f(z) : z;
          }
          c.store(Y2 + o);
        }
        __syncthreads();   // the scratch may be zeroed again by the next tile's epilogue
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * NT_;
      const int row = idx / OCH, ch = idx - row * OCH;
      const int m = m0 + row, n = n0 + ch * EPV;
      if (m >= p.M || n >= p.Cout) continue;
      if (p.dev_flags & 1) continue;   // development knob 1 (a kernel ARGUMENT: see conv_args.hpp)
      size_t o = (size_t)m * p.Cout + n;
      if (p.ps_cout > 0) {   // (phase, co) column of low-res pixel (n, bh, bw) -> its place in the 2x larger image
        const int ph = n / p.ps_cout, co = n - ph * p.ps_cout;
        const int bw = m % p.ps_W, t_ = m / p.ps_W;
        const int bh = t_ % p.ps_H, img = t_ / p.ps_H;
        o = (((size_t)img * (2 * p.ps_H) + 2 * bh + (ph >> 1)) * (2 * p.ps_W) + 2 * bw + (ph & 1)) * p.ps_cout + co;
      }
      if (act == 0 && !R) {
        *reinterpret_cast<uint4*>(Y + o) = raw[i];
        continue;
      }
      Chunk<T> c;
      c.load(reinterpret_cast<const T*>(&raw[i]));
      if (act == 1) {
        if (Ypre) {
          if constexpr (AUX8) {   // what the backward needs of the pre-activation: gelu'(pre), one byte per element
            const uint2 code = {DGeluCode::enc4(&c.v[0]), DGeluCode::enc4(&c.v[4])};
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.ypre) + o) = code;
          } else {
            *reinterpret_cast<uint4*>(Ypre + o) = raw[i];
          }
        }
        gelu_vec<T, EPV>(c.v);
      } else if (act == 2) {
        if constexpr (AUX8) {
          float g[8];
          DGeluCode::dec4(pre[i].x, g);
          DGeluCode::dec4(pre[i].y, g + 4);
#pragma unroll
          for (int e = 0; e < EPV; ++e) c.v[e] *= g[e];
        } else {
          Chunk<T> ax;
          ax.load(reinterpret_cast<const T*>(&pre[i]));
          mul_dgelu_vec<T, EPV>(c.v, ax.v);
        }
      }
      if (R) {
        Chunk<T> rr;
        rr.load(reinterpret_cast<const T*>(&pre[i]));
#pragma unroll
        for (int e = 0; e < EPV; ++e) c.v[e] += rr.v[e];
      }
      c.store(Y + o);
    }
    return;
  }
  __syncthreads();
  after_lds();
  // generic path (Cout not a multiple of the chunk): per-element stores
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * TM + i * 16 + l16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * TN + j * 16 + quad * 4;
      if (n >= p.Cout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const size_t o = (size_t)m * p.Cout + n;
      const int nv = min(4, p.Cout - n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e >= nv) continue;
        if (p.bias) v[e] += p.bias[n + e];
        if (p.act == 1) {
          // (this path sees the unrounded accumulator; the byte code of gelu' is taken from the value rounded to T, as the
          //  chunked path above takes it)
          if (Ypre) {
            if constexpr (sizeof(T) == 2 && kFfnAuxByte) reinterpret_cast<unsigned char*>(p.ypre)[o + e] = (unsigned char)DGeluCode::enc(to_f32(from_f32<T>(v[e])));
            else Ypre[o + e] = from_f32<T>(v[e]);
          }
          v[e] = gelu_t<T>(v[e]);
        } else if (p.act == 2) {
          if constexpr (sizeof(T) == 2 && kFfnAuxByte) v[e] *= DGeluCode::dec(reinterpret_cast<const unsigned char*>(p.aux)[o + e]);
          else v[e] *= dgelu_t<T>(to_f32(AUX[o + e]));
        }
        if (R) v[e] += to_f32(R[o + e]);
        Y[o + e] = from_f32<T>(v[e]);
      }
    }
  }
}

// B operand of the bf16x3 form when the weights were split into planes ahead of time (mdm_split_weight_planes: every run of
// 8 fp32 values = 32 bytes holds [8 bf16 hi | 8 bf16 lo]): the two 16-byte chunks a lane reads ARE its hi and lo fragments
__device__ __forceinline__ void load_frag_planes(FragSplit& f, const char* t, int row, int quad) {
  f.hi = *reinterpret_cast<const bf16x8*>(t + lds_chunk_off(row, 2 * quad));
  f.lo = *reinterpret_cast<const bf16x8*>(t + lds_chunk_off(row, 2 * quad + 1));
}
template <typename F> __device__ __forceinline__ void load_frag_planes(F&, const char*, int, int) {}

// Double-buffered: the next tile's LDS-DMA overlaps this tile's MFMAs; __syncthreads() drains it (vmcnt(0)).
// SPLIT (float only): 1 = the products as three bf16 MFMAs on hi / lo halves of the fp32 operands (common.hpp FragSplit),
// both operands split in the loop; 2 = the same with the weight operand already stored as hi / lo planes (half the
// conversion work of the loop: 97.8 -> 88.9 ms per iteration of the 64x64 sampler at batch 64, eager)
template <typename T, int BM, int BN, int WM, int WN, int MODE, int SPLIT = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_kernel(ConvArgs p) {
  constexpr int NSTAGE = 2;
  constexpr int EPV = Tr<T>::EPV, BK = Tr<T>::BK, KSTEPS = Tr<T>::KSTEPS;
  constexpr int NT_ = WM * WN * 64;                  // threads per block
  constexpr int RPP = NT_ / 8;                       // tile rows staged per pass (8 chunk lanes per row)
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int AJ = BM / RPP, BJ = BN / RPP;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int quad = lane >> 4, l16 = lane & 15;

  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.w);

  // ---- loader state -------------------------------------------------------
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ (lrow & 7);   // logical chunk this thread fetches (physical slot = tid & 7)
  int a_pix[AJ];   // n * H * W  (pixel index base of the image), -1 if row >= M
  int a_oh[AJ], a_ow[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + lrow + RPP * j;
    if (m < p.M) {
      if (MODE == MODE_1x1) {
        a_pix[j] = m; a_oh[j] = 0; a_ow[j] = 0;
      } else {
        const int hw = p.Ho * p.Wo;
        const int n = m / hw, r = m - n * hw;
        const int oh = r / p.Wo;
        a_pix[j] = n * p.H * p.W; a_oh[j] = oh; a_ow[j] = r - oh * p.Wo;
      }
    } else {
      a_pix[j] = -1; a_oh[j] = 0; a_ow[j] = 0;
    }
  }
  int kcur = lchunk * EPV;  // this thread's k within the current k-tile (global k)
  int tap = 0, cin = kcur;
  // channel-block-major order (kblk): one k-tile = one (64-channel block, tap); the 9 taps of a channel block are
  // 9 consecutive k-tiles, so the shifted re-reads of the same activation lines hit L1/L2 instead of coming back
  // from the Infinity Cache a whole Cin-sweep later (tap-major order re-reads each line after Cin/64 k-tiles)
  const bool kblk = MODE != MODE_1x1 && p.kblk != 0;
  if (MODE != MODE_1x1 && !kblk) { tap = kcur / p.Cin; cin = kcur - tap * p.Cin; }

  // Global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.  The DMA writes
  // lane-linear (wave base + lane * 16), i.e. thread (row = tid >> 3, slot = tid & 7) of pass j fills physical
  // slot `slot` of row `row + 32 j`; the XOR swizzle therefore moves to the SOURCE side: the thread fetches the
  // logical chunk lchunk = slot ^ (row & 7).  Out-of-range taps / rows / k read a 16-byte zero page.
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
#define MDM_GLDS(src, lds_off)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                  \
                                   (__attribute__((address_space(3))) void*)(lds_off), 16, 0, 0)
#define MDM_STAGE_TILE(stage)                                                                             \
  {                                                                                                       \
    const bool kvalid = kcur < p.K;                                                                       \
    int kh = 0, kw = 0;                                                                                   \
    if (MODE != MODE_1x1) { kh = (tap * 11) >> 5; kw = tap - 3 * kh; }                                    \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                      \
      bool v = kvalid && a_pix[j] >= 0;                                                                   \
      size_t off;                                                                                         \
      if (MODE == MODE_1x1) {                                                                             \
        off = (size_t)a_pix[j] * p.Cin + kcur;                                                            \
      } else if (MODE == MODE_3x3) {                                                                      \
        const int ih = a_oh[j] * p.stride + kh - 1, iw = a_ow[j] * p.stride + kw - 1;                     \
        v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);                        \
        off = (size_t)(a_pix[j] + ih * p.W + iw) * p.Cin + cin;                                           \
      } else {                                                                                            \
        const int th = a_oh[j] + kh - 1, tw = a_ow[j] + kw - 1;                                           \
        const int ih = th >> 1, iw = tw >> 1;                                                             \
        v = v && th >= 0 && tw >= 0 && !(th & 1) && !(tw & 1) && ih < p.H && iw < p.W;                    \
        off = (size_t)(a_pix[j] + ih * p.W + iw) * p.Cin + cin;                                           \
      }                                                                                                   \
      const T* src = v ? X + off : reinterpret_cast<const T*>(g_zero_page);                               \
      MDM_GLDS(src, (stage) + j * (RPP * 128) + wave_lds);                                                \
    }                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                                      \
      const int n = n0 + lrow + RPP * j;                                                                  \
      const bool v = kvalid && n < p.Cout;                                                                \
      const T* src = v ? Wp + (size_t)n * p.K + kcur : reinterpret_cast<const T*>(g_zero_page);           \
      MDM_GLDS(src, (stage) + A_BYTES + j * (RPP * 128) + wave_lds);                                      \
    }                                                                                                     \
    kcur += BK;                                                                                           \
    if (MODE != MODE_1x1) {                                                                               \
      if (kblk) {                                                                                         \
        if (++tap == 9) { tap = 0; cin += BK; }                                                           \
      } else {                                                                                            \
        cin += BK;                                                                                        \
        while (cin >= p.Cin) { cin -= p.Cin; ++tap; }                                                     \
      }                                                                                                   \
    }                                                                                                     \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (p.K + BK - 1) / BK;
#define MDM_COMPUTE_TILE(cur)                                                                             \
  {                                                                                                       \
    const char* As = (cur);                                                                               \
    const char* Bs = (cur) + A_BYTES;                                                                     \
    _Pragma("unroll") for (int ks = 0; ks < KSTEPS; ++ks) {                                               \
      typename FragOf<T, SPLIT != 0>::type af[MT], bfr[NT];                                               \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) load_frag_x(af[i], As, wm * TM + i * 16 + l16, ks, quad);    \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                    \
        if constexpr (SPLIT == 2) load_frag_planes(bfr[j], Bs, wn * TN + j * 16 + l16, quad);             \
        else load_frag_x(bfr[j], Bs, wn * TN + j * 16 + l16, ks, quad);                                   \
      }                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                      \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) mma16(acc[i][j], bfr[j], af[i]);                   \
    }                                                                                                     \
  }
  MDM_STAGE_TILE(smem);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    char* cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < ntiles) MDM_STAGE_TILE(smem + ((kt + 1) & 1) * STAGE);
    MDM_COMPUTE_TILE(cur);
    __syncthreads();   // drains the LDS-DMA of tile kt+1 (vmcnt(0)) and fences the reads of tile kt
  }
#undef MDM_COMPUTE_TILE
#undef MDM_STAGE_TILE
#undef MDM_GLDS

  conv_epilogue<T, BM, BN, WM, WN, NSTAGE * STAGE>(p, acc, smem, m0, n0);
}

// ---------------------------------------------------------------------------
// bf16 main-path kernel: same tiling / LDS image / epilogue as conv_gemm_kernel, different k-loop.
//
// In conv_gemm_kernel every k-tile opens with a block of ~70 VALU + 8 LDS-DMA issues per lane that computes the
// implicit-GEMM gather addresses (64-bit, halo tests, zero-page select) -- with all eight waves of the block in
// lock-step behind the per-tile barrier the matrix pipe idles through it (measured: k-loop without the staging
// block 1.42 PF, staging alone 1.77 PF-equivalent, both 0.97 PF).  Here the gather is expressed as a buffer
// address: a per-lane 32-bit byte offset fixed for the whole k-loop (row of the virtual im2col matrix, swizzled
// 16-byte chunk) plus a wave-uniform SGPR offset per k-tile (tap shift + channel block), so a k-tile costs 8
// `buffer_load_dwordx4 ... lds` whose only vector work is the halo select (a 9-bit tap mask per row, 3x3 only).
// Rows / taps outside the image use an offset beyond num_records: the buffer range check returns zeros without a
// memory access.  The 8 DMA issues and the second k-step's fragment reads are interleaved with the first k-step's
// MFMAs (sched_group_barrier), so the load phase runs in the shadow of the matrix pipe.
// Requirements (host-checked, else conv_gemm_kernel): bf16, 1x1 or channel-block-major 3x3 (s1 / s2),
// K % 64 == 0, operands < 0x7F000000 bytes.
// ---------------------------------------------------------------------------
// issue pattern of a phase: per MFMA row, NT MFMAs, DMA LDS-DMA issues, one fragment read (literal arguments only)
template <int NT, int DMA, int... I>
__device__ __forceinline__ void sched_rows(std::integer_sequence<int, I...>) {
  ((__builtin_amdgcn_sched_group_barrier(0x008, NT, I * 0), __builtin_amdgcn_sched_group_barrier(0x010, DMA, 0),
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)), ...);
}

// Grouped launch (GROUPED = true, 1x1 only): G problems of one shape (M, K, Cout) -- the text-state key / value
// projections of every attention layer, unet.py:264 -- in ONE launch: tile t belongs to problem t / tiles_per_problem and
// takes its operand / output pointers from the group table.  A single such problem (M = B * 32 text tokens) is 96-192
// tiles of 128x128: a quarter of the chip, 28 us each for 31 layers.
constexpr int CG_MAXG = 32;
struct ConvGroup {
  const void* x[CG_MAXG];
  const void* w[CG_MAXG];
  const float* bias[CG_MAXG];
  void* y[CG_MAXG];
};
struct NoConvGroup {};

// SEL4 (3x3 only): the reduction walks only the four taps (kh, kw) in {1, 2}^2 -- pixel offsets {0, +1} -- of every
// channel block: k-tile kt = (channel block kt >> 2, tap 4 + (kt & 1) + 3 * ((kt >> 1) & 1)), K = 4 * Cin.  That is the
// input gradient of a STRIDE-2 3x3 convolution in its pixel-unshuffled form (mdm_conv_s2_dgrad): dx[2b + p] only draws
// from dy[b] and dy[b + 1], so over the 2x2-blocked dx (4 Cin channels per block) it is a 2x2 stride-1 correlation.
// GN: the epilogue also normalises the output (ConvArgs::gn_*; its own instantiation, so the extra registers of that
// epilogue do not touch the other kernels' allocation).
// NSTG: LDS stages of the k-loop (2 in every training instantiation; 4 for the under-filled grids of small-batch sampling,
// one block per CU: there a k-tile is 0.25 us of MFMA work behind a ~1.3 us DMA round trip, and with two stages every
// k-tile waits for its own fetch -- with four, three fetches are in flight and the loop runs at the DMA's throughput).
template <int BM, int BN, int WM, int WN, int MODE, bool GROUPED = false, bool SPLITK = false, bool SEL4 = false, bool GN = false, int NSTG = 2>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_gemm_bl_kernel(ConvArgs p, std::conditional_t<GROUPED, ConvGroup, NoConvGroup> gr) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type has no host-side counterpart: the host pass only needs the stub
  using T = bf16;
  static_assert(!GROUPED || MODE == MODE_1x1, "grouped launches: 1x1 / linear only");
  static_assert(!(GROUPED && SPLITK), "split-K launches are single problems");
  static_assert(!SEL4 || (MODE == MODE_3x3 && !GROUPED && !SPLITK), "tap selection: plain 3x3 launches only");
  constexpr int NT_ = WM * WN * 64;
  constexpr int RPP = NT_ / 8;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int AJ = BM / RPP, BJ = BN / RPP;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned INVALID = 0x7F000000u;
  static_assert(MODE == MODE_1x1 || MODE == MODE_3x3, "buffer-addressed loader: 1x1 and 3x3 only");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int quad = lane >> 4, l16 = lane & 15;

  // Persistent blocks: block b works through output tiles b, b + G, b + 2G, ... (G = gridDim.x <= resident blocks);
  // within each group of G the XCD-aware remap keeps neighbouring tiles (shared operand panels) on one XCD's L2.
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tiles_per_problem = ((p.M + BM - 1) / BM) * tiles_n;
  const int tiles_total = tiles_per_problem * (GROUPED ? p.groups : 1) * (SPLITK ? p.ksplit : 1);
  const int G = gridDim.x;
  const int b_in_group = xcd_remap(blockIdx.x, G);

  // ---- per-lane gather offsets (bytes) of one output tile, fixed for its whole k-loop -------------
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ (lrow & 7);   // logical chunk this lane fetches (physical slot = tid & 7)
  const unsigned bias = MODE == MODE_3x3 ? (unsigned)(p.W + 1) * p.Cin * 2u : 0u;   // base shift that keeps the per-tap scalar offset non-negative
  unsigned a_voff[AJ], a_mask[AJ], b_voff[BJ];
  int m0 = 0, n0 = 0, grp = 0;
  int sel_base_tile = p.sel_base;                     // SEL4: tap base of the current output tile
  const int kpp_ = (p.sel_cout >> 6) * 4;             // SEL4, sel_mode 2: k-tiles per input phase
  int split = 0, kt0 = 0, nloc = p.K / 64;   // SPLITK: this tile's reduction range = k-tiles [kt0, kt0 + nloc)
  char* a_base = const_cast<char*>(reinterpret_cast<const char*>(p.x)) - bias;
  char* b_base = const_cast<char*>(reinterpret_cast<const char*>(p.w));
#define MDM_TILE_SETUP(tile_)                                                                               \
  {                                                                                                         \
    int tl_ = (tile_);                                                                                      \
    if constexpr (GROUPED) {                                                                                \
      grp = tl_ / tiles_per_problem; tl_ -= grp * tiles_per_problem;                                        \
      a_base = const_cast<char*>(reinterpret_cast<const char*>(gr.x[grp]));                                 \
      b_base = const_cast<char*>(reinterpret_cast<const char*>(gr.w[grp]));                                 \
    }                                                                                                       \
    if constexpr (SPLITK) {                                                                                 \
      split = tl_ / tiles_per_problem; tl_ -= split * tiles_per_problem;                                    \
      kt0 = split * p.kt_per; nloc = min(p.kt_per, p.K / 64 - kt0);                                         \
    }                                                                                                       \
    m0 = (tl_ / tiles_n) * BM; n0 = (tl_ % tiles_n) * BN;                                                   \
    if constexpr (SEL4) {                                                                                   \
      if (p.sel_mode == 1) { const int ph_ = n0 / p.sel_cout; sel_base_tile = (ph_ >> 1) * 3 + (ph_ & 1); } \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                                        \
      const int m = m0 + lrow + RPP * j;                                                                    \
      a_voff[j] = INVALID; a_mask[j] = 0u;                                                                  \
      if (m < p.M) {                                                                                        \
        if (MODE == MODE_1x1) {                                                                             \
          a_voff[j] = (unsigned)m * p.Cin * 2u + lchunk * 16u;                                              \
          a_mask[j] = 0x1ffu;                                                                               \
        } else {                                                                                            \
          const int hw = p.Ho * p.Wo;                                                                       \
          const int n = m / hw, r = m - n * hw;                                                             \
          const int oh = r / p.Wo, ow = r - oh * p.Wo;                                                      \
          const int ih0 = oh * p.stride, iw0 = ow * p.stride;                                               \
          a_voff[j] = (unsigned)(n * p.H * p.W + ih0 * p.W + iw0) * p.Cin * 2u + lchunk * 16u;              \
          unsigned mk = 0u;                                                                                 \
          _Pragma("unroll") for (int tp = 0; tp < 9; ++tp) {                                                \
            const int ih = ih0 + tp / 3 - 1, iw = iw0 + tp % 3 - 1;                                         \
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mk |= 1u << tp;               \
          }                                                                                                 \
          a_mask[j] = mk;                                                                                   \
        }                                                                                                   \
      }                                                                                                     \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                                        \
      const int n = n0 + lrow + RPP * j;                                                                    \
      b_voff[j] = n < p.Cout ? (unsigned)n * p.K * 2u + lchunk * 16u : INVALID;                             \
    }                                                                                                       \
    if (p.dev_flags & 2) {  /* development knob 0: the LDS-DMA fetches nothing (what the k-loop costs without memory) */ \
      _Pragma("unroll") for (int j = 0; j < AJ; ++j) a_voff[j] = INVALID;                                   \
      _Pragma("unroll") for (int j = 0; j < BJ; ++j) b_voff[j] = INVALID;                                   \
    }                                                                                                       \
  }
  const unsigned a_bytes = (unsigned)p.N * p.H * p.W * p.Cin * 2u + bias;
  const unsigned b_bytes = (unsigned)p.Cout * p.K * 2u;
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
  const int ntiles = p.K / 64;

#define MDM_BLDS(rs, lds_off, voff, soff)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_off), 16, voff, soff, 0, 0)
  // Per-k-tile wave-uniform state of the loader: buffer descriptors (empty past the last tile, so the same loads
  // fetch nothing), the A-side scalar offset (tap shift + channel block) and the tap's bit in the halo masks.
#define MDM_TILE_STATE(kt_)                                                                                 \
  const bool more_ = (kt_) < (SPLITK ? nloc : ntiles);                                                      \
  const int kta_ = SPLITK ? kt0 + (kt_) : (kt_);                                                            \
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(a_base, 0, more_ ? a_bytes : 0u, 0x00020000);         \
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(b_base, 0, more_ ? b_bytes : 0u, 0x00020000);         \
  const int b_soff = kta_ * 128;                                                                            \
  int a_soff = b_soff, tapbit = 1;                                                                          \
  if (MODE == MODE_3x3) {                                                                                   \
    int cb, tap;                                                                                            \
    if constexpr (SEL4) {                                                                                   \
      int r_ = kta_, base_ = sel_base_tile, coff_ = 0;                                                      \
      if (p.sel_mode == 2) {   /* phase of this k-tile: kpp_ k-tiles per phase */                           \
        const int ph_ = (kta_ >= kpp_) + (kta_ >= 2 * kpp_) + (kta_ >= 3 * kpp_);                           \
        r_ = kta_ - ph_ * kpp_;                                                                             \
        base_ = (1 - (ph_ >> 1)) * 3 + (1 - (ph_ & 1));                                                     \
        coff_ = ph_ * (p.sel_cout >> 6);                                                                    \
      }                                                                                                     \
      cb = coff_ + (r_ >> 2);                                                                               \
      tap = base_ + (r_ & 1) + 3 * ((r_ >> 1) & 1);                                                         \
    } else {                                                                                                \
      cb = kta_ / 9; tap = kta_ - 9 * cb;                                                                   \
    }                                                                                                       \
    const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;                                                      \
    a_soff = (kh * p.W + kw) * p.Cin * 2 + cb * 128;                                                        \
    tapbit = 1 << tap;                                                                                      \
  }
  // piece q of a k-tile's AJ + BJ LDS-DMA issues (1 KiB per wave each)
#define MDM_DMA_PIECE(stage, q)                                                                             \
  if ((q) < AJ) {                                                                                           \
    const unsigned vo = (MODE == MODE_1x1 || (a_mask[(q)] & tapbit)) ? a_voff[(q)] : INVALID;               \
    MDM_BLDS(rsA, (stage) + (q) * (RPP * 128) + wave_lds, vo, a_soff);                                      \
  } else if ((q) < AJ + BJ) {                                                                               \
    MDM_BLDS(rsB, (stage) + A_BYTES + ((q) - AJ) * (RPP * 128) + wave_lds, b_voff[(q) - AJ], b_soff);       \
  }

  f32x4 acc[MT][NT];

  // Pipeline (one barrier per k-tile, in the middle of it):
  //   phase A(kt): MFMAs of k-step 0 of tile kt; under them the fragment reads of k-step 1 (row i's A fragment is
  //                re-loaded right after row i's MFMAs, B fragments double-buffered)
  //   barrier    : every wave has issued all its LDS reads of tile kt; the DMA of tile kt+1 has landed
  //   phase B(kt): MFMAs of k-step 1; under them the DMA of tile kt+2 into tile kt's buffer and the fragment
  //                reads of k-step 0 of tile kt+1
  // so neither the fragment reads nor the DMA issue ever run with the matrix pipe idle.
  // Across output tiles: the first two k-tiles of the NEXT output tile are requested from inside the epilogue, as
  // soon as the staged tile has left the LDS, so they land while this tile's stores drain.
#define MDM_TILE_PROLOGUE()                                                                                 \
  {                                                                                                         \
    MDM_TILE_STATE(0);                                                                                      \
    _Pragma("unroll") for (int q = 0; q < AJ + BJ; ++q) { MDM_DMA_PIECE(smem, q); }                         \
  }                                                                                                         \
  {                                                                                                         \
    MDM_TILE_STATE(1);                                                                                      \
    _Pragma("unroll") for (int q = 0; q < AJ + BJ; ++q) { MDM_DMA_PIECE(smem + STAGE, q); }                 \
  }                                                                                                         \
  if constexpr (NSTG > 2) {                                                                                 \
    {                                                                                                       \
      MDM_TILE_STATE(2);                                                                                    \
      _Pragma("unroll") for (int q = 0; q < AJ + BJ; ++q) { MDM_DMA_PIECE(smem + 2 * STAGE, q); }           \
    }                                                                                                       \
    {                                                                                                       \
      MDM_TILE_STATE(3);                                                                                    \
      _Pragma("unroll") for (int q = 0; q < AJ + BJ; ++q) { MDM_DMA_PIECE(smem + 3 * STAGE, q); }           \
    }                                                                                                       \
  }

  int tile = b_in_group;
  if (tile >= tiles_total) return;
  MDM_TILE_SETUP(tile);
  MDM_TILE_PROLOGUE();
  for (;;) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this tile's first k-tile (and the previous tile's stores, which are younger: one in-order counter); with NSTG > 2 the
    // later prologue fetches stay in flight: every k-tile is AJ + BJ LDS-DMA instructions per wave
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (AJ + BJ) + (NSTG > 2 ? AJ + BJ : 0)) : "memory");
    __syncthreads();
    Frag<T> af[MT], b0[NT], b1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) load_frag<T>(b0[j], smem + A_BYTES, wn * TN + j * 16 + l16, 0, quad);
#pragma unroll
    for (int i = 0; i < MT; ++i) load_frag<T>(af[i], smem, wm * TM + i * 16 + l16, 0, quad);
    constexpr int DMA_PER_ROW = (AJ + BJ + MT - 1) / MT;
    const int kt_end = SPLITK ? nloc : ntiles;
    for (int kt = 0; kt < kt_end; ++kt) {
      const char* As = smem + (kt % NSTG) * STAGE;
      const char* Bs = As + A_BYTES;
      // ---- phase A
#pragma unroll
      for (int j = 0; j < NT; ++j) load_frag<T>(b1[j], Bs, wn * TN + j * 16 + l16, 1, quad);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) mma16(acc[i][j], b0[j], af[i]);
        load_frag<T>(af[i], As, wm * TM + i * 16 + l16, 1, quad);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
      sched_rows<NT, 0>(std::make_integer_sequence<int, MT>{});
      __builtin_amdgcn_sched_barrier(0);
      // hipcc does not count the loop-carried LDS-DMA of the previous phase B at this barrier: retire it explicitly
      // (tile kt + 1 must have landed; tiles kt + 2 .. kt + NSTG - 1 may still be in flight)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * (AJ + BJ)) : "memory");
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B
      const char* An = smem + ((kt + 1) % NSTG) * STAGE;
      const char* Bn = An + A_BYTES;
      char* dst = smem + (kt % NSTG) * STAGE;
      MDM_TILE_STATE(kt + NSTG);
#pragma unroll
      for (int j = 0; j < NT; ++j) load_frag<T>(b0[j], Bn, wn * TN + j * 16 + l16, 0, quad);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) mma16(acc[i][j], b1[j], af[i]);
#pragma unroll
        for (int d = 0; d < DMA_PER_ROW; ++d) { MDM_DMA_PIECE(dst, i * DMA_PER_ROW + d); }
        load_frag<T>(af[i], An, wm * TM + i * 16 + l16, 0, quad);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
      sched_rows<NT, DMA_PER_ROW>(std::make_integer_sequence<int, MT>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (empty) LDS-DMA issued past the last k-tile
    __syncthreads();   // LDS is reused by the epilogue
    const int cur_m0 = m0, cur_n0 = n0;
    const int next = tile + G;
    if constexpr (SPLITK) {
      // raw fp32 tile of this reduction range -> part[split]; the LDS is not needed, so the next tile's DMA starts first
      float* __restrict__ P = p.part + (size_t)split * p.M * p.Cout;
      if (next < tiles_total) {
        MDM_TILE_SETUP(next);
        MDM_TILE_PROLOGUE();
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int m = cur_m0 + wm * TM + i * 16 + l16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = cur_n0 + wn * TN + j * 16 + quad * 4;
          if (n < p.Cout) *reinterpret_cast<f32x4*>(P + (size_t)m * p.Cout + n) = acc[i][j];   // Cout % 8 == 0 (host-checked)
        }
      }
    } else {
      ConvArgs pe = p;   // the epilogue's view of this tile's problem (the prefetch below already moves on to the next)
      if constexpr (GROUPED) { pe.bias = gr.bias[grp]; pe.y = gr.y[grp]; }
      auto prefetch_next = [&]() {
        if (next < tiles_total) {
          MDM_TILE_SETUP(next);
          MDM_TILE_PROLOGUE();
        }
      };
      conv_epilogue<T, BM, BN, WM, WN, NSTG * STAGE, decltype(prefetch_next), GN>(pe, acc, smem, cur_m0, cur_n0, prefetch_next);
    }
    if (next >= tiles_total) break;
    tile = next;
  }
#undef MDM_TILE_PROLOGUE
#undef MDM_TILE_SETUP
#undef MDM_DMA_PIECE
#undef MDM_TILE_STATE
#undef MDM_BLDS
#endif
}

// ---------------------------------------------------------------------------
// wgrad: slab[s][n][k] = sum_{m in split s} dY[m, n] * A(m, k)   (fp32)
// Both operands are pixel-major in HBM (reduction index m is the slow one), so
// the loader transposes EPV x EPV register blocks before writing the same
// swizzled [row][reduction] LDS image the forward kernel uses.
// ---------------------------------------------------------------------------
struct WgradArgs {
  const void* x;    // [N, H, W, Cin]
  const void* dy;   // [M, Cout]
  float* slab;      // [splits][Cout][K]
  float* bslab;     // [splits * share][Cout] column sums of dY (bias gradient partials) or null; the int at bslab[-WG_BHDR]
                    // receives `share` (rows per split) from the kernel that fills the rows -- the reduce kernels read it
  int N, H, W, Cin, Ho, Wo, Cout, stride;
  int M, K;
  int splits, mtiles_per_split;
  int groups;       // conv_wgrad_bl_kernel only: > 0 = grouped launch over `groups` same-shape problems (WgradGroup)
  int bias_share;   // conv_wgrad_bl_kernel: the column sums are spread over this many k-tile blocks of every (split, n-tile)
  int skip_cout;    // conv_wgrad_bl_kernel, 3x3 only: > 0 = dY is a 2x2-blocked gradient with skip_cout channels per phase
                    // (sub-pixel form of upsample2x -> conv3x3): output tile (phase, tap) is computed only for the four
                    // taps {ph, ph+1} x {pw, pw+1} its phase reads; the other tiles of the slab stay unwritten
};

// Grouped weight gradient (1x1, bf16): G same-shape layers in ONE launch.  The 26 attention layers of the 16x16 level
// each have four 1x1 convolutions whose dW is 9-36 tiles of 256x256: alone, such a problem fills the 256 CUs only by
// splitting the pixel reduction 7-28 ways into fp32 slabs that a second kernel has to add up again (a third of the
// weight-gradient time of a step).  26 of them together are 234-936 tiles: no split, no slabs, no reduce kernel, a
// 256-k-tile loop per block, and the result is added straight into the parameters' gradient-arena slots.
constexpr int WG_MAXG = 32;
constexpr int WG_BHDR = 64;      // floats ahead of the bias-gradient partials: [0] = rows per split, as an int
constexpr int WG_BSHARE = 4;     // most k-tile blocks of one (split, n-tile) the column sums are spread over
struct WgradGroup {
  const void* x[WG_MAXG];
  const void* dy[WG_MAXG];
  float* out[WG_MAXG];    // dW destination (reference layout == packed layout for 1x1), accumulated into
  float* bout[WG_MAXG];   // bias-gradient destination or null
};

template <typename T, int MODE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  constexpr int EPV = Tr<T>::EPV, BKM = Tr<T>::BK, KSTEPS = Tr<T>::KSTEPS;
  constexpr int BM = 128, BN = 128;  // BM over Cout (rows of dW), BN over k = tap*Cin+cin
  constexpr int TM = 64, TN = 64, MT = 4, NT = 4;
  constexpr int A_BYTES = BM * 128, STAGE = 2 * A_BYTES;
  constexpr int CPR = 128 / EPV;              // channel chunks per operand row-block (16 / 32)
  constexpr int MB = BKM / EPV;               // pixel blocks per reduction tile (8)
  constexpr int BLOCKS = CPR * MB;            // EPV x EPV blocks per operand (128 / 256)
  constexpr int NB = 2 * BLOCKS / 256;        // blocks per thread (1 / 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int quad = lane >> 4, l16 = lane & 15;

  const int tiles_k = (p.K + BN - 1) / BN;
  const int tiles_n = (p.Cout + BM - 1) / BM;
  const int tiles = tiles_k * tiles_n;
  // XCD-aware order: all output tiles of one m-range (split) run on the same XCD at the same time, so the dY / X
  // rows they stream in lock-step are fetched from HBM once and then hit that XCD's L2
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / tiles;
  const int t = lid - split * tiles;
  const int n0 = (t / tiles_k) * BM, k0 = (t % tiles_k) * BN;

  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

  // per-thread block assignment
  int b_op[NB], b_cc[NB], b_mb[NB];
  int b_tap[NB], b_cin[NB];
  bool b_cvalid[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int id = tid + 256 * i;
    b_op[i] = id / BLOCKS;
    const int r = id - b_op[i] * BLOCKS;
    // 8 consecutive lanes = the 8 pixel blocks of one channel chunk: their transposed 16-byte stores fill one
    // whole 128-byte LDS row (conflict-free), and their 8 x (8 channel-chunk lanes) loads cover full 128-byte lines
    b_mb[i] = r % MB;
    b_cc[i] = r / MB;
    if (b_op[i] == 0) {
      b_cvalid[i] = (n0 + b_cc[i] * EPV) < p.Cout;
      b_tap[i] = 0; b_cin[i] = 0;
    } else {
      const int k = k0 + b_cc[i] * EPV;
      b_cvalid[i] = k < p.K;
      if (MODE == MODE_1x1) { b_tap[i] = 0; b_cin[i] = k; }
      else { b_tap[i] = k / p.Cin; b_cin[i] = k - b_tap[i] * p.Cin; }
    }
  }

  Blk<T> blk[NB];
  const int mt_begin = split * p.mtiles_per_split;
  const int mt_total = (p.M + BKM - 1) / BKM;
  const int mt_end = min(mt_total, mt_begin + p.mtiles_per_split);

// Straight-line (select-only) gather: the operand role of a block (dY rows vs im2col rows) is folded into
// address selects, so the EPV loads of a block issue back to back and are waited for once.
#define MDM_WG_LOAD(mt)                                                                                  \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                     \
      const bool is_a = b_op[i] == 0;                                                                    \
      const T* base = is_a ? DY : X;                                                                     \
      const int mbase = (mt) * BKM + b_mb[i] * EPV;                                                      \
      int pn = 0, poh = 0, pow_ = 0;                                                                     \
      if (MODE != MODE_1x1) {                                                                            \
        const int hw = p.Ho * p.Wo;                                                                      \
        pn = mbase / hw;                                                                                 \
        const int r = mbase - pn * hw;                                                                   \
        poh = r / p.Wo; pow_ = r - poh * p.Wo;                                                           \
      }                                                                                                  \
      const int kh = (b_tap[i] * 11) >> 5, kw = b_tap[i] - 3 * kh;                                       \
      const size_t a_col = (size_t)(n0 + b_cc[i] * EPV);                                                 \
      _Pragma("unroll") for (int e = 0; e < EPV; ++e) {                                                  \
        const int m = mbase + e;                                                                         \
        bool v = b_cvalid[i] && m < p.M;                                                                 \
        size_t off_b;                                                                                    \
        bool vb = true;                                                                                  \
        if (MODE == MODE_1x1) {                                                                          \
          off_b = (size_t)m * p.Cin + b_cin[i];                                                          \
        } else {                                                                                         \
          const int ih = poh * p.stride + kh - 1, iw = pow_ * p.stride + kw - 1;                         \
          vb = ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);                         \
          off_b = (size_t)((pn * p.H + ih) * p.W + iw) * p.Cin + b_cin[i];                               \
          const bool wrap_w = (pow_ + 1 == p.Wo);                                                        \
          const bool wrap_h = wrap_w && (poh + 1 == p.Ho);                                               \
          pow_ = wrap_w ? 0 : pow_ + 1;                                                                  \
          poh = wrap_h ? 0 : (wrap_w ? poh + 1 : poh);                                                   \
          pn = wrap_h ? pn + 1 : pn;                                                                     \
        }                                                                                                \
        const size_t off_a = (size_t)m * p.Cout + a_col;                                                 \
        v = v && (is_a || vb);                                                                           \
        const size_t off = is_a ? off_a : off_b;                                                         \
        blk[i].load_row_sel(e, base + (v ? off : (size_t)0), v);                                         \
      }                                                                                                  \
    }                                                                                                    \
  }
#define MDM_WG_STORE(stage)                                                                              \
  {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                     \
      char* base = (stage) + b_op[i] * A_BYTES;                                                          \
      _Pragma("unroll") for (int c = 0; c < EPV; ++c) {                                                  \
        const int row = b_cc[i] * EPV + c;                                                               \
        *reinterpret_cast<uint4*>(base + lds_chunk_off(row, b_mb[i])) = blk[i].col(c);                   \
      }                                                                                                  \
    }                                                                                                    \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (mt_begin < mt_end) {
    MDM_WG_LOAD(mt_begin);
    MDM_WG_STORE(smem);
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
      const int it = mt - mt_begin;
      char* cur = smem + (it & 1) * STAGE;
      const bool more = mt + 1 < mt_end;
      if (more) MDM_WG_LOAD(mt + 1);
      const char* As = cur;
      const char* Bs = cur + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        Frag<T> af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) load_frag<T>(af[i], As, wm * TM + i * 16 + l16, ks, quad);
#pragma unroll
        for (int j = 0; j < NT; ++j) load_frag<T>(bfr[j], Bs, wn * TN + j * 16 + l16, ks, quad);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) mma16(acc[i][j], bfr[j], af[i]);
      }
      if (more) MDM_WG_STORE(smem + ((it + 1) & 1) * STAGE);
      __syncthreads();
    }
  }
#undef MDM_WG_LOAD
#undef MDM_WG_STORE

  // acc[i][j][e] = dW[n = n0 + wm*64 + i*16 + l16][k = k0 + wn*64 + j*16 + quad*4 + e]
  float* __restrict__ S = p.slab + (size_t)split * p.Cout * p.K;
  const bool vec_ok = (p.K & 3) == 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int n = n0 + wm * TM + i * 16 + l16;
    if (n >= p.Cout) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k = k0 + wn * TN + j * 16 + quad * 4;
      if (k >= p.K) continue;
      float* o = S + (size_t)n * p.K + k;
      if (vec_ok) {
        *reinterpret_cast<f32x4*>(o) = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] = acc[i][j][e];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 wgrad, LDS-DMA + transpose-read variant.  Both operands are staged in their NATURAL layout
// ([64 pixels][128 channels], 256-byte rows) straight from HBM by global_load_lds -- no staging registers, no
// in-register transposes, no ds_write pass -- and the MFMA fragments (8 consecutive *pixels* of one channel)
// are assembled by the LDS transpose read ds_read_b64_tr_b16: in a 16-lane group lane i supplies the address
// of row (i >> 2), 4-element column segment (i & 3) of a 4 x 16 block and receives column i (probe:
// tools/probes/ds_read_tr_probe.hip).  Rows are swizzled at 32-byte granularity,
// seg' = seg ^ ((row & 3) | ((row >> 3) & 1) << 2), which makes the 8 rows a half-wave touches land on 8
// distinct 32-byte bank groups; the DMA applies the inverse permutation on the source address.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ int tr_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

typedef __attribute__((ext_vector_type(8))) short s16x8;

// Eight transpose-read fragments (4 of operand A, 4 of operand B) of one 32-pixel k-step, issued as ONE asm
// statement with its own lgkmcnt wait.  Inline asm on purpose: for the ds_read_tr builtin hipcc assumes the read
// may alias an in-flight LDS-DMA and drains vmcnt(0) in front of it, which would serialise the next tile's
// global_load_lds behind this tile's MFMAs (cdna_hip_programming.md section 5.7).  Here the reads only ever touch
// the buffer that the previous __syncthreads() (vmcnt(0) + barrier) completed.
// a0..a3 / b0..b3: per-lane byte addresses (LDS) of fragment column tiles 0..3 for reduction row quad*8 + (l16>>2);
// OFF = compile-time byte offset of the k-step inside the tile (ks * 32 rows * 256 B).
template <int OFF>
__device__ __forceinline__ void load_frags_tr(Frag<bf16> (&af)[4], Frag<bf16> (&bf_)[4], unsigned a0, unsigned a1,
                                              unsigned a2, unsigned a3, unsigned b0, unsigned b1, unsigned b2,
                                              unsigned b3) {
  s16x4 r[16];
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%24\n\t"
      "ds_read_b64_tr_b16 %1, %16 offset:%25\n\t"
      "ds_read_b64_tr_b16 %2, %17 offset:%24\n\t"
      "ds_read_b64_tr_b16 %3, %17 offset:%25\n\t"
      "ds_read_b64_tr_b16 %4, %18 offset:%24\n\t"
      "ds_read_b64_tr_b16 %5, %18 offset:%25\n\t"
      "ds_read_b64_tr_b16 %6, %19 offset:%24\n\t"
      "ds_read_b64_tr_b16 %7, %19 offset:%25\n\t"
      "ds_read_b64_tr_b16 %8, %20 offset:%24\n\t"
      "ds_read_b64_tr_b16 %9, %20 offset:%25\n\t"
      "ds_read_b64_tr_b16 %10, %21 offset:%24\n\t"
      "ds_read_b64_tr_b16 %11, %21 offset:%25\n\t"
      "ds_read_b64_tr_b16 %12, %22 offset:%24\n\t"
      "ds_read_b64_tr_b16 %13, %22 offset:%25\n\t"
      "ds_read_b64_tr_b16 %14, %23 offset:%24\n\t"
      "ds_read_b64_tr_b16 %15, %23 offset:%25\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
        "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "i"(OFF), "i"(OFF + 1024)
      : "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s16x8 va = {r[2 * i][0], r[2 * i][1], r[2 * i][2], r[2 * i][3], r[2 * i + 1][0], r[2 * i + 1][1], r[2 * i + 1][2], r[2 * i + 1][3]};
    af[i].v = *reinterpret_cast<bf16x8*>(&va);
    s16x8 vb = {r[8 + 2 * i][0], r[8 + 2 * i][1], r[8 + 2 * i][2], r[8 + 2 * i][3], r[9 + 2 * i][0], r[9 + 2 * i][1], r[9 + 2 * i][2], r[9 + 2 * i][3]};
    bf_[i].v = *reinterpret_cast<bf16x8*>(&vb);
  }
}

// 256 x 256 output tile, 8 waves (2 over Cout x 4 over k), wave tile 128 x 64: the same LDS-DMA + transpose-read
// scheme with half the LDS bytes per FLOP of the 128 x 128 kernel.  Natural-layout rows are 512 bytes here.
template <int OFF>
__device__ __forceinline__ void load_frags_tr_big(Frag<bf16> (&af)[8], Frag<bf16> (&bf_)[4], const unsigned (&a)[8],
                                                  const unsigned (&b)[4]) {
  // OFF = byte offset of the k-step; the upper 4 pixel rows of a fragment are +4 * 512 = 2048 bytes
  s16x4 r[24];
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16 offset:%24\n\t"
      "ds_read_b64_tr_b16 %1, %16 offset:%25\n\t"
      "ds_read_b64_tr_b16 %2, %17 offset:%24\n\t"
      "ds_read_b64_tr_b16 %3, %17 offset:%25\n\t"
      "ds_read_b64_tr_b16 %4, %18 offset:%24\n\t"
      "ds_read_b64_tr_b16 %5, %18 offset:%25\n\t"
      "ds_read_b64_tr_b16 %6, %19 offset:%24\n\t"
      "ds_read_b64_tr_b16 %7, %19 offset:%25\n\t"
      "ds_read_b64_tr_b16 %8, %20 offset:%24\n\t"
      "ds_read_b64_tr_b16 %9, %20 offset:%25\n\t"
      "ds_read_b64_tr_b16 %10, %21 offset:%24\n\t"
      "ds_read_b64_tr_b16 %11, %21 offset:%25\n\t"
      "ds_read_b64_tr_b16 %12, %22 offset:%24\n\t"
      "ds_read_b64_tr_b16 %13, %22 offset:%25\n\t"
      "ds_read_b64_tr_b16 %14, %23 offset:%24\n\t"
      "ds_read_b64_tr_b16 %15, %23 offset:%25"
      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]),
        "=&v"(r[8]), "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "i"(OFF), "i"(OFF + 2048)
      : "memory");
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
      "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
      "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
      "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
      "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:%13"
      : "=&v"(r[16]), "=&v"(r[17]), "=&v"(r[18]), "=&v"(r[19]), "=&v"(r[20]), "=&v"(r[21]), "=&v"(r[22]), "=&v"(r[23])
      : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "i"(OFF), "i"(OFF + 2048)
      : "memory");
  // one wait for all 24 reads; every destination is named so nothing is consumed (or moved) before it
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]),
                 "+v"(r[16]), "+v"(r[17]), "+v"(r[18]), "+v"(r[19]), "+v"(r[20]), "+v"(r[21]), "+v"(r[22]), "+v"(r[23])
               :
               : "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s16x8 v = {r[2 * i][0], r[2 * i][1], r[2 * i][2], r[2 * i][3], r[2 * i + 1][0], r[2 * i + 1][1], r[2 * i + 1][2], r[2 * i + 1][3]};
    af[i].v = *reinterpret_cast<bf16x8*>(&v);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s16x8 v = {r[16 + 2 * i][0], r[16 + 2 * i][1], r[16 + 2 * i][2], r[16 + 2 * i][3], r[17 + 2 * i][0], r[17 + 2 * i][1], r[17 + 2 * i][2], r[17 + 2 * i][3]};
    bf_[i].v = *reinterpret_cast<bf16x8*>(&v);
  }
}

// BIG = 0: 128 x 128 tile, 4 waves (wave tile 64 x 64), 256-byte rows, 2 blocks / CU
// BIG = 1: 256 x 256 tile, 8 waves (wave tile 128 x 64), 512-byte rows, 1 block / CU
template <int MODE, int BIG>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void conv_wgrad_tr_kernel(WgradArgs p) {
  typedef bf16 T;
  constexpr int EPV = 8, BKM = 64;
  constexpr int NTHR = BIG ? 512 : 256;
  constexpr int BM = BIG ? 256 : 128, BN = BM;                 // BM over Cout, BN over k
  constexpr int TM = BIG ? 128 : 64, TN = 64, MT = TM / 16, NT = 4;
  constexpr int ROWB = BM * 2;                                 // bytes per natural-layout row (256 / 512)
  constexpr int CPRW = ROWB / 16;                              // 16-byte chunks per row (16 / 32)
  constexpr int RPP = NTHR / CPRW;                             // rows staged per pass (16)
  constexpr int A_BYTES = BKM * ROWB, STAGE = 2 * A_BYTES;     // 16 / 32 KB per operand
  constexpr int NWN = BIG ? 4 : 2;                             // waves across k
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int quad = lane >> 4, l16 = lane & 15;

  const int tiles_k = (p.K + BN - 1) / BN;
  const int tiles_n = (p.Cout + BM - 1) / BM;
  const int tiles = tiles_k * tiles_n;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / tiles;
  const int t = lid - split * tiles;
  const int n0 = (t / tiles_k) * BM, k0 = (t % tiles_k) * BN;

  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

  // staging role: physical 16-byte slot `ps` of rows srow + RPP * j (j = 0..3); RPP = 16 keeps row bits 0,1,3 that
  // feed tr_swz unchanged across passes, so one logical chunk (hence one (tap, cin)) serves all four passes
  const int ps = tid % CPRW;
  const int srow = tid / CPRW;
  const int lc = ((((ps >> 1) ^ tr_swz(srow)) << 1) | (ps & 1));
  const int a_col = n0 + lc * EPV;
  const bool a_ok = a_col < p.Cout;
  const int kk = k0 + lc * EPV;
  const bool b_ok = kk < p.K;
  int tap = 0, cin = kk;
  if (MODE != MODE_1x1) { tap = kk / p.Cin; cin = kk - tap * p.Cin; }
  const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);

  const int mt_begin = split * p.mtiles_per_split;
  const int mt_total = (p.M + BKM - 1) / BKM;
  const int mt_end = min(mt_total, mt_begin + p.mtiles_per_split);

  int pn[4], poh[4], pow_[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = mt_begin * BKM + j * RPP + srow;
    if (MODE != MODE_1x1) {
      const int hw = p.Ho * p.Wo;
      pn[j] = m / hw;
      const int r = m - pn[j] * hw;
      poh[j] = r / p.Wo; pow_[j] = r - poh[j] * p.Wo;
    } else { pn[j] = 0; poh[j] = 0; pow_[j] = 0; }
  }
  int m_next = mt_begin * BKM;

#define MDM_GLDS(src, lds_off)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                  \
                                   (__attribute__((address_space(3))) void*)(lds_off), 16, 0, 0)
#define MDM_WG_STAGE(stage)                                                                               \
  {                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                       \
      const int m = m_next + j * RPP + srow;                                                              \
      const bool mv = m < p.M;                                                                            \
      const T* sa = (mv && a_ok) ? DY + (size_t)m * p.Cout + a_col : reinterpret_cast<const T*>(g_zero_page); \
      MDM_GLDS(sa, (stage) + j * (RPP * ROWB) + wave_lds);                                                \
      bool v = mv && b_ok;                                                                                \
      size_t off;                                                                                         \
      if (MODE == MODE_1x1) {                                                                             \
        off = (size_t)m * p.Cin + cin;                                                                    \
      } else {                                                                                            \
        const int ih = poh[j] * p.stride + kh - 1, iw = pow_[j] * p.stride + kw - 1;                      \
        v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);                        \
        off = (size_t)((pn[j] * p.H + ih) * p.W + iw) * p.Cin + cin;                                      \
        pow_[j] += BKM;                                                                                   \
        while (pow_[j] >= p.Wo) { pow_[j] -= p.Wo; ++poh[j]; }                                            \
        while (poh[j] >= p.Ho) { poh[j] -= p.Ho; ++pn[j]; }                                               \
      }                                                                                                   \
      const T* sb = v ? X + off : reinterpret_cast<const T*>(g_zero_page);                                \
      MDM_GLDS(sb, (stage) + A_BYTES + j * (RPP * ROWB) + wave_lds);                                      \
    }                                                                                                     \
    m_next += BKM;                                                                                        \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of dY: the dY^T fragments are already in registers, so the blocks of k-tile 0
  // (waves wn == 0) multiply them by an all-ones fragment -- MT extra MFMAs per k-step in 1/tiles_k of the blocks
  const bool do_bias = p.bslab != nullptr && k0 == 0 && wn == 0;
  f32x4 bacc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  Frag<T> ones;
  {
    const bf16 o1 = (bf16)1.0f;
    ones.v = bf16x8{o1, o1, o1, o1, o1, o1, o1, o1};
  }

  // per-lane LDS byte addresses of the transpose reads: reduction row quad*8 + (l16>>2) (+4 for the upper half,
  // +32 rows per k-step: immediates), 16-channel column tile `c`, 4-element segment l16 & 3
  unsigned fa[MT], fb[NT];
  {
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int r0 = quad * 8 + (l16 >> 2);
    const int fsw = (l16 >> 2) | ((quad & 1) << 2);
#pragma unroll
    for (int c = 0; c < MT; ++c) fa[c] = smem_base + r0 * ROWB + (((wm * MT + c) ^ fsw) << 5) + ((l16 & 3) << 3);
#pragma unroll
    for (int c = 0; c < NT; ++c) fb[c] = smem_base + A_BYTES + r0 * ROWB + (((wn * NT + c) ^ fsw) << 5) + ((l16 & 3) << 3);
  }

  if (mt_begin < mt_end) {
    MDM_WG_STAGE(smem);
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
      const int it = mt - mt_begin;
      if (mt + 1 < mt_end) MDM_WG_STAGE(smem + ((it + 1) & 1) * STAGE);
      const unsigned so = (unsigned)((it & 1) * STAGE);
      if constexpr (BIG) {
        unsigned a8[8], b4[4];
#pragma unroll
        for (int c = 0; c < 8; ++c) a8[c] = fa[c] + so;
#pragma unroll
        for (int c = 0; c < 4; ++c) b4[c] = fb[c] + so;
        {
          Frag<T> af[8], bfr[4];
          load_frags_tr_big<0>(af, bfr, a8, b4);
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(acc[i][j], bfr[j], af[i]);
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mma16(bacc[i], ones, af[i]);
          }
        }
        {
          Frag<T> af[8], bfr[4];
          load_frags_tr_big<32 * ROWB>(af, bfr, a8, b4);
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(acc[i][j], bfr[j], af[i]);
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mma16(bacc[i], ones, af[i]);
          }
        }
      } else {
        {
          Frag<T> af[4], bfr[4];
          load_frags_tr<0>(af, bfr, fa[0] + so, fa[1] + so, fa[2] + so, fa[3] + so, fb[0] + so, fb[1] + so, fb[2] + so, fb[3] + so);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(acc[i][j], bfr[j], af[i]);
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mma16(bacc[i], ones, af[i]);
          }
        }
        {
          Frag<T> af[4], bfr[4];
          load_frags_tr<8192>(af, bfr, fa[0] + so, fa[1] + so, fa[2] + so, fa[3] + so, fb[0] + so, fb[1] + so, fb[2] + so, fb[3] + so);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma16(acc[i][j], bfr[j], af[i]);
          if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mma16(bacc[i], ones, af[i]);
          }
        }
      }
      __syncthreads();
    }
  }
#undef MDM_WG_STAGE
#undef MDM_GLDS

  if (do_bias && quad == 0) {
    // D = ones * dY^T-fragment: every row of the 16x16 result holds the column sums; lane l16 of quad 0 owns Cout n
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int n = n0 + wm * TM + i * 16 + l16;
      if (n < p.Cout) p.bslab[(size_t)split * p.Cout + n] = bacc[i][0];
    }
  }
  if (p.bslab != nullptr && blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(p.bslab - WG_BHDR)[0] = 1;   // one row per split
  float* __restrict__ S = p.slab + (size_t)split * p.Cout * p.K;
  const bool vec_ok = (p.K & 3) == 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int n = n0 + wm * TM + i * 16 + l16;
    if (n >= p.Cout) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k = k0 + wn * TN + j * 16 + quad * 4;
      if (k >= p.K) continue;
      float* o = S + (size_t)n * p.K + k;
      if (vec_ok) {
        *reinterpret_cast<f32x4*>(o) = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] = acc[i][j][e];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// bf16 wgrad main path: the tile / LDS image / transpose reads of conv_wgrad_tr_kernel with the loader and k-loop
// of conv_gemm_bl_kernel.
//   * Loader: both operands are rows of a pixel-major matrix whose row index advances by 64 per reduction tile, so
//     a lane's gather offset is fixed for the whole loop (row-in-tile, channel chunk, tap shift) and the tile is a
//     wave-uniform SGPR offset.  Per tile the only vector work is the halo / tail test of the 4 rows a lane stages
//     (H, W powers of two: shifts and masks); rows that fail use an offset past num_records (reads zeros).
//   * k-loop: phase A = MFMAs of k-step 0 with the k-step-1 fragments re-loaded row by row behind them, barrier
//     (LDS reads of the tile drained, next tile landed), phase B = MFMAs of k-step 1 with the DMA of tile + 2 and
//     the k-step-0 fragments of tile + 1 behind them.  Transpose reads stay inline asm (the builtin makes hipcc
//     drain vmcnt before every read); their completion is counted by hand: lgkmcnt(N) ahead of phase-A rows (N
//     = LDS ops issued after that row's reads, capped at 15), lgkmcnt(0) at the barrier.
// Host-checked requirements (else conv_wgrad_tr_kernel): 1x1, or 3x3 stride 1 with H, W powers of two; operands
// below 0x7F000000 bytes.
// ---------------------------------------------------------------------------
#define MDM_TR2(lo, hi, addr, OFF)                                                                          \
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                \
               : "=&v"(lo), "=&v"(hi)                                                                       \
               : "v"(addr), "n"(OFF), "n"((OFF) + 4 * ROWB)                                                 \
               : "memory")

// the two 64-bit halves of a transpose-read fragment stay dword vectors until they are concatenated (16-bit element
// shuffles would make hipcc touch the registers while the read is still in flight)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ Frag<bf16> frag_of(const u32x2& lo, const u32x2& hi) {
  const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, v);
  return f;
}

// acc + sum of the 8 bf16 values of a fragment
__device__ __forceinline__ float frag_sum(const Frag<bf16>& f, float acc) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 one = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
  for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{f.v[2 * q], f.v[2 * q + 1]}, one, acc, false);
  return acc;
}

// the same sum as one opaque block: the compiler may not speculate it, so a wave-uniform `if` around it stays a scalar
// branch (as plain code hipcc if-converts the four dot products + a v_cndmask into every wave's instruction stream)
__device__ __forceinline__ float frag_sum_guarded(const Frag<bf16>& f, float acc) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t r = __builtin_bit_cast(u32x4_t, f.v);
  asm volatile("v_dot2c_f32_bf16 %0, 0x3f803f80, %1\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %2\n\t"
               "v_dot2c_f32_bf16 %0, 0x3f803f80, %3\n\tv_dot2c_f32_bf16 %0, 0x3f803f80, %4"
               : "+v"(acc) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
  return acc;
}

template <int MODE, int BIG>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void conv_wgrad_bl_kernel(WgradArgs p, WgradGroup gr) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef bf16 T;
  constexpr int EPV = 8, BKM = 64;
  constexpr int NTHR = BIG ? 512 : 256;
  constexpr int BM = BIG ? 256 : 128, BN = BM;
  constexpr int TM = BIG ? 128 : 64, TN = 64, MT = TM / 16, NT = 4;
  constexpr int ROWB = BM * 2;
  constexpr int CPRW = ROWB / 16;
  constexpr int RPP = NTHR / CPRW;                             // 16 rows per staging pass
  constexpr int A_BYTES = BKM * ROWB, STAGE = 2 * A_BYTES;
  constexpr int NWN = BIG ? 4 : 2;
  constexpr int KS1 = 32 * ROWB;                               // byte offset of k-step 1 inside a tile
  constexpr unsigned INVALID = 0x7F000000u;
  constexpr int WAITN = 2 * (MT + NT - 1) > 15 ? 15 : 2 * (MT + NT - 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int quad = lane >> 4, l16 = lane & 15;

  const int tiles_k = (p.K + BN - 1) / BN;
  const int tiles_n = (p.Cout + BM - 1) / BM;
  const int tiles = tiles_k * tiles_n;
  int lid = xcd_remap(blockIdx.x, gridDim.x);
  int grp = 0;
  if (p.groups > 0) { grp = lid / (tiles * p.splits); lid -= grp * tiles * p.splits; }
  const int split = lid / tiles;
  const int t = lid - split * tiles;
  const int n0 = (t / tiles_k) * BM, k0 = (t % tiles_k) * BN;
  if (MODE == MODE_3x3 && p.skip_cout > 0) {
    const int ph = n0 / p.skip_cout, tp = k0 / p.Cin;   // tiles never straddle a phase / a tap (host-checked)
    const int dj = tp / 3 - (ph >> 1), dk = tp % 3 - (ph & 1);
    if ((unsigned)dj > 1u || (unsigned)dk > 1u) return;   // block-uniform, before any barrier
  }
  const void* const x_ptr = p.groups > 0 ? gr.x[grp] : p.x;
  const void* const dy_ptr = p.groups > 0 ? gr.dy[grp] : p.dy;

  // ---- loader role (see conv_wgrad_tr_kernel): physical slot ps of rows srow + 16 j --------------------------
  const int ps = tid % CPRW;
  const int srow = tid / CPRW;
  const int lc = ((((ps >> 1) ^ tr_swz(srow)) << 1) | (ps & 1));
  const int a_col = n0 + lc * EPV;
  const int kk = k0 + lc * EPV;
  int tap = 0, cin = kk;
  if (MODE != MODE_1x1) { tap = kk / p.Cin; cin = kk - tap * p.Cin; }
  const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
  const unsigned bias = MODE == MODE_3x3 ? (unsigned)(p.W + 1) * p.Cin * 2u : 0u;
  unsigned a_vo[4], b_vo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned r = j * RPP + srow;
    a_vo[j] = a_col < p.Cout ? (r * p.Cout + a_col) * 2u : INVALID;
    b_vo[j] = kk < p.K ? (r * p.Cin + cin) * 2u + (MODE == MODE_3x3 ? (unsigned)(kh * p.W + kw) * p.Cin * 2u : 0u) : INVALID;
  }
  const unsigned a_bytes = (unsigned)p.M * p.Cout * 2u;
  const unsigned b_bytes = (unsigned)p.N * p.H * p.W * p.Cin * 2u + bias;
  char* const a_base = const_cast<char*>(reinterpret_cast<const char*>(dy_ptr));
  char* const b_base = const_cast<char*>(reinterpret_cast<const char*>(x_ptr)) - bias;
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
  const int logW = 31 - __builtin_clz(p.W);

  const int mt_begin = split * p.mtiles_per_split;
  const int mt_total = (p.M + BKM - 1) / BKM;
  const int mt_end = min(mt_total, mt_begin + p.mtiles_per_split);
  const int nt = mt_end - mt_begin;

#define MDM_BLDS(rs, lds_off, voff, soff)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds_off), 16, voff, soff, 0, 0)
  // wave-uniform state of reduction tile `it_` of this split (empty descriptors past the last one)
#define MDM_WG_TILE_STATE(it_)                                                                              \
  const bool more_ = (it_) < nt;                                                                            \
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(a_base, 0, more_ ? a_bytes : 0u, 0x00020000);         \
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(b_base, 0, more_ ? b_bytes : 0u, 0x00020000);         \
  const int mb_ = (mt_begin + (it_)) * BKM;                                                                 \
  const int a_soff = mb_ * p.Cout * 2, b_soff = mb_ * p.Cin * 2;
  // piece q (0..7) of a tile: row pass q >> 1, operand q & 1 (dY, then X)
#define MDM_WG_PIECE(stage, q)                                                                              \
  {                                                                                                         \
    constexpr int j_ = (q) >> 1;                                                                            \
    const int m_ = mb_ + j_ * RPP + srow;                                                                   \
    bool v_ = m_ < p.M;                                                                                     \
    if (((q) & 1) == 0) {                                                                                   \
      MDM_BLDS(rsA, (stage) + j_ * (RPP * ROWB) + wave_lds, v_ ? a_vo[j_] : INVALID, a_soff);               \
    } else {                                                                                                \
      if (MODE == MODE_3x3) {                                                                               \
        const int ow_ = m_ & (p.W - 1), oh_ = (m_ >> logW) & (p.H - 1);                                     \
        v_ = v_ && (unsigned)(oh_ + kh - 1) < (unsigned)p.H && (unsigned)(ow_ + kw - 1) < (unsigned)p.W;    \
      }                                                                                                     \
      MDM_BLDS(rsB, (stage) + A_BYTES + j_ * (RPP * ROWB) + wave_lds, v_ ? b_vo[j_] : INVALID, b_soff);     \
    }                                                                                                       \
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (with tile skipping the first tile a phase computes is its first tap's, not tap 0's)
  const int bias_k0 = (MODE == MODE_3x3 && p.skip_cout > 0) ? (((n0 / p.skip_cout) >> 1) * 3 + ((n0 / p.skip_cout) & 1)) * p.Cin : 0;
  // bias gradient = column sums of dY: a wave adds up dY^T fragments it already holds (8 pixels of one channel per lane)
  // with v_dot2c_f32_bf16 against (1, 1) -- one fp32 register per fragment row.  A wave that sums all of its MT rows is
  // ~30 % slower in that reduction tile, and when one k-tile block in tiles_k did that for all of its tiles the whole
  // launch took +28 % (that block is the tail of a one-round launch; r05).  Every k-tile block of one (split, n-tile) and
  // every wn wave of a block holds the SAME fragments, so the work is dealt out twice:
  //   * wave wn sums only rows wn * BR .. wn * BR + BR - 1 of its MT (BR = MT / NWN = 2; the block's waves then cover
  //     every channel exactly once -- no cross-wave sum), and
  //   * k-tile block kt_rel of the first `share` sums only the kt_rel-th chunk of the split's reduction tiles: plain loop
  //     up to the chunk, summing loop over it, accumulators written out (row split * share + kt_rel of bslab; the reduce
  //     kernel adds the rows), plain loop for the rest -- the accumulators live in the summing loop only.  (Walking the
  //     tiles in the same order as the other k-tile blocks matters: they stream the same dY rows through the XCD's L2;
  //     a block that started at its chunk instead cost the launch +13-16 %.)
  // Grouped launches add straight into the arena (one writer per channel): share = 1.
  constexpr int BR = MT / NWN;
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  const int bias_share = p.groups > 0 ? 1 : p.bias_share;
  const int kt_rel = k0 >= bias_k0 ? (k0 - bias_k0) / BN : bias_share;
  const bool bias_blk = (p.groups > 0 ? gr.bout[grp] != nullptr : p.bslab != nullptr) && kt_rel < bias_share;   // block-uniform
  const int bias_chunk = (nt + bias_share - 1) / bias_share;
  const int bias_b0 = bias_blk ? min(kt_rel * bias_chunk, nt) : nt;    // reduction tiles [bias_b0, bias_b1) are summed here
  const int bias_b1 = bias_blk ? min(bias_b0 + bias_chunk, nt) : nt;

  unsigned fa[MT], fb[NT];
  {
    const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int r0 = quad * 8 + (l16 >> 2);
    const int fsw = (l16 >> 2) | ((quad & 1) << 2);
#pragma unroll
    for (int c = 0; c < MT; ++c) fa[c] = smem_base + r0 * ROWB + (((wm * MT + c) ^ fsw) << 5) + ((l16 & 3) << 3);
#pragma unroll
    for (int c = 0; c < NT; ++c) fb[c] = smem_base + A_BYTES + r0 * ROWB + (((wn * NT + c) ^ fsw) << 5) + ((l16 & 3) << 3);
  }

  if (nt > 0) {
    {
      MDM_WG_TILE_STATE(0);
      MDM_WG_PIECE(smem, 0); MDM_WG_PIECE(smem, 1); MDM_WG_PIECE(smem, 2); MDM_WG_PIECE(smem, 3);
      MDM_WG_PIECE(smem, 4); MDM_WG_PIECE(smem, 5); MDM_WG_PIECE(smem, 6); MDM_WG_PIECE(smem, 7);
    }
    {
      MDM_WG_TILE_STATE(1);
      MDM_WG_PIECE(smem + STAGE, 0); MDM_WG_PIECE(smem + STAGE, 1); MDM_WG_PIECE(smem + STAGE, 2); MDM_WG_PIECE(smem + STAGE, 3);
      MDM_WG_PIECE(smem + STAGE, 4); MDM_WG_PIECE(smem + STAGE, 5); MDM_WG_PIECE(smem + STAGE, 6); MDM_WG_PIECE(smem + STAGE, 7);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // raw transpose-read results in flight: [fragment][pixels 0-3 / 4-7]
    u32x2 ra[MT][2], rb0[NT][2], rb1[NT][2];
#pragma unroll
    for (int c = 0; c < NT; ++c) MDM_TR2(rb0[c][0], rb0[c][1], fb[c], 0);
#pragma unroll
    for (int c = 0; c < MT; ++c) MDM_TR2(ra[c][0], ra[c][1], fa[c], 0);
    __builtin_amdgcn_sched_barrier(0);

    // one reduction tile; BIAS = this wave also accumulates the column sums of dY (ones x dY^T fragment)
#define MDM_WG_ITER(BIAS)                                                                                   \
    {                                                                                                       \
      const unsigned so = (unsigned)((it & 1) * STAGE), sn = (unsigned)(((it + 1) & 1) * STAGE);            \
      /* ---- phase A: k-step 0 of tile it */                                                               \
      _Pragma("unroll") for (int c = 0; c < NT; ++c) { const unsigned ad = fb[c] + so; MDM_TR2(rb1[c][0], rb1[c][1], ad, KS1); } \
      Frag<T> bq[NT];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                      \
        if (i == 0) {                                                                                       \
          asm volatile("s_waitcnt lgkmcnt(%8)"                                                              \
                       : "+v"(rb0[0][0]), "+v"(rb0[0][1]), "+v"(rb0[1][0]), "+v"(rb0[1][1]), "+v"(rb0[2][0]), \
                         "+v"(rb0[2][1]), "+v"(rb0[3][0]), "+v"(rb0[3][1])                                  \
                       : "n"(WAITN) : "memory");                                                            \
          _Pragma("unroll") for (int j = 0; j < NT; ++j) bq[j] = frag_of(rb0[j][0], rb0[j][1]);             \
        }                                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ra[i][0]), "+v"(ra[i][1]) : "n"(WAITN) : "memory");     \
        const Frag<T> aq = frag_of(ra[i][0], ra[i][1]);                                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) mma16(acc[i][j], bq[j], aq);                         \
        if (BIAS && i / BR == wn_s) bsum[i % BR] = frag_sum_guarded(aq, bsum[i % BR]);                      \
        { const unsigned ad = fa[i] + so; MDM_TR2(ra[i][0], ra[i][1], ad, KS1); }                           \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
      /* every LDS read of tile it has returned; tile it + 1 has landed */                                  \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
      __builtin_amdgcn_s_barrier();                                                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      /* ---- phase B: k-step 1 of tile it; DMA of tile it + 2; k-step-0 fragments of tile it + 1 */        \
      MDM_WG_TILE_STATE(it + 2);                                                                            \
      char* const dst = smem + so;                                                                          \
      _Pragma("unroll") for (int c = 0; c < NT; ++c) { const unsigned ad = fb[c] + sn; MDM_TR2(rb0[c][0], rb0[c][1], ad, 0); } \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) bq[j] = frag_of(rb1[j][0], rb1[j][1]);                 \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                      \
        const Frag<T> aq = frag_of(ra[i][0], ra[i][1]);                                                     \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) mma16(acc[i][j], bq[j], aq);                         \
        if (BIAS && i / BR == wn_s) bsum[i % BR] = frag_sum_guarded(aq, bsum[i % BR]);                      \
        if (MT == 8) { MDM_WG_PIECE_RT(dst, i) } else { MDM_WG_PIECE_RT(dst, 2 * i) MDM_WG_PIECE_RT(dst, 2 * i + 1) } \
        { const unsigned ad = fa[i] + sn; MDM_TR2(ra[i][0], ra[i][1], ad, 0); }                             \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
      }                                                                                                     \
    }
    // piece index known only after unrolling: dispatch to the literal forms
#define MDM_WG_PIECE_RT(stage, q)                                                                           \
    switch (q) {                                                                                            \
      case 0: MDM_WG_PIECE(stage, 0) break; case 1: MDM_WG_PIECE(stage, 1) break;                           \
      case 2: MDM_WG_PIECE(stage, 2) break; case 3: MDM_WG_PIECE(stage, 3) break;                           \
      case 4: MDM_WG_PIECE(stage, 4) break; case 5: MDM_WG_PIECE(stage, 5) break;                           \
      case 6: MDM_WG_PIECE(stage, 6) break; default: MDM_WG_PIECE(stage, 7) break;                          \
    }
    // the two loop bodies execute the same number of barriers, so the waves of a block may take different ones
    // the two loop bodies execute the same number of barriers
    int it = 0;
    float bsum[BR];
    for (; it < bias_b0; ++it) MDM_WG_ITER(false)
#pragma unroll
    for (int i = 0; i < BR; ++i) bsum[i] = 0.f;
    for (; it < bias_b1; ++it) MDM_WG_ITER(true)
    if (bias_blk) {
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        float v = bsum[r];                     // lanes l16, l16 + 16, + 32, + 48 hold the four pixel groups of a k-step
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int n = n0 + wm * TM + (wn * BR + r) * 16 + l16;
        if (quad == 0 && n < p.Cout) {
          if (p.groups > 0) gr.bout[grp][n] += v;   // splits == 1, share == 1: this wave is the only writer of channel n
          else p.bslab[((size_t)split * bias_share + kt_rel) * p.Cout + n] = v;
        }
      }
    }
    for (; it < nt; ++it) MDM_WG_ITER(false)
    // the fragments pre-read for the tile after the last are never used; retire them before the registers die
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#undef MDM_WG_PIECE_RT
#undef MDM_WG_ITER
  } else if (bias_blk && p.groups == 0 && quad == 0) {
    // a split without reduction tiles (the host's split counts do not produce one): its rows of partial sums are zeros
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      const int n = n0 + wm * TM + (wn * BR + r) * 16 + l16;
      if (n < p.Cout) p.bslab[((size_t)split * bias_share + kt_rel) * p.Cout + n] = 0.f;
    }
  }
#undef MDM_WG_PIECE
#undef MDM_WG_TILE_STATE
#undef MDM_BLDS

  if (bias_blk && p.groups == 0 && split == 0 && kt_rel == 0 && n0 == 0 && tid == 0)
    reinterpret_cast<int*>(p.bslab - WG_BHDR)[0] = bias_share;
  // grouped launches (splits == 1) add the finished tile straight into the layer's gradient-arena slot: every output
  // element has exactly one owner, so a plain read-modify-write suffices
  // The two forms are separate loops on purpose (round 6): written as one loop with `direct ? *o + acc : acc`, the wait for
  // the (possibly absent) load sat at the join of the two branches, in front of EVERY store -- and gfx950's single in-order
  // vmcnt turns a wait behind a store into a wait for that store's acknowledgement: the 32-128 stores of a thread went
  // out one memory round trip at a time, in the slab form too (tools/store_wait_scan.py: 162 such waits in this kernel).
  const bool direct = p.groups > 0;
  float* __restrict__ S = direct ? gr.out[grp] : p.slab + (size_t)split * p.Cout * p.K;
  const bool vec_ok = (p.K & 3) == 0;
  if (!direct) {
    // slab form: plain stores, nothing is loaded
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int n = n0 + wm * TM + i * 16 + l16;
      if (n >= p.Cout) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int k = k0 + wn * TN + j * 16 + quad * 4;
        if (k >= p.K) continue;
        float* o = S + (size_t)n * p.K + k;
        if (vec_ok) {
          *reinterpret_cast<f32x4*>(o) = acc[i][j];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] = acc[i][j][e];
        }
      }
    }
  } else if (vec_ok && n0 + BM <= p.Cout && k0 + BN <= p.K) {
    // arena form, whole tile (the grouped launches' layers are multiples of the tile): row i + 1's old values are
    // requested BEFORE row i's sums are stored, so no load is ever younger than a store it has to wait behind
    float* const o0 = S + (size_t)(n0 + wm * TM + l16) * p.K + k0 + wn * TN + quad * 4;
    f32x4 old[2][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) old[0][j] = *reinterpret_cast<const f32x4*>(o0 + j * 16);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i + 1 < MT) {
#pragma unroll
        for (int j = 0; j < NT; ++j) old[(i + 1) & 1][j] = *reinterpret_cast<const f32x4*>(o0 + (size_t)(i + 1) * 16 * p.K + j * 16);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(o0 + (size_t)i * 16 * p.K + j * 16) = old[i & 1][j] + acc[i][j];
    }
  } else {
    // arena form, ragged tile: element by element (every output element has exactly one owner)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int n = n0 + wm * TM + i * 16 + l16;
      if (n >= p.Cout) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int k = k0 + wn * TN + j * 16 + quad * 4;
        if (k >= p.K) continue;
        float* o = S + (size_t)n * p.K + k;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < p.K) o[e] += acc[i][j][e];
      }
    }
  }
#endif
}
#undef MDM_TR2

// Bias-gradient part of the reduce kernels: dbias[o] (+)= sum over the splits * share rows of partials (share: the int the
// producing kernel left at bslab[-WG_BHDR]).  32 channels per block, the rows dealt to the block's NT / 32 thread groups
// (fixed order -> reproducible), the groups' sums added up through LDS.
template <int NT>
__device__ __forceinline__ void wgrad_bias_reduce(const float* __restrict__ bslab, float* __restrict__ dbias, int splits,
                                                  int Cout, int accumulate, int blk, float* lds) {
  constexpr int G = NT / 32;
  const int tid = threadIdx.x, c = tid & 31, g = tid >> 5;
  const int rows = splits * reinterpret_cast<const int*>(bslab - WG_BHDR)[0];
  const int o = blk * 32 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (o < Cout) {
    const float* src = bslab + o;
    int r = g;
    for (; r + 3 * G < rows; r += 4 * G) {
      s0 += src[(size_t)r * Cout];
      s1 += src[(size_t)(r + G) * Cout];
      s2 += src[(size_t)(r + 2 * G) * Cout];
      s3 += src[(size_t)(r + 3 * G) * Cout];
    }
    for (; r < rows; r += G) s0 += src[(size_t)r * Cout];
  }
  lds[tid] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && o < Cout) {
    float s = lds[c];
#pragma unroll
    for (int q = 1; q < G; ++q) s += lds[q * 32 + c];
    dbias[o] = accumulate ? dbias[o] + s : s;
  }
}

// dW_oihw[o][i][t] = sum_s slab[s][o][t*Cin + i]      (taps = 1 or 9)
// One block per (o, 64-channel block): the 9 x 64 slab values are read as 9 contiguous runs, transposed through
// LDS and written as one contiguous run of 576 floats, so both sides are coalesced.  The first `bblocks` blocks of the
// grid reduce the bias-gradient partials (wgrad_bias_reduce).
// Round 4: this kernel read its slabs at 0.5 TB/s (66 MB in 133 us per 3x3 layer of the 64x64 level, 5.3 ms per step):
// 4 output channels per 256-thread block meant 256 blocks for a 256 x 256 layer, each thread walking three groups one
// after the other with four loads in flight.  Now ONE output channel per block (taps x 16 = 144 active lanes, one 16-byte
// group each: 1024+ blocks) and the split loop unrolled 8x with independent accumulators.
__global__ __launch_bounds__(192) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           const float* __restrict__ bslab, float* __restrict__ dbias,
                                                           int splits, int Cout, int Cin, int taps, int accumulate,
                                                           int bblocks) {
  __shared__ float tile[64 * 9 + 4];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < bblocks) {   // bias part first (its blocks walk splits * share rows): 32 output channels per block
    wgrad_bias_reduce<192>(bslab, dbias, splits, Cout, accumulate, (int)blockIdx.x, tile);
    return;
  }
  const int wb = (int)blockIdx.x - bblocks;
  const int iblocks = (Cin + 63) / 64;
  const int o = wb / iblocks, i0 = (wb - o * iblocks) * 64;
  const int ni = min(64, Cin - i0);            // Cin % 4 == 0 (host-checked) -> ni % 4 == 0
  const size_t total = (size_t)Cout * Cin * taps;
  const int K = Cin * taps;
  const int per_o = taps * 16;                 // float4 groups of this block (<= 144)
  if (tid < per_o) {
    const int tp = tid >> 4, i4 = (tid & 15) * 4;
    f32x4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i4 < ni) {
      const float* src = slab + (size_t)o * K + (size_t)tp * Cin + i0 + i4;
      int sp = 0;
      for (; sp + 8 <= splits; sp += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += *reinterpret_cast<const f32x4*>(src + (size_t)(sp + u) * total);
      }
      for (; sp < splits; ++sp) acc[0] += *reinterpret_cast<const f32x4*>(src + (size_t)sp * total);
    }
    const f32x4 s4 = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[(i4 + q) * taps + tp] = s4[q];
  }
  __syncthreads();
  const int run = ni * taps;                   // contiguous floats of this output channel
  float* const dst = dw + ((size_t)o * Cin + i0) * taps;
  for (int r = tid; r < run; r += 192) dst[r] = accumulate ? dst[r] + tile[r] : tile[r];
}

// taps == 1: packed and reference layouts coincide -> flat, fully coalesced reduction (grid-stride); the first
// `bblocks` blocks reduce the bias partials
__global__ __launch_bounds__(256) void wgrad_reduce_flat_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                const float* __restrict__ bslab,
                                                                float* __restrict__ dbias, int splits, int Cout,
                                                                size_t total, int accumulate, int wblocks, int bblocks) {
  if ((int)blockIdx.x < bblocks) {
    __shared__ float part[256];
    wgrad_bias_reduce<256>(bslab, dbias, splits, Cout, accumulate, (int)blockIdx.x, part);
    return;
  }
  const size_t n4 = total / 4;   // Cin % 4 == 0 -> total % 4 == 0
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
  f32x4* d4 = reinterpret_cast<f32x4*>(dw);
  for (size_t i = (size_t)(blockIdx.x - bblocks) * 256 + threadIdx.x; i < n4; i += (size_t)wblocks * 256) {
    f32x4 a0 = s4[i], a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1, a3 = a1;
    int sp = 1;
    for (; sp + 4 <= splits; sp += 4) {   // four independent loads in flight per lane
      a0 += s4[(size_t)sp * n4 + i];
      a1 += s4[(size_t)(sp + 1) * n4 + i];
      a2 += s4[(size_t)(sp + 2) * n4 + i];
      a3 += s4[(size_t)(sp + 3) * n4 + i];
    }
    for (; sp < splits; ++sp) a0 += s4[(size_t)sp * n4 + i];
    const f32x4 a = (a0 + a1) + (a2 + a3);
    d4[i] = accumulate ? d4[i] + a : a;
  }
}

// Weight packing from the reference layout (OIHW fp32, unet.py state_dict) to the
// two kernel layouts:  fwd[o][t][i] and dgrad[i][t'][o] with t' = taps-1-t (flip).
// Cin_pad >= Cin lets the 3-channel stem be zero-padded to a chunk multiple.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                   int Cout, int Cin, int taps, int Cin_pad, int Cout_pad, int kbf, int kbd) {
  // wf[o][kpos], kpos over (tap, cin) in tap-major (kbf == 0) or channel-block-major (kbf = block) order;
  // wd[i][kpos], kpos over (flipped tap, cout) likewise with kbd.
  const size_t Kf = (size_t)taps * Cin_pad, Kd = (size_t)taps * Cout_pad;
  const size_t total_f = (size_t)Cout * Kf;
  const size_t total_d = wd ? (size_t)Cin * Kd : 0;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_d;
       idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < total_f) {
      const int o = (int)(idx / Kf);
      const int kpos = (int)(idx - (size_t)o * Kf);
      int tp, i;
      if (kbf) { const int cb = kpos / (taps * kbf), r = kpos - cb * taps * kbf; tp = r / kbf; i = cb * kbf + (r - tp * kbf); }
      else { tp = kpos / Cin_pad; i = kpos - tp * Cin_pad; }
      wf[idx] = from_f32<T>(i < Cin ? w[((size_t)o * Cin + i) * taps + tp] : 0.f);
    } else {
      const size_t d = idx - total_f;
      const int i = (int)(d / Kd);
      const int kpos = (int)(d - (size_t)i * Kd);
      int tp, o;
      if (kbd) { const int ob = kpos / (taps * kbd), r = kpos - ob * taps * kbd; tp = r / kbd; o = ob * kbd + (r - tp * kbd); }
      else { tp = kpos / Cout_pad; o = kpos - tp * Cout_pad; }
      wd[d] = from_f32<T>(o < Cout ? w[((size_t)o * Cin + i) * taps + (taps - 1 - tp)] : 0.f);
    }
  }
}

// Tiled pack (Cout % 32 == 0, Cin % 32 == 0, no padding): one block moves a [32 o][32 i][taps] brick.  The reference
// layout is read in contiguous runs of 32 * taps floats per output channel, staged in LDS, and written as 64-byte
// runs of 32 consecutive channels in both kernel layouts -- the element-wise kernel above reads the dgrad side with
// a stride of Cin * taps floats (one cache line per element).
template <typename T, int TAPS>
__global__ __launch_bounds__(256) void pack_weight_tiled_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                                                T* __restrict__ wd, int Cout, int Cin, int kbf,
                                                                int kbd) {
  __shared__ float tile[32][32 * TAPS + 1];
  const int ib = blockIdx.x * 32, ob = blockIdx.y * 32;
  const int tid = threadIdx.x;
  for (int e = tid; e < 32 * 32 * TAPS; e += 256) {
    const int ol = e / (32 * TAPS), r = e - ol * (32 * TAPS);
    tile[ol][r] = w[((size_t)(ob + ol) * Cin + ib) * TAPS + r];   // r = il * TAPS + tap
  }
  __syncthreads();
  const size_t Kf = (size_t)TAPS * Cin, Kd = (size_t)TAPS * Cout;
  // forward pack: for each (o, tap) 32 consecutive i
  for (int e = tid; e < 32 * TAPS * 32; e += 256) {
    const int il = e & 31, rest = e >> 5;
    const int tp = rest % TAPS, ol = rest / TAPS;
    const int i = ib + il;
    const size_t kpos = kbf ? (size_t)(i / kbf) * TAPS * kbf + (size_t)tp * kbf + (i % kbf) : (size_t)tp * Cin + i;
    wf[(size_t)(ob + ol) * Kf + kpos] = from_f32<T>(tile[ol][il * TAPS + tp]);
  }
  if (wd) {
    // dgrad pack: for each (i, flipped tap) 32 consecutive o
    for (int e = tid; e < 32 * TAPS * 32; e += 256) {
      const int ol = e & 31, rest = e >> 5;
      const int tp = rest % TAPS, il = rest / TAPS;
      const int o = ob + ol;
      const size_t kpos = kbd ? (size_t)(o / kbd) * TAPS * kbd + (size_t)tp * kbd + (o % kbd) : (size_t)tp * Cout + o;
      wd[(size_t)(ib + il) * Kd + kpos] = from_f32<T>(tile[ol][il * TAPS + (TAPS - 1 - tp)]);
    }
  }
}

// All packed weights of a model in ONE launch (after an optimizer step every kernel-layout copy is stale: 227 launches
// of ~7 us each for UNet-64 otherwise).  A descriptor per weight; block b finds its weight by binary search over the
// prefix sums of the per-weight brick counts and then moves one [32 o][32 i][taps] brick exactly like the kernel above.
struct PackDesc {
  const float* w;   // reference layout (Cout, Cin, k, k)
  void* wf;         // forward pack  [Cout][taps][Cin]  (block-major when kbf != 0)
  void* wd;         // dgrad pack    [Cin][taps flipped][Cout], or null
  int Cout, Cin, taps, kbf, kbd;
  int first_block;  // prefix sum of (Cout / 32) * (Cin / 32)
};

template <typename T, int TAPS>
__device__ __forceinline__ void pack_brick(const PackDesc& d, int brick, float (*tile)[32 * 9 + 1]) {
  // 16-byte global accesses on both sides: float4 loads of the brick's 32 rows (32 * TAPS contiguous floats each),
  // V = 16 / sizeof(T) packed elements per store (runs of 32 output elements are contiguous in both packs: channel
  // bricks are 32-aligned and the k-blocks are multiples of 32)
  constexpr int V = 16 / (int)sizeof(T), NV = 32 / V;
  const int ibn = d.Cin / 32;
  const int ib = (brick % ibn) * 32, ob = (brick / ibn) * 32;
  const int tid = threadIdx.x;
  const int Cin = d.Cin, Cout = d.Cout;
  for (int e = tid; e < 32 * 8 * TAPS; e += 256) {
    const int ol = e / (8 * TAPS), q = e - ol * (8 * TAPS);
    const f32x4 v = *reinterpret_cast<const f32x4*>(d.w + ((size_t)(ob + ol) * Cin + ib) * TAPS + q * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[ol][q * 4 + j] = v[j];
  }
  __syncthreads();
  const size_t Kf = (size_t)TAPS * Cin, Kd = (size_t)TAPS * Cout;
  T* wf = reinterpret_cast<T*>(d.wf);
  T* wd = reinterpret_cast<T*>(d.wd);
  for (int e = tid; e < 32 * TAPS * NV; e += 256) {
    const int iv = e % NV, rest = e / NV;
    const int tp = rest % TAPS, ol = rest / TAPS;
    const int i = ib + iv * V;
    const size_t kpos = d.kbf ? (size_t)(i / d.kbf) * TAPS * d.kbf + (size_t)tp * d.kbf + (i % d.kbf) : (size_t)tp * Cin + i;
    T out[V];
#pragma unroll
    for (int j = 0; j < V; ++j) out[j] = from_f32<T>(tile[ol][(iv * V + j) * TAPS + tp]);
    *reinterpret_cast<uint4*>(wf + (size_t)(ob + ol) * Kf + kpos) = *reinterpret_cast<const uint4*>(out);
  }
  if (wd) {
    for (int e = tid; e < 32 * TAPS * NV; e += 256) {
      const int ov = e % NV, rest = e / NV;
      const int tp = rest % TAPS, il = rest / TAPS;
      const int o = ob + ov * V;
      const size_t kpos = d.kbd ? (size_t)(o / d.kbd) * TAPS * d.kbd + (size_t)tp * d.kbd + (o % d.kbd) : (size_t)tp * Cout + o;
      T out[V];
#pragma unroll
      for (int j = 0; j < V; ++j) out[j] = from_f32<T>(tile[ov * V + j][il * TAPS + (TAPS - 1 - tp)]);
      *reinterpret_cast<uint4*>(wd + (size_t)(ib + il) * Kd + kpos) = *reinterpret_cast<const uint4*>(out);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackDesc* __restrict__ table, int n) {
  __shared__ float tile[32][32 * 9 + 1];
  int lo = 0, hi = n - 1;
  const int b = blockIdx.x;
  while (lo < hi) {   // last descriptor whose first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = table[lo];
  if (d.taps == 9) pack_brick<T, 9>(d, b - d.first_block, tile);
  else pack_brick<T, 1>(d, b - d.first_block, tile);
}

// Column sums: out[c] = sum_m x[m, c]  (bias gradients).  Two deterministic stages: a
// (column group) x (row slab) grid of partial sums, then a per-channel sum over the slabs.
// A block is 8 chunk columns (128 bytes of a row) x 32 row lanes.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                             int M, int C, int rows_per_slab) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[32][8][8];
  const int tid = threadIdx.x, tc = tid & 7, tr = tid >> 3;
  const int chunk = blockIdx.x * 8 + tc;
  const bool cok = chunk * EPV < C;
  const int m_begin = blockIdx.y * rows_per_slab, m_end = min(M, m_begin + rows_per_slab);
  float s[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = 0.f;
  if (cok) {
    for (int m = m_begin + tr; m < m_end; m += 32) {
      Chunk<T> ch;
      ch.load(x + (size_t)m * C + (size_t)chunk * EPV);
#pragma unroll
      for (int e = 0; e < EPV; ++e) s[e] += ch.v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) red[tr][tc][e] = s[e];
  __syncthreads();
  if (tid < 8 * EPV) {
    const int c = tid / EPV, e = tid - c * EPV;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) a += red[r][c][e];
    const int ch = (blockIdx.x * 8 + c) * EPV + e;
    if (ch < C) part[(size_t)blockIdx.y * C + ch] = a;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nslabs, int C,
                                    int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nslabs; ++b) s += part[(size_t)b * C + c];
  out[c] = accumulate ? out[c] + s : s;
}

}  // namespace mdm

using namespace mdm;

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// name (as rocprofv3 prints it, without the argument list) of the GEMM-class kernel the calling thread launched last:
// lets bench.py label its per-launch HIP-event timings with the kernel that actually ran
// ---- split-K second stage: y = epilogue(sum_s part[s] + bias) --------------------------------------------------
// Same arithmetic as conv_epilogue: the sum (+ bias) is rounded to T first, the activation / pre-activation store /
// residual add see the rounded value.  One thread per 4 consecutive channels of one output row.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part, int ksplit, size_t mc,
                                                              int Cout, const float* __restrict__ bias,
                                                              const T* __restrict__ res, const T* __restrict__ aux,
                                                              T* __restrict__ y, T* __restrict__ ypre, int act) {
  const size_t idx = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (idx >= mc) return;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + idx);
  for (int sp = 1; sp < ksplit; ++sp) s += *reinterpret_cast<const f32x4*>(part + (size_t)sp * mc + idx);
  const int n = (int)(idx % (size_t)Cout);
  if (bias) s += *reinterpret_cast<const f32x4*>(bias + n);
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = to_f32(from_f32<T>(s[e]));
  if (act == 1) {
    if (ypre) {
      if constexpr (sizeof(T) == 2 && kFfnAuxByte) {   // one byte per element: the code of gelu'(pre) (common.hpp DGeluCode)
        *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(ypre) + idx) = DGeluCode::enc4(v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) ypre[idx + e] = from_f32<T>(v[e]);
      }
    }
    gelu_vec<T, 4>(v);
  } else if (act == 2) {
    if constexpr (sizeof(T) == 2 && kFfnAuxByte) {
      float g[4];
      DGeluCode::dec4(*reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(aux) + idx), g);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= g[e];
    } else {
      float ax[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) ax[e] = to_f32(aux[idx + e]);
      mul_dgelu_vec<T, 4>(v, ax);
    }
  }
  if (res) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += to_f32(res[idx + e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) y[idx + e] = from_f32<T>(v[e]);
}

// ---- sub-pixel form of upsample2x -> conv3x3: weight packs and the weight-gradient fold ----------------------------
// Row taps of the 3x3 kernel that land on low-resolution offset index a (0, 1) of output phase ph (0, 1):
//   ph = 0: a = 0 (row b - 1) <- {kh 0},  a = 1 (row b) <- {kh 1, 2};   ph = 1: a = 0 (row b) <- {kh 0, 1},  a = 1 (row b + 1) <- {kh 2}
__device__ __forceinline__ bool up_tap_in(int ph, int a, int kh) {
  return ph == 0 ? (a == 0 ? kh == 0 : kh >= 1) : (a == 0 ? kh <= 1 : kh == 2);
}
// one thread per (co, ci), ci the fast index: w (Cout, Cin, 3, 3) fp32 -> w_ph [4 Cout][Cin / 64][4][64] (bf16)
__global__ __launch_bounds__(256) void upconv_pack_kernel(const float* __restrict__ w, bf16* __restrict__ w_ph,
                                                          int Cout, int Cin) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cout * Cin) return;
  const int co = idx / Cin, ci = idx - co * Cin;
  float t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = w[(size_t)idx * 9 + k];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float v = 0.f;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              if (up_tap_in(ph, a, kh) && up_tap_in(pw, b, kw)) v += t[kh * 3 + kw];
          const int phase = ph * 2 + pw;
          // forward: row (phase, co), k = (ci / 64, tap j = 2a + b, ci % 64)
          w_ph[((size_t)(phase * Cout + co) * (Cin >> 6) + (ci >> 6)) * 256 + (a * 2 + b) * 64 + (ci & 63)] = (bf16)v;
        }
}
// ... and its input-gradient pack w_t [Cin][4 phases][Cout / 64][4][64]: row ci, k = (phase, co / 64, tap j = 2 dj + dk with
// dj = 1 - a, dk = 1 - b, co % 64) -- co is the fast index there, so it is written from a [64 co][16 ci][9] brick staged in
// LDS (64 runs of 256 contiguous bf16 per brick), like s2dgrad_pack_kernel below.  (As part of the kernel above, whose
// threads run along ci, these were scattered two-byte stores: 54-58 us per weight.)
__global__ __launch_bounds__(256) void upconv_pack_t_kernel(const float* __restrict__ w, bf16* __restrict__ w_t, int Cout,
                                                            int Cin) {
  __shared__ float t[64][16 * 9 + 1];
  const int cib = Cin / 16;
  const int co0 = ((int)blockIdx.x / cib) * 64, ci0 = ((int)blockIdx.x % cib) * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 144; e += 256) {
    const int co_l = e / 144, r = e - co_l * 144;
    t[co_l][r] = w[((size_t)(co0 + co_l) * Cin + ci0) * 9 + r];
  }
  __syncthreads();
  for (int e = tid; e < 4 * 16 * 256; e += 256) {
    const int co_l = e & 63, j = (e >> 6) & 3, ci_l = (e >> 8) & 15, phase = e >> 12;
    const int ph = phase >> 1, pw = phase & 1, a = 1 - (j >> 1), b = 1 - (j & 1);
    float v = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        if (up_tap_in(ph, a, kh) && up_tap_in(pw, b, kw)) v += t[co_l][ci_l * 9 + kh * 3 + kw];
    w_t[(((size_t)(ci0 + ci_l) * 4 + phase) * (Cout >> 6) + (co0 >> 6)) * 256 + j * 64 + co_l] = (bf16)v;
  }
}

// w (Cout, Cin, 3, 3) fp32 -> the input-gradient pack of the stride-2 convolution,
// w_sel [(ph, pw, ci)][Cout / 64][tap j = 2 dh + dw][64] bf16: dx[2b + p] += dy[b + d] * W[k(p, d)] per axis with
// k(0, 0) = 1, k(1, 0) = 2, k(1, 1) = 0 and no tap for (p, d) = (0, 1) (zero block).
// A block moves a [64 co][16 ci][9] brick through LDS: read as 64 runs of 144 contiguous floats, written as 64 runs of
// 256 contiguous bf16 (one per (phase, ci)).  (Rounds 3-5: one thread per (co, ci) with ci the fast index -- every one of
// its 16 two-byte stores went to a different 512-byte row: 108-134 us for a 256 x 256 weight, twice per train step.)
__global__ __launch_bounds__(256) void s2dgrad_pack_kernel(const float* __restrict__ w, bf16* __restrict__ w_sel, int Cout,
                                                           int Cin) {
  __shared__ float t[64][16 * 9 + 1];
  const int cib = Cin / 16;
  const int co0 = ((int)blockIdx.x / cib) * 64, ci0 = ((int)blockIdx.x % cib) * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 144; e += 256) {
    const int co_l = e / 144, r = e - co_l * 144;
    t[co_l][r] = w[((size_t)(co0 + co_l) * Cin + ci0) * 9 + r];
  }
  __syncthreads();
  for (int e = tid; e < 4 * 16 * 256; e += 256) {
    const int co_l = e & 63, j = (e >> 6) & 3, ci_l = (e >> 8) & 15, phase = e >> 12;
    const int ph = phase >> 1, pw = phase & 1, dh = j >> 1, dw = j & 1;
    const int kh = ph == 0 ? (dh == 0 ? 1 : -1) : (dh == 0 ? 2 : 0);
    const int kw = pw == 0 ? (dw == 0 ? 1 : -1) : (dw == 0 ? 2 : 0);
    const float v = (kh >= 0 && kw >= 0) ? t[co_l][ci_l * 9 + kh * 3 + kw] : 0.f;
    w_sel[(((size_t)(phase * Cin + ci0 + ci_l) * (Cout >> 6) + (co0 >> 6)) * 4 + j) * 64 + co_l] = (bf16)v;
  }
}

// dW (Cout, Cin, 3, 3) (+)= the fold of dwb (4 Cout, Cin, 3, 3): the gradient of phase weight (ph, a) x (pw, b) -- stored at
// low-resolution tap (ph + a, pw + b) of row (phase, co) -- flows to every 3x3 tap summed into it.  Only the 16 computed
// (phase, tap) blocks are read.  One thread per (co, ci).
__global__ __launch_bounds__(256) void upconv_wfold_kernel(const float* __restrict__ dwb, float* __restrict__ dw, int Cout,
                                                           int Cin, int accumulate) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cout * Cin) return;
  const int co = idx / Cin, ci = idx - co * Cin;
  float o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = 0.f;
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float g = dwb[((size_t)((ph * 2 + pw) * Cout + co) * Cin + ci) * 9 + (ph + a) * 3 + (pw + b)];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
              if (up_tap_in(ph, a, kh) && up_tap_in(pw, b, kw)) o[kh * 3 + kw] += g;
        }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float* d = dw + (size_t)idx * 9 + k;
    *d = accumulate ? *d + o[k] : o[k];
  }
}

__global__ __launch_bounds__(256) void upconv_bfold_kernel(const float* __restrict__ db4, float* __restrict__ db, int Cout,
                                                           int accumulate) {
  const int co = blockIdx.x * 256 + threadIdx.x;
  if (co >= Cout) return;
  const float v = (db4[co] + db4[Cout + co]) + (db4[2 * Cout + co] + db4[3 * Cout + co]);
  db[co] = accumulate ? db[co] + v : v;
}

static int g_no_deep_pipe = 0;      // development knob 11: 1 = no 4-stage instantiations for under-filled grids
static int g_one_tile_blocks = 0;   // development knob 5
static int g_dev_flags = 0;         // development knobs 0 / 1 -> ConvArgs::dev_flags (bit 1 / bit 0) of every GEMM launch
static inline ConvArgs with_dev_flags(const ConvArgs& a) { ConvArgs b = a; b.dev_flags = g_dev_flags; return b; }
static thread_local char g_last_gemm[96] = "";
extern "C" const char* mdm_last_gemm_kernel(void) { return g_last_gemm; }
#define MDM_NOTE_KERNEL(...) snprintf(g_last_gemm, sizeof(g_last_gemm), __VA_ARGS__)
template <typename T, int BM, int BN, int WM, int WN, int MODE, int SPLIT = 0>
static int launch_conv_cfg(const ConvArgs& a_, hipStream_t st) {
  const ConvArgs a = with_dev_flags(a_);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_kernel<T, BM, BN, WM, WN, MODE, SPLIT>;
  ensure_dynamic_lds(kern, smem);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), smem, st, a);
  MDM_NOTE_KERNEL("conv_gemm_kernel<%s, %d, %d, %d, %d, %d>", sizeof(T) == 2 ? "bf16" : (SPLIT == 2 ? "float (bf16x3, weight planes)" : (SPLIT ? "float (bf16x3)" : "float")), BM, BN, WM, WN, MODE);
  MDM_LAUNCH_STATUS();
}

template <int BM, int BN, int WM, int WN, int MODE>
static int launch_conv_bl(const ConvArgs& a_, hipStream_t st) {
  const ConvArgs a = with_dev_flags(a_);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE>;
  ensure_dynamic_lds(kern, smem);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  if constexpr (BM == 128 && BN == 128) {
    // under-filled grid (sampling at batch 1-4): one block per CU, four LDS stages (see the kernel's NSTG note)
    if (tiles <= device_cus() && !g_no_deep_pipe) {
      auto kern4 = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE, false, false, false, false, 4>;
      ensure_dynamic_lds(kern4, 2 * smem);
      hipLaunchKernelGGL(kern4, dim3(tiles), dim3(WM * WN * 64), 2 * smem, st, a, NoConvGroup{});
      MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d, 4 stages>", BM, BN, WM, WN, MODE);
      MDM_LAUNCH_STATUS();
    }
  }
  const int resident = device_cus() * (WM * WN == 4 ? 2 : 1);   // persistent blocks: one (8 waves) or two (4 waves) per CU
  // development knob 5: 1 = one block per output tile (a block's stores drain while its successor on the CU starts), 0 =
  // persistent blocks (the next tile's first DMA is issued from inside the epilogue)
  const int grid = (g_one_tile_blocks || tiles < resident) ? tiles : resident;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), smem, st, a, NoConvGroup{});
  MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d>", BM, BN, WM, WN, MODE);
  MDM_LAUNCH_STATUS();
}

template <int MODE>
static int launch_conv_bl_gn(const ConvArgs& a_, hipStream_t st) {
  const ConvArgs a = with_dev_flags(a_);
  constexpr int BM = 256, BN = 192, WM = 2, WN = 4;
  constexpr int smem = 2 * (BM + BN) * 128 + 4096;   // + the epilogue's statistics scratch
  auto kern = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE, false, false, false, true>;
  ensure_dynamic_lds(kern, smem);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  const int resident = device_cus();
  hipLaunchKernelGGL(kern, dim3(tiles < resident ? tiles : resident), dim3(WM * WN * 64), smem, st, a, NoConvGroup{});
  MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d, +gn>", BM, BN, WM, WN, MODE);
  MDM_LAUNCH_STATUS();
}

template <int BM, int BN, int WM, int WN>
static int launch_conv_bl_sel4(const ConvArgs& a_, hipStream_t st) {
  const ConvArgs a = with_dev_flags(a_);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE_3x3, false, false, true>;
  ensure_dynamic_lds(kern, smem);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  const int resident = device_cus() * (WM * WN == 4 ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3(tiles < resident ? tiles : resident), dim3(WM * WN * 64), smem, st, a, NoConvGroup{});
  MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d, sel4>", BM, BN, WM, WN, MODE_3x3);
  MDM_LAUNCH_STATUS();
}

template <int BM, int BN, int WM, int WN>
static int launch_conv_bl_grouped(const ConvArgs& a_, const ConvGroup& gr, hipStream_t st) {
  const ConvArgs a = with_dev_flags(a_);
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE_1x1, true>;
  ensure_dynamic_lds(kern, smem);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN) * a.groups;
  const int resident = device_cus() * (WM * WN == 4 ? 2 : 1);
  hipLaunchKernelGGL(kern, dim3(tiles < resident ? tiles : resident), dim3(WM * WN * 64), smem, st, a, gr);
  MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d, grouped>", BM, BN, WM, WN, MODE_1x1);
  MDM_LAUNCH_STATUS();
}

// buffer-addressed k-loop (conv_gemm_bl_kernel) usable for this problem?
template <typename T, int MODE>
static bool conv_bl_ok(const ConvArgs& a) {
  if (sizeof(T) != 2 || MODE == MODE_3x3_T2) return false;
  if (a.K % 64 != 0 || (MODE == MODE_3x3 && a.kblk == 0)) return false;
  const size_t lim = 0x7F000000u;
  const size_t bias = MODE == MODE_3x3 ? (size_t)(a.W + 1) * a.Cin * 2 : 0;
  if ((size_t)a.N * a.H * a.W * a.Cin * 2 + bias > lim || (size_t)a.Cout * a.K * 2 > lim) return false;
  return 2 * bias + (size_t)a.K * 2 < 0x00F00000u;   // INVALID + any tile offset stays below 2^31
}

// Split-K for problems that cannot fill the chip with output tiles (sampling at batch 1-4: M = 1024 at the 16x16 level
// gives 48 tiles of 128x128, each walking all 108 k-tiles of a 3x3 768 -> 768 convolution -- 73 us for 15 us of work).
// Splits: enough blocks for two per CU, at least 6 k-tiles each, at most 16, and only when the split saves at least 16
// k-tiles of serial walk (~10 us) -- the second launch costs about that much.  1 = do not split.
// Considered whenever the tiles fill less than `fill` percent of the 2-per-CU slots: 80 also catches the training shapes
// of the nested model's inner U-Net at batch 16 (M = 4096: 192 tiles of 128x128 at N = 768), 25 was the sampling-only rule.
static int g_split_fill = 80;   // development knob 6
static int g_no_direct = 0;     // development knob 7: 1 = narrow 3x3 convolutions back on the implicit-GEMM kernel
static int g_no_wgrad_direct = 0;   // development knob 8 (mdm_dev_set_knob): 1 = no wgrad_direct_kernel (split GEMM for the narrow weight gradients too)
static int g_split_minkt = 6, g_split_minsave = 16;   // development knobs 9, 10
static int g_split_per_cu = 0;                        // development knob 12: blocks per CU a split aims for (0 = by M, below)
static int conv_ksplit(int M, int Cout, int K, int dtype) {
  if (dtype != DT_BF16 || K % 64 != 0 || Cout % 8 != 0 || Cout <= 64) return 1;
  const int fill = g_split_fill;   // development knob 6 (mdm_dev_set_knob), default 80
  const long tiles = (long)((M + 127) / 128) * ((Cout + 127) / 128);
  const int nt = K / 64, cus = device_cus();
  if (tiles * 100 > (long)fill * 2 * cus || nt < 2 * g_split_minkt) return 1;
  // blocks per CU a split aims for: two 2-stage blocks for the training shapes (M = 4096 of the nested models' inner U-Net),
  // ONE 4-stage block for the sampling shapes (M <= 2048: 5.98 -> 5.84 ms per graphed UNet-64 iteration at batch 4,
  // 20.8 -> 20.4 nested-1024; the nested-256 train step is indifferent, 59.4 vs 59.4)
  const int per_cu = g_split_per_cu ? g_split_per_cu : (M <= 2048 ? 1 : 2);
  long sp = ((long)per_cu * cus) / tiles;
  if (sp > nt / g_split_minkt) sp = nt / g_split_minkt;
  if (sp > 16) sp = 16;
  if (sp < 2 || nt - (nt + sp - 1) / sp < g_split_minsave) return 1;
  return (int)sp;
}

extern "C" int mdm_conv_fwd_plan(int M, int Cout, int K, int dtype, int* splits, size_t* ws_bytes) {
  MDM_CHECK_ARG(splits && ws_bytes && M > 0 && Cout > 0 && K > 0);
  *splits = conv_ksplit(M, Cout, K, dtype);
  *ws_bytes = *splits > 1 ? (size_t)*splits * M * Cout * sizeof(float) : 0;
  return 0;
}

template <int MODE>
static int launch_conv_bl_splitk(ConvArgs a, int splits, float* ws, hipStream_t st) {
  a.dev_flags = g_dev_flags;
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE, false, true>;
  ensure_dynamic_lds(kern, smem);
  const int nt = a.K / 64;
  a.kt_per = (nt + splits - 1) / splits;
  a.ksplit = (nt + a.kt_per - 1) / a.kt_per;   // every range non-empty
  a.part = ws;
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN) * a.ksplit;
  const int resident = device_cus() * 2;
  if (tiles <= device_cus() && !g_no_deep_pipe) {   // one block per CU: four LDS stages
    auto kern4 = conv_gemm_bl_kernel<BM, BN, WM, WN, MODE, false, true, false, false, 4>;
    ensure_dynamic_lds(kern4, 2 * smem);
    hipLaunchKernelGGL(kern4, dim3(tiles), dim3(WM * WN * 64), 2 * smem, st, a, NoConvGroup{});
  } else
  hipLaunchKernelGGL(kern, dim3(tiles < resident ? tiles : resident), dim3(WM * WN * 64), smem, st, a, NoConvGroup{});
  const size_t mc = (size_t)a.M * a.Cout;
  hipLaunchKernelGGL(splitk_epilogue_kernel<bf16>, dim3((unsigned)((mc / 4 + 255) / 256)), dim3(256), 0, st, ws, a.ksplit, mc,
                     a.Cout, a.bias, (const bf16*)a.res, (const bf16*)a.aux, (bf16*)a.y, (bf16*)a.ypre, a.act);
  MDM_NOTE_KERNEL("conv_gemm_bl_kernel<%d, %d, %d, %d, %d, splitk x%d>", BM, BN, WM, WN, MODE, a.ksplit);
  MDM_LAUNCH_STATUS();
}

// Block tile (BM * 1000 + BN) of the forward / dgrad kernel for a problem.  bf16: the candidates are 128x128
// (2 blocks / CU), 256x192 and 256x256 (1 block / CU); the cheapest by rounds x tile area / relative efficiency wins
// (relative efficiencies measured with tools/kbench.py: the larger tiles move fewer LDS bytes per FLOP; N = 768 --
// 130 GEMMs of a step -- is exactly one round of 256x192 tiles at M = 16384).
static int g_force_tile = 0;   // development knob 2 (mdm_dev_set_knob): 128128 / 256192 / 256256, 0 = cost model
static int conv_tile_code(int M, int Cout, int dtype) {
  if (Cout <= 32) return 128032;
  if (Cout <= 64) return 128064;
  if (dtype != DT_BF16) return 128128;
  if (g_force_tile) return g_force_tile;
  const long mt128 = (M + 127) / 128, mt256 = (M + 255) / 256;
  const long t128 = mt128 * ((Cout + 127) / 128), t192 = mt256 * ((Cout + 191) / 192), t256 = mt256 * ((Cout + 255) / 256);
  const double c128 = (double)((t128 + 511) / 512) * 2.0 * 1.0 / 0.80;
  const double c192 = (double)((t192 + 255) / 256) * 3.0 / 0.95;
  const double c256 = (double)((t256 + 255) / 256) * 4.0 / 1.00;
  if (c256 <= c192 && c256 <= c128) return 256256;
  if (c192 <= c128) return 256192;
  return 128128;
}

extern "C" int mdm_conv_fwd_tile(int M, int Cout, int dtype) { return conv_tile_code(M, Cout, dtype); }

// ---- direct 3x3 convolution for the narrow outer levels of the nested models (32 / 64 channels at 256^2 ... 1024^2) ----
// There the implicit GEMM is the wrong shape: N = 32 fills a quarter of an MFMA-friendly tile, and every tap re-reads its
// shifted copy of the input through L2 -- 280 us for a 1024^2 x 32 -> 32 layer at batch 4 whose HBM bound (read x once,
// write y once) is 117 us.  This kernel is built around that bound instead:
//   * a block owns a TH x TW pixel tile and stages the (TH + 2) x (TW + 2) halo of x in LDS ONCE; the nine taps are nine
//     shifted views of it (zeros outside the image are written during staging);
//   * the operands are swapped: A = the weights (16 output channels x 32 reduction elements per MFMA, read ONCE per block
//     into registers -- the whole filter of the wave's output channels: 72-144 VGPRs -- and kept over a persistent tile loop),
//     B = 16 pixels of the staged tile; D[cout][pixel] then leaves every lane with 4 consecutive output channels of one
//     pixel, which go to HBM as 8-byte stores without a trip through LDS;
//   * w is the forward pack [Cout][tap][Cin] (for the input gradient: the dgrad pack with the roles of the channel counts
//     swapped), bias / residual as in the GEMM epilogue.
struct DirectArgs {
  const bf16* x; const bf16* w; const float* bias; const bf16* res; bf16* y;
  int N, H, W, tiles_x, tiles_y, tiles;
};

template <int CIN, int COUT, int TH, int TW, int WSPLIT>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(DirectArgs p) {
  constexpr int PITCH = CIN * 2 + (CIN == 64 ? 16 : 0);        // bytes per staged pixel (+16: 128-byte rows would 2-way conflict)
  constexpr int TWH = TW + 2, THH = TH + 2;
  constexpr int CHUNKS = CIN / 8;                              // 16-byte chunks per pixel
  constexpr int MB = COUT / WSPLIT / 16;                       // 16-channel output blocks per wave
  constexpr int KPT = CIN / 32;                                // MFMA reduction steps per tap
  constexpr int KS = 9 * KPT;
  constexpr int NB = TW / 16;                                  // 16-pixel blocks per tile row
  constexpr int RSETS = 4 / WSPLIT;                            // waves sharing the rows of a tile
  static_assert(TH % RSETS == 0, "rows per wave");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int cout0 = (wid / RSETS) * (COUT / WSPLIT);
  const int rset = wid % RSETS;

  // the wave's filter, MFMA A-fragment order: lane (l15, quad) holds w[cout0 + mb * 16 + l15][ks * 32 + quad * 8 .. + 8]
  Frag<bf16> wf[MB][KS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[mb][ks].v = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(cout0 + mb * 16 + l15) * (9 * CIN) + ks * 32 + quad * 8);
  float bv[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[mb][i] = p.bias ? p.bias[cout0 + mb * 16 + quad * 4 + i] : 0.f;

  // The halo of a tile travels HBM -> registers -> LDS; the loads of tile t + 1 are issued before the MFMAs of tile t and
  // land while it computes and stores (one memory round trip per tile, hidden, instead of three exposed ones).
  // Staging map: thread `tid` owns the 16-byte chunk column `tid` of every staged row (a staged row is TWH * CHUNKS = 264 /
  // 272 chunks, contiguous in HBM), the few columns past 256 are spread over the first threads -- so the addresses are
  // affine in the row (one pointer + a row stride), the row test is scalar, and nothing per load has to be kept in VGPRs.
  constexpr int COLS = TWH * CHUNKS, REM = COLS - 256;
  static_assert(REM > 0 && REM * THH <= 256, "staging map");
  constexpr int NLD = THH + 1;
  uint4 pre[NLD];
  const int mpx = tid / CHUNKS, mch = tid % CHUNKS;                                   // main column
  const int er = tid / REM, ecol = 256 + tid % REM, epx = ecol / CHUNKS, ech = ecol % CHUNKS;   // extra chunk (tid < REM * THH)
  const bool has_extra = tid < REM * THH;
  auto issue = [&](int tile) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int gx0 = tx * TW - 1, gy0 = ty * TH - 1;
    const bf16* xn = p.x + (size_t)n * p.H * p.W * CIN;
    const int gx = gx0 + mpx;
    const bool okx = gx >= 0 && gx < p.W;
    const bf16* col = xn + ((ptrdiff_t)gy0 * p.W + gx) * CIN + mch * 8;
#pragma unroll
    for (int r = 0; r < THH; ++r) {
      const int gy = gy0 + r;
      pre[r] = uint4{0u, 0u, 0u, 0u};
      if (okx && gy >= 0 && gy < p.H) pre[r] = *reinterpret_cast<const uint4*>(col + (ptrdiff_t)r * p.W * CIN);
    }
    pre[THH] = uint4{0u, 0u, 0u, 0u};
    if (has_extra) {
      const int gy = gy0 + er, gxe = gx0 + epx;
      if (gy >= 0 && gy < p.H && gxe >= 0 && gxe < p.W)
        pre[THH] = *reinterpret_cast<const uint4*>(xn + ((size_t)gy * p.W + gxe) * CIN + ech * 8);
    }
  };
  if ((int)blockIdx.x < p.tiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
#pragma unroll
    for (int r = 0; r < THH; ++r)
      *reinterpret_cast<uint4*>(dsm + (r * TWH + mpx) * PITCH + mch * 16) = pre[r];
    if (has_extra) *reinterpret_cast<uint4*>(dsm + (er * TWH + epx) * PITCH + ech * 16) = pre[THH];
    __syncthreads();
    if (tile + (int)gridDim.x < p.tiles) issue(tile + gridDim.x);
    auto do_row = [&](int r) {
      f32x4 acc[NB][MB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks / KPT, half = ks % KPT;
        const int dy = tap / 3, dx = tap % 3;
        const char* row = dsm + ((r + dy) * TWH + dx + l15) * PITCH + half * 64 + quad * 16;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          Frag<bf16> xb;
          xb.v = *reinterpret_cast<const bf16x8*>(row + nb * 16 * PITCH);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) mma16(acc[nb][mb], wf[mb][ks], xb);
        }
      }
      // The compiler (ROCm 7.2) reads the accumulators of the LAST MFMA back two instructions after issuing it when the
      // read sits behind a branch (the residual test): no wait states for the XDL write -> VALU read hazard, and the
      // first-read components came back stale in the <64, 64, .., 4> instantiation (found by the parity test: channels
      // 4 q + 2, 4 q + 3 of every second pixel block wrong).  Fence the k-loop from the epilogue and pay the wait states here.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // acc[nb][mb][i] = y[pixel nb * 16 + l15][cout0 + mb * 16 + quad * 4 + i]
      const int gy = ty * TH + r;
      const size_t rowbase = ((size_t)n * p.H + gy) * p.W + tx * TW;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const size_t off = (rowbase + nb * 16 + l15) * COUT + cout0 + mb * 16 + quad * 4;
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = acc[nb][mb][i] + bv[mb][i];
          if (p.res) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.res + off);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] += (float)rv[i];
          }
          bf16x4 ov;
#pragma unroll
          for (int i = 0; i < 4; ++i) ov[i] = (bf16)o[i];
          *reinterpret_cast<bf16x4*>(p.y + off) = ov;
        }
    };
#pragma unroll 1
    for (int r = rset; r < TH; r += RSETS) do_row(r);
    __syncthreads();   // the next tile's staging overwrites what the slower waves may still read
  }
}

template <int CIN, int COUT, int TH, int TW, int WSPLIT>
static int launch_conv_direct(const ConvArgs& a, hipStream_t st) {
  constexpr int pitch = CIN * 2 + (CIN == 64 ? 16 : 0);
  constexpr int smem = (TH + 2) * (TW + 2) * pitch;
  auto kern = conv3x3_direct_kernel<CIN, COUT, TH, TW, WSPLIT>;
  ensure_dynamic_lds(kern, smem);
  DirectArgs d;
  d.x = (const bf16*)a.x; d.w = (const bf16*)a.w; d.bias = a.bias; d.res = (const bf16*)a.res; d.y = (bf16*)a.y;
  d.N = a.N; d.H = a.H; d.W = a.W; d.tiles_x = a.W / TW; d.tiles_y = a.H / TH; d.tiles = d.tiles_x * d.tiles_y * a.N;
  const int per_cu = 160 * 1024 / smem >= 3 ? 3 : (160 * 1024 / smem >= 2 ? 2 : 1);   // 3 waves / SIMD by registers
  const int resident = device_cus() * per_cu * 2;                                       // two rounds of blocks: the second balances the tail
  hipLaunchKernelGGL(kern, dim3(d.tiles < resident ? d.tiles : resident), dim3(256), smem, st, d);
  MDM_NOTE_KERNEL("conv3x3_direct_kernel<%d, %d, %d, %d, %d>", CIN, COUT, TH, TW, WSPLIT);
  MDM_LAUNCH_STATUS();
}

// usable for this problem?  (bf16, 3x3 stride 1, plain epilogue, 32 / 64 channels both sides, tile-aligned image)
static bool conv_direct_ok(const ConvArgs& a, int ksize, int transposed, int dtype) {
  if (g_no_direct || dtype != DT_BF16 || ksize != 3 || transposed || a.stride != 1 || a.act != 0 || a.aux || a.ypre) return false;
  if (!(a.Cin == 32 || a.Cin == 64) || !(a.Cout == 32 || a.Cout == 64)) return false;
  if (a.kblk != 0 && !(a.kblk == 64 && a.Cin == 64)) return false;      // [Cout][tap][Cin] either way
  if (a.Ho != a.H || a.Wo != a.W || a.H % 8 != 0 || a.W % 64 != 0) return false;
  return (long)a.N * a.H * a.W >= 65536 && ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.w & 15) == 0;
}

// WSPLIT = groups of output channels among the 4 waves (each wave keeps the filter of 16-32 output channels in registers and
// walks 8 / (4 / WSPLIT) rows of the tile): chosen so that filter + accumulators + the prefetched halo stay under 256 VGPRs
// (2 blocks per CU)
static int launch_conv_direct_any(const ConvArgs& a, hipStream_t st) {
  if (a.Cin == 32 && a.Cout == 32) return launch_conv_direct<32, 32, 8, 64, 2>(a, st);
  if (a.Cin == 32 && a.Cout == 64) return launch_conv_direct<32, 64, 8, 64, 4>(a, st);
  if (a.Cin == 64 && a.Cout == 32) return launch_conv_direct<64, 32, 8, 32, 2>(a, st);
  return launch_conv_direct<64, 64, 8, 32, 4>(a, st);
}

template <typename T, int MODE, int SPLIT = 0>
static int launch_conv_mode(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 32) return launch_conv_cfg<T, 128, 32, 4, 1, MODE, SPLIT>(a, st);
  if (a.Cout <= 64) return launch_conv_cfg<T, 128, 64, 2, 2, MODE, SPLIT>(a, st);
  const int code = conv_tile_code(a.M, a.Cout, sizeof(T) == 2 ? DT_BF16 : DT_F32);
  if constexpr (sizeof(T) == 2) {
    if constexpr (MODE != MODE_3x3_T2) {
      if (conv_bl_ok<T, MODE>(a)) {
        if (code == 256256) return launch_conv_bl<256, 256, 2, 4, MODE>(a, st);
        if (code == 256192) return launch_conv_bl<256, 192, 2, 4, MODE>(a, st);
        return launch_conv_bl<128, 128, 2, 2, MODE>(a, st);
      }
    }
    // problems the buffer-addressed loader cannot express (ragged K, transposed stride-2 gradient, > 2 GiB operands)
    if (code != 128128) return launch_conv_cfg<T, 256, 256, 2, 4, MODE>(a, st);
  }
  // (weight planes with 4 x 1 waves -- a wave splits 2 activation fragments per k-step instead of 4, and reads 8 weight
  // fragments instead of 4 -- measured 88.3 against 88.7 ms per iteration: profiles/r06_did_not_pay.md)
  return launch_conv_cfg<T, 128, 128, 2, 2, MODE, SPLIT>(a, st);
}

template <typename T, int SPLIT = 0>
static int launch_conv_t(const ConvArgs& a, int ks, int transposed, hipStream_t st) {
  if (ks == 1) return launch_conv_mode<T, MODE_1x1, SPLIT>(a, st);
  if (transposed) return launch_conv_mode<T, MODE_3x3_T2, SPLIT>(a, st);
  return launch_conv_mode<T, MODE_3x3, SPLIT>(a, st);
}

// ws / ws_bytes: optional fp32 workspace sized by mdm_conv_fwd_plan; with it, problems too small to fill the chip with
// output tiles run split over the reduction (two launches: partial tiles, then sum + epilogue).  NULL = never split.
extern "C" int mdm_conv_fwd_ws(const void* x, const void* w_packed, const float* bias, const void* res,
                               const void* aux, void* y, void* y_pre, int N, int H, int W, int Cin, int Ho, int Wo,
                               int Cout, int ksize, int stride, int transposed, int act, int kblock, int dtype,
                               float* ws, size_t ws_bytes, void* stream) {
  MDM_CHECK_ARG(x && w_packed && y);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16 || dtype == DT_F32_SPLIT || dtype == DT_F32_SPLIT_W);
  // fp32 tensors, bf16x3 products (include/mdm_hip.h MDM_F32_SPLIT; _W: w_packed went through mdm_split_weight_planes)
  const int split_products = dtype == DT_F32_SPLIT ? 1 : (dtype == DT_F32_SPLIT_W ? 2 : 0);
  if (split_products) dtype = DT_F32;
  MDM_CHECK_ARG(split_products != 2 || (ksize * ksize * Cin) % 8 == 0);   // planes are runs of 8 along the reduction
  MDM_CHECK_ARG(act >= 0 && act <= 2);
  MDM_CHECK_ARG(act != 2 || aux);
  MDM_CHECK_ARG(!(act == 2 && res));   // one elementwise operand per launch (conv_epilogue keeps loads out of its store loop)
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(Cin % epv == 0);
  MDM_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cout > 0);
  if (ksize == 1) { MDM_CHECK_ARG(Ho == H && Wo == W && stride == 1 && !transposed); }
  else if (transposed) { MDM_CHECK_ARG(Ho == 2 * H && Wo == 2 * W); }
  else { MDM_CHECK_ARG(stride == 1 || stride == 2); MDM_CHECK_ARG(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1); }
  ConvArgs a = {};
  a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.aux = aux; a.y = y; a.ypre = y_pre;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride;
  a.M = N * Ho * Wo; a.K = ksize * ksize * Cin; a.act = act; a.groups = 0;
  const int bk = dtype == DT_F32 ? 32 : 64;
  MDM_CHECK_ARG(kblock == 0 || (ksize == 3 && kblock == bk && Cin % bk == 0));
  a.kblk = kblock;
  a.part = nullptr; a.ksplit = 1; a.kt_per = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (conv_direct_ok(a, ksize, transposed, dtype)) return launch_conv_direct_any(a, st);
  if (ws && dtype == DT_BF16 && !transposed && stride == 1) {
    const int sp = conv_ksplit(a.M, Cout, a.K, dtype);
    if (sp > 1 && ws_bytes >= (size_t)sp * a.M * Cout * sizeof(float)) {
      if (ksize == 1 && conv_bl_ok<bf16, MODE_1x1>(a)) return launch_conv_bl_splitk<MODE_1x1>(a, sp, ws, st);
      if (ksize == 3 && conv_bl_ok<bf16, MODE_3x3>(a)) return launch_conv_bl_splitk<MODE_3x3>(a, sp, ws, st);
    }
  }
  if (split_products == 2) return launch_conv_t<float, 2>(a, ksize, transposed, st);
  if (split_products) return launch_conv_t<float, 1>(a, ksize, transposed, st);
  return dtype == DT_F32 ? launch_conv_t<float>(a, ksize, transposed, st) : launch_conv_t<bf16>(a, ksize, transposed, st);
}

extern "C" int mdm_conv_fwd(const void* x, const void* w_packed, const float* bias, const void* res,
                            const void* aux, void* y, void* y_pre, int N, int H, int W, int Cin, int Ho, int Wo,
                            int Cout, int ksize, int stride, int transposed, int act, int kblock, int dtype,
                            void* stream) {
  return mdm_conv_fwd_ws(x, w_packed, bias, res, aux, y, y_pre, N, H, W, Cin, Ho, Wo, Cout, ksize, stride, transposed, act,
                         kblock, dtype, nullptr, 0, stream);
}


// n fp32 values of a packed weight (n % 8 == 0) -> the same bytes as hi / lo bf16 planes: run r of 8 values becomes
// [bf16 hi(v0..v7) | bf16 lo(v0..v7)], lo = bf16(v - float(hi)) -- what FragSplit::from_f32 computes in the k-loop
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w, uint4* __restrict__ out, size_t runs) {
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= runs) return;
  const f32x4 a = *reinterpret_cast<const f32x4*>(w + r * 8), b = *reinterpret_cast<const f32x4*>(w + r * 8 + 4);
  FragSplit f;
  f.from_f32(a, b);
  out[2 * r] = *reinterpret_cast<const uint4*>(&f.hi);
  out[2 * r + 1] = *reinterpret_cast<const uint4*>(&f.lo);
}
extern "C" int mdm_split_weight_planes(const float* w_packed, void* planes, size_t n, void* stream) {
  MDM_CHECK_ARG(w_packed && planes && w_packed != planes && n > 0 && n % 8 == 0);
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w_packed, (uint4*)planes, n / 8);
  MDM_LAUNCH_STATUS();
}

// w (Cout, Cin, 3, 3) fp32 -> the two bf16 packs of the sub-pixel upsample convolution (mdm_conv_up_fwd / _dgrad)
extern "C" int mdm_upconv_pack(const float* w_oihw, void* w_ph, void* w_t, int Cout, int Cin, void* stream) {
  MDM_CHECK_ARG(w_oihw && w_ph && w_t && Cout % 64 == 0 && Cin % 64 == 0);
  hipLaunchKernelGGL(upconv_pack_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w_oihw, (bf16*)w_ph, Cout, Cin);
  hipLaunchKernelGGL(upconv_pack_t_kernel, dim3((Cout / 64) * (Cin / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w_oihw, (bf16*)w_t, Cout, Cin);
  MDM_LAUNCH_STATUS();
}
// w (Cout, Cin, 3, 3) fp32 -> the bf16 pack mdm_conv_s2_dgrad reads
extern "C" int mdm_s2dgrad_pack(const float* w_oihw, void* w_sel, int Cout, int Cin, void* stream) {
  MDM_CHECK_ARG(w_oihw && w_sel && Cout % 64 == 0 && Cin > 0 && Cin % 16 == 0);
  hipLaunchKernelGGL(s2dgrad_pack_kernel, dim3((Cout / 64) * (Cin / 16)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     w_oihw, (bf16*)w_sel, Cout, Cin);
  MDM_LAUNCH_STATUS();
}
// dW (Cout, Cin, 3, 3) (+)= fold of dwb (4 Cout, Cin, 3, 3) (see mdm_conv_wgrad_blocked); dbias (Cout) (+)= the four
// phase sums of dbias4 (4 Cout) when both are given
extern "C" int mdm_upconv_wfold(const float* dwb, float* dw, const float* dbias4, float* dbias, int Cout, int Cin,
                                int accumulate, void* stream) {
  MDM_CHECK_ARG(dwb && dw && Cout > 0 && Cin > 0 && ((dbias4 == nullptr) == (dbias == nullptr)));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(upconv_wfold_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, st, dwb, dw, Cout, Cin, accumulate);
  if (dbias) hipLaunchKernelGGL(upconv_bfold_kernel, dim3((Cout + 255) / 256), dim3(256), 0, st, dbias4, dbias, Cout, accumulate);
  MDM_LAUNCH_STATUS();
}

// ---- sub-pixel forms of the resampling convolutions (bf16) -------------------------------------------------------
// A stride-2 3x3 convolution and a 3x3 convolution of a nearest-2x-upsampled image both touch, per 2x2 block of the
// high-resolution side, only 2x2 pixels of the low-resolution side per output phase: written over 2x2-blocked tensors
// they are 2x2 correlations, 16 (phase, offset) weight blocks instead of 36 (phase, tap) -- 2.25x fewer multiply-adds
// than the dense 3x3 over the high-resolution grid -- and the 4x larger upsampled tensor never exists.
// The launches below are conv_gemm_bl_kernel<.., SEL4> (4 taps per channel block); a column tile must lie inside one
// phase, so the per-phase width picks the tile.
static int sel4_tile(int cols_per_phase) {
  return cols_per_phase % 256 == 0 ? 256 : (cols_per_phase % 192 == 0 ? 192 : (cols_per_phase % 128 == 0 ? 128 : 0));
}
static int launch_sel4(const ConvArgs& a, int tile, hipStream_t st) {
  if (tile == 256) return launch_conv_bl_sel4<256, 256, 2, 4>(a, st);
  if (tile == 192) return launch_conv_bl_sel4<256, 192, 2, 4>(a, st);
  return launch_conv_bl_sel4<128, 128, 2, 2>(a, st);
}

// Input gradient of a stride-2 3x3 convolution (pad 1, even H and W): dx [N, 2 Ho, 2 Wo, Cin] from dy [N, Ho, Wo, Cout]:
//   dx[n, 2 bh + ph, 2 bw + pw, ci] = sum over (dh, dw) in {0, 1}^2, co of dy[n, bh + dh, bw + dw, co] * w_sel[(ph, pw, ci)][(dh, dw)][co]
// w_sel: [4 Cin][Cout / 64][4 taps][64] (channel-block-major, tap j = 2 dh + dw), zero where the (phase, offset) pair has
// no tap of the 3x3 kernel (7 of 16).  Cout % 64 == 0, Cin % 128 == 0.
extern "C" int mdm_conv_s2_dgrad_res(const void* dy, const void* w_sel, const void* res, void* dx, int N, int Ho, int Wo,
                                     int Cout, int Cin, int dtype, void* stream);
extern "C" int mdm_conv_s2_dgrad(const void* dy, const void* w_sel, void* dx, int N, int Ho, int Wo, int Cout, int Cin,
                                 int dtype, void* stream) {
  return mdm_conv_s2_dgrad_res(dy, w_sel, nullptr, dx, N, Ho, Wo, Cout, Cin, dtype, stream);
}
// ... + res [N, 2Ho, 2Wo, Cin] (or NULL): a second gradient of the same input -- the skip connection that taps the tensor a
// down-sampling convolution reads (unet.py:566-567, 883-897) -- added in the epilogue instead of by a separate pass
extern "C" int mdm_conv_s2_dgrad_res(const void* dy, const void* w_sel, const void* res, void* dx, int N, int Ho, int Wo,
                                     int Cout, int Cin, int dtype, void* stream) {
  MDM_CHECK_ARG(dy && w_sel && dx && dtype == DT_BF16);
  MDM_CHECK_ARG(N > 0 && Ho > 0 && Wo > 0 && Cout % 64 == 0 && Cin > 0);
  const int tile = sel4_tile(Cin);
  MDM_CHECK_ARG(tile != 0);
  ConvArgs a = {};
  a.x = dy; a.w = w_sel; a.y = dx; a.res = res;
  a.N = N; a.H = Ho; a.W = Wo; a.Cin = Cout; a.Ho = Ho; a.Wo = Wo; a.Cout = 4 * Cin; a.stride = 1;
  a.M = N * Ho * Wo; a.K = 4 * Cout; a.kblk = 64; a.ksplit = 1;
  a.sel_mode = 0; a.sel_base = 4;
  a.ps_cout = Cin; a.ps_H = Ho; a.ps_W = Wo;
  MDM_CHECK_ARG((conv_bl_ok<bf16, MODE_3x3>(a)));
  return launch_sel4(a, tile, reinterpret_cast<hipStream_t>(stream));
}

// y [N, 2H, 2W, Cout] = conv3x3(upsample2x_nearest(x), w) + bias from the LOW-resolution x [N, H, W, Cin]:
//   y[n, 2 bh + ph, 2 bw + pw, co] = bias[co] + sum over (dj, dk) in {0, 1}^2, ci of x[n, bh + ph - 1 + dj, bw + pw - 1 + dk, ci] * w_ph[(ph, pw, co)][(dj, dk)][ci]
// w_ph: [4 Cout][Cin / 64][4][64]: the 3x3 taps that fall on the same low-resolution pixel for this phase, summed.
// bias4: [4 Cout] (the bias repeated per phase) or NULL.  Cin % 64 == 0, Cout % 128 == 0.
extern "C" int mdm_conv_up_fwd(const void* x, const void* w_ph, const float* bias4, void* y, int N, int H, int W, int Cin,
                               int Cout, int dtype, void* stream) {
  MDM_CHECK_ARG(x && w_ph && y && dtype == DT_BF16);
  MDM_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin % 64 == 0 && Cout > 0);
  const int tile = sel4_tile(Cout);
  MDM_CHECK_ARG(tile != 0);
  ConvArgs a = {};
  a.x = x; a.w = w_ph; a.bias = bias4; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = 4 * Cout; a.stride = 1;
  a.M = N * H * W; a.K = 4 * Cin; a.kblk = 64; a.ksplit = 1;
  a.sel_mode = 1; a.sel_cout = Cout;
  a.ps_cout = Cout; a.ps_H = H; a.ps_W = W;
  MDM_CHECK_ARG((conv_bl_ok<bf16, MODE_3x3>(a)));
  return launch_sel4(a, tile, reinterpret_cast<hipStream_t>(stream));
}

// Input gradient of the same operation: dx [N, H, W, Cin] from the 2x2-BLOCKED output gradient dyb [N, H, W, 4 Cout]
// (channel (ph, pw, co) = dy[n, 2 bh + ph, 2 bw + pw, co]; mdm_space_to_depth2x makes it):
//   dx[n, bh, bw, ci] = sum over phases, (dj, dk), co of dyb[n, bh + dj - ph, bw + dk - pw, (ph, pw, co)] * w_t[ci][(ph, pw)][(dj, dk)][co]
// w_t: [Cin][4 phases][Cout / 64][4][64].  Cout % 64 == 0, Cin % 8 == 0.
extern "C" int mdm_conv_up_dgrad(const void* dyb, const void* w_t, void* dx, int N, int H, int W, int Cout, int Cin,
                                 int dtype, void* stream) {
  MDM_CHECK_ARG(dyb && w_t && dx && dtype == DT_BF16);
  MDM_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cout % 64 == 0 && Cin % 8 == 0);
  ConvArgs a = {};
  a.x = dyb; a.w = w_t; a.y = dx;
  a.N = N; a.H = H; a.W = W; a.Cin = 4 * Cout; a.Ho = H; a.Wo = W; a.Cout = Cin; a.stride = 1;
  a.M = N * H * W; a.K = 16 * Cout; a.kblk = 64; a.ksplit = 1;
  a.sel_mode = 2; a.sel_cout = Cout;
  MDM_CHECK_ARG((conv_bl_ok<bf16, MODE_3x3>(a)));
  const int code = conv_tile_code(a.M, a.Cout, DT_BF16);
  return launch_sel4(a, code == 256256 ? 256 : (code == 256192 ? 192 : 128), reinterpret_cast<hipStream_t>(stream));
}

// ---- convolution + GroupNorm of its output in one launch ------------------------------------------------------
// host-only: does (problem, norm) fit the fused epilogue?  bf16; 16x16 images (a 256-row tile = one sample); 24 channels
// per group and Cout a multiple of 192 (a 192-column tile = 8 whole groups); a problem the buffer-addressed loader takes
extern "C" int mdm_conv_fwd_gn_ok(int N, int H, int W, int Cin, int Cout, int ksize, int kblock, int groups, int dtype) {
  if (dtype != DT_BF16 || H * W != 256 || Cout % 192 != 0 || groups <= 0 || Cout != groups * 24 || Cin % 64 != 0) return 0;
  if (ksize != 1 && !(ksize == 3 && kblock == 64)) return 0;
  ConvArgs a = {};
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.K = ksize * ksize * Cin; a.kblk = kblock;
  return (ksize == 1 ? conv_bl_ok<bf16, MODE_1x1>(a) : conv_bl_ok<bf16, MODE_3x3>(a)) ? 1 : 0;
}

// y = conv(x, w_packed) + bias (+ res)  AND  y_norm = act(GroupNorm(y; gamma, beta, groups, eps)), stats [N][G][2],
// coef [N][Cout][2] (the outputs of mdm_gn_fwd on y) from ONE launch -- replaces nn.Conv2d followed by nn.GroupNorm
// (models/unet.py:310-311 proj_out -> ffn[0]; :312 -> the next layer's norm; :238 -> :300).  stride 1, no activation on y.
extern "C" int mdm_conv_fwd_gn(const void* x, const void* w_packed, const float* bias, const void* res, void* y, int N, int H,
                               int W, int Cin, int Cout, int ksize, int kblock, const float* gamma, const float* beta,
                               int groups, float eps, int gn_act, void* y_norm, float* stats, float* coef, int dtype,
                               void* stream) {
  MDM_CHECK_ARG(x && w_packed && y && gamma && beta && y_norm && stats && coef);
  MDM_CHECK_ARG(mdm_conv_fwd_gn_ok(N, H, W, Cin, Cout, ksize, kblock, groups, dtype));
  ConvArgs a = {};
  a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = H; a.Wo = W; a.Cout = Cout; a.stride = 1;
  a.M = N * H * W; a.K = ksize * ksize * Cin; a.kblk = kblock; a.ksplit = 1;
  a.gn_y = y_norm; a.gn_gamma = gamma; a.gn_beta = beta; a.gn_stats = stats; a.gn_coef = coef; a.gn_eps = eps;
  a.gn_act = gn_act; a.gn_groups = groups;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return ksize == 1 ? launch_conv_bl_gn<MODE_1x1>(a, st) : launch_conv_bl_gn<MODE_3x3>(a, st);
}

// y[g] [M, Cout] = x[g] [M, Cin] * w_packed[g]^T + bias[g] for `groups` (<= 32) linear layers of one shape, bf16, in ONE
// launch.  The pointer arrays are HOST arrays of device pointers; bias may be NULL (no bias at all) .  Passing the
// dgrad packs and the output gradients computes the input gradients of the same layers.  Cin % 64 == 0.
extern "C" int mdm_linear_grouped(const void* const* x, const void* const* w_packed, const float* const* bias,
                                  void* const* y, int groups, int M, int Cin, int Cout, int dtype, void* stream) {
  MDM_CHECK_ARG(x && w_packed && y && groups >= 1 && groups <= CG_MAXG);
  MDM_CHECK_ARG(dtype == DT_BF16 && Cin % 64 == 0 && Cout % 8 == 0 && M > 0);
  ConvArgs a = {};
  a.N = M; a.H = 1; a.W = 1; a.Cin = Cin; a.Ho = 1; a.Wo = 1; a.Cout = Cout; a.stride = 1;
  a.M = M; a.K = Cin; a.act = 0; a.kblk = 0; a.groups = groups;
  MDM_CHECK_ARG((size_t)M * Cin * 2 <= 0x7F000000u && (size_t)Cout * Cin * 2 <= 0x7F000000u);
  ConvGroup gr = {};
  for (int g = 0; g < groups; ++g) {
    MDM_CHECK_ARG(x[g] && w_packed[g] && y[g]);
    gr.x[g] = x[g]; gr.w[g] = w_packed[g]; gr.bias[g] = bias ? bias[g] : nullptr; gr.y[g] = y[g];
  }
  a.x = gr.x[0]; a.w = gr.w[0]; a.y = gr.y[0];
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // tile by the same cost model as a single problem of groups x M rows
  const int code = conv_tile_code(M * groups, Cout, DT_BF16);
  if (code == 256256) return launch_conv_bl_grouped<256, 256, 2, 4>(a, gr, st);
  if (code == 256192) return launch_conv_bl_grouped<256, 192, 2, 4>(a, gr, st);
  return launch_conv_bl_grouped<128, 128, 2, 2>(a, gr, st);
}

extern "C" int mdm_dev_ffn_aux_bytes(void) { return kFfnAuxByte ? 1 : 2; }

static int g_skip_wgrad_reduce = 0;   // development knob 13: timing-only ablation, the slab reduces are not launched (WRONG gradients)
extern "C" int mdm_dev_set_knob(int idx, int value) {
  MDM_CHECK_ARG(idx >= 0 && idx < 14);
  if (idx == 13) { g_skip_wgrad_reduce = value; return 0; }
  if (idx == 12) { g_split_per_cu = value > 0 ? value : 0; return 0; }
  if (idx == 11) { g_no_deep_pipe = value; return 0; }
  if (idx == 8) { g_no_wgrad_direct = value; return 0; }
  if (idx == 9) { g_split_minkt = value > 0 ? value : 6; return 0; }
  if (idx == 10) { g_split_minsave = value > 0 ? value : 16; return 0; }
  if (idx == 5) { g_one_tile_blocks = value; return 0; }
  if (idx == 3 || idx == 4) return 0;   // (were conv_gemm_x_kernel switches; the kernel was removed in round 6)
  if (idx == 6) { g_split_fill = value > 0 ? value : 80; return 0; }
  if (idx == 7) { g_no_direct = value; return 0; }
  if (idx == 2) { g_force_tile = value; return 0; }
  if (idx == 0) { g_dev_flags = (g_dev_flags & ~2) | ((value & 1) ? 2 : 0); return 0; }
  if (idx == 1) { g_dev_flags = (g_dev_flags & ~1) | (value ? 1 : 0); return 0; }
  return 0;
}

// ---- direct weight gradient for the narrow outer levels of the nested models (64 output channels at 128^2 ... 1024^2) ----
// dW[64][9 x 64] of a 3x3 64 -> 64 convolution over a million pixels is an HBM
// problem: x and dY are 134 MB each at batch 16, 256^2, the product is 37 K numbers.  The split GEMM above pads the 64
// output channels to a 128-wide tile and walks K = 576 in five column tiles, each of which streams dY again and its
// shifted copy of x through L2 -- 345 us per layer against a 60 us HBM bound (profiles/r04_shapes_nested256.txt, 5.8 ms
// of the nested 64+256 step).  Here a block owns the WHOLE 64 x K gradient in registers and streams pixel tiles past it:
//   * a TH x 32 pixel tile of dY and the (TH + 2) x 34 halo of x are staged in LDS once (HBM -> registers -> LDS, the next
//     tile's loads in flight under this tile's MFMAs); the nine taps are nine shifted views of the halo;
//   * the reduction index is the PIXEL, so both MFMA operands are read with the LDS transpose read from the natural
//     [pixel][channel] images (lane i of a 16-lane group supplies pixel i >> 2, 4-channel segment i & 3 and receives
//     channel i: 8 consecutive pixels of one channel per lane = one operand of v_mfma_f32_16x16x32_bf16).  The pixel
//     pitch is padded so that the eight 32-byte rows one 32-lane half reads (pixels p .. p+3 and p+8 .. p+11) fall on
//     eight disjoint 8-bank windows: pitch / 4 = 20 or 52 (mod 64);
//   * 8 waves = 2 halves of the output channels x 4 quarters of the K columns; A = x^T fragment (16 K columns), B = dY^T
//     fragment (16 output channels), so a lane ends up with 4 consecutive K columns of one output channel (16-byte stores);
//   * every block writes ONE slab [64][K] (and its column sums of dY for the bias gradient) after its last tile; the
//     slabs are added by mdm_conv_wgrad_reduce like the split ranges of the GEMM path.
// Measured (nested 64+256, batch 16): 345 -> 155 us per layer (500 TF, 1.7 TB/s of operands); what is left is one exposed
// memory round trip per tile -- the staging registers leave room for ONE tile of prefetch.  The nested model's step does
// not move (59.4 vs 59.4 ms, alternating): its weight gradients run on the side stream, which is not its critical path.
struct WgDirectArgs {
  const bf16* x; const bf16* dy; float* slab; float* bslab;
  int N, H, W, tiles_x, tiles_y, tiles;
};
constexpr int wgd_pitch(int bytes) {
  int d = bytes / 4;
  while (d % 64 != 20 && d % 64 != 52) ++d;
  return d * 4;
}
typedef __attribute__((ext_vector_type(4))) short wg_s16x4;
__device__ __forceinline__ Frag<bf16> wgd_tr2(const char* p, int second) {
  const wg_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p));
  const wg_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s16x4*)(p + second));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  Frag<bf16> f;
  f.v = __builtin_bit_cast(bf16x8, v);
  return f;
}

template <int CIN, int TAPS, int TH>
__global__ __launch_bounds__(512, 1) void wgrad_direct_kernel(WgDirectArgs p) {
  constexpr int COUT = 64, TW = 32, HALO = TAPS == 9 ? 1 : 0;
  constexpr int TWH = TW + 2 * HALO, THH = TH + 2 * HALO;
  constexpr int XCH = CIN / 8, DCH = COUT / 8;                       // 16-byte chunks per pixel
  constexpr int PX = wgd_pitch(CIN * 2), PD = wgd_pitch(COUT * 2);   // bytes per staged pixel
  constexpr int K = TAPS * CIN;
  constexpr int NBW = K / 16 / 4;                                    // 16-column blocks of dW per wave
  static_assert((K / 16) % 4 == 0, "K columns split over 4 waves");
  constexpr int XROW = TWH * XCH, XCHUNKS = THH * XROW, DCHUNKS = TH * TW * DCH;
  constexpr int NXL = (XCHUNKS + 511) / 512, NDL = (DCHUNKS + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  char* const xs = dsm;
  char* const ys = dsm + THH * TWH * PX;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const int mw = wid >> 2, nw = wid & 3;

  // per-lane offsets of the transpose reads: pixel 8 quad + (l15 >> 2) of a 32-pixel row segment, 4-channel segment l15 & 3
  const int trow = 8 * quad + (l15 >> 2), tseg = (l15 & 3) * 8;
  int y_off[2], x_off[NBW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) y_off[mi] = trow * PD + (mw * 2 + mi) * 32 + tseg;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int nbg = nw * NBW + nb, tap = nbg / (CIN / 16), cb = nbg - tap * (CIN / 16);
    const int dy_ = TAPS == 9 ? tap / 3 : 0, dx_ = TAPS == 9 ? tap % 3 : 0;
    x_off[nb] = (dy_ * TWH + dx_ + trow) * PX + cb * 32 + tseg;
  }

  f32x4 acc[NBW][2];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) acc[nb][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = p.bslab != nullptr && nw == 0;   // wave-uniform

  uint4 prex[NXL], prey[NDL];
  auto issue = [&](int tile) {
    const int tx = tile % p.tiles_x, ty = (tile / p.tiles_x) % p.tiles_y, n = tile / (p.tiles_x * p.tiles_y);
    const int gx0 = tx * TW - HALO, gy0 = ty * TH - HALO;
    const bf16* xn = p.x + (size_t)n * p.H * p.W * CIN;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int c = tid + 512 * i;
      const int row = c / XROW, col = c - row * XROW, px = col / XCH, ch = col - px * XCH;
      const int gy = gy0 + row, gx = gx0 + px;
      prex[i] = uint4{0u, 0u, 0u, 0u};
      if (c < XCHUNKS && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
        prex[i] = *reinterpret_cast<const uint4*>(xn + ((size_t)gy * p.W + gx) * CIN + ch * 8);
    }
    const bf16* yn = p.dy + (((size_t)n * p.H + ty * TH) * p.W + tx * TW) * COUT;
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const int c = tid + 512 * i;
      const int row = c / (TW * DCH), col = c - row * (TW * DCH);
      prey[i] = uint4{0u, 0u, 0u, 0u};
      if (c < DCHUNKS) prey[i] = *reinterpret_cast<const uint4*>(yn + (size_t)row * p.W * COUT + col * 8);
    }
  };
  if ((int)blockIdx.x < p.tiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int c = tid + 512 * i;
      const int row = c / XROW, col = c - row * XROW, px = col / XCH, ch = col - px * XCH;
      if (c < XCHUNKS) *reinterpret_cast<uint4*>(xs + (row * TWH + px) * PX + ch * 16) = prex[i];
    }
#pragma unroll
    for (int i = 0; i < NDL; ++i) {
      const int c = tid + 512 * i;
      const int pix = c / DCH, ch = c - pix * DCH;
      if (c < DCHUNKS) *reinterpret_cast<uint4*>(ys + pix * PD + ch * 16) = prey[i];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < p.tiles) issue(tile + gridDim.x);
#pragma unroll 2
    for (int r = 0; r < TH; ++r) {
      Frag<bf16> yf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) yf[mi] = wgd_tr2(ys + r * (TW * PD) + y_off[mi], 4 * PD);
      if (do_bias) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) bsum[mi] = frag_sum(yf[mi], bsum[mi]);
      }
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const Frag<bf16> xf = wgd_tr2(xs + r * (TWH * PX) + x_off[nb], 4 * PX);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) mma16(acc[nb][mi], xf, yf[mi]);
      }
    }
    __syncthreads();   // the next tile's staging overwrites what the slower waves may still read
  }
  // acc[nb][mi][i] = dW[cout = (mw * 2 + mi) * 16 + l15][k = (nw * NBW + nb) * 16 + quad * 4 + i]
  float* const S = p.slab + (size_t)blockIdx.x * COUT * K;
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      *reinterpret_cast<f32x4*>(S + (size_t)((mw * 2 + mi) * 16 + l15) * K + (nw * NBW + nb) * 16 + quad * 4) = acc[nb][mi];
  if (do_bias) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float v = bsum[mi];                     // the four quads hold the four 8-pixel groups of every row segment
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (quad == 0) p.bslab[(size_t)blockIdx.x * COUT + (mw * 2 + mi) * 16 + l15] = v;
    }
  }
  if (p.bslab != nullptr && blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(p.bslab - WG_BHDR)[0] = 1;   // one row per slab
}

// slabs (= blocks) of the direct kernel for a problem, 0 = not a shape it is built for.  Decided from (M, Cout, K) alone so
// that mdm_conv_wgrad and mdm_conv_wgrad_reduce agree; a problem of these sizes whose geometry the kernel cannot take
// (1x1 with Cin = 576, an image not made of 8 x 32 tiles) runs the split GEMM with the same number of ranges.
constexpr int WGD_SLABS = 256;
static int wgrad_direct_slabs(int M, int Cout, int K, int dtype) {
  if (g_no_wgrad_direct || dtype != DT_BF16 || Cout != 64 || M < 262144) return 0;
  return (K == 576 || K == 128 || K == 192) ? WGD_SLABS : 0;
}
static bool wgrad_direct_ok(const WgradArgs& a, int ksize) {
  if (a.stride != 1 || a.Ho != a.H || a.Wo != a.W || a.Cout != 64 || a.skip_cout) return false;
  // 3x3 only.  The 1x1 shortcuts (128 / 192 -> 64) have 4-6 MFMAs per wave and tile row: there the tile loop is one
  // exposed memory round trip per 256 pixels (261 us measured) and the split GEMM with the same 256 ranges is faster
  // (225 us; 448 us with the 10-range split the cost model gave it before).
  if (!(ksize == 3 && a.Cin == 64)) return false;
  if (((uintptr_t)a.x & 15) || ((uintptr_t)a.dy & 15)) return false;
  if (a.H % 8 != 0 || a.W % 32 != 0) return false;
  return (long)a.N * (a.H / 8) * (a.W / 32) >= WGD_SLABS;
}
template <int CIN, int TAPS, int TH>
static int launch_wgrad_direct(const WgradArgs& a, hipStream_t st) {
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int smem = (TH + 2 * HALO) * (32 + 2 * HALO) * wgd_pitch(CIN * 2) + TH * 32 * wgd_pitch(128);
  static_assert(smem <= 160 * 1024, "LDS");
  auto kern = wgrad_direct_kernel<CIN, TAPS, TH>;
  ensure_dynamic_lds(kern, smem);
  WgDirectArgs d;
  d.x = (const bf16*)a.x; d.dy = (const bf16*)a.dy; d.slab = a.slab; d.bslab = a.bslab;
  d.N = a.N; d.H = a.H; d.W = a.W; d.tiles_x = d.W / 32; d.tiles_y = d.H / TH; d.tiles = d.tiles_x * d.tiles_y * d.N;
  hipLaunchKernelGGL(kern, dim3(WGD_SLABS), dim3(512), smem, st, d);
  MDM_NOTE_KERNEL("wgrad_direct_kernel<%d, %d, %d>", CIN, TAPS, TH);
  MDM_LAUNCH_STATUS();
}

// workspace size (bytes) the caller must provide to mdm_conv_wgrad
// Tile edge (128: 4 waves, 2 blocks / CU; 256: bf16 8-wave kernel, 1 block / CU) and split count of a problem, by a
// small cost model in microseconds: rounds of resident blocks x (reduction tiles per split x tile time + per-block
// prologue and slab write) + the slab traffic of the reduce kernel.  Replaces "about 2 blocks per CU": at
// Cout x K = 768 x 3072 that rule gave 15 splits = 540 blocks = 2.1 rounds of the 256 CUs.
static void wgrad_choose(int M, int Cout, int K, int dtype, int* te_out, int* splits_out) {
  if (const int slabs = wgrad_direct_slabs(M, Cout, K, dtype)) { *te_out = 128; *splits_out = slabs; return; }
  const int bkm = dtype == DT_F32 ? 32 : 64;
  const int mt_total = (M + bkm - 1) / bkm;
  double best = 1e30;
  int best_te = 128, best_s = 1;
  for (int te = 128; te <= 256; te += 128) {
    if (te == 256 && !(dtype == DT_BF16 && Cout >= 192 && K >= 192)) continue;
    const int tiles = ((Cout + te - 1) / te) * ((K + te - 1) / te);
    const int slots = te == 256 ? 256 : 512;
    const double t_tile = (te == 256 ? 1.7 : 1.06) * (dtype == DT_F32 ? 8.0 : 1.0), t_fix = te == 256 ? 12.0 : 5.0;
    const int smax = mt_total < 64 ? mt_total : 64;
    for (int sp = 1; sp <= smax; ++sp) {
      const int per = (mt_total + sp - 1) / sp;
      if ((mt_total + per - 1) / per != sp) continue;          // canonical split counts only
      const long blocks = (long)tiles * sp;
      const double rounds = (double)((blocks + slots - 1) / slots);
      const double cost = rounds * (per * t_tile + t_fix) + sp * (double)Cout * K * 4.0 / 5.0e6 + (sp > 1 ? 4.0 : 0.0);
      if (cost < best) { best = cost; best_te = te; best_s = sp; }
    }
  }
  *te_out = best_te;
  *splits_out = best_s;
}

static int wgrad_tile(int M, int Cout, int K, int dtype) {
  int te, sp;
  wgrad_choose(M, Cout, K, dtype, &te, &sp);
  return te;
}

extern "C" int mdm_conv_wgrad_tile(int M, int Cout, int K, int dtype) { return wgrad_tile(M, Cout, K, dtype); }

extern "C" int mdm_conv_wgrad_plan(int M, int Cout, int K, int dtype, int* splits_out, size_t* ws_bytes) {
  MDM_CHECK_ARG(splits_out && ws_bytes);
  int te, splits;
  wgrad_choose(M, Cout, K, dtype, &te, &splits);
  *splits_out = splits;
  // weight slabs + bias-gradient partials (or the column-sum workspace of the fp32 path)
  const int brows = splits * WG_BSHARE > 64 ? splits * WG_BSHARE : 64;
  *ws_bytes = ((size_t)splits * Cout * K + WG_BHDR + (size_t)brows * Cout) * sizeof(float);
  return 0;
}

extern "C" int mdm_colsum(const void* x, float* out, float* ws, int M, int C, int accumulate, int dtype, void* stream);

// buffer-addressed wgrad (conv_wgrad_bl_kernel) usable for this bf16 problem?
static bool wgrad_bl_ok(const WgradArgs& a, int ksize) {
  if (ksize == 3) {
    if (a.stride != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if ((a.H & (a.H - 1)) || (a.W & (a.W - 1))) return false;
  }
  const size_t lim = 0x7F000000u;
  const size_t bias = ksize == 3 ? (size_t)(a.W + 1) * a.Cin * 2 : 0;
  return (size_t)a.M * a.Cout * 2 <= lim && (size_t)a.N * a.H * a.W * a.Cin * 2 + bias <= lim;
}

static int conv_wgrad_impl(const void* x, const void* dy, int want_bias, float* ws, int N, int H, int W, int Cin,
                           int Ho, int Wo, int Cout, int ksize, int stride, int dtype, int skip_cout, void* stream) {
  MDM_CHECK_ARG(x && dy && ws);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(Cin % epv == 0 && Cout % epv == 0);
  WgradArgs a = {};
  a.x = x; a.dy = dy; a.slab = ws; a.skip_cout = skip_cout;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride;
  a.M = N * Ho * Wo; a.K = ksize * ksize * Cin; a.groups = 0;
  size_t wsb;
  int rc = mdm_conv_wgrad_plan(a.M, Cout, a.K, dtype, &a.splits, &wsb);
  if (rc) return rc;
  const int bkm = dtype == DT_F32 ? 32 : 64;
  const int mt_total = (a.M + bkm - 1) / bkm;
  a.mtiles_per_split = (mt_total + a.splits - 1) / a.splits;
  float* const bias_ws = ws + (size_t)a.splits * Cout * a.K;
  a.bslab = (want_bias && dtype == DT_BF16) ? bias_ws + WG_BHDR : nullptr;   // the bf16 kernels fold the column sums in
  const int te = wgrad_tile(a.M, Cout, a.K, dtype);
  const int tiles = ((Cout + te - 1) / te) * ((a.K + te - 1) / te);
  {   // k-tile blocks that share the column sums: those of the whole row of tiles, or of a phase's first tap (blocked form)
    const int kt = skip_cout > 0 ? Cin / te : (a.K + te - 1) / te;
    a.bias_share = kt < WG_BSHARE ? (kt < 1 ? 1 : kt) : WG_BSHARE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (wgrad_direct_slabs(a.M, Cout, a.K, dtype) && wgrad_direct_ok(a, ksize)) {
    return launch_wgrad_direct<64, 9, 8>(a, st);
  }
  constexpr int smem = 4 * 128 * 128, smem_big = 4 * 64 * 512;
  ensure_dynamic_lds(conv_wgrad_kernel<float, MODE_1x1>, smem);
  ensure_dynamic_lds(conv_wgrad_kernel<float, MODE_3x3>, smem);
  ensure_dynamic_lds(conv_wgrad_tr_kernel<MODE_1x1, 0>, smem);
  ensure_dynamic_lds(conv_wgrad_tr_kernel<MODE_3x3, 0>, smem);
  ensure_dynamic_lds(conv_wgrad_tr_kernel<MODE_1x1, 1>, smem_big);
  ensure_dynamic_lds(conv_wgrad_tr_kernel<MODE_3x3, 1>, smem_big);
  ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_1x1, 0>, smem);
  ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_3x3, 0>, smem);
  ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_1x1, 1>, smem_big);
  ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_3x3, 1>, smem_big);
  dim3 grid(tiles * a.splits);
  const int mode = ksize == 1 ? MODE_1x1 : MODE_3x3;
  if (dtype == DT_F32) MDM_NOTE_KERNEL("conv_wgrad_kernel<float, %d>", mode);
  else if (wgrad_bl_ok(a, ksize)) MDM_NOTE_KERNEL("conv_wgrad_bl_kernel<%d, %d>", mode, te == 256 ? 1 : 0);
  else MDM_NOTE_KERNEL("conv_wgrad_tr_kernel<%d, %d>", mode, te == 256 ? 1 : 0);
  if (dtype == DT_F32) {
    if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_kernel<float, MODE_1x1>), grid, dim3(256), smem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<float, MODE_3x3>), grid, dim3(256), smem, st, a);
  } else if (wgrad_bl_ok(a, ksize)) {
    if (te == 256) {
      if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_1x1, 1>), grid, dim3(512), smem_big, st, a, WgradGroup{});
      else hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_3x3, 1>), grid, dim3(512), smem_big, st, a, WgradGroup{});
    } else {
      if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_1x1, 0>), grid, dim3(256), smem, st, a, WgradGroup{});
      else hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_3x3, 0>), grid, dim3(256), smem, st, a, WgradGroup{});
    }
  } else if (te == 256) {
    if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_tr_kernel<MODE_1x1, 1>), grid, dim3(512), smem_big, st, a);
    else hipLaunchKernelGGL((conv_wgrad_tr_kernel<MODE_3x3, 1>), grid, dim3(512), smem_big, st, a);
  } else {
    if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_tr_kernel<MODE_1x1, 0>), grid, dim3(256), smem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_tr_kernel<MODE_3x3, 0>), grid, dim3(256), smem, st, a);
  }
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_conv_wgrad(const void* x, const void* dy, int want_bias, float* ws, int N, int H, int W, int Cin,
                              int Ho, int Wo, int Cout, int ksize, int stride, int dtype, void* stream) {
  return conv_wgrad_impl(x, dy, want_bias, ws, N, H, W, Cin, Ho, Wo, Cout, ksize, stride, dtype, 0, stream);
}

// Weight gradient of upsample2x -> conv3x3 in its sub-pixel form: the split GEMM of mdm_conv_wgrad over the LOW-resolution
// x [N, H, W, Cin] and the 2x2-blocked gradient dyb [N, H, W, 4 Cout] (channel (ph, pw, co)), computing for every phase
// only the four low-resolution taps it reads (16 of the 36 (phase, tap) blocks).  ws / splits as for
// mdm_conv_wgrad_plan(M = N H W, 4 Cout, 9 Cin); mdm_conv_wgrad_reduce(.., Cout = 4 Cout, ksize 3, ..) then gives
// dwb (4 Cout, Cin, 3, 3) whose computed blocks mdm_upconv_wfold adds up into dW (Cout, Cin, 3, 3).
// bf16, H and W powers of two, Cin % 256 == 0, Cout % 256 == 0.
extern "C" int mdm_conv_wgrad_blocked(const void* x, const void* dyb, int want_bias, float* ws, int N, int H, int W, int Cin,
                                      int Cout, int dtype, void* stream) {
  MDM_CHECK_ARG(dtype == DT_BF16 && Cin % 256 == 0 && Cout % 256 == 0);
  MDM_CHECK_ARG((H & (H - 1)) == 0 && (W & (W - 1)) == 0);
  return conv_wgrad_impl(x, dyb, want_bias, ws, N, H, W, Cin, H, W, 4 * Cout, 3, 1, dtype, Cout, stream);
}

// Tile edge of a grouped 1x1 weight gradient, or 0 when grouping `groups` problems of this shape would not fill the
// chip without a split (the caller then launches them one by one through mdm_conv_wgrad).
static int wgrad_group_tile(int M, int Cout, int K, int groups) {
  if (groups < 2 || groups > WG_MAXG || M < 4096) return 0;
  const long t256 = (long)((Cout + 255) / 256) * ((K + 255) / 256) * groups;
  const long t128 = (long)((Cout + 127) / 128) * ((K + 127) / 128) * groups;
  const int cus = device_cus();
  if (Cout >= 192 && K >= 192) {
    const double eff = (double)t256 / (double)(((t256 + cus - 1) / cus) * cus);   // occupancy of the last round
    if (t256 >= (long)(0.7 * cus) && eff >= 0.7) return 256;
  }
  const double eff = (double)t128 / (double)(((t128 + 2 * cus - 1) / (2 * cus)) * 2 * cus);
  if (t128 >= (long)(1.4 * cus) && eff >= 0.7) return 128;
  return 0;
}

extern "C" int mdm_conv_wgrad_group_plan(int M, int Cout, int K, int dtype, int groups, int* tile_out) {
  MDM_CHECK_ARG(tile_out);
  *tile_out = dtype == DT_BF16 ? wgrad_group_tile(M, Cout, K, groups) : 0;
  return 0;
}

// x[g] [M, Cin], dy[g] [M, Cout] (bf16, device), dw[g] (Cout, Cin) fp32 and dbias[g] (Cout) fp32 or null: the weight /
// bias gradients of `groups` 1x1 convolutions (or linear layers) of one shape are ADDED into dw[g] / dbias[g].
// The pointer arrays themselves are HOST arrays.
extern "C" int mdm_conv_wgrad_grouped(const void* const* x, const void* const* dy, float* const* dw,
                                      float* const* dbias, int groups, int M, int Cin, int Cout, int dtype,
                                      void* stream) {
  MDM_CHECK_ARG(x && dy && dw && groups >= 1 && groups <= WG_MAXG);
  MDM_CHECK_ARG(dtype == DT_BF16 && Cin % 8 == 0 && Cout % 8 == 0 && M > 0);
  WgradArgs a = {};
  a.N = M; a.H = 1; a.W = 1; a.Cin = Cin; a.Ho = 1; a.Wo = 1; a.Cout = Cout; a.stride = 1;
  a.M = M; a.K = Cin; a.splits = 1; a.groups = groups;
  a.mtiles_per_split = (M + 63) / 64;
  MDM_CHECK_ARG(wgrad_bl_ok(a, 1));
  WgradGroup gr = {};
  for (int g = 0; g < groups; ++g) {
    MDM_CHECK_ARG(x[g] && dy[g] && dw[g]);
    gr.x[g] = x[g]; gr.dy[g] = dy[g]; gr.out[g] = dw[g]; gr.bout[g] = dbias ? dbias[g] : nullptr;
  }
  int te = wgrad_group_tile(M, Cout, Cin, groups);
  if (!te) te = (Cout >= 192 && Cin >= 192) ? 256 : 128;   // legal for any group size; the plan decides when it pays
  const int tiles = ((Cout + te - 1) / te) * ((Cin + te - 1) / te);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int smem = 4 * 128 * 128, smem_big = 4 * 64 * 512;
  if (te == 256) {
    ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_1x1, 1>, smem_big);
    hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_1x1, 1>), dim3(tiles * groups), dim3(512), smem_big, st, a, gr);
  } else {
    ensure_dynamic_lds(conv_wgrad_bl_kernel<MODE_1x1, 0>, smem);
    hipLaunchKernelGGL((conv_wgrad_bl_kernel<MODE_1x1, 0>), dim3(tiles * groups), dim3(256), smem, st, a, gr);
  }
  MDM_NOTE_KERNEL("conv_wgrad_bl_kernel<%d, %d>", MODE_1x1, te == 256 ? 1 : 0);
  MDM_LAUNCH_STATUS();
}

// Second half of the weight gradient: dw (Cout, Cin, k, k) (+)= sum over the split slabs written by mdm_conv_wgrad
// (also converts the packed [o][tap][i] slab order to the reference OIHW layout), dbias (+)= bias partials.
// Must be called with the same geometry / dtype / workspace right after mdm_conv_wgrad on the same stream.
extern "C" int mdm_conv_wgrad_reduce(const float* ws, float* dw_oihw, float* dbias, const void* dy, int M, int Cin,
                                     int Cout, int ksize, int accumulate, int dtype, void* stream) {
  MDM_CHECK_ARG(ws && dw_oihw && (ksize == 1 || ksize == 3));
  if (g_skip_wgrad_reduce) return 0;
  const int K = ksize * ksize * Cin;
  int splits; size_t wsb;
  int rc = mdm_conv_wgrad_plan(M, Cout, K, dtype, &splits, &wsb);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* const bias_ws = const_cast<float*>(ws) + (size_t)splits * Cout * K;
  const float* bslab = (dbias && dtype == DT_BF16) ? bias_ws + WG_BHDR : nullptr;
  const int bblocks = bslab ? (Cout + 31) / 32 : 0;
  if (ksize == 1) {
    const size_t total = (size_t)Cout * Cin;
    const int wblocks = (int)((total / 4 + 255) / 256 > 2048 ? 2048 : (total / 4 + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(wblocks + bblocks), dim3(256), 0, st, ws, dw_oihw, bslab, dbias,
                       splits, Cout, total, accumulate, wblocks, bblocks);
  } else {
    const int wblocks = Cout * ((Cin + 63) / 64);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + bblocks), dim3(192), 0, st, ws, dw_oihw, bslab, dbias,
                       splits, Cout, Cin, ksize * ksize, accumulate, bblocks);
  }
  if (dbias && !bslab) {
    MDM_CHECK_ARG(dy);
    return mdm_colsum(dy, dbias, bias_ws, M, Cout, accumulate, dtype, stream);
  }
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_pack_weight(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int ksize,
                               int Cin_pad, int Cout_pad, int kblock_fwd, int kblock_dgrad, int dtype,
                               void* stream) {
  MDM_CHECK_ARG(w_oihw && w_fwd);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(Cin_pad >= Cin && Cout_pad >= Cout);
  MDM_CHECK_ARG(kblock_fwd == 0 || (ksize == 3 && Cin_pad % kblock_fwd == 0));
  MDM_CHECK_ARG(kblock_dgrad == 0 || (ksize == 3 && Cout_pad % kblock_dgrad == 0));
  const int taps = ksize * ksize;
  const size_t total = (size_t)Cout * taps * Cin_pad + (w_dgrad ? (size_t)Cin * taps * Cout_pad : 0);
  const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (Cin_pad == Cin && Cout_pad == Cout && Cin % 32 == 0 && Cout % 32 == 0) {
    dim3 grid(Cin / 32, Cout / 32);
#define MDM_PACK_TILED(TT, TAPS)                                                                               \
  hipLaunchKernelGGL((pack_weight_tiled_kernel<TT, TAPS>), grid, dim3(256), 0, st, w_oihw, (TT*)w_fwd, (TT*)w_dgrad, \
                     Cout, Cin, kblock_fwd, kblock_dgrad)
    if (dtype == DT_F32) { if (taps == 9) MDM_PACK_TILED(float, 9); else MDM_PACK_TILED(float, 1); }
    else if (dtype == DT_BF16) { if (taps == 9) MDM_PACK_TILED(bf16, 9); else MDM_PACK_TILED(bf16, 1); }
    else MDM_CHECK_ARG(false);
#undef MDM_PACK_TILED
    MDM_LAUNCH_STATUS();
  }
  if (dtype == DT_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(nb), dim3(256), 0, st, w_oihw, (float*)w_fwd, (float*)w_dgrad, Cout, Cin, taps, Cin_pad, Cout_pad, kblock_fwd, kblock_dgrad);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(nb), dim3(256), 0, st, w_oihw, (bf16*)w_fwd, (bf16*)w_dgrad, Cout, Cin, taps, Cin_pad, Cout_pad, kblock_fwd, kblock_dgrad);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

// table: DEVICE array of n descriptors {const float* w; void* w_fwd; void* w_dgrad; int Cout, Cin, taps, kblock_fwd,
// kblock_dgrad, first_block} (48 bytes each, first_block = prefix sum of (Cout/32)*(Cin/32)); every weight must satisfy
// Cout % 32 == 0 and Cin % 32 == 0 (no channel padding).  total_blocks = the sum of all brick counts.
extern "C" int mdm_pack_weights_multi(const void* table, int n, int total_blocks, int dtype, void* stream) {
  MDM_CHECK_ARG(table && n > 0 && total_blocks > 0);
  static_assert(sizeof(PackDesc) == 48, "descriptor layout is part of the ABI");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32) hipLaunchKernelGGL(pack_weights_multi_kernel<float>, dim3(total_blocks), dim3(256), 0, st, (const PackDesc*)table, n);
  else if (dtype == DT_BF16) hipLaunchKernelGGL(pack_weights_multi_kernel<bf16>, dim3(total_blocks), dim3(256), 0, st, (const PackDesc*)table, n);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

static inline int colsum_slabs(int M, int C, int epv) {
  const int groups = (C + 8 * epv - 1) / (8 * epv);
  int slabs = (1024 + groups - 1) / groups;
  const int max_slabs = (M + 63) / 64;
  if (slabs > max_slabs) slabs = max_slabs;
  if (slabs > 64) slabs = 64;
  if (slabs < 1) slabs = 1;
  return slabs;
}

extern "C" int mdm_colsum_plan(int M, int C, int* nblocks, size_t* ws_bytes) {
  MDM_CHECK_ARG(nblocks && ws_bytes);
  const int slabs = colsum_slabs(M, C, 4);  // upper bound over both dtypes
  *nblocks = slabs;
  *ws_bytes = (size_t)64 * C * sizeof(float);
  return 0;
}

extern "C" int mdm_colsum(const void* x, float* out, float* ws, int M, int C, int accumulate, int dtype,
                          void* stream) {
  MDM_CHECK_ARG(x && out && ws);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0);
  const int slabs = colsum_slabs(M, C, epv);
  const int rps = (M + slabs - 1) / slabs;
  const int groups = (C + 8 * epv - 1) / (8 * epv);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(groups, slabs), dim3(256), 0, st, (const float*)x, ws, M, C, rps);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(groups, slabs), dim3(256), 0, st, (const bf16*)x, ws, M, C, rps);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, out, slabs, C, accumulate);
  MDM_LAUNCH_STATUS();
}
