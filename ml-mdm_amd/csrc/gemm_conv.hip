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
#include "common.hpp"

namespace mdm {

struct ConvArgs {
  const void* x;      // activation (A source) [N, H, W, Cin]
  const void* w;      // packed weight [Cout][K]
  const float* bias;  // [Cout] or null
  const void* res;    // residual [M, Cout] or null (added last)
  const void* aux;    // pre-activation for act==2 [M, Cout]
  void* y;            // output [M, Cout]
  void* ypre;         // optional pre-GELU output (act==1) or null
  int N, H, W, Cin;   // geometry of x
  int Ho, Wo, Cout;   // geometry of y
  int stride;
  int M, K;
  int act;            // 0 none, 1 y=gelu(v), 2 y=v*gelu'(aux)
};

enum { MODE_1x1 = 0, MODE_3x3 = 1, MODE_3x3_T2 = 2 };

template <typename T, int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs p) {
  constexpr int EPV = Tr<T>::EPV, BK = Tr<T>::BK, KSTEPS = Tr<T>::KSTEPS;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int AJ = BM / 32, BJ = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(WM * WN == 4, "4 waves");
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
  const int lrow = tid >> 3, lchunk = tid & 7;
  const int pchunk = lchunk ^ (lrow & 7);
  int a_pix[AJ];   // n * H * W  (pixel index base of the image), -1 if row >= M
  int a_oh[AJ], a_ow[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + lrow + 32 * j;
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
  if (MODE != MODE_1x1) { tap = kcur / p.Cin; cin = kcur - tap * p.Cin; }

  uint4 ra[AJ], rb[BJ];
  auto load_tile = [&]() {
    const bool kvalid = kcur < p.K;
    int kh = 0, kw = 0;
    if (MODE != MODE_1x1) { kh = (tap * 11) >> 5; kw = tap - 3 * kh; }
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      bool v = kvalid && a_pix[j] >= 0;
      size_t off;
      if (MODE == MODE_1x1) {
        off = (size_t)a_pix[j] * p.Cin + kcur;
      } else if (MODE == MODE_3x3) {
        const int ih = a_oh[j] * p.stride + kh - 1, iw = a_ow[j] * p.stride + kw - 1;
        v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
        off = (size_t)(a_pix[j] + ih * p.W + iw) * p.Cin + cin;
      } else {  // transposed stride 2: source index = (o + tap - 1) / 2 when even
        const int th = a_oh[j] + kh - 1, tw = a_ow[j] + kw - 1;
        const int ih = th >> 1, iw = tw >> 1;
        v = v && th >= 0 && tw >= 0 && !(th & 1) && !(tw & 1) && ih < p.H && iw < p.W;
        off = (size_t)(a_pix[j] + ih * p.W + iw) * p.Cin + cin;
      }
      uint4 z = {0u, 0u, 0u, 0u};
      ra[j] = v ? *reinterpret_cast<const uint4*>(X + off) : z;
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int n = n0 + lrow + 32 * j;
      const bool v = kvalid && n < p.Cout;
      uint4 z = {0u, 0u, 0u, 0u};
      rb[j] = v ? *reinterpret_cast<const uint4*>(Wp + (size_t)n * p.K + kcur) : z;
    }
  };
  auto advance_k = [&]() {
    kcur += BK;
    if (MODE != MODE_1x1) {
      cin += BK;
      while (cin >= p.Cin) { cin -= p.Cin; ++tap; }
    }
  };
  auto store_tile = [&](char* stage) {
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      *reinterpret_cast<uint4*>(stage + (lrow + 32 * j) * 128 + pchunk * 16) = ra[j];
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      *reinterpret_cast<uint4*>(stage + A_BYTES + (lrow + 32 * j) * 128 + pchunk * 16) = rb[j];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = (p.K + BK - 1) / BK;
  load_tile();
  advance_k();
  store_tile(smem);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    char* cur = smem + (kt & 1) * STAGE;
    const bool more = kt + 1 < ntiles;
    if (more) { load_tile(); advance_k(); }
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
    if (more) store_tile(smem + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  T* __restrict__ Ypre = reinterpret_cast<T*>(p.ypre);
  const T* __restrict__ R = reinterpret_cast<const T*>(p.res);
  const T* __restrict__ AUX = reinterpret_cast<const T*>(p.aux);
  const bool vec_ok = (p.Cout & 3) == 0;
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
      const int nv = vec_ok ? 4 : min(4, p.Cout - n);
      if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] += p.bias[n + e];
      }
      if (p.act == 1) {
        if (Ypre) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (e < nv) Ypre[o + e] = from_f32<T>(v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
      } else if (p.act == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] *= dgelu_f(to_f32(AUX[o + e]));
      }
      if (R) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] += to_f32(R[o + e]);
      }
      if (vec_ok) {
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          bf16x4 b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *reinterpret_cast<bf16x4*>(Y + o) = b;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) Y[o + e] = from_f32<T>(v[e]);
      }
    }
  }
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
  int N, H, W, Cin, Ho, Wo, Cout, stride;
  int M, K;
  int splits, mtiles_per_split;
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
  const int split = blockIdx.x / tiles;
  const int t = blockIdx.x - split * tiles;
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
    b_cc[i] = r % CPR;
    b_mb[i] = r / CPR;
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

  auto load_tile = [&](int mt) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int mbase = mt * BKM + b_mb[i] * EPV;
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        const int m = mbase + e;
        bool v = b_cvalid[i] && m < p.M;
        if (b_op[i] == 0) {
          if (v) blk[i].load_row(e, DY + (size_t)m * p.Cout + n0 + b_cc[i] * EPV);
          else blk[i].zero_row(e);
        } else {
          size_t off;
          if (MODE == MODE_1x1) {
            off = (size_t)m * p.Cin + b_cin[i];
          } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, r = m - n * hw;
            const int oh = r / p.Wo, ow = r - oh * p.Wo;
            const int kh = (b_tap[i] * 11) >> 5, kw = b_tap[i] - 3 * kh;
            const int ih = oh * p.stride + kh - 1, iw = ow * p.stride + kw - 1;
            v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
            off = (size_t)((n * p.H + ih) * p.W + iw) * p.Cin + b_cin[i];
          }
          if (v) blk[i].load_row(e, X + off);
          else blk[i].zero_row(e);
        }
      }
    }
  };
  auto store_tile = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      char* base = stage + b_op[i] * A_BYTES;
#pragma unroll
      for (int c = 0; c < EPV; ++c) {
        const int row = b_cc[i] * EPV + c;
        *reinterpret_cast<uint4*>(base + lds_chunk_off(row, b_mb[i])) = blk[i].col(c);
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (mt_begin < mt_end) {
    load_tile(mt_begin);
    store_tile(smem);
    __syncthreads();
    for (int mt = mt_begin; mt < mt_end; ++mt) {
      const int it = mt - mt_begin;
      char* cur = smem + (it & 1) * STAGE;
      const bool more = mt + 1 < mt_end;
      if (more) load_tile(mt + 1);
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
      if (more) store_tile(smem + ((it + 1) & 1) * STAGE);
      __syncthreads();
    }
  }

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

// dW_oihw[o][i][t] = sum_s slab[s][o][t*Cin + i]      (taps = 1 or 9)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits,
                                    int Cout, int Cin, int taps) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    // idx enumerates the *packed* layout (coalesced slab reads)
    const int K = Cin * taps;
    const int o = (int)(idx / K), k = (int)(idx - (size_t)o * K);
    const int tp = k / Cin, i = k - tp * Cin;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[(size_t)sp * total + idx];
    dw[((size_t)o * Cin + i) * taps + tp] = s;
  }
}

// Weight packing from the reference layout (OIHW fp32, unet.py state_dict) to the
// two kernel layouts:  fwd[o][t][i] and dgrad[i][t'][o] with t' = taps-1-t (flip).
// Cin_pad >= Cin lets the 3-channel stem be zero-padded to a chunk multiple.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd,
                                   int Cout, int Cin, int taps, int Cin_pad, int Cout_pad) {
  const size_t total_f = (size_t)Cout * taps * Cin_pad;
  const size_t total_d = wd ? (size_t)Cin * taps * Cout_pad : 0;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total_f + total_d;
       idx += (size_t)gridDim.x * blockDim.x) {
    if (idx < total_f) {
      const int i = (int)(idx % Cin_pad);
      const size_t r = idx / Cin_pad;
      const int tp = (int)(r % taps), o = (int)(r / taps);
      wf[idx] = from_f32<T>(i < Cin ? w[((size_t)o * Cin + i) * taps + tp] : 0.f);
    } else {
      const size_t d = idx - total_f;
      const int o = (int)(d % Cout_pad);
      const size_t r = d / Cout_pad;
      const int tp = (int)(r % taps), i = (int)(r / taps);
      wd[d] = from_f32<T>(o < Cout ? w[((size_t)o * Cin + i) * taps + (taps - 1 - tp)] : 0.f);
    }
  }
}

// Column sums: out[c] = sum_m x[m, c]  (bias gradients).  Two deterministic stages.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ part,
                                                             int M, int C, int rows_per_block) {
  constexpr int EPV = Tr<T>::EPV;
  __shared__ float red[256 * 8];
  const int nchunks = C / EPV;                 // host guarantees C % EPV == 0
  const int tid = threadIdx.x;
  const int m_begin = blockIdx.x * rows_per_block, m_end = min(M, m_begin + rows_per_block);
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    // threads: tc = chunk lane, tr = row lane; nchunks may be < 256 -> several rows per pass
    const int cw = min(256, nchunks - c0);
    const int rows_par = 256 / cw;             // >= 1
    const int tc = tid % cw, tr = tid / cw;
    float s[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) s[e] = 0.f;
    if (tr < rows_par) {
      for (int m = m_begin + tr; m < m_end; m += rows_par) {
        Chunk<T> ch;
        ch.load(x + (size_t)m * C + (size_t)(c0 + tc) * EPV);
#pragma unroll
        for (int e = 0; e < EPV; ++e) s[e] += ch.v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) red[tid * 8 + e] = s[e];
    __syncthreads();
    if (tr == 0) {
      for (int r = 1; r < rows_par; ++r)
#pragma unroll
        for (int e = 0; e < EPV; ++e) s[e] += red[(r * cw + tc) * 8 + e];
#pragma unroll
      for (int e = 0; e < EPV; ++e) part[(size_t)blockIdx.x * C + (size_t)(c0 + tc) * EPV + e] = s[e];
    }
    __syncthreads();
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblocks, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * C + c];
  out[c] = s;
}

}  // namespace mdm

using namespace mdm;

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int MODE>
static int launch_conv_cfg(const ConvArgs& a, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * 128;
  auto kern = conv_gemm_kernel<T, BM, BN, WM, WN, MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, st, a);
  MDM_LAUNCH_STATUS();
}

template <typename T, int MODE>
static int launch_conv_mode(const ConvArgs& a, hipStream_t st) {
  if (a.Cout <= 32) return launch_conv_cfg<T, 128, 32, 4, 1, MODE>(a, st);
  if (a.Cout <= 64) return launch_conv_cfg<T, 128, 64, 2, 2, MODE>(a, st);
  return launch_conv_cfg<T, 128, 128, 2, 2, MODE>(a, st);
}

template <typename T>
static int launch_conv_t(const ConvArgs& a, int ks, int transposed, hipStream_t st) {
  if (ks == 1) return launch_conv_mode<T, MODE_1x1>(a, st);
  if (transposed) return launch_conv_mode<T, MODE_3x3_T2>(a, st);
  return launch_conv_mode<T, MODE_3x3>(a, st);
}

extern "C" int mdm_conv_fwd(const void* x, const void* w_packed, const float* bias, const void* res,
                            const void* aux, void* y, void* y_pre, int N, int H, int W, int Cin, int Ho, int Wo,
                            int Cout, int ksize, int stride, int transposed, int act, int dtype, void* stream) {
  MDM_CHECK_ARG(x && w_packed && y);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  MDM_CHECK_ARG(act >= 0 && act <= 2);
  MDM_CHECK_ARG(act != 2 || aux);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(Cin % epv == 0);
  MDM_CHECK_ARG(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && Cout > 0);
  if (ksize == 1) { MDM_CHECK_ARG(Ho == H && Wo == W && stride == 1 && !transposed); }
  else if (transposed) { MDM_CHECK_ARG(Ho == 2 * H && Wo == 2 * W); }
  else { MDM_CHECK_ARG(stride == 1 || stride == 2); MDM_CHECK_ARG(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1); }
  ConvArgs a;
  a.x = x; a.w = w_packed; a.bias = bias; a.res = res; a.aux = aux; a.y = y; a.ypre = y_pre;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride;
  a.M = N * Ho * Wo; a.K = ksize * ksize * Cin; a.act = act;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return dtype == DT_F32 ? launch_conv_t<float>(a, ksize, transposed, st) : launch_conv_t<bf16>(a, ksize, transposed, st);
}

// workspace size (bytes) the caller must provide to mdm_conv_wgrad
extern "C" int mdm_conv_wgrad_plan(int M, int Cout, int K, int dtype, int* splits_out, size_t* ws_bytes) {
  MDM_CHECK_ARG(splits_out && ws_bytes);
  const int bkm = dtype == DT_F32 ? 32 : 64;
  const int tiles = ((Cout + 127) / 128) * ((K + 127) / 128);
  const int mt_total = (M + bkm - 1) / bkm;
  int splits = (1024 + tiles - 1) / tiles;           // aim at ~4 workgroups per CU
  if (splits > mt_total) splits = mt_total;
  const int max_by_work = (mt_total + 7) / 8;          // >= 8 reduction tiles per split
  if (splits > max_by_work) splits = max_by_work;
  if (splits < 1) splits = 1;
  const int per = (mt_total + splits - 1) / splits;
  splits = (mt_total + per - 1) / per;
  *splits_out = splits;
  *ws_bytes = (size_t)splits * Cout * K * sizeof(float);
  return 0;
}

extern "C" int mdm_conv_wgrad(const void* x, const void* dy, float* dw_oihw, float* ws, int N, int H, int W,
                              int Cin, int Ho, int Wo, int Cout, int ksize, int stride, int dtype, void* stream) {
  MDM_CHECK_ARG(x && dy && dw_oihw && ws);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(Cin % epv == 0 && Cout % epv == 0);
  WgradArgs a;
  a.x = x; a.dy = dy; a.slab = ws;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.stride = stride;
  a.M = N * Ho * Wo; a.K = ksize * ksize * Cin;
  size_t wsb;
  int rc = mdm_conv_wgrad_plan(a.M, Cout, a.K, dtype, &a.splits, &wsb);
  if (rc) return rc;
  const int bkm = dtype == DT_F32 ? 32 : 64;
  const int mt_total = (a.M + bkm - 1) / bkm;
  a.mtiles_per_split = (mt_total + a.splits - 1) / a.splits;
  const int tiles = ((Cout + 127) / 128) * ((a.K + 127) / 128);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int smem = 4 * 128 * 128;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<float, MODE_1x1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<float, MODE_3x3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<bf16, MODE_1x1>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<bf16, MODE_3x3>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  dim3 grid(tiles * a.splits), block(256);
  if (dtype == DT_F32) {
    if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_kernel<float, MODE_1x1>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<float, MODE_3x3>), grid, block, smem, st, a);
  } else {
    if (ksize == 1) hipLaunchKernelGGL((conv_wgrad_kernel<bf16, MODE_1x1>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<bf16, MODE_3x3>), grid, block, smem, st, a);
  }
  const size_t total = (size_t)Cout * a.K;
  const int rb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, ws, dw_oihw, a.splits, Cout, Cin, ksize * ksize);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_pack_weight(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int ksize,
                               int Cin_pad, int Cout_pad, int dtype, void* stream) {
  MDM_CHECK_ARG(w_oihw && w_fwd);
  MDM_CHECK_ARG(ksize == 1 || ksize == 3);
  MDM_CHECK_ARG(Cin_pad >= Cin && Cout_pad >= Cout);
  const int taps = ksize * ksize;
  const size_t total = (size_t)Cout * taps * Cin_pad + (w_dgrad ? (size_t)Cin * taps * Cout_pad : 0);
  const int nb = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(nb), dim3(256), 0, st, w_oihw, (float*)w_fwd, (float*)w_dgrad, Cout, Cin, taps, Cin_pad, Cout_pad);
  else if (dtype == DT_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(nb), dim3(256), 0, st, w_oihw, (bf16*)w_fwd, (bf16*)w_dgrad, Cout, Cin, taps, Cin_pad, Cout_pad);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_colsum_plan(int M, int C, int* nblocks, size_t* ws_bytes) {
  MDM_CHECK_ARG(nblocks && ws_bytes);
  int nb = (M + 127) / 128;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  *nblocks = nb;
  *ws_bytes = (size_t)nb * C * sizeof(float);
  return 0;
}

extern "C" int mdm_colsum(const void* x, float* out, float* ws, int M, int C, int dtype, void* stream) {
  MDM_CHECK_ARG(x && out && ws);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0);
  int nb; size_t wsb;
  mdm_colsum_plan(M, C, &nb, &wsb);
  const int rpb = (M + nb - 1) / nb;
  nb = (M + rpb - 1) / rpb;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, ws, M, C, rpb);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(nb), dim3(256), 0, st, (const bf16*)x, ws, M, C, rpb);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, out, nb, C);
  MDM_LAUNCH_STATUS();
}
