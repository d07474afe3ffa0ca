// conv_gemm_x_kernel -- the forward / input-gradient implicit GEMM for problems with MANY output tiles per CU (bf16).
// Included by gemm_conv.hip after ConvArgs; same operands, packs, k-order and rounding points as conv_gemm_bl_kernel.
//
// Why a second structure.  conv_gemm_bl_kernel gives a CU to ONE 8-wave block (8 x 256 registers): while that block runs
// its epilogue (accumulators -> LDS -> rows -> HBM, plus the activation) nothing on the CU issues MFMAs, and the next
// tile's first LDS-DMA cannot be waited for before the stores have retired (vmcnt is in-order).  On the 1x1 / FFN GEMMs
// (K = 512 ... 768: 8-12 k-tiles per output tile) that serial tail is 30-45 % of the launch
// (profiles/r03_gemm_store_phase_cost.txt).  Here the k-tile stream of a block never stops:
//   * 4 waves, ONE per SIMD, 512 registers each; a wave owns a 128 x 64 piece of a 256 x 128 output tile = 8 blocks of
//     v_mfma_f32_32x32x16_bf16 = 128 accumulator registers, and it has TWO such sets: tile n+1 accumulates into one while
//     tile n is drained from the other, a 32 x 64 slice per k-tile iteration -- v_accvgpr_read, +bias, bf16, a 4 KB
//     per-wave LDS scratch that turns the MFMA layout into whole 128-byte rows, activation / residual, global stores --
//     all of it issued in the gaps between the next tile's MFMAs;
//   * the LDS ring has 3 stages of one k-tile (64 reduction elements: 256 + 128 rows of 128 bytes = 48 KB) and runs
//     across output tiles: the loader is always 2-3 k-tiles ahead, whatever tile those belong to, so there is no
//     pipeline fill per tile either;
//   * every wait is counted: the barrier of iteration g (between its k-steps 2 and 3) waits for vmcnt(12) = everything
//     but the 12 LDS-DMA pieces of k-tile g+2, which are the youngest vector-memory operations by construction (the
//     drain's stores and operand loads are issued right after a barrier, before the first piece of the next k-tile).
// LDS: 3 x 48 KB + 4 x 4 KB = 160 KB.  Host-checked requirements (else conv_gemm_bl_kernel): bf16, 1x1 or channel-block-
// major 3x3, M % 256 == 0, Cout % 128 == 0, K % 64 == 0, K >= 384, plain [M][Cout] output.
#pragma once

namespace mdm {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct XG {
  static constexpr int BM = 256, BN = 128;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int NSTG = 3, SCR = NSTG * STAGE, SCR_WAVE = 4096, LDS = SCR + 4 * SCR_WAVE;
  static constexpr int PIECES = 12;   // LDS-DMA issues per wave per k-tile: 8 of the activation rows, 4 of the weight rows
};

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// The drain's LDS scratch traffic goes through inline asm: hipcc orders an ordinary ds_write behind every LDS-DMA in
// flight (it emits s_waitcnt vmcnt(0) in front of it: the DMA is a pending LDS write it cannot tell apart), which would
// drain the loader's pipeline once per slice.  The scratch is private to a wave and LDS operations of one wave execute
// in order, so write -> read needs no wait; the read results are consumed only behind the s_waitcnt lgkmcnt(0) in front
// of the iteration's barrier, which names them as operands so that nothing can be scheduled across it.
__device__ __forceinline__ void lds_write_b64_asm(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v));
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_b128_asm(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

// Global loads with register destinations (bias, aux / residual rows) are inline asm for the same reason: next to LDS-DMA
// in flight hipcc waits vmcnt(0) at the first use of an ordinary load.  They are issued right after a barrier, before the
// loader's next piece, so the counted wait of the NEXT barrier retires them; that wait names them as operands.
__device__ __forceinline__ u32x4 global_load_b128_asm(const void* ptr) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr));
  return r;
}

// One accumulator element, read where the drain needs it.  The "a" constraint keeps the register allocator from
// moving a whole finished tile (128 registers) into VGPRs at the start of the next pass, which it otherwise does
// (every use is a VALU instruction) -- and then spills.
__device__ __forceinline__ float acc_read(float a) {
  float r;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(a));
  return r;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  bf16x2_ t = {(bf16)a, (bf16)b};
  return *reinterpret_cast<uint32_t*>(&t);
}

// ACT: 0 none, 1 y = gelu(v) (+ the pre-activation to ypre), 2 y = v * gelu'(aux); RES: + residual (ACT 0 only)
template <int MODE, int ACT, bool RES>
__global__ __launch_bounds__(256, 1) void conv_gemm_x_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  using T = bf16;
  constexpr int BM = XG::BM, BN = XG::BN, A_BYTES = XG::A_BYTES, STAGE = XG::STAGE;
  constexpr unsigned INVALID = 0x7F000000u;
  constexpr bool OPND = ACT == 2 || RES;      // the drain reads a second [M][Cout] tensor (aux or the residual)
  static_assert(MODE == MODE_1x1 || MODE == MODE_3x3, "1x1 and channel-block-major 3x3 only");
  static_assert(!(ACT != 0 && RES), "residual: plain epilogue only");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l32 = lane & 31, hi = lane >> 5;

  const int tiles_n = p.Cout / BN;
  const int tiles_total = (p.M / BM) * tiles_n;
  const int G = gridDim.x;
  const int nk = p.K / 64;
  const int first = xcd_remap(blockIdx.x, G);
  if (first >= tiles_total) return;
  // development knob 0 (mdm_dev_set_knob; 0 in the product): bit 0 = the LDS-DMA fetches nothing (offsets out of range),
  // bit 1 = the barrier does not wait for the DMA -- both give wrong results, they price the memory side of the k-loop
  const int xknob = __builtin_amdgcn_readfirstlane(g_knobs[0]);
  // Tile order: the 32 consecutive tile ids an XCD works on at a time form SA x SB super-tiles (SA row tiles x SB column
  // tiles): SA activation panels + SB weight panels stay in its L2 instead of one activation panel + every weight panel
  const int SB = p.sel_base > 0 ? p.sel_base : tiles_n;     // host: a divisor of tiles_n (ConvArgs::sel_base is unused here)
  const int SA = p.sel_cout > 0 ? p.sel_cout : 1;           // host: a divisor of the row-tile count
  const int sup = SA * SB, sup_per_row = tiles_n / SB;
  auto tile_m0 = [&](int t) { const int s_ = t / sup, w_ = t - s_ * sup; return ((s_ / sup_per_row) * SA + w_ / SB) * BM; };
  auto tile_n0 = [&](int t) { const int s_ = t / sup, w_ = t - s_ * sup; return ((s_ % sup_per_row) * SB + w_ % SB) * BN; };

  // ---- loader: per-lane gather offsets of one output tile (bytes), fixed for its k-loop ------------------------------
  // piece j of a wave = rows 32 j + 8 wave + (lane >> 3) of the tile, 16-byte slot lane & 7 of each row; the XOR swizzle
  // sits on the source side: the lane fetches logical chunk slot ^ f(row), f(row) = (row >> 1) & 7 (conflict-free for
  // the 32-row fragments of the 32x32x16 MFMA, whose ds_read_b128 lane groups span rows {0-3, 12-15, 20-27} / ...)
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const unsigned abias = MODE == MODE_3x3 ? (unsigned)(p.W + 1) * p.Cin * 2u : 0u;
  // descriptor inputs as provably wave-uniform scalars (readfirstlane): under SGPR pressure hipcc keeps uniform values
  // in VGPRs, and a buffer descriptor or scalar offset in a VGPR turns every LDS-DMA into a waterfall loop
  auto uni_ptr = [](const void* q) -> char* {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<char*>(((uint64_t)hi << 32) | lo);
  };
  char* const a_base = uni_ptr(reinterpret_cast<const char*>(p.x) - abias);
  char* const b_base = uni_ptr(p.w);
  const unsigned a_bytes = (unsigned)p.N * p.H * p.W * p.Cin * 2u + abias;
  const unsigned b_bytes = (unsigned)p.Cout * p.K * 2u;
  unsigned a_voff[8], a_mask[8], b_voff[4];
#define MDX_DMA_SETUP(tile_)                                                                                \
  {                                                                                                         \
    const int tl_ = (tile_);                                                                                \
    const int m0_ = tile_m0(tl_), n0_ = tile_n0(tl_);                                                       \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                         \
      const int m = m0_ + lrow + 32 * j;                                                                    \
      if (MODE == MODE_1x1) {                                                                               \
        a_voff[j] = (xknob & 1) ? INVALID : (unsigned)m * p.Cin * 2u + lchunk * 16u;                        \
        a_mask[j] = 0x1ffu;                                                                                 \
      } else {                                                                                              \
        const int hw = p.Ho * p.Wo;                                                                         \
        const int n = m / hw, r = m - n * hw;                                                               \
        const int oh = r / p.Wo, ow = r - oh * p.Wo;                                                        \
        const int ih0 = oh * p.stride, iw0 = ow * p.stride;                                                 \
        a_voff[j] = (xknob & 1) ? INVALID : (unsigned)(n * p.H * p.W + ih0 * p.W + iw0) * p.Cin * 2u + lchunk * 16u; \
        unsigned mk = 0u;                                                                                   \
        _Pragma("unroll") for (int tp = 0; tp < 9; ++tp) {                                                  \
          const int ih = ih0 + tp / 3 - 1, iw = iw0 + tp % 3 - 1;                                           \
          if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) mk |= 1u << tp;                 \
        }                                                                                                   \
        a_mask[j] = mk;                                                                                     \
      }                                                                                                     \
    }                                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                           \
      b_voff[j] = (xknob & 1) ? INVALID : (unsigned)(n0_ + lrow + 32 * j) * p.K * 2u + lchunk * 16u;        \
  }
  // loader position: output tile d_tile, its k-tile d_kt (3x3: = channel block d_cb, tap d_tap), target stage d_stage
  int d_tile = first, d_kt = 0, d_cb = 0, d_tap = 0, d_stage = 0;
  bool d_live = true;
  const int wave_lds = wave * 1024;
#define MDX_BLDS(rs, lds_off, voff, soff)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (lds_off)), 16, voff, soff, 0, 0)
  // wave-uniform state of the loader's current k-tile
#define MDX_KTILE_STATE()                                                                                   \
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(a_base, 0, __builtin_amdgcn_readfirstlane(d_live ? a_bytes : 0u), 0x00020000); \
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(b_base, 0, __builtin_amdgcn_readfirstlane(d_live ? b_bytes : 0u), 0x00020000); \
  const int b_soff = __builtin_amdgcn_readfirstlane(d_kt * 128);                                            \
  int a_soff = b_soff;                                                                                      \
  unsigned tapbit = 1u;                                                                                     \
  if (MODE == MODE_3x3) {                                                                                   \
    const int kh = (d_tap * 11) >> 5, kw = d_tap - 3 * kh;                                                  \
    a_soff = __builtin_amdgcn_readfirstlane((kh * p.W + kw) * p.Cin * 2 + d_cb * 128);                      \
    tapbit = __builtin_amdgcn_readfirstlane(1u << d_tap);                                                   \
  }
#define MDX_PIECE(q)                                                                                        \
  if ((q) < 8) {                                                                                            \
    const unsigned vo = (MODE == MODE_1x1 || (a_mask[(q)] & tapbit)) ? a_voff[(q)] : INVALID;               \
    MDX_BLDS(rsA, __builtin_amdgcn_readfirstlane(d_stage + (q) * 4096 + wave_lds), vo, a_soff);                                             \
  } else {                                                                                                  \
    MDX_BLDS(rsB, __builtin_amdgcn_readfirstlane(d_stage + A_BYTES + ((q) - 8) * 4096 + wave_lds), b_voff[(q) - 8], b_soff);                \
  }
  // next k-tile of the stream (next output tile of this block after the last k-tile; past the last tile: empty
  // descriptors, the pieces are still issued -- they fetch nothing -- so the vmcnt arithmetic never changes)
#define MDX_DMA_ADVANCE()                                                                                   \
  {                                                                                                         \
    d_stage = d_stage == 2 * STAGE ? 0 : d_stage + STAGE;                                                   \
    ++d_kt;                                                                                                 \
    if (MODE == MODE_3x3) { if (++d_tap == 9) { d_tap = 0; ++d_cb; } }                                      \
    if (d_kt == nk) {                                                                                       \
      d_kt = 0; d_cb = 0; d_tap = 0;                                                                        \
      d_tile += G;                                                                                          \
      d_live = d_tile < tiles_total;                                                                        \
      if (d_live) MDX_DMA_SETUP(d_tile);                                                                    \
    }                                                                                                       \
  }

  // ---- compute side ---------------------------------------------------------------------------------------------------
  // fragment addresses within a stage: row (l32 of a 32-row block), 16-byte chunk 2 s + hi of k-step s, swizzled
  const int fsw = (l32 >> 1) & 7;
  int xa[4], wa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int slot = (2 * s + hi) ^ fsw;
    xa[s] = (wm * 128 + l32) * 128 + slot * 16;
    wa[s] = A_BYTES + (wn * 64 + l32) * 128 + slot * 16;
  }
  // drain: per-wave scratch [32 pixel rows][64 channels] bf16, 16-byte chunk index XOR (row & 7)
  const unsigned scr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + XG::SCR) + wave * XG::SCR_WAVE;
  // write side: row l32, 8-byte unit u = 8 j + 2 g + hi at unit u ^ ((l32 & 7) << 1): address = sw_addr ^ ((8 j + 2 g) << 3)
  const unsigned sw_addr = scr + l32 * 128 + ((hi | ((l32 & 7) << 1)) << 3);
  // read side: row (lane >> 3) + 8 q, chunk lane & 7 at chunk (lane & 7) ^ (row & 7): address = sr_addr + 1024 q
  const unsigned sr_addr = scr + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);

  f32x16 accA[4][2], accB[4][2];
  bf16x8 xf0[4], wf0[2], xf1[4], wf1[2];
  u32x4 dr_raw[2][4], dr_op[2][4], dr_pre[2][4]; // drain slices in flight (slice parity): staged rows, aux / residual rows, pre-activation rows
  u32x4 dr_bias[2][4];                           // bias of the tile being drained, MFMA layout: [j][g] -> channels 32 j + 8 g + 4 hi + e
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dr_raw[i][q] = u32x4{0u, 0u, 0u, 0u}; dr_op[i][q] = u32x4{0u, 0u, 0u, 0u};
      dr_pre[i][q] = u32x4{0u, 0u, 0u, 0u}; dr_bias[i][q] = u32x4{0u, 0u, 0u, 0u};
    }
  int pm0 = 0, pn0 = 0;                          // coordinates of the tile being drained
  bool have_prev = false;
  const bool do_store = g_knobs[1] == 0;         // development knob (mdm_dev_set_knob 1): skip the drain's stores
  int c_tile = first;
  int c_stage = 0;                               // byte offset of the stage the compute side reads

  T* __restrict__ const Y = reinterpret_cast<T*>(p.y);
  T* __restrict__ const Ypre = reinterpret_cast<T*>(p.ypre);
  const T* __restrict__ const OP = reinterpret_cast<const T*>(ACT == 2 ? p.aux : p.res);

#define MDX_READ_FRAGS(XF, WF, S, STG)                                                                      \
  {                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) WF[j] = *reinterpret_cast<const bf16x8*>(smem + (STG) + wa[(S)] + j * 4096); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) XF[i] = *reinterpret_cast<const bf16x8*>(smem + (STG) + xa[(S)] + i * 4096); \
  }
#define MDX_MFMA_STEP(CUR, XF, WF)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      CUR[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[j], XF[i], CUR[i][j], 0, 0, 0);
#define MDX_MFMA_STEP_Z(CUR, XF, WF)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      CUR[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[j], XF[i], zero16, 0, 0, 0);

  // address of the drain's 16-byte chunk q of slice D (element offset in the [M][Cout] tensors)
#define MDX_ROW_OFF(D, q) ((size_t)(pm0 + wm * 128 + (D) * 32 + (lane >> 3) + 8 * (q)) * p.Cout + pn0 + wn * 64 + (lane & 7) * 8)

  // drain, stage 1 of slice D (= 32-row block D of PRV): accumulators + bias -> bf16 -> scratch (column half J), then
  // whole rows back into dr_raw
#define MDX_DRAIN_W(PRV, D, J)                                                                              \
  {                                                                                                         \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
      const u32x4 bv = dr_bias[(J)][g];                                                                     \
      u32x2 v;                                                                                              \
      v.x = pack_bf16x2(acc_read(PRV[(D)][(J)][4 * g + 0]) + __uint_as_float(bv.x),                         \
                        acc_read(PRV[(D)][(J)][4 * g + 1]) + __uint_as_float(bv.y));                        \
      v.y = pack_bf16x2(acc_read(PRV[(D)][(J)][4 * g + 2]) + __uint_as_float(bv.z),                         \
                        acc_read(PRV[(D)][(J)][4 * g + 3]) + __uint_as_float(bv.w));                        \
      lds_write_b64_asm(sw_addr ^ (unsigned)((8 * (J) + 2 * g) << 3), v);                                   \
    }                                                                                                       \
  }
#define MDX_DRAIN_R(D)                                                                                      \
  {                                                                                                         \
    dr_raw[(D) & 1][0] = lds_read_b128_asm<0>(sr_addr);                                                     \
    dr_raw[(D) & 1][1] = lds_read_b128_asm<1024>(sr_addr);                                                  \
    dr_raw[(D) & 1][2] = lds_read_b128_asm<2048>(sr_addr);                                                  \
    dr_raw[(D) & 1][3] = lds_read_b128_asm<3072>(sr_addr);                                                  \
  }
  // the wait that makes the rows of slice D usable (they are operands of the statement: nothing moves across it)
#define MDX_WAIT_ROWS(D, WAITS)                                                                             \
  asm volatile(WAITS : "+v"(dr_raw[(D) & 1][0]), "+v"(dr_raw[(D) & 1][1]), "+v"(dr_raw[(D) & 1][2]),       \
               "+v"(dr_raw[(D) & 1][3])::"memory");
  // ... and the aux / residual rows of slice D, the bias
#define MDX_WAIT_OP(D, WAITS)                                                                               \
  asm volatile(WAITS : "+v"(dr_op[(D) & 1][0]), "+v"(dr_op[(D) & 1][1]), "+v"(dr_op[(D) & 1][2]),          \
               "+v"(dr_op[(D) & 1][3])::"memory");
#define MDX_WAIT_BIAS(WAITS)                                                                                \
  asm volatile(WAITS : "+v"(dr_bias[0][0]), "+v"(dr_bias[0][1]), "+v"(dr_bias[0][2]), "+v"(dr_bias[0][3]), \
               "+v"(dr_bias[1][0]), "+v"(dr_bias[1][1]), "+v"(dr_bias[1][2]), "+v"(dr_bias[1][3])::"memory");
  // drain, stage 2 of slice D: activation / residual on the staged rows, in registers (values already rounded to bf16, as
  // the reference's autocast graph has them); ACT 1 keeps the pre-activation rows for their own store
#define MDX_DRAIN_ACT(D)                                                                                    \
  if constexpr (ACT != 0 || RES) {                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
      u32x4 r = dr_raw[(D) & 1][q];                                                                         \
      if constexpr (ACT == 1) dr_pre[(D) & 1][q] = r;                                                       \
      Chunk<T> c;                                                                                           \
      c.load(reinterpret_cast<const T*>(&r));                                                               \
      if constexpr (ACT == 1) {                                                                             \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] = gelu_f(c.v[e]);                              \
      } else {                                                                                              \
        Chunk<T> ax;                                                                                        \
        ax.load(reinterpret_cast<const T*>(&dr_op[(D) & 1][q]));                                            \
        if constexpr (ACT == 2) {                                                                           \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] *= dgelu_f(ax.v[e]);                         \
        } else {                                                                                            \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] += ax.v[e];                                  \
        }                                                                                                   \
      }                                                                                                     \
      c.store(reinterpret_cast<T*>(&r));                                                                    \
      dr_raw[(D) & 1][q] = r;                                                                               \
    }                                                                                                       \
  }
  // drain, stage 3 of slice D: the stores (issued right after a barrier, before the loader's next piece)
#define MDX_DRAIN_STORE(D)                                                                                  \
  if (do_store) {                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
      const size_t o = MDX_ROW_OFF(D, q);                                                                   \
      if constexpr (ACT == 1) {                                                                             \
        if (Ypre) *reinterpret_cast<u32x4*>(Ypre + o) = dr_pre[(D) & 1][q];                                 \
      }                                                                                                     \
      *reinterpret_cast<u32x4*>(Y + o) = dr_raw[(D) & 1][q];                                                \
    }                                                                                                       \
  }
#define MDX_DRAIN_LOAD_OP(D)                                                                                \
  if constexpr (OPND) {                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                           \
      dr_op[(D) & 1][q] = global_load_b128_asm(OP + MDX_ROW_OFF(D, q));                          \
  }
  // bias of the tile that is accumulating NOW (n-range N0_), MFMA layout; read when that tile drains, one pass later
#define MDX_DRAIN_LOAD_BIAS(N0_)                                                                            \
  {                                                                                                         \
    const float* bsrc = p.bias ? p.bias + (N0_) + wn * 64 + 4 * hi : reinterpret_cast<const float*>(g_zero_page); \
    const int bstep = p.bias ? 1 : 0;                                                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                         \
        dr_bias[j][g] = global_load_b128_asm(bsrc + bstep * (32 * j + 8 * g));                              \
  }

  // One k-tile iteration.  TS = position in the output tile's k-loop: 0 first (accumulators start from zero), 1..5 carry
  // the drain of the previous tile, 6 plain.  Loop-carried: xf0 / wf0 = fragments of k-step 0, the loader state.
  //   region 1: k-steps 0-2 + their fragment prefetch, pieces 3-11 of the loader's k-tile, the drain's LDS pass
  //   barrier : every wave holds all fragments of this k-tile (its stage is free), k-tile g+1 has landed
  //   region 2: the drain's stores / operand loads (BEFORE any piece), loader moves on, pieces 0-2, k-step 3
#define MDX_ITER(CUR, PRV, TS)                                                                              \
  {                                                                                                         \
    const int n_stage = c_stage == 2 * STAGE ? 0 : c_stage + STAGE;                                         \
    {                                                                                                       \
      MDX_KTILE_STATE();                                                                                    \
      if constexpr ((TS) >= 2 && (TS) <= 5) { MDX_DRAIN_ACT(((TS) - 2) & 3); }                              \
      MDX_READ_FRAGS(xf1, wf1, 1, c_stage);                                                                 \
      if constexpr ((TS) == 0) { MDX_MFMA_STEP_Z(CUR, xf0, wf0); } else { MDX_MFMA_STEP(CUR, xf0, wf0); }   \
      MDX_PIECE(3); MDX_PIECE(4); MDX_PIECE(5);                                                             \
      if constexpr ((TS) >= 1 && (TS) <= 4) { MDX_DRAIN_W(PRV, ((TS) - 1) & 3, 0); }                        \
      MDX_READ_FRAGS(xf0, wf0, 2, c_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf1, wf1);                                                                         \
      MDX_PIECE(6); MDX_PIECE(7); MDX_PIECE(8);                                                             \
      if constexpr ((TS) >= 1 && (TS) <= 4) { MDX_DRAIN_W(PRV, ((TS) - 1) & 3, 1); }                        \
      MDX_READ_FRAGS(xf1, wf1, 3, c_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf0, wf0);                                                                         \
      MDX_PIECE(9); MDX_PIECE(10); MDX_PIECE(11);                                                           \
      if constexpr ((TS) >= 1 && (TS) <= 4) { MDX_DRAIN_R(((TS) - 1) & 3); }                                \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if constexpr ((TS) >= 1 && (TS) <= 4) {                                                                 \
      MDX_WAIT_ROWS(((TS) - 1) & 3, "s_waitcnt vmcnt(12) lgkmcnt(0)");                                      \
      if constexpr (OPND) { MDX_WAIT_OP(((TS) - 1) & 3, ""); }                                              \
    } else if constexpr ((TS) == 0) {                                                                       \
      MDX_WAIT_BIAS("s_waitcnt vmcnt(12) lgkmcnt(0)");                                                      \
    } else {                                                                                                \
      if (xknob & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }                                 \
      else { asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); }                                 \
    }                                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if constexpr ((TS) >= 2 && (TS) <= 5) { if (have_prev) { MDX_DRAIN_STORE(((TS) - 2) & 3); } }           \
    if constexpr ((TS) == 5) { MDX_DRAIN_LOAD_BIAS(tile_n0(c_tile)); }                                      \
    if constexpr ((TS) <= 3) { if (have_prev) { MDX_DRAIN_LOAD_OP((TS) & 3); } }                            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    MDX_DMA_ADVANCE();                                                                                      \
    {                                                                                                       \
      MDX_KTILE_STATE();                                                                                    \
      MDX_READ_FRAGS(xf0, wf0, 0, n_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf1, wf1);                                                                         \
      MDX_PIECE(0); MDX_PIECE(1); MDX_PIECE(2);                                                             \
    }                                                                                                       \
    c_stage = n_stage;                                                                                      \
  }
  // the k-loop of one output tile into CUR while PRV (the previous tile) drains
#define MDX_TILE_PASS(CUR, PRV)                                                                             \
  {                                                                                                         \
    MDX_ITER(CUR, PRV, 0);                                                                                  \
    MDX_ITER(CUR, PRV, 1);                                                                                  \
    MDX_ITER(CUR, PRV, 2);                                                                                  \
    MDX_ITER(CUR, PRV, 3);                                                                                  \
    MDX_ITER(CUR, PRV, 4);                                                                                  \
    MDX_ITER(CUR, PRV, 5);                                                                                  \
    for (int t = 6; t < nk; ++t) MDX_ITER(CUR, PRV, 6);                                                     \
    pm0 = tile_m0(c_tile); pn0 = tile_n0(c_tile);                                                           \
    have_prev = true;                                                                                       \
    c_tile += G;                                                                                            \
  }
  // the last tile of the block: nothing left to hide its drain behind
#define MDX_FINAL_SLICE(PRV, D)                                                                             \
  {                                                                                                         \
    MDX_DRAIN_LOAD_OP(D);                                                                                   \
    MDX_DRAIN_W(PRV, D, 0);                                                                                 \
    MDX_DRAIN_W(PRV, D, 1);                                                                                 \
    MDX_DRAIN_R(D);                                                                                         \
    MDX_WAIT_ROWS(D, "s_waitcnt vmcnt(0) lgkmcnt(0)");                                                      \
    if constexpr (OPND) { MDX_WAIT_OP(D, ""); }                                                             \
    MDX_DRAIN_ACT(D);                                                                                       \
    MDX_DRAIN_STORE(D);                                                                                     \
  }
#define MDX_FINAL_DRAIN(PRV)                                                                                \
  {                                                                                                         \
    MDX_WAIT_BIAS("s_waitcnt vmcnt(0)");   /* the bias; the empty pieces issued past the end of the stream */ \
    MDX_FINAL_SLICE(PRV, 0);                                                                                \
    MDX_FINAL_SLICE(PRV, 1);                                                                                \
    MDX_FINAL_SLICE(PRV, 2);                                                                                \
    MDX_FINAL_SLICE(PRV, 3);                                                                                \
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- prologue: k-tiles 0 and 1 of the stream, pieces 0-2 of k-tile 2 ---------------------------------------------
  MDX_DMA_SETUP(first);
  {
    MDX_KTILE_STATE();
#pragma unroll
    for (int q = 0; q < 12; ++q) { MDX_PIECE(q); }
  }
  MDX_DMA_ADVANCE();
  {
    MDX_KTILE_STATE();
#pragma unroll
    for (int q = 0; q < 12; ++q) { MDX_PIECE(q); }
  }
  MDX_DMA_ADVANCE();
  {
    MDX_KTILE_STATE();
    MDX_PIECE(0); MDX_PIECE(1); MDX_PIECE(2);
  }
  asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // k-tile 0 has landed (k-tile 1 and the 3 pieces stay in flight)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  MDX_READ_FRAGS(xf0, wf0, 0, 0);

  for (;;) {
    MDX_TILE_PASS(accA, accB);
    if (c_tile >= tiles_total) { MDX_FINAL_DRAIN(accA); break; }
    MDX_TILE_PASS(accB, accA);
    if (c_tile >= tiles_total) { MDX_FINAL_DRAIN(accB); break; }
  }
#undef MDX_FINAL_DRAIN
#undef MDX_TILE_PASS
#undef MDX_ITER
#undef MDX_DRAIN_LOAD_BIAS
#undef MDX_DRAIN_LOAD_OP
#undef MDX_DRAIN_STORE
#undef MDX_DRAIN_ACT
#undef MDX_FINAL_SLICE
#undef MDX_WAIT_BIAS
#undef MDX_WAIT_OP
#undef MDX_WAIT_ROWS
#undef MDX_DRAIN_R
#undef MDX_DRAIN_W
#undef MDX_ROW_OFF
#undef MDX_MFMA_STEP_Z
#undef MDX_MFMA_STEP
#undef MDX_READ_FRAGS
#undef MDX_DMA_ADVANCE
#undef MDX_PIECE
#undef MDX_KTILE_STATE
#undef MDX_BLDS
#undef MDX_DMA_SETUP
#endif
}

}  // namespace mdm
