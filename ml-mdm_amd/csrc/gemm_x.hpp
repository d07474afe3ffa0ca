// conv_gemm_x_kernel -- the forward / input-gradient implicit GEMM for problems with MANY output tiles per CU (bf16).
// Included by gemm_conv.hip after ConvArgs; same operands, packs, k-order and rounding points as conv_gemm_bl_kernel.
//
// Why a second structure.  conv_gemm_bl_kernel gives a CU to ONE 8-wave block whose registers are all accumulators and
// fragments: while that block runs its epilogue (accumulators -> LDS -> rows -> HBM, plus the activation) nothing on the CU
// issues MFMAs, and the next tile's first LDS-DMA cannot be waited for before the stores have retired (vmcnt is in-order).
// On the 1x1 / FFN GEMMs (K = 512 ... 768: 8-12 k-tiles per output tile) that serial tail is 30-45 % of the launch
// (profiles/r03_gemm_store_phase_cost.txt).  Here the k-tile stream of a block never stops:
//   * 8 waves, two per SIMD (what hides a wave's LDS-DMA issue, scalar blocks and waits: a first version with 4 waves of
//     512 registers, one per SIMD, was correct and 5-50 % slower -- profiles/r04_gemm_x4_one_wave_per_simd_probe.txt);
//     a wave owns a 64 x 64 piece of a 256 x 128 output tile = 2 x 2 blocks of v_mfma_f32_32x32x16_bf16 = 64 accumulator
//     registers, and it has TWO such sets: tile n+1 accumulates into one while tile n is drained from the other, one
//     32 x 32 block per k-tile iteration -- bf16, a 2 KB per-wave LDS scratch that turns the MFMA layout into 64-byte row
//     pieces, activation / residual, global stores -- all of it issued in the gaps between the next tile's MFMAs;
//   * the bias is not added in the drain: the accumulators of a tile START from it (operand C of the first MFMAs);
//   * the LDS ring has 3 stages of one k-tile (64 reduction elements: 256 + 128 rows of 128 bytes = 48 KB) and runs
//     across output tiles: the loader is always 2-3 k-tiles ahead, whatever tile those belong to, so there is no
//     pipeline fill per tile either;
//   * every wait is counted: the barrier of iteration g (between its k-steps 2 and 3) needs k-tile g+1; younger than its
//     pieces are only the drain's stores / loads issued right after the previous barrier (a compile-time number) and the
//     6 pieces of k-tile g+2, so it waits for vmcnt(6 + that number): the stores get two iterations to retire.
// LDS: 3 x 48 KB + 8 x 2 KB = 160 KB.  Host-checked requirements (else conv_gemm_bl_kernel): bf16, 1x1 or channel-block-
// major 3x3, M % 256 == 0, Cout % 128 == 0, K % 64 == 0, K >= 512, plain [M][Cout] output.
#pragma once

namespace mdm {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct XG {
  static constexpr int BM = 256, BN = 128, THREADS = 512;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static constexpr int NSTG = 3, SCR = NSTG * STAGE, SCR_WAVE = 2048, LDS = SCR + 8 * SCR_WAVE;
  static constexpr int PIECES = 6;    // LDS-DMA issues per wave per k-tile: 4 of the activation rows, 2 of the weight rows
  static constexpr int MIN_KTILES = 8;
};

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
// (address = wave-uniform base in SGPRs + 32-bit per-lane byte offset: no 64-bit address registers)
__device__ __forceinline__ u32x4 global_load_b128_asm(const void* base, unsigned byte_off) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(byte_off), "s"(base));
  return r;
}
// lane id recomputed where it is needed (volatile: cannot be hoisted): a lane-constant offset computed at kernel entry and
// used once per tile is otherwise spilled around the k-loop, and the reload's s_waitcnt vmcnt(0) drains the LDS-DMA queue
__device__ __forceinline__ unsigned lane_id_here() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  bf16x2_ t = {(bf16)a, (bf16)b};
  return *reinterpret_cast<uint32_t*>(&t);
}

// Drain schedule of slice d (0..3) of the previous tile, in iterations TS of the current tile's k-loop (DS = 1 when the
// drain reads a second tensor, else 0).  A vector-memory operation issued right after barrier t is only known complete
// behind barrier t + 2 (the counted waits let it fly for two iterations):
//   after barrier d            : aux / residual rows requested
//   iteration d + 1 + DS       : accumulators -> scratch -> row pieces; usable behind its barrier
//   after barrier d + 1        : pre-activation rows stored (ACT 1)
//   iteration d + 2 + DS       : activation / residual in registers
//   after barrier d + 2 + DS   : y rows stored
// vector-memory operations issued right after the barrier of iteration ts, NOT counting the next tile's bias loads
// (issued first after barrier 5 and retired by barrier 6)
template <int ACT, bool OPND>
__host__ __device__ constexpr int xg_post_ops(int ts, bool have_prev) {
  constexpr int DS = OPND ? 1 : 0;
  int n = 0;
  if (have_prev) {
    if (ts - 2 - DS >= 0 && ts - 2 - DS <= 3) n += 2;       // y rows of slice ts - 2 - DS
    if (ACT == 1 && ts >= 1 && ts <= 4) n += 2;              // pre-activation rows of slice ts - 1
    if (OPND && ts <= 3) n += 2;                             // aux / residual rows of slice ts
  }
  return n;
}

// ACT: 0 none, 1 y = gelu(v) (+ the pre-activation to ypre, must be non-null), 2 y = v * gelu'(aux); RES: + residual (ACT 0 only)
template <int MODE, int ACT, bool RES>
__global__ __launch_bounds__(512, 2) void conv_gemm_x_kernel(ConvArgs p) {
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
  // bit 1 = the plain iterations' barriers do not wait for the DMA -- wrong results, they price the memory side of the k-loop
  const int xknob = __builtin_amdgcn_readfirstlane(g_knobs[0]);
  // Tile order: the 32 consecutive tile ids an XCD works on at a time form SA x SB super-tiles (SA row tiles x SB column
  // tiles): SA activation panels + SB weight panels stay in its L2 instead of one activation panel + every weight panel
  const int SB = p.sel_base > 0 ? p.sel_base : tiles_n;     // host: a divisor of tiles_n (ConvArgs::sel_base is unused here)
  const int SA = p.sel_cout > 0 ? p.sel_cout : 1;           // host: a divisor of the row-tile count
  const int sup = SA * SB, sup_per_row = tiles_n / SB;
  auto tile_m0 = [&](int t) { const int s_ = t / sup, w_ = t - s_ * sup; return ((s_ / sup_per_row) * SA + w_ / SB) * BM; };
  auto tile_n0 = [&](int t) { const int s_ = t / sup, w_ = t - s_ * sup; return ((s_ % sup_per_row) * SB + w_ % SB) * BN; };

  // ---- loader: per-lane gather offsets of one output tile (bytes), fixed for its k-loop ------------------------------
  // piece j of a wave = rows 64 j + 8 wave + (lane >> 3) of the tile, 16-byte slot lane & 7 of each row; the XOR swizzle
  // sits on the source side: the lane fetches logical chunk slot ^ f(row), f(row) = (row >> 1) & 7 (conflict-free for
  // the 32-row fragments of the 32x32x16 MFMA, whose ds_read_b128 lane groups span rows {0-3, 12-15, 20-27} / ...)
  const int lrow = tid >> 3;
  const int lchunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const unsigned abias = MODE == MODE_3x3 ? (unsigned)(p.W + 1) * p.Cin * 2u : 0u;
  // descriptor inputs as provably wave-uniform scalars (readfirstlane): under SGPR pressure hipcc keeps uniform values
  // in VGPRs, and a buffer descriptor or scalar offset in a VGPR turns every LDS-DMA into a waterfall loop
  auto uni_ptr = [](const void* q) -> char* {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi_ = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<char*>(((uint64_t)hi_ << 32) | lo);
  };
  char* const a_base = uni_ptr(reinterpret_cast<const char*>(p.x) - abias);
  char* const b_base = uni_ptr(p.w);
  const unsigned a_bytes = (unsigned)p.N * p.H * p.W * p.Cin * 2u + abias;
  const unsigned b_bytes = (unsigned)p.Cout * p.K * 2u;
  unsigned a_voff[4], a_mask[4], b_voff[2];
#define MDX_DMA_SETUP(tile_)                                                                                \
  {                                                                                                         \
    const int tl_ = (tile_);                                                                                \
    const int m0_ = tile_m0(tl_), n0_ = tile_n0(tl_);                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                         \
      const int m = m0_ + lrow + 64 * j;                                                                    \
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
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      b_voff[j] = (xknob & 1) ? INVALID : (unsigned)(n0_ + lrow + 64 * j) * p.K * 2u + lchunk * 16u;        \
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
  if ((q) < 4) {                                                                                            \
    const unsigned vo = (MODE == MODE_1x1 || (a_mask[(q)] & tapbit)) ? a_voff[(q)] : INVALID;               \
    MDX_BLDS(rsA, __builtin_amdgcn_readfirstlane(d_stage + (q) * 8192 + wave_lds), vo, a_soff);             \
  } else {                                                                                                  \
    MDX_BLDS(rsB, __builtin_amdgcn_readfirstlane(d_stage + A_BYTES + ((q) - 4) * 8192 + wave_lds), b_voff[(q) - 4], b_soff); \
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
  // (slot = (2 s + hi) ^ fsw: the address of k-step s is the address of k-step 0 XOR (s << 5) -- one register per operand)
  const int fsw = (l32 >> 1) & 7;
  const int xa0 = (wm * 64 + l32) * 128 + ((hi ^ fsw) << 4);
  const int wa0 = A_BYTES + (wn * 64 + l32) * 128 + ((hi ^ fsw) << 4);
  // drain: per-wave scratch [32 pixel rows][32 channels] bf16 (64-byte rows), 16-byte chunk index XOR ((row >> 1) & 3)
  const unsigned scr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + XG::SCR) + wave * XG::SCR_WAVE;
  // write side: row l32, channels 8 g + 4 hi .. + 3 = chunk g, half hi: address = sw_addr ^ (g << 4)
  const unsigned sw_addr = scr + l32 * 64 + (((l32 >> 1) & 3) << 4) + hi * 8;
  // read side: row (lane >> 2) + 16 q, chunk lane & 3: address = sr_addr + 1024 q
  const unsigned sr_addr = scr + (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 3) & 3)) << 4);

  f32x16 accA[2][2], accB[2][2];
  bf16x8 xf0[2], wf0[2], xf1[2], wf1[2];
  u32x4 dr_raw[2][2], dr_op[3][2];               // drain slices in flight: staged rows (slice % 2), aux / residual rows (slice % 3)
  u32x4 cb[2][4];                                // bias of the NEXT tile as the MFMA's C operand: [j][g] -> channels 32 j + 8 g + 4 hi + e
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) { dr_raw[i][q] = u32x4{0u, 0u, 0u, 0u}; dr_op[i][q] = u32x4{0u, 0u, 0u, 0u}; dr_op[2][q] = u32x4{0u, 0u, 0u, 0u}; }
  const unsigned lane_row_off = (unsigned)(((lane >> 2) * p.Cout + (lane & 3) * 8) * 2);
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
    _Pragma("unroll") for (int j = 0; j < 2; ++j) WF[j] = *reinterpret_cast<const bf16x8*>(smem + (STG) + (wa0 ^ ((S) << 5)) + j * 4096); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) XF[i] = *reinterpret_cast<const bf16x8*>(smem + (STG) + (xa0 ^ ((S) << 5)) + i * 4096); \
  }
#define MDX_MFMA_STEP(CUR, XF, WF)                                                                          \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      CUR[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[j], XF[i], CUR[i][j], 0, 0, 0);
  // first k-step of an output tile: the accumulators start from the bias (cb, loaded one pass earlier)
#define MDX_MFMA_STEP_B(CUR, XF, WF)                                                                        \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                           \
    f32x16 c0;                                                                                              \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
      c0[4 * g + 0] = __uint_as_float(cb[j][g].x); c0[4 * g + 1] = __uint_as_float(cb[j][g].y);             \
      c0[4 * g + 2] = __uint_as_float(cb[j][g].z); c0[4 * g + 3] = __uint_as_float(cb[j][g].w);             \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
      CUR[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(WF[j], XF[i], c0, 0, 0, 0);                       \
  }

  // drain slice SL = 32 x 32 block (i = SL >> 1, j = SL & 1) of the finished tile; element offset of its 16-byte chunk q
  // (BYTE offset in the [M][Cout] tensors = per-lane part + wave-uniform part, 32 bits: host-checked < 2^31)
#define MDX_ROW_OFF(SL, q)                                                                                  \
  (lane_row_off + (unsigned)(((pm0 + wm * 64 + ((SL) >> 1) * 32 + 16 * (q)) * p.Cout + pn0 + wn * 64 + ((SL) & 1) * 32) * 2))
  // stage 1: accumulators -> bf16 -> scratch, then 64-byte row pieces back into dr_raw
#define MDX_DRAIN_WR(PRV, SL)                                                                               \
  {                                                                                                         \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                         \
      u32x2 v;                                                                                              \
      v.x = pack_bf16x2(PRV[(SL) >> 1][(SL) & 1][4 * g + 0], PRV[(SL) >> 1][(SL) & 1][4 * g + 1]);          \
      v.y = pack_bf16x2(PRV[(SL) >> 1][(SL) & 1][4 * g + 2], PRV[(SL) >> 1][(SL) & 1][4 * g + 3]);          \
      lds_write_b64_asm(sw_addr ^ (unsigned)(g << 4), v);                                                   \
    }                                                                                                       \
    dr_raw[(SL) & 1][0] = lds_read_b128_asm<0>(sr_addr);                                                    \
    dr_raw[(SL) & 1][1] = lds_read_b128_asm<1024>(sr_addr);                                                 \
  }
  // the waits that make asm-loaded registers usable (they are operands of the statement: nothing moves across it)
#define MDX_WAIT_ROWS(SL, WAITS) asm volatile(WAITS : "+v"(dr_raw[(SL) & 1][0]), "+v"(dr_raw[(SL) & 1][1])::"memory")
#define MDX_WAIT_ROWS_N(SL, N_) asm volatile("s_waitcnt vmcnt(%c2) lgkmcnt(0)" : "+v"(dr_raw[(SL) & 1][0]), "+v"(dr_raw[(SL) & 1][1]) : "n"(N_) : "memory")
#define MDX_WAIT_OP(SL) asm volatile("" : "+v"(dr_op[(SL) % 3][0]), "+v"(dr_op[(SL) % 3][1])::"memory")
#define MDX_WAIT_BIAS(WAITS)                                                                                \
  asm volatile(WAITS : "+v"(cb[0][0]), "+v"(cb[0][1]), "+v"(cb[0][2]), "+v"(cb[0][3]), "+v"(cb[1][0]),      \
               "+v"(cb[1][1]), "+v"(cb[1][2]), "+v"(cb[1][3])::"memory")
  // stage 2: activation / residual on the staged rows, in registers (values already rounded to bf16, as the reference's
  // autocast graph has them)
#define MDX_DRAIN_ACT(SL)                                                                                   \
  if constexpr (ACT != 0 || RES) {                                                                          \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                         \
      u32x4 r = dr_raw[(SL) & 1][q];                                                                        \
      Chunk<T> c;                                                                                           \
      c.load(reinterpret_cast<const T*>(&r));                                                               \
      if constexpr (ACT == 1) {                                                                             \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] = gelu_t<T>(c.v[e]);                              \
      } else {                                                                                              \
        Chunk<T> ax;                                                                                        \
        ax.load(reinterpret_cast<const T*>(&dr_op[(SL) % 3][q]));                                           \
        if constexpr (ACT == 2) {                                                                           \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] *= dgelu_t<T>(ax.v[e]);                         \
        } else {                                                                                            \
          _Pragma("unroll") for (int e = 0; e < 8; ++e) c.v[e] += ax.v[e];                                  \
        }                                                                                                   \
      }                                                                                                     \
      c.store(reinterpret_cast<T*>(&r));                                                                    \
      dr_raw[(SL) & 1][q] = r;                                                                              \
    }                                                                                                       \
    /* pin the result here: it is only stored inside the `have_prev` block behind the barrier, and hipcc would sink */ \
    /* the whole activation into that block -- out of the MFMAs' shadow                                            */ \
    asm volatile("" : "+v"(dr_raw[(SL) & 1][0]), "+v"(dr_raw[(SL) & 1][1]));                                \
  }
  // stage 3: the stores (issued right after a barrier, before the loader's next piece)
#define MDX_DRAIN_STORE(DST, SL)                                                                            \
  if (do_store) {                                                                                           \
    if (xknob & 4) {   /* development knob 0, bit 2: non-temporal stores */                                 \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                         \
        __builtin_nontemporal_store(dr_raw[(SL) & 1][q], reinterpret_cast<u32x4*>(reinterpret_cast<char*>(DST) + MDX_ROW_OFF(SL, q))); \
    } else {                                                                                                \
      _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                         \
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(DST) + MDX_ROW_OFF(SL, q)) = dr_raw[(SL) & 1][q]; \
    }                                                                                                       \
  }
#define MDX_DRAIN_LOAD_OP(SL)                                                                               \
  if constexpr (OPND) {                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) dr_op[(SL) % 3][q] = global_load_b128_asm(OP, MDX_ROW_OFF(SL, q)); \
  }
  // bias of output tile TILE_ (clamped to the last one), MFMA layout
#define MDX_LOAD_BIAS(TILE_)                                                                                \
  {                                                                                                         \
    const int tb_ = (TILE_) < tiles_total ? (TILE_) : tiles_total - 1;                                      \
    const float* bsrc = p.bias ? p.bias : reinterpret_cast<const float*>(g_zero_page);                      \
    const unsigned boff = p.bias ? (unsigned)((tile_n0(tb_) + wn * 64) * 4) + (lane_id_here() >> 5) * 16u : 0u; \
    const unsigned bstep = p.bias ? 4u : 0u;                                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                         \
        cb[j][g] = global_load_b128_asm(bsrc, boff + bstep * (32 * j + 8 * g));                             \
  }

  // One k-tile iteration.  TS = position in the output tile's k-loop: 0 first (accumulators start from the bias), 1..7
  // carry the drain of the previous tile (schedule above), 8 plain.  Loop-carried: xf0 / wf0 = fragments of k-step 0, the
  // loader state.
  //   region 1: k-steps 0-2 + their fragment prefetch, pieces 2-5 of the loader's k-tile, the drain's LDS pass / activation
  //   barrier : every wave holds all fragments of this k-tile (its stage is free), k-tile g+1 has landed
  //   region 2: the drain's stores / operand loads (BEFORE any piece), loader moves on, pieces 0-1, k-step 3
#define MDX_ITER(CUR, PRV, TS)                                                                              \
  {                                                                                                         \
    constexpr int DS = OPND ? 1 : 0;                                                                        \
    constexpr int S_WR = (TS) - 1 - DS, S_ACT = (TS) - 2 - DS;   /* slices of this iteration's LDS pass / activation */ \
    const int n_stage = c_stage == 2 * STAGE ? 0 : c_stage + STAGE;                                         \
    {                                                                                                       \
      MDX_KTILE_STATE();                                                                                    \
      if constexpr (S_ACT >= 0 && S_ACT <= 3) { MDX_DRAIN_ACT(S_ACT & 3); }                                 \
      MDX_READ_FRAGS(xf1, wf1, 1, c_stage);                                                                 \
      if constexpr ((TS) == 0) { MDX_MFMA_STEP_B(CUR, xf0, wf0); } else { MDX_MFMA_STEP(CUR, xf0, wf0); }   \
      MDX_PIECE(2); MDX_PIECE(3);                                                                           \
      MDX_READ_FRAGS(xf0, wf0, 2, c_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf1, wf1);                                                                         \
      MDX_PIECE(4);                                                                                         \
      if constexpr (S_WR >= 0 && S_WR <= 3) { MDX_DRAIN_WR(PRV, S_WR & 3); }                                \
      MDX_READ_FRAGS(xf1, wf1, 3, c_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf0, wf0);                                                                         \
      MDX_PIECE(5);                                                                                         \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    {                                                                                                       \
      /* younger than k-tile g+1: what the previous iteration issued after its barrier (a pass ends with TS 7 or 8: */ \
      /* nothing; the bias loads of TS 5 come first in their block and are NOT left in flight) + 6 pieces          */ \
      constexpr int TP = (TS) == 0 ? 8 : (TS) - 1;                                                          \
      constexpr int NA = XG::PIECES + xg_post_ops<ACT, OPND>(TP, true), NB = XG::PIECES + xg_post_ops<ACT, OPND>(TP, false); \
      static_assert(NA <= 63 && NB <= 63, "vmcnt field");                                                   \
      if constexpr (S_WR >= 0 && S_WR <= 3) {                                                               \
        if (have_prev) { MDX_WAIT_ROWS_N(S_WR & 3, NA); } else { MDX_WAIT_ROWS_N(S_WR & 3, NB); }           \
      } else if constexpr ((TS) == 8) {                                                                     \
        if (xknob & 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }                               \
        else { asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)" ::"n"(NA) : "memory"); }                      \
      } else {                                                                                              \
        if (have_prev) { asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)" ::"n"(NA) : "memory"); }            \
        else { asm volatile("s_waitcnt vmcnt(%c0) lgkmcnt(0)" ::"n"(NB) : "memory"); }                      \
      }                                                                                                     \
      if constexpr ((TS) == 6) { MDX_WAIT_BIAS(""); }                      /* requested after barrier 5 */  \
      if constexpr (OPND && (TS) >= 2 && (TS) <= 5) { MDX_WAIT_OP(((TS) - 2) & 3); }   /* requested after barrier TS - 2 */ \
    }                                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if constexpr ((TS) == 5) { MDX_LOAD_BIAS(c_tile + G); }                                                 \
    if constexpr (S_ACT >= 0 && S_ACT <= 3) { if (have_prev) { MDX_DRAIN_STORE(Y, S_ACT & 3); } }           \
    if constexpr (ACT == 1 && (TS) >= 1 && (TS) <= 4) { if (have_prev) { MDX_DRAIN_STORE(Ypre, ((TS) - 1) & 3); } } \
    if constexpr ((TS) <= 3) { if (have_prev) { MDX_DRAIN_LOAD_OP((TS) & 3); } }                            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    MDX_DMA_ADVANCE();                                                                                      \
    {                                                                                                       \
      MDX_KTILE_STATE();                                                                                    \
      MDX_READ_FRAGS(xf0, wf0, 0, n_stage);                                                                 \
      MDX_MFMA_STEP(CUR, xf1, wf1);                                                                         \
      MDX_PIECE(0); MDX_PIECE(1);                                                                           \
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
    MDX_ITER(CUR, PRV, 6);                                                                                  \
    MDX_ITER(CUR, PRV, 7);                                                                                  \
    for (int t = 8; t < nk; ++t) MDX_ITER(CUR, PRV, 8);                                                     \
    pm0 = tile_m0(c_tile); pn0 = tile_n0(c_tile);                                                           \
    have_prev = true;                                                                                       \
    c_tile += G;                                                                                            \
  }
  // the last tile of the block: nothing left to hide its drain behind
#define MDX_FINAL_SLICE(PRV, SL)                                                                            \
  {                                                                                                         \
    MDX_DRAIN_LOAD_OP(SL);                                                                                  \
    MDX_DRAIN_WR(PRV, SL);                                                                                  \
    MDX_WAIT_ROWS(SL, "s_waitcnt vmcnt(0) lgkmcnt(0)");                                                     \
    if constexpr (OPND) { MDX_WAIT_OP(SL); }                                                                \
    if constexpr (ACT == 1) { MDX_DRAIN_STORE(Ypre, SL); }                                                  \
    MDX_DRAIN_ACT(SL);                                                                                      \
    MDX_DRAIN_STORE(Y, SL);                                                                                 \
  }
#define MDX_FINAL_DRAIN(PRV)                                                                                \
  {                                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* the empty pieces issued past the end of the stream */ \
    MDX_FINAL_SLICE(PRV, 0);                                                                                \
    MDX_FINAL_SLICE(PRV, 1);                                                                                \
    MDX_FINAL_SLICE(PRV, 2);                                                                                \
    MDX_FINAL_SLICE(PRV, 3);                                                                                \
  }

  // ---- prologue: the first tile's bias, k-tiles 0 and 1 of the stream, pieces 0-1 of k-tile 2 --------------------------
  MDX_LOAD_BIAS(first);
  MDX_DMA_SETUP(first);
  {
    MDX_KTILE_STATE();
#pragma unroll
    for (int q = 0; q < 6; ++q) { MDX_PIECE(q); }
  }
  MDX_DMA_ADVANCE();
  {
    MDX_KTILE_STATE();
#pragma unroll
    for (int q = 0; q < 6; ++q) { MDX_PIECE(q); }
  }
  MDX_DMA_ADVANCE();
  {
    MDX_KTILE_STATE();
    MDX_PIECE(0); MDX_PIECE(1);
  }
  MDX_WAIT_BIAS("s_waitcnt vmcnt(8)");   // the bias and k-tile 0 have landed (k-tile 1 and the 2 pieces stay in flight)
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
#undef MDX_FINAL_SLICE
#undef MDX_TILE_PASS
#undef MDX_ITER
#undef MDX_LOAD_BIAS
#undef MDX_DRAIN_LOAD_OP
#undef MDX_DRAIN_STORE
#undef MDX_DRAIN_ACT
#undef MDX_WAIT_BIAS
#undef MDX_WAIT_OP
#undef MDX_WAIT_ROWS_N
#undef MDX_WAIT_ROWS
#undef MDX_DRAIN_WR
#undef MDX_ROW_OFF
#undef MDX_MFMA_STEP_B
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
