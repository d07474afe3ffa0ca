// Argument block shared by the forward / input-gradient implicit-GEMM kernels (gemm_conv.hip).
#pragma once
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
  int groups;         // grouped launches only (ConvGroup)
  int act;            // 0 none, 1 y=gelu(v), 2 y=v*gelu'(aux)
  int kblk;           // 3x3 k-order: 0 = tap-major (k = tap*Cin + cin); B = channel-block-major,
                      // k = (cin / B) * 9 * B + tap * B + cin % B with B = the k-tile (64 bf16 / 32 fp32)
  // split-K launches (conv_gemm_bl_kernel<..., SPLITK>): the reduction is cut into `ksplit` ranges of `kt_per` k-tiles,
  // range s writes its raw fp32 tile to part[s][M][Cout]; splitk_epilogue_kernel adds them up and applies the epilogue
  float* part;
  int ksplit, kt_per;
  // tap selection (conv_gemm_bl_kernel<..., SEL4>): 4 of the 9 taps per channel block, tap = base + (j & 1) + 3 * (j >> 1)
  //   sel_mode 0: one base for the whole launch (sel_base; 4 = the stride-2 input gradient)
  //   sel_mode 1: base by OUTPUT phase, phase = column tile's n0 / sel_cout (sub-pixel form of upsample2x -> conv3x3:
  //               output columns are (phase, co), phase (ph, pw) reads the low-res taps {ph, ph+1} x {pw, pw+1})
  //   sel_mode 2: base by INPUT phase: the reduction runs over (phase, channel block, tap) of a 2x2-blocked input with
  //               sel_cout channels per phase (input gradient of the same op); base = (1 - ph) * 3 + (1 - pw)
  int sel_mode, sel_base, sel_cout;
  // pixel-shuffled store (ps_cout > 0): output column (phase, co), row (n, bh, bw) of a ps_H x ps_W grid goes to
  // y[n, 2 bh + ph, 2 bw + pw, co] of a [N, 2 ps_H, 2 ps_W, ps_cout] tensor (a column tile lies inside one phase)
  int ps_cout, ps_H, ps_W;
  // GroupNorm of the OUTPUT fused into the epilogue (gn_y != null; 256x192 tile, bf16): a row tile is the 256 pixels of
  // one sample and a column tile 8 whole groups of 24 channels, so the tile holds everything the statistics need.
  // Besides y the launch writes gn_y = act(GroupNorm(y)), the norm's stats [N][G][2] and coef [N][C][2] (what
  // mdm_gn_fwd would have produced from y: the standalone norm kernel and its read of y disappear).
  void* gn_y;
  const float* gn_gamma;
  const float* gn_beta;
  float* gn_stats;
  float* gn_coef;
  float gn_eps;
  int gn_act, gn_groups;
  // development flags (include/mdm_hip_dev.h knobs 0 and 1; always 0 in the product), filled in by the launch helpers.
  // They travel in the ARGUMENT block -- a wave-uniform scalar load at kernel start -- and NOT in a __device__ variable:
  // rounds 3-5 read `g_knobs[1]` (a mutable global, hence a VECTOR load) inside the epilogue's store loop, and the
  // `s_waitcnt vmcnt(0)` in front of its use made every chunk's store wait for the acknowledgement of the previous one
  // (round 6; HISTORY.md section 4.1)
  //   bit 0: the epilogue skips its global stores (timing only);  bit 1: the LDS-DMA fetches nothing (timing only)
  int dev_flags;
};

enum { MODE_1x1 = 0, MODE_3x3 = 1, MODE_3x3_T2 = 2 };

}  // namespace mdm
