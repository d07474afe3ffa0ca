/* libmdm_hip -- development / profiling aids.  NOT part of the drop-in boundary (include/mdm_hip.h): nothing here
 * replaces a reference call site; bench.py and tools/ use these to label per-launch timings. */
#ifndef MDM_HIP_DEV_H_
#define MDM_HIP_DEV_H_
#ifdef __cplusplus
extern "C" {
#endif

/* name (as a profiler prints it, without arguments) of the GEMM-class kernel the calling thread launched last */
const char* mdm_last_gemm_kernel(void);
/* host-only: block tile (BM * 1000 + BN) mdm_conv_fwd will use for (M, Cout, dtype) */
int mdm_conv_fwd_tile(int M, int Cout, int dtype);
/* host-only: 128 or 256 (square output tile edge) mdm_conv_wgrad will use */
int mdm_conv_wgrad_tile(int M, int Cout, int K, int dtype);

/* development knobs of the GEMM kernels (all 0 / default in the product; host-side state of the calling process, never
 * read from the environment inside an entry point).  Experiments recorded in DESIGN.md / profiles/.
 *   0  conv_gemm_bl_kernel: bit 0 = its LDS-DMA fetches nothing (timing only: what the k-loop costs without memory)
 *   1  epilogues skip their global stores (what the store phase costs)
 *   2  force the forward tile of conv_gemm_bl_kernel (128128 / 256192 / 256256)
 *   3, 4  unused (were the switches of conv_gemm_x_kernel, removed in round 6: profiles/r04_gemm_x8_probe.txt is its record)
 *   6  tile-fill percentage below which a forward GEMM may split its reduction (default 80; 25 = the sampling-only rule)
 *   7  1 = the narrow 3x3 convolutions of the nested models go back to the implicit-GEMM kernel (no conv3x3_direct_kernel)
 *   8  1 = the narrow weight gradients (64 output channels, >= 262144 pixels) go back to the split GEMM (no wgrad_direct_kernel)
 *   9  forward split-K: minimum k-tiles per range (default 6);  10: minimum k-tiles of serial walk a split must save (default 16)
 *  11  1 = no 4-stage instantiations of conv_gemm_bl_kernel<128, 128> for under-filled grids (small-batch sampling)
 *  12  forward split-K: blocks per CU a split aims for (0 = by problem size: 1 for M <= 2048, else 2)
 *  13  1 = mdm_conv_wgrad_reduce launches nothing (timing-only ablation: what the slab-reduce launches cost a step; WRONG gradients) */
int mdm_dev_set_knob(int idx, int value);
/* attention backward kernel choice: 0 = by shape, 1 = always the split (dQ + dK/dV) kernels on 16x16x32 MFMAs, 2 = the
 * one-block-per-head kernel whenever the shape allows (tests), 3 = as 2 but the 16x16x32 one, 4 = the streaming kernels on
 * 32x32x16 MFMAs (csrc/attn32.hpp) whenever the shape allows */
int mdm_dev_set_attn_bwd(int mode);
/* GroupNorm backward, two-kernel path (images too large for the register-resident kernel): bytes of x + dy, in MiB,
 * above which the batch is walked in chunks of samples so that the second kernel's reads still find them in the Infinity
 * Cache (default 160; 0 = chunk whenever the batch allows it -- the tests; negative = never) */
int mdm_dev_set_gn_chunk_mb(int mb);
/* bytes per element of the operand a bf16 MDM_ACT_GELU launch leaves for its backward: 1 (the product: the byte code of
 * gelu', include/mdm_hip.h) or 2 (a variant library built with -DMDM_FFN_AUX_BF16 for in-call A/B: the bf16 pre-activation) */
int mdm_dev_ffn_aux_bytes(void);
/* phase time stamps of attn_bwd_small32_kernel: a device buffer of [blocks][8][16] 64-bit words (tools/attn_debug.py), or
 * null (the default) */
int mdm_dev_set_attn_dbg(void* buf);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_DEV_H_ */
