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

/* development knobs of the GEMM kernels (all 0 in the product): 1 = epilogues skip their global stores (measures what
 * the store phase costs), 2 = force the forward tile (128128 / 256192 / 256256).  Experiments recorded in DESIGN.md. */
int mdm_dev_set_knob(int idx, int value);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_DEV_H_ */
