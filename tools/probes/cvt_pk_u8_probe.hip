// What does v_cvt_pk_u8_f32 do with fractions, negatives, > 255 and NaN?  (round 6: the byte code of gelu' in the FFN epilogue)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cvt_pk_u8_probe.hip -o /tmp/cvt_probe && /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float* in, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
  const float vals[] = {0.f, 0.49f, 0.5f, 0.51f, 1.5f, 2.5f, 2.51f, 3.5f, 27.99f, 254.5f, 254.51f, 255.5f, 300.f, -0.4f, -0.6f, -5.f, NAN, INFINITY};
  const int n = sizeof(vals) / sizeof(float);
  float* d; unsigned* o; unsigned h[64];
  hipMalloc(&d, sizeof(vals)); hipMalloc(&o, n * 4);
  hipMemcpy(d, vals, sizeof(vals), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%g -> %u\n", vals[i], h[i]);
  return 0;
}
