#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(int* out, int mode){
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
  __syncthreads();
  int l = threadIdx.x;
  // hypothesis: lane i of a 16-group supplies row (i/4), col seg (i%4) of a 4x16 block with row stride RS elements
  int g = l >> 4, i = l & 15;
  int RS = 64;  // row stride in elements
  int addr_elems;
  if (mode == 0) addr_elems = g * 16 + (i >> 2) * RS + (i & 3) * 4;        // blocks side by side in columns
  else addr_elems = g * 16 + (i & 3) * RS + (i >> 2) * 4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr_elems));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main(){
  int* d; hipMalloc(&d, 64*4*4); int h[256];
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1,64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j=0;j<4;++j) printf(" %4d (r%d c%d)", h[l*4+j], h[l*4+j]/64, h[l*4+j]%64); printf("\n"); }
  }
  return 0;
}
