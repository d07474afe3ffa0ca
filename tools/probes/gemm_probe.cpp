// Development probe (no torch, starts in a second on a fresh GPU box): the forward / input-gradient GEMM entry
// mdm_conv_fwd through the C ABI on the U-Net's layer shapes, conv_gemm_x_kernel (dev knob 3 = 2) against
// conv_gemm_bl_kernel (knob 3 = 1) -- same operands, outputs compared element by element, both timed with HIP events --
// plus a naive fp32 reference on sampled rows.
//   hipcc -O2 tools/probes/gemm_probe.cpp -Iinclude -Lml-mdm_amd/mdm_hip -lmdm_hip -Wl,-rpath,$PWD/ml-mdm_amd/mdm_hip -o /tmp/gemm_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "mdm_hip.h"
#include "mdm_hip_dev.h"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
  } while (0)

typedef __bf16 bf16;

__device__ __host__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void fill_bf16(bf16* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
  p[i] = (bf16)(((float)(h & 0xffff) / 32768.f - 1.f) * scale);
}
__global__ void fill_f32(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = hash32((uint32_t)i * 2654435761u + seed);
  p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * scale;
}
// element-wise comparison: max |a - b| / (|b| + floor), number of elements beyond tol
__global__ void compare_bf16(const bf16* a, const bf16* b, size_t n, float tol, float floor_, unsigned long long* bad, float* maxrel) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = (float)a[i], y = (float)b[i];
  const float r = fabsf(x - y) / (fabsf(y) + floor_);
  if (!(r <= tol)) atomicAdd(bad, 1ull);
  atomicMax(reinterpret_cast<int*>(maxrel), __float_as_int(r));   // r >= 0: integer order == float order
}
// naive reference of sampled output rows (every `rstep`-th row): same k-order-free fp32 sum, same rounding points
__global__ void ref_rows(const bf16* x, const bf16* w, const float* bias, const bf16* res, const bf16* aux, bf16* y, bf16* ypre,
                         int N, int H, int W, int Cin, int Cout, int ks, int act, int kblk, int rstep) {
  const int row = blockIdx.x * rstep, co = blockIdx.y * blockDim.x + threadIdx.x;
  if (co >= Cout) return;
  const int hw = H * W, n = row / hw, r = row % hw, oh = r / W, ow = r % W;
  const int K = ks * ks * Cin;
  float s = 0.f;
  for (int tap = 0; tap < ks * ks; ++tap) {
    const int ih = ks == 3 ? oh + tap / 3 - 1 : oh, iw = ks == 3 ? ow + tap % 3 - 1 : ow;
    if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
    const bf16* xp = x + ((size_t)(n * H + ih) * W + iw) * Cin;
    for (int c = 0; c < Cin; ++c) {
      const int k = ks == 1 ? c : (kblk ? (c / 64) * 9 * 64 + tap * 64 + c % 64 : tap * Cin + c);
      s += (float)xp[c] * (float)w[(size_t)co * K + k];
    }
  }
  if (bias) s += bias[co];
  const size_t o = (size_t)row * Cout + co;
  float v = (float)(bf16)s;
  if (act == 1) {
    if (ypre) ypre[o] = (bf16)v;
    v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
  } else if (act == 2) {
    const float z = (float)aux[o];
    v *= 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
  }
  if (res) v += (float)res[o];
  y[o] = (bf16)v;
}
__global__ void compare_rows(const bf16* a, const bf16* ref, int M, int Cout, int rstep, float tol, float floor_,
                             unsigned long long* bad, float* maxrel) {
  const int row = blockIdx.x * rstep, co = blockIdx.y * blockDim.x + threadIdx.x;
  if (co >= Cout || row >= M) return;
  const size_t o = (size_t)row * Cout + co;
  const float x = (float)a[o], y = (float)ref[o];
  const float r = fabsf(x - y) / (fabsf(y) + floor_);
  if (!(r <= tol)) atomicAdd(bad, 1ull);
  atomicMax(reinterpret_cast<int*>(maxrel), __float_as_int(r));
}

struct Shape { const char* name; int H, Cin, Cout, ks, act, res; };

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  const char* only = argc > 3 ? argv[3] : nullptr;
  const Shape shapes[] = {
      {"1x1 768->3072 @16", 16, 768, 3072, 1, 0, 0},      {"1x1 768->3072 @16 gelu", 16, 768, 3072, 1, 1, 0},
      {"1x1 768->3072 @16 dgelu", 16, 768, 3072, 1, 2, 0}, {"1x1 768->2304 @16", 16, 768, 2304, 1, 0, 0},
      {"1x1 3072->768 @16 +res", 16, 3072, 768, 1, 0, 1},  {"1x1 768->768 @16 +res", 16, 768, 768, 1, 0, 1},
      {"1x1 2304->768 @16", 16, 2304, 768, 1, 0, 0},       {"1x1 512->2048 @32 gelu", 32, 512, 2048, 1, 1, 0},
      {"1x1 512->2048 @32 dgelu", 32, 512, 2048, 1, 2, 0}, {"1x1 2048->512 @32 +res", 32, 2048, 512, 1, 0, 1},
      {"1x1 512->1536 @32", 32, 512, 1536, 1, 0, 0},       {"3x3 256->256 @64", 64, 256, 256, 3, 0, 0},
      {"3x3 256->256 @64 +res", 64, 256, 256, 3, 0, 1},    {"3x3 512->512 @32", 32, 512, 512, 3, 0, 0},
      {"3x3 768->768 @16 +res", 16, 768, 768, 3, 0, 1},    {"3x3 1536->768 @16", 16, 1536, 768, 3, 0, 0},
      {"3x3 512->256 @64", 64, 512, 256, 3, 0, 0},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* d_bad; float* d_max;
  CK(hipMalloc(&d_bad, 8)); CK(hipMalloc(&d_max, 4));
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const int M = B * s.H * s.H, K = s.ks * s.ks * s.Cin;
    const size_t nx = (size_t)M * s.Cin, nw = (size_t)s.Cout * K, ny = (size_t)M * s.Cout;
    // RING > 0 (argv[4]): every launch of the timing loops writes a different output buffer of a ring larger than the
    // 256 MB Infinity Cache -- fresh destination lines, as inside a train step (rewriting one buffer lets the caches
    // absorb the stores)
    const int RING = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<bf16*> ring_y, ring_p;
    bf16 *x, *w, *y[2], *yp[2], *yr, *ypr, *res = nullptr, *aux = nullptr;
    float* bias;
    CK(hipMalloc(&x, nx * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, s.Cout * 4));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&y[v], ny * 2)); CK(hipMalloc(&yp[v], ny * 2)); CK(hipMemset(y[v], 0x7f, ny * 2)); CK(hipMemset(yp[v], 0x7f, ny * 2)); }
    CK(hipMalloc(&yr, ny * 2)); CK(hipMalloc(&ypr, ny * 2));
    for (int r = 0; r < RING; ++r) { bf16 *a_, *b_; CK(hipMalloc(&a_, ny * 2)); CK(hipMalloc(&b_, ny * 2)); ring_y.push_back(a_); ring_p.push_back(b_); }
    int ring_i = 0;
    fill_bf16<<<(nx + 255) / 256, 256, 0, st>>>(x, nx, 1u, 1.f);
    fill_bf16<<<(nw + 255) / 256, 256, 0, st>>>(w, nw, 2u, 1.7f / sqrtf((float)K));
    fill_f32<<<(s.Cout + 255) / 256, 256, 0, st>>>(bias, s.Cout, 3u, 0.5f);
    if (s.res) { CK(hipMalloc(&res, ny * 2)); fill_bf16<<<(ny + 255) / 256, 256, 0, st>>>(res, ny, 4u, 1.f); }
    if (s.act == 2) { CK(hipMalloc(&aux, ny * 2)); fill_bf16<<<(ny + 255) / 256, 256, 0, st>>>(aux, ny, 5u, 2.f); }
    const int kblk = s.ks == 3 ? 64 : 0;
    // variants: 0 = conv_gemm_bl_kernel, 1 = conv_gemm_x_kernel (super-tile order), then timing-only experiments:
    // 2 = x, row-major tile order; 3 = x, LDS-DMA fetches nothing; 4 = x, barriers do not wait for the DMA; 5 = both;
    // 6 = x without the drain's stores
    // 7 = conv_gemm_bl_kernel with the full wait for its stores before the next tile (dev knob 5: the round-3 behaviour)
    // 8 = conv_gemm_bl_kernel whose LDS-DMA fetches nothing, 9 = the same without stores either
    const int NV = 10;
    double us[NV] = {0};
    char kname[2][96];
    // interleaved rounds (variant order rotates; the median of the rounds is reported): a variant timed once, first, right
    // after the allocations runs 5-10 % slower than the same variant timed later
    const int ROUNDS = 5;
    std::vector<double> tv[NV];
    for (int r = 0; r < ROUNDS; ++r)
      for (int vi = 0; vi < NV; ++vi) {
        const int v = (vi + r) % NV;
        mdm_dev_set_knob(3, (v == 0 || v >= 7) ? 1 : 2);
        mdm_dev_set_knob(4, 1);
        mdm_dev_set_knob(0, (v == 3 || v >= 8) ? 1 : v == 4 ? 2 : v == 5 ? 3 : v == 2 ? 4 : 0);
        mdm_dev_set_knob(1, (v == 6 || v == 9) ? 1 : 0);
        bf16* yy = v < 2 ? y[v] : yr;              // experiments write into scratch outputs (the reference rows come later)
        bf16* ypp = v < 2 ? yp[v] : ypr;
        auto run = [&]() {
          bf16* yy_ = yy; bf16* ypp_ = ypp;
          if (RING > 0) { yy_ = ring_y[ring_i % RING]; ypp_ = ring_p[ring_i % RING]; ++ring_i; }
          int rc = mdm_conv_fwd(x, w, bias, res, aux, yy_, s.act == 1 ? ypp_ : nullptr, B, s.H, s.H, s.Cin, s.H, s.H, s.Cout, s.ks, 1, 0,
                                s.act, kblk, 1, st);
          if (rc) { printf("mdm_conv_fwd rc=%d: %s\n", rc, mdm_last_error()); exit(1); }
        };
        run();
        CK(hipStreamSynchronize(st));
        if (v < 2) strncpy(kname[v], mdm_last_gemm_kernel(), 95);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        tv[v].push_back(ms * 1e3 / iters);
      }
    for (int v = 0; v < NV; ++v) { std::sort(tv[v].begin(), tv[v].end()); us[v] = tv[v][ROUNDS / 2]; }
    mdm_dev_set_knob(0, 0); mdm_dev_set_knob(1, 0); mdm_dev_set_knob(4, 1);
    mdm_dev_set_knob(3, 0);
    // X vs the 8-wave kernel, every element
    unsigned long long bad[3] = {0, 0, 0};
    float mr[3] = {0, 0, 0};
    CK(hipMemsetAsync(d_bad, 0, 8, st)); CK(hipMemsetAsync(d_max, 0, 4, st));
    compare_bf16<<<(ny + 255) / 256, 256, 0, st>>>(y[1], y[0], ny, 0.02f, 0.02f, d_bad, d_max);
    CK(hipMemcpyAsync(&bad[0], d_bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&mr[0], d_max, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    if (s.act == 1) {
      CK(hipMemsetAsync(d_bad, 0, 8, st)); CK(hipMemsetAsync(d_max, 0, 4, st));
      compare_bf16<<<(ny + 255) / 256, 256, 0, st>>>(yp[1], yp[0], ny, 0.02f, 0.02f, d_bad, d_max);
      CK(hipMemcpyAsync(&bad[1], d_bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&mr[1], d_max, 4, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    }
    // X vs a naive reference on every 61st row
    const int rstep = 61, nrows = (M + rstep - 1) / rstep;
    ref_rows<<<dim3(nrows, (s.Cout + 127) / 128), 128, 0, st>>>(x, w, bias, res, aux, yr, ypr, B, s.H, s.H, s.Cin, s.Cout, s.ks, s.act, kblk, rstep);
    CK(hipMemsetAsync(d_bad, 0, 8, st)); CK(hipMemsetAsync(d_max, 0, 4, st));
    compare_rows<<<dim3(nrows, (s.Cout + 127) / 128), 128, 0, st>>>(y[1], yr, M, s.Cout, rstep, 0.03f, 0.03f, d_bad, d_max);
    CK(hipMemcpyAsync(&bad[2], d_bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&mr[2], d_max, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    unsigned long long badbl = 0; float mrbl = 0;
    CK(hipMemsetAsync(d_bad, 0, 8, st)); CK(hipMemsetAsync(d_max, 0, 4, st));
    compare_rows<<<dim3(nrows, (s.Cout + 127) / 128), 128, 0, st>>>(y[0], yr, M, s.Cout, rstep, 0.03f, 0.03f, d_bad, d_max);
    CK(hipMemcpyAsync(&badbl, d_bad, 8, hipMemcpyDeviceToHost, st)); CK(hipMemcpyAsync(&mrbl, d_max, 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    const double fl = 2.0 * M * s.Cout * K;
    printf("%-26s M=%-6d N=%-4d K=%-5d | bl %7.1f us %6.0f TF | x %7.1f us %6.0f TF | x/bl %.3f | x-nt-stores %7.1f | noDMA %7.1f | nowait %7.1f | both %7.1f | nostore %7.1f | bl(again) %7.1f | bl noDMA %7.1f | bl noDMA nostore %7.1f | bl vs naive: bad %llu maxrel %.3g | x vs bl: bad %llu maxrel %.3g%s | vs naive: bad %llu maxrel %.3g | %s / %s\n",
           s.name, M, s.Cout, K, us[0], fl / us[0] * 1e-6, us[1], fl / us[1] * 1e-6, us[1] / us[0], us[2], us[3], us[4], us[5], us[6], us[7], us[8], us[9], badbl, mrbl, bad[0], mr[0],
           s.act == 1 ? (bad[1] ? " (ypre BAD)" : " (ypre ok)") : "", bad[2], mr[2], kname[0], kname[1]);
    fflush(stdout);
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(bias)); CK(hipFree(yr)); CK(hipFree(ypr));
    for (int v = 0; v < 2; ++v) { CK(hipFree(y[v])); CK(hipFree(yp[v])); }
    for (int r = 0; r < RING; ++r) { CK(hipFree(ring_y[r])); CK(hipFree(ring_p[r])); }
    if (res) CK(hipFree(res));
    if (aux) CK(hipFree(aux));
  }
  return 0;
}
