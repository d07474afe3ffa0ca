// Streaming (HBM-bound) helper kernels around the conv / attention core:
// layout conversion at the model boundary (reference tensors are NCHW fp32,
// unet.py:971-987), skip-concat (unet.py:545-547), nearest 2x upsample
// (unet.py:567-569), SiLU on the time embedding (unet.py:227,844), the sinusoidal
// timestep embedding (unet.py:834-839), masked mean of the text states
// (unet.py:854-861) and dtype casts.  All use 16-byte accesses where the layout
// allows; everything computes in fp32.
#include "common.hpp"
#include "../../include/mdm_hip.h"

namespace mdm {

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W,
                                    int Cpad) {
  const size_t total = (size_t)N * H * W * Cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const size_t pix = i / Cpad;
    const int hw = (int)(pix % ((size_t)H * W));
    const int n = (int)(pix / ((size_t)H * W));
    dst[i] = from_f32<T>(c < C ? src[((size_t)n * C + c) * H * W + hw] : 0.f);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int Cs) {
  const size_t total = (size_t)N * C * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int hw = (int)(i % ((size_t)H * W));
    const size_t r = i / ((size_t)H * W);
    const int c = (int)(r % C), n = (int)(r / C);
    dst[i] = to_f32(src[((size_t)n * H * W + hw) * Cs + c]);
  }
}

// out[m, 0:C1] = a[m, :], out[m, C1:C1+C2] = b[m, :]   (dir = 0)
// a[m, :] = out[m, 0:C1], b[m, :] = out[m, C1:]         (dir = 1, the backward split)
template <typename T>
__global__ void concat_kernel(T* a, T* b, T* out, size_t M, int C1, int C2, int dir) {
  constexpr int EPV = Tr<T>::EPV;
  const int k1 = C1 / EPV, k = (C1 + C2) / EPV;
  const size_t total = M * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % k);
    const size_t m = i / k;
    T* part = cc < k1 ? a + (m * k1 + cc) * EPV : b + (m * (k - k1) + (cc - k1)) * EPV;
    uint4* po = reinterpret_cast<uint4*>(out + i * EPV);
    uint4* pp = reinterpret_cast<uint4*>(part);
    if (dir == 0) *po = *pp; else *pp = *po;
  }
}

// nearest-neighbour 2x upsample: y[n, 2h+a, 2w+b, :] = x[n, h, w, :]
template <typename T>
__global__ void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  constexpr int EPV = Tr<T>::EPV;
  const int k = C / EPV;
  const size_t total = (size_t)N * 4 * H * W * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % k);
    size_t p = i / k;
    const int ow = (int)(p % (2 * W)); p /= (2 * W);
    const int oh = (int)(p % (2 * H));
    const int n = (int)(p / (2 * H));
    const size_t s = (((size_t)n * H + (oh >> 1)) * W + (ow >> 1)) * k + cc;
    reinterpret_cast<uint4*>(y)[i] = reinterpret_cast<const uint4*>(x)[s];
  }
}
// 2x2 pixel blocks -> channels: y[n, h, w, (2a + b) * C + c] = x[n, 2h + a, 2w + b, c]
template <typename T>
__global__ void space_to_depth2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  constexpr int EPV = Tr<T>::EPV;
  const int k = C / EPV;
  const size_t total = (size_t)N * 4 * H * W * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % k);
    size_t p = i / k;
    const int ph = (int)(p % 4); p /= 4;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    const size_t s = ((((size_t)n * 2 * H + 2 * h + (ph >> 1)) * 2 * W) + 2 * w + (ph & 1)) * k + cc;
    reinterpret_cast<uint4*>(y)[i] = reinterpret_cast<const uint4*>(x)[s];
  }
}
// backward of the above: dx[n, h, w, :] = sum of the 2x2 block of dy
template <typename T>
__global__ void downsum2x_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
  constexpr int EPV = Tr<T>::EPV;
  const int k = C / EPV;
  const size_t total = (size_t)N * H * W * k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % k);
    size_t p = i / k;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    Chunk<T> acc, t;
    const size_t base = (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + (size_t)cc * EPV;
    acc.load(dy + base);
    t.load(dy + base + C);
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc.v[e] += t.v[e];
    t.load(dy + base + (size_t)2 * W * C);
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc.v[e] += t.v[e];
    t.load(dy + base + (size_t)2 * W * C + C);
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc.v[e] += t.v[e];
    acc.store(dx + i * EPV);
  }
}

// op: 0 out = silu(a); 1 out = b * silu'(a)  (a = pre-activation, b = upstream grad);
//     2 out = a + b;   3 out = a (copy / cast source == dest type)
template <typename T, int OP>
__global__ void ew_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = to_f32(a[i]);
    float r;
    if (OP == 0) r = silu_f(x);
    else if (OP == 1) r = to_f32(b[i]) * dsilu_f(x);
    else if (OP == 2) r = x + to_f32(b[i]);
    else r = x;
    out[i] = from_f32<T>(r);
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ a, TO* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = from_f32<TO>(to_f32(a[i]));
}

// out[b, 0:half] = sin(t[b] * f[i]); out[b, half:2half] = cos(t[b] * f[i])
template <typename T>
__global__ void sincos_kernel(const float* __restrict__ t, const float* __restrict__ freq, T* __restrict__ out, int B,
                              int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float a = t[b] * freq[j];
  out[(size_t)b * 2 * half + j] = from_f32<T>(sinf(a));
  out[(size_t)b * 2 * half + half + j] = from_f32<T>(cosf(a));
}

// y[b, d] = sum_s m[b,s] x[b,s,d] / sum_s m[b,s]   (mask == null -> plain mean)
template <typename T>
__global__ void masked_mean_kernel(const T* __restrict__ x, const float* __restrict__ mask, T* __restrict__ y, int B,
                                   int S, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  float acc = 0.f, den = 0.f;
  for (int s = 0; s < S; ++s) {
    const float m = mask ? mask[b * S + s] : 1.f;
    acc += m * to_f32(x[((size_t)b * S + s) * D + d]);
    den += m;
  }
  y[i] = from_f32<T>(acc / den);
}
// dx[b, s, d] (+)= m[b,s] * dy[b,d] / sum_s m[b,s];  accumulate != 0 adds into dx
template <typename T>
__global__ void masked_mean_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ mask, T* __restrict__ dx,
                                       int B, int S, int D, int accumulate) {
  const size_t total = (size_t)B * S * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const size_t bs = i / D;
    const int b = (int)(bs / S);
    float den = 0.f;
    if (mask) { for (int s = 0; s < S; ++s) den += mask[b * S + s]; } else den = (float)S;
    const float m = mask ? mask[bs] : 1.f;
    float v = m * to_f32(dy[(size_t)b * D + d]) / den;
    if (accumulate) v += to_f32(dx[i]);
    dx[i] = from_f32<T>(v);
  }
}

}  // namespace mdm

using namespace mdm;

static inline int ew_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

#define MDM_DISPATCH_T(dtype, ...)                                                  \
  if ((dtype) == DT_F32) { typedef float TT; __VA_ARGS__; }                         \
  else if ((dtype) == DT_BF16) { typedef bf16 TT; __VA_ARGS__; }                    \
  else { MDM_CHECK_ARG(false); }

extern "C" int mdm_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int Cpad, int dtype,
                                void* stream) {
  MDM_CHECK_ARG(src && dst && Cpad >= C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * H * W * Cpad;
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, src, (TT*)dst, N, C, H, W, Cpad));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int Cs, int dtype,
                                void* stream) {
  MDM_CHECK_ARG(src && dst && Cs >= C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * C * H * W;
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (const TT*)src, dst, N, C, H, W, Cs));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_concat(void* a, void* b, void* out, size_t M, int C1, int C2, int dir, int dtype, void* stream) {
  MDM_CHECK_ARG(a && b && out);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C1 % epv == 0 && C2 % epv == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = M * ((C1 + C2) / epv);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(concat_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (TT*)a, (TT*)b, (TT*)out, M, C1, C2, dir));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_upsample2x(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  MDM_CHECK_ARG(x && y);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * 4 * H * W * (C / epv);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(upsample2x_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (const TT*)x, (TT*)y, N, H, W, C));
  MDM_LAUNCH_STATUS();
}

// y [N, H, W, 4C] (2x2 blocks, channel (2a + b, c)) from x [N, 2H, 2W, C]
extern "C" int mdm_space_to_depth2x(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  MDM_CHECK_ARG(x && y);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * 4 * H * W * (C / epv);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(space_to_depth2x_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (const TT*)x, (TT*)y, N, H, W, C));
  MDM_LAUNCH_STATUS();
}

// dx [N,H,W,C] = 2x2 block sums of dy [N,2H,2W,C]
extern "C" int mdm_downsum2x(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream) {
  MDM_CHECK_ARG(dy && dx);
  const int epv = dtype == DT_F32 ? 4 : 8;
  MDM_CHECK_ARG(C % epv == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * H * W * (C / epv);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(downsum2x_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (const TT*)dy, (TT*)dx, N, H, W, C));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_elementwise(const void* a, const void* b, void* out, size_t n, int op, int dtype, void* stream) {
  MDM_CHECK_ARG(a && out && op >= 0 && op <= 3);
  MDM_CHECK_ARG(!(op == 1 || op == 2) || b);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = ew_blocks(n);
  MDM_DISPATCH_T(dtype,
    if (op == 0) hipLaunchKernelGGL((ew_kernel<TT, 0>), dim3(nb), dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)out, n);
    else if (op == 1) hipLaunchKernelGGL((ew_kernel<TT, 1>), dim3(nb), dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)out, n);
    else if (op == 2) hipLaunchKernelGGL((ew_kernel<TT, 2>), dim3(nb), dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)out, n);
    else hipLaunchKernelGGL((ew_kernel<TT, 3>), dim3(nb), dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)out, n));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_cast(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, void* stream) {
  MDM_CHECK_ARG(src && dst);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = ew_blocks(n);
  if (src_dtype == DT_F32 && dst_dtype == DT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(nb), dim3(256), 0, st, (const float*)src, (bf16*)dst, n);
  else if (src_dtype == DT_BF16 && dst_dtype == DT_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(nb), dim3(256), 0, st, (const bf16*)src, (float*)dst, n);
  else if (src_dtype == DT_F32 && dst_dtype == DT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (src_dtype == DT_BF16 && dst_dtype == DT_BF16) hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)src, (bf16*)dst, n);
  else MDM_CHECK_ARG(false);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_sincos_emb(const float* times, const float* freqs, void* out, int B, int half, int dtype,
                              void* stream) {
  MDM_CHECK_ARG(times && freqs && out);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(sincos_kernel<TT>, dim3((B * half + 255) / 256), dim3(256), 0, st, times, freqs, (TT*)out, B, half));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_masked_mean(const void* x, const float* mask, void* y, int B, int S, int D, int dtype,
                               void* stream) {
  MDM_CHECK_ARG(x && y);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(masked_mean_kernel<TT>, dim3((B * D + 255) / 256), dim3(256), 0, st, (const TT*)x, mask, (TT*)y, B, S, D));
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_masked_mean_bwd(const void* dy, const float* mask, void* dx, int B, int S, int D, int accumulate,
                                   int dtype, void* stream) {
  MDM_CHECK_ARG(dy && dx);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * S * D;
  MDM_DISPATCH_T(dtype, hipLaunchKernelGGL(masked_mean_bwd_kernel<TT>, dim3(ew_blocks(total)), dim3(256), 0, st, (const TT*)dy, mask, (TT*)dx, B, S, D, accumulate));
  MDM_LAUNCH_STATUS();
}

// ---- library-wide state -------------------------------------------------------------
#include <stdio.h>
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void mdm_set_error(const char* file, int line, const char* what) {
  const char* base = strrchr(file, '/');
  snprintf(g_err, sizeof(g_err), "%s:%d: %s", base ? base + 1 : file, line, what);
}
extern "C" const char* mdm_last_error(void) { return g_err; }
extern "C" int mdm_abi_version(void) { return MDM_HIP_ABI_VERSION; }
