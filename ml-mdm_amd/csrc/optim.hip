// Optimizer tail of the train step as two streaming passes over flat fp32 arenas.
//
// Reference semantics (ml-mdm-matryoshka/ml_mdm/):
//   trainer.py:52-58,79-86      total_norm = clip_grad_norm_(params, clip); optimizer.step()
//   clis/train_parallel.py:122  AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
//   models/model_ema.py:25-34   ema = ema * d + p * (1 - d)
//   trainer.py:92               optimizer.zero_grad()
// In the reference this is ~10 full passes over 1.85 GB executed as thousands of per-tensor kernels.
// Here: (1) sum of squares of the gradient arena -> one device scalar (no host sync);
//       (2) one fused pass: clip-scale, AdamW moments + update, EMA, and zeroing of the gradient arena
//           for the next step: 5 reads + 5 writes of 4 bytes per parameter = algorithmic minimum.
// HBM-bound; 16-byte accesses; deterministic (fixed block partition of the reduction).
#include "common.hpp"

namespace mdm {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, float* __restrict__ part,
                                                            size_t n4) {
  __shared__ float sh[4];
  float s = 0.f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ part, int nparts,
                                                          const float* __restrict__ g, size_t n, size_t tail_from,
                                                          float* __restrict__ out, int* __restrict__ step_counter) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  for (size_t i = tail_from + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float q = sh[0] + sh[1] + sh[2] + sh[3];
    out[0] = q;
    // the optimizer's step number lives on the device and advances only when the update will be applied
    if (step_counter && q == q && q <= 3.0e38f) step_counter[0] += 1;
  }
}

struct AdamArgs {
  float* p; float* g; float* m; float* v; float* ema;
  const float* gnorm_sq;   // device scalar (sum of squares of g) or null = no clipping
  const int* step_dev;     // device step number (1-based) or null = bc1 / bc2 computed on the host
  size_t n;
  float lr, beta1, beta2, eps, wd, bc1, bc2, clip, ema_decay;
  int zero_grad;
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float& e, const AdamArgs& a, float gs,
                                         bool has_ema) {
  const float gr = g * gs;
  p *= (1.f - a.lr * a.wd);
  m = a.beta1 * m + (1.f - a.beta1) * gr;
  v = a.beta2 * v + (1.f - a.beta2) * gr * gr;
  const float denom = sqrtf(v) / sqrtf(a.bc2) + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
  if (has_ema) e = e * a.ema_decay + p * (1.f - a.ema_decay);
  if (a.zero_grad) g = 0.f;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(AdamArgs a) {
  if (a.step_dev) {
    const float st = (float)a.step_dev[0];
    a.bc1 = 1.f - powf(a.beta1, st);
    a.bc2 = 1.f - powf(a.beta2, st);
  }
  float gs = 1.f;
  bool poisoned = false;
  if (a.gnorm_sq) {
    const float q = a.gnorm_sq[0];
    poisoned = !(q == q) || q > 3.0e38f;       // NaN / inf gradient norm: one bad sample must not wipe out p, m, v, ema
    const float nrm = sqrtf(q);
    gs = fminf(a.clip / (nrm + 1e-6f), 1.f);   // torch.nn.utils.clip_grad_norm_ coefficient
  }
  const bool has_ema = a.ema != nullptr;
  const size_t n4 = a.n / 4;
  f32x4* p4 = reinterpret_cast<f32x4*>(a.p);
  f32x4* g4 = reinterpret_cast<f32x4*>(a.g);
  if (poisoned) {   // skip the update, only clear the gradients
    if (a.zero_grad) {
      for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) g4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < a.n; i += 256) a.g[i] = 0.f;
    }
    return;
  }
  f32x4* m4 = reinterpret_cast<f32x4*>(a.m);
  f32x4* v4 = reinterpret_cast<f32x4*>(a.v);
  f32x4* e4 = reinterpret_cast<f32x4*>(a.ema);
  // (Round 6: software-pipelining the trips -- the next trip's five vectors requested before this trip's stores, so that the
  // wait for them is not also a wait for the stores' acknowledgement on gfx950's single in-order vmcnt -- was 5 % SLOWER:
  // 3.65 vs 3.47 ms on 415 M parameters.  With 8 waves per SIMD resident the other waves cover that round trip already.)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
    f32x4 e = has_ema ? e4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float pp = p[k], gg = g[k], mm = m[k], vv = v[k], ee = e[k];
      adam_one(pp, gg, mm, vv, ee, a, gs, has_ema);
      p[k] = pp; g[k] = gg; m[k] = mm; v[k] = vv; e[k] = ee;
    }
    p4[i] = p; m4[i] = m; v4[i] = v;
    if (has_ema) e4[i] = e;
    if (a.zero_grad) g4[i] = g;
  }
  // scalar tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (size_t i = n4 * 4 + threadIdx.x; i < a.n; i += 256) {
      float e = has_ema ? a.ema[i] : 0.f;
      adam_one(a.p[i], a.g[i], a.m[i], a.v[i], e, a, gs, has_ema);
      if (has_ema) a.ema[i] = e;
    }
  }
}

}  // namespace mdm

using namespace mdm;

// out[0] = sum_i g[i]^2.  ws: fp32 [1024].  step_counter (device int, may be NULL) += 1 when the sum is finite.
extern "C" int mdm_sumsq(const float* g, float* out, float* ws, size_t n, int* step_counter, void* stream) {
  MDM_CHECK_ARG(g && out && ws);
  MDM_CHECK_ARG(((size_t)g & 15) == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t n4 = n / 4;
  int nb = (int)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, g, ws, n4);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, ws, nb, g, n, n4 * 4, out, step_counter);
  MDM_LAUNCH_STATUS();
}

// One fused optimizer step over flat fp32 arenas of n elements (all 16-byte aligned):
//   gs = gnorm_sq ? min(1, clip / (sqrt(*gnorm_sq) + 1e-6)) : 1          (gradient-norm clipping)
//   AdamW(lr, beta1, beta2, eps, weight_decay) on (p, g*gs, m, v) for step number `step` (1-based, bias correction), or
//   for the device-resident step number *step_dev when that is given (mdm_sumsq advances it only for finite norms)
//   ema = ema * ema_decay + p * (1 - ema_decay)   (skipped when ema == NULL)
//   g = 0 when zero_grad != 0
extern "C" int mdm_adamw_ema_step(float* p, float* g, float* m, float* v, float* ema, const float* gnorm_sq, size_t n,
                                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  const int* step_dev, float clip, float ema_decay, int zero_grad, void* stream) {
  MDM_CHECK_ARG(p && g && m && v && (step >= 1 || step_dev));
  MDM_CHECK_ARG((((size_t)p | (size_t)g | (size_t)m | (size_t)v | (size_t)ema) & 15) == 0);
  AdamArgs a;
  a.p = p; a.g = g; a.m = m; a.v = v; a.ema = ema; a.gnorm_sq = gnorm_sq; a.n = n;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.step_dev = step_dev;
  a.bc1 = 1.f - powf(beta1, (float)(step >= 1 ? step : 1));
  a.bc2 = 1.f - powf(beta2, (float)(step >= 1 ? step : 1));
  a.clip = clip; a.ema_decay = ema_decay; a.zero_grad = zero_grad;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t n4 = n / 4;
  int nb = (int)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(nb), dim3(256), 0, st, a);
  MDM_LAUNCH_STATUS();
}
