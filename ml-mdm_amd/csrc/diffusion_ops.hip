// Per-pixel arithmetic AROUND the denoiser: the sampling update, the train-side noising / loss, the image pyramid,
// the per-sample input normalisation of a nested level and the uint8 input stage.  (SURVEY.md section 8f rows N1,
// N3, N4 and row a7's x / std.)  In the reference each of these is a chain of 5-40 elementwise ATen launches on
// [B, 3, H, W] fp32 images; here each is one streaming kernel (16-byte accesses, fp32 math, NCHW fp32 at the model
// boundary like the reference).
//
// Reference semantics (paths relative to ml-mdm-matryoshka/ml_mdm/):
//   samplers.py:281-345   get_prediction_xt_last: v/eps -> x0, clip, DDPM posterior mean or DDIM(eta) step, noise
//   samplers.py:445-456   classifier-free guidance combine  p = p_u + w (p_c - p_u)
//   samplers.py:461-508   clip_sample: CLIP = clamp(x0 s, -1, 1) / s; DYNAMIC = clamp(x0 s, -q, q) / q / s
//   samplers.py:233-279   get_eps_time / get_xt / get_prediction_targets (+ 347-390 x0 <-> prediction conversions)
//   diffusion.py:144-168, 315-387  get_loss: MSE(pred_target_space, target).mean(C, H, W); F.avg_pool2d pyramid
//   models/unet.py:871-872 x_t / x_t.std((1, 2, 3))   (unbiased), nested level with skip_normalization = false
//   clis/train_parallel.py:194-195  images = (uint8 - 127) / 128, NHWC -> NCHW
#include "common.hpp"

namespace mdm {

// ---------------------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11) + Box-Muller.  Element i of a draw takes lane (i & 3) of
// the block with counter (offset + i / 4, stream): a draw is a pure function of (seed, offset, stream, i), so the
// same numbers can be regenerated on the host (oracle/philox_ref.py) -- "device RNG replayable from a CPU seed".
// ---------------------------------------------------------------------------------------------------------
struct RngState { unsigned long long seed, offset; };

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// four standard normals of block `blk` of stream `stream`
__device__ __forceinline__ void normal4(const RngState& st, unsigned long long blk, uint32_t stream, float (&out)[4]) {
  const unsigned long long ctr = st.offset + blk;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0u};
  philox4x32_10(c, (uint32_t)st.seed, (uint32_t)(st.seed >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)c[2 * h] + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
    const float u2 = (float)c[2 * h + 1] * 2.3283064365386963e-10f;            // [0, 1]
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    out[2 * h] = r * cs; out[2 * h + 1] = r * sn;
  }
}

__global__ void rng_advance_kernel(RngState* st, unsigned long long blocks) { st->offset += blocks; }

template <typename F>
__device__ __forceinline__ void for_each_vec4(size_t total4, F f) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) f(i);
}

enum { PT_EPS = 0, PT_V = 2 };   // PredictionType: DDPM (0) and DDIM (1) both predict eps; V_PREDICTION = 2

// x0 from the model output (samplers.py:347-367)
__device__ __forceinline__ float x0_of(float xt, float pred, float sg, float s1g, int ptype) {
  return ptype == PT_V ? xt * sg - pred * s1g : (xt - pred * s1g) / sg;
}

// ---------------------------------------------------------------------------------------------------------
// N1: one reverse step on a [B, chw] image.  mode 0 = ancestral DDPM posterior mean (ddim_eta is None),
// 1 = DDIM with eta (eta == 0: deterministic, no noise).  clip: 0 none, 1 CLIP, 2 DYNAMIC (thr[b] given), 3 = emit
// the UNCLIPPED x0 * scale into x0_out and stop (first pass of dynamic thresholding: the quantile needs it).
// ---------------------------------------------------------------------------------------------------------
struct StepArgs {
  const float* x_t; const float* pred; const float* pred_uncond; float guidance;
  const float* gamma; const float* gamma_last; const float* noise; const float* noise_gate; const float* thr;
  const RngState* rng; uint32_t rng_stream;
  float* x0_out; float* x_last_out;
  size_t chw4; int B; int ptype, mode; float eta; int need_noise, clip; float scale;
};

__global__ __launch_bounds__(256) void sampler_step_kernel(StepArgs a) {
  const size_t total4 = (size_t)a.B * a.chw4;
  RngState st = {0ull, 0ull};
  const bool gen = a.need_noise && !a.noise && a.rng;
  if (gen) st = *a.rng;
  const float gate = a.noise_gate ? a.noise_gate[0] : 1.f;
  for_each_vec4(total4, [&](size_t i) {
    const int b = (int)(i / a.chw4);
    const float g = a.gamma[b], gl = a.gamma_last[b];
    const float sg = sqrtf(g), s1g = sqrtf(1.f - g), sgl = sqrtf(gl);
    const f32x4 xt = reinterpret_cast<const f32x4*>(a.x_t)[i];
    f32x4 p = reinterpret_cast<const f32x4*>(a.pred)[i];
    if (a.pred_uncond) {
      const f32x4 pu = reinterpret_cast<const f32x4*>(a.pred_uncond)[i];
      p = pu + a.guidance * (p - pu);
    }
    const float alpha = g / gl, beta = 1.f - alpha;
    float beta_t = beta * (1.f - gl) / (1.f - g);
    f32x4 x0, xl;
    float nz[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.need_noise) {
      if (a.noise) { const f32x4 t = reinterpret_cast<const f32x4*>(a.noise)[i]; nz[0] = t[0]; nz[1] = t[1]; nz[2] = t[2]; nz[3] = t[3]; }
      else if (gen) normal4(st, i, a.rng_stream, nz);
    }
    if (a.mode == 1 && a.eta > 0.f) beta_t *= a.eta * a.eta;
    const float thr = a.clip == 2 ? a.thr[b] : 1.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = x0_of(xt[e], p[e], sg, s1g, a.ptype);
      if (a.clip == 1) v = fminf(fmaxf(v * a.scale, -1.f), 1.f) / a.scale;
      else if (a.clip == 2) v = fminf(fmaxf(v * a.scale, -thr), thr) / thr / a.scale;
      else if (a.clip == 3) v = v * a.scale;
      x0[e] = v;
      float x;
      if (a.mode == 0) {
        x = v * beta * sgl / (1.f - g) + xt[e] * sqrtf(alpha) * (1.f - gl) / (1.f - g);
      } else {
        const float eps = (xt[e] - v * sg) / s1g;
        x = v * sgl + eps * sqrtf(a.eta > 0.f ? 1.f - gl - beta_t : 1.f - gl);
      }
      if (a.need_noise && !(a.mode == 1 && a.eta <= 0.f)) x += sqrtf(beta_t) * gate * nz[e];
      xl[e] = x;
    }
    reinterpret_cast<f32x4*>(a.x0_out)[i] = x0;
    if (a.clip != 3) reinterpret_cast<f32x4*>(a.x_last_out)[i] = xl;
  });
}

// ---------------------------------------------------------------------------------------------------------
// N3 (a): x_t = sqrt(g) * images * inv_scale + sqrt(1 - g) * eps; eps given, or drawn here (and stored)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void noise_images_kernel(const float* __restrict__ images, const float* __restrict__ eps_in,
                                                           const float* __restrict__ gamma, float inv_scale,
                                                           float* __restrict__ x_t, float* __restrict__ eps_out,
                                                           const RngState* __restrict__ rng, uint32_t stream, int B,
                                                           size_t chw4) {
  RngState st = {0ull, 0ull};
  if (!eps_in) st = *rng;
  for_each_vec4((size_t)B * chw4, [&](size_t i) {
    const int b = (int)(i / chw4);
    const float g = gamma[b], sg = sqrtf(g), s1g = sqrtf(1.f - g);
    const f32x4 im = reinterpret_cast<const f32x4*>(images)[i];
    float nz[4];
    if (eps_in) { const f32x4 t = reinterpret_cast<const f32x4*>(eps_in)[i]; nz[0] = t[0]; nz[1] = t[1]; nz[2] = t[2]; nz[3] = t[3]; }
    else normal4(st, i, stream, nz);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = sg * (im[e] * inv_scale) + s1g * nz[e];
    reinterpret_cast<f32x4*>(x_t)[i] = o;
    if (eps_out) reinterpret_cast<f32x4*>(eps_out)[i] = f32x4{nz[0], nz[1], nz[2], nz[3]};
  });
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, const RngState* __restrict__ rng, uint32_t stream,
                                                    size_t n4) {
  const RngState st = *rng;
  for_each_vec4(n4, [&](size_t i) {
    float nz[4];
    normal4(st, i, stream, nz);
    reinterpret_cast<f32x4*>(out)[i] = f32x4{nz[0], nz[1], nz[2], nz[3]};
  });
}

// ---------------------------------------------------------------------------------------------------------
// N3 (b): per-sample loss.  p = al * x_t + be * pred (the prediction in the loss-target space), t = ce * eps + ci * img
//   prediction V, target eps:  al = sqrt(1-g), be = sqrt(g)          target eps: ce = 1,       ci = 0
//   prediction eps, target V:  al = -sqrt(1-g)/sqrt(g), be = 1/sqrt(g)   target V:   ce = sqrt(g), ci = -sqrt(1-g)
//   same space:                al = 0, be = 1
// loss[b] = mean_i (p - t)^2 ;  dpred = gloss[b] * (2 / chw) * be * (p - t)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void loss_coefs(float g, int ptype, int ttype, float& al, float& be, float& ce, float& ci) {
  const float sg = sqrtf(g), s1g = sqrtf(1.f - g);
  const bool pv = ptype == PT_V, tv = ttype == PT_V;
  if (pv == tv) { al = 0.f; be = 1.f; }
  else if (pv) { al = s1g; be = sg; }
  else { al = -s1g / sg; be = 1.f / sg; }
  if (tv) { ce = sg; ci = -s1g; } else { ce = 1.f; ci = 0.f; }
}

// grid (slabs, B): part[b][slab] = sum over the slab's elements of (p - t)^2
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* __restrict__ x_t, const float* __restrict__ pred,
                                                           const float* __restrict__ images, const float* __restrict__ eps,
                                                           const float* __restrict__ gamma, float inv_scale,
                                                           float* __restrict__ part, size_t chw4, int ptype, int ttype) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  float al, be, ce, ci;
  loss_coefs(gamma[b], ptype, ttype, al, be, ce, ci);
  const size_t base = (size_t)b * chw4;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < chw4; i += (size_t)gridDim.x * 256) {
    const f32x4 xt = reinterpret_cast<const f32x4*>(x_t)[base + i], pr = reinterpret_cast<const f32x4*>(pred)[base + i];
    const f32x4 im = reinterpret_cast<const f32x4*>(images)[base + i], ep = reinterpret_cast<const f32x4*>(eps)[base + i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = (al * xt[e] + be * pr[e]) - (ce * ep[e] + ci * im[e] * inv_scale);
      s += d * d;
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void loss_final_kernel(const float* __restrict__ part, float* __restrict__ loss, int B, int slabs, float inv_n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int k = 0; k < slabs; ++k) s += part[(size_t)b * slabs + k];
  loss[b] = s * inv_n;
}
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ x_t, const float* __restrict__ pred,
                                                       const float* __restrict__ images, const float* __restrict__ eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ gloss,
                                                       float inv_scale, float* __restrict__ dpred, int B, size_t chw4,
                                                       int ptype, int ttype, float two_over_n) {
  for_each_vec4((size_t)B * chw4, [&](size_t i) {
    const int b = (int)(i / chw4);
    float al, be, ce, ci;
    loss_coefs(gamma[b], ptype, ttype, al, be, ce, ci);
    const float k = gloss[b] * two_over_n * be;
    const f32x4 xt = reinterpret_cast<const f32x4*>(x_t)[i], pr = reinterpret_cast<const f32x4*>(pred)[i];
    const f32x4 im = reinterpret_cast<const f32x4*>(images)[i], ep = reinterpret_cast<const f32x4*>(eps)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = k * ((al * xt[e] + be * pr[e]) - (ce * ep[e] + ci * im[e] * inv_scale));
    reinterpret_cast<f32x4*>(dpred)[i] = o;
  });
}

// ---------------------------------------------------------------------------------------------------------
// F.avg_pool2d(x, r) on NCHW fp32 (diffusion.py:346-348); one thread per output pixel
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, size_t planes, int H,
                                                      int W, int r) {
  const int Ho = H / r, Wo = W / r;
  const size_t total = planes * Ho * Wo;
  const float inv = 1.f / (float)(r * r);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % Wo);
    const size_t t = i / Wo;
    const int oh = (int)(t % Ho);
    const size_t pl = t / Ho;
    const float* src = x + (pl * H + (size_t)oh * r) * W + (size_t)ow * r;
    float s = 0.f;
    for (int dy = 0; dy < r; ++dy)
      for (int dx = 0; dx < r; ++dx) s += src[(size_t)dy * W + dx];
    y[i] = s * inv;
  }
}

// ---------------------------------------------------------------------------------------------------------
// per-sample std normalisation y = x / std(x) (unbiased, over C*H*W) and its gradient.
//   partial: grid (slabs, N): (sum (x - K), sum (x - K)^2) with K = x[n][0]   [bwd: sum dy * x]
//   apply:   every block re-reduces the slab partials of its sample (<= 64 pairs) and scales its range
// stats [N][2] = (mean, 1 / std) are kept for the backward pass.
//   dx = dy / s - (x - mean) * (sum dy x) / ((n - 1) s^3)
// ---------------------------------------------------------------------------------------------------------
constexpr int STD_SLABS = 64;

__global__ __launch_bounds__(256) void std_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ part, size_t n4) {
  __shared__ float sh[2][4];
  const int n = blockIdx.y;
  const f32x4* xn = reinterpret_cast<const f32x4*>(x) + (size_t)n * n4;
  const f32x4* gn = dy ? reinterpret_cast<const f32x4*>(dy) + (size_t)n * n4 : nullptr;
  const float K = x[(size_t)n * n4 * 4];
  float s1 = 0.f, s2 = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = xn[i];
    if (gn) {
      const f32x4 g = gn[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) s1 += g[e] * v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - K; s1 += d; s2 += d * d; }
    }
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)n * gridDim.x + blockIdx.x) * 2;
    o[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    o[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

__global__ __launch_bounds__(256) void std_apply_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                        float* __restrict__ y, float* __restrict__ stats, size_t n4,
                                                        int slabs) {
  const int n = blockIdx.y;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < slabs; ++k) { s1 += part[((size_t)n * slabs + k) * 2]; s2 += part[((size_t)n * slabs + k) * 2 + 1]; }
  const float cnt = (float)(n4 * 4);
  const float K = x[(size_t)n * n4 * 4];
  const float dm = s1 / cnt;                                   // mean - K
  const float var = fmaxf((s2 - cnt * dm * dm) / (cnt - 1.f), 0.f);
  const float inv = rsqrtf(var);
  if (blockIdx.x == 0 && threadIdx.x == 0) { stats[2 * n] = K + dm; stats[2 * n + 1] = inv; }
  const f32x4* xn = reinterpret_cast<const f32x4*>(x) + (size_t)n * n4;
  f32x4* yn = reinterpret_cast<f32x4*>(y) + (size_t)n * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) yn[i] = xn[i] * inv;
}

__global__ __launch_bounds__(256) void std_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ part, const float* __restrict__ stats,
                                                            float* __restrict__ dx, size_t n4, int slabs) {
  const int n = blockIdx.y;
  float sgx = 0.f;
  for (int k = 0; k < slabs; ++k) sgx += part[((size_t)n * slabs + k) * 2];
  const float cnt = (float)(n4 * 4);
  const float mean = stats[2 * n], inv = stats[2 * n + 1];
  const float c = sgx * inv * inv * inv / (cnt - 1.f);
  const f32x4* xn = reinterpret_cast<const f32x4*>(x) + (size_t)n * n4;
  const f32x4* gn = reinterpret_cast<const f32x4*>(dy) + (size_t)n * n4;
  f32x4* on = reinterpret_cast<f32x4*>(dx) + (size_t)n * n4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 v = xn[i], g = gn[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] * inv - (v[e] - mean) * c;
    on[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// N4: uint8 NHWC [B, H, W, 3] -> fp32 NCHW [B, 3, H, W], (u - 127) / 128   (clis/train_parallel.py:194-195)
// one thread per 4 consecutive pixels of one image row segment: 12 bytes in, three 16-byte stores out
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void input_stage_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B,
                                                          size_t hw) {
  const size_t hw4 = hw / 4, total = (size_t)B * hw4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / hw4, q = i - b * hw4;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src + (b * hw + q * 4) * 3);
    const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];
    const uint8_t px[12] = {(uint8_t)w0, (uint8_t)(w0 >> 8), (uint8_t)(w0 >> 16), (uint8_t)(w0 >> 24),
                            (uint8_t)w1, (uint8_t)(w1 >> 8), (uint8_t)(w1 >> 16), (uint8_t)(w1 >> 24),
                            (uint8_t)w2, (uint8_t)(w2 >> 8), (uint8_t)(w2 >> 16), (uint8_t)(w2 >> 24)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      f32x4 o;
#pragma unroll
      for (int p = 0; p < 4; ++p) o[p] = ((float)px[p * 3 + c] - 127.0f) * 0.0078125f;
      reinterpret_cast<f32x4*>(dst + (b * 3 + c) * hw)[q] = o;
    }
  }
}

}  // namespace mdm

using namespace mdm;

static inline int stream_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" int mdm_rng_advance(unsigned long long* state, unsigned long long blocks, void* stream) {
  MDM_CHECK_ARG(state);
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<RngState*>(state), blocks);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_randn(float* out, size_t n, const unsigned long long* rng_state, int rng_stream, void* stream) {
  MDM_CHECK_ARG(out && rng_state && n % 4 == 0);
  hipLaunchKernelGGL(randn_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out,
                     reinterpret_cast<const RngState*>(rng_state), (uint32_t)rng_stream, n / 4);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_sampler_step(const float* x_t, const float* pred, const float* pred_uncond, float guidance,
                                const float* gamma, const float* gamma_last, const float* noise,
                                const float* noise_gate, const float* thr, const unsigned long long* rng_state,
                                int rng_stream, float* x0_out, float* x_last_out, int B, size_t chw, int pred_type,
                                int mode, float ddim_eta, int need_noise, int clip, float image_scale, void* stream) {
  MDM_CHECK_ARG(x_t && pred && gamma && gamma_last && x0_out);
  MDM_CHECK_ARG(B > 0 && chw > 0 && chw % 4 == 0);
  MDM_CHECK_ARG(pred_type >= 0 && pred_type <= 2 && (mode == 0 || mode == 1) && clip >= 0 && clip <= 3);
  MDM_CHECK_ARG(clip == 3 || x_last_out);
  MDM_CHECK_ARG(clip != 2 || thr);
  MDM_CHECK_ARG(image_scale > 0.f);
  StepArgs a;
  a.x_t = x_t; a.pred = pred; a.pred_uncond = pred_uncond; a.guidance = guidance;
  a.gamma = gamma; a.gamma_last = gamma_last; a.noise = noise; a.noise_gate = noise_gate; a.thr = thr;
  a.rng = reinterpret_cast<const RngState*>(rng_state); a.rng_stream = (uint32_t)rng_stream;
  a.x0_out = x0_out; a.x_last_out = x_last_out;
  a.chw4 = chw / 4; a.B = B; a.ptype = pred_type == 2 ? PT_V : PT_EPS; a.mode = mode; a.eta = ddim_eta;
  a.need_noise = need_noise; a.clip = clip; a.scale = image_scale;
  MDM_CHECK_ARG(!need_noise || (mode == 1 && ddim_eta <= 0.f) || noise || rng_state);
  hipLaunchKernelGGL(sampler_step_kernel, dim3(stream_blocks((size_t)B * a.chw4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), a);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_noise_images(const float* images, const float* eps, const float* gamma, float inv_scale, float* x_t,
                                float* eps_out, const unsigned long long* rng_state, int rng_stream, int B, size_t chw,
                                void* stream) {
  MDM_CHECK_ARG(images && gamma && x_t && B > 0 && chw > 0 && chw % 4 == 0);
  MDM_CHECK_ARG(eps || (rng_state && eps_out));
  hipLaunchKernelGGL(noise_images_kernel, dim3(stream_blocks((size_t)B * (chw / 4))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), images, eps, gamma, inv_scale, x_t, eps_out,
                     reinterpret_cast<const RngState*>(rng_state), (uint32_t)rng_stream, B, chw / 4);
  MDM_LAUNCH_STATUS();
}

static inline int loss_slabs(int B, size_t chw4) {
  int s = (512 + B - 1) / B;
  const size_t max_s = (chw4 + 1023) / 1024;
  if ((size_t)s > max_s) s = (int)max_s;
  return s < 1 ? 1 : (s > 256 ? 256 : s);
}

extern "C" int mdm_diffusion_loss_plan(int B, size_t chw, size_t* ws_bytes) {
  MDM_CHECK_ARG(ws_bytes && B > 0 && chw % 4 == 0);
  *ws_bytes = (size_t)B * loss_slabs(B, chw / 4) * sizeof(float);
  return 0;
}

extern "C" int mdm_diffusion_loss_fwd(const float* x_t, const float* pred, const float* images, const float* eps,
                                      const float* gamma, float inv_scale, float* loss, float* ws, int B, size_t chw,
                                      int pred_type, int target_type, void* stream) {
  MDM_CHECK_ARG(x_t && pred && images && eps && gamma && loss && ws && B > 0 && chw > 0 && chw % 4 == 0);
  MDM_CHECK_ARG(pred_type >= 0 && pred_type <= 2 && target_type >= 0 && target_type <= 2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = loss_slabs(B, chw / 4);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(slabs, B), dim3(256), 0, st, x_t, pred, images, eps, gamma, inv_scale, ws,
                     chw / 4, pred_type == 2 ? PT_V : PT_EPS, target_type == 2 ? PT_V : PT_EPS);
  hipLaunchKernelGGL(loss_final_kernel, dim3((B + 63) / 64), dim3(64), 0, st, ws, loss, B, slabs, 1.0f / (float)chw);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_diffusion_loss_bwd(const float* x_t, const float* pred, const float* images, const float* eps,
                                      const float* gamma, const float* gloss, float inv_scale, float* dpred, int B,
                                      size_t chw, int pred_type, int target_type, void* stream) {
  MDM_CHECK_ARG(x_t && pred && images && eps && gamma && gloss && dpred && B > 0 && chw > 0 && chw % 4 == 0);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(stream_blocks((size_t)B * (chw / 4))), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x_t, pred, images, eps, gamma, gloss, inv_scale, dpred, B,
                     chw / 4, pred_type == 2 ? PT_V : PT_EPS, target_type == 2 ? PT_V : PT_EPS, 2.0f / (float)chw);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_avgpool(const float* x, float* y, int N, int C, int H, int W, int r, void* stream) {
  MDM_CHECK_ARG(x && y && N > 0 && C > 0 && r >= 1 && H % r == 0 && W % r == 0);
  const size_t total = (size_t)N * C * (H / r) * (W / r);
  hipLaunchKernelGGL(avgpool_kernel, dim3(stream_blocks(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y,
                     (size_t)N * C, H, W, r);
  MDM_LAUNCH_STATUS();
}

static inline int std_slabs(int N, size_t n4) {
  int s = (512 + N - 1) / N;
  const size_t max_s = (n4 + 1023) / 1024;
  if ((size_t)s > max_s) s = (int)max_s;
  return s < 1 ? 1 : (s > STD_SLABS ? STD_SLABS : s);
}

/* ws: fp32 [N][64][2] */
extern "C" int mdm_sample_std_fwd(const float* x, float* y, float* stats, float* ws, int N, size_t chw, void* stream) {
  MDM_CHECK_ARG(x && y && stats && ws && N > 0 && chw >= 8 && chw % 4 == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = std_slabs(N, chw / 4);
  hipLaunchKernelGGL(std_partial_kernel, dim3(slabs, N), dim3(256), 0, st, x, (const float*)nullptr, ws, chw / 4);
  hipLaunchKernelGGL(std_apply_kernel, dim3(slabs, N), dim3(256), 0, st, x, ws, y, stats, chw / 4, slabs);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_sample_std_bwd(const float* dy, const float* x, const float* stats, float* dx, float* ws, int N,
                                  size_t chw, void* stream) {
  MDM_CHECK_ARG(dy && x && stats && dx && ws && N > 0 && chw >= 8 && chw % 4 == 0);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int slabs = std_slabs(N, chw / 4);
  hipLaunchKernelGGL(std_partial_kernel, dim3(slabs, N), dim3(256), 0, st, x, dy, ws, chw / 4);
  hipLaunchKernelGGL(std_bwd_apply_kernel, dim3(slabs, N), dim3(256), 0, st, dy, x, ws, stats, dx, chw / 4, slabs);
  MDM_LAUNCH_STATUS();
}

extern "C" int mdm_input_stage(const void* u8_nhwc, float* out_nchw, int B, int H, int W, void* stream) {
  MDM_CHECK_ARG(u8_nhwc && out_nchw && B > 0 && H > 0 && W > 0);
  const size_t hw = (size_t)H * W;
  MDM_CHECK_ARG(hw % 4 == 0 && ((size_t)u8_nhwc & 3) == 0);
  hipLaunchKernelGGL(input_stage_kernel, dim3(stream_blocks((size_t)B * hw / 4)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const uint8_t*>(u8_nhwc), out_nchw, B, hw);
  MDM_LAUNCH_STATUS();
}


// ---------------------------------------------------------------------------------------------------------
// Dropout (reference models/unet.py:208,234: nn.Dropout between norm2 + SiLU and conv2 of a ResNet block).
// y[i] = keep(i) ? x[i] / (1 - p) : 0 with keep(i) = (Philox word of element i) >= p * 2^32: a pure function of
// (seed, offset, i), so backward regenerates the same mask from the two integers instead of storing it.  The same
// launch serves both directions (dx = dy masked and scaled the same way).  8 elements per thread = two Philox blocks.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, size_t n8, unsigned thresh,
                                                      float scale, unsigned long long seed, unsigned long long offset) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    if constexpr (sizeof(T) == 2) {
      Chunk<T> c;
      c.load(x + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c.v[e];
    } else {
      Chunk<T> c0, c1;
      c0.load(x + i * 8); c1.load(x + i * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = c0.v[e]; v[4 + e] = c1.v[e]; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned long long ctr = offset + 2 * i + h;
      uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x44524f50u /* "DROP" */, 0u};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * h + e] = c[e] >= thresh ? v[4 * h + e] * scale : 0.f;
    }
    if constexpr (sizeof(T) == 2) {
      Chunk<T> c;
#pragma unroll
      for (int e = 0; e < 8; ++e) c.v[e] = v[e];
      c.store(y + i * 8);
    } else {
      Chunk<T> c0, c1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { c0.v[e] = v[e]; c1.v[e] = v[4 + e]; }
      c0.store(y + i * 8); c1.store(y + i * 8 + 4);
    }
  }
}

extern "C" int mdm_dropout(const void* x, void* y, size_t n, float p, unsigned long long seed, unsigned long long offset,
                           int dtype, void* stream) {
  MDM_CHECK_ARG(x && y && n % 8 == 0 && p >= 0.f && p < 1.f);
  MDM_CHECK_ARG(dtype == DT_F32 || dtype == DT_BF16);
  if (n == 0) return 0;
  const size_t n8 = n / 8;
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  const float scale = 1.f / (1.f - p);
  const unsigned blocks = (unsigned)((n8 + 255) / 256 > 8192 ? 8192 : (n8 + 255) / 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == DT_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, n8, thresh, scale, seed, offset);
  else hipLaunchKernelGGL(dropout_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)x, (bf16*)y, n8, thresh, scale, seed, offset);
  MDM_LAUNCH_STATUS();
}
