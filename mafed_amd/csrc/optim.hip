// Flat-buffer optimiser kernels: global gradient L2 norm + clip scale (Lightning gradient_clip_val,
// mafed/train.py:288) and HF-style AdamW (mafed/optim/adamw.py:86-111).  HBM-bound streaming, 16-byte accesses.
#include "common.h"

namespace mafed {

constexpr int GN_BLOCKS = 1024;

__global__ __launch_bounds__(256) void gradnorm_partial_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float sm[4];
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s0 = 0.f, s1 = 0.f;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 a = load4(g + i * 4), b = load4(g + (i + stride) * 4);
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    s1 += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
  }
  for (; i < n4; i += stride) {
    const float4 a = load4(g + i * 4);
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const float a = g[n4 * 4 + threadIdx.x];
    s0 += a * a;
  }
  const float s = block_sum<256>(s0 + s1, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// sumsq16[blockIdx & 15] += sum of squares of this block's share of x (the unfused form of the weight-gradient epilogue's squares)
__global__ __launch_bounds__(256) void sumsq_accumulate_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ sumsq16) {
  __shared__ float sm[4];
  const int64_t n4 = n / 4, stride = (int64_t)gridDim.x * blockDim.x;
  float s0 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = load4(x + i * 4);
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const float a = x[n4 * 4 + threadIdx.x];
    s0 += a * a;
  }
  const float s = block_sum<256>(s0, sm);
  if (threadIdx.x == 0) atomicAdd(sumsq16 + (blockIdx.x & 15), s);
}

__global__ __launch_bounds__(256) void gradnorm_finish_kernel(const float* __restrict__ partial, int nblk, float max_norm,
                                                              float* __restrict__ out2) {
  __shared__ float sm[4];
  // eight loads in flight per thread (the piecewise norm leaves ~27k partials at 410M: one load per trip was a 50 us chain on the
  // optimiser step's critical path); fixed association, so the result does not depend on the launch
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = threadIdx.x;
  for (; b + 7 * 256 < nblk; b += 8 * 256) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[b + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += v[u];
  }
  for (; b < nblk; b += 256) acc[0] += partial[b];
  float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  s = block_sum<256>(s, sm);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out2[0] = norm;
    // torch clip_grad_norm_: clamp(max_norm / (norm + 1e-6), max=1); a non-finite norm marks the step as skipped (scale -1: the AdamW
    // kernel then leaves parameters and optimiser state alone -- fminf(1, max_norm / NaN) would be 1 and NaN gradients would be applied)
    out2[1] = isfinite(norm) ? fminf(1.0f, max_norm / (norm + 1e-6f)) : -1.0f;
  }
}

template <bool SHADOW, bool ZERO_G = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, const float* __restrict__ lr_dev, float beta1,
                                                    float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    const float* __restrict__ clip_dev, float grad_mul, bf16_t* __restrict__ p_bf16,
                                                    int64_t zero_n) {
  // ZERO_G: g[0 .. zero_n) is zeroed in the same pass (zero_n = n: the whole chunk; the layer chunks pass their LayerNorm part only)
  const int64_t zero4 = zero_n / 4;
  const float lr = lr_dev[0];
  if (bc1 <= 0.f) {  // hyper-parameters of this step live in device memory (hipGraph replay): {lr, 1-b1^t, sqrt(1-b2^t)}
    bc1 = lr_dev[1];
    bc2_sqrt = lr_dev[2];
  }
  const float clip = clip_dev ? clip_dev[1] : 1.0f;
  const bool skip = clip < 0.f;   // non-finite gradient norm (gradnorm_finish): no update, optimiser state untouched; g is still zeroed
  const float gs = grad_mul * clip;
  const float step_size = lr * bc2_sqrt / bc1;
  const float decay = lr * wd;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (skip) {
    if (ZERO_G) {
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4 && i < zero4; i += stride) store4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));
      if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4) && n4 * 4 + threadIdx.x < zero_n) g[n4 * 4 + threadIdx.x] = 0.f;
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = load4(p + i * 4), gg = load4(g + i * 4), mm = load4(m + i * 4), vv = load4(v + i * 4);
    float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * gs;
      ma[k] = ma[k] * beta1 + (1.0f - beta1) * gk;
      va[k] = va[k] * beta2 + (1.0f - beta2) * gk * gk;
      const float denom = sqrtf(va[k]) + eps;
      float x = pa[k] - step_size * (ma[k] / denom);
      if (wd > 0.f) x = x - decay * x;
      pa[k] = x;
    }
    store4(p + i * 4, pp); store4(m + i * 4, mm); store4(v + i * 4, vv);
    if (SHADOW) store4(p_bf16 + i * 4, pp);
    if (ZERO_G && i < zero4) store4(g + i * 4, make_float4(0.f, 0.f, 0.f, 0.f));  // optimizer.zero_grad() of the next window, same pass
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gk = g[i] * gs;
    const float mk = m[i] * beta1 + (1.0f - beta1) * gk;
    const float vk = v[i] * beta2 + (1.0f - beta2) * gk * gk;
    float x = p[i] - step_size * (mk / (sqrtf(vk) + eps));
    if (wd > 0.f) x = x - decay * x;
    p[i] = x; m[i] = mk; v[i] = vk;
    if (SHADOW) p_bf16[i] = f32_to_bf16(x);
    if (ZERO_G && i < zero_n) g[i] = 0.f;
  }
}

// One thread: advance the optimiser step counter and publish this step's scalars {lr, 1-b1^t, sqrt(1-b2^t)} in double
// precision, so that a whole training step (schedule included) replays from a hipGraph without host involvement.
__device__ __forceinline__ void optim_advance_body(long long* __restrict__ state, double base_lr, long long warmup, long long total, double b1,
                                                   double b2, float* __restrict__ hyper) {
  const long long t = state[0] + 1;
  state[0] = t;
  const long long s = t - 1;  // LambdaLR epoch in force during optimiser step t (scheduler steps once per optimiser step)
  double lam = 1.0;
  if (total > 0) {
    if (s < warmup) lam = (double)s / (double)(warmup > 1 ? warmup : 1);
    else {
      const long long den = total - warmup > 1 ? total - warmup : 1;
      lam = (double)(total - s) / (double)den;
      if (lam < 0.0) lam = 0.0;
    }
  }
  hyper[0] = (float)(base_lr * lam);
  hyper[1] = (float)(1.0 - pow(b1, (double)t));
  hyper[2] = (float)sqrt(1.0 - pow(b2, (double)t));
}

__global__ void optim_advance_kernel(long long* __restrict__ state, double base_lr, long long warmup, long long total, double b1,
                                     double b2, float* __restrict__ hyper, const float* __restrict__ clip_dev) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (clip_dev && clip_dev[1] < 0.f) return;   // skipped step (non-finite gradient norm): the counter and the schedule stand still
  optim_advance_body(state, base_lr, warmup, total, b1, b2, hyper);
}

// gradnorm_finish_kernel + optim_advance_kernel as ONE launch (the two sat back to back, with a 4-byte device copy of the norm for the
// step's log between them, on the optimiser step's critical path): also leaves the norm in `norm_log` (a slot the caller owns).
__global__ __launch_bounds__(256) void gradnorm_finish_advance_kernel(const float* __restrict__ partial, int nblk, float max_norm,
                                                                      float* __restrict__ out2, float* __restrict__ norm_log,
                                                                      long long* __restrict__ state, double base_lr, long long warmup,
                                                                      long long total, double b1, double b2, float* __restrict__ hyper) {
  __shared__ float sm[4];
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int b = threadIdx.x;
  for (; b + 7 * 256 < nblk; b += 8 * 256) {   // (same association as gradnorm_finish_kernel: bit-identical norm)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[b + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += v[u];
  }
  for (; b < nblk; b += 256) acc[0] += partial[b];
  float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  s = block_sum<256>(s, sm);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out2[0] = norm;
    const bool ok = isfinite(norm);
    out2[1] = ok ? fminf(1.0f, max_norm / (norm + 1e-6f)) : -1.0f;
    if (norm_log) norm_log[0] = norm;
    if (ok) optim_advance_body(state, base_lr, warmup, total, b1, b2, hyper);
  }
}

}  // namespace mafed

using namespace mafed;

extern "C" int mafed_gradnorm_blocks(int64_t n);

extern "C" int mafed_optim_advance(int64_t* state_dev, double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1,
                                   double beta2, float* hyper3_dev, void* stream) {
  MAFED_CHECK_ARG(state_dev && hyper3_dev, "optim_advance: null pointer");
  optim_advance_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>((long long*)state_dev, base_lr, (long long)warmup_steps,
                                                                    (long long)total_steps, beta1, beta2, hyper3_dev, nullptr);
  MAFED_CHECK_LAUNCH("optim_advance");
  return MAFED_OK;
}

extern "C" int mafed_optim_advance_guarded(int64_t* state_dev, double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1,
                                           double beta2, float* hyper3_dev, const float* clip_dev, void* stream) {
  MAFED_CHECK_ARG(state_dev && hyper3_dev, "optim_advance: null pointer");
  optim_advance_kernel<<<dim3(1), dim3(64), 0, as_stream(stream)>>>((long long*)state_dev, base_lr, (long long)warmup_steps,
                                                                    (long long)total_steps, beta1, beta2, hyper3_dev, clip_dev);
  MAFED_CHECK_LAUNCH("optim_advance");
  return MAFED_OK;
}

extern "C" int mafed_sumsq_accumulate(const float* x, int64_t n, float* sumsq16, void* stream) {
  MAFED_CHECK_ARG(x && sumsq16 && n >= 0 && ((uintptr_t)x & 15) == 0, "sumsq_accumulate: bad arguments");
  if (n == 0) return MAFED_OK;
  launch(K_GRADNORM, (double)n * 4.0, sumsq_accumulate_kernel, dim3((unsigned)mafed_gradnorm_blocks(n)), dim3(256), 0, as_stream(stream), x, n, sumsq16);
  MAFED_CHECK_LAUNCH("sumsq_accumulate");
  return MAFED_OK;
}

extern "C" size_t mafed_gradnorm_workspace_bytes(int64_t n) { (void)n; return (size_t)GN_BLOCKS * sizeof(float); }

extern "C" int mafed_gradnorm_clip(const float* g, int64_t n, float max_norm, float* out2, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  MAFED_CHECK_ARG(g && out2 && n >= 0, "gradnorm_clip: bad arguments");
  MAFED_CHECK_ARG(((uintptr_t)g & 15) == 0, "gradnorm_clip: g must be 16-byte aligned");
  if (!workspace || workspace_bytes < GN_BLOCKS * sizeof(float)) {
    set_error("gradnorm_clip: workspace %zu < %zu", workspace_bytes, (size_t)GN_BLOCKS * sizeof(float));
    return MAFED_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  int64_t nb = cdiv(n / 4 + 1, 256 * 4);
  if (nb > GN_BLOCKS) nb = GN_BLOCKS;
  if (nb < 1) nb = 1;
  launch(K_GRADNORM, (double)n * 4.0, gradnorm_partial_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, n, (float*)workspace);
  MAFED_CHECK_LAUNCH("gradnorm(partial)");
  launch(K_SMALL, 0.0, gradnorm_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)nb, max_norm, out2);
  MAFED_CHECK_LAUNCH("gradnorm(finish)");
  return MAFED_OK;
}

// Piecewise form of the same norm: sum-of-squares partials of one range of the gradient buffer (as soon as that range is final,
// on whatever stream finished it), then one finish over all partials.  With these the optimiser step keeps only the last range and
// the finish on its critical path instead of a 1.6 GB pass (0.28 ms at 410M).  Deterministic: every range has a fixed block count
// and the finish adds the partials in index order.
extern "C" int mafed_gradnorm_blocks(int64_t n) {
  int64_t nb = cdiv(n / 4 + 1, 256 * 4);
  if (nb > GN_BLOCKS) nb = GN_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int mafed_gradnorm_partial(const float* g, int64_t n, float* partial_out, void* stream) {
  MAFED_CHECK_ARG(g && partial_out && n >= 0, "gradnorm_partial: bad arguments");
  MAFED_CHECK_ARG(((uintptr_t)g & 15) == 0, "gradnorm_partial: g must be 16-byte aligned");
  launch(K_GRADNORM, (double)n * 4.0, gradnorm_partial_kernel, dim3((unsigned)mafed_gradnorm_blocks(n)), dim3(256), 0, as_stream(stream), g, n, partial_out);
  MAFED_CHECK_LAUNCH("gradnorm_partial");
  return MAFED_OK;
}

extern "C" int mafed_gradnorm_finish(const float* partial, int n_partials, float max_norm, float* out2, void* stream) {
  MAFED_CHECK_ARG(partial && out2 && n_partials >= 1, "gradnorm_finish: bad arguments");
  launch(K_SMALL, 0.0, gradnorm_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, n_partials, max_norm, out2);
  MAFED_CHECK_LAUNCH("gradnorm_finish");
  return MAFED_OK;
}

extern "C" int mafed_gradnorm_finish_advance(const float* partial, int n_partials, float max_norm, float* out2, float* norm_log,
                                             int64_t* state_dev, double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1,
                                             double beta2, float* hyper3_dev, void* stream) {
  MAFED_CHECK_ARG(partial && out2 && n_partials >= 1 && state_dev && hyper3_dev, "gradnorm_finish_advance: bad arguments");
  launch(K_SMALL, 0.0, gradnorm_finish_advance_kernel, dim3(1), dim3(256), 0, as_stream(stream), partial, n_partials, max_norm, out2, norm_log,
         (long long*)state_dev, base_lr, (long long)warmup_steps, (long long)total_steps, beta1, beta2, hyper3_dev);
  MAFED_CHECK_LAUNCH("gradnorm_finish_advance");
  return MAFED_OK;
}

static int adamw_impl(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2, float eps,
                      float weight_decay, int step, const float* clip_dev, float grad_mul, void* p_bf16, int64_t zero_n, void* stream) {
  MAFED_CHECK_ARG(zero_n >= 0 && zero_n <= n && (zero_n == n || zero_n % 4 == 0), "adamw_step: zero_n must be 0 .. n and a multiple of 4 (or n)");
  const bool zero_g = zero_n > 0;
  MAFED_CHECK_ARG(p && g && m && v && lr_dev && n >= 0 && step >= 0, "adamw_step: bad arguments");
  MAFED_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw_step: buffers must be 16-byte aligned");
  MAFED_CHECK_ARG(!p_bf16 || ((uintptr_t)p_bf16 & 7) == 0, "adamw_step: p_bf16 must be 8-byte aligned");
  if (n == 0) return MAFED_OK;
  // bias corrections in double on the host, exactly as math.sqrt(1 - b2**t) / (1 - b1**t) (adamw.py:94-97)
  // step == 0: the kernel reads {lr, 1-b1^t, sqrt(1-b2^t)} from lr_dev[0..2] (graph-replayable form)
  const double bc1 = step > 0 ? 1.0 - pow((double)beta1, (double)step) : 0.0;
  const double bc2 = step > 0 ? 1.0 - pow((double)beta2, (double)step) : 0.0;
  hipStream_t st = as_stream(stream);
  int64_t nb = cdiv(n / 4 + 1, 256);
  if (nb > 4096) nb = 4096;
  // algorithmic bytes: p, m, v read + written, g read (+ the bf16 shadow weight written, + the gradient zeroed) per parameter
  const double bytes = (double)n * (28.0 + (p_bf16 ? 2.0 : 0.0)) + (double)zero_n * 4.0;
#define MAFED_ADAMW(SH, ZG)                                                                                                          \
  launch(K_ADAMW, bytes, adamw_kernel<SH, ZG>, dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, n, lr_dev, beta1, beta2, eps, weight_decay, \
         (float)bc1, (float)sqrt(bc2), clip_dev, grad_mul, (bf16_t*)p_bf16, zero_n)
  if (p_bf16 && zero_g) MAFED_ADAMW(true, true);
  else if (p_bf16) MAFED_ADAMW(true, false);
  else if (zero_g) MAFED_ADAMW(false, true);
  else MAFED_ADAMW(false, false);
#undef MAFED_ADAMW
  MAFED_CHECK_LAUNCH("adamw_step");
  return MAFED_OK;
}

extern "C" int mafed_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                                float eps, float weight_decay, int step, const float* clip_dev, float grad_mul, void* p_bf16,
                                void* stream) {
  return adamw_impl(p, const_cast<float*>(g), m, v, n, lr_dev, beta1, beta2, eps, weight_decay, step, clip_dev, grad_mul, p_bf16, 0, stream);
}

extern "C" int mafed_adamw_step_zero_grad(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                                          float eps, float weight_decay, int step, const float* clip_dev, float grad_mul, void* p_bf16,
                                          void* stream) {
  return adamw_impl(p, g, m, v, n, lr_dev, beta1, beta2, eps, weight_decay, step, clip_dev, grad_mul, p_bf16, n, stream);
}

extern "C" int mafed_adamw_step_partial_zero(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                                             float eps, float weight_decay, int step, const float* clip_dev, float grad_mul, void* p_bf16,
                                             int64_t zero_n, void* stream) {
  return adamw_impl(p, g, m, v, n, lr_dev, beta1, beta2, eps, weight_decay, step, clip_dev, grad_mul, p_bf16, zero_n, stream);
}
