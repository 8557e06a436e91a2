// Online-EWC per-step penalty over the flat parameter buffer (mafed/methods/ewc.py:105-127):
//   loss += 0.5 * lambda * sum_i F_i (p_i - p*_i)^2          grad_i += dL * lambda * F_i (p_i - p*_i)
// The reference walks named_parameters() and launches ~6 torch kernels per tensor; here it is two HBM-bound passes over
// three (forward) / four (backward) flat fp32 streams, 12 and 20 bytes per parameter.
#include "common.h"

namespace mafed {

constexpr int EWC_BLOCKS = 2048;

__global__ __launch_bounds__(256) void ewc_partial_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                          const float* __restrict__ f, int64_t n, float* __restrict__ partial) {
  __shared__ float sm[4];
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s0 = 0.f, s1 = 0.f;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto term = [](const float4& a, const float4& b, const float4& w) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
    return (w.x * dx * dx + w.y * dy * dy) + (w.z * dz * dz + w.w * dw * dw);
  };
  for (; i + stride < n4; i += 2 * stride) {
    s0 += term(load4(p + i * 4), load4(q + i * 4), load4(f + i * 4));
    s1 += term(load4(p + (i + stride) * 4), load4(q + (i + stride) * 4), load4(f + (i + stride) * 4));
  }
  for (; i < n4; i += stride) s0 += term(load4(p + i * 4), load4(q + i * 4), load4(f + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const int64_t j = n4 * 4 + threadIdx.x;
    const float d = p[j] - q[j];
    s0 += f[j] * d * d;
  }
  const float s = block_sum<256>(s0 + s1, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void ewc_finish_kernel(const float* __restrict__ partial, int nblk, float half_lambda, float beta,
                                                         float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) s += partial[b];
  s = block_sum<256>(s, sm);
  if (threadIdx.x == 0) out[0] = (beta != 0.f ? beta * out[0] : 0.f) + half_lambda * s;
}

__global__ __launch_bounds__(256) void ewc_bwd_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                      const float* __restrict__ f, int64_t n, float lambda,
                                                      const float* __restrict__ coef_dev, float* __restrict__ grad) {
  const float c = coef_dev[0] * lambda;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = load4(p + i * 4), b = load4(q + i * 4), w = load4(f + i * 4);
    float4 g = load4(grad + i * 4);
    g.x += c * w.x * (a.x - b.x); g.y += c * w.y * (a.y - b.y); g.z += c * w.z * (a.z - b.z); g.w += c * w.w * (a.w - b.w);
    store4(grad + i * 4, g);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const int64_t j = n4 * 4 + threadIdx.x;
    grad[j] += c * f[j] * (p[j] - q[j]);
  }
}

}  // namespace mafed

using namespace mafed;

extern "C" size_t mafed_ewc_workspace_bytes(int64_t n) { (void)n; return (size_t)EWC_BLOCKS * sizeof(float); }

extern "C" int mafed_ewc_penalty_fwd(const float* p, const float* p_old, const float* fisher, int64_t n, float half_lambda, float beta,
                                     float* out, void* workspace, size_t workspace_bytes, void* stream) {
  MAFED_CHECK_ARG(p && p_old && fisher && out && n >= 0, "ewc_penalty_fwd: bad arguments");
  MAFED_CHECK_ARG((((uintptr_t)p | (uintptr_t)p_old | (uintptr_t)fisher) & 15) == 0, "ewc_penalty_fwd: buffers must be 16-byte aligned");
  if (!workspace || workspace_bytes < EWC_BLOCKS * sizeof(float)) {
    set_error("ewc_penalty_fwd: workspace %zu < %zu", workspace_bytes, (size_t)EWC_BLOCKS * sizeof(float));
    return MAFED_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  int64_t nb = cdiv(n / 4 + 1, 256 * 4);
  if (nb > EWC_BLOCKS) nb = EWC_BLOCKS;
  if (nb < 1) nb = 1;
  ewc_partial_kernel<<<dim3((unsigned)nb), dim3(256), 0, st>>>(p, p_old, fisher, n, (float*)workspace);
  MAFED_CHECK_LAUNCH("ewc_penalty_fwd(partial)");
  ewc_finish_kernel<<<dim3(1), dim3(256), 0, st>>>((const float*)workspace, (int)nb, half_lambda, beta, out);
  MAFED_CHECK_LAUNCH("ewc_penalty_fwd(finish)");
  return MAFED_OK;
}

extern "C" int mafed_ewc_penalty_bwd(const float* p, const float* p_old, const float* fisher, int64_t n, float lambda,
                                     const float* coef_dev, float* grad, void* stream) {
  MAFED_CHECK_ARG(p && p_old && fisher && coef_dev && grad && n >= 0, "ewc_penalty_bwd: bad arguments");
  MAFED_CHECK_ARG((((uintptr_t)p | (uintptr_t)p_old | (uintptr_t)fisher | (uintptr_t)grad) & 15) == 0,
                  "ewc_penalty_bwd: buffers must be 16-byte aligned");
  if (n == 0) return MAFED_OK;
  int64_t nb = cdiv(n / 4 + 1, 256);
  if (nb > 4096) nb = 4096;
  ewc_bwd_kernel<<<dim3((unsigned)nb), dim3(256), 0, as_stream(stream)>>>(p, p_old, fisher, n, lambda, coef_dev, grad);
  MAFED_CHECK_LAUNCH("ewc_penalty_bwd");
  return MAFED_OK;
}
