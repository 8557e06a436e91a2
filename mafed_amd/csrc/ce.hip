// Shifted, per-sample-normalised cross-entropy over the text positions (mafed/model/vl_pythia.py:44-96).
// HBM-bound: one 256-thread block per (b,t) row, 4-element vector loads, online log-sum-exp in fp32.
#include "common.h"

namespace mafed {

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels, int B, int Tn,
                                                     int64_t V, float* __restrict__ lse_out, float* __restrict__ row_loss) {
  __shared__ float sm[8];
  const int64_t r = blockIdx.x;
  const int b = (int)(r / Tn), t = (int)(r - (int64_t)b * Tn);
  if (t == Tn - 1) {  // the last position predicts nothing (logits[..., :-1, :], vl_pythia.py:91)
    if (threadIdx.x == 0) { row_loss[r] = 0.f; lse_out[r] = 0.f; }
    return;
  }
  const T* x = logits + r * V;
  float m = -INFINITY, s = 0.f;
  for (int64_t c = (int64_t)threadIdx.x * 4; c < V; c += 256 * 4) {
    const float4 v = load4(x + c);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { s *= expf(m - mx); m = mx; }
    s += (expf(v.x - m) + expf(v.y - m)) + (expf(v.z - m) + expf(v.w - m));
  }
  const float gm = block_max<256>(m, sm);
  s = (m == -INFINITY) ? 0.f : s * expf(m - gm);
  const float gs = block_sum<256>(s, sm);
  if (threadIdx.x == 0) {
    const float lse = gm + logf(gs);
    lse_out[r] = lse;
    const int64_t lab = labels[(int64_t)b * Tn + t + 1];
    row_loss[r] = (lab >= 0 && lab < V) ? (lse - Elem<T>::load(x + lab)) : 0.f;  // ignore_index -100 -> 0
  }
}

// loss = mean_b( sum_t row_loss[b,t] / max(count_b, 1e-13) ); one wave per sample, single block
// `poison` (may be null): a device flag; non-zero turns the loss into NaN (mafed_ce_fwd_guarded)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ labels, int B,
                                                          int Tn, int64_t V, float* __restrict__ loss_out, const int* __restrict__ poison) {
  __shared__ float sm[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = wave; b < B; b += 4) {
    float s = 0.f, c = 0.f;
    for (int t = lane; t < Tn - 1; t += 64) {
      const int64_t lab = labels[(int64_t)b * Tn + t + 1];
      if (lab != -100) { c += 1.f; s += row_loss[(int64_t)b * Tn + t]; }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    acc += s / fmaxf(c, 1e-13f);
  }
  if (lane == 0) sm[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float l = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)B;
    loss_out[0] = (poison && poison[0] != 0) ? __int_as_float(0x7fc00000) : l;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                     const float* __restrict__ lse, int B, int Tn, int64_t V,
                                                     const float* __restrict__ gloss, T* __restrict__ dlogits) {
  const int64_t r = blockIdx.x;
  const int b = (int)(r / Tn), t = (int)(r - (int64_t)b * Tn);
  const T* x = logits + r * V;
  T* dx = dlogits + r * V;
  int64_t lab = -100;
  if (t < Tn - 1) lab = labels[(int64_t)b * Tn + t + 1];
  if (lab == -100) {
    for (int64_t c = (int64_t)threadIdx.x * 4; c < V; c += 256 * 4) store4(dx + c, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  float cnt = 0.f;
  for (int tt = 1; tt < Tn; ++tt) cnt += (labels[(int64_t)b * Tn + tt] != -100) ? 1.f : 0.f;
  const float g = gloss[0] / ((float)B * fmaxf(cnt, 1e-13f));
  const float l = lse[r];
  for (int64_t c = (int64_t)threadIdx.x * 4; c < V; c += 256 * 4) {
    const float4 v = load4(x + c);
    float4 o = make_float4(g * expf(v.x - l), g * expf(v.y - l), g * expf(v.z - l), g * expf(v.w - l));
    if (lab >= c && lab < c + 4) {
      const int k = (int)(lab - c);
      if (k == 0) o.x -= g; else if (k == 1) o.y -= g; else if (k == 2) o.z -= g; else o.w -= g;
    }
    store4(dx + c, o);
  }
}

}  // namespace mafed

using namespace mafed;

static int ce_fwd_impl(const void* logits, mafed_dtype dtype, const int64_t* labels, int B, int T, int64_t V, float* lse, float* row_loss,
                       float* loss_out, const int* poison, void* stream) {
  MAFED_CHECK_ARG(logits && labels && lse && row_loss && loss_out, "ce_fwd: null pointer");
  MAFED_CHECK_ARG(B > 0 && T > 0 && V > 0 && V % 4 == 0, "ce_fwd: bad shape (V must be a multiple of 4)");
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)((int64_t)B * T)), block(256);
  const double ce_bytes = (double)B * T * V * (dtype == MAFED_F32 ? 4.0 : 2.0);  // the logits, read once
  if (dtype == MAFED_F32) launch(K_CE_FWD, ce_bytes, ce_fwd_kernel<float>, grid, block, 0, st, (const float*)logits, labels, B, T, V, lse, row_loss);
  else launch(K_CE_FWD, ce_bytes, ce_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, labels, B, T, V, lse, row_loss);
  MAFED_CHECK_LAUNCH("ce_fwd");
  launch(K_SMALL, 0.0, ce_finalize_kernel, dim3(1), block, 0, st, row_loss, labels, B, T, V, loss_out, poison);
  MAFED_CHECK_LAUNCH("ce_fwd(finalize)");
  return MAFED_OK;
}

extern "C" int mafed_ce_fwd(const void* logits, mafed_dtype dtype, const int64_t* labels, int B, int T, int64_t V, float* lse,
                            float* row_loss, float* loss_out, void* stream) {
  return ce_fwd_impl(logits, dtype, labels, B, T, V, lse, row_loss, loss_out, nullptr, stream);
}

extern "C" int mafed_ce_fwd_guarded(const void* logits, mafed_dtype dtype, const int64_t* labels, int B, int T, int64_t V, float* lse,
                                    float* row_loss, float* loss_out, const int* poison_flag, void* stream) {
  MAFED_CHECK_ARG(poison_flag, "ce_fwd_guarded: null flag");
  return ce_fwd_impl(logits, dtype, labels, B, T, V, lse, row_loss, loss_out, poison_flag, stream);
}

extern "C" int mafed_ce_bwd(const void* logits, mafed_dtype dtype, const int64_t* labels, const float* lse, int B, int T, int64_t V,
                            const float* gloss_dev, void* dlogits, void* stream) {
  MAFED_CHECK_ARG(logits && labels && lse && gloss_dev && dlogits, "ce_bwd: null pointer");
  MAFED_CHECK_ARG(B > 0 && T > 0 && V > 0 && V % 4 == 0, "ce_bwd: bad shape (V must be a multiple of 4)");
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)((int64_t)B * T)), block(256);
  const double ce_bytes = 2.0 * B * T * V * (dtype == MAFED_F32 ? 4.0 : 2.0);  // logits in, gradient out
  if (dtype == MAFED_F32) launch(K_CE_BWD, ce_bytes, ce_bwd_kernel<float>, grid, block, 0, st, (const float*)logits, labels, lse, B, T, V, gloss_dev, (float*)dlogits);
  else launch(K_CE_BWD, ce_bytes, ce_bwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, labels, lse, B, T, V, gloss_dev, (bf16_t*)dlogits);
  MAFED_CHECK_LAUNCH("ce_bwd");
  return MAFED_OK;
}
