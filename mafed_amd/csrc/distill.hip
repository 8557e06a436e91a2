// MAFED modality-aware feature distillation (mafed/methods/distillation.py:124-166, 226-257).
// One pass over the student/teacher hidden states of a layer serves BOTH modality masks: each token row is
// classified from its position (image prefix / valid text / left pad) and its per-row distance is added to that
// class's sum.  HBM-bound: 2 * rows * h * 4 B algorithmic read per layer; one wave per row, 16-byte loads;
// deterministic two-stage reduction (per-block partials -> one finishing block), no float atomics.
#include "common.h"

namespace mafed {

constexpr int DS_MAX_BLOCKS = 1024;

// Row sums over h columns, 1024 columns (4 x 16 B per lane and tensor) per step: the eight loads of a step are unconditional
// (columns past h re-read column 0 and are masked out afterwards) so that they are all in flight together -- a `c < h` loop with one
// load pair per trip ran as h / 256 dependent round trips per row.
__device__ __forceinline__ void row_stats(const float* __restrict__ s, const float* __restrict__ t, int h, int lane, float& dd,
                                          float& ss, float& tt, float& st) {
  dd = ss = tt = st = 0.f;
  for (int c0 = 0; c0 < h; c0 += 1024) {
    float4 a[4], b[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + lane * 4 + 256 * u;
      in[u] = c < h;
      a[u] = load4(s + (in[u] ? c : 0));
      b[u] = load4(t + (in[u] ? c : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m = in[u] ? 1.f : 0.f;
      const float d0 = a[u].x - b[u].x, d1 = a[u].y - b[u].y, d2 = a[u].z - b[u].z, d3 = a[u].w - b[u].w;
      dd += m * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      ss += m * ((a[u].x * a[u].x + a[u].y * a[u].y) + (a[u].z * a[u].z + a[u].w * a[u].w));
      tt += m * ((b[u].x * b[u].x + b[u].y * b[u].y) + (b[u].z * b[u].z + b[u].w * b[u].w));
      st += m * ((a[u].x * b[u].x + a[u].y * b[u].y) + (a[u].z * b[u].z + a[u].w * b[u].w));
    }
  }
}

// torch cosine_embedding_loss(target=1): 1 - st / sqrt((ss + eps) * (tt + eps)), eps = 1e-12 (aten EPSILON)
__device__ __forceinline__ float cos_dist(float ss, float tt, float st) {
  const float EPS = 1e-12f;
  return 1.0f - st / sqrtf((ss + EPS) * (tt + EPS));
}

template <bool COSINE>
__global__ __launch_bounds__(256) void distill_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                          const int64_t* __restrict__ attention_mask, int64_t rows, int S, int P,
                                                          int T, int h, float* __restrict__ partial) {
  __shared__ float sm[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // lang_sum, vision_sum, n_lang, n_vision
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    // mask word and row fetched together (the class decides afterwards whether the row counts): pad rows cost their loads, but a
    // row no longer waits for its mask word first
    const int64_t bi = row / S;
    const int si = (int)(row - bi * S);
    const int64_t mv = attention_mask[bi * T + (si < P ? 0 : si - P)];
    float dd, ss, tt, st;
    row_stats(s + row * h, t + row * h, h, lane, dd, ss, tt, st);
    const int cls = si < P ? 1 : (mv != 0 ? 0 : 2);
    if (cls == 2) continue;
    float d;
    if (COSINE) {
      ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
      d = cos_dist(ss, tt, st);
    } else {
      d = wave_sum(dd) / (float)h;
    }
    acc[cls] += d;
    acc[2 + cls] += 1.f;
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] =
      (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void distill_finish_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out4) {
  __shared__ float sm[4];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += partial[(size_t)b * 4 + k];
    s = block_sum<256>(s, sm);
    if (threadIdx.x == 0) out4[k] = s;
  }
}

template <bool COSINE>
__global__ __launch_bounds__(256) void distill_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                          const int64_t* __restrict__ attention_mask, int64_t rows, int S, int P,
                                                          int T, int h, const float* __restrict__ coef, float* __restrict__ ds,
                                                          int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int cls = modality_class(row, S, P, T, attention_mask);
  const float c = cls == 2 ? 0.f : coef[cls];
  const float* sr = s + row * h;
  const float* tr = t + row * h;
  float* dr = ds + row * h;
  if (c == 0.f) {
    if (!accumulate)
      for (int k = lane * 4; k < h; k += 256) store4(dr + k, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  float a = 0.f, bq = 0.f;  // ds = a * t + bq * s   (cosine)   or   a * (s - t)   (mse)
  if (COSINE) {
    float dd, ss, tt, st;
    row_stats(sr, tr, h, lane, dd, ss, tt, st);
    ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
    const float EPS = 1e-12f;
    const float denom = sqrtf((ss + EPS) * (tt + EPS));
    // d/ds [1 - st/denom] = -( t/denom - st * (tt+eps) * s / denom^3 )
    a = -c / denom;
    bq = c * st * (tt + EPS) / (denom * denom * denom);
  } else {
    a = c * 2.0f / (float)h;
  }
  for (int k = lane * 4; k < h; k += 256) {
    const float4 x = load4(sr + k), y = load4(tr + k);
    float4 o;
    if (COSINE) o = make_float4(a * y.x + bq * x.x, a * y.y + bq * x.y, a * y.z + bq * x.z, a * y.w + bq * x.w);
    else o = make_float4(a * (x.x - y.x), a * (x.y - y.y), a * (x.z - y.z), a * (x.w - y.w));
    if (accumulate) {
      const float4 p = load4(dr + k);
      o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
    }
    store4(dr + k, o);
  }
}

// CLS variant: token 0 of every sample, cosine, mean over the batch.  B is small: one block, one wave per sample.
__global__ __launch_bounds__(256) void distill_cls_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t, int B, int S,
                                                              int h, float* __restrict__ out1) {
  __shared__ float sm[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = wave; b < B; b += 4) {
    float dd, ss, tt, st;
    row_stats(s + (int64_t)b * S * h, t + (int64_t)b * S * h, h, lane, dd, ss, tt, st);
    ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
    acc += cos_dist(ss, tt, st);
  }
  if (lane == 0) sm[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out1[0] = ((sm[0] + sm[1]) + (sm[2] + sm[3])) / (float)B;
}

__global__ __launch_bounds__(256) void distill_cls_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t, int B, int S,
                                                              int h, const float* __restrict__ coef, float* __restrict__ ds,
                                                              int accumulate) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // over all B*S rows so that non-CLS rows get zeros when !accumulate
  if (row >= (int64_t)B * S) return;
  float* dr = ds + row * h;
  if (row % S != 0) {
    if (!accumulate)
      for (int k = lane * 4; k < h; k += 256) store4(dr + k, make_float4(0.f, 0.f, 0.f, 0.f));
    return;
  }
  const float* sr = s + row * h;
  const float* tr = t + row * h;
  float dd, ss, tt, st;
  row_stats(sr, tr, h, lane, dd, ss, tt, st);
  ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
  const float EPS = 1e-12f, c = coef[0];
  const float denom = sqrtf((ss + EPS) * (tt + EPS));
  const float a = -c / denom, bq = c * st * (tt + EPS) / (denom * denom * denom);
  for (int k = lane * 4; k < h; k += 256) {
    const float4 x = load4(sr + k), y = load4(tr + k);
    float4 o = make_float4(a * y.x + bq * x.x, a * y.y + bq * x.y, a * y.z + bq * x.z, a * y.w + bq * x.w);
    if (accumulate) {
      const float4 p = load4(dr + k);
      o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
    }
    store4(dr + k, o);
  }
}

// Scalar tail of distill() (mafed/methods/distillation.py:105-122, 147-162) for all distilled layers in ONE launch:
//   lang[l] = sums[l][0] / sums[l][2], vis[l] = sums[l][1] / sums[l][3]                      (masked means, :248)
//   lw[l]   = n_lang / (n_lang + n_vis) | const | lang_coeff[l] ;  vw[l] = 1 - lw[l]         (distillation_loss_weights.py:71-79)
//   per_layer[l] = lw * lang + vw * vis ;  loss = sum_l coeff[l] * per_layer[l]              (coeff = layer coefficient x distillation_coeff)
// and the per-layer gradient coefficients the model's backward injects: inj[l] = {coeff lw / n_lang, coeff vw / n_vis, 0, 0}
// (= d loss / d {sum_lang, sum_vis}).  One wave; layers strided over lanes; fixed-order reduction.
__global__ __launch_bounds__(64) void distill_combine_kernel(const float* __restrict__ sums, int nl, const float* __restrict__ coeff, int mode,
                                                             float lang_const, const float* __restrict__ lang_vec, float* __restrict__ loss,
                                                             float* __restrict__ per_layer, float* __restrict__ modality,
                                                             float* __restrict__ inj) {
  const int lane = threadIdx.x;
  const float n0 = sums[2], n1 = sums[3];  // counts of the first distilled layer (every layer sees the same masks)
  float acc = 0.f;
  for (int l = lane; l < nl; l += 64) {
    const float4 s = load4(sums + 4 * l);
    const float lang = s.x / s.z, vis = s.y / s.w;
    float lw;
    if (mode == 0) lw = n0 / (n0 + n1);
    else if (mode == 1) lw = lang_const;
    else lw = lang_vec[l];
    const float vw = 1.0f - lw;
    const float pl = lw * lang + vw * vis;
    const float c = coeff[l];
    per_layer[l] = pl;
    modality[2 * l] = lang;
    modality[2 * l + 1] = vis;
    store4(inj + 4 * l, make_float4(c * lw / s.z, c * vw / s.w, 0.f, 0.f));
    acc += c * pl;
  }
  acc = wave_sum(acc);
  if (lane == 0) loss[0] = acc;
}

static int ds_blocks(int64_t rows) {
  int64_t nb = cdiv(rows, 4);
  if (nb > DS_MAX_BLOCKS) nb = DS_MAX_BLOCKS;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace mafed

using namespace mafed;

extern "C" size_t mafed_distill_workspace_bytes(int64_t rows) { return (size_t)ds_blocks(rows) * 4 * sizeof(float); }

extern "C" int mafed_distill_fwd(const float* s, const float* t, const int64_t* attention_mask, int B, int S, int P, int h, int cosine,
                                 float* out4, void* workspace, size_t workspace_bytes, void* stream) {
  MAFED_CHECK_ARG(s && t && attention_mask && out4, "distill_fwd: null pointer");
  MAFED_CHECK_ARG(B > 0 && S > 0 && P >= 0 && P <= S && h > 0 && h % 4 == 0, "distill_fwd: bad shape (h must be a multiple of 4)");
  const int64_t rows = (int64_t)B * S;
  const int nblk = ds_blocks(rows);
  if (!workspace || workspace_bytes < (size_t)nblk * 4 * sizeof(float)) {
    set_error("distill_fwd: workspace %zu < %zu", workspace_bytes, (size_t)nblk * 4 * sizeof(float));
    return MAFED_EWORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const double dbytes = 2.0 * rows * h * 4.0;  // student + teacher hidden state of one layer
  if (cosine) launch(K_DISTILL_FWD, dbytes, distill_fwd_kernel<true>, dim3(nblk), dim3(256), 0, st, s, t, attention_mask, rows, S, P, S - P, h, (float*)workspace);
  else launch(K_DISTILL_FWD, dbytes, distill_fwd_kernel<false>, dim3(nblk), dim3(256), 0, st, s, t, attention_mask, rows, S, P, S - P, h, (float*)workspace);
  MAFED_CHECK_LAUNCH("distill_fwd");
  launch(K_SMALL, 0.0, distill_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nblk, out4);
  MAFED_CHECK_LAUNCH("distill_fwd(finish)");
  return MAFED_OK;
}

extern "C" int mafed_distill_combine(const float* sums, int n_layers, const float* layer_coeff_dev, int modality_mode, float lang_weight,
                                     const float* lang_weight_vec_dev, float* loss_out, float* per_layer_out, float* modality_out,
                                     float* inject_out, void* stream) {
  MAFED_CHECK_ARG(sums && layer_coeff_dev && loss_out && per_layer_out && modality_out && inject_out, "distill_combine: null pointer");
  MAFED_CHECK_ARG(n_layers > 0 && modality_mode >= 0 && modality_mode <= 2, "distill_combine: n_layers=%d mode=%d", n_layers, modality_mode);
  MAFED_CHECK_ARG(modality_mode != 2 || lang_weight_vec_dev, "distill_combine: adaptive weights need the per-layer vector");
  MAFED_CHECK_ARG((((uintptr_t)sums | (uintptr_t)inject_out) & 15) == 0, "distill_combine: sums / inject must be 16-byte aligned");
  launch(K_SMALL, 0.0, distill_combine_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, n_layers, layer_coeff_dev, modality_mode, lang_weight,
         lang_weight_vec_dev, loss_out, per_layer_out, modality_out, inject_out);
  MAFED_CHECK_LAUNCH("distill_combine");
  return MAFED_OK;
}

extern "C" int mafed_distill_bwd(const float* s, const float* t, const int64_t* attention_mask, int B, int S, int P, int h, int cosine,
                                 const float* coef_dev, float* ds, int accumulate, void* stream) {
  MAFED_CHECK_ARG(s && t && attention_mask && coef_dev && ds, "distill_bwd: null pointer");
  MAFED_CHECK_ARG(B > 0 && S > 0 && P >= 0 && P <= S && h > 0 && h % 4 == 0, "distill_bwd: bad shape (h must be a multiple of 4)");
  const int64_t rows = (int64_t)B * S;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(rows, 4)), block(256);
  const double dbytes = (accumulate ? 4.0 : 3.0) * rows * h * 4.0;
  if (cosine) launch(K_DISTILL_BWD, dbytes, distill_bwd_kernel<true>, grid, block, 0, st, s, t, attention_mask, rows, S, P, S - P, h, coef_dev, ds, accumulate);
  else launch(K_DISTILL_BWD, dbytes, distill_bwd_kernel<false>, grid, block, 0, st, s, t, attention_mask, rows, S, P, S - P, h, coef_dev, ds, accumulate);
  MAFED_CHECK_LAUNCH("distill_bwd");
  return MAFED_OK;
}

extern "C" int mafed_distill_cls_fwd(const float* s, const float* t, int B, int S, int h, float* out1, void* stream) {
  MAFED_CHECK_ARG(s && t && out1 && B > 0 && S > 0 && h > 0 && h % 4 == 0, "distill_cls_fwd: bad arguments");
  distill_cls_fwd_kernel<<<dim3(1), dim3(256), 0, as_stream(stream)>>>(s, t, B, S, h, out1);
  MAFED_CHECK_LAUNCH("distill_cls_fwd");
  return MAFED_OK;
}

extern "C" int mafed_distill_cls_bwd(const float* s, const float* t, int B, int S, int h, const float* coef_dev, float* ds,
                                     int accumulate, void* stream) {
  MAFED_CHECK_ARG(s && t && coef_dev && ds && B > 0 && S > 0 && h > 0 && h % 4 == 0, "distill_cls_bwd: bad arguments");
  const int64_t rows = (int64_t)B * S;
  distill_cls_bwd_kernel<<<dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, as_stream(stream)>>>(s, t, B, S, h, coef_dev, ds, accumulate);
  MAFED_CHECK_LAUNCH("distill_cls_bwd");
  return MAFED_OK;
}
