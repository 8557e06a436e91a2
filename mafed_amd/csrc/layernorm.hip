// Dual-affine LayerNorm forward/backward for the fp32 residual stream (tf:243-244,261,272,313).
// HBM-bound: one wave per token row, 16-byte loads, x read once for both affine outputs.
#include "common.h"

namespace mafed {

constexpr int LN_ROWS_PER_BLOCK = 4;  // 4 waves of 64

template <int NV, typename OutT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t rows, int h, float eps,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            OutT* __restrict__ y1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, OutT* __restrict__ y2,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK + wave;
  if (row >= rows) return;
  const float* xr = x + row * h;
  // Every load of the row is issued up front and unconditionally (columns past h re-read column 0 and are discarded by selects):
  // with the loads inside `c < h` regions the compiler drained vmcnt(0) at the end of every region -- four dependent HBM round trips
  // for x, then eight dependent L2 round trips for the affine parameters, per wave.
  float4 v[NV], g1[NV], o1[NV], g2[NV], o2[NV];
  bool in[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    in[i] = c < h;
    v[i] = load4(xr + (in[i] ? c : 0));
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = in[i] ? (lane + 64 * i) * 4 : 0;
    g1[i] = load4(w1 + c);
    o1[i] = load4(b1 + c);
    if (y2) {  // uniform
      g2[i] = load4(w2 + c);
      o2[i] = load4(b2 + c);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (!in[i]) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)h;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
    q += in[i] ? (a * a + b * b) + (cc * cc + d * d) : 0.f;
  }
  const float var = wave_sum(q) / (float)h;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (in[i]) {
      float4 n = make_float4((v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd);
      store4(y1 + row * h + c, make_float4(n.x * g1[i].x + o1[i].x, n.y * g1[i].y + o1[i].y, n.z * g1[i].z + o1[i].z, n.w * g1[i].w + o1[i].w));
      if (y2) store4(y2 + row * h + c, make_float4(n.x * g2[i].x + o2[i].x, n.y * g2[i].y + o2[i].y, n.z * g2[i].z + o2[i].z, n.w * g2[i].w + o2[i].w));
    }
  }
}

// Backward.  Each block walks rows with stride gridDim.x*4; per-lane register partials of the four parameter
// gradients are combined across the block's 4 waves through LDS and written to workspace [gridDim.x][4][h];
// ln_param_reduce_kernel then sums the slabs deterministically and accumulates into dw/db.
// DXSUM: additionally accumulate the column sums of dx (= the bias gradients of the two Linear layers that produced the
// residual branches of the layer below: both are sum_rows(dY) with dY = this dx), saving two column-sum passes per layer.
template <int NV, typename DyT, bool DUAL, bool DXSUM>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const DyT* __restrict__ dy1, const DyT* __restrict__ dy2,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, int64_t rows, int h,
                                                            const float* __restrict__ dres, float* __restrict__ dx,
                                                            DyT* __restrict__ dx_lp, const float* __restrict__ teacher,
                                                            const int64_t* __restrict__ attention_mask, int S, int P, int T,
                                                            const float* __restrict__ inj_scale, float inj_mul,
                                                            float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [3 waves][h] staging for the cross-wave sum
  constexpr int NP = (DUAL ? 4 : 2) + (DXSUM ? 1 : 0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 aw1[NV], ab1[NV], aw2[DUAL ? NV : 1], ab2[DUAL ? NV : 1], adx[DXSUM ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    aw1[i] = ab1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DUAL) aw2[i] = ab2[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DXSUM) adx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 g1[NV], g2[DUAL ? NV : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    g1[i] = (c < h) ? load4(w1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (DUAL) g2[i] = (c < h) ? load4(w2 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float ls = 0.f, vs = 0.f;
  // inj_mul > 0: MSE injection inj_mul * scale * (x - teacher); inj_mul < 0: cosine-distance injection |inj_mul| * scale * d/dx [1 - cos(x, teacher)]
  const bool inj_cos = inj_mul < 0.f;
  if (teacher) { const float m = fabsf(inj_mul); ls = inj_scale[0] * m; vs = inj_scale[1] * m; }
  for (int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_BLOCK + wave; row < rows; row += (int64_t)gridDim.x * LN_ROWS_PER_BLOCK) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NV], g[NV];
    // the residual-path gradient and the teacher row are only consumed after the two row reductions: fetch them now, with
    // the first phase's loads, so that their latency hides behind the reductions instead of following them
    float4 rres[NV], rte[NV];
    float inj = 0.f;
    if (teacher) {
      const int cls = modality_class(row, S, P, T, attention_mask);
      inj = cls == 0 ? ls : (cls == 1 ? vs : 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      rres[i] = (dres && c < h) ? load4(dres + row * h + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      rte[i] = (teacher && c < h) ? load4(teacher + row * h + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < h) {
        const float4 xv = load4(x + row * h + c);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        const float4 d1 = load4(dy1 + row * h + c);
        aw1[i].x += d1.x * xh[i].x; aw1[i].y += d1.y * xh[i].y; aw1[i].z += d1.z * xh[i].z; aw1[i].w += d1.w * xh[i].w;
        ab1[i].x += d1.x; ab1[i].y += d1.y; ab1[i].z += d1.z; ab1[i].w += d1.w;
        g[i] = make_float4(d1.x * g1[i].x, d1.y * g1[i].y, d1.z * g1[i].z, d1.w * g1[i].w);
        if (DUAL) {
          const float4 d2 = load4(dy2 + row * h + c);
          aw2[i].x += d2.x * xh[i].x; aw2[i].y += d2.y * xh[i].y; aw2[i].z += d2.z * xh[i].z; aw2[i].w += d2.w * xh[i].w;
          ab2[i].x += d2.x; ab2[i].y += d2.y; ab2[i].z += d2.z; ab2[i].w += d2.w;
          g[i].x += d2.x * g2[i].x; g[i].y += d2.y * g2[i].y; g[i].z += d2.z * g2[i].z; g[i].w += d2.w * g2[i].w;
        }
        sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        sgx += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      } else {
        xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float mg = wave_sum(sg) / (float)h, mgx = wave_sum(sgx) / (float)h;
    const float istd = 1.0f / rs;
    // cosine distance (mafed_distill_fwd: 1 - st / sqrt((ss + eps)(tt + eps))): d/dx = ca * teacher + cb * x, from three more row sums
    float ca = 0.f, cb = 0.f;
    if (teacher && inj_cos && inj != 0.f) {   // (wave-uniform: one row per wave)
      float ss = 0.f, tt = 0.f, st = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < h) {
          const float4 tv = rte[i];
          const float4 xv = make_float4(mu + xh[i].x * istd, mu + xh[i].y * istd, mu + xh[i].z * istd, mu + xh[i].w * istd);
          ss += (xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w);
          tt += (tv.x * tv.x + tv.y * tv.y) + (tv.z * tv.z + tv.w * tv.w);
          st += (xv.x * tv.x + xv.y * tv.y) + (xv.z * tv.z + xv.w * tv.w);
        }
      }
      ss = wave_sum(ss); tt = wave_sum(tt); st = wave_sum(st);
      const float EPS = 1e-12f;
      const float denom = sqrtf((ss + EPS) * (tt + EPS));
      ca = -inj / denom;
      cb = inj * st * (tt + EPS) / (denom * denom * denom);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < h) {
        float4 o = make_float4(rs * (g[i].x - mg - xh[i].x * mgx), rs * (g[i].y - mg - xh[i].y * mgx),
                               rs * (g[i].z - mg - xh[i].z * mgx), rs * (g[i].w - mg - xh[i].w * mgx));
        if (dres) { o.x += rres[i].x; o.y += rres[i].y; o.z += rres[i].z; o.w += rres[i].w; }
        if (teacher && inj != 0.f) {
          // x = mu + xh / rstd (the row itself is no longer in registers)
          const float4 tv = rte[i];
          const float4 xv = make_float4(mu + xh[i].x * istd, mu + xh[i].y * istd, mu + xh[i].z * istd, mu + xh[i].w * istd);
          if (inj_cos) {
            o.x += ca * tv.x + cb * xv.x; o.y += ca * tv.y + cb * xv.y; o.z += ca * tv.z + cb * xv.z; o.w += ca * tv.w + cb * xv.w;
          } else {
            o.x += inj * (xv.x - tv.x); o.y += inj * (xv.y - tv.y); o.z += inj * (xv.z - tv.z); o.w += inj * (xv.w - tv.w);
          }
        }
        store4(dx + row * h + c, o);
        if (dx_lp) store4(dx_lp + row * h + c, o);
        if (DXSUM) { adx[i].x += o.x; adx[i].y += o.y; adx[i].z += o.z; adx[i].w += o.w; }
      }
    }
  }
  // cross-wave sum of the parameter partials, one array at a time: waves 1..3 stage it to LDS ([3][h] floats = 12 KiB at
  // h = 1024), wave 0 adds and writes the block's slab.  Staging all NP arrays at once needed 60 KiB of LDS, which kept
  // these blocks off every CU that already held two GEMM blocks; at 12 KiB they slot in beside the weight-gradient GEMMs
  // of the side streams (HBM-bound work under MFMA-bound work).
  float* out = partial + (size_t)blockIdx.x * NP * h;
  auto fold = [&](float4 (&acc)[NV], int slot) {
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < h) store4(lds + (size_t)(wave - 1) * h + c, acc[i]);
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < h) {
          float4 a = acc[i];
          for (int w = 0; w < 3; ++w) {
            const float4 t = load4(lds + (size_t)w * h + c);
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
          }
          store4(out + (size_t)slot * h + c, a);
        }
      }
    }
    __syncthreads();
  };
  fold(aw1, 0);
  fold(ab1, 1);
  if constexpr (DUAL) {
    fold(aw2, 2);
    fold(ab2, 3);
  }
  if constexpr (DXSUM) fold(adx, NP - 1);
}

// out_k[c] += sum_b partial[b][k][c]   (k = 0..NP-1 -> dw1, db1, dw2, db2); 64 columns x 4 slab groups per block
__global__ __launch_bounds__(256) void ln_param_reduce_kernel(const float* __restrict__ partial, int nblk, int np, int h,
                                                              float* __restrict__ o0, float* __restrict__ o1,
                                                              float* __restrict__ o2, float* __restrict__ o3, int dxsum_slot,
                                                              float* __restrict__ dxs_a, float* __restrict__ dxs_b) {
  __shared__ float sm[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + col;
  const int tot = np * h;
  float s = 0.f;
  if (idx < tot) {
    // sixteen slabs in flight per thread (this kernel is a chain of dependent-latency loads on the backward's critical path:
    // 512 slabs through 4 accumulators took 13 us, 16 accumulators bring it to the latency of a few round trips)
    float a[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] = 0.f;
    int b = grp;
    for (; b + 60 < nblk; b += 64) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += partial[(size_t)(b + 4 * u) * tot + idx];
    }
    for (; b < nblk; b += 4) a[0] += partial[(size_t)b * tot + idx];
#pragma unroll
    for (int u = 8; u > 0; u >>= 1)
#pragma unroll
      for (int v = 0; v < u; ++v) a[v] += a[v + u];
    s = a[0];
  }
  sm[grp][col] = s;
  __syncthreads();
  if (grp == 0 && idx < tot) {
    const float r = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]);
    const int k = idx / h, c = idx - k * h;
    if (k == dxsum_slot) {
      if (dxs_a) dxs_a[c] += r;
      if (dxs_b) dxs_b[c] += r;
    } else {
      float* o = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3));
      o[c] += r;
    }
  }
}

static int ln_nv(int h) {
  const int need = (h + 255) / 256;
  if (need <= 1) return 1;
  if (need <= 2) return 2;
  if (need <= 3) return 3;
  if (need <= 4) return 4;
  if (need <= 8) return 8;
  return 0;
}

static int ln_bwd_blocks(int64_t rows) {
  int64_t nb = cdiv(rows, LN_ROWS_PER_BLOCK);
  if (nb > 512) nb = 512;
  return (int)nb;
}

}  // namespace mafed

using namespace mafed;

extern "C" int mafed_layernorm_fwd(const float* x, int64_t rows, int h, float eps, const float* w1, const float* b1, void* y1,
                                   const float* w2, const float* b2, void* y2, mafed_dtype out_dtype, float* mean,
                                   float* rstd, void* stream) {
  MAFED_CHECK_ARG(x && w1 && b1 && y1, "layernorm_fwd: null pointer");
  MAFED_CHECK_ARG(rows >= 0 && h > 0 && h % 4 == 0, "layernorm_fwd: h=%d must be a positive multiple of 4", h);
  MAFED_CHECK_ARG((y2 == nullptr) == (w2 == nullptr) && (y2 == nullptr) == (b2 == nullptr), "layernorm_fwd: w2/b2/y2 must be all set or all NULL");
  const int nv = ln_nv(h);
  MAFED_CHECK_ARG(nv > 0, "layernorm_fwd: h=%d > 2048 unsupported", h);
  if (rows == 0) return MAFED_OK;
  dim3 grid((unsigned)cdiv(rows, LN_ROWS_PER_BLOCK)), block(256);
  hipStream_t st = as_stream(stream);
#define LAUNCH(NV, T) \
  launch(K_LN_FWD, (double)rows * h * (4.0 + (y2 ? 2.0 : 1.0) * sizeof(T)), layernorm_fwd_kernel<NV, T>, grid, block, 0, st, x, rows, h, eps, w1, b1, (T*)y1, w2, b2, (T*)y2, mean, rstd)
#define DISPATCH_T(NV)                      \
  if (out_dtype == MAFED_F32) LAUNCH(NV, float); \
  else LAUNCH(NV, bf16_t)
  switch (nv) {
    case 1: DISPATCH_T(1); break;
    case 2: DISPATCH_T(2); break;
    case 3: DISPATCH_T(3); break;
    case 4: DISPATCH_T(4); break;
    default: DISPATCH_T(8); break;
  }
#undef DISPATCH_T
#undef LAUNCH
  MAFED_CHECK_LAUNCH("layernorm_fwd");
  return MAFED_OK;
}

extern "C" size_t mafed_layernorm_bwd_workspace_bytes(int64_t rows, int h) {
  return (size_t)ln_bwd_blocks(rows) * 5 * (size_t)h * sizeof(float);
}

// phase: 0 = both kernels on `stream`; 1 = the row kernel only (dx, dx_lp and the per-block parameter partials in `workspace`);
// 2 = the parameter reduction only (workspace -> dw / db / dxsum, on `stream`: a side stream, off the dX chain)
static int layernorm_bwd_impl(int phase, const void* dy1, const void* dy2, mafed_dtype dy_dtype, const float* x, const float* mean,
                              const float* rstd, const float* w1, const float* w2, int64_t rows, int h,
                              const float* dres, float* dx, void* dx_lp, float* dw1, float* db1, float* dw2, float* db2,
                              const float* teacher, const int64_t* attention_mask, int S, int P, int T,
                              const float* inj_scale_dev, float inj_mul, float* dxsum_a, float* dxsum_b, void* workspace,
                              size_t workspace_bytes, void* stream);

extern "C" int mafed_layernorm_bwd(const void* dy1, const void* dy2, mafed_dtype dy_dtype, const float* x, const float* mean,
                                   const float* rstd, const float* w1, const float* w2, int64_t rows, int h,
                                   const float* dres, float* dx, void* dx_lp, float* dw1, float* db1, float* dw2, float* db2,
                                   const float* teacher, const int64_t* attention_mask, int S, int P, int T,
                                   const float* inj_scale_dev, float inj_mul, float* dxsum_a, float* dxsum_b, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return layernorm_bwd_impl(0, dy1, dy2, dy_dtype, x, mean, rstd, w1, w2, rows, h, dres, dx, dx_lp, dw1, db1, dw2, db2, teacher, attention_mask, S, P,
                            T, inj_scale_dev, inj_mul, dxsum_a, dxsum_b, workspace, workspace_bytes, stream);
}

extern "C" int mafed_layernorm_bwd_rows(const void* dy1, const void* dy2, mafed_dtype dy_dtype, const float* x, const float* mean,
                                        const float* rstd, const float* w1, const float* w2, int64_t rows, int h, const float* dres, float* dx,
                                        void* dx_lp, const float* teacher, const int64_t* attention_mask, int S, int P, int T,
                                        const float* inj_scale_dev, float inj_mul, int want_dxsum, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  float dummy = 0.f;   // the row kernel never dereferences the parameter-gradient pointers: non-null stands for "wanted"
  return layernorm_bwd_impl(1, dy1, dy2, dy_dtype, x, mean, rstd, w1, w2, rows, h, dres, dx, dx_lp, &dummy, &dummy, dy2 ? &dummy : nullptr,
                            dy2 ? &dummy : nullptr, teacher, attention_mask, S, P, T, inj_scale_dev, inj_mul, want_dxsum ? &dummy : nullptr, nullptr,
                            workspace, workspace_bytes, stream);
}

extern "C" int mafed_layernorm_bwd_params(int64_t rows, int h, float* dw1, float* db1, float* dw2, float* db2, float* dxsum_a, float* dxsum_b,
                                          const void* workspace, size_t workspace_bytes, void* stream) {
  float dummy = 0.f;   // operands of the row kernel: only tested for presence in this phase
  return layernorm_bwd_impl(2, &dummy, dw2 ? &dummy : nullptr, MAFED_F32, &dummy, &dummy, &dummy, &dummy, dw2 ? &dummy : nullptr, rows, h, nullptr, &dummy,
                            nullptr, dw1, db1, dw2, db2, nullptr, nullptr, 0, 0, 0, nullptr, 0.f, dxsum_a, dxsum_b, const_cast<void*>(workspace),
                            workspace_bytes, stream);
}

static int layernorm_bwd_impl(int phase, const void* dy1, const void* dy2, mafed_dtype dy_dtype, const float* x, const float* mean,
                              const float* rstd, const float* w1, const float* w2, int64_t rows, int h,
                              const float* dres, float* dx, void* dx_lp, float* dw1, float* db1, float* dw2, float* db2,
                              const float* teacher, const int64_t* attention_mask, int S, int P, int T,
                              const float* inj_scale_dev, float inj_mul, float* dxsum_a, float* dxsum_b, void* workspace,
                              size_t workspace_bytes, void* stream) {
  MAFED_CHECK_ARG(dy1 && x && mean && rstd && w1 && dx && dw1 && db1, "layernorm_bwd: null pointer");
  MAFED_CHECK_ARG(h > 0 && h % 4 == 0, "layernorm_bwd: h=%d must be a positive multiple of 4", h);
  const bool dual = dy2 != nullptr;
  MAFED_CHECK_ARG(!dual || (w2 && dw2 && db2), "layernorm_bwd: dual LN needs w2, dw2, db2");
  MAFED_CHECK_ARG(!teacher || (attention_mask && inj_scale_dev && S > 0 && P >= 0 && T == S - P && rows % S == 0),
                  "layernorm_bwd: distillation injection needs attention_mask, inj_scale, S=P+T, rows %% S == 0");
  const int nv = ln_nv(h);
  MAFED_CHECK_ARG(nv > 0, "layernorm_bwd: h=%d > 2048 unsupported", h);
  if (rows == 0) return MAFED_OK;
  const int nblk = ln_bwd_blocks(rows);
  const bool dxsum = dxsum_a != nullptr || dxsum_b != nullptr;
  const int np = (dual ? 4 : 2) + (dxsum ? 1 : 0);
  if (workspace_bytes < (size_t)nblk * np * h * sizeof(float) || !workspace) {
    set_error("layernorm_bwd: workspace %zu < %zu", workspace_bytes, (size_t)nblk * np * h * sizeof(float));
    return MAFED_EWORKSPACE;
  }
  const size_t lds_bytes = (size_t)3 * h * sizeof(float);
  MAFED_CHECK_ARG(lds_bytes <= 160 * 1024, "layernorm_bwd: LDS staging %zu too large", lds_bytes);
  hipStream_t st = as_stream(stream);
  float* partial = (float*)workspace;
  dim3 grid(nblk), block(256);
#define LAUNCH(NV, T, DUAL, DXS)                                                                                         \
  do {                                                                                                                   \
    auto kfn = layernorm_bwd_kernel<NV, T, DUAL, DXS>;                                                                        \
    if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
    launch(K_LN_BWD, ln_bwd_bytes, kfn, grid, block, lds_bytes, st, (const T*)dy1, (const T*)dy2, x, mean, rstd, w1, w2, rows, h, dres, dx, (T*)dx_lp, \
           teacher, attention_mask, S, P, T_, inj_scale_dev, inj_mul, partial);                                               \
  } while (0)
  const int T_ = T;
  // algorithmic bytes: dy1 (+ dy2) and x in, the residual gradient and the teacher rows where present, dx (+ its low-precision copy) out
  const double esz = dy_dtype == MAFED_F32 ? 4.0 : 2.0;
  const double ln_bwd_bytes = (double)rows * h * ((dual ? 2.0 : 1.0) * esz + 4.0 + (dres ? 4.0 : 0.0) + (teacher ? 4.0 : 0.0) + 4.0 + (dx_lp ? esz : 0.0));
#define DISPATCH_D(NV, TT)                       \
  if (dual && dxsum) LAUNCH(NV, TT, true, true);  \
  else if (dual) LAUNCH(NV, TT, true, false);     \
  else if (dxsum) LAUNCH(NV, TT, false, true);    \
  else LAUNCH(NV, TT, false, false)
#define DISPATCH_T(NV)                          \
  if (dy_dtype == MAFED_F32) { DISPATCH_D(NV, float); } \
  else { DISPATCH_D(NV, bf16_t); }
  if (phase != 2) {
    switch (nv) {
      case 1: DISPATCH_T(1); break;
      case 2: DISPATCH_T(2); break;
      case 3: DISPATCH_T(3); break;
      case 4: DISPATCH_T(4); break;
      default: DISPATCH_T(8); break;
    }
    MAFED_CHECK_LAUNCH("layernorm_bwd");
  }
#undef DISPATCH_T
#undef DISPATCH_D
#undef LAUNCH
  if (phase == 1) return MAFED_OK;
  const int tot = np * h;
  launch(K_LN_BWD_REDUCE, (double)nblk * np * h * 4.0, ln_param_reduce_kernel, dim3((tot + 63) / 64), dim3(256), 0, st, partial, nblk, np, h, dw1,
         db1, dw2, db2, dxsum ? np - 1 : -1, dxsum_a, dxsum_b);
  MAFED_CHECK_LAUNCH("layernorm_bwd(param reduce)");
  return MAFED_OK;
}
