// Small HBM-bound utilities: cast, GELU, embedding gather + image/text concat, column sums (bias gradients).
#include "common.h"

namespace mafed {

template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void cast_kernel(const SrcT* __restrict__ src, DstT* __restrict__ dst, int64_t n4, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) store4(dst + i * 4, load4(src + i * 4));
  // tail (n not a multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    Elem<DstT>::store(dst + i, Elem<SrcT>::load(src + i));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    Elem<T>::store(y + i, gelu_erf(Elem<T>::load(x + i)));
}

// h0[b,s,:] = s < P ? image[b,s,:] : embed_in[input_ids[b,s-P],:]     one wave per token row, 16-byte accesses
template <typename ImgT>
__global__ __launch_bounds__(256) void embed_concat_fwd_kernel(const ImgT* __restrict__ image, const float* __restrict__ embed_in,
                                                               const int64_t* __restrict__ input_ids, int B, int P, int T, int h,
                                                               int64_t V, float* __restrict__ h0) {
  const int S = P + T;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * S) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = row / S;
  const int s = (int)(row - b * S);
  float* dst = h0 + row * h;
  if (s < P) {
    const ImgT* src = image + (b * P + s) * (int64_t)h;
    for (int c = lane * 4; c < h; c += 256) store4(dst + c, load4(src + c));
  } else {
    int64_t id = input_ids[b * T + (s - P)];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const float* src = embed_in + id * h;
    for (int c = lane * 4; c < h; c += 256) store4(dst + c, load4(src + c));
  }
}

template <typename ImgT>
__global__ __launch_bounds__(256) void embed_concat_bwd_kernel(const float* __restrict__ dh0, const int64_t* __restrict__ input_ids,
                                                               int B, int P, int T, int h, int64_t V, ImgT* __restrict__ d_image,
                                                               float* __restrict__ d_embed_in) {
  const int S = P + T;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * S) return;
  const int lane = threadIdx.x & 63;
  const int64_t b = row / S;
  const int s = (int)(row - b * S);
  const float* src = dh0 + row * h;
  if (s < P) {
    if (d_image) {
      ImgT* dst = d_image + (b * P + s) * (int64_t)h;
      for (int c = lane * 4; c < h; c += 256) store4(dst + c, load4(src + c));
    }
  } else if (d_embed_in) {
    int64_t id = input_ids[b * T + (s - P)];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    float* dst = d_embed_in + id * h;
    // duplicate ids (pad id 0, repeated tokens) collide: fp32 atomics, contiguous 256 B per wave-instruction
    for (int c = lane; c < h; c += 64) atomicAdd(dst + c, src[c]);
  }
}

// out[n] += sum_m X[m,n].  Block = 256 rows x 128 columns: 32 column-quads x 8 row-groups of threads, 8/16-byte
// loads, LDS tree over the row groups, then ONE fp32 atomic per column per block (M/256 adders per address, spread
// over N addresses -- far from the contended regime of MI355X_MICROARCH "Global float atomics").
constexpr int CS_ROWS = 256, CS_COLS = 128;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t M, int64_t N, int64_t ldx, float* __restrict__ out) {
  __shared__ float4 sm[8][32];
  const int cq = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int64_t n = (int64_t)blockIdx.x * CS_COLS + cq * 4;
  const int64_t m0 = (int64_t)blockIdx.y * CS_ROWS;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < N) {
    const int64_t m1 = m0 + CS_ROWS < M ? m0 + CS_ROWS : M;
    for (int64_t m = m0 + rg; m < m1; m += 8) {
      const float4 v = load4(X + m * ldx + n);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sm[rg][cq] = s;
  __syncthreads();
  if (rg == 0 && n < N) {
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 v = sm[r][cq];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    atomicAdd(out + n, s.x); atomicAdd(out + n + 1, s.y); atomicAdd(out + n + 2, s.z); atomicAdd(out + n + 3, s.w);
  }
}

static int grid_for(int64_t n, int per_thread = 1) {
  int64_t g = cdiv(n, 256 * (int64_t)per_thread);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- frozen CLIP vision tower: patch embedding as a GEMM (CLIPVisionEmbeddings, clip:148-154, 202-218) ---------------------
// im2col of a stride = kernel convolution: out[b * np + p][c * ps * ps + i * ps + j] = pix[b][c][py * ps + i][px * ps + j],
// zero for the K .. Kpad-1 padding columns (Kpad a multiple of 64 lets the product take the LDS-DMA GEMM) and for pad rows.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* __restrict__ pix, int B, int C, int H, int W, int ps, int gw, int np, int K,
                                                       int Kpad, int64_t rows_out, TO* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows_out * Kpad) return;
  const int64_t row = idx / Kpad;
  const int k = (int)(idx - row * Kpad);
  float v = 0.f;
  if (k < K && row < (int64_t)B * np) {
    const int b = (int)(row / np), p = (int)(row - (int64_t)b * np);
    const int py = p / gw, px = p - py * gw;
    const int c = k / (ps * ps), r = k - c * ps * ps;
    const int i = r / ps, j = r - i * ps;
    v = Elem<TI>::load(pix + (((int64_t)b * C + c) * H + py * ps + i) * W + px * ps + j);
  }
  Elem<TO>::store(out + idx, v);
}

// tokens[b][0] = class_embedding + pos[0]; tokens[b][1 + p] = patch_emb[b * np + p] + pos[1 + p]  (fp32 residual stream)
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_kernel(const T* __restrict__ pe, int64_t ld_pe, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, int B, int np, int h, float* __restrict__ out) {
  const int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t total = (int64_t)B * (np + 1) * h;
  if (idx >= total) return;
  const int64_t row = idx / h;
  const int d = (int)(idx - row * h);
  const int b = (int)(row / (np + 1)), s = (int)(row - (int64_t)b * (np + 1));
  const float4 pv = load4(pos + (int64_t)s * h + d);
  float4 x = s == 0 ? load4(cls + d) : load4(pe + ((int64_t)b * np + s - 1) * ld_pe + d);
  x.x += pv.x; x.y += pv.y; x.z += pv.z; x.w += pv.w;
  store4(out + idx, x);
}

}  // namespace mafed

namespace mafed {
// dst[b, s, :] = s < P ? 0 : src[b, s - P, :]  (fp32), plus the same rows in bf16 when dst_lp != NULL: the gradient of the residual
// stream entering the top layer's backward -- only the T text positions reach the LM head -- and the compute-dtype copy its GEMMs
// read, in one pass (was a 38 MB fill, a strided copy and a cast: three small kernels on the dX chain between forward and backward).
__global__ __launch_bounds__(256) void pad_text_rows_kernel(const float* __restrict__ src, int S, int P, int h4, int64_t n4, float* __restrict__ dst,
                                                            bf16_t* __restrict__ dst_lp) {
  const int T = S - P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / h4;
    const int c = (int)(i - row * h4);
    const int64_t b = row / S;
    const int s_ = (int)(row - b * S);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s_ >= P) v = load4(src + ((b * T + (s_ - P)) * (int64_t)h4 + c) * 4);
    store4(dst + i * 4, v);
    if (dst_lp) store4(dst_lp + i * 4, v);
  }
}
// Row-sparse LM head (training only): the shifted cross-entropy looks at position t of a sample only when labels[b, t + 1] is a real
// token -- four answer tokens of 32 text positions in the VQA batches -- so the head's three GEMMs and the two CE passes need only those
// rows.  One thread per sample lists them: compact slot (b, n) <- text row (b, t), n < Rc - 1 in order of t; labels_c[b, n + 1] is the
// slot's label, i.e. the compact [B, Rc] problem is again a "position n predicts label n + 1" problem and goes through the same
// CE kernels (same per-sample counts, same normalisation).  overflow[0] is set when a sample has more labelled rows than Rc - 1.
__global__ void label_rows_kernel(const int64_t* __restrict__ labels, int B, int T, int Rc, int* __restrict__ row_of_slot,
                                  int* __restrict__ slot_of_row, int64_t* __restrict__ labels_c, int* __restrict__ overflow) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = 0;
  labels_c[(int64_t)b * Rc] = -100;
  for (int t = 0; t < T; ++t) {
    const bool has = t + 1 < T && labels[(int64_t)b * T + t + 1] != -100;
    if (has && n < Rc - 1) {
      row_of_slot[b * Rc + n] = b * T + t;
      slot_of_row[b * T + t] = b * Rc + n;
      labels_c[(int64_t)b * Rc + n + 1] = labels[(int64_t)b * T + t + 1];
      ++n;
    } else {
      slot_of_row[b * T + t] = -1;
      if (has) overflow[0] = 1;
    }
  }
  for (int k = n; k < Rc; ++k) {
    row_of_slot[b * Rc + k] = -1;
    if (k + 1 < Rc) labels_c[(int64_t)b * Rc + k + 1] = -100;
  }
}

// dst[r, :] = idx[r] >= 0 ? src[idx[r], :] : 0   (16-byte pieces; h % 8 == 0 for bf16, % 4 for fp32)
template <typename TT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TT* __restrict__ src, const int* __restrict__ idx, int64_t n_out, int hv,
                                                          TT* __restrict__ dst) {
  constexpr int EPV = 16 / sizeof(TT);
  const int64_t total = n_out * hv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / hv;
    const int c = (int)(i - r * hv);
    const int j = idx[r];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (j >= 0) v = *reinterpret_cast<const uint4*>(src + ((int64_t)j * hv + c) * EPV);
    *reinterpret_cast<uint4*>(dst + i * EPV) = v;
  }
}
}  // namespace mafed

using namespace mafed;

extern "C" int mafed_label_rows(const int64_t* labels, int B, int T, int Rc, int* row_of_slot, int* slot_of_row, int64_t* labels_c, int* overflow,
                                void* stream) {
  MAFED_CHECK_ARG(labels && row_of_slot && slot_of_row && labels_c && overflow && B > 0 && T > 1 && Rc >= 2, "label_rows: bad arguments");
  label_rows_kernel<<<dim3((B + 63) / 64), dim3(64), 0, as_stream(stream)>>>(labels, B, T, Rc, row_of_slot, slot_of_row, labels_c, overflow);
  MAFED_CHECK_LAUNCH("label_rows");
  return MAFED_OK;
}

extern "C" int mafed_gather_rows(const void* src, mafed_dtype dtype, const int* idx, int64_t n_out, int h, void* dst, void* stream) {
  MAFED_CHECK_ARG(src && idx && dst && n_out >= 0 && h > 0, "gather_rows: bad arguments");
  MAFED_CHECK_ARG(h % (dtype == MAFED_F32 ? 4 : 8) == 0, "gather_rows: h=%d must fill 16-byte pieces", h);
  if (n_out == 0) return MAFED_OK;
  const int hv = h / (dtype == MAFED_F32 ? 4 : 8);
  const dim3 grid(grid_for(n_out * hv)), block(256);
  if (dtype == MAFED_F32) gather_rows_kernel<float><<<grid, block, 0, as_stream(stream)>>>((const float*)src, idx, n_out, hv, (float*)dst);
  else gather_rows_kernel<bf16_t><<<grid, block, 0, as_stream(stream)>>>((const bf16_t*)src, idx, n_out, hv, (bf16_t*)dst);
  MAFED_CHECK_LAUNCH("gather_rows");
  return MAFED_OK;
}

extern "C" int mafed_pad_text_rows(const float* src, int B, int S, int P, int h, float* dst, void* dst_lp, void* stream) {
  MAFED_CHECK_ARG(src && dst && B > 0 && S > 0 && P >= 0 && P < S && h > 0 && h % 4 == 0, "pad_text_rows: bad arguments");
  const int64_t n4 = (int64_t)B * S * (h / 4);
  launch(K_CAST, (double)B * h * ((S - P) * 4.0 + S * (4.0 + (dst_lp ? 2.0 : 0.0))), pad_text_rows_kernel, dim3(grid_for(n4)), dim3(256), 0, as_stream(stream), src, S, P,
         h / 4, n4, dst, (bf16_t*)dst_lp);
  MAFED_CHECK_LAUNCH("pad_text_rows");
  return MAFED_OK;
}

extern "C" int mafed_cast(const void* src, mafed_dtype sd, void* dst, mafed_dtype dd, int64_t n, void* stream) {
  MAFED_CHECK_ARG(src && dst && n >= 0, "cast: bad arguments");
  if (n == 0) return MAFED_OK;
  hipStream_t st = as_stream(stream);
  const int64_t n4 = n / 4;
  dim3 grid(grid_for(n4)), block(256);
  const double cbytes = (double)n * ((sd == MAFED_F32 ? 4.0 : 2.0) + (dd == MAFED_F32 ? 4.0 : 2.0));
  if (sd == MAFED_F32 && dd == MAFED_BF16) launch(K_CAST, cbytes, cast_kernel<float, bf16_t>, grid, block, 0, st, (const float*)src, (bf16_t*)dst, n4, n);
  else if (sd == MAFED_BF16 && dd == MAFED_F32) launch(K_CAST, cbytes, cast_kernel<bf16_t, float>, grid, block, 0, st, (const bf16_t*)src, (float*)dst, n4, n);
  else if (sd == MAFED_F32 && dd == MAFED_F32) launch(K_CAST, cbytes, cast_kernel<float, float>, grid, block, 0, st, (const float*)src, (float*)dst, n4, n);
  else launch(K_CAST, cbytes, cast_kernel<bf16_t, bf16_t>, grid, block, 0, st, (const bf16_t*)src, (bf16_t*)dst, n4, n);
  MAFED_CHECK_LAUNCH("cast");
  return MAFED_OK;
}

extern "C" int mafed_gelu(const void* x, void* y, mafed_dtype dtype, int64_t n, void* stream) {
  MAFED_CHECK_ARG(x && y && n >= 0, "gelu: bad arguments");
  if (n == 0) return MAFED_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid(grid_for(n)), block(256);
  if (dtype == MAFED_F32) gelu_kernel<float><<<grid, block, 0, st>>>((const float*)x, (float*)y, n);
  else gelu_kernel<bf16_t><<<grid, block, 0, st>>>((const bf16_t*)x, (bf16_t*)y, n);
  MAFED_CHECK_LAUNCH("gelu");
  return MAFED_OK;
}

extern "C" int mafed_embed_concat_fwd(const void* image, mafed_dtype img_dtype, const float* embed_in, const int64_t* input_ids,
                                      int B, int P, int T, int h, int64_t V, float* h0, void* stream) {
  MAFED_CHECK_ARG(image && embed_in && input_ids && h0, "embed_concat_fwd: null pointer");
  MAFED_CHECK_ARG(B >= 0 && P >= 0 && T >= 0 && h > 0 && h % 4 == 0 && V > 0, "embed_concat_fwd: bad shape (h must be a multiple of 4)");
  const int64_t rows = (int64_t)B * (P + T);
  if (rows == 0) return MAFED_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(rows, 4)), block(256);
  const double ebytes = (double)B * h * (P * (img_dtype == MAFED_F32 ? 4.0 : 2.0) + T * 4.0 + (P + T) * 4.0);
  if (img_dtype == MAFED_F32) launch(K_EMBED_FWD, ebytes, embed_concat_fwd_kernel<float>, grid, block, 0, st, (const float*)image, embed_in, input_ids, B, P, T, h, V, h0);
  else launch(K_EMBED_FWD, ebytes, embed_concat_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)image, embed_in, input_ids, B, P, T, h, V, h0);
  MAFED_CHECK_LAUNCH("embed_concat_fwd");
  return MAFED_OK;
}

extern "C" int mafed_embed_concat_bwd(const float* dh0, const int64_t* input_ids, int B, int P, int T, int h, int64_t V,
                                      void* d_image, mafed_dtype img_dtype, float* d_embed_in, void* stream) {
  MAFED_CHECK_ARG(dh0 && input_ids, "embed_concat_bwd: null pointer");
  MAFED_CHECK_ARG(B >= 0 && P >= 0 && T >= 0 && h > 0 && h % 4 == 0 && V > 0, "embed_concat_bwd: bad shape (h must be a multiple of 4)");
  const int64_t rows = (int64_t)B * (P + T);
  if (rows == 0) return MAFED_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(rows, 4)), block(256);
  const double ebytes = (double)B * h * ((P + T) * 4.0 + P * (img_dtype == MAFED_F32 ? 4.0 : 2.0) + T * 4.0);
  if (img_dtype == MAFED_F32) launch(K_EMBED_BWD, ebytes, embed_concat_bwd_kernel<float>, grid, block, 0, st, dh0, input_ids, B, P, T, h, V, (float*)d_image, d_embed_in);
  else launch(K_EMBED_BWD, ebytes, embed_concat_bwd_kernel<bf16_t>, grid, block, 0, st, dh0, input_ids, B, P, T, h, V, (bf16_t*)d_image, d_embed_in);
  MAFED_CHECK_LAUNCH("embed_concat_bwd");
  return MAFED_OK;
}

extern "C" size_t mafed_colsum_workspace_bytes(int64_t M, int64_t N) { (void)M; (void)N; return 0; }

extern "C" int mafed_colsum(const void* X, mafed_dtype dtype, int64_t M, int64_t N, int64_t ldx, float* out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  MAFED_CHECK_ARG(X && out && M >= 0 && N > 0 && ldx >= N, "colsum: bad arguments");
  MAFED_CHECK_ARG(N % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 7) == 0, "colsum: N and ldx must be multiples of 4, X 8-byte aligned");
  if (M == 0) return MAFED_OK;
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)cdiv(N, CS_COLS), (unsigned)cdiv(M, CS_ROWS)), block(256);
  const double sbytes = (double)M * N * (dtype == MAFED_F32 ? 4.0 : 2.0);
  if (dtype == MAFED_F32) launch(K_COLSUM, sbytes, colsum_kernel<float>, grid, block, 0, st, (const float*)X, M, N, ldx, out);
  else launch(K_COLSUM, sbytes, colsum_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)X, M, N, ldx, out);
  MAFED_CHECK_LAUNCH("colsum");
  return MAFED_OK;
}

extern "C" int mafed_patchify(const void* pixels, mafed_dtype pix_dtype, int B, int C, int H, int W, int patch, int64_t rows_out, int k_pad,
                              void* out, mafed_dtype out_dtype, void* stream) {
  MAFED_CHECK_ARG(pixels && out, "patchify: null pointer");
  MAFED_CHECK_ARG(B > 0 && C > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "patchify: image %dx%d is not a multiple of the patch %d", H, W, patch);
  const int gw = W / patch, np = gw * (H / patch), K = C * patch * patch;
  MAFED_CHECK_ARG(k_pad >= K && rows_out >= (int64_t)B * np, "patchify: k_pad=%d < %d or rows_out too small", k_pad, K);
  const int64_t total = rows_out * k_pad;
  dim3 grid((unsigned)cdiv(total, 256)), block(256);
  hipStream_t st = as_stream(stream);
  const double bytes = (double)B * C * H * W * (pix_dtype == MAFED_F32 ? 4.0 : 2.0) + (double)total * (out_dtype == MAFED_F32 ? 4.0 : 2.0);
  if (pix_dtype == MAFED_F32 && out_dtype == MAFED_BF16)
    launch(K_CAST, bytes, patchify_kernel<float, bf16_t>, grid, block, 0, st, (const float*)pixels, B, C, H, W, patch, gw, np, K, k_pad, rows_out, (bf16_t*)out);
  else if (pix_dtype == MAFED_F32)
    launch(K_CAST, bytes, patchify_kernel<float, float>, grid, block, 0, st, (const float*)pixels, B, C, H, W, patch, gw, np, K, k_pad, rows_out, (float*)out);
  else if (out_dtype == MAFED_BF16)
    launch(K_CAST, bytes, patchify_kernel<bf16_t, bf16_t>, grid, block, 0, st, (const bf16_t*)pixels, B, C, H, W, patch, gw, np, K, k_pad, rows_out, (bf16_t*)out);
  else
    launch(K_CAST, bytes, patchify_kernel<bf16_t, float>, grid, block, 0, st, (const bf16_t*)pixels, B, C, H, W, patch, gw, np, K, k_pad, rows_out, (float*)out);
  MAFED_CHECK_LAUNCH("patchify");
  return MAFED_OK;
}

extern "C" int mafed_vit_assemble(const void* patch_emb, mafed_dtype pe_dtype, int64_t ld_pe, const float* class_embedding,
                                  const float* position_embedding, int B, int num_patches, int h, float* tokens, void* stream) {
  MAFED_CHECK_ARG(patch_emb && class_embedding && position_embedding && tokens, "vit_assemble: null pointer");
  MAFED_CHECK_ARG(B > 0 && num_patches > 0 && h > 0 && h % 4 == 0 && ld_pe >= h && ld_pe % 4 == 0, "vit_assemble: bad shape");
  const int64_t total4 = (int64_t)B * (num_patches + 1) * h / 4;
  dim3 grid((unsigned)cdiv(total4, 256)), block(256);
  hipStream_t st = as_stream(stream);
  const double bytes = (double)B * (num_patches + 1) * h * (4.0 + (pe_dtype == MAFED_F32 ? 4.0 : 2.0));
  if (pe_dtype == MAFED_F32)
    launch(K_EMBED_FWD, bytes, vit_assemble_kernel<float>, grid, block, 0, st, (const float*)patch_emb, ld_pe, class_embedding, position_embedding, B, num_patches, h, tokens);
  else
    launch(K_EMBED_FWD, bytes, vit_assemble_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)patch_emb, ld_pe, class_embedding, position_embedding, B, num_patches, h, tokens);
  MAFED_CHECK_LAUNCH("vit_assemble");
  return MAFED_OK;
}
