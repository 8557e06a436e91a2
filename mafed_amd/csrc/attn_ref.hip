// Exact-fp32 attention kernels (parity mode of the north star's 1e-3 gate, any head size <= 256, and the on-GPU
// cross-check of the MFMA kernels).  One wave per query row (forward / dQ) or per key row (dK, dV); scores are
// lane-parallel over the other sequence index, outputs lane-parallel over the head dimension.
// Semantics: GPTNeoXAttention eager path (tf:154-236): partial rotary on q,k; scale D^-0.5; mask = causal AND
// key-padding; softmax in fp32; fully causal over the whole [image | text] sequence.
#include "attn.h"

namespace mafed {

// element d of the rotated row (tf:111-151): first `rot` dims rotate with the NeoX half pairing d <-> d +- rot/2
template <typename T>
__device__ __forceinline__ float rot_elem(const T* __restrict__ row, int d, int rot, const float* __restrict__ c, const float* __restrict__ s) {
  const float x = Elem<T>::load(row + d);
  if (d >= rot) return x;
  const int half = rot >> 1;
  if (d < half) return x * c[d] - Elem<T>::load(row + d + half) * s[d];
  return x * c[d - half] + Elem<T>::load(row + d - half) * s[d - half];
}
// transpose of the rotation applied to a gradient row g (indexed through LDS)
__device__ __forceinline__ float unrot_elem(const float* __restrict__ g, int d, int rot, const float* __restrict__ c, const float* __restrict__ s) {
  const float x = g[d];
  if (d >= rot) return x;
  const int half = rot >> 1;
  if (d < half) return x * c[d] + g[d + half] * s[d];
  return x * c[d - half] - g[d - half] * s[d - half];
}

__device__ __forceinline__ bool key_valid(const int64_t* __restrict__ am, int b, int j, int P, int T) {
  return j < P || am[(int64_t)b * T + (j - P)] != 0;
}

// LDS per wave: qrow[D] + sc[S] floats
template <typename T>
__global__ __launch_bounds__(256) void attn_ref_fwd_kernel(const T* __restrict__ qkv, AttnShape sh, const float* __restrict__ rc,
                                                           const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                           T* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = sh.S, H = sh.H, D = sh.D, rot = sh.rot, half = sh.rot >> 1;
  const int q = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  float* qrow = lds + (size_t)wave * (D + S);
  float* sc = qrow + D;
  if (q >= S) return;
  const int64_t rstride = (int64_t)H * 3 * D;
  const T* base = qkv + ((int64_t)b * S * H + h) * 3 * D;  // + s * rstride + {0,D,2D}
  const T* qp = base + (int64_t)q * rstride;
  for (int d = lane; d < D; d += 64) qrow[d] = rot_elem(qp, d, rot, rc + (int64_t)q * half, rs + (int64_t)q * half);
  __builtin_amdgcn_wave_barrier();
  const float scale = rsqrtf((float)D);
  float m = -INFINITY;
  const int jlast = sh.causal ? q : S - 1;  // bidirectional: every key
  for (int j = lane; j <= jlast; j += 64) {
    float s = -INFINITY;
    if (key_valid(am, b, j, sh.P, sh.T)) {
      const T* kp = base + (int64_t)j * rstride + D;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc = fmaf(qrow[d], rot_elem(kp, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half), acc);
      s = acc * scale;
    }
    sc[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j <= jlast; j += 64) {
    const float p = expf(sc[j] - m);
    sc[j] = p;
    l += p;
  }
  l = wave_sum(l);
  __builtin_amdgcn_wave_barrier();
  const float inv = 1.0f / l;
  T* op = out + ((int64_t)b * S + q) * H * D + (int64_t)h * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j <= jlast; ++j) acc = fmaf(sc[j], Elem<T>::load(base + (int64_t)j * rstride + 2 * D + d), acc);
    Elem<T>::store(op + d, acc * inv);
  }
  if (lane == 0) lse[((int64_t)b * H + h) * S + q] = m + logf(l);
}

// dQ (+ delta): one wave per query row.  LDS per wave: qrow[D] + dorow[D] + gq[D] + sc[S]
template <typename T>
__global__ __launch_bounds__(256) void attn_ref_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ out, const T* __restrict__ dout,
                                                              const float* __restrict__ lse, AttnShape sh, const float* __restrict__ rc,
                                                              const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                              T* __restrict__ dqkv, float* __restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = sh.S, H = sh.H, D = sh.D, rot = sh.rot, half = sh.rot >> 1;
  const int q = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  float* qrow = lds + (size_t)wave * (3 * D + S);
  float* dorow = qrow + D;
  float* gq = dorow + D;
  float* sc = gq + D;
  if (q >= S) return;
  const int64_t rstride = (int64_t)H * 3 * D;
  const T* base = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const T* qp = base + (int64_t)q * rstride;
  const T* op = out + ((int64_t)b * S + q) * H * D + (int64_t)h * D;
  const T* dop = dout + ((int64_t)b * S + q) * H * D + (int64_t)h * D;
  float dl = 0.f;
  for (int d = lane; d < D; d += 64) {
    qrow[d] = rot_elem(qp, d, rot, rc + (int64_t)q * half, rs + (int64_t)q * half);
    const float g = Elem<T>::load(dop + d);
    dorow[d] = g;
    dl += g * Elem<T>::load(op + d);
  }
  dl = wave_sum(dl);
  if (lane == 0) delta[((int64_t)b * H + h) * S + q] = dl;
  __builtin_amdgcn_wave_barrier();
  const float scale = rsqrtf((float)D);
  const float L = lse[((int64_t)b * H + h) * S + q];
  for (int j = lane; j <= q; j += 64) {
    float ds = 0.f;
    if (key_valid(am, b, j, sh.P, sh.T)) {
      const T* kp = base + (int64_t)j * rstride + D;
      const T* vp = kp + D;
      float s = 0.f, dp = 0.f;
      for (int d = 0; d < D; ++d) {
        s = fmaf(qrow[d], rot_elem(kp, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half), s);
        dp = fmaf(dorow[d], Elem<T>::load(vp + d), dp);
      }
      const float p = expf(s * scale - L);
      ds = p * (dp - dl) * scale;
    }
    sc[j] = ds;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j <= q; ++j) {
      const T* kp = base + (int64_t)j * rstride + D;
      acc = fmaf(sc[j], rot_elem(kp, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half), acc);
    }
    gq[d] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  T* dqp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)q * rstride;
  for (int d = lane; d < D; d += 64) Elem<T>::store(dqp + d, unrot_elem(gq, d, rot, rc + (int64_t)q * half, rs + (int64_t)q * half));
}

// dK, dV: one wave per key row j; queries q >= j.  LDS per wave: krow[D] + vrow[D] + gk[D] + pq[S] + dsq[S]
template <typename T>
__global__ __launch_bounds__(256) void attn_ref_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               AttnShape sh, const float* __restrict__ rc, const float* __restrict__ rs,
                                                               const int64_t* __restrict__ am, T* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = sh.S, H = sh.H, D = sh.D, rot = sh.rot, half = sh.rot >> 1;
  const int j = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  float* krow = lds + (size_t)wave * (3 * D + 2 * S);
  float* vrow = krow + D;
  float* gk = vrow + D;
  float* pq = gk + D;
  float* dsq = pq + S;
  if (j >= S) return;
  const int64_t rstride = (int64_t)H * 3 * D;
  const T* base = qkv + ((int64_t)b * S * H + h) * 3 * D;
  T* dkp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)j * rstride + D;
  T* dvp = dkp + D;
  if (!key_valid(am, b, j, sh.P, sh.T)) {  // masked key: no query attends to it
    for (int d = lane; d < D; d += 64) { Elem<T>::store(dkp + d, 0.f); Elem<T>::store(dvp + d, 0.f); }
    return;
  }
  const T* kp = base + (int64_t)j * rstride + D;
  for (int d = lane; d < D; d += 64) {
    krow[d] = rot_elem(kp, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half);
    vrow[d] = Elem<T>::load(kp + D + d);
  }
  __builtin_amdgcn_wave_barrier();
  const float scale = rsqrtf((float)D);
  for (int q = j + lane; q < S; q += 64) {
    const T* qp = base + (int64_t)q * rstride;
    const T* dop = dout + ((int64_t)b * S + q) * H * D + (int64_t)h * D;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < D; ++d) {
      s = fmaf(rot_elem(qp, d, rot, rc + (int64_t)q * half, rs + (int64_t)q * half), krow[d], s);
      dp = fmaf(Elem<T>::load(dop + d), vrow[d], dp);
    }
    const int64_t li = ((int64_t)b * H + h) * S + q;
    const float p = expf(s * scale - lse[li]);
    pq[q] = p;
    dsq[q] = p * (dp - delta[li]) * scale;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < D; d += 64) {
    float av = 0.f, ak = 0.f;
    for (int q = j; q < S; ++q) {
      const T* qp = base + (int64_t)q * rstride;
      const T* dop = dout + ((int64_t)b * S + q) * H * D + (int64_t)h * D;
      av = fmaf(pq[q], Elem<T>::load(dop + d), av);
      ak = fmaf(dsq[q], rot_elem(qp, d, rot, rc + (int64_t)q * half, rs + (int64_t)q * half), ak);
    }
    Elem<T>::store(dvp + d, av);
    gk[d] = ak;
  }
  __builtin_amdgcn_wave_barrier();
  for (int d = lane; d < D; d += 64) Elem<T>::store(dkp + d, unrot_elem(gk, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half));
}

template <typename T>
int attn_ref_fwd_launch(const void* qkv, const AttnShape& sh, const float* rc, const float* rs, const int64_t* am, void* out, float* lse,
                        hipStream_t st) {
  const size_t lds = (size_t)4 * (sh.D + sh.S) * sizeof(float);
  if (lds > 160 * 1024) { set_error("attn_fwd(f32): S=%d too long for the parity kernel", sh.S); return MAFED_EINVAL; }
  auto k = attn_ref_fwd_kernel<T>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<<<dim3((sh.S + 3) / 4, sh.H, sh.B), dim3(256), lds, st>>>((const T*)qkv, sh, rc, rs, am, (T*)out, lse);
  return MAFED_OK;
}

template <typename T>
int attn_ref_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, const AttnShape& sh, const float* rc,
                        const float* rs, const int64_t* am, void* dqkv, float* delta, hipStream_t st) {
  const size_t lds1 = (size_t)4 * (3 * sh.D + sh.S) * sizeof(float), lds2 = (size_t)4 * (3 * sh.D + 2 * sh.S) * sizeof(float);
  if (lds2 > 160 * 1024) { set_error("attn_bwd(f32): S=%d too long for the parity kernel", sh.S); return MAFED_EINVAL; }
  auto k1 = attn_ref_bwd_dq_kernel<T>;
  auto k2 = attn_ref_bwd_dkv_kernel<T>;
  if (lds1 > 64 * 1024) (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
  if (lds2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  dim3 grid((sh.S + 3) / 4, sh.H, sh.B), block(256);
  k1<<<grid, block, lds1, st>>>((const T*)qkv, (const T*)out, (const T*)dout, lse, sh, rc, rs, am, (T*)dqkv, delta);
  k2<<<grid, block, lds2, st>>>((const T*)qkv, (const T*)dout, lse, delta, sh, rc, rs, am, (T*)dqkv);
  return MAFED_OK;
}

// ------------------------------------------------------------------------------------------------------------
// KV-cached decode: ONE new query per sample against every earlier key.  The cache is the
// [B,S0,H,3,D] qkv tensor the prefill's fused QKV GEMM left behind (k stays un-rotated; the rotation is applied on load,
// position = key index, exactly like the training kernels) plus a small [B,cap,H,3,D] tensor that receives one row per
// generated token.  One wave per (batch, head): scores lane-parallel over keys, output lane-parallel over the head dim.
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ qkv_pre, int S0, const T* __restrict__ qkv_new, int cap, int t,
                                                          int B, int H, int D, int rot, int P, int Tm, const float* __restrict__ rc,
                                                          const float* __restrict__ rs, const int64_t* __restrict__ am, T* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int id = blockIdx.x * 4 + wave;
  const int nk = S0 + t + 1, half = rot >> 1;
  float* qrow = lds + (size_t)wave * (D + nk);
  float* sc = qrow + D;
  if (id >= B * H) return;
  const int b = id / H, h = id - b * H;
  const int64_t rstride = (int64_t)H * 3 * D;
  const T* pre = qkv_pre + ((int64_t)b * S0 * H + h) * 3 * D;
  const T* neu = qkv_new + ((int64_t)b * cap * H + h) * 3 * D;
  const int qpos = S0 + t;
  const T* qp = neu + (int64_t)t * rstride;
  for (int d = lane; d < D; d += 64) qrow[d] = rot_elem(qp, d, rot, rc + (int64_t)qpos * half, rs + (int64_t)qpos * half);
  __builtin_amdgcn_wave_barrier();
  const float scale = rsqrtf((float)D);
  float m = -INFINITY;
  for (int j = lane; j < nk; j += 64) {
    float s = -INFINITY;
    if (j >= S0 || key_valid(am, b, j, P, Tm)) {
      const T* kp = (j < S0 ? pre + (int64_t)j * rstride : neu + (int64_t)(j - S0) * rstride) + D;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc = fmaf(qrow[d], rot_elem(kp, d, rot, rc + (int64_t)j * half, rs + (int64_t)j * half), acc);
      s = acc * scale;
    }
    sc[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < nk; j += 64) {
    const float p = expf(sc[j] - m);
    sc[j] = p;
    l += p;
  }
  l = wave_sum(l);
  __builtin_amdgcn_wave_barrier();
  const float inv = 1.0f / l;
  T* op = out + (int64_t)b * H * D + (int64_t)h * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < S0; ++j) acc = fmaf(sc[j], Elem<T>::load(pre + (int64_t)j * rstride + 2 * D + d), acc);
    for (int j = S0; j < nk; ++j) acc = fmaf(sc[j], Elem<T>::load(neu + (int64_t)(j - S0) * rstride + 2 * D + d), acc);
    Elem<T>::store(op + d, acc * inv);
  }
}

// Block-per-(batch, head) form of the same step, used whenever rot % 16 == 0 (every VLPythia head size): 256 threads score one
// key each with 16-byte row loads (rotary applied chunk-wise to the first rot dims), a block-wide softmax, then thread
// (key group, 8-dim chunk) accumulates p.V with 16-byte loads and the key groups are folded through LDS.  The wave-per-head
// kernel above did 2-byte loads in dependent chains: 112 us per layer at B = 32, H = 16, ~300 keys; this one is bound by
// the 39 MB of K/V it streams.
template <typename T>
__device__ __forceinline__ void load_row8(const T* __restrict__ p, float (&v)[8]);
template <>
__device__ __forceinline__ void load_row8<float>(const float* __restrict__ p, float (&v)[8]) {
  const float4 a = load4(p), b = load4(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load_row8<bf16_t>(const bf16_t* __restrict__ p, float (&v)[8]) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

template <typename T>
__device__ __forceinline__ void store_row8(T* __restrict__ p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store_row8<float>(float* __restrict__ p, const float (&v)[8]) {
  store4(p, make_float4(v[0], v[1], v[2], v[3]));
  store4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}
template <>
__device__ __forceinline__ void store_row8<bf16_t>(bf16_t* __restrict__ p, const float (&v)[8]) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  r.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  r.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  r.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = r;
}

// chunk c (8 dims) of a q / k row, rotated for position pos (tf:111-151); rot % 16 == 0
template <typename T>
__device__ __forceinline__ void load_chunk_rot8(const T* __restrict__ row, int c, int rot, const float* __restrict__ rc, const float* __restrict__ rs,
                                                int pos, float (&o)[8]) {
  load_row8<T>(row + c * 8, o);
  if (c * 8 >= rot) return;
  const int hc = rot >> 4, half = rot >> 1;
  const bool first = c < hc;
  float y[8];
  load_row8<T>(row + (first ? c + hc : c - hc) * 8, y);
  const float* cp = rc + (int64_t)pos * half + (first ? c : c - hc) * 8;
  const float* sp = rs + (int64_t)pos * half + (first ? c : c - hc) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = first ? o[e] * cp[e] - y[e] * sp[e] : o[e] * cp[e] + y[e] * sp[e];
}

__device__ __forceinline__ void unpack_bf16x8(const uint4& r, float (&v)[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

// One pass over the K/V cache: `chunks` = D/8 lanes share a key row (lane c takes the 16-byte chunk c of k AND of v: the row's k | v
// are 4*D contiguous bytes), 256/chunks rows per step and four steps of loads in flight per thread; the row's score is folded over
// those lanes with DPP/permute adds and every row group keeps an online-softmax state (m, l, acc[8]) that the block merges at the
// end.  Replaces the thread-per-key / exp / V three-phase form (19 us at B = 32, S = 288: its 256 threads covered 289+ keys in two
// dependent rounds, the second one with 33 busy lanes).
// PREROT (round 4): the cache holds ROTATED keys -- the prefix was rotated in place once behind the prefill (rotate_k_rows_kernel), every
// generated row by the step that appended it (this kernel rotates row t, uses it from LDS and writes it back for the later steps).  A step
// then loads k and v only: the on-load form fetched the rotary partner chunk and 64 bytes of cos / sin per lane and row on top of the
// 32 bytes of k | v -- most of the load instructions of a kernel that is bound by how many loads it keeps in flight.
template <typename T, int D, bool PREROT = false>
__global__ __launch_bounds__(256) void attn_decode_fused_kernel(const T* __restrict__ qkv_pre, int S0, T* __restrict__ qkv_new, int cap, int t,
                                                                int H, int rot, int P, int Tm, const float* __restrict__ rc,
                                                                const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                                T* __restrict__ out) {
  constexpr int chunks = D / 8, groups = 256 / chunks;
  __shared__ float q_s[D];
  __shared__ float knew_s[D];
  __shared__ float red[groups][D];
  __shared__ float ml[groups][2];
  const int tid = threadIdx.x;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nk = S0 + t + 1;
  const int64_t rstride = (int64_t)H * 3 * D;
  const T* pre = qkv_pre + ((int64_t)b * S0 * H + h) * 3 * D;
  T* neu = qkv_new + ((int64_t)b * cap * H + h) * 3 * D;
  const int c = tid % chunks, kg = tid / chunks;
  const float scale = rsqrtf((float)D);
  float qr[8];
  constexpr int UNR = D == 64 ? (PREROT ? 10 : 5) : (PREROT ? 8 : 4);  // rows in flight per thread: 5 x 32 row groups cover S <= 320 keys in two steps (pre-rotated cache: one)
  float m = -INFINITY, l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // Every load of a step is unconditional (clamped row, partner chunk and cos / sin rows fetched by every lane, the mask word too) and
  // the selects come afterwards: with the loads inside `if (valid)` / `if (chunk < rot)` regions the compiler drained vmcnt(0) at the
  // end of each region -- mask word -> k chunk -> rotary operands -> v chunk became four dependent round trips per row.
  const int hc = rot >> 4, half = rot >> 1;
  const bool inrot = c * 8 < rot, first = c < hc;
  const int cpart = inrot ? (first ? c + hc : c - hc) : c;     // rotary partner chunk (itself outside the rotary range)
  const int ccs = inrot ? (first ? c : c - hc) * 8 : 0;       // offset into the cos / sin row
  const float sgn = first ? -1.f : 1.f;
  int j0 = kg;
  do {  // at least one step per thread (rows past nk are clamped loads with p = 0): the barrier below is reached by every thread
    uint4 kraw[UNR], praw[PREROT ? 1 : UNR], vraw[UNR];
    float4 cs0[PREROT ? 1 : UNR], cs1[PREROT ? 1 : UNR], sn0[PREROT ? 1 : UNR], sn1[PREROT ? 1 : UNR];
    int64_t amv[UNR];
    int jj[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * groups;
      const int jc = j < nk ? j : nk - 1;
      jj[u] = j;
      const T* row = jc < S0 ? pre + (int64_t)jc * rstride : neu + (int64_t)(jc - S0) * rstride;
      if constexpr (sizeof(T) == 2) {
        kraw[u] = *reinterpret_cast<const uint4*>(row + D + c * 8);
        if constexpr (!PREROT) praw[u] = *reinterpret_cast<const uint4*>(row + D + cpart * 8);
        vraw[u] = *reinterpret_cast<const uint4*>(row + 2 * D + c * 8);
      }
      if constexpr (!PREROT) {
        const float* cp = rc + (int64_t)jc * half + ccs;
        const float* sp = rs + (int64_t)jc * half + ccs;
        cs0[u] = load4(cp); cs1[u] = load4(cp + 4); sn0[u] = load4(sp); sn1[u] = load4(sp + 4);
      }
      const int ti = jc >= P && jc < S0 ? jc - P : 0;
      amv[u] = am[(int64_t)b * Tm + ti];
    }
    if (j0 == kg) {
      // the query row is fetched and rotated behind the first step's K / V requests (they do not depend on it); every thread takes
      // this branch in its first iteration, so the barrier is reached by the whole block
      if (tid < chunks) {
        float v[8];
        load_chunk_rot8<T>(neu + (int64_t)t * rstride, tid, rot, rc, rs, S0 + t, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) q_s[tid * 8 + e] = v[e];
        if constexpr (PREROT) {
          // this step's own key row: rotated here (it arrives un-rotated from the QKV GEMM), used from LDS below, written back rotated
          // for the steps to come (this block is the only reader and writer of the (b, h) slice of row t)
          float kv[8];
          load_chunk_rot8<T>(neu + (int64_t)t * rstride + D, tid, rot, rc, rs, S0 + t, kv);
#pragma unroll
          for (int e = 0; e < 8; ++e) knew_s[tid * 8 + e] = kv[e];
        }
      }
      __syncthreads();
      if constexpr (PREROT) {
        if (tid < chunks) {
          float kv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) kv[e] = knew_s[tid * 8 + e];
          store_row8<T>(neu + (int64_t)t * rstride + D + tid * 8, kv);
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) qr[e] = q_s[c * 8 + e] * scale;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = jj[u];
      const bool ok = j < nk && (j < P || j >= S0 || amv[u] != 0);
      float x[8], y[8], vv[8];
      if constexpr (sizeof(T) == 2) {
        unpack_bf16x8(kraw[u], x);
        if constexpr (!PREROT) unpack_bf16x8(praw[u], y);
        unpack_bf16x8(vraw[u], vv);
      } else {
        const int jc = j < nk ? j : nk - 1;
        const T* row = jc < S0 ? pre + (int64_t)jc * rstride : neu + (int64_t)(jc - S0) * rstride;
        load_row8<T>(row + D + c * 8, x);
        if constexpr (!PREROT) load_row8<T>(row + D + cpart * 8, y);
        load_row8<T>(row + 2 * D + c * 8, vv);
      }
      float s = 0.f;
      if constexpr (PREROT) {
        const bool own = j == nk - 1;   // the row this step appended: its rotated key is in LDS (the copy in memory may still be the un-rotated one)
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qr[e], own ? knew_s[c * 8 + e] : x[e], s);
      } else {
        const float cs[8] = {cs0[u].x, cs0[u].y, cs0[u].z, cs0[u].w, cs1[u].x, cs1[u].y, cs1[u].z, cs1[u].w};
        const float sn[8] = {sn0[u].x, sn0[u].y, sn0[u].z, sn0[u].w, sn1[u].x, sn1[u].y, sn1[u].z, sn1[u].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float kr = inrot ? x[e] * cs[e] + sgn * y[e] * sn[e] : x[e];
          s = fmaf(qr[e], kr, s);
        }
      }
#pragma unroll
      for (int o = 1; o < chunks; o <<= 1) s += __shfl_xor(s, o, 64);
      // masked or out-of-range rows: p = 0 and the state is left as it is (selects, no branches)
      const float mn = ok ? fmaxf(m, s) : m;
      const float corr = (ok && m != -INFINITY) ? expf(m - mn) : (ok ? 0.f : 1.f);
      const float p = ok ? expf(s - mn) : 0.f;
      l = l * corr + p;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(acc[e], corr, p * vv[e]);
      m = mn;
    }
    j0 += UNR * groups;
  } while (j0 < nk);
#pragma unroll
  for (int e = 0; e < 8; ++e) red[kg][c * 8 + e] = acc[e];
  if (c == 0) {
    ml[kg][0] = m;
    ml[kg][1] = l;
  }
  __syncthreads();
  if (tid < D) {
    float mx = -INFINITY;
#pragma unroll 8
    for (int g2 = 0; g2 < groups; ++g2) mx = fmaxf(mx, ml[g2][0]);
    float o = 0.f, lt = 0.f;
#pragma unroll 8
    for (int g2 = 0; g2 < groups; ++g2) {
      const float w = ml[g2][0] == -INFINITY ? 0.f : expf(ml[g2][0] - mx);  // a group whose rows were all masked contributes nothing
      o = fmaf(w, red[g2][tid], o);
      lt = fmaf(w, ml[g2][1], lt);
    }
    Elem<T>::store(out + (int64_t)b * H * D + (int64_t)h * D + tid, o / lt);
  }
}

// Round 4, second form for the pre-rotated bf16 cache: every key row of the (b, h) slice is in flight at once (UNR x 256/chunks rows cover
// nk), so the softmax needs no running state -- scores first, one block-wide max, then exp and the V sum; the row groups are merged with
// register shuffles inside a wave and four LDS rows across the waves (the online form above merged 32 (m, l, acc) states with 32 dependent
// expf per output element, and carried two expf per key row).  The prompt mask (Tm <= 256 text positions) is read once into LDS.
__device__ long long* g_decode_trace_dev = nullptr;   // tools: stamps of the next decode attention launches (mafed_attn_decode_set_trace)

template <int D, int UNR>
__global__ __launch_bounds__(256) void attn_decode_flat_kernel(const bf16_t* __restrict__ qkv_pre, int S0, bf16_t* __restrict__ qkv_new, int cap, int t,
                                                               int H, int rot, int P, int Tm, const float* __restrict__ rc,
                                                               const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                               bf16_t* __restrict__ out) {
  constexpr int chunks = D / 8, groups = 256 / chunks;
  __shared__ float red[4][D + 1];
  __shared__ float wmax[4];
  __shared__ unsigned char msk[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  long long* tr = g_decode_trace_dev ? g_decode_trace_dev + ((size_t)b * gridDim.x + h) * 8 : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
  const int nk = S0 + t + 1;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* pre = qkv_pre + ((int64_t)b * S0 * H + h) * 3 * D;
  bf16_t* neu = qkv_new + ((int64_t)b * cap * H + h) * 3 * D;
  const int c = tid % chunks, kg = tid / chunks;
  // first in the queue (they come back first): the prompt mask word and this step's own q | k row -- every lane fetches chunk c of both
  // and rotates it in registers (no wave waits on another one's round trip; all of it L2 hits after the first wave of the block)
  const int64_t mword = am[(int64_t)b * Tm + (tid < Tm ? tid : Tm - 1)];   // (clamped, not predicated: a predicated load drains vmcnt(0) at the end of its region)
  const int hc = rot >> 4, half = rot >> 1;
  const bool inrot = c * 8 < rot, first = c < hc;
  const int cpart = inrot ? (first ? c + hc : c - hc) : c;
  const int ccs = inrot ? (first ? c : c - hc) * 8 : 0;
  const bf16_t* qrow = neu + (int64_t)t * rstride;
  const uint4 q0 = *reinterpret_cast<const uint4*>(qrow + c * 8), q1 = *reinterpret_cast<const uint4*>(qrow + cpart * 8);
  const uint4 k0 = *reinterpret_cast<const uint4*>(qrow + D + c * 8), k1 = *reinterpret_cast<const uint4*>(qrow + D + cpart * 8);
  const float* cp = rc + (int64_t)(S0 + t) * half + ccs;
  const float* sp = rs + (int64_t)(S0 + t) * half + ccs;
  const float4 cs0 = load4(cp), cs1 = load4(cp + 4), sn0 = load4(sp), sn1 = load4(sp + 4);
  uint4 kraw[UNR], vraw[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {   // unconditional clamped loads: every key row of the slice in flight before anything waits
    const int j = kg + u * groups;
    const int jc = j < nk ? j : nk - 1;
    const bf16_t* row = jc < S0 ? pre + (int64_t)jc * rstride : neu + (int64_t)(jc - S0) * rstride;
    kraw[u] = *reinterpret_cast<const uint4*>(row + D + c * 8);
    vraw[u] = *reinterpret_cast<const uint4*>(row + 2 * D + c * 8);
  }
  if (tid < Tm) msk[tid] = mword != 0;
  const float scale = rsqrtf((float)D);
  float qr[8], knew[8];
  {
    float a0[8], a1[8], b0[8], b1[8];
    unpack_bf16x8(q0, a0); unpack_bf16x8(q1, a1); unpack_bf16x8(k0, b0); unpack_bf16x8(k1, b1);
    const float cs[8] = {cs0.x, cs0.y, cs0.z, cs0.w, cs1.x, cs1.y, cs1.z, cs1.w};
    const float sn[8] = {sn0.x, sn0.y, sn0.z, sn0.w, sn1.x, sn1.y, sn1.z, sn1.w};
    const float sgn = first ? -1.f : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qr[e] = (inrot ? a0[e] * cs[e] + sgn * a1[e] * sn[e] : a0[e]) * scale;
      knew[e] = inrot ? b0[e] * cs[e] + sgn * b1[e] * sn[e] : b0[e];
    }
  }
  if (tr && tid == 0) tr[1] = wall_clock64();   // q | k row rotated (first loads back)
  __syncthreads();   // mask bytes; also: every lane has read row t's un-rotated key before the write-back below
  if (tid < chunks) store_row8<bf16_t>(neu + (int64_t)t * rstride + D + tid * 8, knew);   // rotated, for the steps to come (only this block touches the slice)
  float sc[UNR];
  float tmax = -INFINITY;
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = kg + u * groups;
    float x[8];
    unpack_bf16x8(kraw[u], x);
    const bool own = j == nk - 1;   // the row this step appended: its rotated key is in LDS (the copy in memory may still be the un-rotated one)
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(qr[e], own ? knew[e] : x[e], s);
#pragma unroll
    for (int o = 1; o < chunks; o <<= 1) s += __shfl_xor(s, o, 64);
    const int ti = j >= P && j < S0 ? j - P : 0;
    const bool ok = j < nk && (j < P || j >= S0 || msk[ti] != 0);
    sc[u] = ok ? s : -INFINITY;
    tmax = fmaxf(tmax, sc[u]);
  }
#pragma unroll
  for (int o = chunks; o < 64; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
  if (tr && tid == 0) tr[2] = wall_clock64();   // scores done (all K rows in)
  if (lane == 0) wmax[wave] = tmax;
  __syncthreads();
  float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  if (mx == -INFINITY) mx = 0.f;   // every key masked: p = 0 everywhere, the output is 0
  float l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    float vv[8];
    unpack_bf16x8(vraw[u], vv);
    const float p = __expf(sc[u] - mx);
    l += p;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vv[e], acc[e]);
  }
#pragma unroll
  for (int o = chunks; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (tr && tid == 0) tr[3] = wall_clock64();   // V sum and shuffles done
  if (lane < chunks) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c * 8 + e] = acc[e];
    if (lane == 0) red[wave][D] = l;
  }
  __syncthreads();
  if (tid < D) {
    const float o = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    const float lt = (red[0][D] + red[1][D]) + (red[2][D] + red[3][D]);
    out[(int64_t)b * H * D + (int64_t)h * D + tid] = f32_to_bf16(lt > 0.f ? o / lt : 0.f);
  }
  if (tr && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[4] = wall_clock64(); }
}

// In place: k part of every row of a [B,S,H,3,D] qkv tensor rotated for its position (row index within the sample); rot % 16 == 0.
template <typename T>
__global__ __launch_bounds__(256) void rotate_k_rows_kernel(T* __restrict__ qkv, int64_t rows, int S, int H, int D, int rot, const float* __restrict__ rc,
                                                            const float* __restrict__ rs) {
  const int hc = rot >> 4;                       // 8-element chunks in half the rotary range; chunk pair (c, c + hc) is one work item
  const int64_t items = rows * H * hc, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= items) return;
  const int c = (int)(i % hc);
  const int64_t rh = i / hc, row = rh / H;
  const int h = (int)(rh - row * H), pos = (int)(row % S), half = rot >> 1;
  T* kp = qkv + (row * H + h) * 3 * D + D;
  float a[8], b[8];
  load_row8<T>(kp + c * 8, a);
  load_row8<T>(kp + (c + hc) * 8, b);
  const float* cp = rc + (int64_t)pos * half + c * 8;
  const float* sp = rs + (int64_t)pos * half + c * 8;
  float lo[8], hi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { lo[e] = a[e] * cp[e] - b[e] * sp[e]; hi[e] = b[e] * cp[e] + a[e] * sp[e]; }
  store_row8<T>(kp + c * 8, lo);
  store_row8<T>(kp + (c + hc) * 8, hi);
}

template <typename T>
int rotate_k_rows_launch(void* qkv, int64_t rows, int S, int H, int D, int rot, const float* rc, const float* rs, hipStream_t st) {
  if (rot == 0) return MAFED_OK;
  if (rot % 16 != 0) { set_error("rotate_k_rows: rot %% 16 != 0"); return MAFED_EINVAL; }
  const int64_t items = rows * H * (rot >> 4);
  rotate_k_rows_kernel<T><<<dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st>>>((T*)qkv, rows, S, H, D, rot, rc, rs);
  return MAFED_OK;
}
template int rotate_k_rows_launch<float>(void*, int64_t, int, int, int, int, const float*, const float*, hipStream_t);
template int rotate_k_rows_launch<bf16_t>(void*, int64_t, int, int, int, int, const float*, const float*, hipStream_t);

int g_attn_decode_flat = 1;   // mafed_gemm_set_variant(740 / 741)
int attn_decode_set_trace(void* buf) {
  long long* p = (long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_decode_trace_dev), &p, sizeof(p)) == hipSuccess ? MAFED_OK : MAFED_ELAUNCH;
}

template <typename T>
int attn_decode_launch(const void* qkv_pre, int S0, const void* qkv_new, int cap, int t, int B, int H, int D, int rot, int P, int Tm,
                       const float* rc, const float* rs, const int64_t* am, void* out, hipStream_t st, bool prerot) {
  if (rot % 16 == 0 && (D == 64 || D == 128 || D == 256)) {
#define GO(DV, PR) attn_decode_fused_kernel<T, DV, PR><<<dim3(H, B), dim3(256), 0, st>>>((const T*)qkv_pre, S0, (T*)const_cast<void*>(qkv_new), cap, t, H, rot, P, Tm, rc, rs, am, (T*)out)
    if constexpr (sizeof(T) == 2) {
      // every key row in flight at once when the slice fits (see attn_decode_flat_kernel); g_attn_decode_flat = 0 keeps the online form (A/B)
      if (prerot && g_attn_decode_flat && Tm <= 256 && (D == 64 || D == 128)) {
        const int nk = S0 + t + 1, need = (nk + (256 / (D / 8)) - 1) / (256 / (D / 8));
#define GOF(DV, UV) attn_decode_flat_kernel<DV, UV><<<dim3(H, B), dim3(256), 0, st>>>((const bf16_t*)qkv_pre, S0, (bf16_t*)const_cast<void*>(qkv_new), cap, t, H, rot, P, Tm, rc, rs, am, (bf16_t*)out)
        if (D == 64 && need <= 10) { GOF(64, 10); return MAFED_OK; }
        if (D == 64 && need <= 16) { GOF(64, 16); return MAFED_OK; }
        if (D == 64 && need <= 24) { GOF(64, 24); return MAFED_OK; }
        if (D == 128 && need <= 12) { GOF(128, 12); return MAFED_OK; }
        if (D == 128 && need <= 24) { GOF(128, 24); return MAFED_OK; }
#undef GOF
      }
    }
    if (prerot) {
      if (D == 64) GO(64, true);
      else if (D == 128) GO(128, true);
      else GO(256, true);
    } else {
      if (D == 64) GO(64, false);
      else if (D == 128) GO(128, false);
      else GO(256, false);
    }
#undef GO
    return MAFED_OK;
  }
  if (prerot) { set_error("attn_decode: the pre-rotated cache needs rot %% 16 == 0 and a head size of 64 / 128 / 256"); return MAFED_EINVAL; }
  const size_t lds = (size_t)4 * (D + S0 + t + 1) * sizeof(float);
  if (lds > 160 * 1024) { set_error("attn_decode: %d keys too many for this kernel", S0 + t + 1); return MAFED_EINVAL; }
  auto k = attn_decode_kernel<T>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<<<dim3((B * H + 3) / 4), dim3(256), lds, st>>>((const T*)qkv_pre, S0, (const T*)qkv_new, cap, t, B, H, D, rot, P, Tm, rc, rs, am, (T*)out);
  return MAFED_OK;
}
template int attn_decode_launch<float>(const void*, int, const void*, int, int, int, int, int, int, int, int, const float*, const float*,
                                       const int64_t*, void*, hipStream_t, bool);
template int attn_decode_launch<bf16_t>(const void*, int, const void*, int, int, int, int, int, int, int, int, const float*, const float*,
                                        const int64_t*, void*, hipStream_t, bool);

template int attn_ref_fwd_launch<float>(const void*, const AttnShape&, const float*, const float*, const int64_t*, void*, float*, hipStream_t);
template int attn_ref_fwd_launch<bf16_t>(const void*, const AttnShape&, const float*, const float*, const int64_t*, void*, float*, hipStream_t);
template int attn_ref_bwd_launch<float>(const void*, const void*, const void*, const float*, const AttnShape&, const float*, const float*,
                                        const int64_t*, void*, float*, hipStream_t);
template int attn_ref_bwd_launch<bf16_t>(const void*, const void*, const void*, const float*, const AttnShape&, const float*, const float*,
                                         const int64_t*, void*, float*, hipStream_t);

}  // namespace mafed
