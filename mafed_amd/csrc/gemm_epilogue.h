// Shared fused epilogue of the dense contractions: bias -> GELU / GELU' -> residual adds -> beta*C -> store.
#pragma once
#include "common.h"

namespace mafed {

struct GemmEpi {
  const float* bias;   // [N] or null
  int mode;            // MAFED_EPI_*
  void* aux;           // pre-activation [M,N] (ld = ldc), same dtype as C
  const float* res1;   // [M,N] (ld = ldc) or null; fp32, or bf16 when res1_bf16 is set
  const float* res2;   // fp32 [M,N] or null
  int res1_bf16;
  float beta;          // C = v + beta * C_old  (fp32 C only)
  int64_t ldc;
  float* colsum;       // [N] or null: colsum[n] += sum_m C[m,n] (the bias gradient of the layer that produced this GEMM's input)
};

// 4 consecutive columns n..n+3 of row m (n % 4 == 0, n + 3 < N guaranteed by the caller)
template <typename CT, bool FAST = false>
__device__ __forceinline__ void epilogue_store4(const GemmEpi& e, CT* __restrict__ C, int64_t m, int64_t n, float4 v) {
  const int64_t off = m * e.ldc + n;
  if (e.bias) {
    const float4 b = load4(e.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (e.mode == MAFED_EPI_GELU) {
    if (e.aux) store4(reinterpret_cast<CT*>(e.aux) + off, v);
    if (FAST) v = make_float4(gelu_erf_fast(v.x), gelu_erf_fast(v.y), gelu_erf_fast(v.z), gelu_erf_fast(v.w));
    else v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
  } else if (e.mode == MAFED_EPI_QUICK_GELU) {
    v = make_float4(quick_gelu(v.x), quick_gelu(v.y), quick_gelu(v.z), quick_gelu(v.w));
  } else if (e.mode == MAFED_EPI_GELU_BWD) {
    const float4 u = load4(reinterpret_cast<const CT*>(e.aux) + off);
    if (FAST) v = make_float4(v.x * gelu_erf_grad_fast(u.x), v.y * gelu_erf_grad_fast(u.y), v.z * gelu_erf_grad_fast(u.z), v.w * gelu_erf_grad_fast(u.w));
    else v = make_float4(v.x * gelu_erf_grad(u.x), v.y * gelu_erf_grad(u.y), v.z * gelu_erf_grad(u.z), v.w * gelu_erf_grad(u.w));
  }
  if (e.res1) {
    const float4 r = e.res1_bf16 ? load4(reinterpret_cast<const bf16_t*>(e.res1) + off) : load4(e.res1 + off);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (e.res2) {
    const float4 r = load4(e.res2 + off);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (e.beta != 0.f) {
    const float4 c = load4(C + off);
    v.x += e.beta * c.x; v.y += e.beta * c.y; v.z += e.beta * c.z; v.w += e.beta * c.w;
  }
  store4(C + off, v);
}

// epilogue_store4 with the bias / residual operands already in registers (epi_fetch4, issued before the product's main loop):
// behind the cross-wave fold of the skinny kernel each of them was a dependent round trip (bias, res1, res2 per 16-row tile: the
// 4h -> h projection of a decode step spent more time there than streaming its 8 MB of weights).  No beta, no GELU' (not on that path).
struct EpiPre4 {
  float4 b, r1, r2;
};
__device__ __forceinline__ void epi_fetch4(const GemmEpi& e, int64_t m, int64_t n, EpiPre4& p) {
  const int64_t off = m * e.ldc + n;
  p.b = e.bias ? load4(e.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  p.r1 = e.res1 ? (e.res1_bf16 ? load4(reinterpret_cast<const bf16_t*>(e.res1) + off) : load4(e.res1 + off)) : make_float4(0.f, 0.f, 0.f, 0.f);
  p.r2 = e.res2 ? load4(e.res2 + off) : make_float4(0.f, 0.f, 0.f, 0.f);
}
template <typename CT>
__device__ __forceinline__ void epilogue_store4_pre(const GemmEpi& e, CT* __restrict__ C, int64_t m, int64_t n, float4 v, const EpiPre4& p) {
  const int64_t off = m * e.ldc + n;
  v.x += p.b.x; v.y += p.b.y; v.z += p.b.z; v.w += p.b.w;
  if (e.mode == MAFED_EPI_GELU) {
    if (e.aux) store4(reinterpret_cast<CT*>(e.aux) + off, v);
    v = make_float4(gelu_erf_fast(v.x), gelu_erf_fast(v.y), gelu_erf_fast(v.z), gelu_erf_fast(v.w));
  } else if (e.mode == MAFED_EPI_QUICK_GELU) {
    v = make_float4(quick_gelu(v.x), quick_gelu(v.y), quick_gelu(v.z), quick_gelu(v.w));
  }
  v.x += p.r1.x; v.y += p.r1.y; v.z += p.r1.z; v.w += p.r1.w;   // same order of additions as epilogue_store4
  v.x += p.r2.x; v.y += p.r2.y; v.z += p.r2.z; v.w += p.r2.w;
  store4(C + off, v);
}

// 8 consecutive columns n..n+7 of row m (n % 8 == 0): the coalesced form used after the accumulators went through LDS
// (16-byte bf16 / 32-byte fp32 stores, 128/256 contiguous bytes per row per 8 lanes).
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = load4(p), b = load4(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const uint4 r = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  store4(p, make_float4(v[0], v[1], v[2], v[3]));
  store4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  r.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  r.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
  r.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = r;
}

template <typename CT>
__device__ __forceinline__ void epilogue_store8(const GemmEpi& e, CT* __restrict__ C, int64_t m, int64_t n, float (&v)[8]) {
  const int64_t off = m * e.ldc + n;
  if (e.bias) {
    float b[8];
    load8(e.bias + n, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if (e.mode == MAFED_EPI_GELU) {
    if (e.aux) store8(reinterpret_cast<CT*>(e.aux) + off, v);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = gelu_erf_fast2((f32x2){v[i], v[i + 1]});
      v[i] = r[0]; v[i + 1] = r[1];
    }
  } else if (e.mode == MAFED_EPI_QUICK_GELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = quick_gelu(v[i]);
  } else if (e.mode == MAFED_EPI_GELU_BWD) {
    float u[8];
    load8(reinterpret_cast<const CT*>(e.aux) + off, u);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = gelu_erf_grad_fast2((f32x2){u[i], u[i + 1]});
      v[i] *= r[0]; v[i + 1] *= r[1];
    }
  }
  if (e.res1) {
    float r[8];
    if (e.res1_bf16) load8(reinterpret_cast<const bf16_t*>(e.res1) + off, r);
    else load8(e.res1 + off, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (e.res2) {
    float r[8];
    load8(e.res2 + off, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (e.beta != 0.f) {
    float c[8];
    load8(C + off, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += e.beta * c[i];
  }
  store8(C + off, v);
}

// Operands the epilogue READS per 8-column group (saved pre-activation of GELU', residuals, the old C of beta != 0), fetched for a
// whole pass of groups BEFORE that pass's stores: inside epilogue_store8 each load sits behind the previous group's store (the
// compiler cannot prove C / aux / residual pointers distinct), so a wave paid one HBM round trip per group -- 12 in a row in
// the GELU' GEMM, whose epilogue alone ran 55 us for 151 MB.  One bf16 slot (aux, or the bf16 residual) and one fp32 slot
// (res2, or the old C) per group; whatever does not fit a slot is loaded in place as before.
struct EpiPre {
  uint4 h;        // 8 bf16
  float4 f0, f1;  // 8 fp32
};
template <typename CT>
__device__ __forceinline__ bool epi_pre_aux(const GemmEpi& e) { return sizeof(CT) == 2 && e.mode == MAFED_EPI_GELU_BWD; }
template <typename CT>
__device__ __forceinline__ bool epi_pre_res1(const GemmEpi& e) { return e.res1 && e.res1_bf16 && !epi_pre_aux<CT>(e); }
template <typename CT>
__device__ __forceinline__ bool epi_pre_cold(const GemmEpi& e) { return sizeof(CT) == 4 && e.beta != 0.f && !e.res2; }

// F32SLOT = false: only the bf16 slot is fetched ahead (tiles whose accumulators leave no room for 8 more registers per group)
template <typename CT, bool F32SLOT = true>
__device__ __forceinline__ void epi_prefetch(const GemmEpi& e, const CT* __restrict__ C, int64_t m, int64_t n, EpiPre& p) {
  const int64_t off = m * e.ldc + n;
  if (epi_pre_aux<CT>(e)) p.h = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(e.aux) + off);
  else if (epi_pre_res1<CT>(e)) p.h = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(e.res1) + off);
  if (!F32SLOT) return;
  if (e.res2) {
    p.f0 = load4(e.res2 + off);
    p.f1 = load4(e.res2 + off + 4);
  } else if (epi_pre_cold<CT>(e)) {
    p.f0 = load4(reinterpret_cast<const float*>(C) + off);
    p.f1 = load4(reinterpret_cast<const float*>(C) + off + 4);
  }
}

__device__ __forceinline__ void unpack8(const uint4& r, float (&v)[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

// epilogue_store8 with the read operands taken from `p` (same arithmetic, same order of additions)
// `bv`: the 8 bias values of columns n .. n+7, loaded ONCE per wave by the caller (epi_bias8): the column group of a lane is the same
// for every row group and pass of the tile, but a reload after each store (the compiler cannot hoist it past them) put an L2 round
// trip on the dependency chain of every group.
__device__ __forceinline__ void epi_bias8(const GemmEpi& e, int64_t n, float (&bv)[8]) {
  if (e.bias) load8(e.bias + n, bv);
  else {
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = 0.f;
  }
}
template <typename CT, bool F32SLOT = true>
__device__ __forceinline__ void epilogue_store8_pre(const GemmEpi& e, CT* __restrict__ C, int64_t m, int64_t n, float (&v)[8], const EpiPre& p,
                                                    const float (&bv)[8]) {
  const int64_t off = m * e.ldc + n;
  if (e.bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += bv[i];
  }
  if (e.mode == MAFED_EPI_GELU) {
    if (e.aux) store8(reinterpret_cast<CT*>(e.aux) + off, v);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = gelu_erf_fast2((f32x2){v[i], v[i + 1]});
      v[i] = r[0]; v[i + 1] = r[1];
    }
  } else if (e.mode == MAFED_EPI_QUICK_GELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = quick_gelu(v[i]);
  } else if (e.mode == MAFED_EPI_GELU_BWD) {
    float u[8];
    if (epi_pre_aux<CT>(e)) unpack8(p.h, u);
    else load8(reinterpret_cast<const CT*>(e.aux) + off, u);
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const f32x2 r = gelu_erf_grad_fast2((f32x2){u[i], u[i + 1]});
      v[i] *= r[0]; v[i + 1] *= r[1];
    }
  }
  if (e.res1) {
    float r[8];
    if (epi_pre_res1<CT>(e)) unpack8(p.h, r);
    else if (e.res1_bf16) load8(reinterpret_cast<const bf16_t*>(e.res1) + off, r);
    else load8(e.res1 + off, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (e.res2) {
    float r[8] = {p.f0.x, p.f0.y, p.f0.z, p.f0.w, p.f1.x, p.f1.y, p.f1.z, p.f1.w};
    if (!F32SLOT) load8(e.res2 + off, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += r[i];
  }
  if (e.beta != 0.f) {
    float c[8];
    if (F32SLOT && epi_pre_cold<CT>(e)) {
      c[0] = p.f0.x; c[1] = p.f0.y; c[2] = p.f0.z; c[3] = p.f0.w; c[4] = p.f1.x; c[5] = p.f1.y; c[6] = p.f1.z; c[7] = p.f1.w;
    } else {
      load8(C + off, c);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += e.beta * c[i];
  }
  store8(C + off, v);
}

// Fused column sums of the stored tile: cs[i] holds this lane's partial for column 8*(lane % JL)+i over the rows it
// stored; lanes that share lane % JL are folded, then JL lanes issue 8 atomics each (one per column per wave per tile).
template <int JL>
__device__ __forceinline__ void colsum_flush(float (&cs)[8], float* __restrict__ colsum, int64_t ncol0, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = cs[i];
#pragma unroll
    for (int o = JL; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    cs[i] = v;
  }
  if (lane < JL) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(colsum + ncol0 + 8 * lane + i, cs[i]);
  }
}

// Block-level form: the waves of a block first fold their sums in LDS (`lds_sum`: TN floats at the start of the block's LDS), then the block issues
// TN / 64 full-width atomic instructions (256 contiguous bytes each: the full-rate shape of MI355X_MICROARCH "Global float atomics")
// instead of 8 (or 4) eight-lane instructions per wave -- the cost of a float atomic is per INSTRUCTION (~50 ns per CU), and the
// per-wave form cost the GELU' GEMM of the step 18 us of its 132.  `wcol0` = first tile column of this wave.  Contains barriers:
// every thread of the block calls it.
template <int JL, int TN>
__device__ __forceinline__ void colsum_flush_block(float (&cs)[8], float* __restrict__ lds_sum, float* __restrict__ colsum, int64_t n0, int wcol0,
                                                   int lane, int tid) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v = cs[i];
#pragma unroll
    for (int o = JL; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    cs[i] = v;
  }
  __syncthreads();  // every wave is done with its staging region: `lds_sum` re-uses the start of that memory
  if (tid < TN) lds_sum[tid] = 0.f;
  __syncthreads();
  if (lane < JL) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(lds_sum + wcol0 + 8 * lane + i, cs[i]);  // ds_add_f32
  }
  __syncthreads();
  if (tid < TN) atomicAdd(colsum + n0 + tid, lds_sum[tid]);
}

// decode-path product (gemm_skinny.hip)
bool gemm_skinny_ok(int transA, int transB, int64_t M, int64_t N, int64_t K, float beta, const void* colsum);
int gemm_skinny_launch(int64_t M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, mafed_dtype c_dtype,
                       const GemmEpi& epi, hipStream_t st);

}  // namespace mafed
