// Shared fused epilogue of the dense contractions: bias -> GELU / GELU' -> residual adds -> beta*C -> store.
#pragma once
#include "common.h"

namespace mafed {

struct GemmEpi {
  const float* bias;   // [N] or null
  int mode;            // MAFED_EPI_*
  void* aux;           // pre-activation [M,N] (ld = ldc), same dtype as C
  const float* res1;   // fp32 [M,N] (ld = ldc) or null
  const float* res2;
  float beta;          // C = v + beta * C_old  (fp32 C only)
  int64_t ldc;
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
  } else if (e.mode == MAFED_EPI_GELU_BWD) {
    const float4 u = load4(reinterpret_cast<const CT*>(e.aux) + off);
    if (FAST) v = make_float4(v.x * gelu_erf_grad_fast(u.x), v.y * gelu_erf_grad_fast(u.y), v.z * gelu_erf_grad_fast(u.z), v.w * gelu_erf_grad_fast(u.w));
    else v = make_float4(v.x * gelu_erf_grad(u.x), v.y * gelu_erf_grad(u.y), v.z * gelu_erf_grad(u.z), v.w * gelu_erf_grad(u.w));
  }
  if (e.res1) {
    const float4 r = load4(e.res1 + off);
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

}  // namespace mafed
