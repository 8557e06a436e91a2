// Shared device/host helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

#include "../../include/mafed_hip.h"

namespace mafed {

void set_error(const char* fmt, ...);

#define MAFED_CHECK_ARG(cond, ...)          \
  do {                                      \
    if (!(cond)) {                          \
      mafed::set_error(__VA_ARGS__);        \
      return MAFED_EINVAL;                  \
    }                                       \
  } while (0)

#define MAFED_CHECK_LAUNCH(name)                                             \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      mafed::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return MAFED_ELAUNCH;                                                  \
    }                                                                        \
  } while (0)

// ---- in-library kernel profiler (mafed_prof_*, include/mafed_hip.h) ---------------------------------------------------------
// Every hot kernel goes through launch(): a plain launch normally; while a profile is open the same kernel is launched with
// hipExtLaunchKernelGGL and a start/stop event pair, whose elapsed time is the kernel's own execution time on the GPU (the
// dispatch's begin/end timestamps -- what rocprofv3 --kernel-trace reports), not a stream bracket that would also count the
// time spent queued behind other streams' kernels.  `work` is the launch's algorithmic work in the unit of its roofline:
// flops for the MFMA kernels (GEMM, attention), bytes for the HBM-bound ones.
enum KernelTag {
  K_GEMM_BF16 = 0, K_GEMM_F32, K_GEMM_SKINNY, K_ATTN_FWD, K_ATTN_BWD_DQ, K_ATTN_BWD_DKV, K_ATTN_EXACT, K_LN_FWD, K_LN_BWD, K_LN_BWD_REDUCE,
  K_CE_FWD, K_CE_BWD, K_DISTILL_FWD, K_DISTILL_BWD, K_ADAMW, K_GRADNORM, K_EMBED_FWD, K_EMBED_BWD, K_COLSUM, K_CAST, K_EWC, K_SMALL, K_GEMM_PP, K_TAG_COUNT
};
extern std::atomic<bool> g_prof_on;   // read on every launch (caller thread and autograd's backward thread), written under the profiler's mutex
bool prof_events(int tag, double work, hipEvent_t* e0, hipEvent_t* e1);

template <typename... KArgs, typename... Args>
inline void launch(int tag, double work, void (*kfn)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
  hipEvent_t e0, e1;
  if (g_prof_on.load(std::memory_order_relaxed) && prof_events(tag, work, &e0, &e1)) {
    hipExtLaunchKernelGGL(kfn, grid, block, (uint32_t)lds, st, e0, e1, 0, static_cast<KArgs>(args)...);
    return;
  }
  kfn<<<grid, block, lds, st>>>(static_cast<KArgs>(args)...);
}

typedef uint16_t bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32; keeps NaN a NaN, MI355X_MICROARCH correctness table)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 4 consecutive elements <-> float4 (16-byte fp32 / 8-byte bf16 accesses)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
  uint2 r;
  r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
  r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); sm must hold NT/64 floats; result valid in every thread
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += sm[i];
  return r;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  float r = sm[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, sm[i]);
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Fast erf-GELU for the bf16 MFMA epilogues: Phi(x) - 1/2 = x P(s) and gelu'(x) - 1/2 = x Q(s) (both odd), x clamped to [-R, R],
// s = 2 x^2 / R^2 - 1, P / Q = Chebyshev interpolants in monomial form (tools/gelu_poly_fit.py prints the arrays and their errors:
// |Phi err| <= 2.3e-6, |gelu err| <= 1.8e-5, |gelu' err| <= 1.2e-5 over [-8, 8] in fp32 Horner -- below the bf16 rounding of the
// stored activation / gradient everywhere).  Everything is packed v_pk_{mul,fma}_f32 on element pairs plus one v_med3 per element for the
// clamp: ~60 issue cycles per pair, no transcendental.  The Abramowitz-Stegun form used until round 2 (one v_rcp + one v_exp per element,
// both quarter rate) cost ~140 and made the GELU epilogues VALU-bound: 22.8 us of the 4h up-projection's 99.8 (11 k cycles per tile
// against a 29 k-cycle K loop).  The fp32 parity path keeps erff.
constexpr float GELU_P_R = 4.5f, GELU_Q_R = 5.0f;
constexpr float GELU_P[11] = {1.569049965e-01f, -7.719375885e-02f, 5.469785678e-02f, -4.011058319e-02f, 2.831077579e-02f, -1.901597806e-02f,
                              1.136370917e-02f, -5.258188585e-03f, 2.802886231e-03f, -2.337056659e-03f, 9.459542584e-04f};
constexpr float GELU_Q[13] = {1.421339443e-01f, -7.509349723e-02f, 6.655416207e-02f, -7.222797948e-02f, 8.061209880e-02f, -8.095120240e-02f,
                              7.859707870e-02f, -7.980066804e-02f, 5.597591266e-02f, -1.508289572e-02f, 1.306347472e-02f, -2.578379958e-02f,
                              1.200570532e-02f};
template <int N>
__device__ __forceinline__ f32x2 gelu_odd_poly2(f32x2 x, const float (&c)[N], float R) {   // -> 1/2 + xc * poly(s), xc = clamp(x, -R, R)
  f32x2 xc;
  xc[0] = __builtin_amdgcn_fmed3f(x[0], -R, R); xc[1] = __builtin_amdgcn_fmed3f(x[1], -R, R);
  const f32x2 s = xc * xc * (2.0f / (R * R)) + -1.0f;
  f32x2 r = s * c[N - 1] + c[N - 2];
#pragma unroll
  for (int i = N - 3; i >= 0; --i) r = r * s + c[i];
  return xc * r + 0.5f;
}
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) { return x * gelu_odd_poly2(x, GELU_P, GELU_P_R); }
__device__ __forceinline__ f32x2 gelu_erf_grad_fast2(f32x2 x) { return gelu_odd_poly2(x, GELU_Q, GELU_Q_R); }
__device__ __forceinline__ float gelu_erf_fast(float x) {
  f32x2 v = {x, x};
  return gelu_erf_fast2(v)[0];
}
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  f32x2 v = {x, x};
  return gelu_erf_grad_fast2(v)[0];
}

// x * sigmoid(1.702 x): hidden_act "quick_gelu" of the OpenAI CLIP towers (transformers/activations.py QuickGELUActivation)
__device__ __forceinline__ float quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x)); }

// modality class of a token row (mafed/methods/distillation.py:134-144): 0 = language (valid text), 1 = vision, 2 = none (pad)
__device__ __forceinline__ int modality_class(int64_t row, int S, int P, int T, const int64_t* attention_mask) {
  const int64_t b = row / S;
  const int s = (int)(row - b * S);
  if (s < P) return 1;
  return attention_mask[b * T + (s - P)] != 0 ? 0 : 2;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mafed
