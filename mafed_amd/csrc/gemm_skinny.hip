// Skinny NT product for the decode path: C[M,N] = X[M,K] . W[N,K]^T with M <= 64 rows (one token
// per sample).  The work is streaming W once from HBM; a 128-row tile kernel would launch N/128 = 8..32 blocks for it.
// Here one block owns a 16-column strip of C: its eight waves each take an eighth of K, stream their 16 x K/8 slab of W
// straight from global memory into MFMA operands (16 B per lane, no LDS), multiply it with the matching columns of the
// L2-resident X, fold the eight partial tiles through LDS and run the ordinary fused epilogue (bias, GELU, residuals).
// N/16 blocks x 8 waves = 512 .. 25k waves in flight.  Strips narrower than the MFMA's 16 columns (NS = 8 or 4: the other W rows of the
// operand are zeros that are never loaded) put a block on every CU when N is small: the h-wide projections (N = 1024) ran on 64
// of the 256 CUs with 16-column strips (17.5 -> 14.3 us for the 8 MB of the 4h -> h weight, 7.0 -> 5.5 for the 2 MB of dense).
#include "gemm_epilogue.h"

namespace mafed {

template <int MT, int NW, typename CT, int NS = 16>  // NS: 16 or 4
__global__ __launch_bounds__(NW * 64) void gemm_skinny_nt_kernel(int M, int64_t N, int64_t K, const bf16_t* __restrict__ X, int64_t ldx,
                                                             const bf16_t* __restrict__ W, int64_t ldw, CT* __restrict__ C, GemmEpi epi) {
  __shared__ f32x4 red[NW - 1][MT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * NS;
  const bool win = NS == 16 || i < NS;  // this lane's W row exists
  const int64_t kper = K / NW, kb = wave * kper;
  const bf16_t* wp = W + (n0 + (win ? i : 0)) * ldw + kb + 8 * g;
  const bf16_t* xp[MT];
  bool xin[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + i;
    xin[mt] = row < M;
    xp[mt] = X + (int64_t)(xin[mt] ? row : 0) * ldx + kb + 8 * g;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the storing wave requests its bias / residual operands now: they arrive under the weight stream
  EpiPre4 epre[MT];
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 16 + i;
      epi_fetch4(epi, m < M ? m : 0, n0 + (4 * g < NS ? 4 * g : 0), epre[mt]);
    }
  }
  const bf16x8 zero = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
  int64_t k = 0;
  for (; k + 128 <= kper; k += 128) {  // four k-steps per trip: every load of the trip is in flight before the first MFMA
    bf16x8 wf[4], xf[4][MT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wf[u] = win ? *reinterpret_cast<const bf16x8*>(wp + k + 32 * u) : zero;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][mt] = xin[mt] ? *reinterpret_cast<const bf16x8*>(xp[mt] + k + 32 * u) : zero;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][mt], acc[mt], 0, 0, 0);
  }
  for (; k < kper; k += 32) {
    const bf16x8 wf = win ? *reinterpret_cast<const bf16x8*>(wp + k) : zero;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 xf = xin[mt] ? *reinterpret_cast<const bf16x8*>(xp[mt] + k) : zero;
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mt], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[wave - 1][mt][lane] = acc[mt];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 a = acc[mt];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) a += red[w][mt][lane];
      // lane holds C[m = mt*16 + (lane & 15)][n0 + 4g .. 4g+3]
      const int m = mt * 16 + i;
      if (m < M && 4 * g < NS) {
        if (epi.mode == MAFED_EPI_GELU_BWD) epilogue_store4<CT, true>(epi, C, m, n0 + 4 * g, make_float4(a[0], a[1], a[2], a[3]));
        else epilogue_store4_pre<CT>(epi, C, m, n0 + 4 * g, make_float4(a[0], a[1], a[2], a[3]), epre[mt]);
      }
    }
  }
}

int g_skinny_ns = 0, g_skinny_wide = -1;  // tuning overrides (mafed_gemm_set_variant 500 + ns, 600 + wide)

template <typename CT>
static int skinny_dispatch(int M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, const GemmEpi& epi,
                           hipStream_t st) {
  const int mt = (M + 15) / 16;
  // sixteen waves per strip once a wave's K slice would need more than one trip of loads (K >= 2048: the 4h -> h projection)
  bool wide = K >= 2048 && K % 512 == 0;
  // strip width: the widest that still gives every CU a block
  int ns = (N / 16 >= 128 || N % 4 != 0) ? 16 : 4;  // measured in the decode step: N = 1024 5.5 (NS 4) vs 7.0 us (16); N = 3072 8.7 (NS 8) vs 7.0 (16)
  if (g_skinny_ns) ns = g_skinny_ns;
  if (g_skinny_wide >= 0) wide = g_skinny_wide && K % 512 == 0;
#define GO2(MTV, NSV)                                                                                                                          \
  do {                                                                                                                                         \
    const dim3 grid((unsigned)(N / NSV));                                                                                                      \
    if (wide) gemm_skinny_nt_kernel<MTV, 16, CT, NSV><<<grid, dim3(1024), 0, st>>>(M, N, K, (const bf16_t*)X, ldx, (const bf16_t*)W, ldw, (CT*)C, epi); \
    else gemm_skinny_nt_kernel<MTV, 8, CT, NSV><<<grid, dim3(512), 0, st>>>(M, N, K, (const bf16_t*)X, ldx, (const bf16_t*)W, ldw, (CT*)C, epi);       \
  } while (0)
#define GO(MTV)                  \
  do {                           \
    if (ns == 16) GO2(MTV, 16);  \
    else GO2(MTV, 4);            \
  } while (0)
  switch (mt) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    default: GO(4); break;
  }
#undef GO2
#undef GO
  return MAFED_OK;
}

// C = X . W^T (NT), bf16 operands, M <= 64, N % 16 == 0, K % 256 == 0 (eight waves x whole 32-deep k-steps)
bool gemm_skinny_ok(int transA, int transB, int64_t M, int64_t N, int64_t K, float beta, const void* colsum) {
  return !transA && transB && M >= 1 && M <= 64 && N % 16 == 0 && K % 256 == 0 && beta == 0.f && colsum == nullptr && N / 16 <= 0x7fffffff;
}

int gemm_skinny_launch(int64_t M, int64_t N, int64_t K, const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, mafed_dtype c_dtype,
                       const GemmEpi& epi, hipStream_t st) {
  return c_dtype == MAFED_F32 ? skinny_dispatch<float>((int)M, N, K, X, ldx, W, ldw, C, epi, st)
                              : skinny_dispatch<bf16_t>((int)M, N, K, X, ldx, W, ldw, C, epi, st);
}

}  // namespace mafed
