// bf16 MFMA GEMM, persistent form with deferred stores (see the header comment of the kernel).  Split from gemm.hip so that the two
// files build in parallel; the LDS images and DMA addressing are shared through gemm_tiles.h.
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_tiles.h"

namespace mafed {

// ------------------------------------------------------------------------------------------------------------
// Persistent LDS-DMA kernel with DEFERRED stores (bf16 outputs, >= 2 tiles per block).
// The kernel above is store-issue-bound in its epilogue: every block of the chip finishes its K loop at the same time, then all of
// them push 12-15 GB/s per CU of output through the store path while the matrix pipe idles (fc1 + GELU: 3 rounds x 13 us of a
// 114 us kernel; total epilogue time = output bytes / ~3.6 TB/s whatever the tiling).  Here a block walks several tiles; when a
// tile's K loop ends its accumulators are rounded to bf16 (+ bias) into `pend` registers (48 VGPRs for a 96 x 64 wave tile) and the
// block goes straight on to the next tile -- whose first K-tile was already requested by DMA during the previous tile's last K step,
// so the DMA pipeline never drains between tiles.  The pending tile is then stored piece by piece (one 16-row group = NT 8-byte
// stores per lane, completing whole 128-byte lines) during the first MT iterations of the next tile's K loop, straight from the
// accumulator layout: no LDS staging, no epilogue phase; GELU / GELU' are evaluated at store time on the VALU, beside the MFMAs.
//   GELU:     pend = bf16(acc + bias) = the saved pre-activation (aux); C = gelu(pend)   (what bf16 autocast computes: the
//             Linear's output is rounded to bf16 before the activation)
//   GELU_BWD: pend = bf16(acc); C = pend * gelu'(aux), aux rows fetched one slot ahead
// Slots are compile-time (the first MT iterations are peeled) so that `pend` is indexed statically (cdna_hip_programming rule 20).
// ------------------------------------------------------------------------------------------------------------
template <int... I, typename F>
__device__ __forceinline__ __attribute__((always_inline)) void static_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

__device__ __forceinline__ float4 unpack4(uint2 p) {
  return make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16), __uint_as_float(p.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 pack4(float4 v) {
  uint2 r;
  r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
  r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
  return r;
}
// sum over the 16 lanes of a DPP row; every lane ends with the row total
__device__ __forceinline__ float gemm_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, bool BWD, bool CSUM>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_bf16_persist_kernel(int64_t M, int64_t N, int64_t K, const bf16_t* __restrict__ A,
                                                                             int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                             bf16_t* __restrict__ C, GemmEpi epi, int tiles_n, int ntiles) {
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16, NW = WM * WN;
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = TM / 8, NB = TN / 8;
  constexpr int A_PER_WAVE = (NA + NW - 1) / NW, B_PER_WAVE = (NB + NW - 1) / NW;
  static_assert(!A_KS, "row-major activations only");
  static_assert(!B_KS || TN == 64 || TN % 128 == 0, "the [k][row] image needs 64 or 128-row multiples");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int GROUP_M = tiles_n >> 20;
  const int tiles_n_ = tiles_n & 0xfffff;
  const int tiles_m = ntiles / tiles_n_;
  const int mode = epi.mode;

  // virtual block id -> tile: the XCD-aware remap and L2-patch order of gemm_bf16_glds_kernel (v and v + gridDim.x share an XCD)
  auto tile_of = [&](int v, int64_t& m0, int64_t& n0) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    const int gsz = GROUP_M * tiles_n_;
    const int group = bid / gsz, first_m = group * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int within = bid - group * gsz;
    m0 = (int64_t)(first_m + within % gm) * TM;
    n0 = (int64_t)(within / gm) * TN;
  };

  // per-lane DMA sources as 32-bit ELEMENT offsets from A / B (the launcher checks that both operands span < 2^31 elements):
  // half the address registers of 64-bit pointers, and the uniform base stays in SGPRs
  int asrc[A_PER_WAVE];
  int bsrc[B_PER_WAVE];
  auto set_sources = [&](int64_t m0, int64_t n0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) asrc[i] = (int)glds_src_off<A_KS, TM>(wave + i * NW < NA ? wave + i * NW : 0, lane, lda, m0);
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) bsrc[i] = (int)glds_src_off<B_KS, TN>(wave + i * NW < NB ? wave + i * NW : 0, lane, ldb, n0);
  };
  const int a_step = A_KS ? 64 * (int)lda : 64, b_step = B_KS ? 64 * (int)ldb : 64;
  auto issue = [&](int stage, int kt) __attribute__((always_inline)) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const bf16_t* Ak = A + (int64_t)kt * a_step;  // uniform: the K advance rides on the scalar base
    const bf16_t* Bk = B + (int64_t)kt * b_step;
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i)
      if (NA % NW == 0 || wave + i * NW < NA)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(Ak + asrc[i]), (lds_void_ptr)(sa + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i)
      if (NB % NW == 0 || wave + i * NW < NB)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(Bk + bsrc[i]), (lds_void_ptr)(sb + (wave + i * NW) * 1024), 16, 0, 0);
  };

  f32x4 acc[NT][MT];
  uint2 pend[MT * NT];       // the finished tile, bf16, accumulator layout: piece mt * NT + nt = rows (lane & 15) of row group mt, 4 columns
  uint2 uraw[BWD ? NT : 1];  // GELU': saved pre-activations of the next row group to store (fetched one slot ahead)
  float cs[CSUM ? NT * 4 : 1];  // fused column sums of the stored values (this lane's row, 4 columns per nt)
  int64_t pm0 = 0, pn0 = 0;  // origin of the pending tile
  bool have_pend = false;
  if constexpr (CSUM) {
#pragma unroll
    for (int i = 0; i < NT * 4; ++i) cs[i] = 0.f;
  }

  auto aux_row = [&](int mt) __attribute__((always_inline)) -> const bf16_t* {
    return reinterpret_cast<const bf16_t*>(epi.aux) + (pm0 + (wm * MT + mt) * 16 + (lane & 15)) * epi.ldc + pn0 + wn * NT * 16 + 4 * g;
  };
  auto fetch_u = [&](int mt) __attribute__((always_inline)) {  // GELU' only
    if constexpr (BWD) {
      const bf16_t* up = aux_row(mt);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) uraw[nt] = *reinterpret_cast<const uint2*>(up + nt * 16);
    }
  };
  // store row group MTI of the pending tile (NT 8-byte stores per lane; the NT stores of the 4 lanes that share a row fill
  // NT * 32 contiguous bytes -- whole 128-byte lines at NT = 4)
  auto store_group = [&](auto mti_tag) __attribute__((always_inline)) {
    constexpr int MTI = decltype(mti_tag)::value;
    const int64_t off0 = (pm0 + (wm * MT + MTI) * 16 + (lane & 15)) * epi.ldc + pn0 + wn * NT * 16 + 4 * g;
    uint2 u_now[BWD ? NT : 1];
    if constexpr (BWD) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) u_now[nt] = uraw[nt];
      if (MTI + 1 < MT) fetch_u(MTI + 1);  // lands before the next slot: a barrier (vmcnt(0)) lies between
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      uint2 pv = pend[MTI * NT + nt];
      float4 v = unpack4(pv);
      if (!BWD && mode == MAFED_EPI_GELU) {
        if (epi.aux) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(epi.aux) + off0 + nt * 16) = pv;
        const f32x2 a = gelu_erf_fast2((f32x2){v.x, v.y}), b = gelu_erf_fast2((f32x2){v.z, v.w});
        v = make_float4(a[0], a[1], b[0], b[1]);
        pv = pack4(v);
      } else if (BWD) {
        const float4 u = unpack4(u_now[BWD ? nt : 0]);
        const f32x2 a = gelu_erf_grad_fast2((f32x2){u.x, u.y}), b = gelu_erf_grad_fast2((f32x2){u.z, u.w});
        v = make_float4(v.x * a[0], v.y * a[1], v.z * b[0], v.w * b[1]);
        pv = pack4(v);
      } else if (!BWD && mode == MAFED_EPI_QUICK_GELU) {
        v = make_float4(quick_gelu(v.x), quick_gelu(v.y), quick_gelu(v.z), quick_gelu(v.w));
        pv = pack4(v);
      }
      *reinterpret_cast<uint2*>(C + off0 + nt * 16) = pv;
      if constexpr (CSUM) {
        cs[nt * 4 + 0] += v.x; cs[nt * 4 + 1] += v.y; cs[nt * 4 + 2] += v.z; cs[nt * 4 + 3] += v.w;
      }
    }
    if constexpr (CSUM && MTI == MT - 1) {  // the tile is out: fold the 16 rows of each column group and add to the bias gradient
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        const float t = gemm_row16_sum(cs[i]);
        if ((lane & 15) == 0) atomicAdd(epi.colsum + pn0 + (wn * NT + (i >> 2)) * 16 + 4 * g + (i & 3), t);
        cs[i] = 0.f;
      }
    }
  };

  // The bias enters through the accumulators' initial value (acc = bias + sum, one rounding at the end): the bias row of the tile
  // about to start is fetched into acc[.][0] -- free registers at that point -- and spread over the row groups behind the first
  // barrier of the tile, whose vmcnt(0) covers the load.
  auto fetch_bias = [&](int64_t n0_) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (epi.bias) {
        const float4 b = load4(epi.bias + n0_ + (wn * NT + nt) * 16 + 4 * g);
        acc[nt][0] = (f32x4){b.x, b.y, b.z, b.w};
      } else {
        acc[nt][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  const int nkt = (int)(K / BK);
  int v = blockIdx.x;
  int64_t m0, n0;
  tile_of(v, m0, n0);
  set_sources(m0, n0);
  issue(0, 0);
  fetch_bias(n0);
  int sbase = 0;  // LDS stage of K-tile 0 of the current tile
  for (;;) {
    const int vnext = v + (int)gridDim.x;
    const bool has_next = vnext < ntiles;
    int64_t cm0 = m0, cn0 = n0;
    // one K step; SLOT >= 0: the iteration also stores row group SLOT of the pending tile
    auto kstep = [&](int kt, auto slot_tag) __attribute__((always_inline)) {
      constexpr int SLOT = decltype(slot_tag)::value;
      __syncthreads();  // (vmcnt(0) first) K-tile kt has landed for every wave; every wave is done with K-tile kt-1 and its stores
      if (kt + 1 < nkt) {
        issue((sbase + kt + 1) & 1, kt + 1);
      } else if (has_next) {  // the next tile's first K-tile goes into the stage K-tile nkt-2 just left: the pipeline never drains
        tile_of(vnext, m0, n0);
        set_sources(m0, n0);
        issue((sbase + nkt) & 1, 0);
      }
      if constexpr (SLOT == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 1; mt < MT; ++mt) acc[nt][mt] = acc[nt][0];
      }
      if constexpr (SLOT >= 0) {
        if (have_pend) store_group(slot_tag);
      }
      const char* sa = smem + ((sbase + kt) & 1) * STAGE;
      const char* sb = sa + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[MT], fb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = glds_read_frag<B_KS, TN>(sb, wn * NT + t, ks, lane);
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = glds_read_frag<A_KS, TM>(sa, wm * MT + t, ks, lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
      }
    };
    static_for(std::make_integer_sequence<int, MT>{}, [&](auto s) __attribute__((always_inline)) { kstep(decltype(s)::value, s); });
    for (int kt = MT; kt < nkt; ++kt) kstep(kt, std::integral_constant<int, -1>{});
    // tile done: accumulators -> bf16 -> pend; from here on the next tile's loop carries its stores
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        pend[mt * NT + nt] = pack4(make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]));
    pm0 = cm0;
    pn0 = cn0;
    have_pend = true;
    if constexpr (BWD) fetch_u(0);
    sbase = (sbase + nkt) & 1;
    if (!has_next) break;
    v = vnext;
    fetch_bias(n0);  // (n0 already names the next tile: set during the last K step)
  }
  // last tile of this block: nothing left to hide the stores under
  static_for(std::make_integer_sequence<int, MT>{}, [&](auto s) __attribute__((always_inline)) { store_group(s); });
}

template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, bool BWD, bool CSUM>
static int launch_bf16_persist(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                               const GemmEpi& epi, int grid, int group_m, hipStream_t st) {
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16, LDS = 2 * (TM + TN) * 128;
  const int64_t tm = M / TM, tn = N / TN, ntiles = tm * tn;
  auto kfn = gemm_bf16_persist_kernel<WM, WN, MT, NT, A_KS, B_KS, BWD, CSUM>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  if (grid > ntiles) grid = (int)ntiles;  // every block owns at least one tile
  launch(K_GEMM_BF16, 2.0 * M * N * K, kfn, dim3((unsigned)grid), dim3(WM * WN * 64), LDS, st, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
         (bf16_t*)C, epi, (int)tn | (group_m << 20), (int)ntiles);
  return MAFED_OK;
}

// cfg 11 = 144 x 128 tiles (4 waves of 144 x 32), cfg 17 = 192 x 128 tiles (4 waves of 96 x 64); A row-major ([M][K]).
// Instantiated for the forward products (plain / bias / GELU + saved pre-activation / quick-GELU) and for the GELU' product of the
// backward with and without the fused column sums (what the step uses: dX of dense_4h_to_h feeding dense_h_to_4h.bias).
bool gemm_persist_ok(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, bool b_ks, const GemmEpi& epi) {
  const int64_t a_span = M * lda, b_span = b_ks ? K * ldb : N * ldb;
  if (a_span >= (1ll << 31) || b_span >= (1ll << 31)) return false;  // 32-bit element offsets inside the kernel
  if (epi.colsum && epi.mode != MAFED_EPI_GELU_BWD) return false;
  return true;
}

int gemm_persist_launch(int cfg, bool b_ks, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                        const GemmEpi& epi, int grid, int group_m, hipStream_t st) {
  const bool bwd = epi.mode == MAFED_EPI_GELU_BWD, cs = epi.colsum != nullptr;
#define MAFED_PERSIST(WM, WN, MT, NT)                                                                                              \
  do {                                                                                                                           \
    if (bwd && cs) return b_ks ? launch_bf16_persist<WM, WN, MT, NT, false, true, true, true>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st)   \
                               : launch_bf16_persist<WM, WN, MT, NT, false, false, true, true>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st); \
    if (bwd) return b_ks ? launch_bf16_persist<WM, WN, MT, NT, false, true, true, false>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st)        \
                         : launch_bf16_persist<WM, WN, MT, NT, false, false, true, false>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st);      \
    return b_ks ? launch_bf16_persist<WM, WN, MT, NT, false, true, false, false>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st)                \
                : launch_bf16_persist<WM, WN, MT, NT, false, false, false, false>(M, N, K, A, lda, B, ldb, C, epi, grid, group_m, st);              \
  } while (0)
  if (cfg == 11) MAFED_PERSIST(1, 4, 9, 2);
  MAFED_PERSIST(2, 2, 6, 4);
#undef MAFED_PERSIST
}

}  // namespace mafed
