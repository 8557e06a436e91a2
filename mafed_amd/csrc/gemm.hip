// Dense contractions C = op(A).op(B) with fused epilogue.
//
//  bf16 path (the product path): v_mfma_f32_16x16x32_bf16, 128x128x64 block tile, 4 waves (2x2), each wave a 64x64
//  sub-tile = 4x4 accumulators of 16x16.  Operands are staged global -> registers -> LDS (16-byte accesses,
//  register prefetch of tile t+1 issued before the MFMAs of tile t, written after them: one barrier per K-tile,
//  two LDS stages).  An operand whose reduction index is contiguous in memory ("KC": A of X.W^T, both of nothing
//  else) is kept as [row][k] and read with ds_read_b128; an operand whose reduction index is the memory row
//  ("KS": W in dX = dY.W, and BOTH operands of dW = dY^T.X) is kept as [k][row] and read with the gfx950 transposing
//  read ds_read_b64_tr_b16, so no transposed copy of an activation or a weight is ever materialised in HBM.
//  Both images are XOR-swizzled so that the fragment reads are bank-conflict free (see lds_off_*).
//  The accumulator is computed transposed (mfma(Bfrag, Afrag)) so that each lane owns 4 CONSECUTIVE columns of one
//  output row: 8/16-byte epilogue stores, float4 bias/residual loads.
//
//  fp32 path (parity mode, 1e-3 gate of the north star): plain FMA tile kernel, exact fp32 products.
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_tiles.h"
#include "gemm_pp.h"

#ifndef MAFED_GEMM_SPREAD_DMA
#define MAFED_GEMM_SPREAD_DMA 0  // measured twice (also with a hand-ordered, fence-pinned row schedule): a DMA piece between MFMA rows is 1.1-1.6x SLOWER than the burst (fc1 114 -> 177 us)
#endif

namespace mafed {

// ------------------------------------------------------------------------------------------------------------
// fp32 parity kernel: 64x64 tile, BK = 16, 256 threads, 4x4 outputs per thread
// ------------------------------------------------------------------------------------------------------------
template <typename CT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int transA, int transB, int64_t M, int64_t N, int64_t K,
                                                       const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                       int64_t ldb, CT* __restrict__ C, GemmEpi epi) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int64_t k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (!transA) { k = tid & 15; m = (tid >> 4) + 16 * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
      const int64_t gm = m0 + m, gk = k0 + k;
      float v = 0.f;
      if (gm < M && gk < K) v = transA ? A[gk * lda + gm] : A[gm * lda + gk];
      As[k][m] = v;
      int n, kb;
      if (transB) { kb = tid & 15; n = (tid >> 4) + 16 * i; } else { n = tid & 63; kb = (tid >> 6) + 4 * i; }
      const int64_t gn = n0 + n, gkb = k0 + kb;
      float w = 0.f;
      if (gn < N && gkb < K) w = transB ? B[gn * ldb + gkb] : B[gkb * ldb + gn];
      Bs[kb][n] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i, n = n0 + tx * 4;
    if (m < M && n < N) epilogue_store4<CT>(epi, C, m, n, make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]));
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * 64 * 2;              // one operand tile, either image: 16 KiB
constexpr int GEMM_LDS_BYTES = 2 * 2 * TILE_BYTES;    // 2 stages x (A + B) = 64 KiB -> 2 blocks / CU

// global -> registers for one 128 x 64 operand tile (4 x 16 B per thread), zero-filled out of range
template <bool KS>
__device__ __forceinline__ void gemm_load_tile(const bf16_t* __restrict__ X, int64_t ld, int64_t r0, int64_t R, int64_t k0, int64_t K,
                                               int tid, uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!KS) {
      const int row = c >> 3, kc = c & 7;
      const int64_t gr = r0 + row, gk = k0 + kc * 8;
      if (gr < R && gk < K) v = *reinterpret_cast<const uint4*>(X + gr * ld + gk);  // K % 8 == 0
    } else {
      const int k = c >> 4, rc = c & 15;
      const int64_t gk = k0 + k, gr = r0 + rc * 8;
      if (gk < K && gr < R) v = *reinterpret_cast<const uint4*>(X + gk * ld + gr);  // R % 8 == 0
    }
    reg[i] = v;
  }
}

template <bool KS>
__device__ __forceinline__ void gemm_store_tile(char* __restrict__ img, int tid, const uint4 (&reg)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    int off;
    if (!KS) off = lds_off_kc(c >> 3, c & 7);
    else off = lds_off_ks(c >> 4, (c & 15) >> 1, ((c & 15) & 1) << 4);
    *reinterpret_cast<uint4*>(img + off) = reg[i];
  }
}

// fragment of a 16-row tile for k-step ks (32 wide): lane l gets X(row = rt*16 + (l&15), k = ks*32 + 8*(l>>4) + j), j = 0..7
template <bool KS>
__device__ __forceinline__ bf16x8 gemm_read_frag(const char* __restrict__ img, int rt, int ks, int lane) {
  if (!KS) {
    const int row = rt * 16 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + lds_off_kc(row, ks * 4 + (lane >> 4)));
  } else {
    // two transposing reads: k rows 8g+q and 8g+4+q (q = (lane&15)>>2 supplies the row address), 4 columns (lane&3)*4..+3
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int ka = ks * 32 + 8 * g + q, kb = ka + 4;
    typedef __attribute__((address_space(3))) bf16x4* lptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(img + lds_off_ks(ka, rt, p * 8)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(img + lds_off_ks(kb, rt, p * 8)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

template <bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(int64_t M, int64_t N, int64_t K, const bf16_t* __restrict__ A, int64_t lda,
                                                           const bf16_t* __restrict__ B, int64_t ldb, CT* __restrict__ C, GemmEpi epi,
                                                           int tiles_n, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware remap (bijective form, cdna_hip_programming T1): blocks that share an XCD (same bid % 8) take a
  // contiguous run of tiles, n fastest, so an XCD's L2 keeps the A row panel and the whole of W.
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int64_t m0 = (int64_t)(bid / tiles_n) * BM, n0 = (int64_t)(bid % tiles_n) * BN;

  f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nkt = (int)((K + BK - 1) / BK);
  gemm_load_tile<A_KS>(A, lda, m0, M, 0, K, tid, ra);
  gemm_load_tile<B_KS>(B, ldb, n0, N, 0, K, tid, rb);
  gemm_store_tile<A_KS>(smem, tid, ra);
  gemm_store_tile<B_KS>(smem + TILE_BYTES, tid, rb);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const char* sa = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* sb = sa + TILE_BYTES;
    const bool more = kt + 1 < nkt;
    if (more) {
      gemm_load_tile<A_KS>(A, lda, m0, M, (int64_t)(kt + 1) * BK, K, tid, ra);
      gemm_load_tile<B_KS>(B, ldb, n0, N, (int64_t)(kt + 1) * BK, K, tid, rb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = gemm_read_frag<A_KS>(sa, wm * 4 + t, ks, lane);
        fb[t] = gemm_read_frag<B_KS>(sb, wn * 4 + t, ks, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
    }
    if (more) {
      char* da = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      gemm_store_tile<A_KS>(da, tid, ra);
      gemm_store_tile<B_KS>(da + TILE_BYTES, tid, rb);
    }
    __syncthreads();
  }
  // epilogue: lane owns row m = .. + (lane&15), columns n = .. + 4*(lane>>4) + {0..3} of every 16x16 tile
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int64_t m = m0 + wm * 64 + mt * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int64_t n = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
      if (n < N) epilogue_store4<CT, true>(epi, C, m, n, make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel, LDS-DMA staging (global_load_lds_dwordx4): the fast path for tile-aligned shapes.
// Block tile (WM*64) x (WN*64) x 64, WM*WN waves of 64x64.  Operand tiles go global -> LDS without touching VGPRs
// (ds_write_b128 staging tops out near 79 B/clk/CU and was the v1 bottleneck); the LDS images and their swizzles
// are those of the register-staged kernel -- the DMA writes LDS linearly, so the XOR is applied to each lane's
// SOURCE address (cdna_hip_programming rule 21).  Two LDS stages, one barrier per K-tile: the DMA of tile t+1 is
// issued right after the barrier that publishes tile t and flies under tile t's MFMAs.
// ------------------------------------------------------------------------------------------------------------
#ifdef MAFED_GEMM_TRACE
// Tuning builds only (tools/gemm_trace.py): per-block phase timestamps {hw_id, xcc_id, start, first tile landed, loop end,
// block end} on the 100 MHz s_memrealtime clock, to see how the blocks that share a CU line up in time.
__device__ unsigned long long* g_gemm_trace = nullptr;
extern "C" int mafed_gemm_set_trace(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define MAFED_TRACE_MARK(slot) \
  do { if (g_gemm_trace && threadIdx.x == 0 && blockIdx.y == 0) g_gemm_trace[(int64_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MAFED_TRACE_MARK(slot) do { } while (0)
#endif

// ABL (tuning only): 0 = real kernel; 1 = no DMA inside the loop; 2 = no MFMAs (fragments kept live); 3 = no fragment reads
// KG (1 or 2): K groups inside the block.  With KG = 2 a second set of WM x WN waves works on the other half of the
// block's K range through its own pair of LDS stages and hands its accumulators over through LDS at the end: two waves
// per SIMD on GEMMs whose output has only as many tiles as the chip has CUs (dW += dY^T.X), without split-K atomics.
template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, typename CT, int ABL = 0, int KG = 1>
__global__ __launch_bounds__(WM * WN * KG * 64) void gemm_bf16_glds_kernel(int64_t M, int64_t N, int64_t K, const bf16_t* __restrict__ A,
                                                                       int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                                       CT* __restrict__ C, GemmEpi epi, int tiles_n, int nwg) {
  // block tile TM x TN x 64; WM x WN waves, each owning (MT*16) x (NT*16) outputs = MT x NT accumulators of 16x16.
  // TM need not be a power of two: 144-row tiles make the tile count of every M = 9216 GEMM of the 410M / B32 step a
  // multiple of the 512 resident blocks (9216 = 64 x 144), which 128-row tiles miss by up to 44 % (576 tiles = 1.125 rounds).
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16, NW = WM * WN;
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NA = TM / 8, NB = TN / 8;                               // 1 KiB DMA instructions per operand per K-tile
  constexpr int A_PER_WAVE = (NA + NW - 1) / NW, B_PER_WAVE = (NB + NW - 1) / NW;
  static_assert(TM % 8 == 0 && TN % 8 == 0, "tile must be a multiple of the 8-row DMA piece");
  static_assert(!A_KS || TM == 64 || TM % 128 == 0, "the [k][row] image needs 64 or 128-row multiples");
  static_assert(!B_KS || TN == 64 || TN % 128 == 0, "the [k][row] image needs 64 or 128-row multiples");
  static_assert(KG == 1 || (KG == 2 && MT == 4 && NT == 4), "K groups: 64 x 64 wave tiles only");
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 1 ? 0 : wave_all / NW, wave = KG == 1 ? wave_all : wave_all % NW;
  char* const smem = smem_all + kg * (2 * STAGE);   // this K group's two operand stages
  const int wm = wave / WN, wn = wave % WN;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  // grouped tile order inside the XCD's run: GROUP_M row-tiles x all column tiles, walked column-major, so that the ~64
  // blocks resident on one XCD form an ~8 x 8 patch (8 A panels + 8 B panels per K-step through its 4 MiB L2) instead of
  // a 2 x 32 strip (2 + 32 panels): FETCH_SIZE showed the weight operand re-read ~24x per GEMM with the strip order
  int tile_m, tile_n;
  {
    const int GROUP_M = tiles_n >> 20;
    const int tiles_n_ = tiles_n & 0xfffff;
    const int tiles_m = nwg / tiles_n_, gsz = GROUP_M * tiles_n_;
    const int group = bid / gsz, first_m = group * GROUP_M;
    const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int within = bid - group * gsz;
    tile_m = first_m + within % gm;
    tile_n = within / gm;
  }
  const int64_t m0 = (int64_t)tile_m * TM, n0 = (int64_t)tile_n * TN;
#ifdef MAFED_GEMM_TRACE
  if (g_gemm_trace && threadIdx.x == 0 && blockIdx.y == 0) {
    g_gemm_trace[(int64_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
    g_gemm_trace[(int64_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
  }
  MAFED_TRACE_MARK(2);
#endif

  // per-lane DMA sources (piece j = wave + i * NW); k-tile advance is +64 elements (KC) or +64 rows (KS)
  const bf16_t* asrc[A_PER_WAVE];
  const bf16_t* bsrc[B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) asrc[i] = A + glds_src_off<A_KS, TM>(wave + i * NW < NA ? wave + i * NW : 0, lane, lda, m0);
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) bsrc[i] = B + glds_src_off<B_KS, TN>(wave + i * NW < NB ? wave + i * NW : 0, lane, ldb, n0);
  const int64_t a_step = A_KS ? 64 * lda : 64, b_step = B_KS ? 64 * ldb : 64;

  auto issue = [&](int stage, int kt) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i)
      if (NA % NW == 0 || wave + i * NW < NA)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(asrc[i] + kt * a_step), (lds_void_ptr)(sa + (wave + i * NW) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i)
      if (NB % NW == 0 || wave + i * NW < NB)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(bsrc[i] + kt * b_step), (lds_void_ptr)(sb + (wave + i * NW) * 1024), 16, 0, 0);
  };
  // one DMA piece (compile-time index p over this wave's A then B pieces): lets the main loop spread the pieces between
  // MFMA groups instead of issuing them as one burst (a piece costs ~60 issue cycles among MFMAs, 100-185 in a burst)
  auto issue_piece = [&](int stage, int kt, int p) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    if (p < A_PER_WAVE) {
      if (NA % NW == 0 || wave + p * NW < NA)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(asrc[p] + kt * a_step), (lds_void_ptr)(sa + (wave + p * NW) * 1024), 16, 0, 0);
    } else {
      const int q = p - A_PER_WAVE;
      if (NB % NW == 0 || wave + q * NW < NB)
        __builtin_amdgcn_global_load_lds((glb_void_ptr)(bsrc[q] + kt * b_step), (lds_void_ptr)(sb + (wave + q * NW) * 1024), 16, 0, 0);
    }
  };
  constexpr int NPIECE = A_PER_WAVE + B_PER_WAVE;
  constexpr bool SPREAD = MAFED_GEMM_SPREAD_DMA && ABL == 0 && NPIECE <= 2 * MT;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K (gridDim.y > 1, accumulate-only fp32 outputs): this block owns K-tiles [kt0, kt0 + nkt)
  const int nkt_all = (int)(K / BK);
  int kt0 = (int)((int64_t)nkt_all * blockIdx.y / gridDim.y);
  int nkt = (int)((int64_t)nkt_all * (blockIdx.y + 1) / gridDim.y) - kt0;
  if (KG > 1) {  // the dispatcher only picks KG = 2 when every block's range halves evenly (same barrier count in both groups)
    nkt /= KG;
    kt0 += kg * nkt;
  }
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) asrc[i] += kt0 * a_step;
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) bsrc[i] += kt0 * b_step;
  if (ABL != 7) issue(0, 0);
  for (int kt = 0; kt < (ABL == 7 ? 0 : nkt); ++kt) {
    if (ABL != 6) __syncthreads();  // (hipcc drains vmcnt(0) first) tile kt has landed for every wave; everyone is done with tile kt-1
    if (kt == 0) MAFED_TRACE_MARK(3);
    const bool more = kt + 1 < nkt;
    if (!SPREAD && ABL != 1 && ABL != 5 && ABL != 6 && more) issue((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT], fb[NT];
      if (ABL == 3 || ABL == 5 || ABL == 6) {
#pragma unroll
        for (int t = 0; t < NT; ++t) { fb[t] = __builtin_bit_cast(bf16x8, make_uint4(kt, ks, t, lane)); asm volatile("" : "+v"(fb[t])); }
#pragma unroll
        for (int t = 0; t < MT; ++t) { fa[t] = __builtin_bit_cast(bf16x8, make_uint4(kt, ks, t, lane)); asm volatile("" : "+v"(fa[t])); }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) fb[t] = glds_read_frag<B_KS, TN>(sb, wn * NT + t, ks, lane);
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[t] = glds_read_frag<A_KS, TM>(sa, wm * MT + t, ks, lane);
      }
      if (ABL == 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(fb[t]));
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" ::"v"(fa[t]));
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[nt][mt], 0, 0, 0);
          if (SPREAD) {
            const int p = ks * MT + mt;  // one piece after each row of MFMAs until this wave's pieces are out
            if (p < NPIECE && more) issue_piece((kt + 1) & 1, kt + 1, p);
          }
        }
      }
    }
  }
  MAFED_TRACE_MARK(4);
  if (ABL == 4 || ABL == 5 || ABL == 6) {  // timing only: no C traffic (one element per wave keeps the accumulators live)
    float sacc = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sacc += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
    if (lane == 0) Elem<CT>::store(C + (m0 + wm * MT * 16) * epi.ldc + n0 + wn * NT * 16, sacc);
    return;
  }
  if constexpr (MT == 4 && NT == 4 && STAGE * 2 >= NW * 16384) {
    // Epilogue through LDS: the accumulator layout gives each lane 4 columns of 16 different rows (8-byte bf16 stores in
    // 32-byte row segments); re-read as 8 consecutive columns per lane so that C / aux / residual traffic moves in
    // 128-256 contiguous bytes per row.  Wave-private 64x64 fp32 region, float4 slot XOR (row & 15): conflict-free both ways.
    __syncthreads();  // every wave is done with the operand stages
    if constexpr (KG > 1) {
      // group 1 hands its partial sums to the same-numbered wave of group 0 (wave-private 16 KiB slot in group 0's stages,
      // one f32x4 per lane and accumulator: conflict-free) and leaves; from here on group 0's LDS use is wave-private
      f32x4* xch = reinterpret_cast<f32x4*>(smem_all) + wave * 1024;
      if (kg == 1) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xch[(nt * 4 + mt) * 64 + lane] = acc[nt][mt];
      }
      __syncthreads();
      if (kg == 1) return;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] += xch[(nt * 4 + mt) * 64 + lane];
      __builtin_amdgcn_wave_barrier();
    }
    float* reg = reinterpret_cast<float*>(smem) + wave * 4096;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = mt * 16 + (lane & 15);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int c4 = (nt * 4 + (lane >> 4)) ^ (row & 15);
        *reinterpret_cast<f32x4*>(reg + row * 64 + c4 * 4) = acc[nt][mt];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (gridDim.y > 1) {
      // split-K partial: C += acc with fp32 atomics, one 256-byte row segment per wave-instruction (the full-rate shape
      // of MI355X_MICROARCH "Global float atomics"); the gradient buffer already holds the running sum (beta = 1)
      float* cbase = reinterpret_cast<float*>(C) + (m0 + wm * 64) * epi.ldc + n0 + wn * 64 + lane;
#pragma unroll 8
      for (int row = 0; row < 64; ++row) {
        const float v = reg[row * 64 + ((((lane >> 2) ^ (row & 15)) << 2) | (lane & 3))];
        atomicAdd(cbase + (int64_t)row * epi.ldc, v);
      }
      return;
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    EpiPre pre[8];
    float bv[8];
    epi_bias8(epi, n0 + wn * 64 + 8 * (lane & 7), bv);
#pragma unroll
    for (int it = 0; it < 8; ++it) epi_prefetch<CT>(epi, C, m0 + wm * 64 + it * 8 + (lane >> 3), n0 + wn * 64 + 8 * (lane & 7), pre[it]);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 8 + (lane >> 3), j = lane & 7;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(reg + row * 64 + (((2 * j) ^ (row & 15)) << 2));
      const f32x4 hi = *reinterpret_cast<const f32x4*>(reg + row * 64 + (((2 * j + 1) ^ (row & 15)) << 2));
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      epilogue_store8_pre<CT>(epi, C, m0 + wm * 64 + row, n0 + wn * 64 + 8 * j, v, pre[it], bv);
      if (epi.colsum) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cs[i] += v[i];
      }
    }
    if (epi.colsum) colsum_flush_block<8, TN>(cs, reinterpret_cast<float*>(smem), epi.colsum, n0, wn * 64, lane, tid);
  } else if constexpr ((MT == 3 || MT == 6 || MT == 9) && NT == 4) {
    // 48 / 96 / 144 x 64 wave tiles (144 x 128 with six waves, 192 x 128, 288 x 256): MT / 3 passes of 3 row tiles through a
    // wave-private 48 x 64 fp32 region (the launcher sizes the dynamic LDS for it when the operand stages are smaller)
    constexpr int NPASS = MT / 3;
    __syncthreads();
    float* reg = reinterpret_cast<float*>(smem) + wave * 3072;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bv[8];
    epi_bias8(epi, n0 + wn * 64 + 8 * (lane & 7), bv);
    EpiPre pre[2][6];  // bf16 slot only, ping-pong: pass p + 1's operands are requested before pass p's stores go out
#pragma unroll
    for (int it = 0; it < 6; ++it)
      epi_prefetch<CT, false>(epi, C, m0 + (wm * MT) * 16 + it * 8 + (lane >> 3), n0 + wn * 64 + 8 * (lane & 7), pre[0][it]);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      if (pass + 1 < NPASS) {
#pragma unroll
        for (int it = 0; it < 6; ++it)
          epi_prefetch<CT, false>(epi, C, m0 + (wm * MT + (pass + 1) * 3) * 16 + it * 8 + (lane >> 3), n0 + wn * 64 + 8 * (lane & 7), pre[(pass + 1) & 1][it]);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int row = t * 16 + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int c4 = (nt * 4 + (lane >> 4)) ^ (row & 15);
          *reinterpret_cast<f32x4*>(reg + row * 64 + c4 * 4) = acc[nt][pass * 3 + t];
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int row = it * 8 + (lane >> 3), j = lane & 7;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(reg + row * 64 + (((2 * j) ^ (row & 15)) << 2));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(reg + row * 64 + (((2 * j + 1) ^ (row & 15)) << 2));
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        epilogue_store8_pre<CT, false>(epi, C, m0 + (wm * MT + pass * 3) * 16 + row, n0 + wn * 64 + 8 * j, v, pre[pass & 1][it], bv);
        if (epi.colsum) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cs[i] += v[i];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (epi.colsum) colsum_flush_block<8, TN>(cs, reinterpret_cast<float*>(smem), epi.colsum, n0, wn * 64, lane, tid);
  } else if constexpr (MT == 9 && NT == 2 && WM == 1 && STAGE * 2 >= NW * 10240) {
    // 144 x 32 wave tile: same idea in two passes (row tiles 0-4, then 5-8) through a wave-private 80 x 32 fp32 region
    __syncthreads();
    float* reg = reinterpret_cast<float*>(smem) + wave * 2560;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bv[8];
    epi_bias8(epi, n0 + wn * 32 + 8 * (lane & 3), bv);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int mt0 = pass * 5, nmt = pass == 0 ? 5 : 4;
      EpiPre pre[5];
#pragma unroll
      for (int it = 0; it < 5; ++it)
        if (it < nmt) epi_prefetch<CT>(epi, C, m0 + mt0 * 16 + it * 16 + (lane >> 2), n0 + wn * 32 + 8 * (lane & 3), pre[it]);
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (t < nmt) {
          const int row = t * 16 + (lane & 15);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int c4 = (nt * 4 + (lane >> 4)) ^ (row & 7);
            *reinterpret_cast<f32x4*>(reg + row * 32 + c4 * 4) = acc[nt][mt0 + t];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        if (it < nmt) {
          const int row = it * 16 + (lane >> 2), j = lane & 3;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(reg + row * 32 + (((2 * j) ^ (row & 7)) << 2));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(reg + row * 32 + (((2 * j + 1) ^ (row & 7)) << 2));
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          epilogue_store8_pre<CT>(epi, C, m0 + mt0 * 16 + row, n0 + wn * 32 + 8 * j, v, pre[it], bv);
          if (epi.colsum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[i] += v[i];
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (epi.colsum) colsum_flush_block<4, TN>(cs, reinterpret_cast<float*>(smem), epi.colsum, n0, wn * 32, lane, tid);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int64_t m = m0 + (wm * MT + mt) * 16 + (lane & 15);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int64_t n = n0 + (wn * NT + nt) * 16 + 4 * (lane >> 4);
        epilogue_store4<CT, true>(epi, C, m, n, make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]));
      }
    }
  }
  MAFED_TRACE_MARK(5);
}

static int g_gemm_group_m = 4;  // row-tiles per L2 patch of the grouped tile order (tuning: variant 300 + g)
static int g_gemm_nsplit = 1;  // set by the dispatcher for the next launch_bf16_glds<2,2,4,4,...> (split-K, accumulate-only)

template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, typename CT, int ABL = 0, int KG = 1>
static int launch_bf16_glds(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                            const GemmEpi& epi, hipStream_t st) {
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16, STAGES = KG * 2 * (TM + TN) * 128;
  // the staged epilogue of the 3 / 6 / 9-row-tile waves needs 12 KiB per wave: more than the operand stages of the six-wave 144 x 128 tile
  constexpr int EPI_LDS = ((MT == 3 || MT == 6 || MT == 9) && NT == 4) ? WM * WN * 12288 : 0;
  constexpr int LDS = STAGES > EPI_LDS ? STAGES : EPI_LDS;
  const int64_t tm = M / TM, tn = N / TN, nwg = tm * tn;
  if (nwg > 0x7fffffff) { set_error("gemm: grid too large"); return MAFED_EINVAL; }
  auto kfn = gemm_bf16_glds_kernel<WM, WN, MT, NT, A_KS, B_KS, CT, ABL, KG>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int nsplit = (MT == 4 && NT == 4 && WM == 2 && WN == 2) ? g_gemm_nsplit : 1;
  if (KG > 1 && ((K / BK) % (nsplit * KG) != 0 || epi.colsum)) { set_error("gemm: K groups need an even K-tile count per block and no fused column sums"); return MAFED_EINVAL; }
  launch(K_GEMM_BF16, 2.0 * M * N * K, kfn, dim3((unsigned)nwg, (unsigned)nsplit), dim3(WM * WN * KG * 64), LDS, st, M, N, K, (const bf16_t*)A, lda,
         (const bf16_t*)B, ldb, (CT*)C, epi, (int)tn | (g_gemm_group_m << 20), (int)nwg);
  return MAFED_OK;
}

// tile configurations of the LDS-DMA kernel: 0 = 128x128 (4 waves of 64x64, 2 blocks/CU), 1 = 256x256 (8 waves of 128x64,
// 1 block/CU: one K-tile of MFMAs per SIMD then covers the DMA latency), 2 = 256x128 (8 waves of 64x64)
template <bool A_KS, bool B_KS, typename CT>
static int launch_bf16_glds_cfg(int cfg, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                                const GemmEpi& epi, hipStream_t st) {
  switch (cfg) {
    case 1: return launch_bf16_glds<2, 4, 8, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 2: return launch_bf16_glds<4, 2, 4, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 11: if constexpr (!A_KS) return launch_bf16_glds<1, 4, 9, 2, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 144x128, 4 waves of 144x32
    case 12: if constexpr (!A_KS) return launch_bf16_glds<3, 2, 3, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 144x128, 6 waves of 48x64
    case 13: if constexpr (!A_KS) return launch_bf16_glds<1, 8, 9, 2, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 144x256, 8 waves of 144x32
    case 16: if constexpr (!A_KS) return launch_bf16_glds<2, 4, 9, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 288x256, 8 waves of 144x64 (136 KiB, 1 block / CU)
    case 17: if constexpr (!A_KS) return launch_bf16_glds<2, 2, 6, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 192x128, 4 waves of 96x64 (80 KiB: 2 blocks / CU)
    case 18: if constexpr (sizeof(CT) == 4) return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 0, 2>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 128x128, two K groups of 4 waves (128 KiB, 1 block / CU)
    case 14: return launch_bf16_glds<2, 2, 4, 2, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st);  // 128x64, 4 waves of 64x32 (48 KiB: 3 blocks / CU)
    case 15: return launch_bf16_glds<2, 2, 2, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st);  // 64x128
    case 21: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 1>(M, N, K, A, lda, B, ldb, C, epi, st);  // ablations (timing only)
    case 22: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 2>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 23: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 3>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 24: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 4>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 25: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 5>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 26: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 6>(M, N, K, A, lda, B, ldb, C, epi, st);
    case 29: return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT, 7>(M, N, K, A, lda, B, ldb, C, epi, st);  // 128x128, epilogue only (timing)
    case 30: if constexpr (!A_KS) return launch_bf16_glds<1, 4, 9, 2, A_KS, B_KS, CT, 7>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 144x128, epilogue only (timing)
    case 27: if constexpr (!A_KS) return launch_bf16_glds<2, 2, 6, 4, A_KS, B_KS, CT, 7>(M, N, K, A, lda, B, ldb, C, epi, st); else break;  // 192x128, epilogue only (timing)
    default: break;
  }
  return launch_bf16_glds<2, 2, 4, 4, A_KS, B_KS, CT>(M, N, K, A, lda, B, ldb, C, epi, st);
}

template <bool A_KS, bool B_KS, typename CT>
static int launch_bf16(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                       const GemmEpi& epi, hipStream_t st) {
  const int64_t tm = cdiv(M, BM), tn = cdiv(N, BN);
  const int64_t nwg = tm * tn;
  if (nwg > 0x7fffffff) { set_error("gemm: grid too large"); return MAFED_EINVAL; }
  auto kfn = gemm_bf16_kernel<A_KS, B_KS, CT>;
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    attr_set = true;
  }
  launch(K_GEMM_BF16, 2.0 * M * N * K, kfn, dim3((unsigned)nwg), dim3(256), GEMM_LDS_BYTES, st, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
         (CT*)C, epi, (int)tn, (int)nwg);
  return MAFED_OK;
}

}  // namespace mafed

using namespace mafed;

// test / tuning hook: 0 = automatic, 1 = force the register-staged kernel
static int g_gemm_variant = 0;
static int g_gemm_big = 0;    // 288x256 configuration in automatic mode (200 = off, 201 = on): faster alone, slower beside the side streams
static int g_gemm_kgroups = 0;  // two K groups per block for the one-tile-per-CU weight gradients (400 = off, 401 = on): 9 % faster alone, 2.5 % slower step (no dX block fits beside a 128 KiB block)
static int g_gemm_split = 0;  // 0 automatic, 1 never split K, n > 1 force n splits where legal
static int g_gemm_pp = 1;        // persistent ping-pong kernel (gemm_pp.hip): 700 = off, 701 = automatic (default), 710 + c = force configuration c
static int g_gemm_pp_force = -1;
static int g_gemm_pp_launches = 0;   // test hook: how many launches took the ping-pong kernel
static int g_gemm_pp_fc2 = 0;        // tuning (730 off / 731 on): the fp32 + two-residual epilogue (4h -> h product) on the persistent kernel too
extern "C" int mafed_gemm_pp_launches(void) { return g_gemm_pp_launches; }
namespace mafed { extern int g_skinny_ns, g_skinny_wide, g_attn_decode_flat, g_decode_lds; }
extern "C" int mafed_gemm_get_variant(int which) {   // which: 0 = tile-configuration variant, 7 = persistent-kernel mode (700 / 701 / 710 + c)
  if (which == 7) return g_gemm_pp_force >= 0 ? 710 + g_gemm_pp_force : (g_gemm_pp ? 701 : 700);
  if (which == 72) return 720 + gemm_pp_ticket_mode();
  if (which == 73) return gemm_pp_ticket_launches();
  return g_gemm_variant;
}
extern "C" int mafed_gemm_set_variant(int v) {
  if (v >= 720 && v <= 722) { gemm_pp_set_ticket_mode(v - 720); return MAFED_OK; }
  if (v == 730 || v == 731) { g_gemm_pp_fc2 = v - 730; return MAFED_OK; }
  if (v == 760 || v == 761) { g_decode_lds = v - 760; return MAFED_OK; }   // decode layer kernels: register-direct / LDS-staged loads
  if (v == 740 || v == 741) { g_attn_decode_flat = v - 740; return MAFED_OK; }   // decode attention: online / all-rows-in-flight form   // persistent kernels: static / ticketed tile order / per call
  if (v >= 700 && v < 800) { g_gemm_pp = v == 700 ? 0 : 1; g_gemm_pp_force = v >= 710 ? v - 710 : -1; return MAFED_OK; }
  if (v >= 600) { g_skinny_wide = v == 699 ? -1 : v - 600; return MAFED_OK; }
  if (v >= 500) { g_skinny_ns = v - 500; return MAFED_OK; }
  if (v >= 400) { g_gemm_kgroups = v - 400; return MAFED_OK; }
  if (v >= 300) { g_gemm_group_m = v - 300 > 0 ? v - 300 : 1; return MAFED_OK; }
  if (v >= 200) { g_gemm_big = v - 200; return MAFED_OK; }
  if (v >= 100) { g_gemm_split = v - 100; return MAFED_OK; }  // 100 = automatic split-K, 101 = off, 100 + n = force n
  g_gemm_variant = v;
  return MAFED_OK;
}

static int gemm_impl(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                     const void* B, int64_t ldb, void* C, int64_t ldc, mafed_dtype c_dtype, const float* bias, int epilogue,
                     void* aux, const float* res1, const float* res2, float beta, float* colsum, void* stream);
extern "C" int mafed_sumsq_accumulate(const float* x, int64_t n, float* sumsq16, void* stream);   // optim.hip

extern "C" int mafed_gemm(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                          const void* B, int64_t ldb, void* C, int64_t ldc, mafed_dtype c_dtype, const float* bias, int epilogue,
                          void* aux, const float* res1, const float* res2, float beta, void* stream) {
  return gemm_impl(in_dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, epilogue, aux, res1, res2, beta, nullptr, stream);
}

extern "C" int mafed_gemm_colsum(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                                 const void* B, int64_t ldb, void* C, int64_t ldc, mafed_dtype c_dtype, const float* bias, int epilogue,
                                 void* aux, const float* res1, const float* res2, float beta, float* colsum, void* stream) {
  return gemm_impl(in_dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, bias, epilogue, aux, res1, res2, beta, colsum, stream);
}

// Problem record of the ping-pong kernel; false when the epilogue / alignment is outside what that kernel stores directly
// (16-byte row segments: every pointer 16-byte aligned, ldc a multiple of 8; fused column sums only in its prefetching epilogues).
static bool pp_fill_problem(PPProblem& pr, bool a_ks, bool b_ks, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                            int64_t ldb, void* C, int64_t ldc, mafed_dtype c_dtype, const GemmEpi& epi, float* colsum) {
  (void)M; (void)N; (void)K;
  const uintptr_t al = (uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)epi.aux | (uintptr_t)epi.res1 | (uintptr_t)epi.res2 |
                       (uintptr_t)epi.bias | (uintptr_t)colsum;
  if ((al & 15) || ldc % 8 || lda % 8 || ldb % 8) return false;
  const bool r1 = epi.res1 != nullptr, r2 = epi.res2 != nullptr, bt = epi.beta != 0.f, pair = c_dtype == MAFED_BF16;
  const bool fast = (epi.mode == MAFED_EPI_NONE && !r1 && !r2 && !bt) || (epi.mode == MAFED_EPI_GELU && !r1 && !r2 && !bt) ||
                    (pair && epi.mode == MAFED_EPI_GELU_BWD && !r1 && !r2 && !bt) ||
                    (epi.mode == MAFED_EPI_NONE && r1 && epi.res1_bf16 && r2 && !bt) || (!pair && epi.mode == MAFED_EPI_NONE && !r1 && !r2 && bt);
  if (colsum && !fast) return false;
  // the weight-gradient instantiations (both operands reduction-major) carry the bias / beta epilogues only (PPEpilogue PLAIN_ONLY)
  if (a_ks && b_ks && (epi.mode != MAFED_EPI_NONE || r1 || r2 || colsum)) return false;
  pr.A = (const bf16_t*)A; pr.B = (const bf16_t*)B; pr.C = C;
  pr.bias = epi.bias; pr.aux = epi.aux; pr.res1 = epi.res1; pr.res2 = epi.res2; pr.colsum = colsum; pr.sumsq = nullptr;
  pr.lda = lda; pr.ldb = ldb; pr.ldc = ldc;
  pr.beta = epi.beta; pr.mode = epi.mode; pr.res1_bf16 = epi.res1_bf16;
  pr.tiles_m = pr.tiles_n = pr.nkt = pr.tile_begin = pr.pad_ = 0;
  return true;
}

/* Several independent products C_i = op(A_i).op(B_i) (same operand layouts, input and output types) as ONE launch of the persistent
 * kernel where their shapes tile it; otherwise one launch each. */
// 1 if mafed_gemm_grouped would run these `n` products (dense operands: lda / ldb = the contiguous extent) as ONE persistent launch whose
// epilogue emits the squares of C (mafed_gemm_problem.sumsq fused), 0 if they would be taken in a pass over C behind the product(s).
extern "C" int mafed_gemm_grouped_fuses_sumsq(mafed_dtype in_dtype, int transA, int transB, mafed_dtype c_dtype, const int64_t* M, const int64_t* N,
                                              const int64_t* K, int n) {
  if (!M || !N || !K || n < 1 || n > PP_MAXP || in_dtype != MAFED_BF16 || c_dtype != MAFED_F32) return 0;
  if (!((g_gemm_variant == 0 && g_gemm_pp) || g_gemm_pp_force >= 0)) return 0;
  const bool a_ks = transA != 0, b_ks = transB == 0;
  if (!(a_ks && b_ks)) return 0;
  int64_t ldas[PP_MAXP], ldbs[PP_MAXP];
  for (int i = 0; i < n; ++i) { ldas[i] = M[i]; ldbs[i] = N[i]; }
  double fill = 0.0;
  const int cfg = gemm_pp_pick(a_ks, b_ks, c_dtype, n, M, N, K, ldas, ldbs, g_gemm_pp_force, &fill);
  return cfg != PP_NONE && cfg != PP_256x256 && (g_gemm_pp_force >= 0 || fill >= 0.7) ? 1 : 0;
}

extern "C" int mafed_gemm_grouped(mafed_dtype in_dtype, int transA, int transB, mafed_dtype c_dtype, const mafed_gemm_problem* problems, int n,
                                  void* stream) {
  MAFED_CHECK_ARG(problems && n >= 1, "gemm_grouped: no problems");
  bool one = in_dtype == MAFED_BF16 && n <= PP_MAXP && ((g_gemm_variant == 0 && g_gemm_pp) || g_gemm_pp_force >= 0);
  for (int i = 0; one && i < n; ++i) one = !(problems[i].epilogue & MAFED_EPI_NO_PERSISTENT);   // per-call opt-out (mafed_hip.h)
  bool want_tickets = false;
  for (int i = 0; i < n; ++i) want_tickets = want_tickets || (problems[i].epilogue & MAFED_EPI_TICKETED) != 0;
  const bool a_ks = transA != 0, b_ks = transB == 0;
  PPProblem pr[PP_MAXP];
  int64_t Ms[PP_MAXP], Ns[PP_MAXP], Ks[PP_MAXP], ldas[PP_MAXP], ldbs[PP_MAXP];
  int cfg = PP_NONE;
  for (int i = 0; one && i < n; ++i) {
    const mafed_gemm_problem& q = problems[i];
    const int res1_bf16 = (q.epilogue & MAFED_EPI_RES1_BF16) ? 1 : 0;
    GemmEpi epi{q.bias, q.epilogue & ~(MAFED_EPI_RES1_BF16 | MAFED_EPI_NO_PERSISTENT | MAFED_EPI_TICKETED), q.aux, q.res1, q.res2, res1_bf16, q.beta, q.ldc, nullptr};
    one = q.A && q.B && q.C && q.M > 0 && (c_dtype == MAFED_F32 || q.beta == 0.f) && !(q.colsum && q.beta != 0.f) &&
          pp_fill_problem(pr[i], a_ks, b_ks, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, c_dtype, epi, q.colsum);
    // squares of the stored C: fused into the weight-gradient kernels' epilogue only (both operands reduction-major, fp32 C, not the
    // 256 x 256 kernel); any other launch computes them in a pass over C behind the product (below)
    if (one) pr[i].sumsq = q.sumsq;
    if (one && q.sumsq && !(a_ks && b_ks && c_dtype == MAFED_F32)) one = false;
    Ms[i] = q.M; Ns[i] = q.N; Ks[i] = q.K; ldas[i] = q.lda; ldbs[i] = q.ldb;
  }
  if (one) {
    double fill = 0.0;
    cfg = gemm_pp_pick(a_ks, b_ks, c_dtype, n, Ms, Ns, Ks, ldas, ldbs, g_gemm_pp_force, &fill);
    one = cfg != PP_NONE && (g_gemm_pp_force >= 0 || fill >= 0.7);
  }
  if (one) {
    // gemm_z.hip (256 x 256 tiles: the h = 768 / 2048 weight gradients) carries no fused squares: the group still goes out as ONE launch and
    // the squares are taken in a pass over each C behind it (round 4 first sent such a group back to one launch per product: 96 launches
    // instead of 12 at the 1.4B shape, 295 -> 246 samples/s)
    bool squares_after = false;
    if (cfg == PP_256x256)
      for (int i = 0; i < n; ++i)
        if (pr[i].sumsq) {
          MAFED_CHECK_ARG(problems[i].ldc == problems[i].N, "gemm_grouped: sumsq needs a dense fp32 C");
          pr[i].sumsq = nullptr;
          squares_after = true;
        }
    const int rc = gemm_pp_launch(cfg, a_ks, b_ks, c_dtype, pr, n, Ms, Ns, Ks, as_stream(stream), want_tickets);
    if (rc != MAFED_OK) return rc;
    MAFED_CHECK_LAUNCH("gemm_grouped(ping-pong)");
    ++g_gemm_pp_launches;
    if (squares_after)
      for (int i = 0; i < n; ++i)
        if (problems[i].sumsq) {
          const int rc2 = mafed_sumsq_accumulate((const float*)problems[i].C, problems[i].M * problems[i].N, problems[i].sumsq, stream);
          if (rc2 != MAFED_OK) return rc2;
        }
    return MAFED_OK;
  }
  for (int i = 0; i < n; ++i) {
    const mafed_gemm_problem& q = problems[i];
    const int rc = gemm_impl(in_dtype, transA, transB, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, c_dtype, q.bias, q.epilogue, q.aux, q.res1,
                             q.res2, q.beta, q.colsum, stream);
    if (rc != MAFED_OK) return rc;
    if (q.sumsq) {
      MAFED_CHECK_ARG(c_dtype == MAFED_F32 && q.ldc == q.N, "gemm_grouped: sumsq needs a dense fp32 C");
      const int rc2 = mafed_sumsq_accumulate((const float*)q.C, q.M * q.N, q.sumsq, stream);
      if (rc2 != MAFED_OK) return rc2;
    }
  }
  return MAFED_OK;
}

static int gemm_impl(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                     const void* B, int64_t ldb, void* C, int64_t ldc, mafed_dtype c_dtype, const float* bias, int epilogue,
                     void* aux, const float* res1, const float* res2, float beta, float* colsum, void* stream) {
  MAFED_CHECK_ARG(A && B && C, "gemm: null pointer");
  MAFED_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  MAFED_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && ldc >= N, "gemm: N=%lld and ldc=%lld must be multiples of 4 (vector epilogue)",
                  (long long)N, (long long)ldc);
  MAFED_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N), "gemm: leading dimension too small");
  const int res1_bf16 = (epilogue & MAFED_EPI_RES1_BF16) ? 1 : 0;
  const bool no_persistent = (epilogue & MAFED_EPI_NO_PERSISTENT) != 0;   // this call only: stay off the one-block-per-CU kernels
  const bool want_tickets = (epilogue & MAFED_EPI_TICKETED) != 0;          // this call only: persistent kernel in ticketed tile order
  epilogue &= ~(MAFED_EPI_RES1_BF16 | MAFED_EPI_NO_PERSISTENT | MAFED_EPI_TICKETED);
  MAFED_CHECK_ARG(epilogue >= MAFED_EPI_NONE && epilogue <= MAFED_EPI_QUICK_GELU, "gemm: unknown epilogue %d", epilogue);
  MAFED_CHECK_ARG(epilogue != MAFED_EPI_GELU_BWD || aux, "gemm: GELU_BWD epilogue needs aux");
  MAFED_CHECK_ARG(beta == 0.f || c_dtype == MAFED_F32, "gemm: beta != 0 requires an fp32 C");
  MAFED_CHECK_ARG((((uintptr_t)C | (uintptr_t)aux | (uintptr_t)res1 | (uintptr_t)res2 | (uintptr_t)bias) & 7) == 0,
                  "gemm: C/aux/res/bias must be at least 8-byte aligned");
  if (M == 0) return MAFED_OK;
  GemmEpi epi{bias, epilogue, aux, res1, res2, res1_bf16, beta, ldc, nullptr};
  hipStream_t st = as_stream(stream);
  // column sums of C: fused into the LDS-staged epilogues of the MFMA kernel where that exists, otherwise a second pass
  auto colsum_after = [&]() -> int {
    return colsum ? mafed_colsum(C, c_dtype, M, N, ldc, colsum, nullptr, 0, stream) : MAFED_OK;
  };
  MAFED_CHECK_ARG(!colsum || beta == 0.f, "gemm: colsum with beta != 0 is not defined");
  if (in_dtype == MAFED_F32) {
    dim3 grid((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64)), block(256);
    if (c_dtype == MAFED_F32) launch(K_GEMM_F32, 2.0 * M * N * K, gemm_f32_kernel<float>, grid, block, 0, st, transA, transB, M, N, K, (const float*)A, lda, (const float*)B, ldb, (float*)C, epi);
    else launch(K_GEMM_F32, 2.0 * M * N * K, gemm_f32_kernel<bf16_t>, grid, block, 0, st, transA, transB, M, N, K, (const float*)A, lda, (const float*)B, ldb, (bf16_t*)C, epi);
    MAFED_CHECK_LAUNCH("gemm(f32)");
    return colsum_after();
  }
  // bf16 MFMA path: 16-byte operand loads along the contiguous extent
  MAFED_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && lda % 8 == 0 && ldb % 8 == 0,
                  "gemm(bf16): A/B must be 16-byte aligned with leading dimensions multiple of 8");
  MAFED_CHECK_ARG((transA ? M : K) % 8 == 0 && (transB ? K : N) % 8 == 0,
                  "gemm(bf16): contiguous extents must be multiples of 8 (transA=%d M=%lld K=%lld transB=%d N=%lld)", transA,
                  (long long)M, (long long)K, transB, (long long)N);
  const bool a_ks = transA != 0, b_ks = transB == 0;
  int rc;
  if (g_gemm_variant == 0 && gemm_skinny_ok(transA, transB, M, N, K, beta, colsum)) {
    // one token per sample (decode): stream W once with N/16 blocks instead of N/128
    rc = gemm_skinny_launch(M, N, K, A, lda, B, ldb, C, c_dtype, epi, st);
    if (rc != MAFED_OK) return rc;
    MAFED_CHECK_LAUNCH("gemm(bf16, skinny)");
    return MAFED_OK;
  }
  // persistent ping-pong kernel (gemm_pp.hip): tile-aligned shapes whose tile count fills the 256 CUs in whole rounds
  if (!no_persistent && ((g_gemm_variant == 0 && g_gemm_pp) || g_gemm_pp_force >= 0)) {
    PPProblem pr;
    if (pp_fill_problem(pr, a_ks, b_ks, M, N, K, A, lda, B, ldb, C, ldc, c_dtype, epi, colsum)) {
      double fill = 0.0;
      int pcfg = gemm_pp_pick(a_ks, b_ks, c_dtype, 1, &M, &N, &K, &lda, &ldb, g_gemm_pp_force, &fill);
      // one persistent block per CU: a tile count that leaves the last round of the 256 CUs mostly empty (a single weight gradient:
      // 64 - 128 tiles) is the 128 x 128 kernel's job, or that of a grouped launch
      if (pcfg != PP_NONE && g_gemm_pp_force < 0 && fill < 0.85) pcfg = PP_NONE;
      // fp32 C with both residual operands (the 4h -> h product): the register-direct epilogue moves 64-byte (fp32) and 32-byte (bf16
      // residual) row segments, twice the memory requests of the LDS-staged epilogue of the 128 x 128 kernel for the same bytes --
      // 82.5 us against 78.5 for the whole product, at any prefetch depth (3 / 8 / 12 items: 82.4 / 83.0 / 82.4 us)
      if (pcfg != PP_NONE && g_gemm_pp_force < 0 && c_dtype == MAFED_F32 && epi.res1 && epi.res2 && !g_gemm_pp_fc2) pcfg = PP_NONE;
      if (pcfg != PP_NONE) {
        rc = gemm_pp_launch(pcfg, a_ks, b_ks, c_dtype, &pr, 1, &M, &N, &K, st, want_tickets);
        if (rc != MAFED_OK) return rc;
        MAFED_CHECK_LAUNCH("gemm(bf16, ping-pong)");
        ++g_gemm_pp_launches;
        return MAFED_OK;
      }
    }
  }
  // variant: 0 automatic, 1 register-staged kernel, 10 + c forces LDS-DMA tile configuration c
  int cfg = -1;
  if (g_gemm_variant != 1 && K % 64 == 0) {
    const bool ok128 = (M % 128 == 0) && (N % 128 == 0), ok256 = (M % 256 == 0) && (N % 256 == 0), ok256x128 = (M % 256 == 0) && (N % 128 == 0);
    if (g_gemm_variant >= 10) {
      const int want = g_gemm_variant - 10;
      const bool ok144 = (M % 144 == 0) && (N % 128 == 0) && !a_ks, ok144x256 = (M % 144 == 0) && (N % 256 == 0) && !a_ks;
      if (((want == 11 || want == 12) && ok144) || (want == 13 && ok144x256)) cfg = want;
      else if (want == 16 && (M % 288 == 0) && (N % 256 == 0) && !a_ks) cfg = 16;
      else if ((want == 17 || want == 27) && (M % 192 == 0) && (N % 128 == 0) && !a_ks) cfg = want;
      else if (want == 14 && (M % 128 == 0) && (N % 64 == 0)) cfg = 14;
      else if (want == 15 && (M % 64 == 0) && (N % 128 == 0)) cfg = 15;
      else if (want == 18 && ok128 && c_dtype == MAFED_F32 && !colsum && (K / 64) % 2 == 0) cfg = 18;
      else if (want == 29 && ok128) cfg = 29;
      else if (want == 30 && ok144) cfg = 30;
      else if (((want == 0 || (want >= 21 && want <= 26)) && ok128) || (want == 1 && ok256) || (want == 2 && ok256x128)) cfg = want;
      else if (ok128) cfg = 0;
    } else {
      // automatic: 128x128 tiles, unless 144-row tiles fill the 512 resident blocks much better (M = 9216 with N = 1024:
      // 576 tiles = 1.125 rounds vs 512 = exactly one)
      if (ok128) cfg = 0;
      if ((M % 144 == 0) && (N % 128 == 0) && !a_ks) {
        auto eff = [](int64_t t) { return (double)t / (double)(((t + 511) / 512) * 512); };
        const double e144 = eff((M / 144) * (N / 128)), e128 = ok128 ? eff((M / 128) * (N / 128)) : 0.0;
        if (e144 >= e128) cfg = 11;
        // at most one 144 x 128 tile per CU (M = 4608 = 16 x 288 rows with N = 1024: 256 tiles): six waves of 48 x 64 instead of four of
        // 144 x 32 keep a lone block's SIMDs busier -- dense 18.9 vs 22.5 us, fc2 48.8 vs 56.2, dX of fc1 47.9 vs 56.5 (vs 128 x 128)
        if ((M / 144) * (N / 128) <= 256) cfg = 12;
        // 192 x 128 tiles (96 x 64 per wave: 31 % fewer LDS fragment reads and 22 % fewer DMA pieces per MFMA, still two
        // blocks per CU) when they fill the 512 slots in whole rounds: the N = 4096 GEMMs (1536 tiles); 895-917 vs 840-856
        if ((M % 192 == 0) && (((M / 192) * (N / 128)) % 512 == 0)) cfg = 17;
        // 288 x 256 tiles (one 8-wave block per CU, half the DMA and 28 % fewer LDS reads per MFMA) when they fill the
        // 256 CUs in whole rounds: the N = 4096 GEMMs of the step (512 tiles); measured 944-958 vs 846-849 TFLOP/s
        if ((M % 288 == 0) && (N % 256 == 0) && (((M / 288) * (N / 256)) % 256 == 0) && g_gemm_big) cfg = 16;  // measured >= the 128-row tile on every M = 9216 shape once both use the LDS-staged epilogue
      }
    }
  }
  g_gemm_nsplit = 1;
  // weight gradients with one 128 x 128 tile per CU (4096 x 1024 and 1024 x 4096 at K = 9216): two K groups per block give
  // every SIMD a second wave without split-K atomics (isolated 88-94 vs 99-102 us)
  if (cfg == 0 && g_gemm_variant == 0 && g_gemm_kgroups && c_dtype == MAFED_F32 && beta == 1.0f && !bias && epilogue == MAFED_EPI_NONE && !res1 &&
      !res2 && !colsum && (M / 128) * (N / 128) >= 225 && (M / 128) * (N / 128) <= 256 && (K / 64) % 2 == 0 && K >= 2048)
    cfg = 18;
  if ((cfg == 0 || cfg == 18) && c_dtype == MAFED_F32 && beta == 1.0f && !bias && epilogue == MAFED_EPI_NONE && !res1 && !res2 && g_gemm_split != 1) {
    // weight-gradient GEMMs (dW += dY^T.X): few output tiles, very long K.  Split K so that the grid fills the chip.
    const int64_t tiles = (M / 128) * (N / 128), nkt = K / 64;
    int ns = g_gemm_split > 1 ? g_gemm_split : (tiles <= 96 ? (nkt >= 512 ? 8 : 4) : (tiles <= 224 ? 2 : 1));  // measured: 64 tiles x4, 192 tiles x2, 256 tiles x1
    if (g_gemm_split <= 1 && tiles <= 24 && nkt >= 512) ns = 32;  // the row-sparse LM head's dX: 16 tiles over K = 50304
    while (ns > 1 && nkt / ns < 8) ns >>= 1;
    if (cfg == 18) while (ns > 1 && nkt % (2 * ns) != 0) ns >>= 1;
    g_gemm_nsplit = ns;
  }
  if (cfg >= 0) {
    const bool fused_colsum = colsum && g_gemm_nsplit == 1 && (cfg == 0 || cfg == 11 || cfg == 12 || cfg == 16 || cfg == 17);
    if (fused_colsum) epi.colsum = colsum;
#define GOG(AKS, BKS)                                                                                                       \
  rc = (c_dtype == MAFED_F32) ? launch_bf16_glds_cfg<AKS, BKS, float>(cfg, M, N, K, A, lda, B, ldb, C, epi, st)             \
                              : launch_bf16_glds_cfg<AKS, BKS, bf16_t>(cfg, M, N, K, A, lda, B, ldb, C, epi, st)
    if (!a_ks && !b_ks) GOG(false, false);
    else if (!a_ks && b_ks) GOG(false, true);
    else if (a_ks && b_ks) GOG(true, true);
    else GOG(true, false);
#undef GOG
    if (rc != MAFED_OK) return rc;
    MAFED_CHECK_LAUNCH("gemm(bf16, lds-dma)");
    return fused_colsum ? MAFED_OK : colsum_after();
  }
#define GO(AKS, BKS)                                                                                         \
  rc = (c_dtype == MAFED_F32) ? launch_bf16<AKS, BKS, float>(M, N, K, A, lda, B, ldb, C, epi, st)            \
                              : launch_bf16<AKS, BKS, bf16_t>(M, N, K, A, lda, B, ldb, C, epi, st)
  if (!a_ks && !b_ks) GO(false, false);
  else if (!a_ks && b_ks) GO(false, true);
  else if (a_ks && b_ks) GO(true, true);
  else GO(true, false);
#undef GO
  if (rc != MAFED_OK) return rc;
  MAFED_CHECK_LAUNCH("gemm(bf16)");
  return colsum_after();
}
