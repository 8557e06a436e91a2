// Persistent 256 x 256-tile bf16 MFMA GEMM for gfx950, ONE wave per SIMD: C = op(A).op(B) with the fused epilogues of gemm_pp_epilogue.h.
//
// Why a second persistent kernel: the 8-wave kernel of gemm_pp.hip is bound by a CU's L2 -> LDS rate (~65 GB/s: 632 of a phase's 752
// cycles with the MFMAs removed), i.e. by operand bytes per MFMA, and its wave tiles (144 x 32, 128 x 32: 44 - 48 operand bytes per MFMA
// cycle) cannot grow -- two waves per SIMD have 256 registers each, and 128 accumulators plus double-buffered fragments spill.  Here a
// 256-thread block puts ONE wave on each SIMD with the whole 512-register file: a 2 x 2 wave grid of 128 x 128 wave tiles (256
// accumulator registers, 8 x 8 fragments of 16 x 16) computes a 256 x 256 tile -- 32 operand bytes per MFMA cycle through the DMA path
// and 64 B/clk of LDS fragment reads (153 in gemm_pp) -- and hides its own latencies in software instead of behind a partner wave:
//   * K advances in 32-deep steps through a ring of FOUR LDS stages (2 x 16 KiB each); the LDS-DMA of step s+4 is issued during step
//     s into the stage whose fragments were read during step s-1, and waited for (counted vmcnt(16)) before the barrier of step s+2:
//     three steps (~3 us of MFMA work) of cover.  ONE barrier per step (64 MFMAs per wave).
//   * the fragments of step s+1 are read (double-buffered registers) while the MFMAs of step s issue: eight groups of { 8 MFMAs,
//     one A and one B fragment read, one DMA piece } keep the matrix pipe fed from a single in-order instruction stream.
//   * [row][32 k] images with 64-byte rows (XOR of the 16-byte chunk with perm[(row >> 2) & 3], perm = 0 3 2 1: conflict-free
//     ds_read_b128 for natural and pair-mapped rows), [32 k][256] images read with ds_read_b64_tr_b16 for reduction-major operands.
//   * persistent blocks, problem table, tile order, grouped launches, SGPR-cached problem fields, inline-asm LDS-DMA: as gemm_pp.hip.
// Replaces the same reference products as gemm_pp.hip (tf:38-49,192-236) where a launch's tile count fills whole rounds of 256 CUs:
// grouped weight gradients (192 tiles per layer, four layers = three rounds) and pairs of independent forward / dX products.
#include <type_traits>

#include "gemm_pp.h"
#include "gemm_pp_epilogue.h"
#include "gemm_tiles.h"

namespace mafed {

typedef const __attribute__((address_space(4))) PPArgs* z_args_ptr;
typedef __attribute__((address_space(3))) bf16x4* z_lds_bf16x4_ptr;

#ifdef MAFED_PP_TRACE
// Tuning builds only (tools/pp_trace.py --z): one s_memtime stamp per event {0 step start, 1 epilogue start, 2 epilogue end, 3 next tile
// ready}, written to LDS one event late and dumped at the end of the kernel (waves 0 and 2 of the first blocks).
__device__ unsigned long long* g_z_trace = nullptr;
extern "C" int mafed_gemm_z_set_trace(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_z_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
constexpr int Z_TRACE_REC = 240, Z_TRACE_BYTES = 2 * Z_TRACE_REC * 2 * 8;
#else
constexpr int Z_TRACE_BYTES = 0;
#endif

template <int N>
__device__ __forceinline__ void z_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int z_perm(int b) { return (4 - b) & 3; }   // 0 3 2 1

template <bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(256) void gemm_z_kernel(PPArgs args_by_value) {
  constexpr int TM = 256, TN = 256, MT = 8, NT = 8;
  constexpr int OPB = 256 * 64;            // bytes of one operand's 32-deep step image
  constexpr int STAGE = 2 * OPB, NSTG = 4;
  constexpr bool PAIR = sizeof(CT) == 2;
  constexpr int NPAIR = NT / 2;
  constexpr int NST = PPEpilogue<MT, NT, CT, true>::NST;
  constexpr int NA_RD = A_KS ? MT : 1, NB_RD = B_KS ? (PAIR ? NPAIR : NT) : 1, NVB = B_KS ? 2 : 1;
  (void)args_by_value;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  z_args_ptr args = (z_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, q4 = lane >> 4;

  // ---- fragment read offsets (bytes from the start of a stage) ----------------------------------------------------------------
  int a_rd[NA_RD], b_rd[NB_RD];
  {
    const int F = (li >> 2) | ((q4 & 1) << 2);                 // ks_f(k), k = 8 * q4 + (li >> 2) (+4)
    const int kq = (8 * q4 + (li >> 2)) * 512;
    const int sw = (q4 ^ z_perm((li >> 2) & 3)) << 4;          // [row][32 k] image: chunk q4 of a row whose 4-row block is li >> 2
    if constexpr (!A_KS) a_rd[0] = (wr * 128 + li) * 64 + sw;
    else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a_rd[mt] = kq + (((wr * 8 + mt) ^ F) << 5) + (li & 3) * 8;
    }
    if constexpr (!B_KS) {
      if constexpr (!PAIR) b_rd[0] = OPB + (wc * 128 + li) * 64 + sw;
      else b_rd[0] = OPB + (wc * 128 + 8 * (li >> 2) + (li & 3)) * 64 + sw;
    } else if constexpr (!PAIR) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_rd[nt] = OPB + kq + (((wc * 8 + nt) ^ F) << 5) + (li & 3) * 8;
    } else {
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) b_rd[pr] = OPB + kq + (((wc * 8 + 2 * pr + ((li & 3) >> 1)) ^ F) << 5) + 16 * (li & 1);
    }
  }

  // ---- this wave's DMA pieces (1 KiB each): four of A and four of B per step --------------------------------------------------------
  // source address = operand step base (SGPR pair) + piece offset (scalar) + per-lane offset (one register per operand; two for a
  // reduction-major B, whose swizzle depends on the piece's parity)
  uint32_t a_so[4], b_so[4], va[A_KS ? 2 : 1], vb[NVB];
  auto set_offsets = [&](uint32_t lda, uint32_t ldb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pj = wave + 4 * i;
      a_so[i] = (A_KS ? 2u * pj : 16u * pj) * lda * 2u;
      b_so[i] = (B_KS ? 2u * pj : 16u * pj) * ldb * 2u;
    }
    const int r = lane >> 2, ch = lane & 3;
    if constexpr (!A_KS) va[0] = (uint32_t)r * lda * 2u + (uint32_t)((ch ^ z_perm((r >> 2) & 3)) << 4);
    if constexpr (!B_KS) {
      const int f = PAIR ? z_perm((2 * wave + (r >> 3)) & 3) : z_perm((r >> 2) & 3);
      vb[0] = (uint32_t)r * ldb * 2u + (uint32_t)((ch ^ f) << 4);
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {   // k = 2 * (wave + 4 i) + (lane >> 5), v = i & 1
      const int l32 = ((lane & 31) >> 1) ^ (((2 * wave + (lane >> 5)) & 3) | (v << 2));
      const uint32_t colb = (uint32_t)(l32 * 16 + (lane & 1) * 8) * 2u;
      if constexpr (A_KS) va[v] = (uint32_t)(lane >> 5) * lda * 2u + colb;
      if constexpr (B_KS) vb[v] = (uint32_t)(lane >> 5) * ldb * 2u + colb;
    }
  };

  // ---- tile space (as gemm_pp.hip) -----------------------------------------------------------------------------------------------
  const int G = (int)gridDim.x, ntiles = args->ntiles;
  int slot;
  {
    const int b = (int)blockIdx.x, q = G >> 3, r = G & 7, x = b & 7;
    slot = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  struct DmaProb { const char* A; const char* B; uint32_t lda, ldb; int nks, tiles_m, tiles_n, tile_begin; } dq;
  PPEpiProb cq;
  const int GM = args->group_m, nprobs = args->nprobs;
  auto load_dq = [&](int pi) {
    dq.A = reinterpret_cast<const char*>(args->p[pi].A); dq.B = reinterpret_cast<const char*>(args->p[pi].B);
    dq.lda = (uint32_t)args->p[pi].lda; dq.ldb = (uint32_t)args->p[pi].ldb; dq.nks = 2 * args->p[pi].nkt;
    dq.tiles_m = args->p[pi].tiles_m; dq.tiles_n = args->p[pi].tiles_n; dq.tile_begin = args->p[pi].tile_begin;
  };
  auto load_cq = [&](int pi) {
    cq.C = args->p[pi].C; cq.bias = args->p[pi].bias; cq.aux = args->p[pi].aux; cq.res1 = args->p[pi].res1; cq.res2 = args->p[pi].res2;
    cq.colsum = args->p[pi].colsum; cq.sumsq = args->p[pi].sumsq; cq.ldc = args->p[pi].ldc; cq.beta = args->p[pi].beta; cq.mode = args->p[pi].mode;
    cq.res1_bf16 = args->p[pi].res1_bf16; cq.nkt = args->p[pi].nkt;
  };
  auto decode = [&](int id, int& pi, int& tm, int& tn) -> bool {
    bool changed = false;
    if (id < dq.tile_begin || id >= dq.tile_begin + dq.tiles_m * dq.tiles_n) {
      int np = 0;
      for (int j = 1; j < nprobs; ++j)
        if (id >= args->p[j].tile_begin) np = j;
      pi = np;
      load_dq(np);
      changed = true;
    }
    const int lt = id - dq.tile_begin;
    const int gsz = GM * dq.tiles_n, group = lt / gsz, first_m = group * GM;
    const int gm = dq.tiles_m - first_m < GM ? dq.tiles_m - first_m : GM;
    const int within = lt - group * gsz;
    tm = first_m + within % gm;
    tn = within / gm;
    return changed;
  };

  // ---- DMA stream: `pa`, `pb` = operand bytes of the stream's current step -----------------------------------------------------------
  int d_id = slot, d_pi = 0, d_tm, d_tn, d_ks = 0;
  const char *pa, *pb;
  uint32_t a_step, b_step;
  auto dma_set_tile = [&](bool new_ld) {
    if (new_ld) set_offsets(dq.lda, dq.ldb);
    pa = dq.A + (A_KS ? (int64_t)d_tm * TM * 2 : (int64_t)d_tm * TM * dq.lda * 2);
    pb = dq.B + (B_KS ? (int64_t)d_tn * TN * 2 : (int64_t)d_tn * TN * dq.ldb * 2);
    a_step = A_KS ? 64u * dq.lda : 64u;
    b_step = B_KS ? 64u * dq.ldb : 64u;
    d_ks = 0;
  };
  auto dma_advance = [&]() {
    ++d_ks;
    if (d_ks < dq.nks) {
      pa += a_step;
      pb += b_step;
    } else {
      d_id += G;
      if (d_id >= ntiles) d_id = slot;
      const bool changed = decode(d_id, d_pi, d_tm, d_tn);
      dma_set_tile(changed);
    }
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
  auto dma16 = [&](const char* sbase, uint32_t voff, uint32_t lds_dst) {   // (inline asm: gemm_pp.hip header)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  // piece g (0..7) of the stream's current step into stage `st`: A pieces 0..3, then B pieces 0..3
  auto dma_piece = [&](int st, int g) {
    if (g < 4) dma16(pa + a_so[g], va[A_KS ? (g & 1) : 0], lds0 + st * STAGE + (wave + 4 * g) * 1024);
    else dma16(pb + b_so[g - 4], vb[B_KS ? (g & 1) : 0], lds0 + st * STAGE + OPB + (wave + 4 * (g - 4)) * 1024);
  };

  f32x4 acc[NT][MT];
  bf16x8 fa[2][MT], fb[2][NT];   // fragments of the current step and of the next one
#ifdef MAFED_PP_TRACE
  const bool trace_on = g_z_trace != nullptr && blockIdx.x < 4 && (wave & 1) == 0 && lane == 0;
  int trace_n = 0, ts_tag = -1;
  unsigned long long ts_cur = 0;
  auto trace_event = [&](int tag) {
    const unsigned long long prev = ts_cur;
    const int prev_tag = ts_tag;
    asm volatile("s_memtime %0" : "=s"(ts_cur));
    ts_tag = tag;
    if (trace_on && prev_tag >= 0 && trace_n < Z_TRACE_REC) {
      unsigned long long* tr = reinterpret_cast<unsigned long long*>(smem + NSTG * STAGE + 2048) + ((wave >> 1) * Z_TRACE_REC + trace_n) * 2;
      tr[0] = prev; tr[1] = (unsigned long long)prev_tag;
    }
    if (prev_tag >= 0) ++trace_n;
  };
#else
  auto trace_event = [&](int) {};
#endif
  auto read_a1 = [&](const char* st, int mt, bf16x8& dst) {
    if constexpr (!A_KS) {
      dst = *reinterpret_cast<const bf16x8*>(st + a_rd[0] + mt * 1024);
    } else {
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((z_lds_bf16x4_ptr)(st + a_rd[mt]));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((z_lds_bf16x4_ptr)(st + a_rd[mt] + 4 * 512));
      bf16x8 r;
      r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
      r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
      dst = r;
    }
  };
  auto read_b1 = [&](const char* st, int nt, bf16x8& dst) {
    if constexpr (!B_KS) {
      const int imm = PAIR ? (32 * (nt >> 1) + 4 * (nt & 1)) * 64 : nt * 1024;
      dst = *reinterpret_cast<const bf16x8*>(st + b_rd[0] + imm);
    } else {
      const int base = PAIR ? b_rd[nt >> 1] + 8 * (nt & 1) : b_rd[nt];
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((z_lds_bf16x4_ptr)(st + base));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((z_lds_bf16x4_ptr)(st + base + 4 * 512));
      bf16x8 r;
      r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
      r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
      dst = r;
    }
  };

  // One 32-deep step s (stage S = s & 3, fragment set SET = s & 1): eight groups of { 8 MFMAs of row fragment g, the reads of A row
  // fragment g and B column fragment g of step s+1 (stage S+1, other set), DMA piece g of step s+4 into stage S }, then
  // { lgkmcnt(0): this step's reads have retired -- after the barrier other waves re-fill their stage; counted vmcnt: step s+2 has
  // landed; barrier }.  `post`: first two steps after an epilogue (its stores sit between the DMAs in the VMEM queue); `last`: no
  // next step in this tile (the next tile reads its step-0 fragments itself, after the epilogue).
  auto kstep = [&](auto stage_c, bool post, bool last) {
    constexpr int S = decltype(stage_c)::value, SET = S & 1;
    const char* stn = smem + ((S + 1) & 3) * STAGE;
    trace_event(0);
    dma_advance();
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      if (!last) {
        read_a1(stn, g, fa[SET ^ 1][g]);
        read_b1(stn, g, fb[SET ^ 1][g]);
      }
      dma_piece(S, g);
      // in-place accumulate in the accumulator file, through inline asm: with the builtin hipcc gives every MFMA a destination
      // different from its C operand and, the 256 accumulators filling the AGPR file exactly, shuffles tiles through VGPRs -- 108
      // v_accvgpr_write per 64 MFMAs.  (Dependent accumulate chains need no wait states; the fragments come from ds_reads the
      // compiler waits for; the first reader of the accumulators other than an MFMA, the epilogue, sits behind explicit s_nops.)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[nt][g]) : "v"(fb[SET][nt]), "v"(fa[SET][g]));
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (post) z_wait_vmcnt<(16 + NST > 63 ? 63 : 16 + NST)>();
    else z_wait_vmcnt<16>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- main ----------------------------------------------------------------------------------------------------------------
  if (slot >= ntiles) return;
  int id = slot, pi = 0, tm, tn;
  load_dq(0);
  decode(id, d_pi, d_tm, d_tn);
  pi = d_pi; tm = d_tm; tn = d_tn;
  load_cq(pi);
  dma_set_tile(true);
  // prologue: steps 0..3 of the stream into stages 0..3 (the stream cursor ends on step 3; every kstep() advances it first)
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    if (st > 0) dma_advance();
#pragma unroll
    for (int g = 0; g < 8; ++g) dma_piece(st, g);
  }
  z_wait_vmcnt<16>();   // steps 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  bool first = true;
  while (true) {
    const int nks = 2 * cq.nkt;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // step 0's fragments; they must have retired in every wave before step 0 re-fills stage 0
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      read_a1(smem, g, fa[0][g]);
      read_b1(smem, g, fb[0][g]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nks; s += 4) {
      const bool post = !first && s == 0;
      kstep(std::integral_constant<int, 0>{}, post, false);
      kstep(std::integral_constant<int, 1>{}, post, false);
      kstep(std::integral_constant<int, 2>{}, false, false);
      kstep(std::integral_constant<int, 3>{}, false, s + 4 >= nks);
    }
    trace_event(1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> first non-MFMA reader: up to 18 wait states, none inserted for asm
    PPEpilogue<MT, NT, CT, true, (A_KS && B_KS), false>::run(acc, cq, (int64_t)tm * TM + wr * 128 + li, (int64_t)tn * TN + wc * 128, lane,
                                reinterpret_cast<float*>(smem + NSTG * STAGE) + wave * 128);
    trace_event(2);
    first = false;
    id += G;
    if (id >= ntiles) break;
    if (d_pi != pi) load_cq(d_pi);
    pi = d_pi; tm = d_tm; tn = d_tn;
    trace_event(3);
  }
  trace_event(3);
#ifdef MAFED_PP_TRACE
  if (trace_on) {
    const unsigned long long* tr = reinterpret_cast<const unsigned long long*>(smem + NSTG * STAGE + 2048) + (wave >> 1) * Z_TRACE_REC * 2;
    unsigned long long* dst = g_z_trace + ((int64_t)blockIdx.x * 2 + (wave >> 1)) * (Z_TRACE_REC * 2 + 1);
    const int n = trace_n < Z_TRACE_REC ? trace_n : Z_TRACE_REC;
    dst[0] = n;
    for (int i = 0; i < n * 2; ++i) dst[1 + i] = tr[i];
  }
#endif
}

template <bool A_KS, bool B_KS, typename CT>
static int z_launch_t(const PPArgs& a, double flops, hipStream_t st) {
  constexpr int LDS = 4 * 2 * 256 * 64 + 2048 + Z_TRACE_BYTES;
  auto kfn = gemm_z_kernel<A_KS, B_KS, CT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int ncu = gemm_pp_num_cus();
  const int grid = a.ntiles < ncu ? a.ntiles : ncu;   // (static tile order: the ticketed order of gemm_pp.hip is not built into this kernel)
  PPArgs a2 = a;
  a2.grid = grid;
  launch(K_GEMM_PP, flops, kfn, dim3((unsigned)grid), dim3(256), LDS, st, a2);
  return MAFED_OK;
}

// 256 x 256 tiles (PP_256x256): called by gemm_pp_launch
int gemm_z_launch(bool a_ks, bool b_ks, mafed_dtype c_dtype, const PPArgs& a, double flops, hipStream_t st) {
  const bool f32 = c_dtype == MAFED_F32;
  if (a_ks && b_ks && f32) return z_launch_t<true, true, float>(a, flops, st);
  if (!a_ks && b_ks && !f32) return z_launch_t<false, true, bf16_t>(a, flops, st);
  if (!a_ks && !b_ks && !f32) return z_launch_t<false, false, bf16_t>(a, flops, st);
  if (!a_ks && !b_ks && f32) return z_launch_t<false, false, float>(a, flops, st);
  set_error("gemm_z: layout / output type not instantiated");
  return MAFED_EINVAL;
}

}  // namespace mafed
