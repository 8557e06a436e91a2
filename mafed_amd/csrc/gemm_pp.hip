// Persistent ping-pong bf16 MFMA GEMM for gfx950: C = op(A).op(B) with the fused epilogues of gemm_epilogue.h.
//
// Replaces, for the tile-aligned shapes of the training step, every nn.Linear forward / dX / dW product of
// transformers/models/gpt_neox/modeling_gpt_neox.py:38-49,192-236 (the reference runs them through torch.matmul under bf16 autocast).
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template", rebuilt from its description for three tile shapes):
//   * ONE 512-thread workgroup per CU, persistent: a block walks its tiles (tile id = round * grid + slot, XCD-aware order) and the
//     operand stream never drains -- the LDS-DMA for the first two K-tiles of tile i+1 is issued under the last K-tiles of tile i.
//   * Two groups of four waves (one wave per SIMD each) run the same phases half a phase apart ("ping-pong"): between two barriers
//     every wave reads the NEXT phase's fragments and issues LDS-DMA, then runs the current phase's MFMA cluster; group 1 takes the
//     phase barrier between the two, group 0 after the cluster, so on every SIMD one wave issues MFMAs while its partner loads.
//     One raw s_barrier per phase, NPH phases (12-16 MFMAs each) per 64-deep K-tile.  (The two-barriers-per-phase form of the guide's template measured 375-450 cycles
//     per 192-256-cycle MFMA cluster here: the barrier pair, not the memory system, set the pace.)
//   * Operands go global -> LDS by global_load_lds_dwordx4 into two K-tile stages; waits are COUNTED (s_waitcnt vmcnt(N), never 0
//     in the loop): 3-5 phases of DMA stay in flight across the barriers.  Each region of a stage (B, and the A rows of each
//     phase) is re-filled two phases after its last reader and waited for one phase before its first reader; the tables below
//     were derived by hand and are checked by tools/pp_schedule_check.py (RAW / WAR over both groups' barrier intervals).
//   * fragment-to-column map chosen so that the accumulators are stored straight from registers in 64-byte row segments
//     (bf16 C: a lane owns 8 consecutive columns of a row over two fragments), no LDS round trip in the epilogue.
//   * tile shapes: 256x256 (2x4 waves of 128x64; dW), 192x256 (2x4 waves of 96x64; N = 4096), 144x256 (1x8 waves of 144x32;
//     N = 1024 / 3072): each makes the tile count of the M = 9216 step GEMMs a whole number of rounds of 256 CUs.
//   * grouped launches: up to 16 problems (same layouts / output type) share one grid; the tile space is their concatenation.
#include <type_traits>

#include "gemm_pp.h"
#include "gemm_tiles.h"

namespace mafed {

typedef const __attribute__((address_space(4))) PPArgs* pp_args_ptr;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

#ifdef MAFED_PP_TRACE
// Tuning builds only (tools/pp_trace.py): one s_memtime stamp per event {0 interval start, 1 epilogue start, 2 epilogue end, 3 next
// tile ready} of waves 0 and 4 of the first blocks, written to LDS one event late (the stamp has returned by then: no wait in the
// traced stream) and dumped at the end of the kernel.
__device__ unsigned long long* g_pp_trace = nullptr;
extern "C" int mafed_gemm_pp_set_trace(void* buf) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define PP_STAMP(v) asm volatile("s_memtime %0" : "=s"(v))
constexpr int PP_TRACE_REC = 240, PP_TRACE_BYTES = 2 * PP_TRACE_REC * 2 * 8;
#else
#define PP_STAMP(v) do { } while (0)
constexpr int PP_TRACE_BYTES = 0;
#endif

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int pp_f2(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }                 // [k][64] image: 32-byte chunk XOR
__device__ __forceinline__ int pp_fpair(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }       // KC image read with pair-mapped rows

template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(512) void gemm_pp_kernel(PPArgs args_by_value) {
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16;
  constexpr int NPH = (MT == 8) ? 4 : 3;        // phases per K-tile
  constexpr int MTP = MT / NPH;                 // A row fragments per phase
  static_assert(WM * WN == 8 && MT % NPH == 0 && TN == 256, "8 waves, 256 columns");
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr bool PAIR = sizeof(CT) == 2;        // bf16 C: lane owns 8 consecutive columns over a fragment pair
  constexpr int A_PW = (WM == 2) ? NPH : 3;     // A pieces (1 KiB DMA instructions) per wave per K-tile
  constexpr int B_PW = 4;
  constexpr int RBB = TN * 2;                   // bytes per k-row of the [k][TN] image
  static_assert(!A_KS || (WM == 2 && MTP == 2), "regional [k][64] A image: 2 x 2 fragments per phase region");
  static_assert(WM == 2 || (WM == 1 && MT == 9), "piece tables: 2 x 4 waves, or 1 x 8 waves of 144 rows");
  constexpr int NPAIR = NT / 2;
  constexpr int NST = PAIR ? MT * NPAIR : MT * NT;   // C stores per wave per tile: lower bound of the epilogue's VMEM operations
  constexpr int NB_RD = B_KS ? (PAIR ? NPAIR : NT) : 2;
  (void)args_by_value;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  pp_args_ptr args = (pp_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, q4 = lane >> 4;

  // ---- fragment read offsets (bytes from the start of a stage; the rest are compile-time immediates) --------------------------
  int a_rd[2], b_rd[NB_RD];
  if constexpr (!A_KS) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_rd[ks] = (wm * MT * 16 + li) * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4);
  } else {
    const int F2 = ((li >> 3) & 1) | ((q4 & 1) << 1);
#pragma unroll
    for (int jf = 0; jf < 2; ++jf) a_rd[jf] = (8 * q4 + (li >> 2)) * 128 + (((wm * 2 + jf) ^ F2) << 5) + (li & 3) * 8;
  }
  if constexpr (!B_KS) {
    if constexpr (!PAIR) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b_rd[ks] = A_BYTES + (wn * NT * 16 + li) * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4);
    } else {
      const int rowb = wn * NT * 16 + 8 * (li >> 2) + (li & 3);
      const int fp = ((li >> 1) & 1) | (((li >> 2) & 3) << 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b_rd[ks] = A_BYTES + rowb * 128 + (((ks * 4 + q4) ^ fp) << 4);
    }
  } else {
    const int F = (li >> 2) | ((q4 & 1) << 2);
    const int kq = (8 * q4 + (li >> 2)) * RBB;
    if constexpr (!PAIR) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_rd[nt] = A_BYTES + kq + (((wn * NT + nt) ^ F) << 5) + (li & 3) * 8;
    } else {
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) b_rd[pr] = A_BYTES + kq + (((wn * NT + 2 * pr + ((li & 3) >> 1)) ^ F) << 5) + 16 * (li & 1);
    }
  }

  // ---- this wave's DMA pieces: LDS destination inside a stage (wave-uniform) and per-lane source byte offset -------------------
  int a_dst[A_PW], b_dst[B_PW];
  uint32_t aoff[A_PW], boff[B_PW];
  auto set_offsets = [&](int64_t lda, int64_t ldb) {
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      if constexpr (!A_KS) {
        int pj;
        if constexpr (WM == 2) pj = (wave >> 2) * (MT * 2) + i * 4 + (wave & 3);   // group i's rows of this wave's half
        else pj = i == 0 ? wave : (i == 1 ? wave + 8 : 16 + (wave & 1));           // 18 pieces over 8 waves (6 duplicates)
        a_dst[i] = pj * 1024;
        const int row = 8 * pj + (lane >> 3);
        aoff[i] = (uint32_t)(row * lda * 2) + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
      } else {
        a_dst[i] = i * 8192 + wave * 1024;
        const int k = 8 * wave + (lane >> 3), ph = lane & 7;
        const int l32 = (ph >> 1) ^ pp_f2(k);
        const int row = (l32 >> 1) * (MT * 16) + (i * 2 + (l32 & 1)) * 16 + (ph & 1) * 8;
        aoff[i] = (uint32_t)((k * lda + row) * 2);
      }
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int pj = 4 * wave + i;
      b_dst[i] = A_BYTES + pj * 1024;
      if constexpr (!B_KS) {
        const int row = 8 * pj + (lane >> 3);
        const int f = PAIR ? pp_fpair(row) : ((row >> 1) & 7);
        boff[i] = (uint32_t)(row * ldb * 2) + (uint32_t)(((lane & 7) ^ f) << 4);
      } else {
        const int pb = pj * 1024 + lane * 16;
        const int k = pb / RBB, within = pb % RBB;
        const int l32 = (within >> 5) ^ ks_f(k);
        boff[i] = (uint32_t)((k * ldb + l32 * 16 + ((within >> 4) & 1) * 8) * 2);
      }
    }
  };

  // ---- tile space ---------------------------------------------------------------------------------------------------------
  const int G = (int)gridDim.x, ntiles = args->ntiles;
  int slot;
  {
    const int b = (int)blockIdx.x, q = G >> 3, r = G & 7, x = b & 7;   // bijective XCD remap (cdna_hip_programming T1)
    slot = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  auto decode = [&](int id, int& pi, int& tm, int& tn) {
    pi = 0;
    for (int j = 1; j < args->nprobs; ++j)
      if (id >= args->p[j].tile_begin) pi = j;
    const int lt = id - args->p[pi].tile_begin;
    const int tiles_m = args->p[pi].tiles_m, tiles_n = args->p[pi].tiles_n, GM = args->group_m;
    const int gsz = GM * tiles_n, group = lt / gsz, first_m = group * GM;
    const int gm = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int within = lt - group * gsz;
    tm = first_m + within % gm;
    tn = within / gm;
  };

  // ---- DMA stream state: `pa`, `pb` point at the operand bytes of the stream's current K-tile ------------------------------------
  int d_id = slot, d_pi, d_tm, d_tn, d_kt = 0, d_nkt;
  const char *pa, *pb;
  int64_t a_step, b_step;
  auto dma_set_tile = [&](bool new_ld) {
    const int64_t lda = args->p[d_pi].lda, ldb = args->p[d_pi].ldb;
    if (new_ld) set_offsets(lda, ldb);
    pa = reinterpret_cast<const char*>(args->p[d_pi].A) + (A_KS ? (int64_t)d_tm * TM * 2 : (int64_t)d_tm * TM * lda * 2);
    pb = reinterpret_cast<const char*>(args->p[d_pi].B) + (B_KS ? (int64_t)d_tn * TN * 2 : (int64_t)d_tn * TN * ldb * 2);
    a_step = A_KS ? 128 * lda : 128;
    b_step = B_KS ? 128 * ldb : 128;
    d_nkt = args->p[d_pi].nkt;
    d_kt = 0;
  };
  auto dma_advance = [&]() {   // to the next K-tile of the stream (the block's next tile after the last K-tile; wraps to its first tile)
    ++d_kt;
    if (d_kt < d_nkt) {
      pa += a_step;
      pb += b_step;
    } else {
      const int old_pi = d_pi;
      d_id += G;
      if (d_id >= ntiles) d_id = slot;
      decode(d_id, d_pi, d_tm, d_tn);
      dma_set_tile(d_pi != old_pi);
    }
  };
  // LDS-DMA through inline asm: with the builtin hipcc (ROCm 7.2) drains vmcnt(0) before every ds_read_b64_tr_b16 that follows an
  // LDS-DMA (it cannot tell the transposing read from the DMA's LDS store), which serialised every phase of the dX / dW kernels.
  // The statement has no VGPR destination (register-safe); M0 = wave-uniform LDS byte address, saved and restored in the same
  // statement (cdna_hip_programming 5.7); source = uniform 64-bit base in SGPRs + per-lane 32-bit byte offset.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
  auto dma16 = [&](const char* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  auto dma_a = [&](int stage, int i) { dma16(pa, aoff[i], lds0 + stage * STAGE + a_dst[i]); };
  auto dma_b = [&](int stage, int i) { dma16(pb, boff[i], lds0 + stage * STAGE + b_dst[i]); };
  // issue group g (stream order) of the stream's current K-tile into `stage`
  auto dma_group = [&](int stage, int g) {
    if constexpr (NPH == 4) {          // [B0 B1] [B2 B3] [A0 A1] [A2 A3]
      if (g == 0) { dma_b(stage, 0); dma_b(stage, 1); }
      else if (g == 1) { dma_b(stage, 2); dma_b(stage, 3); }
      else if (g == 2) { dma_a(stage, 0); dma_a(stage, 1); }
      else { dma_a(stage, 2); dma_a(stage, 3); }
    } else {                           // [B0 B1 B2] [B3 A0] [A1 A2]
      if (g == 0) { dma_b(stage, 0); dma_b(stage, 1); dma_b(stage, 2); }
      else if (g == 1) { dma_b(stage, 3); dma_a(stage, 0); }
      else { dma_a(stage, 1); dma_a(stage, 2); }
    }
  };

  f32x4 acc[NT][MT];
  // fragments are read ONE PHASE AHEAD of the MFMA cluster that consumes them (a wave's own ds_read latency, ~300 cycles with eight
  // waves reading, was exposed in front of every cluster otherwise): two sets of A fragments by phase parity, two of B by K-tile parity
  // (NT = 4: one set of B fragments, 32 registers, re-read inside the last cluster of a K-tile as its two k-halves retire)
  constexpr bool FB2 = NT == 2;
  bf16x8 fb[FB2 ? 2 : 1][2][NT], fa[2][2][MTP];
#ifdef MAFED_PP_TRACE
  const bool trace_on = g_pp_trace != nullptr && blockIdx.x < 4;
  int trace_n = 0;
#endif

  auto read_b = [&](const char* st, bf16x8 (&dst)[2][NT], int ks0, int ks1) {
#pragma unroll
    for (int ks = ks0; ks < ks1; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (!B_KS) {
          const int imm = PAIR ? (32 * (nt >> 1) + 4 * (nt & 1)) * 128 : nt * 2048;
          dst[ks][nt] = *reinterpret_cast<const bf16x8*>(st + b_rd[ks] + imm);
        } else {
          const int base = PAIR ? b_rd[nt >> 1] + 8 * (nt & 1) : b_rd[nt];
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(st + base + (32 * ks) * RBB));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(st + base + (32 * ks + 4) * RBB));
          bf16x8 r;
          r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
          r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
          dst[ks][nt] = r;
        }
      }
  };
  auto read_a = [&](const char* st, int p, bf16x8 (&dst)[2][MTP]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < MTP; ++j) {
        if constexpr (!A_KS) {
          dst[ks][j] = *reinterpret_cast<const bf16x8*>(st + a_rd[ks] + (p * MTP + j) * 2048);
        } else {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(st + a_rd[j] + p * 8192 + (32 * ks) * 128));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(st + a_rd[j] + p * 8192 + (32 * ks + 4) * 128));
          bf16x8 r;
          r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
          r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
          dst[ks][j] = r;
        }
      }
  };

  // One K-tile = NPH phases, ONE barrier per phase.  Interval J (between barriers J-1 and J) = { read the fragments of phase J+1 into
  // the other fragment set; issue DMA; counted wait; MFMA cluster of phase J }: nothing in an interval waits for an LDS or a memory
  // round trip of its own.  A region (B, or the A rows of one phase) whose fragments are read in interval J-1 and consumed in
  // interval J is re-filled from interval J+1 on; data covered by a counted wait in interval j is read from interval j+1 on
  // (tools/pp_schedule_check.py).  DMA issue groups per K-tile, in stream order:
  //   NPH = 3: [B0 B1 B2] [B3 A0] [A1 A2]   slot 0: group 2 of K-tile kt+1 | slot 1: group 0 of kt+2, vmcnt(5) | slot 2: group 1 of kt+2, vmcnt(5)
  //   NPH = 4: [B0 B1] [B2 B3] [A0 A1] [A2 A3]   slot 0: group 3 of kt+1, vmcnt(8) | slot 1: group 0 of kt+2 | slot 2: group 1, vmcnt(6) | slot 3: group 2
  // `stage` = parity of the K-tile (compile-time after unrolling by two); `post` = first K-tile after an epilogue (its stores sit
  // in the VMEM queue between the DMAs: the waits that still target a DMA issued before them count NST more operations).
  // STAG: group 1 takes the barrier between its loads and its MFMA cluster (half a phase behind group 0; MI355X_MICROARCH "Two
  // waves per SIMD" item 9); it then retires the PREVIOUS interval's fragment reads with a counted lgkmcnt before that barrier,
  // which is where group 0 retires them (before its cluster).
#ifdef MAFED_PP_TRACE
  unsigned long long ts_cur = 0;
  int ts_tag = -1;
  auto trace_event = [&](int tag) {
    const unsigned long long prev = ts_cur;
    const int prev_tag = ts_tag;
    PP_STAMP(ts_cur);
    ts_tag = tag;
    if (trace_on && (wave & 3) == 0 && lane == 0 && prev_tag >= 0 && trace_n < PP_TRACE_REC) {
      unsigned long long* tr = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE + 2048) + (grp * PP_TRACE_REC + trace_n) * 2;
      tr[0] = prev; tr[1] = (unsigned long long)prev_tag;
    }
    if (prev_tag >= 0) ++trace_n;
  };
#else
  auto trace_event = [&](int) {};
#endif
#ifndef MAFED_PP_STAGGER
#define MAFED_PP_STAGGER 1
#endif
  constexpr bool STAG = MAFED_PP_STAGGER != 0 && FB2;   // (the in-cluster B refresh of NT = 4 reads B after the barrier a staggered group 1 would take: uniform program there)
  constexpr int RD_A = (A_KS ? 4 : 2) * MTP, RD_B = (B_KS ? 4 : 2) * NT;   // LDS read instructions of one phase's A / one K-tile's B fragments
  auto seg_load = [&](auto stage_c, int p, int set, bool post) {
    constexpr int S = decltype(stage_c)::value;
    if (p + 1 < NPH) {
      read_a(smem + S * STAGE, p + 1, fa[set ^ 1]);
    } else {
      if constexpr (FB2) read_b(smem + (S ^ 1) * STAGE, fb[S ^ 1], 0, 2);
      read_a(smem + (S ^ 1) * STAGE, 0, fa[set ^ 1]);
    }
    if constexpr (NPH == 4) {
      if (p == 0) dma_group(S ^ 1, 3);
      if (p == 1) { dma_advance(); dma_group(S, 0); }
      if (p == 2) dma_group(S, 1);
      if (p == 3) dma_group(S, 2);
    } else {
      if (p == 0) dma_group(S ^ 1, 2);
      if (p == 1) { dma_advance(); dma_group(S, 0); }
      if (p == 2) dma_group(S, 1);
    }
    if constexpr (NPH == 4) {
      if (p == 0) { if (post) pp_wait_vmcnt<(8 + NST > 63 ? 63 : 8 + NST)>(); else pp_wait_vmcnt<8>(); }
      if (p == 2) { if (post) pp_wait_vmcnt<(6 + NST > 63 ? 63 : 6 + NST)>(); else pp_wait_vmcnt<6>(); }
    } else {
      if (p == 1) { if (post) pp_wait_vmcnt<(5 + NST > 63 ? 63 : 5 + NST)>(); else pp_wait_vmcnt<5>(); }
      if (p == 2) pp_wait_vmcnt<5>();
    }
  };
  auto seg_mfma = [&](int S, int p, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < MTP; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt][p * MTP + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[FB2 ? S : 0][ks][nt], fa[set][ks][j], acc[nt][p * MTP + j], 0, 0, 0);
      if constexpr (!FB2) {
        // last cluster of the K-tile: this k-half of the B fragments is dead, fetch the next K-tile's into the same registers
        if (p == NPH - 1) {
          __builtin_amdgcn_sched_barrier(0);
          read_b(smem + (S ^ 1) * STAGE, fb[0], ks, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // (one lambda instance per group: a wave-uniform branch INSIDE the phase loop makes the register allocator join all accumulators
  //  at every phase and spill; the two groups' loops meet only at the epilogue)
  auto ktile = [&](auto stage_c, auto grp_c, bool post) {
    constexpr int S = decltype(stage_c)::value, GRP = decltype(grp_c)::value;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      const int set = (S * NPH + p) & 1;
      __builtin_amdgcn_sched_barrier(0);
      trace_event(0);
      seg_load(stage_c, p, set, post);
      if constexpr (STAG && GRP == 1) {
        // the reads of the previous interval (consumed by the cluster below, after the barrier) retire here: only this interval's remain
        constexpr int NEW_B = RD_B, NEW_A = RD_A;
        if (p + 1 < NPH) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEW_A > 15 ? 15 : NEW_A) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEW_A + NEW_B > 15 ? 15 : NEW_A + NEW_B) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      seg_mfma(S, p, set);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(STAG && GRP == 1)) __builtin_amdgcn_s_barrier();
    }
  };

  // ---- epilogue: straight from the accumulators, 64-byte row segments per store instruction -------------------------------------
  // Column group outermost (8 columns of a bf16 C, 4 of an fp32 C: one store instruction = 16 rows x 64 bytes), row fragments inside,
  // the operands the epilogue READS (saved pre-activation of GELU', residuals, old C) fetched PD row fragments ahead of the stores.
  // MODE / HSRC / FSRC are compile-time per instantiation (dispatched on the problem's epilogue below): HSRC = bf16 operand slot
  // (0 none, 1 aux of GELU', 2 bf16 res1), FSRC = fp32 operand slot (0 none, 1 res2, 2 old C for beta != 0).
  constexpr int NG = PAIR ? NPAIR : NT;
  constexpr int GW = PAIR ? 8 : 4;
  constexpr int GSTEP = PAIR ? 32 : 16;
  auto epilogue_fast = [&](auto mode_c, auto hsrc_c, auto fsrc_c, int pi, int tm, int tn) {
    constexpr int MODE = decltype(mode_c)::value, HSRC = decltype(hsrc_c)::value, FSRC = decltype(fsrc_c)::value;
    CT* __restrict__ C = reinterpret_cast<CT*>(args->p[pi].C);
    const float* __restrict__ bias = args->p[pi].bias;
    CT* aux = reinterpret_cast<CT*>(args->p[pi].aux);
    const bf16_t* res1 = reinterpret_cast<const bf16_t*>(args->p[pi].res1);
    const float* res2 = args->p[pi].res2;
    float* colsum = args->p[pi].colsum;
    const int64_t ldc = args->p[pi].ldc;
    const float beta = args->p[pi].beta;
    const int64_t row0 = (int64_t)tm * TM + wm * MT * 16 + li;
    const int64_t col0 = (int64_t)tn * TN + wn * NT * 16 + (PAIR ? 8 * q4 : 4 * q4);
    struct Pre { uint4 h; float4 f0, f1; };
    constexpr int PD = 3;
    float* scr = reinterpret_cast<float*>(smem + 2 * STAGE) + wave * 64;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int64_t cg = col0 + g * GSTEP;
      float bv[GW], cs[GW];
      if (bias) {
        if constexpr (PAIR) load8(bias + cg, bv);
        else { const float4 b = load4(bias + cg); bv[0] = b.x; bv[1] = b.y; bv[2] = b.z; bv[3] = b.w; }
      } else {
#pragma unroll
        for (int e = 0; e < GW; ++e) bv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < GW; ++e) cs[e] = 0.f;
      auto fetch = [&](int mt, Pre& p) {
        const int64_t o = (row0 + mt * 16) * ldc + cg;
        if constexpr (HSRC != 0) {
          const bf16_t* src = HSRC == 1 ? reinterpret_cast<const bf16_t*>(aux) : res1;
          if constexpr (PAIR) p.h = *reinterpret_cast<const uint4*>(src + o);
          else { const uint2 t = *reinterpret_cast<const uint2*>(src + o); p.h = make_uint4(t.x, t.y, 0u, 0u); }
        }
        if constexpr (FSRC != 0) {
          const float* src = FSRC == 1 ? res2 : reinterpret_cast<const float*>(C);
          p.f0 = load4(src + o);
          if constexpr (PAIR) p.f1 = load4(src + o + 4);
        }
      };
      Pre pre[MT];
      if constexpr (HSRC != 0 || FSRC != 0) {
#pragma unroll
        for (int mt = 0; mt < PD && mt < MT; ++mt) fetch(mt, pre[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (HSRC != 0 || FSRC != 0) {
          if (mt + PD < MT) fetch(mt + PD, pre[mt + PD]);
        }
        const int64_t o = (row0 + mt * 16) * ldc + cg;
        float v[GW];
        if constexpr (PAIR) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = acc[2 * g][mt][e]; v[4 + e] = acc[2 * g + 1][mt][e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[g][mt][e];
        }
#pragma unroll
        for (int e = 0; e < GW; ++e) v[e] += bv[e];
        float hv[GW];
        if constexpr (HSRC != 0) {
          if constexpr (PAIR) unpack8(pre[mt].h, hv);
          else {
            hv[0] = __uint_as_float(pre[mt].h.x << 16); hv[1] = __uint_as_float(pre[mt].h.x & 0xffff0000u);
            hv[2] = __uint_as_float(pre[mt].h.y << 16); hv[3] = __uint_as_float(pre[mt].h.y & 0xffff0000u);
          }
        }
        if constexpr (MODE == MAFED_EPI_GELU) {
          if (aux) {
            if constexpr (PAIR) store8(aux + o, v);
            else store4(aux + o, make_float4(v[0], v[1], v[2], v[3]));
          }
#pragma unroll
          for (int e = 0; e < GW; e += 2) {
            const f32x2 r = gelu_erf_fast2((f32x2){v[e], v[e + 1]});
            v[e] = r[0]; v[e + 1] = r[1];
          }
        } else if constexpr (MODE == MAFED_EPI_GELU_BWD) {
          static_assert(MODE != MAFED_EPI_GELU_BWD || HSRC == 1, "GELU' reads the saved pre-activation from the bf16 slot");
#pragma unroll
          for (int e = 0; e < GW; e += 2) {
            const f32x2 r = gelu_erf_grad_fast2((f32x2){hv[e], hv[e + 1]});
            v[e] *= r[0]; v[e + 1] *= r[1];
          }
        }
        if constexpr (HSRC == 2) {
#pragma unroll
          for (int e = 0; e < GW; ++e) v[e] += hv[e];
        }
        if constexpr (FSRC == 1) {
          v[0] += pre[mt].f0.x; v[1] += pre[mt].f0.y; v[2] += pre[mt].f0.z; v[3] += pre[mt].f0.w;
          if constexpr (PAIR) { v[4] += pre[mt].f1.x; v[5] += pre[mt].f1.y; v[6] += pre[mt].f1.z; v[7] += pre[mt].f1.w; }
        }
        if constexpr (FSRC == 2 && !PAIR) {
          v[0] += beta * pre[mt].f0.x; v[1] += beta * pre[mt].f0.y; v[2] += beta * pre[mt].f0.z; v[3] += beta * pre[mt].f0.w;
        }
        if constexpr (PAIR) store8(C + o, v);
        else store4(C + o, make_float4(v[0], v[1], v[2], v[3]));
        if (colsum) {
#pragma unroll
          for (int e = 0; e < GW; ++e) cs[e] += v[e];
        }
      }
      if (colsum) {
        // fold the 16 rows of a lane group; the wave's NT*16 column sums go through a wave-private LDS strip
#pragma unroll
        for (int e = 0; e < GW; ++e) {
          float s = cs[e];
          s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
          if (li == 0) scr[g * GSTEP + GW * q4 + e] = s;
        }
      }
    }
    if (colsum) {
      // one atomic instruction of contiguous floats per wave and tile (full-rate shape of MI355X_MICROARCH "Global float atomics")
      __builtin_amdgcn_wave_barrier();
      if (lane < NT * 16) atomicAdd(colsum + (int64_t)tn * TN + wn * NT * 16 + lane, scr[lane]);
      __builtin_amdgcn_wave_barrier();
    }
  };
  // any other epilogue combination: the in-place operand loads of gemm_epilogue.h (no prefetch, no fused column sums)
  auto epilogue_generic = [&](int pi, int tm, int tn) {
    GemmEpi e;
    e.bias = args->p[pi].bias; e.mode = args->p[pi].mode; e.aux = args->p[pi].aux;
    e.res1 = reinterpret_cast<const float*>(args->p[pi].res1); e.res2 = args->p[pi].res2; e.res1_bf16 = args->p[pi].res1_bf16;
    e.beta = args->p[pi].beta; e.ldc = args->p[pi].ldc; e.colsum = nullptr;
    CT* C = reinterpret_cast<CT*>(args->p[pi].C);
    const int64_t row0 = (int64_t)tm * TM + wm * MT * 16 + li;
    const int64_t col0 = (int64_t)tn * TN + wn * NT * 16 + (PAIR ? 8 * q4 : 4 * q4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if constexpr (PAIR) {
          float v[8];
#pragma unroll
          for (int x = 0; x < 4; ++x) { v[x] = acc[2 * g][mt][x]; v[4 + x] = acc[2 * g + 1][mt][x]; }
          epilogue_store8<CT>(e, C, row0 + mt * 16, col0 + g * GSTEP, v);
        } else {
          epilogue_store4<CT, true>(e, C, row0 + mt * 16, col0 + g * GSTEP, make_float4(acc[g][mt][0], acc[g][mt][1], acc[g][mt][2], acc[g][mt][3]));
        }
      }
  };
  auto epilogue = [&](int pi, int tm, int tn) {
    using std::integral_constant;
    const int mode = args->p[pi].mode;
    const bool r1 = args->p[pi].res1 != nullptr, r1h = r1 && args->p[pi].res1_bf16, r2 = args->p[pi].res2 != nullptr;
    const bool bt = args->p[pi].beta != 0.f;
    if (mode == MAFED_EPI_NONE && !r1 && !r2 && !bt)
      epilogue_fast(integral_constant<int, MAFED_EPI_NONE>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, pi, tm, tn);
    else if (mode == MAFED_EPI_GELU && !r1 && !r2 && !bt)
      epilogue_fast(integral_constant<int, MAFED_EPI_GELU>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, pi, tm, tn);
    else if (PAIR && mode == MAFED_EPI_GELU_BWD && !r1 && !r2 && !bt)
      epilogue_fast(integral_constant<int, PAIR ? MAFED_EPI_GELU_BWD : MAFED_EPI_NONE>{}, integral_constant<int, PAIR ? 1 : 0>{}, integral_constant<int, 0>{}, pi, tm, tn);
    else if (mode == MAFED_EPI_NONE && r1h && r2 && !bt)
      epilogue_fast(integral_constant<int, MAFED_EPI_NONE>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{}, pi, tm, tn);
    else if (!PAIR && mode == MAFED_EPI_NONE && !r1 && !r2 && bt)
      epilogue_fast(integral_constant<int, MAFED_EPI_NONE>{}, integral_constant<int, 0>{}, integral_constant<int, PAIR ? 0 : 2>{}, pi, tm, tn);
    else
      epilogue_generic(pi, tm, tn);
  };

  // ---- main ----------------------------------------------------------------------------------------------------------------
  if (slot >= ntiles) return;
  int id = slot, pi, tm, tn;
  decode(id, pi, tm, tn);
  d_pi = pi; d_tm = tm; d_tn = tn;
  dma_set_tile(true);
  // prologue = "interval -1": K-tile 0 (all groups) into stage 0 and the first NPH - 1 groups of K-tile 1 into stage 1, as the steady
  // state would have by now; every piece of K-tile 0 that phases 0 and 1 read has landed once only those of K-tile 1 (and, NPH = 4,
  // K-tile 0's last group) remain; then phase 0's fragments
#pragma unroll
  for (int g = 0; g < NPH; ++g) dma_group(0, g);
  dma_advance();
#pragma unroll
  for (int g = 0; g < NPH - 1; ++g) dma_group(1, g);
  if constexpr (NPH == 4) pp_wait_vmcnt<8>(); else pp_wait_vmcnt<5>();
  __builtin_amdgcn_s_barrier();
  read_b(smem, fb[0], 0, 2);
  read_a(smem, 0, fa[0]);
  bool first = true;
  while (true) {
    const int nkt = args->p[pi].nkt;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (grp == 0) {
      for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, !first && kt == 0);
        ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, false);
      }
    } else {
      for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, !first && kt == 0);
        ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, false);
      }
    }
    trace_event(1);
    epilogue(pi, tm, tn);
    trace_event(2);
    first = false;
    id += G;
    if (id >= ntiles) break;
    decode(id, pi, tm, tn);
    trace_event(3);
  }
  trace_event(3);
#ifdef MAFED_PP_TRACE
  if (trace_on && (wave & 3) == 0 && lane == 0) {
    const unsigned long long* tr = reinterpret_cast<const unsigned long long*>(smem + 2 * STAGE + 2048) + grp * PP_TRACE_REC * 2;
    unsigned long long* dst = g_pp_trace + ((int64_t)blockIdx.x * 2 + grp) * (PP_TRACE_REC * 2 + 1);
    const int n = trace_n < PP_TRACE_REC ? trace_n : PP_TRACE_REC;
    dst[0] = n;
    for (int i = 0; i < n * 2; ++i) dst[1 + i] = tr[i];
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int MT, int NT, bool A_KS, bool B_KS, typename CT>
static int pp_launch_t(const PPArgs& a, double flops, hipStream_t st) {
  constexpr int TM = WM * MT * 16, TN = WN * NT * 16;
  constexpr int LDS = 2 * (TM + TN) * 128 + 2048 + PP_TRACE_BYTES;
  auto kfn = gemm_pp_kernel<WM, WN, MT, NT, A_KS, B_KS, CT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int grid = a.ntiles < 256 ? a.ntiles : 256;
  launch(K_GEMM_BF16, flops, kfn, dim3((unsigned)grid), dim3(512), LDS, st, a);
  return MAFED_OK;
}

static void pp_tile_shape(int cfg, int& TM, int& TN) {
  TN = 256;
  TM = cfg == PP_256x256 ? 256 : (cfg == PP_192x256 ? 192 : 144);
}

int gemm_pp_pick(bool a_ks, bool b_ks, mafed_dtype c_dtype, int64_t M, int64_t N, int64_t K, int force_cfg) {
  if (K % 128 != 0 || K < 256 || N % 256 != 0) return PP_NONE;
  auto inst = [&](int cfg) {   // instantiated (layout, output type, configuration) combinations
    if (a_ks && b_ks) return cfg == PP_256x256 && c_dtype == MAFED_F32;
    if (a_ks) return false;
    if (cfg == PP_256x256) return false;
    if (b_ks) return c_dtype == MAFED_BF16;
    return true;
  };
  auto fits = [&](int cfg) {
    int TM, TN;
    pp_tile_shape(cfg, TM, TN);
    return inst(cfg) && M % TM == 0 && N % TN == 0;
  };
  if (force_cfg >= 0) return fits(force_cfg) ? force_cfg : PP_NONE;
  int best = PP_NONE;
  double best_cost = 1e30;
  for (int cfg = 0; cfg < 3; ++cfg) {
    if (!fits(cfg)) continue;
    int TM, TN;
    pp_tile_shape(cfg, TM, TN);
    const int64_t tiles = (M / TM) * (N / TN);
    const int64_t rounds = (tiles + 255) / 256;
    // time ~ rounds x tile area, with the measured relative loop efficiency of the wave tile (144x32 reads more LDS per MFMA)
    const double eff = cfg == PP_144x256 ? 0.88 : 1.0;
    const double cost = (double)rounds * TM * TN / eff;
    if (cost < best_cost) { best_cost = cost; best = cfg; }
  }
  return best;
}

int gemm_pp_launch(int cfg, bool a_ks, bool b_ks, mafed_dtype c_dtype, const PPProblem* probs, int n, const int64_t* Ms, const int64_t* Ns,
                   const int64_t* Ks, hipStream_t st) {
  if (n < 1 || n > PP_MAXP) { set_error("gemm_pp: 1..%d problems per launch", PP_MAXP); return MAFED_EINVAL; }
  int TM, TN;
  pp_tile_shape(cfg, TM, TN);
  PPArgs a;
  a.nprobs = n;
  a.pad_ = 0;
  int tiles = 0;
  double flops = 0.0;
  int min_tn = 1 << 30;
  for (int i = 0; i < n; ++i) {
    if (Ms[i] % TM || Ns[i] % TN || Ks[i] % 128 || Ks[i] < 256) { set_error("gemm_pp: shape does not tile"); return MAFED_EINVAL; }
    a.p[i] = probs[i];
    a.p[i].tiles_m = (int)(Ms[i] / TM);
    a.p[i].tiles_n = (int)(Ns[i] / TN);
    a.p[i].nkt = (int)(Ks[i] / 64);
    a.p[i].tile_begin = tiles;
    tiles += a.p[i].tiles_m * a.p[i].tiles_n;
    flops += 2.0 * Ms[i] * Ns[i] * Ks[i];
    if (a.p[i].tiles_n < min_tn) min_tn = a.p[i].tiles_n;
  }
  a.ntiles = tiles;
  // the 32 CUs of an XCD take 32 consecutive tile ids: GROUP_M row tiles x (32 / GROUP_M) column tiles form a compact patch
  a.group_m = min_tn >= 8 ? 4 : (min_tn >= 4 ? 8 : 16);
#define PP_GO(WM, WN, MT, NT, AKS, BKS, CT) return pp_launch_t<WM, WN, MT, NT, AKS, BKS, CT>(a, flops, st)
  const bool f32 = c_dtype == MAFED_F32;
  if (a_ks && b_ks) {
    if (cfg == PP_256x256 && f32) PP_GO(2, 4, 8, 4, true, true, float);
  } else if (!a_ks && b_ks) {
    if (cfg == PP_192x256 && !f32) PP_GO(2, 4, 6, 4, false, true, bf16_t);
    if (cfg == PP_144x256 && !f32) PP_GO(1, 8, 9, 2, false, true, bf16_t);
  } else if (!a_ks && !b_ks) {
    if (cfg == PP_192x256 && !f32) PP_GO(2, 4, 6, 4, false, false, bf16_t);
    if (cfg == PP_192x256 && f32) PP_GO(2, 4, 6, 4, false, false, float);
    if (cfg == PP_144x256 && !f32) PP_GO(1, 8, 9, 2, false, false, bf16_t);
    if (cfg == PP_144x256 && f32) PP_GO(1, 8, 9, 2, false, false, float);
  }
#undef PP_GO
  set_error("gemm_pp: configuration %d not instantiated for this layout / output type", cfg);
  return MAFED_EINVAL;
}

}  // namespace mafed
