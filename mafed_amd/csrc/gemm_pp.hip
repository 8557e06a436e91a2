// Persistent ping-pong bf16 MFMA GEMM for gfx950: C = op(A).op(B) with the fused epilogues of gemm_epilogue.h.
//
// Replaces, for the tile-aligned shapes of the training step, every nn.Linear forward / dX / dW product of
// transformers/models/gpt_neox/modeling_gpt_neox.py:38-49,192-236 (the reference runs them through torch.matmul under bf16 autocast).
//
// Structure (cdna_hip_programming.md section 5, "256^2 8-phase template", rebuilt from its description and then changed where the
// in-kernel event trace of this file, tools/pp_trace.py, showed a stall):
//   * ONE 512-thread workgroup per CU, persistent: a block walks its tiles (tile id = round * grid + slot, XCD-aware order) and the
//     operand stream never drains -- the LDS-DMA for the first K-tiles of tile i+1 is issued under the last K-tiles of tile i.
//   * 1 x 8 waves; two groups of four (one wave per SIMD each) run the same phases half a phase apart: in every interval between
//     two barriers a wave reads the NEXT phase's fragments (double-buffered), issues LDS-DMA, takes a counted wait and runs the
//     current phase's MFMA cluster; group 1 takes the phase barrier between the loads and the cluster, group 0 after the cluster,
//     so on every SIMD one wave issues MFMAs while its partner loads.  One raw s_barrier per phase, NPH phases (12-16 MFMAs each)
//     per 64-deep K-tile.  Measured steps that led here: two barriers per phase (the guide's form) 375-450 cycles per 192-256-cycle
//     cluster; fragments read in the phase that consumes them: ~300 cycles of ds_read latency exposed before every cluster.
//   * Operands go global -> LDS by global_load_lds_dwordx4 into a ring of K-tile stages; waits are COUNTED (s_waitcnt vmcnt(N), never 0
//     in the loop): 2-3 phases of DMA stay in flight across the barriers.  Each region of a stage (B, and the A rows of each phase)
//     is re-filled one interval after the barrier that retires its readers and waited for one interval before its first reader;
//     the tables are checked by tools/pp_schedule_check.py (index maps, bank conflicts, RAW / WAR of the schedule).
//     The DMA is issued through inline asm: hipcc (ROCm 7.2) drains vmcnt(0) before every ds_read_b64_tr_b16 that follows an
//     LDS-DMA builtin, and a spilled register anywhere in the kernel puts a vmcnt(0) for its scratch reload into the loop.
//   * fragment-to-column map chosen so that the accumulators are stored straight from registers in 64-byte row segments
//     (bf16 C: a lane owns 8 consecutive columns of a row over its two fragments), no LDS round trip in the epilogue.
//   * tile shapes: 144 x 256 (wave tile 144 x 32, three phases, two stages: every M = 9216 product of the step is a whole number
//     of rounds of 256 CUs -- N = 1024 / 3072 / 4096 -> 256 / 768 / 1024 tiles) and 128 x 256 (wave tile 128 x 32, two phases,
//     three stages: the weight gradients, whose operands are both reduction-major).  The loop is bound by a CU's L2 -> LDS rate
//     (~65 GB/s measured; 632 of a phase's 752 cycles with the MFMAs removed), so a larger tile -- fewer operand bytes per MFMA --
//     is the next step; 192 x 256 and 256 x 256 wave-tile variants of this structure (12 / 16 row fragments, 96 / 128 accumulator
//     registers + double-buffered fragments) were built and spilled 28 - 217 registers: a spilled register anywhere puts a
//     vmcnt(0) for its scratch reload into the loop, so they are not instantiated.
//   * grouped launches: up to 16 problems (same layouts / output type) share one grid; the tile space is their concatenation.
//   * ticketed tile order (round 4): a block's FIRST tile is static (tile id = its XCD-remapped slot, no latency at kernel start); every
//     further tile is drawn from one of eight queues -- queue x = the tiles the static schedule would give the 32 blocks with
//     blockIdx % 8 == x in rounds 1, 2, ... (same L2 patches), one returning atomic per tile on the queue's head.  A block that starts
//     late (its CU was held by another stream's kernel) or runs slowly simply draws fewer tiles; when its queue is empty it exits and the
//     CU is free for the late ones' static tiles.  The static schedule made every launch 1.6 - 1.8x as long with 8 CUs taken
//     (profiles/r03_contention.txt).  The atomic is issued by one lane in front of the epilogue and its value parked in LDS behind it
//     (compiler-counted, the only wait is the epilogue's own); the DMA stream reads it when it wraps to the next tile, one tile later,
//     so a block always holds one claimed tile beyond the one it computes.
//   * problem fields live in SGPRs and are re-read from the kernarg table only when the problem changes (a scalar load is a ~1 us
//     round trip; a chain of them in front of every tile cost a K = 1024 tile a quarter of its time).
#include <mutex>
#include <type_traits>

#include "gemm_pp.h"
#include "gemm_pp_epilogue.h"
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

// Register budget (tuning knob, default = no cap).  A persistent block owns its CU: with two waves per SIMD of 241 - 253 registers
// (allocated as 248 - 256) nothing else fits on the SIMD, and the block cannot even START while any other wave is resident there
// (tools/contention_bench.py, profiles/r03_contention_light.txt).  -DPP_VGPR_CAP=224 leaves 64 registers per SIMD to one wave of a
// side-stream kernel: the bf16-output instantiations then spill 16 - 25 registers in their epilogues only (isolated times unchanged)
// and do co-reside with a light kernel, the fp32-output ones spill 49 - 95 in the loop (grouped dW 731 vs 440 us); in the step the
// capped build gained nothing, so the default stays uncapped.  (amdgpu_num_vgpr counts half-registers of the
// unified VGPR + AGPR file on gfx950: N caps the kernel at 2 N.)
#ifndef PP_VGPR_CAP
#define PP_VGPR_CAP 252   // v252 - v255 belong to the ticket statements (tk_issue / tk_park below), never to the compiler
#endif
// (the weight-gradient instantiation -- both operands reduction-major -- needs every register: it draws its tickets synchronously, below,
//  and is the kernel gemm_pp_kernel_w without the cap; amdgpu_num_vgpr takes a literal, hence two kernels around one body)

// MT x NT fragments of 16 x 16 per wave (wave tile MT*16 x NT*16, block tile MT*16 x 8*NT*16), NPH phases per K-tile, NSTG stages.
// TK: the ticketed tile order is compiled in (gemm_pp_kernel_t / gemm_pp_kernel_w); TK = false is the static-order kernel without a
// single instruction of it (round 4: with the ticket code compiled in but switched off at run time the multi-tile products were 4 - 7 %
// slower than round 3's kernel in a same-box A/B -- code that is never executed still moves the compiler's wait counts and spills).
template <int MT, int NT, int NPH, int NSTG, bool A_KS, bool B_KS, typename CT, bool TK>
__device__ __forceinline__ void gemm_pp_body(const PPArgs& args_by_value) {
  constexpr int TM = MT * 16, TN = 8 * NT * 16;
  constexpr int MTP = MT / NPH;                 // A row fragments per phase
  static_assert(MT % NPH == 0 && TN == 256 && NT == 2, "1 x 8 waves of MT*16 x 32, 256 columns");
  static_assert((NPH == 3 && NSTG == 2) || (NPH == 2 && NSTG == 3), "DMA schedules: three phases over two stages, two phases over three");
  constexpr int A_BYTES = TM * 128, B_BYTES = TN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr bool PAIR = sizeof(CT) == 2;        // bf16 C: lane owns 8 consecutive columns over its fragment pair
  static_assert(!A_KS || MTP == 4, "regional [k][64] A image: four fragments per phase region");
  constexpr int A_PW = A_KS ? NPH : (TM + 63) / 64;   // A pieces (1 KiB DMA instructions) per wave per K-tile
  constexpr int B_PW = 4;
  static_assert(A_PW + B_PW == (NPH == 3 ? 7 : 6), "issue groups: 3+2+2 (3+2+1 for waves 2..7 of a 144-row tile) / 3+3 pieces per wave");
  constexpr int RBB = TN * 2;                   // bytes per k-row of the [k][TN] image
  constexpr int NPAIR = NT / 2;
  constexpr int NST = PPEpilogue<MT, NT, CT>::NST;   // C stores per wave per tile: lower bound of the epilogue's VMEM operations
  constexpr int NA_RD = A_KS ? MTP : 2, NB_RD = B_KS ? (PAIR ? NPAIR : NT) : 2, NVB = (PAIR && !B_KS) ? 4 : 2;
  (void)args_by_value;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  pp_args_ptr args = (pp_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int li = lane & 15, q4 = lane >> 4;

  // ---- fragment read offsets (bytes from the start of a stage; the rest are compile-time immediates) --------------------------
  int a_rd[NA_RD], b_rd[NB_RD];
  if constexpr (!A_KS) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a_rd[ks] = li * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4);
  } else {
    const int F2 = ((li >> 3) & 1) | ((q4 & 1) << 1);
#pragma unroll
    for (int jf = 0; jf < MTP; ++jf) a_rd[jf] = (8 * q4 + (li >> 2)) * 128 + ((jf ^ F2) << 5) + (li & 3) * 8;
  }
  if constexpr (!B_KS) {
    if constexpr (!PAIR) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b_rd[ks] = A_BYTES + (wave * NT * 16 + li) * 128 + (((ks * 4 + q4) ^ ((li >> 1) & 7)) << 4);
    } else {
      const int rowb = wave * NT * 16 + 8 * (li >> 2) + (li & 3);
      const int fp = ((li >> 1) & 1) | (((li >> 2) & 3) << 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b_rd[ks] = A_BYTES + rowb * 128 + (((ks * 4 + q4) ^ fp) << 4);
    }
  } else {
    const int F = (li >> 2) | ((q4 & 1) << 2);
    const int kq = (8 * q4 + (li >> 2)) * RBB;
    if constexpr (!PAIR) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_rd[nt] = A_BYTES + kq + (((wave * NT + nt) ^ F) << 5) + (li & 3) * 8;
    } else {
#pragma unroll
      for (int pr = 0; pr < NPAIR; ++pr) b_rd[pr] = A_BYTES + kq + (((wave * NT + 2 * pr + ((li & 3) >> 1)) ^ F) << 5) + 16 * (li & 1);
    }
  }

  // ---- this wave's DMA pieces ------------------------------------------------------------------------------------------------------
  // source address = operand K-tile base (SGPR pair) + piece offset (scalar, a multiple of the leading dimension) + per-lane offset.
  // The per-lane part is the same for every A piece of a wave (their row index has one parity) and has two (four: pair-mapped KC
  // image) variants over its B pieces, so a change of leading dimension (grouped launch) recomputes 3 - 5 registers, in registers.
  int a_dst[A_PW], b_dst[B_PW];           // LDS destination inside a stage (wave-uniform)
  uint32_t a_so[A_PW], b_so[B_PW];        // scalar byte offset of the piece inside the operand K-tile
  uint32_t va, vb[NVB];
  auto set_offsets = [&](uint32_t lda, uint32_t ldb) {
    const int r = lane >> 3, ph = lane & 7;
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      if constexpr (!A_KS) {
        const int pj = i < TM / 64 ? wave + 8 * i : 16 + (wave & 1);   // 144 rows: 18 pieces, the last two by waves 0 and 1 (ODD = wave < 2)
        a_dst[i] = pj * 1024;
        a_so[i] = (uint32_t)(8 * pj) * lda * 2u;
      } else {
        a_dst[i] = i * 8192 + wave * 1024;
        a_so[i] = (uint32_t)(8 * wave) * lda * 2u + (uint32_t)i * 128u;
      }
    }
    if constexpr (!A_KS) {
      va = (uint32_t)r * lda * 2u + (uint32_t)((ph ^ ((r >> 1) | ((wave & 1) << 2))) << 4);
    } else {
      const int l32 = (ph >> 1) ^ (((r >> 1) & 1) | ((wave & 1) << 1));   // f2(k), k = 8 * wave + r
      va = (uint32_t)r * lda * 2u + (uint32_t)(l32 * 16 + (ph & 1) * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) {
      const int pj = 4 * wave + i;
      b_dst[i] = A_BYTES + pj * 1024;
      if constexpr (!B_KS) b_so[i] = (uint32_t)(8 * pj) * ldb * 2u;
      else b_so[i] = (uint32_t)(2 * pj) * ldb * 2u;
    }
#pragma unroll
    for (int v = 0; v < NVB; ++v) {
      if constexpr (!B_KS) {
        const int f = PAIR ? (((r >> 1) & 1) | (v << 1)) : ((r >> 1) | (v << 2));   // row = 8 * (4 * wave + i) + r, v = i & 3 / i & 1
        vb[v] = (uint32_t)r * ldb * 2u + (uint32_t)((ph ^ f) << 4);
      } else {
        const int kl = 2 * v + (lane >> 5);                                          // k = 8 * wave + 2 * i + (lane >> 5), v = i & 1
        const int l32 = ((lane & 31) >> 1) ^ ((kl & 3) | ((wave & 1) << 2));
        vb[v] = (uint32_t)(lane >> 5) * ldb * 2u + (uint32_t)(l32 * 16 + (lane & 1) * 8) * 2u;
      }
    }
  };

  // ---- tile space ---------------------------------------------------------------------------------------------------------
  const int G = (int)gridDim.x, ntiles = args->ntiles;
  unsigned* const tk_heads = TK ? args->tickets : nullptr;
  unsigned* const tk_clear = TK ? args->tickets_clear : nullptr;
  // ticketed order (header comment): heads of this launch's eight queues, or NULL = static order (host: single round, capture, no slot)
  const bool dyn = TK && tk_heads != nullptr;
  const int tk_per = G >> 3, tk_q = (int)blockIdx.x & 7;
  int* const tk_lds = reinterpret_cast<int*>(smem + NSTG * STAGE + 2048 + PP_TRACE_BYTES);
  const uint32_t tk_lds_addr = (uint32_t)(uintptr_t)(lds_void_ptr)tk_lds;
  constexpr int TKW = 7;            // the wave whose lane 0 draws the tickets
  const bool tk_wave = dyn && wave == TKW;
  // The draw is ONE inline-asm returning atomic by lane 0 (EXEC narrowed inside the statement), invisible to hipcc's wait-count pass: a
  // compiler-visible atomic is waited for with vmcnt(0) where its result register is next written -- an MFMA destination inside the K
  // loop (measured in the ISA) -- which drains the DMA ring every K-tile.  Its value lands in `tk_raw` some 0.3 - 1.3 us later
  // (MI355X_MICROARCH "dequeue").  Where it is issued and where it is read follow from the wave's in-order VMEM queue:
  //   * issued inside the epilogue, behind its bias / operand fetches and in front of its C stores: the K loop's first wait after an
  //     epilogue tolerates the NST stores and nothing older, so it also covers the atomic -- by then the epilogue's arithmetic and the
  //     first phase have passed; no wait in the schedule changes;
  //   * parked in LDS (a plain store of lane 0) one interval after that wait, read by every wave when its DMA stream wraps to the next
  //     tile.  Between the asm and the park nothing may touch the register: tools/pp_ticket_audit.py checks the ISA of every build.
  // The landing register is v255, OUTSIDE what hipcc may allocate: the kernel's register cap (amdgpu_num_vgpr, below 256) leaves v252 - v255
  // to these two statements, so no copy, spill or re-use by the compiler can touch the register while the atomic is in flight.
  // TK_SYNC (the weight-gradient kernel: 144 K-tiles per tile, and no register to give away): the same atomic with its own vmcnt(0) in
  // the statement, parked at once -- a microsecond per 160-us tile.
  constexpr bool TK_SYNC = A_KS && B_KS;
#ifndef MAFED_PP_TK_RELAX
#define MAFED_PP_TK_RELAX 1
#endif
  constexpr bool TK_RELAX = TK && MAFED_PP_TK_RELAX != 0 && NPH == 3 && TKW >= 2 && !TK_SYNC;   // (the first wait behind an epilogue spares the ticket atomic too)
  auto tk_issue = [&]() {
    if constexpr (!TK) return;
    unsigned long long sav;
    const unsigned off = (unsigned)(tk_q * PP_TICKET_STRIDE * 4), one = 1u;
    if constexpr (!TK) {
    } else if constexpr (!TK_SYNC) {
      // (s_nop 4: the queue base may come straight from a v_readlane of a spilled SGPR -- VALU-written SGPR -> VMEM address, 5 wait states)
      asm volatile("s_nop 4\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add v255, %1, %2, %3 sc0\n\ts_mov_b64 exec, %0"
                   : "=&s"(sav)
                   : "v"(off), "v"(one), "s"(tk_heads)
                   : "memory", "v255");
    } else {
      int v;
      asm volatile("s_nop 4\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %1"
                   : "=&v"(v), "=&s"(sav)
                   : "v"(off), "v"(one), "s"(tk_heads)
                   : "memory");
      if (lane == 0) *tk_lds = v;
    }
  };
  // Park: lane 0's value (landed: see above) -> LDS, as ONE asm statement with its own lgkmcnt(0).  A compiler-visible store inside the
  // K loop (round 4, first version) made hipcc merge "store pending / not pending" at the join behind the branch and tighten the counted
  // lgkmcnt waits of EVERY K-tile: +5 % on the three-tile products with the branch never taken (same-box A/B, static order).  The
  // statement leaves no LDS operation outstanding, so the compiler's own counts stay on the safe side; the reads it flushes were issued
  // an interval ago.
  auto tk_park = [&]() {
    if constexpr (TK && !TK_SYNC) {
      unsigned long long sav;
      asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, v255\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, %0"
                   : "=&s"(sav) : "v"(tk_lds_addr) : "memory");
    }
  };
  if constexpr (TK) {
    if (dyn && blockIdx.x == 0 && wave == 0 && lane < 8 && tk_clear)
      __hip_atomic_store(tk_clear + lane * PP_TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (TK) {
    if (tk_wave) tk_issue();   // the block's second tile: oldest operation of the wave, complete behind the prologue's wait
  }
  int slot;
  {
    const int b = (int)blockIdx.x, q = G >> 3, r = G & 7, x = b & 7;   // bijective XCD remap (cdna_hip_programming T1)
    slot = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  // Problem fields live in SGPRs and are re-read from the kernarg table only when a tile belongs to another problem.
  // `dq` = the problem of the DMA stream's tile, `cq` = the problem of the tile being computed.
  struct DmaProb { const char* A; const char* B; uint32_t lda, ldb; int nkt, tiles_m, tiles_n, tile_begin; } dq;
  PPEpiProb cq;
  const int GM = args->group_m, nprobs = args->nprobs;
  auto load_dq = [&](int pi) {
    dq.A = reinterpret_cast<const char*>(args->p[pi].A); dq.B = reinterpret_cast<const char*>(args->p[pi].B);
    dq.lda = (uint32_t)args->p[pi].lda; dq.ldb = (uint32_t)args->p[pi].ldb; dq.nkt = args->p[pi].nkt;
    dq.tiles_m = args->p[pi].tiles_m; dq.tiles_n = args->p[pi].tiles_n; dq.tile_begin = args->p[pi].tile_begin;
  };
  auto load_cq = [&](int pi) {
    cq.C = args->p[pi].C; cq.bias = args->p[pi].bias; cq.aux = args->p[pi].aux; cq.res1 = args->p[pi].res1; cq.res2 = args->p[pi].res2;
    cq.colsum = args->p[pi].colsum; cq.sumsq = args->p[pi].sumsq; cq.ldc = args->p[pi].ldc; cq.beta = args->p[pi].beta; cq.mode = args->p[pi].mode;
    cq.res1_bf16 = args->p[pi].res1_bf16; cq.nkt = args->p[pi].nkt;
  };
  // tile id -> (problem, tile row, tile column) for the DMA stream; `dq` follows the problem.  Returns true when the problem changed.
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

  // ---- DMA stream state: `pa`, `pb` point at the operand bytes of the stream's current K-tile ------------------------------------
  int d_id = slot, d_pi = 0, d_tm, d_tn, d_kt = 0;
  bool has_next = false;   // the DMA stream's tile (decided when it wrapped) is a real tile of this block
  const char *pa, *pb;
  uint32_t a_step, b_step;
  auto dma_set_tile = [&](bool new_ld) {
    if (new_ld) set_offsets(dq.lda, dq.ldb);
    pa = dq.A + (A_KS ? (int64_t)d_tm * TM * 2 : (int64_t)d_tm * TM * dq.lda * 2);
    pb = dq.B + (B_KS ? (int64_t)d_tn * TN * 2 : (int64_t)d_tn * TN * dq.ldb * 2);
    a_step = A_KS ? 128u * dq.lda : 128u;
    b_step = B_KS ? 128u * dq.ldb : 128u;
    d_kt = 0;
  };
  auto dma_advance = [&]() {   // to the next K-tile of the stream (the block's next tile after the last K-tile; wraps to its first tile)
    ++d_kt;
    if (d_kt < dq.nkt) {
      pa += a_step;
      pb += b_step;
    } else {
      // next tile of this block: static order id + G, ticketed order the id parked in LDS behind the previous epilogue (the prologue)
      int nid = d_id + G;
      if (TK && dyn) {   // (one asm statement = read + its own wait: a volatile LDS load makes hipcc drain vmcnt(0) -- the DMA ring -- around it)
        int v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(tk_lds_addr) : "memory");
        const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane(v), r = t / (unsigned)tk_per;   // t-th draw of queue tk_q
        nid = r >= 0x10000u ? 0x7fffffff : (int)((1u + r) * (unsigned)G + (unsigned)(tk_q * tk_per) + (t - r * (unsigned)tk_per));
      }
      has_next = (unsigned)nid < (unsigned)ntiles;   // (unsigned: whatever the ticket word holds, a tile id outside the launch is never decoded)
      d_id = has_next ? nid : slot;   // (no further tile: the stream idles on the block's first tile, its pieces are never read)
      const bool changed = decode(d_id, d_pi, d_tm, d_tn);
      dma_set_tile(changed);
    }
  };
  // LDS-DMA through inline asm (header comment).  No VGPR destination (register-safe); M0 = wave-uniform LDS byte address, saved and
  // restored in the same statement (cdna_hip_programming 5.7); source = uniform 64-bit base in SGPRs + per-lane 32-bit byte offset.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void_ptr)smem;
#ifndef MAFED_PP_ABL
#define MAFED_PP_ABL 0   // tuning builds (timing only, wrong results): 1 no DMA in the K loop, 2 no fragment reads, 3 no MFMA, 4 = 1 + 2,
                         // 5 no epilogue, 6 accumulators not cleared between tiles, 7 = 5 + 6
#endif
  bool abl_dma_on = true;
  auto dma16 = [&](const char* sbase, uint32_t voff, uint32_t lds_dst) {
    if ((MAFED_PP_ABL == 1 || MAFED_PP_ABL == 4) && !abl_dma_on) return;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
  };
  auto dma_a = [&](int so, int i) { dma16(pa + a_so[i], va, lds0 + so + a_dst[i]); };
  auto dma_b = [&](int so, int i) { dma16(pb + b_so[i], vb[(PAIR && !B_KS) ? (i & 3) : (i & 1)], lds0 + so + b_dst[i]); };
  // issue group g (stream order) of the stream's current K-tile into the stage at byte offset `so`
  auto dma_group = [&](int so, int g) {
    if constexpr (NPH == 3) {          // [B0 B1 B2] [B3 A0] [A1 A2]
      if (g == 0) { dma_b(so, 0); dma_b(so, 1); dma_b(so, 2); }
      else if (g == 1) { dma_b(so, 3); dma_a(so, 0); }
      else { dma_a(so, 1); if (TM % 64 == 0 || wave < 2) dma_a(so, 2); }
    } else {                           // [B0 B1 B2] [B3 A0 A1]
      if (g == 0) { dma_b(so, 0); dma_b(so, 1); dma_b(so, 2); }
      else { dma_b(so, 3); dma_a(so, 0); dma_a(so, 1); }
    }
  };

  f32x4 acc[NT][MT];
  // fragments are read ONE PHASE AHEAD of the MFMA cluster that consumes them: two sets of A fragments by phase parity, two of B by
  // K-tile parity
  bf16x8 fb[2][2][NT], fa[2][2][MTP];
  if (MAFED_PP_ABL == 2 || MAFED_PP_ABL == 4) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[a][ks][nt] = __builtin_bit_cast(bf16x8, make_uint4(lane, a, ks, nt));
#pragma unroll
        for (int j = 0; j < MTP; ++j) fa[a][ks][j] = __builtin_bit_cast(bf16x8, make_uint4(lane, a, ks, j));
      }
  }
#ifdef MAFED_PP_TRACE
  const bool trace_on = g_pp_trace != nullptr && blockIdx.x < 4;
  int trace_n = 0;
#endif

  auto read_b = [&](const char* st, bf16x8 (&dst)[2][NT], int ks0 = 0, int ks1 = 2) {
    if (MAFED_PP_ABL == 2 || MAFED_PP_ABL == 4) {
#pragma unroll
      for (int ks = ks0; ks < ks1; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(dst[ks][nt]));
      return;
    }
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
    if (MAFED_PP_ABL == 2 || MAFED_PP_ABL == 4) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < MTP; ++j) asm volatile("" : "+v"(dst[ks][j]));
      return;
    }
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
  // (tools/pp_schedule_check.py).  DMA issue groups per K-tile, in stream order, and the phase slot that issues them:
  //   3 phases, 2 stages: [B0 B1 B2] [B3 A0] [A1 A2]   slot 0: group 2 of K-tile kt+1 | slot 1: group 0 of kt+2, vmcnt(5) | slot 2: group 1 of kt+2, vmcnt(5)
  //     (a three-stage ring for this schedule, 5 - 6 phases of cover instead of 2 - 3, measured SLOWER, 824 vs 920 TFLOP/s on the
  //      qkv product: the loop is bound by the L2 -> LDS rate of a CU, ~65 GB/s -- MI355X_MICROARCH "Indexed rows: gather into LDS"
  //      gives 66-73 -- i.e. 632 of a phase's 752 cycles with the MFMAs removed, not by the latency of a piece)
  //   2 phases, 3 stages: [B0 B1 B2] [B3 A0 A1]        slot 0: group 1 of K-tile kt+2, vmcnt(6) | slot 1: group 0 of kt+3
  // `S` = parity of the K-tile (compile-time after unrolling by two: fragment sets; with two stages also the stage); `post` = 1 / 2 in
  // the first / second K-tile after an epilogue (its stores sit in the VMEM queue between the DMAs: a wait that still targets a DMA
  // issued before them counts NST more operations), else 0; `last` = last K-tile of the tile (the next tile reads its phase-0 fragments itself, after
  // the epilogue: nothing stays live across it).
  // Group 1 takes the barrier between its loads and its cluster (half a phase behind group 0; MI355X_MICROARCH "Two waves per SIMD"
  // item 9) and retires the PREVIOUS interval's fragment reads with a counted lgkmcnt before it, where group 0 retires them (before
  // its cluster).
  int st0 = 0, st1 = STAGE, st2 = 2 * STAGE;   // three stages: byte offsets of the stages of K-tiles kt, kt+1, kt+2 (rotating)
  (void)st2;
#ifdef MAFED_PP_TRACE
  unsigned long long ts_cur = 0;
  int ts_tag = -1;
  auto trace_event = [&](int tag) {
    const unsigned long long prev = ts_cur;
    const int prev_tag = ts_tag;
    PP_STAMP(ts_cur);
    ts_tag = tag;
    if (trace_on && (wave & 3) == 0 && lane == 0 && prev_tag >= 0 && trace_n < PP_TRACE_REC) {
      unsigned long long* tr = reinterpret_cast<unsigned long long*>(smem + NSTG * STAGE + 2048) + (grp * PP_TRACE_REC + trace_n) * 2;
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
  constexpr bool STAG = MAFED_PP_STAGGER != 0;
  constexpr int RD_A = (A_KS ? 4 : 2) * MTP, RD_B = (B_KS ? 4 : 2) * NT;   // LDS read instructions of one phase's A / one K-tile's B fragments
  auto seg_load = [&](auto stage_c, int p, int set, int post, bool last) {
    constexpr int S = decltype(stage_c)::value;
    const int so_c = NSTG == 2 ? S * STAGE : st0, so_n = NSTG == 2 ? (S ^ 1) * STAGE : st1;
    // ticket park: the interval after the first wait that covers the atomic (slot 2 of the first K-tile behind an epilogue -> slot 0 of
    // the second), in front of this interval's fragment reads (the counted lgkmcnt waits below cover "all but the youngest reads")
    if constexpr (TK) {
      if ((TK_RELAX ? (post == 2 && p == 0) : (post == 1 && p == (NPH == 3 ? 2 : 1))) && tk_wave) { tk_park(); __builtin_amdgcn_sched_barrier(0); }
    }
    if (p + 1 < NPH) {
      read_a(smem + so_c, p + 1, fa[set ^ 1]);
    } else if (!last) {
      read_b(smem + so_n, fb[S ^ 1]);
      read_a(smem + so_n, 0, fa[set ^ 1]);
    }
    if constexpr (NPH == 3 && NSTG == 2) {
      if (p == 0) dma_group(so_n, 2);
      if (p == 1) { dma_advance(); dma_group(so_c, 0); }
      if (p == 2) dma_group(so_c, 1);
      // (the wait of slot 1 leaves group 2 of K-tile kt+1 and group 0 of kt+2 in flight: 2 + 3 pieces, 1 + 3 for the waves whose
      //  group 2 is a single piece)
      if (p == 1) {
        if (TM % 64 == 0 || wave < 2) { if (post == 1) pp_wait_vmcnt<(5 + NST > 63 ? 63 : 5 + NST)>(); else pp_wait_vmcnt<5>(); }
        else {
          // (the ticket wave -- wave 7, in this branch -- tolerates ONE more operation here: its ticket atomic sits in the queue just in
          //  front of the epilogue's stores, and what this wait has to cover is older than it; the atomic then has until the wait of
          //  slot 2 -- the epilogue plus two intervals, ~1.7 us with the shortest epilogue -- before it can hold the wave up)
          if (post == 1) { if (tk_wave && TK_RELAX) pp_wait_vmcnt<(5 + NST > 63 ? 63 : 5 + NST)>(); else pp_wait_vmcnt<(4 + NST > 63 ? 63 : 4 + NST)>(); }
          else pp_wait_vmcnt<4>();
        }
      }
      if (p == 2) pp_wait_vmcnt<5>();
    } else {
      if (p == 0) dma_group(st2, 1);
      if (p == 1) { dma_advance(); dma_group(st0, 0); }
      if (p == 0) { if (post == 1) pp_wait_vmcnt<(6 + NST > 63 ? 63 : 6 + NST)>(); else pp_wait_vmcnt<6>(); }
    }
  };
  auto seg_mfma = [&](int S, int p, int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < MTP; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (MAFED_PP_ABL == 3) { asm volatile("" ::"v"(fb[S][ks][nt]), "v"(fa[set][ks][j])); continue; }
          acc[nt][p * MTP + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[S][ks][nt], fa[set][ks][j], acc[nt][p * MTP + j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
  };
  // (one lambda instance per group: a wave-uniform branch INSIDE the phase loop makes the register allocator join all accumulators
  //  at every phase and spill; the two groups' loops meet only at the epilogue)
  auto ktile = [&](auto stage_c, auto grp_c, int post, bool last) {
    constexpr int S = decltype(stage_c)::value, GRP = decltype(grp_c)::value;
#pragma unroll
    for (int p = 0; p < NPH; ++p) {
      const int set = (S * NPH + p) & 1;
      __builtin_amdgcn_sched_barrier(0);
      trace_event(0);
      seg_load(stage_c, p, set, post, last);
      if constexpr (STAG && GRP == 1) {
        // the reads of the previous interval (consumed by the cluster below, after the barrier) retire here: only this interval's remain
        if (p + 1 < NPH) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RD_A > 15 ? 15 : RD_A) : "memory");
        else if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RD_A + RD_B > 15 ? 15 : RD_A + RD_B) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      seg_mfma(S, p, set);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(STAG && GRP == 1)) __builtin_amdgcn_s_barrier();
    }
    if constexpr (NSTG == 3) { const int t = st0; st0 = st1; st1 = st2; st2 = t; }
  };

  // ---- epilogue (gemm_pp_epilogue.h): straight from the accumulators, 64-byte row segments per store instruction -----------------
  auto epilogue = [&](int tm, int tn) {
    const bool draw = tk_wave && has_next;   // (wave-uniform: the ticket wave of a block that has a further tile)
    PPEpilogue<MT, NT, CT, false, (A_KS && B_KS)>::run(acc, cq, (int64_t)tm * TM + li, (int64_t)tn * TN + wave * NT * 16, lane,
                                reinterpret_cast<float*>(smem + NSTG * STAGE) + wave * 64, [&]() { if constexpr (TK) { if (draw) tk_issue(); } });
  };

  // ---- main ----------------------------------------------------------------------------------------------------------------
  if (slot >= ntiles) return;
  int id = slot, pi = 0, tm, tn;
  load_dq(0);
  decode(id, d_pi, d_tm, d_tn);
  pi = d_pi; tm = d_tm; tn = d_tn;
  load_cq(pi);
  dma_set_tile(true);
  // prologue = "interval -1": what the steady state would have issued by now, then a wait that covers every piece of K-tile 0
  if constexpr (NPH == 3 && NSTG == 2) {
    dma_group(0, 0); dma_group(0, 1); dma_group(0, 2);
    dma_advance();
    dma_group(STAGE, 0); dma_group(STAGE, 1);
    pp_wait_vmcnt<5>();
  } else {
    dma_group(0, 0); dma_group(0, 1);
    dma_advance();
    dma_group(STAGE, 0); dma_group(STAGE, 1);
    dma_advance();
    dma_group(2 * STAGE, 0);
    pp_wait_vmcnt<9>();
  }
  if constexpr (TK) {
    if (tk_wave) tk_park();   // (landed: the atomic is older than every piece the wait above covers)
  }
  __builtin_amdgcn_s_barrier();
  abl_dma_on = false;
  bool first = true;
  while (true) {
    const int nkt = cq.nkt;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (!(MAFED_PP_ABL == 6 || MAFED_PP_ABL == 7) || first) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // phase 0's fragments (K-tile 0 of this tile landed and was published before the previous tile's last barrier / the prologue's)
    read_b(smem + (NSTG == 2 ? 0 : st0), fb[0]);
    read_a(smem + (NSTG == 2 ? 0 : st0), 0, fa[0]);
#ifdef MAFED_PP_ALIGN
    // tuning: start of the K loops on a 64-byte boundary (+ MAFED_PP_ALIGN_PAD dwords): the loop is a hand-scheduled stream of asm
    // statements -- does its placement matter (MI355X_MICROARCH "Two waves per SIMD" item 8)?
#define PP_ALIGN_STR2(x) #x
#define PP_ALIGN_STR(x) PP_ALIGN_STR2(x)
#define PP_LOOP_ALIGN() asm volatile(".p2align 6\n\t.rept " PP_ALIGN_STR(MAFED_PP_ALIGN_PAD) "\n\ts_nop 0\n\t.endr")
#else
#define PP_LOOP_ALIGN() do { } while (0)
#endif
    if (grp == 0) {
      PP_LOOP_ALIGN();
      for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, (!first && kt == 0) ? 1 : 0, false);
        ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, (!first && kt == 0) ? 2 : 0, kt + 2 >= nkt);
      }
    } else {
      PP_LOOP_ALIGN();
      for (int kt = 0; kt < nkt; kt += 2) {
        ktile(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, (!first && kt == 0) ? 1 : 0, false);
        ktile(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, (!first && kt == 0) ? 2 : 0, kt + 2 >= nkt);
      }
    }
    trace_event(1);
    // (ticket for the tile after next: drawn inside the epilogue, parked in the first K-tile of the next tile)
    if (!(MAFED_PP_ABL == 5 || MAFED_PP_ABL == 7) || (TK ? !has_next : id + G >= ntiles)) epilogue(tm, tn);
    else if (TK && tk_wave) tk_issue();
    trace_event(2);
    first = false;
    if constexpr (TK) {
      if (!has_next) break;
      id = d_id;
    } else {   // (the static kernel keeps round 3's own tile counter: nothing of the ticket bookkeeping is live across its K loop)
      id += G;
      if (id >= ntiles) break;
    }
    // the DMA stream switched to this tile K-tiles ago: its coordinates are already decoded
    if (d_pi != pi) load_cq(d_pi);
    pi = d_pi; tm = d_tm; tn = d_tn;
    trace_event(3);
  }
  trace_event(3);
#ifdef MAFED_PP_TRACE
  if (trace_on && (wave & 3) == 0 && lane == 0) {
    const unsigned long long* tr = reinterpret_cast<const unsigned long long*>(smem + NSTG * STAGE + 2048) + grp * PP_TRACE_REC * 2;
    unsigned long long* dst = g_pp_trace + ((int64_t)blockIdx.x * 2 + grp) * (PP_TRACE_REC * 2 + 1);
    const int n = trace_n < PP_TRACE_REC ? trace_n : PP_TRACE_REC;
    dst[0] = n;
    for (int i = 0; i < n * 2; ++i) dst[1 + i] = tr[i];
  }
#endif
}

// static tile order: round 3's kernel (no register cap below 256, no ticket code)
template <int MT, int NT, int NPH, int NSTG, bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(128))) void gemm_pp_kernel(PPArgs args_by_value) {
  gemm_pp_body<MT, NT, NPH, NSTG, A_KS, B_KS, CT, false>(args_by_value);
}
// ticketed tile order, asynchronous draw: v252 - v255 are kept from the compiler (PP_VGPR_CAP), the atomic lands in v255
template <int MT, int NT, int NPH, int NSTG, bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(PP_VGPR_CAP / 2))) void gemm_pp_kernel_t(PPArgs args_by_value) {
  static_assert(!(A_KS && B_KS), "the weight-gradient layout draws synchronously: gemm_pp_kernel_w");
  gemm_pp_body<MT, NT, NPH, NSTG, A_KS, B_KS, CT, true>(args_by_value);
}
// ticketed tile order, synchronous draw (the weight-gradient layout: no register to give away, 144 K-tiles per tile)
template <int MT, int NT, int NPH, int NSTG, bool A_KS, bool B_KS, typename CT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(128))) void gemm_pp_kernel_w(PPArgs args_by_value) {
  static_assert(A_KS && B_KS, "synchronous tickets, no register cap: the weight-gradient layout only");
  gemm_pp_body<MT, NT, NPH, NSTG, A_KS, B_KS, CT, true>(args_by_value);
}

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
// ---- ticket slots ------------------------------------------------------------------------------------------------------------------
// Two slots of eight queue heads per stream, used alternately by that stream's ticketed launches: launches of one stream run in order,
// so when launch n starts launch n - 1 has finished with the other slot -- block 0 of launch n clears it for launch n + 1.  No reset
// launch, no host-side count of the draws, nothing shared between streams.  The memory is a zero-initialised __device__ array of this
// code object (the library never allocates).
constexpr int PP_TK_STREAMS = 64, PP_TK_SLOT = 8 * PP_TICKET_STRIDE;
__device__ unsigned g_pp_ticket_mem[PP_TK_STREAMS * 2 * PP_TK_SLOT];
namespace {
std::mutex g_tk_mu;   // launches come from the caller's thread and from autograd's backward thread
struct TkStream { hipStream_t st; int dev; int parity; };
TkStream g_tk_streams[PP_TK_STREAMS];
int g_tk_n = 0;
unsigned* g_tk_base[16] = {};
int g_tk_mode = 2;     // mafed_gemm_set_variant: 720 = static order everywhere, 721 = ticketed everywhere, 722 = per call (MAFED_EPI_TICKETED; default)
int g_tk_launches = 0; // test hook: launches that ran in ticketed order
int g_num_cus = 0;
}  // namespace
void gemm_pp_set_ticket_mode(int mode) { g_tk_mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
int gemm_pp_ticket_mode() { return g_tk_mode; }
int gemm_pp_ticket_launches() { return g_tk_launches; }

// CUs of the current device (the persistent grid and the dispatcher's fill estimate; 256 on MI355X, fewer in a partition mode)
int gemm_pp_num_cus() {
  if (g_num_cus > 0) return g_num_cus;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) g_num_cus = n;
  else { (void)hipGetLastError(); return 256; }   // (no device: the CPU-side checks of the dispatcher)
  return g_num_cus;
}

// Slot pointers for a ticketed launch on `st` (false: table full / capture in progress / no device symbol -> static order).  Caller holds g_tk_mu.
static bool tk_acquire(hipStream_t st, unsigned** cur, unsigned** other, int* entry) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cs != hipStreamCaptureStatusNone) return false;   // a captured launch would replay with the same slot: static order inside graphs
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); return false; }
  if (!g_tk_base[dev]) {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_pp_ticket_mem)) != hipSuccess || !ptr) { (void)hipGetLastError(); return false; }
    g_tk_base[dev] = reinterpret_cast<unsigned*>(ptr);
  }
  int e = -1, same_dev = 0;
  for (int i = 0; i < g_tk_n; ++i) {
    if (g_tk_streams[i].dev != dev) continue;
    if (g_tk_streams[i].st == st) { e = i; break; }
    ++same_dev;
  }
  if (e < 0) {
    if (g_tk_n >= PP_TK_STREAMS) return false;
    (void)same_dev;
    e = g_tk_n++;
    g_tk_streams[e] = TkStream{st, dev, 0};
  }
  unsigned* base = g_tk_base[dev] + (size_t)e * 2 * PP_TK_SLOT;
  *cur = base + g_tk_streams[e].parity * PP_TK_SLOT;
  *other = base + (g_tk_streams[e].parity ^ 1) * PP_TK_SLOT;
  *entry = e;
  return true;
}

template <int MT, int NT, int NPH, int NSTG, bool A_KS, bool B_KS, typename CT>
static int pp_launch_t(const PPArgs& a_in, double flops, hipStream_t st, bool ticketed) {
  constexpr int TM = MT * 16, TN = 8 * NT * 16;
  constexpr int LDS = NSTG * (TM + TN) * 128 + 2048 + PP_TRACE_BYTES + 64;   // stages | epilogue strips | (trace) | ticket word (ticketed kernels)
  void (*kfn)(PPArgs) = gemm_pp_kernel<MT, NT, NPH, NSTG, A_KS, B_KS, CT>;   // static tile order
  void (*kfn_t)(PPArgs);                                                        // ticketed tile order
  if constexpr (A_KS && B_KS) kfn_t = gemm_pp_kernel_w<MT, NT, NPH, NSTG, A_KS, B_KS, CT>;
  else kfn_t = gemm_pp_kernel_t<MT, NT, NPH, NSTG, A_KS, B_KS, CT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)kfn_t, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  const int ncu = gemm_pp_num_cus();
  const int grid = a_in.ntiles < ncu ? a_in.ntiles : ncu;
  PPArgs a = a_in;
  a.grid = grid;
  a.tickets = a.tickets_clear = nullptr;
  if (ticketed && a.ntiles > grid && grid % 8 == 0) {
    // more than one round: ticketed order.  Slot choice, parity flip and launch are one critical section, so that the alternation of
    // the two slots follows the stream's launch order whichever host thread launches.
    std::lock_guard<std::mutex> lk(g_tk_mu);
    int e = -1;
    if (tk_acquire(st, &a.tickets, &a.tickets_clear, &e)) {
      launch(K_GEMM_PP, flops, kfn_t, dim3((unsigned)grid), dim3(512), LDS, st, a);
      if (hipPeekAtLastError() == hipSuccess) { g_tk_streams[e].parity ^= 1; ++g_tk_launches; }   // (a refused launch never ran: the slot stays the stream's current one)
      return MAFED_OK;
    }
    a.tickets = a.tickets_clear = nullptr;
  }
  launch(K_GEMM_PP, flops, kfn, dim3((unsigned)grid), dim3(512), LDS, st, a);
  return MAFED_OK;
}

int gemm_z_launch(bool a_ks, bool b_ks, mafed_dtype c_dtype, const PPArgs& a, double flops, hipStream_t st);   // gemm_z.hip: 256 x 256 tiles

static void pp_tile_shape(int cfg, int& TM, int& TN) {
  TN = 256;
  TM = cfg == PP_128x256 ? 128 : (cfg == PP_256x256 ? 256 : 144);
}

static bool pp_instantiated(int cfg, bool a_ks, bool b_ks, mafed_dtype c_dtype) {
  if (a_ks && b_ks) return (cfg == PP_128x256 || cfg == PP_256x256) && c_dtype == MAFED_F32;
  if (a_ks) return false;
  if (cfg == PP_128x256) return false;
  if (cfg == PP_256x256) return c_dtype == MAFED_BF16 || !b_ks;
  if (b_ks) return c_dtype == MAFED_BF16;
  return true;
}

// Configuration for a launch of n problems (PP_NONE when none tiles every problem).  Whole rounds of the 256 CUs first, then the
// 256 x 256 kernel: it moves 32 operand bytes per MFMA cycle through the L2 -> LDS path that bounds these loops, the others 44 - 48.
int gemm_pp_pick(bool a_ks, bool b_ks, mafed_dtype c_dtype, int n, const int64_t* Ms, const int64_t* Ns, const int64_t* Ks, const int64_t* ldas,
                 const int64_t* ldbs, int force_cfg, double* fill_out) {
  for (int i = 0; i < n; ++i) {
    if (Ks[i] % 128 != 0 || Ks[i] < 256 || Ns[i] % 256 != 0) return PP_NONE;
    // a tile's fixed cost (pipeline fill + epilogue, ~6-12k cycles) against ~600 cycles per 64-deep k step: below K = 512 the
    // 128 x 128 kernel's many small blocks win (LM-head weight gradient, K = 256 labelled rows: 129 us here, 62 us there)
    if (force_cfg < 0 && Ks[i] < 512) return PP_NONE;
    if (ldas[i] % 64 != 0 || ldbs[i] % 64 != 0 || ldas[i] >= (1 << 22) || ldbs[i] >= (1 << 22)) return PP_NONE;   // 32-bit piece offsets
  }
  int best = PP_NONE;
  double best_score = -1.0, best_fill = 0.0;
  for (int cfg = 0; cfg < PP_NCFG; ++cfg) {
    if (force_cfg >= 0 && cfg != force_cfg) continue;
    if (!pp_instantiated(cfg, a_ks, b_ks, c_dtype)) continue;
    int TM, TN;
    pp_tile_shape(cfg, TM, TN);
    int64_t tiles = 0;
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) {
      ok = Ms[i] % TM == 0 && Ns[i] % TN == 0;
      tiles += (Ms[i] / TM) * (Ns[i] / TN);
    }
    if (!ok) continue;
    const int64_t ncu = gemm_pp_num_cus();
    const int64_t rounds = (tiles + ncu - 1) / ncu;
    const double fill = (double)tiles / (double)(rounds * ncu);
    // 256 x 256 tiles run their k loop ~15 % faster per flop (half the LDS-DMA bytes per MFMA); at equal cost the 8-wave kernel wins
    const double score = fill * (cfg == PP_256x256 ? 1.15 : 1.0);
    if (score > best_score) { best_score = score; best = cfg; best_fill = fill; }
  }
  (void)best_fill;
  if (fill_out) *fill_out = best_score;   // "effective fill": what the callers' thresholds compare
  return best;
}

int gemm_pp_launch(int cfg, bool a_ks, bool b_ks, mafed_dtype c_dtype, const PPProblem* probs, int n, const int64_t* Ms, const int64_t* Ns,
                   const int64_t* Ks, hipStream_t st, bool want_tickets) {
  if (n < 1 || n > PP_MAXP) { set_error("gemm_pp: 1..%d problems per launch", PP_MAXP); return MAFED_EINVAL; }
  int TM, TN;
  pp_tile_shape(cfg, TM, TN);
  PPArgs a;
  a.nprobs = n;
  a.grid = 0;   // (set by the launcher: the kernels read the grid size from the table, not from the dispatch packet)
  a.tickets = a.tickets_clear = nullptr;
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
  if (cfg == PP_256x256) return gemm_z_launch(a_ks, b_ks, c_dtype, a, flops, st);
  // tile order of this launch: the caller's flag (MAFED_EPI_TICKETED, per call) unless the tuning hook forces one (720 static / 721 ticketed)
  const bool ticketed = g_tk_mode == 2 ? want_tickets : g_tk_mode == 1;
#define PP_GO(MT, NT, NPH, NSTG, AKS, BKS, CT) return pp_launch_t<MT, NT, NPH, NSTG, AKS, BKS, CT>(a, flops, st, ticketed)
  const bool f32 = c_dtype == MAFED_F32;
  if (a_ks && b_ks) {
    if (cfg == PP_128x256 && f32) PP_GO(8, 2, 2, 3, true, true, float);
  } else if (!a_ks && b_ks) {
    if (cfg == PP_144x256 && !f32) PP_GO(9, 2, 3, 2, false, true, bf16_t);
  } else if (!a_ks && !b_ks) {
    if (cfg == PP_144x256 && !f32) PP_GO(9, 2, 3, 2, false, false, bf16_t);
    if (cfg == PP_144x256 && f32) PP_GO(9, 2, 3, 2, false, false, float);
  }
#undef PP_GO
  set_error("gemm_pp: configuration %d not instantiated for this layout / output type", cfg);
  return MAFED_EINVAL;
}

}  // namespace mafed
