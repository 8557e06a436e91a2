// Persistent "ping-pong" bf16 MFMA GEMM (gemm_pp.hip): problem table shared by the kernel, the launcher and the dispatcher.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"

namespace mafed {

constexpr int PP_MAXP = 16;  // problems per grouped launch (kernarg: 32 + 16 x 128 bytes)
constexpr int PP_TICKET_STRIDE = 16;   // dwords between the queue heads of a slot (one 64-byte line each: atomics on different heads do not share a line)

// One C = op(A).op(B) problem of a (possibly grouped) persistent launch.  All problems of a launch share the operand
// layouts (A_KS / B_KS), the output type and the tile configuration; shapes, leading dimensions and epilogues are their own.
struct PPProblem {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  void* aux;
  const void* res1;
  const float* res2;
  float* colsum;
  float* sumsq;   // 16 slots or null: += squares of the stored C (weight-gradient kernels only)
  int64_t lda, ldb, ldc;
  float beta;
  int mode;       // MAFED_EPI_*
  int res1_bf16;
  int tiles_m, tiles_n;
  int nkt;        // K / 64 (even)
  int tile_begin; // first tile id of this problem in the launch's tile space
  int pad_;
};

struct PPArgs {
  int nprobs, ntiles, group_m, grid;   // grid = number of workgroups of the launch
  // dynamic tile order (gemm_pp.hip "ticketed order"): eight per-XCD queue heads of THIS launch (16 dwords apart; NULL = static order) and the
  // heads of the stream's other slot, which block 0 clears for the next ticketed launch on the same stream
  unsigned* tickets;
  unsigned* tickets_clear;
  PPProblem p[PP_MAXP];
};

// tile configurations of the ping-pong kernel
enum PPConfig { PP_NONE = -1, PP_144x256 = 0, PP_128x256 = 1, PP_256x256 = 2, PP_NCFG = 3 };

// picks a configuration for a launch of n problems (PP_NONE when no configuration tiles every shape / the mode is not instantiated);
// *fill_out = tiles / (rounds x 256 CUs) of the chosen configuration x its relative loop speed (1.15 for 256 x 256)
int gemm_pp_pick(bool a_ks, bool b_ks, mafed_dtype c_dtype, int n, const int64_t* Ms, const int64_t* Ns, const int64_t* Ks, const int64_t* ldas,
                 const int64_t* ldbs, int force_cfg, double* fill_out);
// launches `n` problems (same layouts / output type / configuration) as ONE persistent grid
int gemm_pp_launch(int cfg, bool a_ks, bool b_ks, mafed_dtype c_dtype, const PPProblem* probs, int n, const int64_t* Ms, const int64_t* Ns,
                   const int64_t* Ks, hipStream_t st, bool want_tickets = false);

// CUs of the current device (grid of the persistent kernels, the dispatcher's fill estimate)
int gemm_pp_num_cus();
// tile order of the persistent launches: 0 static everywhere, 1 ticketed everywhere, 2 per call (default) -- mafed_gemm_set_variant 720 / 721 / 722
void gemm_pp_set_ticket_mode(int mode);
int gemm_pp_ticket_mode();
int gemm_pp_ticket_launches();

}  // namespace mafed
