// Epilogue of the persistent MFMA GEMM kernels (gemm_pp.hip, gemm_z.hip): straight from the accumulators, 64-byte row segments per
// store instruction.  acc[nt][mt] = the 16 x 16 fragment (row fragment mt, column fragment nt) in the TRANSPOSED accumulator layout:
// lane (li = lane & 15, q4 = lane >> 4) holds row li and, bf16 C, columns 32 * (nt >> 1) + 8 * q4 + 4 * (nt & 1) + {0..3} of the
// wave's NT * 16 columns (pair-mapped fragments: 8 consecutive columns over a fragment pair = one 16-byte store), fp32 C, columns
// 16 * nt + 4 * q4 + {0..3}.
#pragma once
#include <type_traits>

#include "common.h"
#include "gemm_epilogue.h"

#ifndef PP_EPI_PD
#define PP_EPI_PD 8   // items (16 rows x 64 bytes per operand) the epilogue's operand loads run ahead of its stores
#endif

namespace mafed {

struct PPEpiProb { void* C; const float* bias; void* aux; const void* res1; const float* res2; float* colsum; float* sumsq; int64_t ldc; float beta; int mode, res1_bf16, nkt; };

// Column group outermost (8 columns of a bf16 C, 4 of an fp32 C: one store instruction = 16 rows x 64 bytes), row fragments inside,
// the operands the epilogue READS (saved pre-activation of GELU', residuals, old C) fetched PD row fragments ahead of the stores.
// MODE / HSRC / FSRC are compile-time per instantiation (dispatched on the problem's epilogue in run()): HSRC = bf16 operand slot
// (0 none, 1 aux of GELU', 2 bf16 res1), FSRC = fp32 operand slot (0 none, 1 res2, 2 old C for beta != 0).
// row0 = first row of the wave's tile + li; colw = first column of the wave's tile; scr = NT * 16 wave-private floats in LDS.
// ACC_AGPR: the accumulators live in the accumulator register file (gemm_z.hip: inline-asm MFMAs with "+a" operands); they are read
// one fragment at a time through v_accvgpr_read -- left to itself hipcc copies all 256 of them to VGPRs in front of the epilogue and
// spills them to scratch.
template <bool ACC_AGPR>
__device__ __forceinline__ f32x4 pp_acc_read(const f32x4& a) {
  if constexpr (!ACC_AGPR) return a;
  else {
    f32x4 r;
    asm("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3])
        : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]));
    return r;
  }
}

// PLAIN_ONLY: the instantiation only ever sees bias / beta epilogues (the weight-gradient kernels: both operands reduction-major);
// leaving the activation code out keeps those kernels below the register count at which hipcc starts to spill (3 VGPRs with the
// polynomial GELU compiled in -- a spill anywhere puts a vmcnt(0) into the K loop).  The dispatcher never sends them anything else
// (pp_fill_problem).
// SUMSQ: the weight-gradient epilogue also folds the squares of what it stores into cq.sumsq (off for the 256 x 256-tile kernel: with it
// compiled in, gemm_z_kernel<true, true, float> spilled 492 bytes and its launches took 357 instead of 284 us at the 1.4B shape)
template <int MT, int NT, typename CT, bool ACC_AGPR = false, bool PLAIN_ONLY = false, bool SUMSQ = PLAIN_ONLY>
struct PPEpilogue {
  static constexpr bool PAIR = sizeof(CT) == 2;
  static constexpr int NPAIR = NT / 2;
  static constexpr int NG = PAIR ? NPAIR : NT;
  static constexpr int GW = PAIR ? 8 : 4;
  static constexpr int GSTEP = PAIR ? 32 : 16;
  static constexpr int NST = PAIR ? MT * NPAIR : MT * NT;   // C stores per wave per tile: lower bound of the epilogue's VMEM operations

  // `hook`: called once, behind the epilogue's bias / first operand fetches and in front of its first C store (gemm_pp.hip issues the
  // ticket atomic there: older than the stores, younger than the fetches this wave is about to wait for)
  template <int MODE, int HSRC, int FSRC, typename Hook>
  static __device__ __forceinline__ void fast(f32x4 (&acc)[NT][MT], const PPEpiProb& cq, int64_t row0, int64_t colw, int lane, float* scr, Hook&& hook) {
    const int li = lane & 15, q4 = lane >> 4;
    CT* __restrict__ C = reinterpret_cast<CT*>(cq.C);
    const float* __restrict__ bias = cq.bias;
    CT* aux = reinterpret_cast<CT*>(cq.aux);
    const bf16_t* res1 = reinterpret_cast<const bf16_t*>(cq.res1);
    const float* res2 = cq.res2;
    float* colsum = cq.colsum;
    const int64_t ldc = cq.ldc;
    const float beta = cq.beta;
    const int64_t col0 = colw + (PAIR ? 8 * q4 : 4 * q4);
    struct Pre { uint4 h; float4 f0, f1; };
    // The items (column group g, row fragment mt) form ONE sequence, g outermost; the operands item i reads are fetched PD items
    // ahead, across group boundaries: with 8 waves per CU and ~2 us to HBM a depth of 3 kept 36 KB in flight per CU -- the
    // residual-add epilogue of the 4h -> h product (57 MB read) then ran 11.6 us, at PD = 8 the reads are a third of that.
    constexpr int NI = NG * MT;
    constexpr int PD = (HSRC != 0 || FSRC != 0) ? (NI < PP_EPI_PD ? NI : PP_EPI_PD) : 0;
    float bv[NG][GW];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int64_t cg = col0 + g * GSTEP;
      if (bias) {
        if constexpr (PAIR) load8(bias + cg, bv[g]);
        else { const float4 b = load4(bias + cg); bv[g][0] = b.x; bv[g][1] = b.y; bv[g][2] = b.z; bv[g][3] = b.w; }
      } else {
#pragma unroll
        for (int e = 0; e < GW; ++e) bv[g][e] = 0.f;
      }
    }
    auto fetch = [&](int idx, Pre& p) {
      const int g = idx / MT, mt = idx % MT;
      const int64_t o = (row0 + mt * 16) * ldc + col0 + g * GSTEP;
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
    Pre pre[NI];
#pragma unroll
    for (int i = 0; i < PD; ++i) fetch(i, pre[i]);
    hook();
    float cs[GW];
    float ssq = 0.f;   // PLAIN_ONLY (weight gradients): squares of what this lane stores, for the clip's norm (cq.sumsq)
#pragma unroll
    for (int idx = 0; idx < NI; ++idx) {
      const int g = idx / MT, mt = idx % MT;
      const int64_t cg = col0 + g * GSTEP;
      if (mt == 0) {
#pragma unroll
        for (int e = 0; e < GW; ++e) cs[e] = 0.f;
      }
      if constexpr (PD != 0) {
        if (idx + PD < NI) fetch(idx + PD, pre[idx + PD]);
      }
      {
        const int64_t o = (row0 + mt * 16) * ldc + cg;
        float v[GW];
        if constexpr (PAIR) {
          const f32x4 t0 = pp_acc_read<ACC_AGPR>(acc[2 * g][mt]), t1 = pp_acc_read<ACC_AGPR>(acc[2 * g + 1][mt]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
        } else {
          const f32x4 t0 = pp_acc_read<ACC_AGPR>(acc[g][mt]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = t0[e];
        }
#pragma unroll
        for (int e = 0; e < GW; ++e) v[e] += bv[g][e];
        float hv[GW];
        if constexpr (HSRC != 0) {
          if constexpr (PAIR) unpack8(pre[idx].h, hv);
          else {
            hv[0] = __uint_as_float(pre[idx].h.x << 16); hv[1] = __uint_as_float(pre[idx].h.x & 0xffff0000u);
            hv[2] = __uint_as_float(pre[idx].h.y << 16); hv[3] = __uint_as_float(pre[idx].h.y & 0xffff0000u);
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
          v[0] += pre[idx].f0.x; v[1] += pre[idx].f0.y; v[2] += pre[idx].f0.z; v[3] += pre[idx].f0.w;
          if constexpr (PAIR) { v[4] += pre[idx].f1.x; v[5] += pre[idx].f1.y; v[6] += pre[idx].f1.z; v[7] += pre[idx].f1.w; }
        }
        if constexpr (FSRC == 2 && !PAIR) {
          v[0] += beta * pre[idx].f0.x; v[1] += beta * pre[idx].f0.y; v[2] += beta * pre[idx].f0.z; v[3] += beta * pre[idx].f0.w;
        }
        if constexpr (PAIR) store8(C + o, v);
        else store4(C + o, make_float4(v[0], v[1], v[2], v[3]));
        if constexpr (PLAIN_ONLY && !PAIR && SUMSQ) ssq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (colsum) {
#pragma unroll
          for (int e = 0; e < GW; ++e) cs[e] += v[e];
        }
      }
      if (mt == MT - 1 && colsum) {
        // fold the 16 rows of a lane group; the wave's NT*16 column sums go through a wave-private LDS strip
#pragma unroll
        for (int e = 0; e < GW; ++e) {
          float s = cs[e];
          s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
          if (li == 0) scr[g * GSTEP + GW * q4 + e] = s;
        }
      }
    }
    if constexpr (PLAIN_ONLY && !PAIR && SUMSQ) {
      if (cq.sumsq) {   // one float atomic per wave and tile, spread over 16 slots by (tile column, tile row)
        ssq = wave_sum(ssq);
        if (lane == 0) atomicAdd(cq.sumsq + (int)(((colw >> 5) + (row0 >> 7)) & 15), ssq);
      }
    }
    if (colsum) {
      // one atomic instruction of contiguous floats per wave and tile (full-rate shape of MI355X_MICROARCH "Global float atomics")
      __builtin_amdgcn_wave_barrier();
      for (int c = lane; c < NT * 16; c += 64) atomicAdd(colsum + colw + c, scr[c]);
      __builtin_amdgcn_wave_barrier();
    }
  }

  static __device__ __forceinline__ void generic(f32x4 (&acc)[NT][MT], const PPEpiProb& cq, int64_t row0, int64_t colw, int lane) {
    const int q4 = lane >> 4;
    GemmEpi e;
    e.bias = cq.bias; e.mode = cq.mode; e.aux = cq.aux;
    e.res1 = reinterpret_cast<const float*>(cq.res1); e.res2 = cq.res2; e.res1_bf16 = cq.res1_bf16;
    e.beta = cq.beta; e.ldc = cq.ldc; e.colsum = nullptr;
    CT* C = reinterpret_cast<CT*>(cq.C);
    const int64_t col0 = colw + (PAIR ? 8 * q4 : 4 * q4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if constexpr (PAIR) {
          float v[8];
          const f32x4 t0 = pp_acc_read<ACC_AGPR>(acc[2 * g][mt]), t1 = pp_acc_read<ACC_AGPR>(acc[2 * g + 1][mt]);
#pragma unroll
          for (int x = 0; x < 4; ++x) { v[x] = t0[x]; v[4 + x] = t1[x]; }
          epilogue_store8<CT>(e, C, row0 + mt * 16, col0 + g * GSTEP, v);
        } else {
          const f32x4 t0 = pp_acc_read<ACC_AGPR>(acc[g][mt]);
          epilogue_store4<CT, true>(e, C, row0 + mt * 16, col0 + g * GSTEP, make_float4(t0[0], t0[1], t0[2], t0[3]));
        }
      }
  }

  struct NoHook { __device__ __forceinline__ void operator()() const {} };
  static __device__ __forceinline__ void run(f32x4 (&acc)[NT][MT], const PPEpiProb& cq, int64_t row0, int64_t colw, int lane, float* scr) {
    run(acc, cq, row0, colw, lane, scr, NoHook{});
  }
  template <typename Hook>
  static __device__ __forceinline__ void run(f32x4 (&acc)[NT][MT], const PPEpiProb& cq, int64_t row0, int64_t colw, int lane, float* scr, Hook&& hook) {
    const int mode = cq.mode;
    const bool r1 = cq.res1 != nullptr, r1h = r1 && cq.res1_bf16, r2 = cq.res2 != nullptr;
    const bool bt = cq.beta != 0.f;
    if constexpr (PLAIN_ONLY) {
      if (!PAIR && bt) fast<MAFED_EPI_NONE, 0, (PAIR ? 0 : 2)>(acc, cq, row0, colw, lane, scr, hook);
      else fast<MAFED_EPI_NONE, 0, 0>(acc, cq, row0, colw, lane, scr, hook);
      return;
    }
    if (mode == MAFED_EPI_NONE && !r1 && !r2 && !bt) fast<MAFED_EPI_NONE, 0, 0>(acc, cq, row0, colw, lane, scr, hook);
    else if (mode == MAFED_EPI_GELU && !r1 && !r2 && !bt) fast<MAFED_EPI_GELU, 0, 0>(acc, cq, row0, colw, lane, scr, hook);
    else if (PAIR && mode == MAFED_EPI_GELU_BWD && !r1 && !r2 && !bt) fast<(PAIR ? MAFED_EPI_GELU_BWD : MAFED_EPI_NONE), (PAIR ? 1 : 0), 0>(acc, cq, row0, colw, lane, scr, hook);
    else if (mode == MAFED_EPI_NONE && r1h && r2 && !bt) fast<MAFED_EPI_NONE, 2, 1>(acc, cq, row0, colw, lane, scr, hook);
    else if (!PAIR && mode == MAFED_EPI_NONE && !r1 && !r2 && bt) fast<MAFED_EPI_NONE, 0, (PAIR ? 0 : 2)>(acc, cq, row0, colw, lane, scr, hook);
    else { hook(); generic(acc, cq, row0, colw, lane); }   // any other combination: the in-place operand loads of gemm_epilogue.h (no prefetch, no fused column sums)
  }
};

}  // namespace mafed
