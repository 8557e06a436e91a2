// One-row-per-sample decode step of a GPT-NeoX layer in three launches (reference call:
// mafed/model/vqa_cont_learner.py:260-277 -> HF greedy search over vl_pythia.py's stack):
//
//   A  decode_ln_qkv_fc1   both LayerNorms of the parallel-residual block folded into the prologue of ONE skinny product over the
//                          concatenated output columns [ q | k | v | 4h ]: qkv row -> K/V cache, gelu(fc1) row -> a
//   B  attention           (attn_ref.hip: attn_decode_flat_kernel)
//   C  decode_out          x' = x + b_dense + b_fc2 + [ ao | a ] . [ W_dense | W_fc2 ]^T: one product over the concatenated K
//
// instead of LayerNorm, QKV, attention, dense, fc1, fc2 (six launches of 3 - 17 us for 29.5 MB of weights and 37.7 MB of K/V per layer
// at 410M / B = 32).  What bounds such a product is not the weight stream alone but what every block reads BESIDE its weight slab: the
// 4-column strips of the round-2 fc2 kernel re-read the whole 256 KB activation row block per block (64 MB of L2 traffic for 8 MB of
// weights, 14.5 us).  Here
//   A: a block owns 32 output columns and the full K = h; its eight waves split K, and each wave's slice of x is loaded ONCE in exactly
//      the MFMA operand layout (fp32, 8 consecutive k per lane), the row statistics are folded over lanes / waves through LDS, and the
//      normalised bf16 operand never leaves registers.  7h / 32 = 224 blocks, 128 KB of x + 64 KB of W each.
//   C: the K = 5h reduction is split over P blocks per 32-column group (h/32 x P ~ 256 blocks; 40 KB of W + 40 KB of activations
//      each); partial tiles go to a workspace and the LAST block of a group to arrive (device-scope counter) adds them in slice order,
//      so the result does not depend on arrival order (no floating-point atomics), adds residual and biases and stores x'.
// Weights are bf16 [N, K] row-major (the optimizer's shadow copy), activations bf16, the residual stream fp32.
#include "common.h"

namespace mafed {

long long* g_decode_trace = nullptr;   // tools: stamps of the next launches (mafed_decode_set_trace)

struct DecodeAArgs {
  const float* x;        // [M, h] fp32 residual stream
  int M, h;
  float eps;
  const float *g1, *b1, *g2, *b2;   // input_layernorm / post_attention_layernorm
  const bf16_t* wqkv;    // [3h, h]
  const float* bqkv;
  bf16_t* qkv_out;       // row m at qkv_out + m * qkv_ld
  int64_t qkv_ld;
  const bf16_t* w1;      // [n1, h]
  const float* bfc1;
  bf16_t* a_out;         // [M, n1]
  int nqkv, n1;          // 3h, intermediate size
  long long* trace;      // tools: optional [grid][8] wall-clock stamps (mafed_decode_set_trace)
};

// lanes (i, g) = (lane & 15, lane >> 4); MFMA 16x16x32: A = W rows n0 + i, k = 8g .. 8g+7; B = x rows (m = i); D[n = 4g + r][m = i]
// NP: pairs of 16-column strips per block (1 for the layer: 7h / 32 blocks fill the chip; 4 for the LM head: the x slab is read once
// per 128 vocabulary columns).  Segment 1 (qkv / head: LN1 operands, bias optional, plain store) then segment 2 (fc1: LN2, GELU).
template <int MT, int KS, int NP>
__global__ __launch_bounds__(512) void decode_ln_qkv_fc1_kernel(DecodeAArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int NS = 2 * NP;                                           // strips per block
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                    // [8 waves][NS strips][MT][64]
  float* part = reinterpret_cast<float*>(smem_raw + 8 * NS * MT * 64 * 16);  // [2 passes][8 waves][MT * 16]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h;
  const int nqb = a.nqkv / (16 * NS);
  const bool is_qkv = (int)blockIdx.x < nqb;
  const int n0 = (is_qkv ? (int)blockIdx.x : (int)blockIdx.x - nqb) * (16 * NS);
  const bf16_t* W = is_qkv ? a.wqkv : a.w1;
  const float* gam = is_qkv ? a.g1 : a.g2;
  const float* bet = is_qkv ? a.b1 : a.b2;
  const int kb = wave * (KS * 32) + 8 * g;   // this lane's first k
  // weight slab first: it does not depend on anything
  bf16x8 wf[NS][KS];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int u = 0; u < KS; ++u) wf[s][u] = *reinterpret_cast<const bf16x8*>(W + (int64_t)(n0 + 16 * s + i) * h + kb + 32 * u);
  float xv[MT][KS][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + i;
    const float* xr = a.x + (int64_t)(row < a.M ? row : 0) * h + kb;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const float4 p = load4(xr + 32 * u), q = load4(xr + 32 * u + 4);
      xv[mt][u][0] = p.x; xv[mt][u][1] = p.y; xv[mt][u][2] = p.z; xv[mt][u][3] = p.w;
      xv[mt][u][4] = q.x; xv[mt][u][5] = q.y; xv[mt][u][6] = q.z; xv[mt][u][7] = q.w;
    }
  }
  // the affine parameters of this lane's k: up front when the registers allow, else fetched (L2 hits) beside the normalisation
  constexpr bool LATE_GB = KS == 8 || MT * KS >= 16 || NP > 1;
  float gv[LATE_GB ? 1 : KS][8], bv[LATE_GB ? 1 : KS][8];
  auto load_gb = [&](int u, float (&gd)[8], float (&bd)[8]) {
    const float4 p = load4(gam + kb + 32 * u), q = load4(gam + kb + 32 * u + 4), r = load4(bet + kb + 32 * u), t = load4(bet + kb + 32 * u + 4);
    gd[0] = p.x; gd[1] = p.y; gd[2] = p.z; gd[3] = p.w; gd[4] = q.x; gd[5] = q.y; gd[6] = q.z; gd[7] = q.w;
    bd[0] = r.x; bd[1] = r.y; bd[2] = r.z; bd[3] = r.w; bd[4] = t.x; bd[5] = t.y; bd[6] = t.z; bd[7] = t.w;
  };
  if constexpr (!LATE_GB) {
#pragma unroll
    for (int u = 0; u < KS; ++u) load_gb(u, gv[u], bv[u]);
  }
  // epilogue operands: wave w stores tiles w, w + 8, ... of the block's NS x MT tiles (strip = tile % NS, row block = tile / NS)
  constexpr int TPW = (NS * MT + 7) / 8;
  const float* bias = is_qkv ? a.bqkv : a.bfc1;
  float4 bia[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = wave + 8 * j, es = tile % NS;
    bia[j] = bias ? load4(bias + n0 + 16 * es + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);   // (uniform; every wave: clamped, not predicated)
  }
  // row statistics: mean, then the centred second moment (the two-pass form of layernorm_fwd_kernel), folded over g, the waves, LDS
  float mean[MT], rstd[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < KS; ++u) s += ((xv[mt][u][0] + xv[mt][u][1]) + (xv[mt][u][2] + xv[mt][u][3])) + ((xv[mt][u][4] + xv[mt][u][5]) + (xv[mt][u][6] + xv[mt][u][7]));
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (g == 0) part[wave * (MT * 16) + mt * 16 + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += part[w * (MT * 16) + mt * 16 + i];
    mean[mt] = s / (float)h;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < KS; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xv[mt][u][e] - mean[mt];
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (g == 0) part[(8 + wave) * (MT * 16) + mt * 16 + i] = q;
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) q += part[(8 + w) * (MT * 16) + mt * 16 + i];
    rstd[mt] = 1.0f / sqrtf(q / (float)h + a.eps);
  }
  bf16x8 xf[MT][KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    float gl[8], bl[8];
    if constexpr (LATE_GB) load_gb(u, gl, bl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ge = LATE_GB ? gl[e] : gv[LATE_GB ? 0 : u][e], be = LATE_GB ? bl[e] : bv[LATE_GB ? 0 : u][e];
        xf[mt][u][e] = (__bf16)(((xv[mt][u][e] - mean[mt]) * rstd[mt]) * ge + be);
      }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < KS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][u], xf[mt][u], acc, 0, 0, 0);
      red[((wave * NS + s) * MT + mt) * 64 + lane] = acc;
    }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tile = wave + 8 * j;
    if (tile < NS * MT) {
      const int es = tile % NS, emt = tile / NS;
      f32x4 v = red[((0 * NS + es) * MT + emt) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) v += red[((w * NS + es) * MT + emt) * 64 + lane];
      const int m = emt * 16 + i;
      if (m < a.M) {
        float4 o = make_float4(v[0] + bia[j].x, v[1] + bia[j].y, v[2] + bia[j].z, v[3] + bia[j].w);
        const int n = n0 + 16 * es + 4 * g;
        if (is_qkv) {
          store4(a.qkv_out + (int64_t)m * a.qkv_ld + n, o);
        } else {
          o = make_float4(gelu_erf_fast(o.x), gelu_erf_fast(o.y), gelu_erf_fast(o.z), gelu_erf_fast(o.w));
          store4(a.a_out + (int64_t)m * a.n1 + n, o);
        }
      }
    }
  }
}

// The same launch with every global load a full-line one (round 4, second form).  The register-direct form above asks, per wave
// instruction, for 64 bytes of each of 16 rows (the MFMA operand layout: lane (i, g) = row i, 16-byte chunk g); measured, a CU takes
// such loads in at 25 - 30 GB/s against ~65 GB/s for 1 KB-contiguous ones, and with the 128 KB fp32 x slab beside the 64 KB weight slab
// that intake -- not HBM -- bounded the kernel (11.8 us).  Here a wave owns whole rows of x (statistics inside the wave, no exchange),
// writes the normalised bf16 row to LDS, the weight slab goes to LDS chunk-by-chunk in address order, and the MFMA operands come from
// LDS (row stride h + 8: the 16 rows of an operand read fall on distinct bank groups).  LDS: (16 MT + 32) x (h + 8) x 2 bytes
// (132 KB at h = 1024, MT = 2); the cross-wave partial tiles re-use the weight slab's space.
template <int MT, int KS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void decode_ln_qkv_fc1_lds_kernel(DecodeAArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int H = 256 * KS, LD = H + 8, MP = 16 * MT;    // LD: padded row length (elements)
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem_raw);                           // [MP][LD]
  bf16_t* Ws = Xs + MP * LD;                                                 // [32][LD]
  f32x4* red = reinterpret_cast<f32x4*>(Ws);                                  // [8 waves][2][MT][64] after the operand reads (<= 32 KB <= the slab)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: M0 and the slab's scalar operands depend on it)
  const int i = lane & 15, g = lane >> 4;
  const int nqb = a.nqkv / 32;
  const bool is_qkv = (int)blockIdx.x < nqb;
  const int n0 = (is_qkv ? (int)blockIdx.x : (int)blockIdx.x - nqb) * 32;
  long long* tr = a.trace ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
  const bf16_t* W = (is_qkv ? a.wqkv : a.w1) + (int64_t)n0 * H;
  const float* gam = is_qkv ? a.g1 : a.g2;
  const float* bet = is_qkv ? a.b1 : a.b2;
  const float* bias = is_qkv ? a.bqkv : a.bfc1;
  const int es = wave & 1, emt = wave >> 1;   // waves 0 .. 2 MT - 1 store tile (strip es, row block emt)
  // One statement issues every global load of the wave -- its RPW rows of x (wave w owns rows w, w + 8, ...: a lane takes columns
  // 4 (lane + 64 jj)) and its eight 1 KB pieces of the weight slab, which go global -> LDS directly (global_load_lds_dwordx4: 64 lanes x
  // 16 bytes land at M0 + 16 lane, a contiguous KB of a row in its padded place, no VGPRs) -- and waits for the rows only (vmcnt(8): the
  // eight pieces, issued last, stay in flight under the row statistics).  Written as asm because the compiler sinks plain loads to their
  // uses (one row at a time: RPW dependent round trips) and because its own vmcnt arithmetic does not know the LDS-DMA pieces.
  // M0 saved / restored inside the statement (cdna_hip_programming 5.7).  Slab piece pc = w + 8 j: source W + 1024 pc, row pc / 2,
  // half pc % 2 -> destination advances by 4 padded rows per j.
  static_assert(KS == 4, "the LDS form is written for h = 1024 (2 KB weight rows = two 1 KB pieces)");
  constexpr int RPW = MP / 8;   // rows per wave (2 MT)
  f32x4 xv[RPW][KS];
  {
    const uint32_t ws0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Ws;
    const uint32_t dst0 = ws0 + (uint32_t)((wave >> 1) * (LD * 2) + (wave & 1) * 1024);
    uint32_t voff = (uint32_t)(wave * 1024 + lane * 16);
    const bf16_t* wb = W;
    const float* xr[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = wave + 8 * r;
      xr[r] = a.x + (int64_t)(row < a.M ? row : 0) * H + 4 * lane;
    }
    unsigned keep;
#define MAFED_DEC_DMA8                                                                                         \
  "s_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"                                                                        \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t" \
  "global_load_lds_dwordx4 %[voff], %[wb]\n\t"                                                                 \
  "s_waitcnt vmcnt(8)\n\ts_mov_b32 m0, %[keep]"
    if constexpr (RPW == 2) {
      asm volatile("s_mov_b32 %[keep], m0\n\t"
                   "global_load_dwordx4 %[x00], %[a0], off\n\tglobal_load_dwordx4 %[x01], %[a0], off offset:1024\n\t"
                   "global_load_dwordx4 %[x02], %[a0], off offset:2048\n\tglobal_load_dwordx4 %[x03], %[a0], off offset:3072\n\t"
                   "global_load_dwordx4 %[x10], %[a1], off\n\tglobal_load_dwordx4 %[x11], %[a1], off offset:1024\n\t"
                   "global_load_dwordx4 %[x12], %[a1], off offset:2048\n\tglobal_load_dwordx4 %[x13], %[a1], off offset:3072\n\t" MAFED_DEC_DMA8
                   : [keep] "=&s"(keep), [voff] "+v"(voff), [x00] "=&v"(xv[0][0]), [x01] "=&v"(xv[0][1]), [x02] "=&v"(xv[0][2]), [x03] "=&v"(xv[0][3]),
                     [x10] "=&v"(xv[1][0]), [x11] "=&v"(xv[1][1]), [x12] "=&v"(xv[1][2]), [x13] "=&v"(xv[1][3])
                   : [a0] "v"(xr[0]), [a1] "v"(xr[1]), [wb] "s"(wb), [dst] "s"(dst0), [dstep] "n"(4 * LD * 2)
                   : "memory", "scc");
    } else {
      static_assert(RPW == 4, "MT <= 2");
      asm volatile("s_mov_b32 %[keep], m0\n\t"
                   "global_load_dwordx4 %[x00], %[a0], off\n\tglobal_load_dwordx4 %[x01], %[a0], off offset:1024\n\t"
                   "global_load_dwordx4 %[x02], %[a0], off offset:2048\n\tglobal_load_dwordx4 %[x03], %[a0], off offset:3072\n\t"
                   "global_load_dwordx4 %[x10], %[a1], off\n\tglobal_load_dwordx4 %[x11], %[a1], off offset:1024\n\t"
                   "global_load_dwordx4 %[x12], %[a1], off offset:2048\n\tglobal_load_dwordx4 %[x13], %[a1], off offset:3072\n\t"
                   "global_load_dwordx4 %[x20], %[a2], off\n\tglobal_load_dwordx4 %[x21], %[a2], off offset:1024\n\t"
                   "global_load_dwordx4 %[x22], %[a2], off offset:2048\n\tglobal_load_dwordx4 %[x23], %[a2], off offset:3072\n\t"
                   "global_load_dwordx4 %[x30], %[a3], off\n\tglobal_load_dwordx4 %[x31], %[a3], off offset:1024\n\t"
                   "global_load_dwordx4 %[x32], %[a3], off offset:2048\n\tglobal_load_dwordx4 %[x33], %[a3], off offset:3072\n\t" MAFED_DEC_DMA8
                   : [keep] "=&s"(keep), [voff] "+v"(voff), [x00] "=&v"(xv[0][0]), [x01] "=&v"(xv[0][1]), [x02] "=&v"(xv[0][2]), [x03] "=&v"(xv[0][3]),
                     [x10] "=&v"(xv[1][0]), [x11] "=&v"(xv[1][1]), [x12] "=&v"(xv[1][2]), [x13] "=&v"(xv[1][3]),
                     [x20] "=&v"(xv[2][0]), [x21] "=&v"(xv[2][1]), [x22] "=&v"(xv[2][2]), [x23] "=&v"(xv[2][3]),
                     [x30] "=&v"(xv[3][0]), [x31] "=&v"(xv[3][1]), [x32] "=&v"(xv[3][2]), [x33] "=&v"(xv[3][3])
                   : [a0] "v"(xr[0]), [a1] "v"(xr[1]), [a2] "v"(xr[2]), [a3] "v"(xr[3]), [wb] "s"(wb), [dst] "s"(dst0), [dstep] "n"(4 * LD * 2)
                   : "memory", "scc");
    }
#undef MAFED_DEC_DMA8
  }
  if (tr && tid == 0) tr[1] = wall_clock64();   // rows have arrived
  // affine parameters and the bias: loads the compiler counts (any wait it inserts for them -- they are the youngest in the queue -- also
  // waits for the slab, which the barrier below needs anyway); requested now, used behind the statistics
  float4 gv[KS], bv[KS];
#pragma unroll
  for (int jj = 0; jj < KS; ++jj) {
    gv[jj] = load4(gam + 4 * (lane + 64 * jj));
    bv[jj] = load4(bet + 4 * (lane + 64 * jj));
  }
  const float4 bia = bias ? load4(bias + n0 + 16 * es + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);   // (uniform; NULL for the LM head)
  // row statistics inside the wave, under the slab's flight; the RPW rows are reduced side by side (one row at a time the twelve
  // dependent cross-lane steps of its two sums were 0.55 us per row: 2.2 of the kernel's 8.8 us, tools/decode_ab_trace.py)
  float mean[RPW], rstd[RPW], part[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float sm = 0.f;
#pragma unroll
    for (int jj = 0; jj < KS; ++jj) sm += (xv[r][jj][0] + xv[r][jj][1]) + (xv[r][jj][2] + xv[r][jj][3]);
    part[r] = sm;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < RPW; ++r) part[r] += __shfl_xor(part[r], o, 64);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    mean[r] = part[r] / (float)H;
    float q = 0.f;
#pragma unroll
    for (int jj = 0; jj < KS; ++jj) {
      const float d0 = xv[r][jj][0] - mean[r], d1 = xv[r][jj][1] - mean[r], d2 = xv[r][jj][2] - mean[r], d3 = xv[r][jj][3] - mean[r];
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    part[r] = q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < RPW; ++r) part[r] += __shfl_xor(part[r], o, 64);
#pragma unroll
  for (int r = 0; r < RPW; ++r) rstd[r] = 1.0f / sqrtf(part[r] / (float)H + a.eps);
  if (tr && tid == 0) tr[2] = wall_clock64();   // statistics done
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = wave + 8 * r;
#pragma unroll
    for (int jj = 0; jj < KS; ++jj) {
      const f32x4 v = xv[r][jj];
      store4(Xs + row * LD + 4 * (lane + 64 * jj),
             make_float4((v[0] - mean[r]) * rstd[r] * gv[jj].x + bv[jj].x, (v[1] - mean[r]) * rstd[r] * gv[jj].y + bv[jj].y,
                         (v[2] - mean[r]) * rstd[r] * gv[jj].z + bv[jj].z, (v[3] - mean[r]) * rstd[r] * gv[jj].w + bv[jj].w));
    }
  }
  if (tr && tid == 0) tr[3] = wall_clock64();   // normalised rows written
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's slab pieces have landed (the compiler does not count them)
  if (tr && tid == 0) tr[4] = wall_clock64();   // slab landed
  __syncthreads();
  if (tr && tid == 0) tr[5] = wall_clock64();
  // wave w multiplies its K slice [w * 32 KS, (w + 1) * 32 KS): both strips x every row block
  const int kb = wave * (KS * 32) + 8 * g;
  f32x4 acc[2][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    acc[0][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[1][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(Ws + i * LD + kb + 32 * u);
    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(Ws + (16 + i) * LD + kb + 32 * u);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xs + (mt * 16 + i) * LD + kb + 32 * u);
      acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf, acc[0][mt], 0, 0, 0);
      acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf, acc[1][mt], 0, 0, 0);
    }
  }
  if (tr && tid == 0) tr[6] = wall_clock64();   // MFMAs issued
  __syncthreads();   // every wave has read its operands: the slab's space becomes the partial tiles
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[((wave * 2 + s) * MT + mt) * 64 + lane] = acc[s][mt];
  __syncthreads();
  if (emt < MT) {
    f32x4 v = red[((0 * 2 + es) * MT + emt) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) v += red[((w * 2 + es) * MT + emt) * 64 + lane];
    const int m = emt * 16 + i;
    if (m < a.M) {
      float4 o = make_float4(v[0] + bia.x, v[1] + bia.y, v[2] + bia.z, v[3] + bia.w);
      const int n = n0 + 16 * es + 4 * g;
      if (is_qkv) {
        store4(a.qkv_out + (int64_t)m * a.qkv_ld + n, o);
      } else {
        o = make_float4(gelu_erf_fast(o.x), gelu_erf_fast(o.y), gelu_erf_fast(o.z), gelu_erf_fast(o.w));
        store4(a.a_out + (int64_t)m * a.n1 + n, o);
      }
    }
  }
  if (tr && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tr[7] = wall_clock64(); }
}

// LM head of a decode step as a persistent kernel (round 4): one 512-thread block per CU normalises the M rows once (final LayerNorm, bf16
// rows in LDS) and then walks 16-column strips of the vocabulary, the next strip's weight slab (16 rows x 2 KB, global -> LDS) in flight
// under the current strip's MFMAs.  The one-slab-per-block form above needs six rounds of blocks for V = 50304, each paying its own
// HBM round trip: 48 us for 103 MB, like LayerNorm + the skinny kernel; this one streams.  h = 1024, M <= 32.
template <int MT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void decode_head_kernel(DecodeAArgs a) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int KS = 4, H = 1024, LD = H + 8, MP = 16 * MT;
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem_raw);                  // [MP][LD]
  bf16_t* Wb = Xs + MP * LD;                                        // [2][16][LD]
  f32x4* red = reinterpret_cast<f32x4*>(Wb + 2 * 16 * LD);           // [8][MT][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nstrips = a.nqkv / 16;
  const uint32_t wb0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Wb;
  const uint32_t dst_w = (uint32_t)((wave >> 1) * (LD * 2) + (wave & 1) * 1024);   // this wave's first piece inside a slab buffer
  // slab of strip s -> buffer b: four 1 KB pieces per wave (piece w + 8 j: row (w + 8 j) / 2, half (w + 8 j) % 2)
  auto dma_slab = [&](int s, int b) {
    const bf16_t* src = a.wqkv + (int64_t)s * 16 * H;
    uint32_t voff = (uint32_t)(wave * 1024 + lane * 16);
    const uint32_t dst = wb0 + (uint32_t)(b * 16 * LD * 2) + dst_w;
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %[voff], %[src]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t"
                 "global_load_lds_dwordx4 %[voff], %[src]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t"
                 "global_load_lds_dwordx4 %[voff], %[src]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t"
                 "global_load_lds_dwordx4 %[voff], %[src]\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep), [voff] "+v"(voff)
                 : [src] "s"(src), [dst] "s"(dst), [dstep] "n"(4 * LD * 2)
                 : "memory", "scc");
  };
  int s = blockIdx.x;
  if (s < nstrips) dma_slab(s, 0);
  // the block's rows: wave w owns rows w, w + 8, ...; statistics inside the wave, side by side
  constexpr int RPW = MP / 8;
  {
    float4 xv[RPW][KS];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int row = wave + 8 * r;
      const float* xr = a.x + (int64_t)(row < a.M ? row : 0) * H;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) xv[r][jj] = load4(xr + 4 * (lane + 64 * jj));
    }
    float4 gv[KS], bv[KS];
#pragma unroll
    for (int jj = 0; jj < KS; ++jj) {
      gv[jj] = load4(a.g1 + 4 * (lane + 64 * jj));
      bv[jj] = load4(a.b1 + 4 * (lane + 64 * jj));
    }
    float mean[RPW], rstd[RPW], part[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float sm = 0.f;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) sm += (xv[r][jj].x + xv[r][jj].y) + (xv[r][jj].z + xv[r][jj].w);
      part[r] = sm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < RPW; ++r) part[r] += __shfl_xor(part[r], o, 64);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      mean[r] = part[r] / (float)H;
      float q = 0.f;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) {
        const float d0 = xv[r][jj].x - mean[r], d1 = xv[r][jj].y - mean[r], d2 = xv[r][jj].z - mean[r], d3 = xv[r][jj].w - mean[r];
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
      part[r] = q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int r = 0; r < RPW; ++r) part[r] += __shfl_xor(part[r], o, 64);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      rstd[r] = 1.0f / sqrtf(part[r] / (float)H + a.eps);
      const int row = wave + 8 * r;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) {
        const float4 v = xv[r][jj];
        store4(Xs + row * LD + 4 * (lane + 64 * jj),
               make_float4((v.x - mean[r]) * rstd[r] * gv[jj].x + bv[jj].x, (v.y - mean[r]) * rstd[r] * gv[jj].y + bv[jj].y,
                           (v.z - mean[r]) * rstd[r] * gv[jj].z + bv[jj].z, (v.w - mean[r]) * rstd[r] * gv[jj].w + bv[jj].w));
      }
    }
  }
  const int kb = wave * (KS * 32) + 8 * g;
  int buf = 0;
  for (; s < nstrips; s += gridDim.x, buf ^= 1) {
    const int sn = s + gridDim.x;
    if (sn < nstrips) {
      dma_slab(sn, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything older than the next slab's four pieces: this strip's slab (and the last stores)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();   // slab (and, first trip, the normalised rows) visible to every wave
    const bf16_t* Ws = Wb + buf * 16 * LD;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + i * LD + kb + 32 * u);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xs + (mt * 16 + i) * LD + kb + 32 * u);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    if (wave < MT) {   // wave emt folds row block emt over the eight K slices (fixed order) and stores 16 x 16 logits
      f32x4 v = red[(0 * MT + wave) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) v += red[(w * MT + wave) * 64 + lane];
      const int m = wave * 16 + i;
      if (m < a.M) store4(a.qkv_out + (int64_t)m * a.qkv_ld + (int64_t)s * 16 + 4 * g, make_float4(v[0], v[1], v[2], v[3]));
    }
    __syncthreads();   // red and the slab buffer of this trip are free again
  }
}

struct DecodeCArgs {
  const float* x;          // [M, h] residual in
  float* x_out;            // [M, h] (may alias x)
  int M, h, n1;            // n1 = intermediate size (K of the second operand)
  const bf16_t* ao;        // [M, h]
  const bf16_t* act;       // [M, n1]
  const bf16_t* wd;        // [h, h]
  const bf16_t* w2;        // [h, n1]
  const float *bd, *b2;
  int P, ksl;              // K slices per column group, k-steps (of 32) per slice
  float* ws;               // [P][MT * 16][h] partial tiles
  unsigned* counters;      // [h / 32], zero between launches
  long long* trace;        // tools: optional [grid][8] wall-clock stamps per workgroup (mafed_decode_set_trace)
};

// Everything behind a block's K loop: the four k lanes' tiles are folded through LDS, the block's partial tile goes to the workspace,
// and the last block of the column group to arrive adds the P partial tiles in slice order, the residual and the biases.
// Partial tiles cross XCDs (each has its own L2): they are written and read with agent-scope accesses (sc1: written through / read
// around the L2), not with plain stores behind a device-wide fence -- __threadfence() is buffer_wbl2 + buffer_inv of the whole L2 by
// every wave of every block (measured: 50 us for this kernel instead of 13; fences by one thread per block: 16).  Order: a wave's
// stores have left (vmcnt(0)) before the block barrier, the arrival counter is bumped after it.
// The last block's read-back of the P partial tiles: ONE statement issues a 16-byte agent-scope load per slice (clamped pointer for the
// slices past P) and waits once -- the dword-at-a-time form (four loads per slice: 48 wave instructions of 16 scattered lines each) took
// 5.6 us of the kernel's 12 (tools/decode_out_trace.py).
#define MAFED_LD_SC1(k) "global_load_dwordx4 %" #k ", %[p" #k "], off sc1\n\t"
__device__ __forceinline__ void ld12_sc1(const float* const (&p)[12], f32x4 (&o)[12]) {
  asm volatile(MAFED_LD_SC1(0) MAFED_LD_SC1(1) MAFED_LD_SC1(2) MAFED_LD_SC1(3) MAFED_LD_SC1(4) MAFED_LD_SC1(5) MAFED_LD_SC1(6) MAFED_LD_SC1(7)
               MAFED_LD_SC1(8) MAFED_LD_SC1(9) MAFED_LD_SC1(10) MAFED_LD_SC1(11) "s_waitcnt vmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]), "=&v"(o[8]), "=&v"(o[9]),
                 "=&v"(o[10]), "=&v"(o[11])
               : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [p4] "v"(p[4]), [p5] "v"(p[5]), [p6] "v"(p[6]), [p7] "v"(p[7]),
                 [p8] "v"(p[8]), [p9] "v"(p[9]), [p10] "v"(p[10]), [p11] "v"(p[11])
               : "memory");
}
#undef MAFED_LD_SC1
// P <= 12 slices in one batch (P = 10 at 410M), up to 24 in two
__device__ __forceinline__ f32x4 decode_out_sum_slices(const float* __restrict__ ws, int P, int64_t slice_stride, int64_t off) {
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int b0 = 0; b0 < P; b0 += 12) {   // (one trip for P <= 12)
    const float* ptr[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) ptr[k] = ws + (int64_t)(b0 + k < P ? b0 + k : 0) * slice_stride + off;
    f32x4 o[12];
    ld12_sc1(ptr, o);
#pragma unroll
    for (int k = 0; k < 12; ++k)
      if (b0 + k < P) v += o[k];   // slice order, whichever block happens to be last (0 + t0 + t1 + ...: the same association every time)
  }
  return v;
}

template <int MT>
__device__ __forceinline__ void decode_out_finish(const DecodeCArgs& a, const f32x4 (&acc)[MT], f32x4 (*red)[MT][64], int* s_last, int grp, int p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h, n0 = grp * 32;
  long long* tr = a.trace ? a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) red[wave][mt][lane] = acc[mt];
  __syncthreads();
  // wave w finishes tile (strip es = w & 1, row block emt = w >> 1): lane holds C[m = emt * 16 + i][n0 + 16 es + 4g .. + 3]
  const int es = wave & 1, emt = wave >> 1;
  const int m = emt * 16 + i, nn = n0 + 16 * es + 4 * g;
  const int Mp = MT * 16;
  // epilogue operands requested now (clamped, by every wave): they are there when the last block needs them
  const int mc = (emt < MT && m < a.M) ? m : 0;
  const float4 r = load4(a.x + (int64_t)mc * h + nn), c0 = load4(a.bd + nn), c1 = load4(a.b2 + nn);
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (emt < MT) {
    v = red[es][emt][lane];
#pragma unroll
    for (int qq = 1; qq < 4; ++qq) v += red[es + 2 * qq][emt][lane];
  }
  if (a.P > 1) {
    if (emt < MT) {
      float* wp = a.ws + ((int64_t)p * Mp + m) * h + nn;
      asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp), "v"(v) : "memory");   // one 16-byte agent-scope store
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (a.trace) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 3] = wall_clock64();
      const unsigned old = __hip_atomic_fetch_add(a.counters + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *s_last = old == (unsigned)(a.P - 1);
      if (a.trace) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 4] = wall_clock64();
      if (old == (unsigned)(a.P - 1)) __hip_atomic_store(a.counters + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    }
    __syncthreads();
    if (!*s_last) return;
    if (emt < MT) {
      const int64_t ss = (int64_t)Mp * h, off = (int64_t)m * h + nn;
      v = decode_out_sum_slices(a.ws, a.P, ss, off);
    }
  }
  if (emt < MT && m < a.M) {
    store4(a.x_out + (int64_t)m * h + nn, make_float4(r.x + (v[0] + c0.x + c1.x), r.y + (v[1] + c0.y + c1.y), r.z + (v[2] + c0.z + c1.z), r.w + (v[3] + c0.w + c1.w)));
  }
  if (tr && threadIdx.x == 0) tr[5] = wall_clock64();
}

// register-direct operands (any served shape): lane (i, g) loads its MFMA fragments straight from global memory
template <int MT>
__global__ __launch_bounds__(512) void decode_out_kernel(DecodeCArgs a) {
  __shared__ f32x4 red[8][MT][64];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h, n1 = a.n1;
  const int grp = blockIdx.x, p = blockIdx.y;
  const int n0 = grp * 32;
  const int s = wave & 1, q = wave >> 1;   // strip, k lane (4 per strip)
  const int ktot = (h + n1) / 32, kh = h / 32;
  const int k_lo = p * a.ksl, k_hi = min(k_lo + a.ksl, ktot);
  const int n = n0 + 16 * s + i;
  const bf16_t* wdr = a.wd + (int64_t)n * h + 8 * g;
  const bf16_t* w2r = a.w2 + (int64_t)n * n1 + 8 * g;
  const bf16_t *aor[MT], *acr[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + i;
    const int rc = row < a.M ? row : 0;
    aor[mt] = a.ao + (int64_t)rc * h + 8 * g;
    acr[mt] = a.act + (int64_t)rc * n1 + 8 * g;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int UN = 5;   // k-steps in flight per wave
  const bf16x8 zero = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));
  for (int kk = k_lo + q; kk < k_hi; kk += 4 * UN) {
    bf16x8 wf[UN], xf[UN][MT];
    bool in[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = kk + 4 * u;
      in[u] = k < k_hi;
      const int kc = in[u] ? k : k_lo;              // clamped (always a valid step of this slice), zeroed below
      const bool first = kc < kh;                   // dense operand / fc2 operand
      const int64_t off = first ? (int64_t)kc * 32 : (int64_t)(kc - kh) * 32;
      wf[u] = *reinterpret_cast<const bf16x8*>((first ? wdr : w2r) + off);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[u][mt] = *reinterpret_cast<const bf16x8*>((first ? aor[mt] : acr[mt]) + off);
    }
    __builtin_amdgcn_sched_barrier(0);   // every load of the trip is in flight before the first MFMA waits
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const bf16x8 w = in[u] ? wf[u] : zero;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xf[u][mt], acc[mt], 0, 0, 0);
    }
  }
  decode_out_finish<MT>(a, acc, red, &s_last, grp, p);
}

// The same product with full-line loads (h = 1024, n1 = 4096): a block owns a 512-deep K slice (16 k-steps; ten slices, none straddles the
// dense | fc2 boundary) of 32 columns; its weight slab (32 rows x 1 KB) and activation slab (16 MT rows x 1 KB) go global -> LDS as
// whole-KB row pieces (global_load_lds_dwordx4, one statement per wave: eight pieces, nothing in VGPRs), the MFMA operands come from
// LDS (padded rows: conflict-free).  66.5 KB of LDS: two blocks per CU, all 320 resident at once.
template <int MT>
__global__ __launch_bounds__(512) void decode_out_lds_kernel(DecodeCArgs a) {
  constexpr int KSL = 16, LDK = KSL * 32 + 8, MP = 16 * MT;
  __shared__ __align__(16) bf16_t Ws[32 * LDK];
  __shared__ __align__(16) bf16_t Xs[MP * LDK];
  __shared__ int s_last;
  f32x4 (*red)[MT][64] = reinterpret_cast<f32x4 (*)[MT][64]>(Ws);   // the k lanes' tiles re-use the weight slab's space (8 MT KB <= 32.5 KB)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h, n1 = a.n1;
  const int grp = blockIdx.x, p = blockIdx.y;
  const int n0 = grp * 32;
  const int k0 = p * (KSL * 32);                   // first k of the slice in the concatenated K
  const bool first = k0 < h;                       // dense operand / fc2 operand (block-uniform)
  const int ld = first ? h : n1, kofs = first ? k0 : k0 - h;
  const bf16_t* wsrc = (first ? a.wd : a.w2) + (int64_t)n0 * ld + kofs;
  const bf16_t* xsrc = (first ? a.ao : a.act) + kofs;
  long long* tr0 = a.trace ? a.trace + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
  if (tr0 && threadIdx.x == 0) tr0[0] = wall_clock64();
  {
    // wave w moves weight rows w, w + 8, w + 16, w + 24 and activation rows w (, w + 8 ...) -- rows past M re-read row M - 1 (never stored)
    const uint32_t wdst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Ws + (uint32_t)(wave * LDK * 2);
    const uint32_t xdst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Xs + (uint32_t)(wave * LDK * 2);
    uint32_t wv = (uint32_t)(wave * ld * 2 + lane * 16);
    const uint32_t wstep = (uint32_t)(8 * ld * 2);
    uint32_t xv[MP / 8];
#pragma unroll
    for (int j = 0; j < MP / 8; ++j) {
      const int row = wave + 8 * j;
      xv[j] = (uint32_t)((row < a.M ? row : a.M - 1) * ld * 2 + lane * 16);
    }
    unsigned keep;
    if constexpr (MT == 1) {
      asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[wdst]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_mov_b32 m0, %[xdst]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x0], %[xsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x1], %[xsrc]\n\t"
                   "s_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep), [wv] "+v"(wv)
                   : [wsrc] "s"(wsrc), [xsrc] "s"(xsrc), [wdst] "s"(wdst), [xdst] "s"(xdst), [wstep] "s"(wstep), [x0] "v"(xv[0]), [x1] "v"(xv[1]),
                     [dstep] "n"(8 * LDK * 2)
                   : "memory", "scc");
    } else {
      static_assert(MT == 2, "the LDS form serves M <= 32");
      asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[wdst]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\tv_add_u32 %[wv], %[wstep], %[wv]\n\t"
                   "global_load_lds_dwordx4 %[wv], %[wsrc]\n\ts_mov_b32 m0, %[xdst]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x0], %[xsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x1], %[xsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x2], %[xsrc]\n\ts_add_u32 m0, m0, %[dstep]\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %[x3], %[xsrc]\n\t"
                   "s_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %[keep]"
                   : [keep] "=&s"(keep), [wv] "+v"(wv)
                   : [wsrc] "s"(wsrc), [xsrc] "s"(xsrc), [wdst] "s"(wdst), [xdst] "s"(xdst), [wstep] "s"(wstep), [x0] "v"(xv[0]), [x1] "v"(xv[1]),
                     [x2] "v"(xv[2]), [x3] "v"(xv[3]), [dstep] "n"(8 * LDK * 2)
                   : "memory", "scc");
    }
  }
  __syncthreads();
  if (tr0 && threadIdx.x == 0) tr0[1] = wall_clock64();
  const int s = wave & 1, q = wave >> 1;   // strip, k lane (4 per strip)
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < KSL / 4; ++u) {
    const int kk = (q + 4 * u) * 32 + 8 * g;
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Ws + (16 * s + i) * LDK + kk);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(Xs + (mt * 16 + i) * LDK + kk);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mt], 0, 0, 0);
    }
  }
  __syncthreads();   // every wave has read its operands before the slab's space is overwritten
  decode_out_finish<MT>(a, acc, red, &s_last, grp, p);
}

int g_decode_lds = 1;    // mafed_gemm_set_variant(760 / 761): register-direct / LDS-staged operand loads
static int g_num_cus = 0;
static int num_cus() {
  if (!g_num_cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cus = n;
  }
  return g_num_cus;
}

}  // namespace mafed

using namespace mafed;

// h % 256 == 0 (eight waves x whole 32-deep k-steps), n1 % 32 == 0, M <= 64 with ceil(M / 16) * h / 256 <= 16 (the x slab lives in registers)
extern "C" int mafed_decode_supported(int M, int h, int n1) {
  if (M < 1 || M > 64 || h % 256 != 0 || n1 % 32 != 0 || h > 4096) return 0;
  const int mt = (M + 15) / 16, ks = h / 256;
  if (!(ks == 1 || ks == 2 || ks == 3 || ks == 4 || ks == 8)) return 0;
  return mt * ks <= 16 ? 1 : 0;
}

template <int NP>
static int decode_a_launch(const DecodeAArgs& a, hipStream_t st) {
  const int M = a.M, h = a.h;
  const int mt = (M + 15) / 16, ks = h / 256;
  const dim3 grid((unsigned)((a.nqkv + a.n1) / (32 * NP))), block(512);
  const size_t lds = (size_t)8 * 2 * NP * mt * 64 * 16 + (size_t)2 * 8 * mt * 16 * 4;
#define GO(MTV, KSV)                                                                                                              \
  do {                                                                                                                            \
    auto k = decode_ln_qkv_fc1_kernel<MTV, KSV, NP>;                                                                              \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
    k<<<grid, block, lds, st>>>(a);                                                                                               \
  } while (0)
#define GOK(MTV)            \
  do {                      \
    if (ks == 1) GO(MTV, 1); \
    else if (ks == 2) GO(MTV, 2); \
    else if (ks == 3) GO(MTV, 3); \
    else GO(MTV, 4);        \
  } while (0)
  if constexpr (NP == 1) {
    if (ks == 8) {   // (mt <= 2: mafed_decode_supported)
      if (mt == 1) GO(1, 8);
      else GO(2, 8);
    } else {
      switch (mt) {
        case 1: GOK(1); break;
        case 2: GOK(2); break;
        case 3: GOK(3); break;
        default: GOK(4); break;
      }
    }
  } else {   // wide blocks: mt <= 2, ks <= 4 (the caller checks)
    if (mt == 1) GOK(1);
    else GOK(2);
  }
#undef GOK
#undef GO
  return MAFED_OK;
}

extern "C" int mafed_decode_ln_qkv_fc1(const float* x, int M, int h, float eps, const float* ln1_w, const float* ln1_b, const float* ln2_w,
                                       const float* ln2_b, const void* wqkv, const float* bqkv, void* qkv_out, int64_t qkv_ld, const void* w1,
                                       const float* b1, int n1, void* a_out, void* stream) {
  MAFED_CHECK_ARG(mafed_decode_supported(M, h, n1), "decode_ln_qkv_fc1: unsupported shape M=%d h=%d n1=%d", M, h, n1);
  MAFED_CHECK_ARG(x && ln1_w && ln1_b && ln2_w && ln2_b && wqkv && bqkv && qkv_out && w1 && b1 && a_out, "decode_ln_qkv_fc1: null operand");
  MAFED_CHECK_ARG(qkv_ld >= 3 * (int64_t)h && qkv_ld % 4 == 0, "decode_ln_qkv_fc1: qkv_ld");
  DecodeAArgs a{x, M, h, eps, ln1_w, ln1_b, ln2_w, ln2_b, (const bf16_t*)wqkv, bqkv, (bf16_t*)qkv_out, qkv_ld, (const bf16_t*)w1, b1, (bf16_t*)a_out, 3 * h, n1, g_decode_trace};
  const int mt = (M + 15) / 16, ks = h / 256;
  const size_t lds = (size_t)(16 * mt + 32) * (size_t)(h + 8) * 2;
  if (g_decode_lds && mt <= 2 && ks == 4 && lds <= 160 * 1024) {   // full-line loads through LDS (B <= 32 at h = 1024)
    const dim3 grid((unsigned)((3 * h + n1) / 32)), block(512);
    hipStream_t st = (hipStream_t)stream;
#define GOL(MTV, KSV)                                                                                                       \
  do {                                                                                                                      \
    auto k = decode_ln_qkv_fc1_lds_kernel<MTV, KSV>;                                                                        \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
    k<<<grid, block, lds, st>>>(a);                                                                                         \
  } while (0)
#define GOLK(MTV) GOL(MTV, 4)
    if (mt == 1) GOLK(1);
    else GOLK(2);
#undef GOLK
#undef GOL
  } else {
    decode_a_launch<1>(a, (hipStream_t)stream);
  }
  MAFED_CHECK_LAUNCH("decode_ln_qkv_fc1");
  return MAFED_OK;
}

// out[m, 0:N] = LN(x[m]) . w^T (+ bias): the final LayerNorm folded into the LM head's product (bf16 out, row stride ldo)
extern "C" int mafed_decode_ln_linear(const float* x, int M, int h, float eps, const float* ln_w, const float* ln_b, const void* w, const float* bias,
                                      int64_t N, void* out, int64_t ldo, void* stream) {
  MAFED_CHECK_ARG(mafed_decode_supported(M, h, 32) && N % 32 == 0 && N >= 32 && N <= (int64_t)1 << 30, "decode_ln_linear: unsupported shape M=%d h=%d N=%lld",
                  M, h, (long long)N);
  MAFED_CHECK_ARG(x && ln_w && ln_b && w && out && ldo >= N && ldo % 4 == 0, "decode_ln_linear: operands");
  DecodeAArgs a{x, M, h, eps, ln_w, ln_b, ln_w, ln_b, (const bf16_t*)w, bias, (bf16_t*)out, ldo, (const bf16_t*)w, bias, (bf16_t*)out, (int)N, 0, nullptr};
  const int mt = (M + 15) / 16, ks = h / 256;
  const size_t lds = (size_t)(16 * mt + 32) * (size_t)(h + 8) * 2;
  if (g_decode_lds && mt <= 2 && ks == 4 && !bias && N % 16 == 0 && N / 16 >= 4 * num_cus()) {   // big vocabulary: persistent strips
    const size_t ldsh = (size_t)(16 * mt + 32) * (size_t)(h + 8) * 2 + (size_t)8 * mt * 64 * 16;
    const dim3 grid((unsigned)num_cus()), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (mt == 1) {
      auto k = decode_head_kernel<1>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh);
      k<<<grid, block, ldsh, st>>>(a);
    } else {
      auto k = decode_head_kernel<2>;
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh);
      k<<<grid, block, ldsh, st>>>(a);
    }
  } else if (g_decode_lds && mt <= 2 && ks == 4 && lds <= 160 * 1024) {   // full-line loads through LDS: the layer kernel with one segment of N columns
    const dim3 grid((unsigned)(N / 32)), block(512);
    hipStream_t st = (hipStream_t)stream;
    if (mt == 1) {
      auto k = decode_ln_qkv_fc1_lds_kernel<1, 4>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      k<<<grid, block, lds, st>>>(a);
    } else {
      auto k = decode_ln_qkv_fc1_lds_kernel<2, 4>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      k<<<grid, block, lds, st>>>(a);
    }
  } else if (N % 128 == 0 && mt <= 2 && ks <= 4 && N / 128 >= 2 * num_cus()) {
    decode_a_launch<4>(a, (hipStream_t)stream);
  } else {
    decode_a_launch<1>(a, (hipStream_t)stream);
  }
  MAFED_CHECK_LAUNCH("decode_ln_linear");
  return MAFED_OK;
}

// tools: [grid][8] stamps {entered, operands in LDS, K loop done, partial tile out, counter bumped, left} of the next decode_out launches
extern "C" int mafed_decode_set_trace(void* buf) { g_decode_trace = (long long*)buf; return MAFED_OK; }

extern "C" size_t mafed_decode_out_workspace_bytes(int M, int h) {
  // [P <= 16][ceil(M / 16) * 16][h] fp32 partial tiles + h / 32 counters (zero-initialised once by the caller; the kernel re-arms them)
  return (size_t)16 * (size_t)(((M + 15) / 16) * 16) * (size_t)h * sizeof(float) + (size_t)(h / 32) * sizeof(unsigned);
}

extern "C" int mafed_decode_out(const float* x, float* x_out, int M, int h, int n1, const void* ao, const void* act, const void* wd,
                                const float* bd, const void* w2, const float* b2, void* workspace, size_t workspace_bytes, void* stream) {
  MAFED_CHECK_ARG(mafed_decode_supported(M, h, n1), "decode_out: unsupported shape M=%d h=%d n1=%d", M, h, n1);
  MAFED_CHECK_ARG(x && x_out && ao && act && wd && bd && w2 && b2 && workspace, "decode_out: null operand");
  MAFED_CHECK_ARG(workspace_bytes >= mafed_decode_out_workspace_bytes(M, h), "decode_out: workspace too small");
  const int mt = (M + 15) / 16, groups = h / 32, ktot = (h + n1) / 32;
  int P = (num_cus() + groups / 2) / groups;
  if (P < 1) P = 1;
  if (P > 16) P = 16;
  int ksl = (ktot + P - 1) / P;
  ksl = (ksl + 3) / 4 * 4;            // whole rounds of the four k lanes
  P = (ktot + ksl - 1) / ksl;
  float* ws = (float*)workspace;
  unsigned* counters = (unsigned*)((char*)workspace + (size_t)16 * (size_t)(mt * 16) * (size_t)h * sizeof(float));
  DecodeCArgs a{x, x_out, M, h, n1, (const bf16_t*)ao, (const bf16_t*)act, (const bf16_t*)wd, (const bf16_t*)w2, bd, b2, P, ksl, ws, counters, g_decode_trace};
  hipStream_t st = (hipStream_t)stream;
  if (g_decode_lds && mt <= 2 && h == 1024 && n1 % 512 == 0) {   // full-line loads through LDS: 512-deep K slices
    a.ksl = 16;
    a.P = (ktot + 15) / 16;
    if (a.P <= 16) {
      const dim3 grid((unsigned)groups, (unsigned)a.P), block(512);
      if (mt == 1) decode_out_lds_kernel<1><<<grid, block, 0, st>>>(a);
      else decode_out_lds_kernel<2><<<grid, block, 0, st>>>(a);
      MAFED_CHECK_LAUNCH("decode_out");
      return MAFED_OK;
    }
    a.ksl = ksl;
    a.P = P;
  }
  const dim3 grid((unsigned)groups, (unsigned)P), block(512);
  switch (mt) {
    case 1: decode_out_kernel<1><<<grid, block, 0, st>>>(a); break;
    case 2: decode_out_kernel<2><<<grid, block, 0, st>>>(a); break;
    case 3: decode_out_kernel<3><<<grid, block, 0, st>>>(a); break;
    default: decode_out_kernel<4><<<grid, block, 0, st>>>(a); break;
  }
  MAFED_CHECK_LAUNCH("decode_out");
  return MAFED_OK;
}
