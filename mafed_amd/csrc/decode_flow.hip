// One launch per decode step (mafed/model/vqa_cont_learner.py:260-277 -> HF greedy search over vl_pythia.py's
// stack): every layer's work items -- LayerNorm rows, 16-column strips of [q|k|v|4h], (batch, head) attention slices, K-slices of the
// [dense|fc2] product -- and the LM head are workgroups of ONE grid, ordered by workgroup id in dependency order, and hand over through
// arrival counters in device memory instead of kernel boundaries.
//
// Why: with one launch per phase (csrc/decode.hip: 3 per layer) a layer costs 35 us for 63 MB of HBM traffic (11.5 us at 5.5 TB/s): every
// launch pays the launch itself (1.7 us in a graph chain), the first HBM byte (~2 us) and its own tail while HBM idles.  Here a workgroup
// requests its weight slab / K|V rows (which depend on nothing) as soon as it is dispatched, THEN waits for its inputs' counter: the 768
// resident workgroups (3 per CU) always hold about half a layer of weights in flight ahead of the dependency front.
//
// Deadlock freedom: workgroups are dispatched in id order, and a workgroup only ever waits for counters bumped by workgroups with
// SMALLER ids (LN rows < strips < attention slices < K-slices of the same layer < next layer) -- all of which are already resident or
// done.  Every spin loop is bounded (~0.3 s) and then raises `err` and carries on, so the grid drains whatever happens.
// Visibility: everything one workgroup writes for another (possibly on another XCD, i.e. behind another L2) is written and read with
// agent-scope accesses (sc1: written through / read around the L2); a producer's stores have left (vmcnt(0)) before its counter moves.
// Determinism: cross-wave and cross-workgroup sums are taken in a fixed order (no floating-point atomics).
//
// Shapes served: bf16, h = 1024 class (h % 512 == 0, h / 128 == 8 k-steps per wave), head size 64, M <= 32 rows, n1 % 512 == 0.
#include "common.h"

namespace mafed {

struct FlowLayer {   // device pointers of one layer (entry L = the head: ln1 = final LayerNorm, wqkv = embed_out)
  const float *ln1w, *ln1b, *ln2w, *ln2b;
  const bf16_t* wqkv;
  const float* bqkv;
  const bf16_t* w1;
  const float* b1;
  const bf16_t* wd;
  const float* bd;
  const bf16_t* w2;
  const float* b2;
  const bf16_t* kv_pre;
  bf16_t* kv_new;
};

constexpr int FLOW_XR = 0, FLOW_LN = 1, FLOW_HEAD = 8, FLOW_AR = 72, FLOW_AO = 104, FLOW_CD = 120, FLOW_STRIDE = 256;

struct FlowArgs {
  const FlowLayer* layers;   // [L + 1]
  int L, M, h, n1, H, S0, cap, t, rot, P, Tm, V;
  float eps;
  float* x;                  // [32, h] fp32 residual stream, in place
  bf16_t *ln1, *ln2;         // [32, h]
  bf16_t* act;               // [32, n1]
  bf16_t* ao;                // [32, h]
  float* ws;                 // [(h + n1) / 512][32][h] partial tiles
  unsigned* flags;           // [L + 1][FLOW_STRIDE], zero at launch
  const float *rc, *rs;
  const int64_t* am;
  bf16_t* logits;            // [M, V]
  unsigned* err;
  int chain;                 // 1: the whole step (every hand-over through counters); 0: one layer's attention + K-slices (decode_attn_out_kernel)
  long long* trace;          // optional [grid][4] wall-clock stamps (dispatch, wait done, role done): tools/decode_flow_trace.py
  int nL, nAq, nAf, nB, nCa, nCo, per_layer;
};

// ---- hand-over primitives --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void flow_wait(const unsigned* p, unsigned target, unsigned* err, long long* tr = nullptr) {
  // ONE wave of the block polls (an agent-scope load goes to the memory side: with every wave of 768 resident workgroups polling the same
  // few words the polls queued up in front of the data loads -- 1.62 ms per step against 0.88 for the three-launch layers); the others
  // wait at the barrier
  if (threadIdx.x < 64) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1 << 21)) {   // never expected: raise the flag and carry on (the grid must drain)
        if (threadIdx.x == 0) atomicOr(err, 1u);
        break;
      }
    }
  }
  __syncthreads();
  if (tr && threadIdx.x == 0) tr[1] = wall_clock64();
  asm volatile("" ::: "memory");
}
// after the block's agent-scope stores: they have left, then the counter moves
__device__ __forceinline__ void flow_signal(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(void* p, uint32_t v) { __hip_atomic_store((uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16); }

// N 16-byte agent-scope loads in flight at once (one statement: issue all, then wait); addresses p + k * STRIDE bytes
#define FLOW_LD(k, off) "global_load_dwordx4 %" #k ", %[p], off offset:" #off " sc1\n\t"
template <int STRIDE>
__device__ __forceinline__ void ld4_sc1(const void* p, uint4 (&o)[4]) {
  static_assert(STRIDE == 256 || STRIDE == 1024, "offsets below are spelled out for these strides");
  if constexpr (STRIDE == 256)
    asm volatile(FLOW_LD(0, 0) FLOW_LD(1, 256) FLOW_LD(2, 512) FLOW_LD(3, 768) "s_waitcnt vmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : [p] "v"(p) : "memory");
  else
    asm volatile(FLOW_LD(0, 0) FLOW_LD(1, 1024) FLOW_LD(2, 2048) FLOW_LD(3, 3072) "s_waitcnt vmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]) : [p] "v"(p) : "memory");
}
#define FLOW_LD2(k, ptr, off) "global_load_dwordx4 %" #k ", %[" #ptr "], off offset:" #off " sc1\n\t"
// two rows x four fragments, 256 bytes apart (the [dense|fc2] K-slice: k-steps q, q + 4, q + 8, q + 12)
__device__ __forceinline__ void ld2x4_sc1(const void* p0, const void* p1, uint4 (&a)[4], uint4 (&b)[4]) {
  asm volatile(FLOW_LD2(0, p0, 0) FLOW_LD2(1, p0, 256) FLOW_LD2(2, p0, 512) FLOW_LD2(3, p0, 768)
               FLOW_LD2(4, p1, 0) FLOW_LD2(5, p1, 256) FLOW_LD2(6, p1, 512) FLOW_LD2(7, p1, 768) "s_waitcnt vmcnt(0)"
               : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
               : [p0] "v"(p0), [p1] "v"(p1) : "memory");
}
// two rows x eight fragments, 64 bytes apart (a wave's K quarter of the normalised rows)
__device__ __forceinline__ void ld2x8_sc1(const void* p0, const void* p1, uint4 (&a)[8], uint4 (&b)[8]) {
  asm volatile(FLOW_LD2(0, p0, 0) FLOW_LD2(1, p0, 64) FLOW_LD2(2, p0, 128) FLOW_LD2(3, p0, 192) FLOW_LD2(4, p0, 256) FLOW_LD2(5, p0, 320)
               FLOW_LD2(6, p0, 384) FLOW_LD2(7, p0, 448) FLOW_LD2(8, p1, 0) FLOW_LD2(9, p1, 64) FLOW_LD2(10, p1, 128) FLOW_LD2(11, p1, 192)
               FLOW_LD2(12, p1, 256) FLOW_LD2(13, p1, 320) FLOW_LD2(14, p1, 384) FLOW_LD2(15, p1, 448) "s_waitcnt vmcnt(0)"
               : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]), "=&v"(b[0]), "=&v"(b[1]),
                 "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7])
               : [p0] "v"(p0), [p1] "v"(p1) : "memory");
}
// five independent 16-byte loads (this step's q | k | v chunks)
__device__ __forceinline__ void ld5_sc1(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, uint4 (&o)[5]) {
  asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
               "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
               : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4) : "memory");
}

__device__ __forceinline__ void flow_unpack8(const uint4& r, float (&v)[8]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

// ---- role L: LayerNorm rows (a wave per row) ---------------------------------------------------------------------------------------------
// x row (agent-scope loads: written by the previous layer's K-slice reducers) -> ln1 (and ln2) bf16 rows
__device__ __forceinline__ void flow_ln(const FlowArgs& a, const FlowLayer& ly, int layer, int blk, bool final_ln) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = a.h, row = blk * 4 + wave;
  unsigned* fl = a.flags + (size_t)layer * FLOW_STRIDE;
  // affine parameters do not depend on anything
  float4 g1[2], o1[2], g2[2], o2[2];   // h = 1024: 16 columns per lane = pieces jj = 0 .. 3 of 4 floats; held as two halves to bound registers
  if (layer > 0) flow_wait(fl + FLOW_XR, (unsigned)(h / 32), a.err, a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr);
  if (row < a.M) {
    uint4 raw[4];
    ld4_sc1<1024>(reinterpret_cast<const char*>(a.x + (size_t)row * h) + lane * 16, raw);
    float v[16];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      v[4 * jj + 0] = __uint_as_float(raw[jj].x); v[4 * jj + 1] = __uint_as_float(raw[jj].y);
      v[4 * jj + 2] = __uint_as_float(raw[jj].z); v[4 * jj + 3] = __uint_as_float(raw[jj].w);
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    const float mean = wave_sum(s) / (float)h;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)h + a.eps);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int jj = half * 2 + j2, col = 4 * (lane + 64 * jj);
        g1[j2] = load4(ly.ln1w + col); o1[j2] = load4(ly.ln1b + col);
        if (!final_ln) { g2[j2] = load4(ly.ln2w + col); o2[j2] = load4(ly.ln2b + col); }
      }
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int jj = half * 2 + j2, col = 4 * (lane + 64 * jj);
        const float n0 = (v[4 * jj] - mean) * rstd, n1 = (v[4 * jj + 1] - mean) * rstd, n2 = (v[4 * jj + 2] - mean) * rstd, n3 = (v[4 * jj + 3] - mean) * rstd;
        bf16_t* d1 = a.ln1 + (size_t)row * h + col;
        st_sc1(d1, pack_bf16(n0 * g1[j2].x + o1[j2].x, n1 * g1[j2].y + o1[j2].y));
        st_sc1(d1 + 2, pack_bf16(n2 * g1[j2].z + o1[j2].z, n3 * g1[j2].w + o1[j2].w));
        if (!final_ln) {
          bf16_t* d2 = a.ln2 + (size_t)row * h + col;
          st_sc1(d2, pack_bf16(n0 * g2[j2].x + o2[j2].x, n1 * g2[j2].y + o2[j2].y));
          st_sc1(d2 + 2, pack_bf16(n2 * g2[j2].z + o2[j2].z, n3 * g2[j2].w + o2[j2].w));
        }
      }
    }
  }
  flow_signal(fl + FLOW_LN);
}

// ---- role A: a 16-column strip of [q|k|v] / fc1 / the LM head (four waves split K = h) ---------------------------------------------------
// MFMA 16x16x32: A = W rows n0 + i (k = 8g..), B = normalised rows (m = i): lane (i, g) holds C[m = mt*16 + i][n0 + 4g .. + 3]
template <int MT>
__device__ __forceinline__ void flow_strip(const FlowArgs& a, const FlowLayer& ly, int layer, int kind, int sidx, f32x4* red) {
  // kind 0: qkv strip (LN1 rows, bias, -> K/V cache row t), 1: fc1 strip (LN2 rows, bias, GELU -> act), 2: head strip (LN1 rows -> logits)
  constexpr int KS = 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h, n0 = sidx * 16;
  unsigned* fl = a.flags + (size_t)layer * FLOW_STRIDE;
  const bf16_t* W = kind == 1 ? ly.w1 : ly.wqkv;
  const int kb = wave * (KS * 32) + 8 * g;
  bf16x8 wf[KS];
#pragma unroll
  for (int u = 0; u < KS; ++u) wf[u] = *reinterpret_cast<const bf16x8*>(W + (size_t)(n0 + i) * h + kb + 32 * u);
  const float* bias = kind == 0 ? ly.bqkv : (kind == 1 ? ly.b1 : nullptr);
  float4 bia = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bia = load4(bias + n0 + 4 * g);   // (uniform branch)
  flow_wait(fl + FLOW_LN, (unsigned)a.nL, a.err, a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr);
  const bf16_t* X = kind == 1 ? a.ln2 : a.ln1;
  uint4 xa[8], xb[8];
  ld2x8_sc1(X + (size_t)i * h + kb, X + (size_t)((MT > 1 ? 16 : 0) + i) * h + kb, xa, xb);
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < KS; ++u) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], __builtin_bit_cast(bf16x8, xa[u]), acc[0], 0, 0, 0);
    if constexpr (MT > 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], __builtin_bit_cast(bf16x8, xb[u]), acc[1], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[((wave - 1) * MT + mt) * 64 + lane] = acc[mt];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v = acc[mt];
#pragma unroll
      for (int w = 0; w < 3; ++w) v += red[(w * MT + mt) * 64 + lane];
      const int m = mt * 16 + i;
      if (m < a.M) {
        float o0 = v[0] + bia.x, o1 = v[1] + bia.y, o2 = v[2] + bia.z, o3 = v[3] + bia.w;
        const int n = n0 + 4 * g;
        if (kind == 0) {
          bf16_t* d = ly.kv_new + ((size_t)m * a.cap + a.t) * (size_t)(3 * h) + n;
          st_sc1(d, pack_bf16(o0, o1));
          st_sc1(d + 2, pack_bf16(o2, o3));
        } else if (kind == 1) {
          bf16_t* d = a.act + (size_t)m * a.n1 + n;
          st_sc1(d, pack_bf16(gelu_erf_fast(o0), gelu_erf_fast(o1)));
          st_sc1(d + 2, pack_bf16(gelu_erf_fast(o2), gelu_erf_fast(o3)));
        } else {
          store4(a.logits + (size_t)m * a.V + n, make_float4(o0, o1, o2, o3));   // read by the next launch only
        }
      }
    }
  }
  if (kind == 0) flow_signal(fl + FLOW_HEAD + (n0 / (3 * 64)));
  else if (kind == 1) flow_signal(fl + FLOW_AR + (n0 / 512));
}

// ---- role B: attention of one (batch, head) slice over the pre-rotated cache (attn_ref.hip: attn_decode_flat_kernel's form) ---------------
template <int UNR, bool WAIT = true>
__device__ __forceinline__ void flow_attn(const FlowArgs& a, const FlowLayer& ly, int layer, int bh, float* lds) {
  constexpr int D = 64, chunks = D / 8, groups = 256 / chunks;
  float(*red)[D + 1] = reinterpret_cast<float(*)[D + 1]>(lds);      // [4][D + 1]
  float* wmax = lds + 4 * (D + 1);                                   // [4]
  unsigned char* msk = reinterpret_cast<unsigned char*>(wmax + 4);   // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, S0 = a.S0, t = a.t, rot = a.rot, P = a.P, Tm = a.Tm;
  const int b = bh / H, hh = bh - b * H;
  unsigned* fl = a.flags + (size_t)layer * FLOW_STRIDE;
  const int nk = S0 + t + 1;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* pre = ly.kv_pre + ((int64_t)b * S0 * H + hh) * 3 * D;
  bf16_t* neu = ly.kv_new + ((int64_t)b * a.cap * H + hh) * 3 * D;
  const int c = tid % chunks, kg = tid / chunks;
  const int64_t mword = a.am[(int64_t)b * Tm + (tid < Tm ? tid : Tm - 1)];
  const int hc = rot >> 4, half = rot >> 1;
  const bool inrot = c * 8 < rot, first = c < hc;
  const int cpart = inrot ? (first ? c + hc : c - hc) : c;
  const int ccs = inrot ? (first ? c : c - hc) * 8 : 0;
  const float* cp = a.rc + (int64_t)(S0 + t) * half + ccs;
  const float* sp = a.rs + (int64_t)(S0 + t) * half + ccs;
  const float4 cs0 = load4(cp), cs1 = load4(cp + 4), sn0 = load4(sp), sn1 = load4(sp + 4);
  // every cached key / value row of the slice is requested before the wait (rows < nk - 1 were written by earlier launches)
  uint4 kraw[UNR], vraw[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = kg + u * groups;
    const int jc = j < nk - 1 ? j : (nk > 1 ? nk - 2 : 0);   // clamped to the last OLD row (row t itself is read after the wait)
    const bf16_t* row = jc < S0 ? pre + (int64_t)jc * rstride : neu + (int64_t)(jc - S0) * rstride;
    kraw[u] = *reinterpret_cast<const uint4*>(row + D + c * 8);
    vraw[u] = *reinterpret_cast<const uint4*>(row + 2 * D + c * 8);
  }
  if (tid < Tm) msk[tid] = mword != 0;
  if constexpr (WAIT) flow_wait(fl + FLOW_HEAD + hh, (unsigned)(3 * D / 16), a.err, a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr);
  else __syncthreads();   // (row t was written by the previous launch; the barrier only publishes the mask bytes early)
  const bf16_t* qrow = neu + (int64_t)t * rstride;
  uint4 nw[5];
  ld5_sc1(qrow + c * 8, qrow + cpart * 8, qrow + D + c * 8, qrow + D + cpart * 8, qrow + 2 * D + c * 8, nw);
  const float scale = rsqrtf((float)D);
  float qr[8], knew[8], vnew[8];
  {
    float a0[8], a1[8], b0[8], b1[8];
    flow_unpack8(nw[0], a0); flow_unpack8(nw[1], a1); flow_unpack8(nw[2], b0); flow_unpack8(nw[3], b1); flow_unpack8(nw[4], vnew);
    const float cs[8] = {cs0.x, cs0.y, cs0.z, cs0.w, cs1.x, cs1.y, cs1.z, cs1.w};
    const float sn[8] = {sn0.x, sn0.y, sn0.z, sn0.w, sn1.x, sn1.y, sn1.z, sn1.w};
    const float sgn = first ? -1.f : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qr[e] = (inrot ? a0[e] * cs[e] + sgn * a1[e] * sn[e] : a0[e]) * scale;
      knew[e] = inrot ? b0[e] * cs[e] + sgn * b1[e] * sn[e] : b0[e];
    }
  }
  __syncthreads();   // mask bytes; every lane has read row t's un-rotated key before the write-back below
  if (tid < chunks) {   // rotated, for the steps to come (read by later launches only)
    uint4 r;
    r.x = pack_bf16(knew[0], knew[1]); r.y = pack_bf16(knew[2], knew[3]); r.z = pack_bf16(knew[4], knew[5]); r.w = pack_bf16(knew[6], knew[7]);
    *reinterpret_cast<uint4*>(neu + (int64_t)t * rstride + D + tid * 8) = r;
  }
  float sc[UNR];
  float tmax = -INFINITY;
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = kg + u * groups;
    float x[8];
    flow_unpack8(kraw[u], x);
    const bool own = j == nk - 1;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(qr[e], own ? knew[e] : x[e], s);
#pragma unroll
    for (int o = 1; o < chunks; o <<= 1) s += __shfl_xor(s, o, 64);
    const int ti = j >= P && j < S0 ? j - P : 0;
    const bool ok = j < nk && (j < P || j >= S0 || msk[ti] != 0);
    sc[u] = ok ? s : -INFINITY;
    tmax = fmaxf(tmax, sc[u]);
  }
#pragma unroll
  for (int o = chunks; o < 64; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, 64));
  if (lane == 0) wmax[wave] = tmax;
  __syncthreads();
  float mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  if (mx == -INFINITY) mx = 0.f;
  float l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = kg + u * groups;
    float vv[8];
    flow_unpack8(vraw[u], vv);
    const bool own = j == nk - 1;
    const float p = __expf(sc[u] - mx);
    l += p;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, own ? vnew[e] : vv[e], acc[e]);
  }
#pragma unroll
  for (int o = chunks; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < chunks) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][c * 8 + e] = acc[e];
    if (lane == 0) red[wave][D] = l;
  }
  __syncthreads();
  if (tid < D / 2) {   // two output columns per lane: one agent-scope dword
    const int c0 = 2 * tid;
    const float lt = (red[0][D] + red[1][D]) + (red[2][D] + red[3][D]);
    const float oa = (red[0][c0] + red[1][c0]) + (red[2][c0] + red[3][c0]), ob = (red[0][c0 + 1] + red[1][c0 + 1]) + (red[2][c0 + 1] + red[3][c0 + 1]);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    st_sc1(a.ao + (size_t)b * a.h + (size_t)hh * D + c0, pack_bf16(lt > 0.f ? oa / lt : 0.f, lt > 0.f ? ob / lt : 0.f));
    (void)inv;
  }
  flow_signal(fl + FLOW_AO + (hh * D) / 512);
}

// ---- role C: a 512-deep K-slice of 32 columns of x + dense(ao) + fc2(act); the last slice of a column group to arrive reduces ----------
template <int MT, bool WAIT = true>
__device__ __forceinline__ void flow_out(const FlowArgs& a, const FlowLayer& ly, int layer, int grp, int slice, f32x4* red, int* s_last) {
  // slice: index into the concatenated K in units of 512 (slices < h / 512 read ao . Wd, the others act . W2)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int h = a.h, n1 = a.n1, n0 = grp * 32;
  unsigned* fl = a.flags + (size_t)layer * FLOW_STRIDE;
  const int nao = h / 512, P = nao + n1 / 512;
  const bool is_ao = slice < nao;
  const int ld = is_ao ? h : n1, kofs = (is_ao ? slice : slice - nao) * 512;
  const bf16_t* Wsrc = is_ao ? ly.wd : ly.w2;
  const bf16_t* Xsrc = is_ao ? a.ao : a.act;
  const int kq = kofs + wave * 32 + 8 * g;   // this wave's k-steps: wave, wave + 4, wave + 8, wave + 12 (128 elements apart)
  bf16x8 wf[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int u = 0; u < 4; ++u) wf[s][u] = *reinterpret_cast<const bf16x8*>(Wsrc + (size_t)(n0 + 16 * s + i) * ld + kq + 128 * u);
  uint4 xa[4], xb[4];
  if constexpr (WAIT) {
    if (is_ao) flow_wait(fl + FLOW_AO + slice, (unsigned)((512 / 64) * a.M), a.err, a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr);
    else flow_wait(fl + FLOW_AR + (slice - nao), 32u, a.err, a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr);
    ld2x4_sc1(Xsrc + (size_t)i * ld + kq, Xsrc + (size_t)((MT > 1 ? 16 : 0) + i) * ld + kq, xa, xb);
  } else {   // the operand was written by the previous launch: plain (L2-cached) loads, nothing to wait for
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xa[u] = *reinterpret_cast<const uint4*>(Xsrc + (size_t)i * ld + kq + 128 * u);
      xb[u] = *reinterpret_cast<const uint4*>(Xsrc + (size_t)((MT > 1 ? 16 : 0) + i) * ld + kq + 128 * u);
    }
  }
  f32x4 acc[2][MT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[s][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][u], __builtin_bit_cast(bf16x8, xa[u]), acc[s][0], 0, 0, 0);
      if constexpr (MT > 1) acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][u], __builtin_bit_cast(bf16x8, xb[u]), acc[s][1], 0, 0, 0);
    }
  // every wave publishes its tiles, then wave w sums tile w (strip w & 1, row block w >> 1) over the four k lanes in a fixed order
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[((wave * 2 + s) * MT + mt) * 64 + lane] = acc[s][mt];
  __syncthreads();
  const int es = wave & 1, emt = wave >> 1;
  const int m = emt * 16 + i, nn = n0 + 16 * es + 4 * g;
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (emt < MT) {
    v = red[((0 * 2 + es) * MT + emt) * 64 + lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) v += red[((q * 2 + es) * MT + emt) * 64 + lane];
    float* wp = a.ws + ((size_t)slice * 32 + m) * h + nn;
#pragma unroll
    for (int e = 0; e < 4; ++e) st_sc1(wp + e, v[e]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(fl + FLOW_CD + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = old == (unsigned)(P - 1);
  }
  __syncthreads();
  if (!*s_last) return;   // (block-uniform)
  if (emt < MT) {
    const float4 c0 = load4(ly.bd + nn), c1 = load4(ly.b2 + nn);
    float r[4], sum[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xr = a.x + (size_t)m * h + nn;
    if (m < a.M) {
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = ld_sc1(xr + e);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = 0.f;
    }
    // all P <= 16 slices requested at once (clamped index: not a dependent round trip per slice), added in slice order whichever block
    // happens to be last
    float tv[16][4];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      const float* rp = a.ws + ((size_t)(pp < P ? pp : 0) * 32 + m) * h + nn;
      if (pp < 12 || P > 12) {   // (P = 10 at 410M: twelve loads in flight)
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[pp][e] = ld_sc1(rp + e);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) tv[pp][e] = 0.f;
      }
    }
#pragma unroll
    for (int pp = 0; pp < 16; ++pp)
      if (pp < P) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += tv[pp][e];
      }
    if (m < a.M) {
      float* xo = a.x + (size_t)m * h + nn;
      st_sc1(xo + 0, r[0] + (sum[0] + c0.x + c1.x));
      st_sc1(xo + 1, r[1] + (sum[1] + c0.y + c1.y));
      st_sc1(xo + 2, r[2] + (sum[2] + c0.z + c1.z));
      st_sc1(xo + 3, r[3] + (sum[3] + c0.w + c1.w));
    }
  }
  if (a.chain) flow_signal(a.flags + (size_t)(layer + 1) * FLOW_STRIDE + FLOW_XR);
}

template <int MT, int UNR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void decode_flow_kernel(FlowArgs a) {
  __shared__ __align__(16) unsigned char lds_raw[4 * 2 * MT * 64 * 16 + 16];
  f32x4* red = reinterpret_cast<f32x4*>(lds_raw);
  int* s_last = reinterpret_cast<int*>(lds_raw + 4 * 2 * MT * 64 * 16);
  const int bid = blockIdx.x;
  const int body = a.L * a.per_layer;
  long long* tr = a.trace ? a.trace + (size_t)bid * 4 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  if (bid < body) {
    const int layer = bid / a.per_layer;
    int r = bid - layer * a.per_layer;
    const FlowLayer ly = a.layers[layer];
    const int groups = a.h / 32, nao = a.h / 512;
    if (r < a.nL) flow_ln(a, ly, layer, r, false);
    else if ((r -= a.nL) < a.nAq) flow_strip<MT>(a, ly, layer, 0, r, red);
    else if ((r -= a.nAq) < a.nAf) flow_strip<MT>(a, ly, layer, 1, r, red);
    else if ((r -= a.nAf) < a.nB) flow_attn<UNR>(a, ly, layer, r, reinterpret_cast<float*>(lds_raw));
    else if ((r -= a.nB) < a.nCa) flow_out<MT>(a, ly, layer, r % groups, nao + r / groups, red, s_last);   // fc2 slices first: they wait for strips only
    else { r -= a.nCa; flow_out<MT>(a, ly, layer, r % groups, r / groups, red, s_last); }
  } else {
    const FlowLayer ly = a.layers[a.L];
    int r = bid - body;
    if (r < a.nL) flow_ln(a, ly, a.L, r, true);
    else flow_strip<MT>(a, ly, a.L, 2, r - a.nL, red);
  }
  if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
}

// One layer's attention and [dense|fc2] product as ONE launch behind mafed_decode_ln_qkv_fc1 (round 4): the fc2 K-slices (they need
// only the previous launch's activation row) come first in id order, then the (batch, head) attention slices, then the dense K-slices,
// which wait for their eight heads' arrival counter.  The K|V stream (38.5 MB) and the 10 MB of weights are requested together from the
// first microsecond; of the 832 workgroups 768 are resident at once, the 64 dense K-slices take the first slots the fc2 slices free.
template <int MT, int UNR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void decode_attn_out_kernel(FlowArgs a) {
  __shared__ __align__(16) unsigned char lds_raw[4 * 2 * MT * 64 * 16 + 16];
  f32x4* red = reinterpret_cast<f32x4*>(lds_raw);
  int* s_last = reinterpret_cast<int*>(lds_raw + 4 * 2 * MT * 64 * 16);
  const FlowLayer ly = a.layers[0];
  const int groups = a.h / 32, nao = a.h / 512;
  int r = blockIdx.x;
  long long* tr = a.trace ? a.trace + (size_t)blockIdx.x * 4 : nullptr;
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  if (r < a.nCa) flow_out<MT, false>(a, ly, 0, r % groups, nao + r / groups, red, s_last);
  else if ((r -= a.nCa) < a.nB) flow_attn<UNR, false>(a, ly, 0, r, reinterpret_cast<float*>(lds_raw));
  else { r -= a.nB; flow_out<MT, true>(a, ly, 0, r % groups, r / groups, red, s_last); }
  if (tr && threadIdx.x == 0) tr[2] = wall_clock64();
}

}  // namespace mafed

using namespace mafed;

static long long* g_flow_trace = nullptr;   // tools: [grid][4] stamps per workgroup of the next launches (mafed_decode_flow_set_trace)
extern "C" int mafed_decode_flow_set_trace(void* buf) { g_flow_trace = (long long*)buf; return MAFED_OK; }
extern "C" size_t mafed_decode_flow_grid(int L, int M, int h, int n1, int H, int V) {
  const int mt = (M + 15) / 16, nl = mt * 4;
  return (size_t)L * (nl + 3 * h / 16 + n1 / 16 + M * H + (h / 32) * (n1 / 512) + (h / 32) * (h / 512)) + nl + V / 16;
}

extern "C" int mafed_decode_flow_supported(int M, int h, int n1, int H, int D, int V, int nk) {
  if (M < 1 || M > 32 || h != 1024 || n1 % 512 != 0 || n1 > 16 * 512 || D != 64 || H * D != h || V % 16 != 0) return 0;
  return (nk + 31) / 32 <= 24 ? 1 : 0;
}

// bytes of the flag block a launch needs: (L + 1) * 256 counters + 1 error word; zero-filled before EVERY launch
extern "C" size_t mafed_decode_flow_flag_bytes(int L) { return ((size_t)(L + 1) * FLOW_STRIDE + 4) * sizeof(unsigned); }
extern "C" size_t mafed_decode_flow_workspace_bytes(int h, int n1) { return (size_t)((h + n1) / 512) * 32 * (size_t)h * sizeof(float); }

// One decode step: x [32, h] fp32 (rows < M hold the embedded tokens) -> logits [M, V] bf16; appends row t to every layer's K/V cache.
// layers: device array of L + 1 records of 14 pointers (struct FlowLayer above; record L: ln1w / ln1b = final LayerNorm, wqkv = embed_out).
extern "C" int mafed_decode_flow_step(const void* layers, int L, int M, int h, int n1, int H, int D, int S0, int cap, int t, int rot, int P, int Tm, int V,
                                      float eps, float* x, void* ln1, void* ln2, void* act, void* ao, void* workspace, size_t workspace_bytes,
                                      void* flags, size_t flag_bytes, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask,
                                      void* logits, void* stream) {
  MAFED_CHECK_ARG(mafed_decode_flow_supported(M, h, n1, H, D, V, S0 + t + 1), "decode_flow: unsupported shape");
  MAFED_CHECK_ARG(layers && x && ln1 && ln2 && act && ao && workspace && flags && rot_cos && rot_sin && attention_mask && logits, "decode_flow: null operand");
  MAFED_CHECK_ARG(L >= 1 && t >= 0 && t < cap && rot % 16 == 0 && Tm >= 1 && Tm <= 256 && S0 >= P, "decode_flow: bad step arguments");
  MAFED_CHECK_ARG(workspace_bytes >= mafed_decode_flow_workspace_bytes(h, n1) && flag_bytes >= mafed_decode_flow_flag_bytes(L), "decode_flow: buffers too small");
  FlowArgs a;
  a.layers = (const FlowLayer*)layers;
  a.L = L; a.M = M; a.h = h; a.n1 = n1; a.H = H; a.S0 = S0; a.cap = cap; a.t = t; a.rot = rot; a.P = P; a.Tm = Tm; a.V = V;
  a.eps = eps;
  a.x = x; a.ln1 = (bf16_t*)ln1; a.ln2 = (bf16_t*)ln2; a.act = (bf16_t*)act; a.ao = (bf16_t*)ao; a.ws = (float*)workspace;
  a.flags = (unsigned*)flags;
  a.err = a.flags + (size_t)(L + 1) * FLOW_STRIDE;
  a.trace = g_flow_trace;
  a.chain = 1;
  a.rc = rot_cos; a.rs = rot_sin; a.am = attention_mask; a.logits = (bf16_t*)logits;
  const int mt = (M + 15) / 16;
  a.nL = mt * 4;
  a.nAq = 3 * h / 16; a.nAf = n1 / 16; a.nB = M * H;
  a.nCa = (h / 32) * (n1 / 512); a.nCo = (h / 32) * (h / 512);
  a.per_layer = a.nL + a.nAq + a.nAf + a.nB + a.nCa + a.nCo;
  const int64_t total = (int64_t)L * a.per_layer + a.nL + V / 16;
  const int need = (S0 + t + 1 + 31) / 32;
  const dim3 grid((unsigned)total), block(256);
  hipStream_t st = (hipStream_t)stream;
#define GO(MTV, UV) decode_flow_kernel<MTV, UV><<<grid, block, 0, st>>>(a)
  if (mt == 1) { if (need <= 10) GO(1, 10); else if (need <= 16) GO(1, 16); else GO(1, 24); }
  else { if (need <= 10) GO(2, 10); else if (need <= 16) GO(2, 16); else GO(2, 24); }
#undef GO
  MAFED_CHECK_LAUNCH("decode_flow_step");
  return MAFED_OK;
}

// Second launch of a decode layer (behind mafed_decode_ln_qkv_fc1): attention over the pre-rotated cache + x <- x + dense(ao) + fc2(act).
// layer_rec: one record of 14 pointers (see mafed_decode_flow_step; ln / qkv / fc1 entries unused).  act [32, n1], ao [32, h] (scratch) bf16;
// flags: 1 KB, ZERO on entry (a launch leaves its counters non-zero: one slot per launch, or re-zeroed between uses).
extern "C" int mafed_decode_attn_out(const void* layer_rec, int M, int h, int n1, int H, int D, int S0, int cap, int t, int rot, int P, int Tm,
                                     float* x, const void* act, void* ao, void* workspace, size_t workspace_bytes, void* flags,
                                     const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, void* stream) {
  MAFED_CHECK_ARG(mafed_decode_flow_supported(M, h, n1, H, D, 16, S0 + t + 1), "decode_attn_out: unsupported shape");
  MAFED_CHECK_ARG(layer_rec && x && act && ao && workspace && flags && rot_cos && rot_sin && attention_mask, "decode_attn_out: null operand");
  MAFED_CHECK_ARG(t >= 0 && t < cap && rot % 16 == 0 && Tm >= 1 && Tm <= 256 && S0 >= P, "decode_attn_out: bad step arguments");
  MAFED_CHECK_ARG(workspace_bytes >= mafed_decode_flow_workspace_bytes(h, n1), "decode_attn_out: workspace too small");
  FlowArgs a;
  a.layers = (const FlowLayer*)layer_rec;
  a.L = 1; a.M = M; a.h = h; a.n1 = n1; a.H = H; a.S0 = S0; a.cap = cap; a.t = t; a.rot = rot; a.P = P; a.Tm = Tm; a.V = 0;
  a.eps = 0.f;
  a.x = x; a.ln1 = nullptr; a.ln2 = nullptr; a.act = (bf16_t*)const_cast<void*>(act); a.ao = (bf16_t*)ao; a.ws = (float*)workspace;
  a.flags = (unsigned*)flags;
  a.err = a.flags + FLOW_STRIDE - 1;   // last word of the slot
  a.rc = rot_cos; a.rs = rot_sin; a.am = attention_mask; a.logits = nullptr;
  a.chain = 0;
  a.trace = g_flow_trace;
  const int mt = (M + 15) / 16;
  a.nL = 0; a.nAq = 0; a.nAf = 0; a.nB = M * H;
  a.nCa = (h / 32) * (n1 / 512); a.nCo = (h / 32) * (h / 512);
  a.per_layer = a.nB + a.nCa + a.nCo;
  const int need = (S0 + t + 1 + 31) / 32;
  const dim3 grid((unsigned)a.per_layer), block(256);
  hipStream_t st = (hipStream_t)stream;
#define GO(MTV, UV) decode_attn_out_kernel<MTV, UV><<<grid, block, 0, st>>>(a)
  if (mt == 1) { if (need <= 10) GO(1, 10); else if (need <= 16) GO(1, 16); else GO(1, 24); }
  else { if (need <= 10) GO(2, 10); else if (need <= 16) GO(2, 16); else GO(2, 24); }
#undef GO
  MAFED_CHECK_LAUNCH("decode_attn_out");
  return MAFED_OK;
}
