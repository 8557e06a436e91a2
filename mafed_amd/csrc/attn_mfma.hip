// bf16 MFMA flash-style attention for the VLPythia decoder (tf:154-236; replaces the flash-attn-2 wheel, README.md:16).
//
// Orientation: every product is computed with the QUERY (forward, dQ) or the KEY (dK/dV) on MFMA column = lane&15, so
// that row statistics (running max / sum, LSE, delta) are lane-local scalars, the S^T / dS^T accumulators are already
// the B operand of the next product (cdna_hip_programming "An accumulator tile as the next MFMA's operand"), and the
// output accumulators hold 4 consecutive head-dim elements per lane (8-byte stores).  The operand that must be read
// along its row index (V and K for O / dQ; dO and Q for dV / dK) comes from the same row-major LDS image through the
// gfx950 transposing read ds_read_b64_tr_b16 -- no transposed copies in LDS or HBM.
// Partial rotary is applied while q / k rows are staged (on load), its transpose while dq / dk are stored.
// Masking: fully causal over S = P + T, key padding from attention_mask (left-padded text); only tiles that touch the
// diagonal or the text range pay for the mask.
#include <type_traits>

#include "attn.h"

namespace mafed {

namespace {

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

// [rows][D] bf16 image, 16-byte chunk swizzle (row reads with ds_read_b128 and transposing reads share it).
// swz<D>(row) only moves aligned 32-byte chunk PAIRS (its low bit is 0), by a value that differs over the rows of every aligned
// group of 8 that share a 256-byte bank row -- D = 64: two rows per bank row, pair index ^ (row >> 1 & 3); D = 128: one row per bank
// row, pair index ^ (row & 7).  Then the 8 x 32-byte pieces of a ds_read_b64_tr_b16 lane group fall on 8 distinct 32-byte slots and
// the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15} at chunk c, rows {4-11} at chunk c ^ 1) on 16 distinct 16-byte slots:
// SQ_LDS_BANK_CONFLICT 0 for both (the round-1 functions (row >> 1) & 7 and (row & 3) << 2 | (row >> 2) & 3 paired rows j and j + 2
// resp. j + 4 on the same slot: every transposing read took two LDS cycles per lane group, profiles/r01_attn_pmc.json).
template <int D>
__device__ __forceinline__ int swz(int row) {
  if (D == 64) return ((row >> 1) & 3) << 1;
  if (D == 128) return (row & 7) << 1;
  return ((row & 7) << 1) | ((row >> 3) & 1);
}
template <int D>
__device__ __forceinline__ int tile_off(int row, int chunk) {
  if (D == 64) return row * 128 + ((chunk ^ swz<64>(row)) << 4);
  if (D == 128) return row * 256 + ((chunk ^ swz<128>(row)) << 4);
  // D = 256 (VLPythia-1B): 512-byte rows, the low four chunk bits XORed with f(row) = (row & 7) << 1 | (row >> 3) & 1 -- a bijection
  // on every aligned group of 16 rows (row reads of one chunk hit 16 distinct 16-byte slots) whose upper three bits differ over
  // every aligned group of 8 rows (the 8 x 32-byte pieces of a transposing read hit 8 distinct 32-byte slots)
  return row * 512 + ((chunk ^ (((row & 7) << 1) | ((row >> 3) & 1))) << 4);
}

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(f[0]) | ((uint32_t)f32_to_bf16(f[1]) << 16);
  r.y = (uint32_t)f32_to_bf16(f[2]) | ((uint32_t)f32_to_bf16(f[3]) << 16);
  r.z = (uint32_t)f32_to_bf16(f[4]) | ((uint32_t)f32_to_bf16(f[5]) << 16);
  r.w = (uint32_t)f32_to_bf16(f[6]) | ((uint32_t)f32_to_bf16(f[7]) << 16);
  return r;
}

// One 16-byte chunk (8 head-dim elements starting at chunk*8) of row `pos` of a q or k matrix, rotated (tf:111-151).
// rowp points at element 0 of that row; rows >= S give zeros.
__device__ __forceinline__ uint4 load_chunk_rot(const bf16_t* __restrict__ rowp, int chunk, int rot, const float* __restrict__ rc,
                                                const float* __restrict__ rs, int pos, bool valid) {
  if (!valid) return make_uint4(0u, 0u, 0u, 0u);
  const uint4 x = *reinterpret_cast<const uint4*>(rowp + chunk * 8);
  if (chunk * 8 >= rot) return x;
  const int hc = rot >> 4;  // chunks per rotary half
  const uint4 y = *reinterpret_cast<const uint4*>(rowp + (chunk ^ hc) * 8);
  const bool first = chunk < hc;
  const int d0 = (first ? chunk : chunk - hc) * 8;
  const float* c = rc + (int64_t)pos * (rot >> 1) + d0;
  const float* s = rs + (int64_t)pos * (rot >> 1) + d0;
  float xf[8], yf[8], o[8];
  unpack8(x, xf);
  unpack8(y, yf);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = first ? (xf[e] * c[e] - yf[e] * s[e]) : (xf[e] * c[e] + yf[e] * s[e]);
  return pack8(o);
}

// Stage `NROWS` rows (global rows r0 .. r0+NROWS-1 of one (b,h) matrix, element stride rstride) into an LDS image.
template <int D, int NROWS, bool ROT>
__device__ __forceinline__ void stage_rows(char* __restrict__ img, const bf16_t* __restrict__ base, int64_t rstride, int r0, int S, int rot,
                                           const float* __restrict__ rc, const float* __restrict__ rs, int tid) {
  constexpr int CPR = D / 8;  // chunks per row
#pragma unroll
  for (int c = tid; c < NROWS * CPR; c += 256) {
    const int row = c / CPR, ch = c % CPR;
    const int gr = r0 + row;
    uint4 v;
    if (ROT) v = load_chunk_rot(base + (int64_t)gr * rstride, ch, rot, rc, rs, gr, gr < S);
    else v = gr < S ? *reinterpret_cast<const uint4*>(base + (int64_t)gr * rstride + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(img + tile_off<D>(row, ch)) = v;
  }
}

// row-read fragment: lane gets X[row = rt*16 + (lane&15)][d = ks*32 + 8*(lane>>4) + e], e = 0..7
template <int D>
__device__ __forceinline__ bf16x8 frag_row(const char* __restrict__ img, int rt, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + tile_off<D>(rt * 16 + (lane & 15), ks * 4 + (lane >> 4)));
}
// transposed fragment for the k-step kk (32 rows): lane (i = lane&15, g = lane>>4) gets X[row(g,e)][d = dt*16 + i] with
// row(g,e) = kk*32 + 4g + e (e < 4) or kk*32 + 16 + 4g + (e-4): the k order in which accumulator tiles 2kk, 2kk+1 pack.
template <int D>
__device__ __forceinline__ bf16x8 frag_tr(const char* __restrict__ img, int dt, int kk, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r_lo = kk * 32 + 4 * g + q, r_hi = r_lo + 16;
  const int ch = 2 * dt + (p >> 1), sub = (p & 1) * 8;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + tile_off<D>(r_lo, ch) + sub));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + tile_off<D>(r_hi, ch) + sub));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

__device__ __forceinline__ bf16x8 pack_acc(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
  r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
  return r;
}

// column-operand fragment straight from global: lane gets X[row0 + (lane&15)][d = ks*32 + 8*(lane>>4) + e] (rotated if ROT)
template <int D, bool ROT>
__device__ __forceinline__ void load_col_frags(bf16x8 (&f)[D / 32], const bf16_t* __restrict__ base, int64_t rstride, int row0, int S, int rot,
                                               const float* __restrict__ rc, const float* __restrict__ rs, int lane) {
  const int r = row0 + (lane & 15);
#pragma unroll
  for (int ks = 0; ks < D / 32; ++ks) {
    const int ch = ks * 4 + (lane >> 4);
    uint4 v;
    if (ROT) v = load_chunk_rot(base + (int64_t)r * rstride, ch, rot, rc, rs, r, r < S);
    else v = r < S ? *reinterpret_cast<const uint4*>(base + (int64_t)r * rstride + ch * 8) : make_uint4(0u, 0u, 0u, 0u);
    f[ks] = __builtin_bit_cast(bf16x8, v);
  }
}

// Two-phase form of load_col_frags for software pipelining across slices: issue the raw 16-byte loads (and the rotary
// partner chunk + cos/sin rows where needed) for the NEXT slice before computing the current one; finish (rotate, cast)
// when the slice starts, a whole slice of MFMAs later.
template <int D>
struct ColRaw {
  uint4 x[D / 32];
  uint4 y;
  float4 c0, c1, s0, s1;
};
template <int D, bool ROT>
__device__ __forceinline__ void col_raw_issue(ColRaw<D>& r, const bf16_t* __restrict__ base, int64_t rstride, int row0, int S, int rot,
                                              const float* __restrict__ rc, const float* __restrict__ rs, int lane) {
  const int row = row0 + (lane & 15), g = lane >> 4;
  const bool valid = row < S;
  const bf16_t* rowp = base + (int64_t)row * rstride;
#pragma unroll
  for (int ks = 0; ks < D / 32; ++ks) r.x[ks] = valid ? *reinterpret_cast<const uint4*>(rowp + (ks * 4 + g) * 8) : make_uint4(0u, 0u, 0u, 0u);
  if (ROT && g * 8 < rot && valid) {
    const int hc = rot >> 4;
    r.y = *reinterpret_cast<const uint4*>(rowp + (g ^ hc) * 8);
    const int d0 = (g < hc ? g : g - hc) * 8;
    const float* c = rc + (int64_t)row * (rot >> 1) + d0;
    const float* sn = rs + (int64_t)row * (rot >> 1) + d0;
    r.c0 = load4(c); r.c1 = load4(c + 4); r.s0 = load4(sn); r.s1 = load4(sn + 4);
  }
}
template <int D, bool ROT>
__device__ __forceinline__ void col_raw_finish(bf16x8 (&f)[D / 32], const ColRaw<D>& r, int row0, int S, int rot, int lane) {
  const int row = row0 + (lane & 15), g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < D / 32; ++ks) f[ks] = __builtin_bit_cast(bf16x8, r.x[ks]);
  if (ROT && g * 8 < rot && row < S) {
    const bool first = g < (rot >> 4);
    float xf[8], yf[8], o[8];
    unpack8(r.x[0], xf);
    unpack8(r.y, yf);
    const float c[8] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w, r.c1.x, r.c1.y, r.c1.z, r.c1.w};
    const float sn[8] = {r.s0.x, r.s0.y, r.s0.z, r.s0.w, r.s1.x, r.s1.y, r.s1.z, r.s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = first ? (xf[e] * c[e] - yf[e] * sn[e]) : (xf[e] * c[e] + yf[e] * sn[e]);
    f[0] = __builtin_bit_cast(bf16x8, pack8(o));
  }
}

// reduce over the four lanes that share lane&15 (the 4 row groups of an accumulator column)
__device__ __forceinline__ float col_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float col_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// store 4 consecutive head-dim elements after undoing the rotary (gradient rows).  acc[dt] holds d = dt*16 + 4g + r.
template <int D>
__device__ __forceinline__ void store_grad_unrot(bf16_t* __restrict__ rowp, f32x4 (&acc)[D / 16], int rot, const float* __restrict__ rc,
                                                 const float* __restrict__ rs, int pos, int lane, bool valid) {
  const int g = lane >> 4;
  const int half = rot >> 1;
  if (rot == 16) {
    // d = 4g + r in tile 0: g in {0,1} first half, g in {2,3} second half; partner is lane ^ 32
    f32x4 mine = acc[0], other;
#pragma unroll
    for (int r = 0; r < 4; ++r) other[r] = __shfl_xor(mine[r], 32, 64);
    if (valid) {
      const bool first = g < 2;
      const int d0 = (first ? 4 * g : 4 * g - 8);
      const float* c = rc + (int64_t)pos * half + d0;
      const float* s = rs + (int64_t)pos * half + d0;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[0][r] = first ? (mine[r] * c[r] + other[r] * s[r]) : (mine[r] * c[r] - other[r] * s[r]);
    }
  } else if (rot >= 32) {
    // half >= 16: partner tile dt + half/16, same lane
    const int ht = half >> 4;
    if (valid) {
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
        if (dt < ht) {
          const int d0 = dt * 16 + 4 * g;
          const float* c = rc + (int64_t)pos * half + d0;
          const float* s = rs + (int64_t)pos * half + d0;
          if (dt + ht < D / 16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = acc[dt][r], b = acc[dt + ht][r];
              acc[dt][r] = a * c[r] + b * s[r];
              acc[dt + ht][r] = b * c[r] - a * s[r];
            }
          }
        }
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt)
      store4(rowp + dt * 16 + 4 * g, make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]));
  }
}

__device__ __forceinline__ bool key_ok(const int64_t* __restrict__ am, int b, int key, int P, int T, int S) {
#if defined(MAFED_ATTN_ABL) && MAFED_ATTN_ABL == 2
  return key < S;
#endif
  return key < P || (key < S && am[(int64_t)b * T + (key - P)] != 0);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// forward: block = 64 query rows of one (b,h), 4 waves x 16 rows; loop over 64-key tiles up to the diagonal
// ------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, AttnShape sh, const float* __restrict__ rc,
                                                            const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                            bf16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 * 64 * D * 2 bytes (64 KiB at D = 256)
  char* kimg = lds;
  char* vimg = lds + 64 * D * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  // heaviest (last) query tiles first: they have the most key tiles
  const int qt = (int)gridDim.x - 1 - (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int q0 = qt * 64 + wave * 16;
  const int myq = q0 + (lane & 15);
  const int g = lane >> 4;

  bf16x8 qf[D / 32];
  load_col_frags<D, true>(qf, qb, rstride, q0, S, rot, rc, rs, lane);

  f32x4 o[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const float scale = rsqrtf((float)D);

  for (int kt = 0; kt <= qt; ++kt) {
    __syncthreads();
    stage_rows<D, 64, true>(kimg, kb, rstride, kt * 64, S, rot, rc, rs, tid);
    stage_rows<D, 64, false>(vimg, vb, rstride, kt * 64, S, rot, rc, rs, tid);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < D / 32; ++ks) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(kimg, j, ks, lane), qf[ks], s[j], 0, 0, 0);
    }
    const bool need_mask = (kt == qt) || (kt * 64 + 63 >= P);
    float tmax = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[j][r] * scale;
        if (need_mask) {
          const int key = kt * 64 + j * 16 + 4 * g + r;
          if (key > myq || !key_ok(am, b, key, P, T, S)) v = -INFINITY;
        }
        s[j][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = col_max(tmax);
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);  // m = -inf on the first tile -> 0
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[j][r] - mn);
        s[j][r] = p;
        ps += p;
      }
    l = l * alpha + ps;
#pragma unroll
    for (int i = 0; i < D / 16; ++i) o[i] *= alpha;
    const bf16x8 p0 = pack_acc(s[0], s[1]), p1 = pack_acc(s[2], s[3]);
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(vimg, dt, 0, lane), p0, o[dt], 0, 0, 0);
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(vimg, dt, 1, lane), p1, o[dt], 0, 0, 0);
    }
  }
  l = col_sum(l);
  if (myq < S) {
    const float inv = 1.0f / l;
    bf16_t* op = out + ((int64_t)b * S + myq) * H * D + (int64_t)h * D;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt)
      store4(op + dt * 16 + 4 * g, make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv));
    if (g == 0) lse[((int64_t)b * H + h) * S + myq] = m + logf(l);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward, query-owned: delta = rowsum(dO * O) and dQ
// ------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                               const bf16_t* __restrict__ dout, const float* __restrict__ lse, AttnShape sh,
                                                               const float* __restrict__ rc, const float* __restrict__ rs,
                                                               const int64_t* __restrict__ am, bf16_t* __restrict__ dqkv,
                                                               float* __restrict__ delta) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 * 64 * D * 2 bytes
  char* kimg = lds;
  char* vimg = lds + 64 * D * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  const int qt = (int)gridDim.x - 1 - (int)blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int q0 = qt * 64 + wave * 16;
  const int myq = q0 + (lane & 15);
  const int g = lane >> 4;
  const int64_t ostride = (int64_t)H * D;
  const bf16_t* ob = out + (int64_t)b * S * ostride + (int64_t)h * D;
  const bf16_t* dob = dout + (int64_t)b * S * ostride + (int64_t)h * D;

  bf16x8 qf[D / 32], dof[D / 32];
  load_col_frags<D, true>(qf, qb, rstride, q0, S, rot, rc, rs, lane);
  load_col_frags<D, false>(dof, dob, ostride, q0, S, 0, rc, rs, lane);
  float dl = 0.f;
  {
    bf16x8 of[D / 32];
    load_col_frags<D, false>(of, ob, ostride, q0, S, 0, rc, rs, lane);
#pragma unroll
    for (int ks = 0; ks < D / 32; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[ks][e];
    dl = col_sum(dl);
  }
  float L = 0.f;
  if (myq < S) {
    L = lse[((int64_t)b * H + h) * S + myq];
    if (g == 0) delta[((int64_t)b * H + h) * S + myq] = dl;
  }
  f32x4 dq[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float scale = rsqrtf((float)D);

  for (int kt = 0; kt <= qt; ++kt) {
    __syncthreads();
    stage_rows<D, 64, true>(kimg, kb, rstride, kt * 64, S, rot, rc, rs, tid);
    stage_rows<D, 64, false>(vimg, vb, rstride, kt * 64, S, rot, rc, rs, tid);
    __syncthreads();
    const bool need_mask = (kt == qt) || (kt * 64 + 63 >= P);
    f32x4 ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < D / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(kimg, j, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(vimg, j, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p = __expf(s[r] * scale - L);
        if (need_mask) {
          const int key = kt * 64 + j * 16 + 4 * g + r;
          if (key > myq || !key_ok(am, b, key, P, T, S)) p = 0.f;
        }
        ds[j][r] = p * (dp[r] - dl) * scale;
      }
    }
    const bf16x8 d0 = pack_acc(ds[0], ds[1]), d1 = pack_acc(ds[2], ds[3]);
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(kimg, dt, 0, lane), d0, dq[dt], 0, 0, 0);
      dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(kimg, dt, 1, lane), d1, dq[dt], 0, 0, 0);
    }
  }
  bf16_t* dqp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)myq * rstride;
  store_grad_unrot<D>(dqp, dq, rot, rc, rs, myq, lane, myq < S);
}

// ------------------------------------------------------------------------------------------------------------
// backward, key-owned: dK and dV of 64 keys (16 per wave); loop over the query tiles at or below the diagonal
// ------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta, AttnShape sh,
                                                                const float* __restrict__ rc, const float* __restrict__ rs,
                                                                const int64_t* __restrict__ am, bf16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 * 64 * D * 2 + 2 * 64 * 4 bytes
  char* qimg = lds;
  char* doimg = lds + 64 * D * 2;
  float* Ls = reinterpret_cast<float*>(lds + 2 * 64 * D * 2);
  float* dls = Ls + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;  // low key tiles (most query tiles) are dispatched first
  const int nqt = (S + 63) / 64;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int64_t ostride = (int64_t)H * D;
  const bf16_t* dob = dout + (int64_t)b * S * ostride + (int64_t)h * D;
  const int k0 = kt * 64 + wave * 16;
  const int mykey = k0 + (lane & 15);
  const int g = lane >> 4;
  const bool mykey_ok = key_ok(am, b, mykey, P, T, S);

  bf16x8 kf[D / 32], vf[D / 32];
  load_col_frags<D, true>(kf, kb, rstride, k0, S, rot, rc, rs, lane);
  load_col_frags<D, false>(vf, vb, rstride, k0, S, 0, rc, rs, lane);
  f32x4 dk[D / 16], dv[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float scale = rsqrtf((float)D);
  const int64_t lbase = ((int64_t)b * H + h) * S;

  for (int qt = kt; qt < nqt; ++qt) {
    __syncthreads();
    stage_rows<D, 64, true>(qimg, qb, rstride, qt * 64, S, rot, rc, rs, tid);
    stage_rows<D, 64, false>(doimg, dob, ostride, qt * 64, S, 0, rc, rs, tid);
    if (tid < 64) {
      const int q = qt * 64 + tid;
      Ls[tid] = q < S ? lse[lbase + q] : 0.f;
      dls[tid] = q < S ? delta[lbase + q] : 0.f;
    }
    __syncthreads();
    f32x4 p[4], ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < D / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(qimg, j, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(doimg, j, ks, lane), vf[ks], dp, 0, 0, 0);
      }
      const float4 Lq = *reinterpret_cast<const float4*>(Ls + j * 16 + 4 * g);
      const float4 Dq = *reinterpret_cast<const float4*>(dls + j * 16 + 4 * g);
      const float La[4] = {Lq.x, Lq.y, Lq.z, Lq.w}, Da[4] = {Dq.x, Dq.y, Dq.z, Dq.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qt * 64 + j * 16 + 4 * g + r;
        float pv = __expf(s[r] * scale - La[r]);
        if (!mykey_ok || mykey > q || q >= S) pv = 0.f;
        p[j][r] = pv;
        ds[j][r] = pv * (dp[r] - Da[r]) * scale;
      }
    }
    const bf16x8 p0 = pack_acc(p[0], p[1]), p1 = pack_acc(p[2], p[3]);
    const bf16x8 d0 = pack_acc(ds[0], ds[1]), d1 = pack_acc(ds[2], ds[3]);
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) {
      dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(doimg, dt, 0, lane), p0, dv[dt], 0, 0, 0);
      dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(doimg, dt, 1, lane), p1, dv[dt], 0, 0, 0);
      dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(qimg, dt, 0, lane), d0, dk[dt], 0, 0, 0);
      dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(qimg, dt, 1, lane), d1, dk[dt], 0, 0, 0);
    }
  }
  bf16_t* dkp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)mykey * rstride + D;
  store_grad_unrot<D>(dkp, dk, rot, rc, rs, mykey, lane, mykey < S);
  if (mykey < S) {
    bf16_t* dvp = dkp + D;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) store4(dvp + dt * 16 + 4 * g, make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]));
  }
}

// ------------------------------------------------------------------------------------------------------------
// "Resident" kernels for short sequences (S = 288 on the MAFED path): ONE block per (batch, head) stages the whole K and
// V (or Q and dO for dK/dV) of that head into LDS once -- 288 x 64 x 2 B x 2 = 72 KiB at D = 64, two blocks per CU --
// and its four waves walk the 16-row slices heaviest first in a boustrophedon order that balances the causal work.
//  * staging is LDS-DMA (global_load_lds_dwordx4, swizzle applied to the source address): every piece of both matrices is
//    in flight at once and no VGPR is touched; the partial rotary is then applied in place on the 2*rot/16 chunks it covers
//  * the key-padding mask becomes an additive 0 / -inf row in LDS, the causal test a compare + select: no branches and no
//    global loads inside the tile loop (the branchy form cost more than the MFMAs); the 16-key sub-tiles of the diagonal
//    tile that lie entirely above the diagonal are skipped
//  * softmax in the exp2 domain (scale * log2 e folded into one multiply; v_exp_f32 is exp2)
//  * the four-lane column reductions use the gfx950 v_permlane{16,32}_swap instead of ds_bpermute
// The tiled kernels above remain the path for sequences whose K/V do not fit.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float col_max_sw(float v) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float col_sum_sw(float v) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// DMA `nrows` rows (multiple of 32) of a [*, D] matrix into the swizzled image; rows >= S re-read row S-1 (finite data that
// the masks turn into exact zeros).  1 KiB per wave instruction = 8 rows (D = 64) or 4 rows (D = 128).
template <int D, int NW = 4>
__device__ __forceinline__ void dma_rows(char* __restrict__ img, const bf16_t* __restrict__ base, int64_t rstride, int nrows, int S, int wave,
                                         int lane) {
  constexpr int RPP = 1024 / (2 * D);
  const int npieces = nrows / RPP;
  for (int j = wave; j < npieces; j += NW) {
    int row, logical;
    if (D == 64) {
      row = j * 8 + (lane >> 3);
      logical = (lane & 7) ^ swz<64>(row);
    } else {
      row = j * 4 + (lane >> 4);
      logical = (lane & 15) ^ swz<128>(row);
    }
    const int srow = row < S ? row : S - 1;
    __builtin_amdgcn_global_load_lds((glb_void_ptr)(base + (int64_t)srow * rstride + logical * 8), (lds_void_ptr)(img + j * 1024), 16, 0, 0);
  }
}

// In-place partial rotary (tf:111-151) of the staged rows: item = (row, chunk c of the first rotary half); the cos / sin
// rows of a thread's first two items are fetched before the DMA wait.
struct RotPre {
  float4 c0, c1, s0, s1;
};
__device__ __forceinline__ void rot_pre_load(RotPre& r, int it, int hc, int rot, const float* __restrict__ rc, const float* __restrict__ rs) {
  const int row = it / hc, c = it - row * hc;
  const float* cp = rc + (int64_t)row * (rot >> 1) + c * 8;
  const float* sp = rs + (int64_t)row * (rot >> 1) + c * 8;
  r.c0 = load4(cp); r.c1 = load4(cp + 4); r.s0 = load4(sp); r.s1 = load4(sp + 4);
}
template <int D>
__device__ __forceinline__ void rot_apply(char* __restrict__ img, int it, int hc, const RotPre& r) {
  const int row = it / hc, c = it - row * hc;
  uint4* p1 = reinterpret_cast<uint4*>(img + tile_off<D>(row, c));
  uint4* p2 = reinterpret_cast<uint4*>(img + tile_off<D>(row, c + hc));
  float x1[8], x2[8], o1[8], o2[8];
  unpack8(*p1, x1);
  unpack8(*p2, x2);
  const float cs[8] = {r.c0.x, r.c0.y, r.c0.z, r.c0.w, r.c1.x, r.c1.y, r.c1.z, r.c1.w};
  const float sn[8] = {r.s0.x, r.s0.y, r.s0.z, r.s0.w, r.s1.x, r.s1.y, r.s1.z, r.s1.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o1[e] = x1[e] * cs[e] - x2[e] * sn[e];
    o2[e] = x2[e] * cs[e] + x1[e] * sn[e];
  }
  *p1 = pack8(o1);
  *p2 = pack8(o2);
}
template <int D, int NT = 256>
__device__ __forceinline__ void rot_fix_rows(char* __restrict__ img, int items, int hc, int rot, const float* __restrict__ rc,
                                             const float* __restrict__ rs, const RotPre (&pre)[2], int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + NT * u < items) rot_apply<D>(img, tid + NT * u, hc, pre[u]);
  for (int it = tid + 2 * NT; it < items; it += NT) {
    RotPre r;
    rot_pre_load(r, it, hc, rot, rc, rs);
    rot_apply<D>(img, it, hc, r);
  }
}

// sum over the 16 lanes of a DPP row; every lane ends with the row total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// slice of wave w in round t: rounds of 4 slices, direction alternating, from the heaviest slice down
__device__ __forceinline__ int snake4(int w, int t) { return t * 4 + ((t & 1) ? 3 - w : w); }
template <int NW>
__device__ __forceinline__ int snake(int w, int t) { return t * NW + ((t & 1) ? NW - 1 - w : w); }

#define MAFED_LOG2E 1.4426950408889634f
#define MAFED_LN2 0.6931471805599453f

// (D = 64: EIGHT waves per block.  The slice loop is a dependency chain -- QK^T MFMAs, the four-lane max, exponentials, PV MFMAs --
// with ~100 VALU instructions per 64-key tile; at 126 VGPRs four waves fit on a SIMD, and twice the waves over the same resident
// K / V image halve each wave's chain and hide twice the latency.  The backward kernels need > 200 VGPRs and stay at four.)
template <int D>
constexpr int attn_fwd_waves() { return D == 64 ? 8 : 4; }

template <int D, bool CAUSAL = true>
__global__ __launch_bounds__(attn_fwd_waves<D>() * 64, (D == 64 ? 2 : 1) * attn_fwd_waves<D>() / 4) void attn_fwd_res_kernel(const bf16_t* __restrict__ qkv, AttnShape sh, const float* __restrict__ rc,
                                                           const float* __restrict__ rs, const int64_t* __restrict__ am,
                                                           bf16_t* __restrict__ out, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NW = attn_fwd_waves<D>(), NTH = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nrows = (S + 31) / 32 * 32;
  char* kimg = lds;
  char* vimg = lds + nrows * D * 2;
  float* kbias = reinterpret_cast<float*>(lds + 2 * nrows * D * 2);
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int nslices = (S + 15) / 16;
  const int hc = rot >> 4, items = S * hc;
#if defined(MAFED_ATTN_ABL) && MAFED_ATTN_ABL == 3
  if (tid == 0) out[((int64_t)b * S) * H * D + (int64_t)h * D] = 0;
  return;
#endif
  dma_rows<D, NW>(kimg, kb, rstride, nrows, S, wave, lane);
#if !defined(MAFED_ATTN_ABL) || MAFED_ATTN_ABL != 4
  dma_rows<D, NW>(vimg, vb, rstride, nrows, S, wave, lane);
#endif
  RotPre pre[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + NTH * u < items) rot_pre_load(pre[u], tid + NTH * u, hc, rot, rc, rs);
  for (int k = tid; k < nrows; k += NTH) kbias[k] = key_ok(am, b, k, P, T, S) ? 0.f : -INFINITY;
  ColRaw<D> qraw;
  {
    const int s0 = nslices - 1 - snake<NW>(wave, 0);
    if (s0 >= 0) col_raw_issue<D, true>(qraw, qb, rstride, s0 * 16, S, rot, rc, rs, lane);
  }
  wait_vm0();
  __syncthreads();
  rot_fix_rows<D, NTH>(kimg, items, hc, rot, rc, rs, pre, tid);
  __syncthreads();
#if defined(MAFED_ATTN_ABL) && (MAFED_ATTN_ABL == 1 || MAFED_ATTN_ABL == 4)
  if (tid < 64) out[((int64_t)b * S + tid) * H * D + (int64_t)h * D] = *reinterpret_cast<const bf16_t*>(kimg + tid * 1024) + *reinterpret_cast<const bf16_t*>(vimg + tid * 1024);
  return;
#endif
  const int g = lane >> 4;
  const float scale2 = rsqrtf((float)D) * MAFED_LOG2E;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
  for (int t = 0;; ++t) {
    const int slice = nslices - 1 - snake<NW>(wave, t);
    if (slice < 0) break;
    const int q0 = slice * 16;
    const int myq = q0 + (lane & 15);
    bf16x8 qf[D / 32];
    col_raw_finish<D, true>(qf, qraw, q0, S, rot, lane);
    {
      const int nxt = nslices - 1 - snake<NW>(wave, t + 1);
      if (nxt >= 0) col_raw_issue<D, true>(qraw, qb, rstride, nxt * 16, S, rot, rc, rs, lane);
    }
    f32x4 o[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY;
    // row sums of p through the matrix pipe: a ones[16 x 32] A operand makes every accumulator row the column sum of the bf16 p tile
    // the PV product consumes (numerator and denominator see the same rounded probabilities); 2 MFMAs per tile on the idle pipe
    // instead of 17 VALU adds in the VALU-bound loop, and no cross-lane reduction at the end
    f32x4 lsum = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int last_kt = CAUSAL ? (q0 + 15) / 64 : (S - 1) / 64;  // bidirectional (CLIP tower): every slice sees every key tile
    // one 64-key tile; MASK = false for tiles that lie wholly below the diagonal and wholly inside the image keys (most of
    // them): no bias read, no compares.  nj = 16-key sub-tiles with a key <= q0 + 15.
    auto tile = [&](const int kt, const int nj, auto mask_tag) {
      constexpr bool MASK = decltype(mask_tag)::value;
      f32x4 s[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!MASK || j < nj) {
#pragma unroll
          for (int ks = 0; ks < D / 32; ++ks)
            s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(kimg, kt * 4 + j, ks, lane), qf[ks], s[j], 0, 0, 0);
        }
      }
      float tmax = -INFINITY;
      if (MASK) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < nj) {
            const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kt * 64 + j * 16 + 4 * g);
            const float kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 64 + j * 16 + 4 * g + r;
              float v = s[j][r] * scale2 + kbv[r];
              if (CAUSAL) v = key > myq ? -INFINITY : v;
              s[j][r] = v;
              tmax = fmaxf(tmax, v);
            }
          } else {
            s[j] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          }
        }
      } else {
        // plain tile: the scale is folded into the exponent's FMA below; the maximum commutes with the positive scale
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[j][r]);
        tmax *= scale2;
      }
      tmax = col_max_sw(tmax);
      // Lazy rescale: the running reference m only moves when some row's maximum outgrew it by more than 2^8 (exact
      // arithmetic either way -- p and l are scaled by the same 2^(m_true - m)); the output accumulators are then touched
      // by the VALU a couple of times per slice instead of once per tile.
      if (__builtin_amdgcn_ballot_w64(tmax - m > 8.0f) != 0) {
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) o[i] *= alpha;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[j][r] = __builtin_amdgcn_exp2f(MASK ? s[j][r] - m : fmaf(s[j][r], scale2, -m));
      const bf16x8 p0 = pack_acc(s[0], s[1]);
      lsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p0, lsum, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(vimg, dt, kt * 2, lane), p0, o[dt], 0, 0, 0);
      if (!MASK || nj > 2) {
        const bf16x8 p1 = pack_acc(s[2], s[3]);
        lsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, p1, lsum, 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(vimg, dt, kt * 2 + 1, lane), p1, o[dt], 0, 0, 0);
      }
    };
    // tiles [0, kt_plain): every key < P and (causal) < q0
    const int kt_plain = CAUSAL ? (last_kt < P / 64 ? last_kt : P / 64) : (P / 64 <= last_kt ? P / 64 : last_kt + 1);
    const int klast = CAUSAL ? q0 + 15 : S - 1;  // last key any row of the slice may attend to
    for (int kt = 0; kt < kt_plain; ++kt) tile(kt, 4, std::false_type{});
    for (int kt = kt_plain; kt <= last_kt; ++kt) tile(kt, kt == last_kt ? ((klast - kt * 64) >> 4) + 1 : 4, std::true_type{});
    const float l = lsum[0];
    if (myq < S) {
      const float inv = 1.0f / l;
      bf16_t* op = out + ((int64_t)b * S + myq) * H * D + (int64_t)h * D;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt)
        store4(op + dt * 16 + 4 * g, make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv));
      if (g == 0) lse[((int64_t)b * H + h) * S + myq] = m * MAFED_LN2 + logf(l);
    }
  }
}

template <int D, int NW = 4>
__global__ __launch_bounds__(NW * 64, (D == 64 ? 2 : 1) * NW / 4) void attn_bwd_dq_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                              const bf16_t* __restrict__ dout, const float* __restrict__ lse, AttnShape sh,
                                                              const float* __restrict__ rc, const float* __restrict__ rs,
                                                              const int64_t* __restrict__ am, bf16_t* __restrict__ dqkv,
                                                              float* __restrict__ delta, float* __restrict__ bsum) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NTH = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nrows = (S + 31) / 32 * 32;
  char* kimg = lds;
  char* vimg = lds + nrows * D * 2;
  float* kbias = reinterpret_cast<float*>(lds + 2 * nrows * D * 2);
  float* red = kbias + 2 * nrows;  // [D] column sums of the stored dq rows (query_key_value.bias gradient), when asked for
  if (bsum && tid < D) red[tid] = 0.f;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int64_t ostride = (int64_t)H * D;
  const bf16_t* ob = out + (int64_t)b * S * ostride + (int64_t)h * D;
  const bf16_t* dob = dout + (int64_t)b * S * ostride + (int64_t)h * D;
  const float* Lrow = lse + ((int64_t)b * H + h) * S;
  const int nslices = (S + 15) / 16;
  const int hc = rot >> 4, items = S * hc;
  dma_rows<D, NW>(kimg, kb, rstride, nrows, S, wave, lane);
  dma_rows<D, NW>(vimg, vb, rstride, nrows, S, wave, lane);
  RotPre pre[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + NTH * u < items) rot_pre_load(pre[u], tid + NTH * u, hc, rot, rc, rs);
  for (int k = tid; k < nrows; k += NTH) kbias[k] = key_ok(am, b, k, P, T, S) ? 0.f : -INFINITY;
  // NW = 4: the next slice's q / dO / O rows are requested a slice ahead (84 VGPRs of raw data).  NW = 8: four waves per SIMD hide
  // that latency themselves, and the registers are what lets them be resident
  constexpr bool PREFETCH = NW == 4;
  ColRaw<D> qraw, doraw, oraw;
  float Lnext = 0.f;
  if (PREFETCH) {
    const int s0 = nslices - 1 - snake<NW>(wave, 0);
    if (s0 >= 0) {
      col_raw_issue<D, true>(qraw, qb, rstride, s0 * 16, S, rot, rc, rs, lane);
      col_raw_issue<D, false>(doraw, dob, ostride, s0 * 16, S, 0, rc, rs, lane);
      col_raw_issue<D, false>(oraw, ob, ostride, s0 * 16, S, 0, rc, rs, lane);
      const int q = s0 * 16 + (lane & 15);
      Lnext = q < S ? Lrow[q] : 0.f;
    }
  }
  wait_vm0();
  __syncthreads();
  rot_fix_rows<D, NTH>(kimg, items, hc, rot, rc, rs, pre, tid);
  __syncthreads();
  const int g = lane >> 4;
  const float scale = rsqrtf((float)D);
  const float scale2 = scale * MAFED_LOG2E;
  for (int t = 0;; ++t) {
    const int slice = nslices - 1 - snake<NW>(wave, t);
    if (slice < 0) break;
    const int q0 = slice * 16;
    const int myq = q0 + (lane & 15);
    if (!PREFETCH) {
      col_raw_issue<D, true>(qraw, qb, rstride, q0, S, rot, rc, rs, lane);
      col_raw_issue<D, false>(doraw, dob, ostride, q0, S, 0, rc, rs, lane);
      col_raw_issue<D, false>(oraw, ob, ostride, q0, S, 0, rc, rs, lane);
      Lnext = myq < S ? Lrow[myq] : 0.f;
    }
    bf16x8 qf[D / 32], dof[D / 32];
    col_raw_finish<D, true>(qf, qraw, q0, S, rot, lane);
    col_raw_finish<D, false>(dof, doraw, q0, S, 0, lane);
    const float L2 = Lnext * MAFED_LOG2E;
    float dl = 0.f;
    {
      bf16x8 of[D / 32];
      col_raw_finish<D, false>(of, oraw, q0, S, 0, lane);
      const int nxt = nslices - 1 - snake<NW>(wave, t + 1);
      if (PREFETCH && nxt >= 0) {
        col_raw_issue<D, true>(qraw, qb, rstride, nxt * 16, S, rot, rc, rs, lane);
        col_raw_issue<D, false>(doraw, dob, ostride, nxt * 16, S, 0, rc, rs, lane);
        col_raw_issue<D, false>(oraw, ob, ostride, nxt * 16, S, 0, rc, rs, lane);
        const int q = nxt * 16 + (lane & 15);
        Lnext = q < S ? Lrow[q] : 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < D / 32; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[ks][e];
      dl = col_sum_sw(dl);
    }
    if (myq < S && g == 0) delta[((int64_t)b * H + h) * S + myq] = dl;
    f32x4 dq[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int last_kt = (q0 + 15) / 64;
    auto tile = [&](const int kt, const int nj, auto mask_tag) {
      constexpr bool MASK = decltype(mask_tag)::value;
      f32x4 ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ds[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!MASK || j < nj) {
          f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < D / 32; ++ks) {
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(kimg, kt * 4 + j, ks, lane), qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(vimg, kt * 4 + j, ks, lane), dof[ks], dp, 0, 0, 0);
          }
          if (MASK) {
            const float4 kb4 = *reinterpret_cast<const float4*>(kbias + kt * 64 + j * 16 + 4 * g);
            const float kbv[4] = {kb4.x, kb4.y, kb4.z, kb4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = kt * 64 + j * 16 + 4 * g + r;
              float x = s[r] * scale2 - L2 + kbv[r];
              x = key > myq ? -INFINITY : x;
              ds[j][r] = __builtin_amdgcn_exp2f(x) * (dp[r] - dl) * scale;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[j][r] = __builtin_amdgcn_exp2f(s[r] * scale2 - L2) * (dp[r] - dl) * scale;
          }
        }
      }
      const bf16x8 d0 = pack_acc(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(kimg, dt, kt * 2, lane), d0, dq[dt], 0, 0, 0);
      if (!MASK || nj > 2) {
        const bf16x8 d1 = pack_acc(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(kimg, dt, kt * 2 + 1, lane), d1, dq[dt], 0, 0, 0);
      }
    };
    const int kt_plain = last_kt < P / 64 ? last_kt : P / 64;
    for (int kt = 0; kt < kt_plain; ++kt) tile(kt, 4, std::false_type{});
    for (int kt = kt_plain; kt <= last_kt; ++kt) tile(kt, kt == last_kt ? ((q0 + 15 - kt * 64) >> 4) + 1 : 4, std::true_type{});
    bf16_t* dqp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)myq * rstride;
    store_grad_unrot<D>(dqp, dq, rot, rc, rs, myq, lane, myq < S);
    if (bsum) {  // fold the slice's 16 queries (one DPP row), accumulate in LDS: no registers live across slices
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = row16_sum(myq < S ? dq[dt][r] : 0.f);
          if ((lane & 15) == 0) atomicAdd(red + dt * 16 + 4 * g + r, v);
        }
    }
  }
  if (bsum) {  // one 4*D-byte run of atomics per block
    __syncthreads();
    if (tid < D) atomicAdd(bsum + (int64_t)h * 3 * D + tid, red[tid]);
  }
}

// dK / dV: Q (rotated) and dO of the head are resident, with the rows' log2-domain LSE and delta beside them; each wave
// owns 16-key slices, heaviest (lowest keys) first
// (two waves per SIMD at D = 64 -- the LDS footprint allows two blocks per CU, so the register budget is capped to match)
template <int D, int NW = 4>
__global__ __launch_bounds__(NW * 64, (D == 64 ? 2 : 1) * NW / 4) void attn_bwd_dkv_res_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta, AttnShape sh,
                                                               const float* __restrict__ rc, const float* __restrict__ rs,
                                                               const int64_t* __restrict__ am, bf16_t* __restrict__ dqkv,
                                                               float* __restrict__ bsum) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NTH = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int S = sh.S, H = sh.H, rot = sh.rot, P = sh.P, T = sh.T;
  const int h = blockIdx.x, b = blockIdx.y;
  const int nrows = (S + 31) / 32 * 32;
  char* qimg = lds;
  char* doimg = lds + nrows * D * 2;
  float* L2s = reinterpret_cast<float*>(lds + 2 * nrows * D * 2);
  float* Ds = L2s + nrows;
  float* red = Ds + nrows;  // [dk D | dv D] column sums of the stored rows, when asked for
  if (bsum && tid < 2 * D) red[tid] = 0.f;
  const int64_t rstride = (int64_t)H * 3 * D;
  const bf16_t* qb = qkv + ((int64_t)b * S * H + h) * 3 * D;
  const bf16_t* kb = qb + D;
  const bf16_t* vb = qb + 2 * D;
  const int64_t ostride = (int64_t)H * D;
  const bf16_t* dob = dout + (int64_t)b * S * ostride + (int64_t)h * D;
  const float* Lrow = lse + ((int64_t)b * H + h) * S;
  const float* Drow = delta + ((int64_t)b * H + h) * S;
  const int nslices = (S + 15) / 16;
  const int hc = rot >> 4, items = S * hc;
  dma_rows<D, NW>(qimg, qb, rstride, nrows, S, wave, lane);
  dma_rows<D, NW>(doimg, dob, ostride, nrows, S, wave, lane);
  RotPre pre[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + NTH * u < items) rot_pre_load(pre[u], tid + NTH * u, hc, rot, rc, rs);
  for (int q = tid; q < nrows; q += NTH) {
    L2s[q] = q < S ? Lrow[q] * MAFED_LOG2E : INFINITY;  // rows past S: exp2(x - inf) = 0
    Ds[q] = q < S ? Drow[q] : 0.f;
  }
  ColRaw<D> kraw, vraw;
  {
    const int s0 = snake<NW>(wave, 0);
    if (s0 < nslices) {
      col_raw_issue<D, true>(kraw, kb, rstride, s0 * 16, S, rot, rc, rs, lane);
      col_raw_issue<D, false>(vraw, vb, rstride, s0 * 16, S, 0, rc, rs, lane);
    }
  }
  wait_vm0();
  __syncthreads();
  rot_fix_rows<D, NTH>(qimg, items, hc, rot, rc, rs, pre, tid);
  __syncthreads();
  const int g = lane >> 4;
  const float scale = rsqrtf((float)D);
  const float scale2 = scale * MAFED_LOG2E;
  const int nqt = (nrows + 63) / 64;
  for (int t = 0;; ++t) {
    const int slice = snake<NW>(wave, t);
    if (slice >= nslices) break;
    const int k0 = slice * 16;
    const int mykey = k0 + (lane & 15);
    const int mykey_eff = key_ok(am, b, mykey, P, T, S) ? mykey : 0x7fffffff;  // a padded key is "after" every query
    bf16x8 kf[D / 32], vf[D / 32];
    col_raw_finish<D, true>(kf, kraw, k0, S, rot, lane);
    col_raw_finish<D, false>(vf, vraw, k0, S, 0, lane);
    {
      const int nxt = snake<NW>(wave, t + 1);
      if (nxt < nslices) {
        col_raw_issue<D, true>(kraw, kb, rstride, nxt * 16, S, rot, rc, rs, lane);
        col_raw_issue<D, false>(vraw, vb, rstride, nxt * 16, S, 0, rc, rs, lane);
      }
    }
    f32x4 dk[D / 16], dv[D / 16];
#pragma unroll
    for (int i = 0; i < D / 16; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int qt0 = k0 / 64;
    const bool text_keys = k0 + 15 >= P;
    // one 64-query tile; EDGE = the diagonal tile, the ragged last tile, or any tile of a slice that holds text keys
    // (sub-tile skipping + causal / padding select); plain tiles below the diagonal need neither
    auto tile = [&](const int qt, const int jlo, const int jhi, auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      f32x4 p[4], ds[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ds[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!EDGE || (j >= jlo && j < jhi)) {
          f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < D / 32; ++ks) {
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(qimg, qt * 4 + j, ks, lane), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<D>(doimg, qt * 4 + j, ks, lane), vf[ks], dp, 0, 0, 0);
          }
          const float4 L4 = *reinterpret_cast<const float4*>(L2s + qt * 64 + j * 16 + 4 * g);
          const float4 D4 = *reinterpret_cast<const float4*>(Ds + qt * 64 + j * 16 + 4 * g);
          const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Dv[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = s[r] * scale2 - Lv[r];
            if (EDGE) {
              const int q = qt * 64 + j * 16 + 4 * g + r;
              x = mykey_eff > q ? -INFINITY : x;
            }
            const float pv = __builtin_amdgcn_exp2f(x);
            p[j][r] = pv;
            ds[j][r] = pv * (dp[r] - Dv[r]) * scale;
          }
        }
      }
      if (!EDGE || jlo < 2) {
        const bf16x8 p0 = pack_acc(p[0], p[1]), d0 = pack_acc(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(doimg, dt, qt * 2, lane), p0, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(qimg, dt, qt * 2, lane), d0, dk[dt], 0, 0, 0);
        }
      }
      if (!EDGE || jhi > 2) {
        const bf16x8 p1 = pack_acc(p[2], p[3]), d1 = pack_acc(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < D / 16; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(doimg, dt, qt * 2 + 1, lane), p1, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<D>(qimg, dt, qt * 2 + 1, lane), d1, dk[dt], 0, 0, 0);
        }
      }
    };
    const int nfull = nrows / 64;  // tiles [0, nfull) have all four query sub-tiles inside the image
    for (int qt = qt0; qt < nqt; ++qt) {
      const int jlo = qt == qt0 ? (k0 - qt * 64) >> 4 : 0;           // query sub-tiles entirely before the slice's keys are skipped
      const int jhi = (nrows - qt * 64) >> 4 < 4 ? (nrows - qt * 64) >> 4 : 4;
      if (qt == qt0 || text_keys || qt >= nfull) tile(qt, jlo, jhi, std::true_type{});
      else tile(qt, 0, 4, std::false_type{});
    }
    bf16_t* dkp = dqkv + ((int64_t)b * S * H + h) * 3 * D + (int64_t)mykey * rstride + D;
    store_grad_unrot<D>(dkp, dk, rot, rc, rs, mykey, lane, mykey < S);
    if (mykey < S) {
      bf16_t* dvp = dkp + D;
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) store4(dvp + dt * 16 + 4 * g, make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]));
    }
    if (bsum) {
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vk = row16_sum(mykey < S ? dk[dt][r] : 0.f), vv = row16_sum(mykey < S ? dv[dt][r] : 0.f);
          if ((lane & 15) == 0) {
            atomicAdd(red + dt * 16 + 4 * g + r, vk);
            atomicAdd(red + D + dt * 16 + 4 * g + r, vv);
          }
        }
    }
  }
  if (bsum) {
    __syncthreads();
    if (tid < 2 * D) atomicAdd(bsum + (int64_t)h * 3 * D + D + tid, red[tid]);
  }
}

// algorithmic forward flops of one launch: QK^T and PV over the causal half, 4 D S (S + 1) / 2 per (batch, head);
// the backward's 2x is booked half on the dQ kernel (dP, dQ) and half on the dK/dV kernel (dK, dV)
static double attn_fwd_flops(const AttnShape& sh) { return 4.0 * sh.D * ((double)sh.S * (sh.S + 1) / 2.0) * sh.H * sh.B; }

static bool attn_resident_fits(const AttnShape& sh, size_t* bytes) {
  const size_t nrows = (size_t)(sh.S + 31) / 32 * 32;
  *bytes = nrows * sh.D * 2 * 2 + nrows * 4 * 2 + (size_t)sh.D * 4 * 2;  // two operand images + two fp32 rows (key bias | LSE, delta) + the column-sum row
  return sh.D <= 128 && *bytes <= 160 * 1024;  // (D = 256: 295 KiB at S = 288 -- K / V go through LDS in 64-row tiles instead)
}
static int g_attn_variant = 0;  // 0 automatic, 1 force the tiled kernels (tests)

template <typename K>
static void set_lds_attr(K kfn, size_t bytes) {
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

int attn_mfma_fwd_launch(const void* qkv, const AttnShape& sh, const float* rc, const float* rs, const int64_t* am, void* out, float* lse,
                         hipStream_t st) {
  size_t bytes;
  if (g_attn_variant != 1 && attn_resident_fits(sh, &bytes)) {
    dim3 grid(sh.H, sh.B), block(attn_fwd_waves<64>() * 64);
    if (!sh.causal) {
      if (sh.D != 64) { set_error("attn_fwd (bidirectional, bf16): head_dim %d has no MFMA kernel (CLIP towers use 64)", sh.D); return MAFED_EINVAL; }
      set_lds_attr(attn_fwd_res_kernel<64, false>, bytes);
      launch(K_ATTN_FWD, 2.0 * attn_fwd_flops(sh), attn_fwd_res_kernel<64, false>, grid, block, bytes, st, (const bf16_t*)qkv, sh, rc, rs, am,
             (bf16_t*)out, lse);
      return MAFED_OK;
    }
    if (sh.D == 64) {
      set_lds_attr(attn_fwd_res_kernel<64>, bytes);
      launch(K_ATTN_FWD, attn_fwd_flops(sh), attn_fwd_res_kernel<64>, grid, block, bytes, st, (const bf16_t*)qkv, sh, rc, rs, am, (bf16_t*)out, lse);
    } else {
      set_lds_attr(attn_fwd_res_kernel<128>, bytes);
      launch(K_ATTN_FWD, attn_fwd_flops(sh), attn_fwd_res_kernel<128>, grid, dim3(attn_fwd_waves<128>() * 64), bytes, st, (const bf16_t*)qkv, sh, rc, rs, am, (bf16_t*)out, lse);
    }
    return MAFED_OK;
  }
  if (!sh.causal) { set_error("attn_fwd (bidirectional, bf16): S=%d D=%d does not fit the resident kernel", sh.S, sh.D); return MAFED_EINVAL; }
  dim3 grid((sh.S + 63) / 64, sh.H, sh.B), block(256);
  const size_t tb = (size_t)2 * 64 * sh.D * 2;
#define MAFED_FWD_TILED(DD)                                                                                                       \
  do {                                                                                                                            \
    set_lds_attr(attn_fwd_mfma_kernel<DD>, tb);                                                                                   \
    launch(K_ATTN_FWD, attn_fwd_flops(sh), attn_fwd_mfma_kernel<DD>, grid, block, tb, st, (const bf16_t*)qkv, sh, rc, rs, am, (bf16_t*)out, lse);                    \
  } while (0)
  if (sh.D == 64) MAFED_FWD_TILED(64);
  else if (sh.D == 128) MAFED_FWD_TILED(128);
  else MAFED_FWD_TILED(256);
#undef MAFED_FWD_TILED
  return MAFED_OK;
}

int attn_mfma_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, const AttnShape& sh, const float* rc,
                         const float* rs, const int64_t* am, void* dqkv, float* delta, float* colsum, bool* colsum_done, hipStream_t st) {
  size_t bytes;
  *colsum_done = false;
  if (g_attn_variant != 1 && attn_resident_fits(sh, &bytes)) {
    dim3 grid(sh.H, sh.B), block(256);
    if (sh.D == 64) {
      set_lds_attr(attn_bwd_dq_res_kernel<64>, bytes);
      set_lds_attr(attn_bwd_dkv_res_kernel<64>, bytes);
      launch(K_ATTN_BWD_DQ, attn_fwd_flops(sh), attn_bwd_dq_res_kernel<64>, grid, block, bytes, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, sh, rc, rs, am,
                                                             (bf16_t*)dqkv, delta, colsum);
      launch(K_ATTN_BWD_DKV, attn_fwd_flops(sh), attn_bwd_dkv_res_kernel<64>, grid, block, bytes, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, sh, rc, rs, am, (bf16_t*)dqkv, colsum);
    } else {
      set_lds_attr(attn_bwd_dq_res_kernel<128>, bytes);
      set_lds_attr(attn_bwd_dkv_res_kernel<128>, bytes);
      launch(K_ATTN_BWD_DQ, attn_fwd_flops(sh), attn_bwd_dq_res_kernel<128>, grid, block, bytes, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, sh, rc, rs, am,
                                                              (bf16_t*)dqkv, delta, colsum);
      launch(K_ATTN_BWD_DKV, attn_fwd_flops(sh), attn_bwd_dkv_res_kernel<128>, grid, block, bytes, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, sh, rc, rs, am, (bf16_t*)dqkv, colsum);
    }
    *colsum_done = colsum != nullptr;
    return MAFED_OK;
  }
  dim3 grid((sh.S + 63) / 64, sh.H, sh.B), block(256);
  const size_t tb = (size_t)2 * 64 * sh.D * 2, tb2 = tb + 2 * 64 * 4;
#define MAFED_BWD_TILED(DD)                                                                                                       \
  do {                                                                                                                            \
    set_lds_attr(attn_bwd_dq_mfma_kernel<DD>, tb);                                                                                \
    set_lds_attr(attn_bwd_dkv_mfma_kernel<DD>, tb2);                                                                              \
    launch(K_ATTN_BWD_DQ, attn_fwd_flops(sh), attn_bwd_dq_mfma_kernel<DD>, grid, block, tb, st, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse, sh, rc, rs, am, \
                                                         (bf16_t*)dqkv, delta);                                                  \
    launch(K_ATTN_BWD_DKV, attn_fwd_flops(sh), attn_bwd_dkv_mfma_kernel<DD>, grid, block, tb2, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta, sh, rc, rs, am,   \
                                                           (bf16_t*)dqkv);                                                       \
  } while (0)
  if (sh.D == 64) MAFED_BWD_TILED(64);
  else if (sh.D == 128) MAFED_BWD_TILED(128);
  else MAFED_BWD_TILED(256);
#undef MAFED_BWD_TILED
  return MAFED_OK;
}

void attn_mfma_set_variant(int v) { g_attn_variant = v; }


}  // namespace mafed
