// LDS operand images of the LDS-DMA MFMA GEMM kernels (gemm.hip): swizzles, per-lane DMA source addresses, fragment reads.
#pragma once
#include "common.h"

namespace mafed {

constexpr int BK = 64;

// KC image: [128 rows][64 k] bf16, 128-byte rows, 16-byte chunk index XORed with (row>>1)&7:
// a ds_read_b128 lane group (16 rows x one chunk, two rows per 256-byte bank row) then touches 16 distinct slots.
__device__ __forceinline__ int lds_off_kc(int row, int kchunk) { return row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4); }
// KS image: [64 k][128 rows] bf16, 256-byte k-rows (= all 64 banks), 32-byte chunk (16 rows) index XORed with
// f(k) = (k&3) | ((k>>3)&1)<<2: the 8 k-rows a 32-lane half reads in one ds_read_b64_tr_b16 get 8 distinct chunks.
__device__ __forceinline__ int ks_f(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
__device__ __forceinline__ int lds_off_ks(int k, int r16, int byte_in_32) { return k * 256 + ((r16 ^ ks_f(k)) << 5) + byte_in_32; }

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// per-lane source element offset (without the k-tile advance) of DMA instruction j of an R-row operand tile
template <bool KS, int R>
__device__ __forceinline__ int64_t glds_src_off(int j, int lane, int64_t ld, int64_t r0) {
  if (!KS) {
    const int row = j * 8 + (lane >> 3), phys = lane & 7;
    const int logical = phys ^ ((row >> 1) & 7);
    return (r0 + row) * ld + logical * 8;
  } else {
    constexpr int RB = 2 * R;                  // bytes per k-row of the image
    const int p = j * 1024 + lane * 16;
    const int k = p / RB, within = p % RB;
    const int logical16 = (within >> 5) ^ (ks_f(k) & (R / 16 - 1));
    return (int64_t)k * ld + r0 + logical16 * 16 + ((within >> 4) & 1) * 8;
  }
}
// [k][R rows] image: 32-byte chunk index XOR f(k), masked to the R/16 chunks of a k-row (R = 64: 2-way conflicts remain)
template <int R>
__device__ __forceinline__ int lds_off_ks_r(int k, int r16, int byte_in_32) {
  return k * (2 * R) + ((r16 ^ (ks_f(k) & (R / 16 - 1))) << 5) + byte_in_32;
}

template <bool KS, int R>
__device__ __forceinline__ bf16x8 glds_read_frag(const char* __restrict__ img, int rt, int ks, int lane) {
  if (!KS) {
    const int row = rt * 16 + (lane & 15);
    return *reinterpret_cast<const bf16x8*>(img + lds_off_kc(row, ks * 4 + (lane >> 4)));
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int ka = ks * 32 + 8 * g + q, kb = ka + 4;
    typedef __attribute__((address_space(3))) bf16x4* lptr;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(img + lds_off_ks_r<R>(ka, rt, p * 8)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lptr)(img + lds_off_ks_r<R>(kb, rt, p * 8)));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
}

}  // namespace mafed
