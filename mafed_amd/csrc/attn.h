// Shared declarations of the attention kernels (exact-fp32 parity kernels + bf16 MFMA kernels).
#pragma once
#include "common.h"

namespace mafed {

struct AttnShape {
  int B, S, H, D, rot, T, P;
  int causal = 1;  // 0: bidirectional (the frozen CLIP vision tower, forward only)
};

template <typename T>
int attn_ref_fwd_launch(const void* qkv, const AttnShape& sh, const float* rc, const float* rs, const int64_t* am, void* out, float* lse,
                        hipStream_t st);
template <typename T>
int attn_ref_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, const AttnShape& sh, const float* rc,
                        const float* rs, const int64_t* am, void* dqkv, float* delta, hipStream_t st);

template <typename T>
int attn_decode_launch(const void* qkv_pre, int S0, const void* qkv_new, int cap, int t, int B, int H, int D, int rot, int P, int Tm,
                       const float* rc, const float* rs, const int64_t* am, void* out, hipStream_t st, bool prerot = false);
// k part of every row of a [rows / S samples, S, H, 3, D] qkv tensor rotated in place for its position (decode cache)
template <typename T>
int rotate_k_rows_launch(void* qkv, int64_t rows, int S, int H, int D, int rot, const float* rc, const float* rs, hipStream_t st);

int attn_mfma_fwd_launch(const void* qkv, const AttnShape& sh, const float* rc, const float* rs, const int64_t* am, void* out, float* lse,
                         hipStream_t st);
int attn_mfma_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, const AttnShape& sh, const float* rc,
                         const float* rs, const int64_t* am, void* dqkv, float* delta, float* colsum, bool* colsum_done, hipStream_t st);

void attn_mfma_set_variant(int v);  // 0 automatic (resident kernels when K/V fit in LDS), 1 tiled kernels only

int attn_decode_set_trace(void* buf);   // tools: [B * H][8] stamps of the next flat decode attention launches

}  // namespace mafed
