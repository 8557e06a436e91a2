// C-ABI entry points of the attention path: argument validation + dispatch (bf16 -> MFMA kernels, f32 -> parity kernels).
#include "attn.h"

using namespace mafed;

static int check_common(const char* who, const void* qkv, int B, int S, int H, int D, int rot, const float* rc, const float* rs,
                        const int64_t* am, int T) {
  MAFED_CHECK_ARG(qkv && am, "%s: null pointer", who);
  MAFED_CHECK_ARG(B > 0 && S > 0 && H > 0 && D > 0 && T >= 0 && T <= S, "%s: bad shape B=%d S=%d H=%d D=%d T=%d", who, B, S, H, D, T);
  MAFED_CHECK_ARG(rot >= 0 && rot <= D && rot % 2 == 0, "%s: rotary dims %d invalid for D=%d", who, rot, D);
  MAFED_CHECK_ARG(rot == 0 || (rc && rs), "%s: rotary tables missing", who);
  return MAFED_OK;
}

static bool mfma_ok(int D, int rot) { return (D == 64 || D == 128 || D == 256) && (rot == 0 || rot == 16 || rot == 32 || rot == 64) && rot * 2 <= D * 2; }

extern "C" int mafed_attn_fwd(const void* qkv, mafed_dtype dtype, int B, int S, int H, int D, int rot, const float* rot_cos,
                              const float* rot_sin, const int64_t* attention_mask, int T, void* out, float* lse, void* stream) {
  int rc = check_common("attn_fwd", qkv, B, S, H, D, rot, rot_cos, rot_sin, attention_mask, T);
  if (rc) return rc;
  MAFED_CHECK_ARG(out && lse, "attn_fwd: null output");
  AttnShape sh{B, S, H, D, rot, T, S - T};
  hipStream_t st = as_stream(stream);
  if (dtype == MAFED_F32) {
    MAFED_CHECK_ARG(D <= 256, "attn_fwd(f32): D=%d > 256", D);
    rc = attn_ref_fwd_launch<float>(qkv, sh, rot_cos, rot_sin, attention_mask, out, lse, st);
  } else if (mfma_ok(D, rot)) {
    MAFED_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "attn_fwd(bf16): qkv/out must be 16-byte aligned");
    rc = attn_mfma_fwd_launch(qkv, sh, rot_cos, rot_sin, attention_mask, out, lse, st);
  } else {
    MAFED_CHECK_ARG(D <= 256, "attn_fwd(bf16): D=%d > 256", D);
    rc = attn_ref_fwd_launch<bf16_t>(qkv, sh, rot_cos, rot_sin, attention_mask, out, lse, st);  // head sizes other than 64 / 128 / 256
  }
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("attn_fwd");
  return MAFED_OK;
}

// Bidirectional attention of the frozen CLIP vision tower (clip:259-277): no mask, no rotary, forward only.
extern "C" int mafed_attn_fwd_bidir(const void* qkv, mafed_dtype dtype, int B, int S, int H, int D, void* out, float* lse, void* stream) {
  MAFED_CHECK_ARG(qkv && out && lse, "attn_fwd_bidir: null pointer");
  MAFED_CHECK_ARG(B > 0 && S > 0 && H > 0 && D > 0 && D <= 256, "attn_fwd_bidir: bad shape B=%d S=%d H=%d D=%d", B, S, H, D);
  AttnShape sh{B, S, H, D, 0, 0, S};
  sh.causal = 0;
  const int64_t* dummy_mask = reinterpret_cast<const int64_t*>(qkv);  // never read: T = 0, every key is an "image" key
  hipStream_t st = as_stream(stream);
  int rc;
  if (dtype == MAFED_F32) {
    rc = attn_ref_fwd_launch<float>(qkv, sh, nullptr, nullptr, dummy_mask, out, lse, st);
  } else {
    MAFED_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "attn_fwd_bidir(bf16): qkv/out must be 16-byte aligned");
    rc = attn_mfma_fwd_launch(qkv, sh, nullptr, nullptr, dummy_mask, out, lse, st);
  }
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("attn_fwd_bidir");
  return MAFED_OK;
}

static int attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype, int B, int S, int H,
                         int D, int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* dqkv,
                         float* delta, float* dqkv_colsum, void* stream);

extern "C" int mafed_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype, int B, int S, int H,
                              int D, int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* dqkv,
                              float* delta, void* stream) {
  return attn_bwd_impl(qkv, out, dout, lse, dtype, B, S, H, D, rot, rot_cos, rot_sin, attention_mask, T, dqkv, delta, nullptr, stream);
}

extern "C" int mafed_attn_bwd_colsum(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype, int B, int S,
                                     int H, int D, int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T,
                                     void* dqkv, float* delta, float* dqkv_colsum, void* stream) {
  return attn_bwd_impl(qkv, out, dout, lse, dtype, B, S, H, D, rot, rot_cos, rot_sin, attention_mask, T, dqkv, delta, dqkv_colsum, stream);
}

static int attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype, int B, int S, int H,
                         int D, int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* dqkv,
                         float* delta, float* dqkv_colsum, void* stream) {
  int rc = check_common("attn_bwd", qkv, B, S, H, D, rot, rot_cos, rot_sin, attention_mask, T);
  if (rc) return rc;
  MAFED_CHECK_ARG(out && dout && lse && dqkv && delta, "attn_bwd: null pointer");
  AttnShape sh{B, S, H, D, rot, T, S - T};
  hipStream_t st = as_stream(stream);
  bool colsum_done = false;  // the resident MFMA kernels fold the column sums of dqkv themselves
  if (dtype == MAFED_F32) {
    MAFED_CHECK_ARG(D <= 256, "attn_bwd(f32): D=%d > 256", D);
    rc = attn_ref_bwd_launch<float>(qkv, out, dout, lse, sh, rot_cos, rot_sin, attention_mask, dqkv, delta, st);
  } else if (mfma_ok(D, rot)) {
    MAFED_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dqkv) & 15) == 0,
                    "attn_bwd(bf16): tensors must be 16-byte aligned");
    rc = attn_mfma_bwd_launch(qkv, out, dout, lse, sh, rot_cos, rot_sin, attention_mask, dqkv, delta, dqkv_colsum, &colsum_done, st);
  } else {
    MAFED_CHECK_ARG(D <= 256, "attn_bwd(bf16): D=%d > 256", D);
    rc = attn_ref_bwd_launch<bf16_t>(qkv, out, dout, lse, sh, rot_cos, rot_sin, attention_mask, dqkv, delta, st);
  }
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("attn_bwd");
  if (dqkv_colsum && !colsum_done)
    return mafed_colsum(dqkv, dtype, (int64_t)B * S, (int64_t)3 * H * D, (int64_t)3 * H * D, dqkv_colsum, nullptr, 0, stream);
  return MAFED_OK;
}

// test / tuning hook: 0 = automatic, 1 = tiled MFMA kernels even when the resident ones fit
extern "C" int mafed_attn_set_variant(int variant) {
  attn_mfma_set_variant(variant);
  return MAFED_OK;
}

// test hook: run the exact kernels on bf16 data (on-GPU cross-check of the MFMA kernels)
extern "C" int mafed_attn_fwd_exact_bf16(const void* qkv, int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                                         const int64_t* attention_mask, int T, void* out, float* lse, void* stream) {
  int rc = check_common("attn_fwd_exact_bf16", qkv, B, S, H, D, rot, rot_cos, rot_sin, attention_mask, T);
  if (rc) return rc;
  AttnShape sh{B, S, H, D, rot, T, S - T};
  rc = attn_ref_fwd_launch<bf16_t>(qkv, sh, rot_cos, rot_sin, attention_mask, out, lse, as_stream(stream));
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("attn_fwd_exact_bf16");
  return MAFED_OK;
}

static int attn_decode_impl(const void* qkv_prefix, int S0, const void* qkv_new, int cap, int t, mafed_dtype dtype, int B, int H, int D,
                            int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* out,
                            void* stream, bool prerot) {
  MAFED_CHECK_ARG(qkv_prefix && qkv_new && out && attention_mask, "attn_decode: null pointer");
  MAFED_CHECK_ARG(B > 0 && H > 0 && D > 0 && D <= 256 && S0 > 0 && T >= 0 && T <= S0 && cap > 0 && t >= 0 && t < cap,
                  "attn_decode: bad shape B=%d H=%d D=%d S0=%d T=%d cap=%d t=%d", B, H, D, S0, T, cap, t);
  MAFED_CHECK_ARG(rot >= 0 && rot <= D && rot % 2 == 0 && (rot == 0 || (rot_cos && rot_sin)), "attn_decode: rotary arguments invalid");
  hipStream_t st = as_stream(stream);
  int rc = dtype == MAFED_F32 ? attn_decode_launch<float>(qkv_prefix, S0, qkv_new, cap, t, B, H, D, rot, S0 - T, T, rot_cos, rot_sin, attention_mask, out, st, prerot)
                              : attn_decode_launch<bf16_t>(qkv_prefix, S0, qkv_new, cap, t, B, H, D, rot, S0 - T, T, rot_cos, rot_sin, attention_mask, out, st, prerot);
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("attn_decode");
  return MAFED_OK;
}

extern "C" int mafed_attn_decode(const void* qkv_prefix, int S0, const void* qkv_new, int cap, int t, mafed_dtype dtype, int B, int H, int D,
                                 int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* out,
                                 void* stream) {
  return attn_decode_impl(qkv_prefix, S0, qkv_new, cap, t, dtype, B, H, D, rot, rot_cos, rot_sin, attention_mask, T, out, stream, false);
}

extern "C" int mafed_attn_decode_prerot(const void* qkv_prefix, int S0, void* qkv_new, int cap, int t, mafed_dtype dtype, int B, int H, int D,
                                        int rot, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T, void* out,
                                        void* stream) {
  MAFED_CHECK_ARG(rot % 16 == 0 && (D == 64 || D == 128 || D == 256), "attn_decode_prerot: needs rot %% 16 == 0 and a head size of 64 / 128 / 256");
  return attn_decode_impl(qkv_prefix, S0, qkv_new, cap, t, dtype, B, H, D, rot, rot_cos, rot_sin, attention_mask, T, out, stream, true);
}

extern "C" int mafed_rotate_k_rows(void* qkv, mafed_dtype dtype, int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                                   void* stream) {
  MAFED_CHECK_ARG(qkv && B > 0 && S > 0 && H > 0 && D > 0 && rot >= 0 && rot <= D && rot % 16 == 0 && (rot == 0 || (rot_cos && rot_sin)),
                  "rotate_k_rows: bad arguments");
  hipStream_t st = as_stream(stream);
  int rc = dtype == MAFED_F32 ? rotate_k_rows_launch<float>(qkv, (int64_t)B * S, S, H, D, rot, rot_cos, rot_sin, st)
                              : rotate_k_rows_launch<bf16_t>(qkv, (int64_t)B * S, S, H, D, rot, rot_cos, rot_sin, st);
  if (rc) return rc;
  MAFED_CHECK_LAUNCH("rotate_k_rows");
  return MAFED_OK;
}

// tools: device buffer of 8 int64 per (batch, head) workgroup that the next mafed_attn_decode_prerot launches (all-rows-in-flight form) fill
// with wall-clock stamps {entered, q | k row rotated, scores done, V sum done, left}; NULL switches it off
extern "C" int mafed_attn_decode_set_trace(void* buf) { return mafed::attn_decode_set_trace(buf); }
