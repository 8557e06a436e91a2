/*
 * mafed_hip.h -- C-ABI of the MI355X-native (gfx950) MAFED training hot path.
 *
 * The reference (MalvinaNikandrou/mafed) is pure Python and has no FFI of its own; the arithmetic of its
 * per-step path lives in torch / transformers / flash-attn calls.  Each entry point below names the reference
 * call site it replaces (paths relative to the reference checkout; "tf:" = transformers
 * models/gpt_neox/modeling_gpt_neox.py).  INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; row-major, contiguous unless an ld is given
 *   - `stream` is a hipStream_t passed as void*; every launch goes on it; nothing allocates, synchronises or throws
 *   - return 0 on success, a negative MAFED_E* code otherwise (mafed_last_error_string() has the detail)
 *   - dtype arguments use mafed_dtype; the fp32 residual stream / statistics / gradients of parameters are float
 *   - parameter-gradient outputs ACCUMULATE (+=): the host zeroes the flat gradient buffer once per
 *     accumulation window (Lightning accumulate_grad_batches semantics, mafed/train.py:286)
 */
#ifndef MAFED_HIP_H
#define MAFED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { MAFED_F32 = 0, MAFED_BF16 = 1 } mafed_dtype;

enum {
  MAFED_OK = 0,
  MAFED_EINVAL = -1,   /* bad shape / alignment / unsupported combination */
  MAFED_ELAUNCH = -2,  /* hipLaunch / hipGetLastError failed */
  MAFED_EWORKSPACE = -3 /* workspace too small */
};

/* GEMM epilogues (mafed_gemm.epilogue) */
enum {
  MAFED_EPI_NONE = 0,
  MAFED_EPI_GELU = 1,     /* C = gelu_erf(acc + bias); aux (if non-NULL) receives the pre-activation acc + bias */
  MAFED_EPI_GELU_BWD = 2, /* C = (acc) * gelu_erf'(aux) ; aux = saved pre-activation, same dtype as C */
  MAFED_EPI_QUICK_GELU = 3, /* C = x * sigmoid(1.702 x), x = acc + bias: MLP activation of the frozen CLIP vision tower (clip:338-350) */
  MAFED_EPI_RES1_BF16 = 0x100, /* flag, OR-ed in: res1 points at bf16 data (the attention branch output under bf16 autocast) */
  MAFED_EPI_NO_PERSISTENT = 0x200, /* flag, OR-ed in: THIS call keeps off the one-block-per-CU persistent kernels (a product that runs beside a
                                      long-resident kernel of another stream, e.g. a collective); per call, no process-wide state */
  MAFED_EPI_TICKETED = 0x400 /* flag, OR-ed in: if THIS call takes a persistent kernel over more than one round of the CUs, the blocks draw
                                their tiles from per-XCD queues instead of walking a static schedule (tolerates CUs held by other streams'
                                kernels: 1.2 - 1.3x instead of 1.7 - 1.9x with 8 - 32 CUs taken; costs a free chip ~2 %) */
};

int mafed_version(void);
const char* mafed_last_error_string(void);

/* ---- dense contractions -------------------------------------------------------------------------------
 * C[M,N] = op(A)[M,K] . op(B)[K,N]  (+ bias[n]) -> epilogue -> (+ res1 + res2) (+ beta * C_old)
 *   transA == 0: A stored [M,K] (lda >= K)     transA == 1: A stored [K,M] (lda >= M)
 *   transB == 0: B stored [K,N] (ldb >= N)     transB == 1: B stored [N,K] (ldb >= K)   (nn.Linear weight)
 * in_dtype MAFED_BF16: MFMA v_mfma_f32_16x16x32_bf16, fp32 accumulate; MAFED_F32: exact fp32 FMA path (parity mode).
 * c_dtype may be F32 or BF16; res1/res2 are fp32 [M,N] (ld = ldc) or NULL (res1 may be bf16: MAFED_EPI_RES1_BF16); beta != 0 requires c_dtype F32.
 * Replaces: every nn.Linear on the path -- query_key_value / dense (tf:192-193,204,233), dense_h_to_4h /
 * dense_4h_to_h + GELU (tf:38-49), vision_embed_tokens (mafed/model/vl_pythia.py:226-234,270), embed_out (:213,310),
 * their autograd backward (dX = dY.W, dW += dY^T.X), and the parallel-residual add (tf:271-274) via res1/res2.
 */
int mafed_gemm(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
               const void* A, int64_t lda, const void* B, int64_t ldb,
               void* C, int64_t ldc, mafed_dtype c_dtype,
               const float* bias, int epilogue, void* aux,
               const float* res1, const float* res2, float beta, void* stream);

/* mafed_gemm that also accumulates the column sums of the stored C: colsum[n] += sum_m C[m,n] (fp32 [N], may be NULL;
 * beta must be 0).  This is the bias gradient of the nn.Linear whose output gradient this GEMM produces
 * (dense_h_to_4h.bias from the GELU' epilogue of dX = dY.W2), fused into the epilogue of the MFMA kernel instead of
 * a second pass over C; where the selected kernel has no fused form the library runs mafed_colsum itself. */
int mafed_gemm_colsum(mafed_dtype in_dtype, int transA, int transB, int64_t M, int64_t N, int64_t K,
                      const void* A, int64_t lda, const void* B, int64_t ldb,
                      void* C, int64_t ldc, mafed_dtype c_dtype,
                      const float* bias, int epilogue, void* aux,
                      const float* res1, const float* res2, float beta, float* colsum, void* stream);

/* Grouped form: n independent products with the same operand layouts (transA / transB), input and output types, each with its own
 * shape, leading dimensions and epilogue (fields as the arguments of mafed_gemm_colsum).  Where every problem tiles the persistent
 * MFMA kernel they run as ONE launch whose tile space is the concatenation of theirs -- the parameter gradients dW += dY^T.X of several
 * layers fill the 256 CUs in whole rounds that way, without split-K -- otherwise as n launches.  Same results as n mafed_gemm_colsum
 * calls.  Replaces: the per-parameter weight-gradient products of autograd's Linear backward (tf:38-49,192-236), batched across layers. */
typedef struct mafed_gemm_problem {
  int64_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  const float* bias; int epilogue; void* aux;
  const float* res1; const float* res2; float beta; float* colsum;
  float* sumsq; /* NULL, or 16 floats: sumsq[k] += the squares of part of the STORED C (fp32 C only; which tile adds to which of the 16 slots
                   is unspecified, their total is sum C^2).  The weight-gradient products of a window's last micro-batch leave the squared
                   norm of the final gradient here, so that the clip's norm pass (mafed_gradnorm_partial, 1.2 GB at 410M) skips the
                   matrices (mafed/train.py:288 clip_grad_norm_). */
} mafed_gemm_problem;
int mafed_gemm_grouped(mafed_dtype in_dtype, int transA, int transB, mafed_dtype c_dtype,
                       const mafed_gemm_problem* problems, int n, void* stream);

/* out[n] += sum_m X[m,n]   (bias gradients; X dtype bf16/f32, out fp32 accumulated with one atomic per column per
 * 256-row block; N, ldx multiples of 4).  workspace is unused (kept for ABI stability; workspace_bytes may be 0). */
size_t mafed_colsum_workspace_bytes(int64_t M, int64_t N);
int mafed_colsum(const void* X, mafed_dtype dtype, int64_t M, int64_t N, int64_t ldx, float* out,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm (tf:243-244,313: nn.LayerNorm, eps 1e-5) ---------------------------------------------------
 * Dual-affine: with use_parallel_residual both LNs of a layer normalise the same tensor (tf:261,272), so x is read
 * once and two outputs are written.  Pass w2 = b2 = y2 = NULL for a single LN (final_layer_norm).
 * x fp32 [rows,h]; y1/y2 in out_dtype; mean/rstd fp32 [rows] saved for backward (may be NULL for inference).
 */
int mafed_layernorm_fwd(const float* x, int64_t rows, int h, float eps,
                        const float* w1, const float* b1, void* y1,
                        const float* w2, const float* b2, void* y2,
                        mafed_dtype out_dtype, float* mean, float* rstd, void* stream);
/* dx = LN'(dy1;w1) + LN'(dy2;w2) + dres ; optional dx_lp = dx cast to dy_dtype (the next GEMM's operand).
 * dy1/dy2 in dy_dtype (dy2, w2, dw2, db2 NULL for single LN); dres fp32 [rows,h] or NULL (the residual-stream
 * gradient flowing around the layer); dx fp32 [rows,h] (may alias dres).  dw1/db1/dw2/db2 fp32 [h], accumulated.
 * Optional fused distillation-gradient injection (the hidden state this LN normalises is a distilled one,
 * mafed/methods/distillation.py:237-249 backward): if teacher != NULL,
 *   dx[row,:] += inj_mul * inj_scale[row_class] * (x - teacher)   with row_class from (row % S): <P image, text-valid, pad=none
 * where inj_scale_dev = {lang, vision} = d(loss)/d(sum_lang d), d(loss)/d(sum_vision d) (coeff * weight / count * upstream grad,
 * left on the device by the loss algebra) and inj_mul = 2/h (MSE).
 * A NEGATIVE inj_mul selects the cosine-distance loss of mafed_distill_fwd(cosine = 1) instead (distillation_loss="cosine",
 * mafed/methods/distillation.py:226-235):
 *   dx[row,:] += |inj_mul| * inj_scale[row_class] * d/dx [ 1 - x.t / sqrt((x.x + 1e-12)(t.t + 1e-12)) ]     (inj_mul = -1 for the plain loss) */
size_t mafed_layernorm_bwd_workspace_bytes(int64_t rows, int h);
int mafed_layernorm_bwd(const void* dy1, const void* dy2, mafed_dtype dy_dtype,
                        const float* x, const float* mean, const float* rstd,
                        const float* w1, const float* w2, int64_t rows, int h,
                        const float* dres, float* dx, void* dx_lp,
                        float* dw1, float* db1, float* dw2, float* db2,
                        const float* teacher, const int64_t* attention_mask, int S, int P, int T,
                        const float* inj_scale_dev /* [2] device: {lang, vision} or NULL */, float inj_mul /* host factor, e.g. 2/h */,
                        float* dxsum_a, float* dxsum_b /* optional fp32 [h]: += column sums of dx (bias grads of the layer below) */,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same backward in two calls, so that the parameter reduction can run on another stream, off the dX chain (the caller orders
 * it behind the row kernel and keeps `workspace` private to this pair):
 *   mafed_layernorm_bwd_rows    dx, dx_lp and the per-block partials of dw / db (/ column sums of dx when want_dxsum) into workspace
 *   mafed_layernorm_bwd_params  workspace -> dw1, db1 (dw2, db2), dxsum_a / dxsum_b; pass dw2 / dxsum_* exactly as "dual" / want_dxsum were */
int mafed_layernorm_bwd_rows(const void* dy1, const void* dy2, mafed_dtype dy_dtype, const float* x, const float* mean, const float* rstd,
                             const float* w1, const float* w2, int64_t rows, int h, const float* dres, float* dx, void* dx_lp,
                             const float* teacher, const int64_t* attention_mask, int S, int P, int T, const float* inj_scale_dev,
                             float inj_mul, int want_dxsum, void* workspace, size_t workspace_bytes, void* stream);
int mafed_layernorm_bwd_params(int64_t rows, int h, float* dw1, float* db1, float* dw2, float* db2, float* dxsum_a, float* dxsum_b,
                               const void* workspace, size_t workspace_bytes, void* stream);

/* ---- attention (tf:154-236 eager path / flash-attn-2 wheel, README.md:16) ------------------------------------
 * qkv: [B,S,H,3,D] exactly as the fused query_key_value GEMM leaves it (per-head {q,k,v} interleave, tf:204-207).
 * Partial rotary (first rot dims, NeoX rotate_half pairing, position = row index, tf:111-151) is applied to q and k
 * on load; rot_cos/rot_sin fp32 [S, rot/2].  Fully causal over S, plus key-padding from attention_mask [B,T]
 * (int64, 1 = valid; LEFT padded so invalid keys are the contiguous range [P, P+pad_b)); P = S - T image keys are
 * always valid.  softmax in fp32; scale = D^-0.5.  out [B,S,H*D] in dtype; lse fp32 [B,H,S] (natural log).
 * dtype BF16: MFMA kernels (D in {64,128}); F32: exact parity kernels (D <= 256).
 */
int mafed_attn_fwd(const void* qkv, mafed_dtype dtype, int B, int S, int H, int D, int rot,
                   const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, int T,
                   void* out, float* lse, void* stream);
/* dqkv [B,S,H,3,D] (dtype) <- d(out); inverse rotation applied to dq, dk.  delta fp32 [B,H,S] scratch. */
int mafed_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype,
                   int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                   const int64_t* attention_mask, int T, void* dqkv, float* delta, void* stream);
/* mafed_attn_bwd that also accumulates dqkv_colsum[3*H*D] += sum over the B*S rows of dqkv: the gradient of
 * query_key_value.bias (tf:192), folded inside the resident MFMA kernels (one run of atomics per block) instead of a
 * second pass over dqkv; other paths run mafed_colsum themselves.  dqkv_colsum may be NULL. */
int mafed_attn_bwd_colsum(const void* qkv, const void* out, const void* dout, const float* lse, mafed_dtype dtype,
                          int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                          const int64_t* attention_mask, int T, void* dqkv, float* delta, float* dqkv_colsum, void* stream);

/* ---- KV-cached greedy decode (mafed/model/vqa_cont_learner.py:260-277 calls
 * generate(max_new_tokens=10, use_cache=False), i.e. ten full forwards over the 256 + T prefix) ---------------------
 * One new query per sample against every earlier key.  qkv_prefix [B,S0,H,3,D] is what the prefill's fused QKV GEMM wrote
 * (k un-rotated; rotation applied on load, position = key index, as in mafed_attn_fwd); qkv_new [B,cap,H,3,D] holds one
 * row per generated token, row t being the current one (its q is the query, position S0 + t; keys = S0 + t + 1).
 * rot_cos / rot_sin cover at least S0 + t + 1 positions.  attention_mask [B,T] is the PROMPT's key-padding mask (P = S0 - T
 * image keys and all generated keys are valid).  out [B,H*D] in dtype. */
int mafed_attn_decode(const void* qkv_prefix, int S0, const void* qkv_new, int cap, int t, mafed_dtype dtype,
                      int B, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                      const int64_t* attention_mask, int T, void* out, void* stream);

/* The same step over a cache of ROTATED keys: qkv_prefix's k part has been rotated in place once behind the prefill
 * (mafed_rotate_k_rows), rows < t of qkv_new by the steps that appended them; this call rotates row t's k (as written by the QKV
 * GEMM), uses it and writes it back rotated.  A step then loads k and v only -- no rotary partner chunk, no cos / sin rows per key
 * (two thirds of the on-load form's load instructions).  rot % 16 == 0, D in {64, 128, 256}.  Same result as mafed_attn_decode up to
 * the rounding of the stored rotated keys (bf16 cache: one more bf16 rounding on the first `rot` dims of k). */
int mafed_attn_decode_prerot(const void* qkv_prefix, int S0, void* qkv_new, int cap, int t, mafed_dtype dtype,
                             int B, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                             const int64_t* attention_mask, int T, void* out, void* stream);
/* k part of every row of a [B,S,H,3,D] qkv tensor rotated in place for its position (row index within the sample): turns a prefill's
 * fused-QKV output into the pre-rotated cache of mafed_attn_decode_prerot.  rot % 16 == 0. */
int mafed_rotate_k_rows(void* qkv, mafed_dtype dtype, int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                        void* stream);

/* ---- fused decode layer (mafed/model/vqa_cont_learner.py:260-277 -> HF greedy search over
 * mafed/model/vl_pythia.py's GPT-NeoX stack, one row per sample and step) ----------------------------------------------
 * A GPT-NeoX layer (parallel residual) of a decode step in three launches: mafed_decode_ln_qkv_fc1, mafed_attn_decode_prerot,
 * mafed_decode_out.  bf16 weights [N, K] row-major, bf16 activations, fp32 residual stream x [M, h], M <= 64 rows.
 * mafed_decode_supported: 1 if the shape is served (h % 256 == 0 with h / 256 in {1, 2, 3, 4, 8}, n1 % 32 == 0, ceil(M/16) * h/256 <= 16). */
int mafed_decode_supported(int M, int h, int n1);
/* qkv_out[m, 0:3h] = LN1(x[m]) . wqkv^T + bqkv   (row m at qkv_out + m * qkv_ld elements: the K/V cache row of this step)
 * a_out[m, 0:n1]   = gelu(LN2(x[m]) . w1^T + b1) (erf form)
 * Both LayerNorms (two-pass statistics in fp32, output rounded to bf16 like mafed_layernorm_fwd) are computed in the prologue. */
int mafed_decode_ln_qkv_fc1(const float* x, int M, int h, float eps, const float* ln1_w, const float* ln1_b, const float* ln2_w,
                            const float* ln2_b, const void* wqkv, const float* bqkv, void* qkv_out, int64_t qkv_ld, const void* w1,
                            const float* b1, int n1, void* a_out, void* stream);
/* out[m, 0:N] = LN(x[m]) . w^T (+ bias, may be NULL): the final LayerNorm folded into the LM head's product (the last two launches of a
 * decode step as one).  bf16 w [N, h] and out (row stride ldo elements); N % 32 == 0; M, h as mafed_decode_supported(M, h, 32). */
int mafed_decode_ln_linear(const float* x, int M, int h, float eps, const float* ln_w, const float* ln_b, const void* w, const float* bias,
                           int64_t N, void* out, int64_t ldo, void* stream);
/* x_out[m] = x[m] + bd + b2 + ao[m] . wd^T + act[m] . w2^T   (attention output projection and 4h -> h projection as one product over
 * the concatenated K = h + n1; x_out may alias x).  The K reduction is split over blocks; partial tiles are added in slice order by the
 * last block of a column group to arrive (no floating-point atomics: bit-identical from run to run).  workspace: at least
 * mafed_decode_out_workspace_bytes(M, h) bytes, ZERO-FILLED once before the first call (it holds the arrival counters, which every
 * launch leaves at zero again); one workspace per stream. */
size_t mafed_decode_out_workspace_bytes(int M, int h);
/* tools: device buffer of 8 int64 per workgroup that the next mafed_decode_out launches fill with wall-clock stamps (NULL = off) */
int mafed_decode_set_trace(void* buf);
int mafed_attn_decode_set_trace(void* buf);   /* the same for mafed_attn_decode_prerot's all-rows-in-flight kernel: 8 int64 per (batch, head) */
int mafed_decode_out(const float* x, float* x_out, int M, int h, int n1, const void* ao, const void* act, const void* wd,
                     const float* bd, const void* w2, const float* b2, void* workspace, size_t workspace_bytes, void* stream);

/* ---- decode step as ONE launch (csrc/decode_flow.hip) ------------------------------------------------------------------
 * Every layer's work items (LayerNorm rows, 16-column strips of [q|k|v|4h], (batch, head) attention slices, K-slices of [dense|fc2]) and
 * the LM head are workgroups of one grid in dependency order that hand over through arrival counters in device memory; a workgroup
 * requests its weights / cached K|V rows before it waits for its inputs.  Same arithmetic as the three-launch layer above (bf16 weights
 * and activations, fp32 residual stream, deterministic summation orders).  Shapes: mafed_decode_flow_supported (h = 1024, head size 64,
 * M <= 32 rows, n1 % 512 == 0, <= 768 keys).
 *   layers  device array of L + 1 records of 14 pointers: {ln1_w, ln1_b, ln2_w, ln2_b, wqkv, bqkv, w1, b1, wd, bd, w2, b2, kv_prefix,
 *           kv_new}; record L describes the head: ln1_w / ln1_b = final LayerNorm, wqkv = embed_out [V, h], the rest unused
 *   x       [32, h] fp32: rows < M hold the embedded tokens on entry (updated in place); ln1 / ln2 / ao [32, h], act [32, n1] bf16 scratch
 *   flags   mafed_decode_flow_flag_bytes(L) bytes, ZERO-FILLED before every call; its last word is an error flag (non-zero: a hand-over
 *           timed out -- the results of that step are undefined)
 *   kv_*    the pre-rotated cache of mafed_attn_decode_prerot; row t of every kv_new receives this step's q | k | v (k rotated)
 *   logits  [M, V] bf16 */
int mafed_decode_flow_supported(int M, int h, int n1, int H, int D, int V, int nk);
/* tools: workgroups of a step's grid, and a device buffer of 4 int64 per workgroup that the next launches fill with wall-clock stamps
 * {dispatched, wait over, done} (NULL switches the trace off) */
size_t mafed_decode_flow_grid(int L, int M, int h, int n1, int H, int V);
int mafed_decode_flow_set_trace(void* buf);
size_t mafed_decode_flow_flag_bytes(int L);
size_t mafed_decode_flow_workspace_bytes(int h, int n1);
int mafed_decode_flow_step(const void* layers, int L, int M, int h, int n1, int H, int D, int S0, int cap, int t, int rot, int P, int T, int V,
                           float eps, float* x, void* ln1, void* ln2, void* act, void* ao, void* workspace, size_t workspace_bytes,
                           void* flags, size_t flag_bytes, const float* rot_cos, const float* rot_sin, const int64_t* attention_mask,
                           void* logits, void* stream);

/* Second launch of a decode layer, behind mafed_decode_ln_qkv_fc1: attention over the pre-rotated cache AND x <- x + dense(ao) + fc2(act)
 * as one grid (fc2 K-slices, then (batch, head) attention slices, then the dense K-slices, which wait for their heads' arrival counter):
 * the K|V stream and the output weights are requested together.  layer_rec: one record of 14 pointers as in mafed_decode_flow_step
 * (only wd, bd, w2, b2, kv_prefix, kv_new are read).  act [32, n1] (rows < M from the first launch), ao [32, h] scratch, both bf16;
 * workspace: mafed_decode_flow_workspace_bytes(h, n1); flags: 1 KB, ZERO on entry (last word: error flag).  Shapes:
 * mafed_decode_flow_supported. */
int mafed_decode_attn_out(const void* layer_rec, int M, int h, int n1, int H, int D, int S0, int cap, int t, int rot, int P, int T,
                          float* x, const void* act, void* ao, void* workspace, size_t workspace_bytes, void* flags,
                          const float* rot_cos, const float* rot_sin, const int64_t* attention_mask, void* stream);

/* ---- online EWC penalty (mafed/methods/ewc.py:105-127) -------------------------------------
 * The reference's compute_regularization over named_parameters(), on the flat fp32 buffers:
 *   fwd: out[0] = beta * out[0] + half_lambda * sum_i fisher[i] * (p[i] - p_old[i])^2     (half_lambda = 0.5 * reg_lambda;
 *        beta = 1 chains the per-task terms of the non-online variant, ewc.py:124-126)
 *   bwd: grad[i] += coef_dev[0] * lambda * fisher[i] * (p[i] - p_old[i])                  (coef_dev = d loss / d penalty)
 * Deterministic two-stage reduction; workspace from mafed_ewc_workspace_bytes. */
size_t mafed_ewc_workspace_bytes(int64_t n);
int mafed_ewc_penalty_fwd(const float* p, const float* p_old, const float* fisher, int64_t n, float half_lambda, float beta,
                          float* out, void* workspace, size_t workspace_bytes, void* stream);
int mafed_ewc_penalty_bwd(const float* p, const float* p_old, const float* fisher, int64_t n, float lambda,
                          const float* coef_dev, float* grad, void* stream);

/* ---- embedding + concat (mafed/model/vl_pythia.py:282-283) ---------------------------------------------------
 * h0[b, :P] = image[b] ; h0[b, P:] = embed_in[input_ids[b]]  -> fp32 [B,P+T,h].  image in img_dtype [B,P,h]. */
int mafed_embed_concat_fwd(const void* image, mafed_dtype img_dtype, const float* embed_in, const int64_t* input_ids,
                           int B, int P, int T, int h, int64_t V, float* h0, void* stream);
/* d_image[b] = dh0[b,:P] (img_dtype) ; d_embed_in[input_ids[b,t]] += dh0[b,P+t] (fp32 atomics) */
int mafed_embed_concat_bwd(const float* dh0, const int64_t* input_ids, int B, int P, int T, int h, int64_t V,
                           void* d_image, mafed_dtype img_dtype, float* d_embed_in, void* stream);

/* ---- per-sample-normalised shifted cross-entropy (mafed/model/vl_pythia.py:44-96) ---------------------------
 * logits [B,T,V] (text positions only), labels int64 [B,T]; row (b,t) predicts labels[b,t+1]; ignore_index -100;
 * loss = mean_b( sum_t ce[b,t] / max(count_b, 1e-13) ).  lse fp32 [B,T] saved.  loss_out fp32 [1]. */
int mafed_ce_fwd(const void* logits, mafed_dtype dtype, const int64_t* labels, int B, int T, int64_t V,
                 float* lse, float* row_loss /* [B,T] scratch */, float* loss_out, void* stream);
/* mafed_ce_fwd whose loss becomes NaN when the device flag *poison_flag is non-zero: the row-sparse LM head (mafed_label_rows) raises
 * its overflow flag when a sample has more labelled positions than the caller's hint promised -- rows were dropped, the loss would be
 * silently wrong -- and the step then fails loudly (NaN loss, caught by every trainer) without a host synchronisation. */
int mafed_ce_fwd_guarded(const void* logits, mafed_dtype dtype, const int64_t* labels, int B, int T, int64_t V,
                         float* lse, float* row_loss, float* loss_out, const int* poison_flag, void* stream);
/* dlogits[b,t,:] = gloss * (softmax - onehot) / (B * count_b) for valid rows, 0 otherwise (incl. t = T-1).
 * gloss_dev: device scalar (upstream dL/dloss).  dlogits may alias logits. */
int mafed_ce_bwd(const void* logits, mafed_dtype dtype, const int64_t* labels, const float* lse, int B, int T, int64_t V,
                 const float* gloss_dev, void* dlogits, void* stream);

/* ---- MAFED per-modality masked distillation (mafed/methods/distillation.py:124-166, 226-257) -----------------
 * s, t: fp32 [B,S,h] student / frozen-teacher hidden state of one layer; attention_mask int64 [B,T]; P image tokens.
 * One pass serves both masks: out[4] = { sum_lang d, sum_vision d, n_lang, n_vision } with
 *   mse:    d[row] = sum((s-t)^2)/h                      (_compute_mse_distillation_loss, :237-249)
 *   cosine: d[row] = 1 - cos(s,t)  (CosineEmbeddingLoss target 1; aten's form: cos = st / sqrt((ss + 1e-12) (tt + 1e-12)), the epsilon on the SQUARED norms; :226-235)
 * Deterministic two-stage reduction (no float atomics).  workspace >= mafed_distill_workspace_bytes(B*S). */
size_t mafed_distill_workspace_bytes(int64_t rows);
int mafed_distill_fwd(const float* s, const float* t, const int64_t* attention_mask, int B, int S, int P, int h,
                      int cosine, float* out4, void* workspace, size_t workspace_bytes, void* stream);
/* ds (+)= g_lang[row] / g_vision[row] weighted gradient of the above:  coef_dev[2] = {c_lang, c_vision} where the
 * host/torch side already folded upstream grad * layer_coeff * distillation_coeff * modality weight / count.
 *   mse:    ds = c * 2/h * (s - t)
 *   cosine: ds = c * -( t/(|s||t|) - cos * s/|s|^2 )
 * accumulate != 0 adds into ds, else overwrites (rows of no modality get 0). */
int mafed_distill_bwd(const float* s, const float* t, const int64_t* attention_mask, int B, int S, int P, int h,
                      int cosine, const float* coef_dev, float* ds, int accumulate, void* stream);
/* CLS variant (:251-257): token 0 only, cosine, mean over batch -> out[1]; bwd with coef_dev[1] = upstream/B */
/* The scalar tail of distill() for every distilled layer in one launch (mafed/methods/distillation.py:105-122,147-162;
 * distillation_loss_weights.py:71-79): from sums[n_layers][4] = {sum_lang, sum_vis, n_lang, n_vis} (mafed_distill_fwd) and the device
 * vector layer_coeff[n_layers] (layer coefficient x distillation_coeff) it writes per_layer[l] = lw * sum_lang / n_lang + vw * sum_vis / n_vis,
 * modality[l] = {lang mean, vision mean}, loss = sum_l coeff[l] * per_layer[l], and inject[l] = d loss / d {sum_lang, sum_vis, ., .}
 * (the coefficients mafed_layernorm_bwd's fused injection takes, to be scaled by the upstream gradient of the loss).
 * modality_mode 0 "equal": lw = n_lang / (n_lang + n_vis); 1 "balanced": lw = lang_weight; 2 "adaptive": lw = lang_weight_vec[l]. */
int mafed_distill_combine(const float* sums, int n_layers, const float* layer_coeff_dev, int modality_mode, float lang_weight,
                          const float* lang_weight_vec_dev, float* loss_out, float* per_layer_out, float* modality_out,
                          float* inject_out, void* stream);

int mafed_distill_cls_fwd(const float* s, const float* t, int B, int S, int h, float* out1, void* stream);
int mafed_distill_cls_bwd(const float* s, const float* t, int B, int S, int h, const float* coef_dev, float* ds,
                          int accumulate, void* stream);

/* ---- optimiser side -------------------------------------------------------------------------------------------
 * Global L2 norm of a flat fp32 gradient buffer (Lightning gradient_clip_val -> clip_grad_norm_, mafed/train.py:288):
 * out[0] = ||g||_2, out[1] = clip scale = min(1, max_norm / (norm + 1e-6)).  workspace >= gradnorm_workspace_bytes. */
size_t mafed_gradnorm_workspace_bytes(int64_t n);
int mafed_gradnorm_clip(const float* g, int64_t n, float max_norm, float* out2, void* workspace, size_t workspace_bytes,
                        void* stream);
/* The same norm in pieces: mafed_gradnorm_partial writes mafed_gradnorm_blocks(n) sum-of-squares partials of one range (call it as soon
 * as that range of the gradient is final, on the stream that finished it); mafed_gradnorm_finish folds n_partials of them (index
 * order) into out2 = {norm, clip scale}.  Replaces the single pass over the whole buffer at the start of the optimiser step. */
int mafed_gradnorm_blocks(int64_t n);
int mafed_gradnorm_partial(const float* g, int64_t n, float* partial_out, void* stream);
int mafed_gradnorm_finish(const float* partial, int n_partials, float max_norm, float* out2, void* stream);
/* HF-style AdamW on a flat segment (mafed/optim/adamw.py:86-111): m,v update; denom = sqrt(v) + eps (not bias
 * corrected); p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/denom; then p -= lr*wd*p.  g is multiplied by clip_scale_dev[1]
 * (from mafed_gradnorm_clip) and by grad_mul (1/world for DDP means).  lr_dev: device scalar (scheduled lr).
 * step >= 1: bias corrections computed on the host from step (double).  step == 0: lr_dev points at three floats
 * {lr, 1-b1^t, sqrt(1-b2^t)} that the host refreshes before each launch -- the form a hipGraph replay needs.
 * If p_bf16 != NULL also writes the bf16 shadow copy used by the MFMA GEMMs.  The host keeps decayed and
 * non-decayed parameters in two contiguous segments of the flat buffer and calls this once per segment. */
int mafed_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                     float eps, float weight_decay, int step, const float* clip_dev /* out2 of gradnorm or NULL */,
                     float grad_mul, void* p_bf16, void* stream);

/* mafed_adamw_step that also writes zeros over g in the same pass (optimizer.zero_grad() of the next accumulation window). */
int mafed_adamw_step_zero_grad(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                               float eps, float weight_decay, int step, const float* clip_scale_dev, float grad_mul, void* p_bf16,
                               void* stream);

/* General form: zeroes only the first `zero_n` elements of g (0 = none, n = all of it = mafed_adamw_step_zero_grad).  For the chunk of one
 * decoder layer -- [LayerNorm weights | the four weight matrices] -- the host passes the LayerNorm part: their gradients are accumulated
 * (+=) by the LayerNorm backward and need a zeroed buffer, the matrices' gradients are OVERWRITTEN by the first micro-batch's weight-
 * gradient GEMMs of the next window (beta = 0), so zero-writing them (1.2 GB per step at 410M) and re-reading them in those GEMMs'
 * epilogues is skipped.  Replaces optimizer.zero_grad() + AdamW.step (mafed/optim/adamw.py:50-113) for that chunk. */
int mafed_adamw_step_partial_zero(float* p, float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2,
                                  float eps, float weight_decay, int step, const float* clip_scale_dev, float grad_mul, void* p_bf16,
                                  int64_t zero_n, void* stream);

/* Device-resident schedule (mafed/optim/sched.py:34-48 + the bias corrections of adamw.py:94-97): state_dev[0] = number of
 * optimiser steps taken so far; increments it to t and writes hyper3_dev = {base_lr * lambda(t-1), 1-b1^t, sqrt(1-b2^t)}
 * (double precision inside) for mafed_adamw_step(step = 0).  total_steps <= 0 means a constant learning rate. */
int mafed_optim_advance(int64_t* state_dev, double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1, double beta2,
                        float* hyper3_dev, void* stream);

/* Guard of an optimiser step against a non-finite gradient (what a GradScaler does on the device): the gradient-norm kernels write the
 * clip scale out2[1] = -1 when the global norm is NaN / infinite (e.g. the poisoned loss of the row-sparse head's overflow flag);
 * mafed_adamw_step* given such a clip_dev leaves p, m, v and the bf16 shadow untouched (the zero_grad form still zeroes g),
 * mafed_gradnorm_finish_advance does not advance the step counter, and this form of mafed_optim_advance does not either when
 * clip_dev[1] < 0 (clip_dev may be NULL = unguarded).  The host sees the skipped step as a non-finite out2[0] / norm_log at its next
 * natural synchronisation.  Reference behaviour (mafed/optim/adamw.py:86-111 applied to NaN gradients) would corrupt every parameter. */
int mafed_optim_advance_guarded(int64_t* state_dev, double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1, double beta2,
                                float* hyper3_dev, const float* clip_dev, void* stream);

/* sumsq16[k] += the squares of part of x (k = 0..15, their total += sum x^2): the unfused form of mafed_gemm_problem.sumsq, and the way
 * to put any fp32 range into the same 16 slots.  x 16-byte aligned. */
int mafed_sumsq_accumulate(const float* x, int64_t n, float* sumsq16, void* stream);
/* 1 if mafed_gemm_grouped would run these n products (dense operands) as ONE persistent launch that fills mafed_gemm_problem.sumsq from its
 * epilogue; 0 if the squares would cost a pass over C behind the product(s) (shapes the 256 x 256-tile kernel takes, groups that do not
 * fill the chip) -- a caller that has a cheaper way to the same norm (mafed_gradnorm_partial over a contiguous range) then skips sumsq. */
int mafed_gemm_grouped_fuses_sumsq(mafed_dtype in_dtype, int transA, int transB, mafed_dtype c_dtype, const int64_t* M, const int64_t* N,
                                   const int64_t* K, int n);

/* mafed_gradnorm_finish followed by mafed_optim_advance as ONE launch (both are single-thread tails on the optimiser step's critical
 * path); norm_log (or NULL) additionally receives the norm -- a slot the caller owns, e.g. for the step's log record, so that no
 * device copy of out2[0] is needed before the next step overwrites it.  Same arithmetic, bit for bit, as the two calls. */
int mafed_gradnorm_finish_advance(const float* partial, int n_partials, float max_norm, float* out2, float* norm_log, int64_t* state_dev,
                                  double base_lr, int64_t warmup_steps, int64_t total_steps, double beta1, double beta2, float* hyper3_dev,
                                  void* stream);

/* ---- small utilities --------------------------------------------------------------------------------------------- */
int mafed_cast(const void* src, mafed_dtype src_dtype, void* dst, mafed_dtype dst_dtype, int64_t n, void* stream);
/* dst [B,S,h] fp32 = zeros for the P image positions of every sample, src [B,S-P,h] for its text positions; dst_lp (bf16, optional): the
 * same rows in the compute dtype.  The start of the residual-stream gradient: only text positions feed the LM head (vl_pythia.py:310). */
int mafed_pad_text_rows(const float* src, int B, int S, int P, int h, float* dst, void* dst_lp, void* stream);
/* Row-sparse LM head (training): the rows of the [B,T] text block whose shifted label is a token (labels[b,t+1] != -100), at most
 * Rc-1 per sample, as a compact [B,Rc] problem for the same CE kernels: row_of_slot [B*Rc] (text row or -1), slot_of_row [B*T] (slot or
 * -1), labels_c [B,Rc] (slot n's label at n+1).  overflow[0] = 1 if a sample had more labelled rows than fit.  mafed_gather_rows moves
 * the rows either way: dst[r,:] = idx[r] >= 0 ? src[idx[r],:] : 0. */
int mafed_label_rows(const int64_t* labels, int B, int T, int Rc, int* row_of_slot, int* slot_of_row, int64_t* labels_c, int* overflow,
                     void* stream);
int mafed_gather_rows(const void* src, mafed_dtype dtype, const int* idx, int64_t n_out, int h, void* dst, void* stream);
/* y = gelu_erf(x) elementwise (used by tests; the product path fuses GELU into mafed_gemm) */
int mafed_gelu(const void* x, void* y, mafed_dtype dtype, int64_t n, void* stream);

/* ---- frozen CLIP vision tower (mafed/model/vl_pythia.py:196-198, 453-475; clip: = transformers/models/clip/modeling_clip.py) ----
 * The tower is ``CLIPVisionModel`` called with output_hidden_states and cut at hidden_states[-2]; its Linear layers go through
 * mafed_gemm (fc1 with MAFED_EPI_QUICK_GELU), its LayerNorms through mafed_layernorm_fwd; the three entry points below are the rest.
 * mafed_patchify: im2col of the stride = kernel patch convolution (clip:148-154, 209-210): pixels [B,C,H,W] ->
 *   out [rows_out, k_pad], out[b * np + p][c * patch^2 + i * patch + j] = pixels[b][c][py * patch + i][px * patch + j]; columns
 *   >= C * patch^2 and rows >= B * np are zero (k_pad a multiple of 64 and rows_out a multiple of the GEMM tile keep the product
 *   on the LDS-DMA GEMM).  The patch embedding is then mafed_gemm(out, W[h, k_pad]^T).
 * mafed_vit_assemble: tokens[b][0] = class_embedding + pos[0], tokens[b][1 + p] = patch_emb[b * np + p] + pos[1 + p] (clip:212-217), fp32.
 * mafed_attn_fwd_bidir: softmax(q k^T D^-0.5) v over all S keys, no mask, no rotary (clip:259-277); qkv [B,S,H,3,D] as in mafed_attn_fwd. */
int mafed_patchify(const void* pixels, mafed_dtype pix_dtype, int B, int C, int H, int W, int patch, int64_t rows_out, int k_pad,
                   void* out, mafed_dtype out_dtype, void* stream);
int mafed_vit_assemble(const void* patch_emb, mafed_dtype pe_dtype, int64_t ld_pe, const float* class_embedding,
                       const float* position_embedding, int B, int num_patches, int h, float* tokens, void* stream);
int mafed_attn_fwd_bidir(const void* qkv, mafed_dtype dtype, int B, int S, int H, int D, void* out, float* lse, void* stream);

/* ---- measurement: per-kernel execution time of the launches this library makes --------------------------------------
 * No reference counterpart (the reference has no profiling); bench.py's `roofline` / `kernels` come from here.
 * Between mafed_prof_begin(max_records) and mafed_prof_end() every hot kernel is launched with a start/stop event pair
 * (hipExtLaunchKernelGGL): the elapsed time of a pair is that dispatch's own execution time on the GPU -- what
 * `rocprofv3 --kernel-trace` reports -- independent of how long the launch sat queued behind other streams.  After the caller
 * has synchronised the device, mafed_prof_collect() returns, in launch order, each record's kernel tag, its algorithmic work
 * (flops for the MFMA kernels, bytes for the HBM-bound ones), its duration in ms and (optionally) its start
 * time in ms relative to the first record's start (host pointers, any may be NULL), and the number of records held.  mafed_prof_tag_name() names a tag.  Launches beyond max_records are not recorded. */
int mafed_prof_begin(int max_records);
int mafed_prof_end(void);
int mafed_prof_collect(int* tags_host, double* work_host, float* ms_host, float* start_ms_host, int max);
const char* mafed_prof_tag_name(int tag);

/* test / tuning hook: 0 = automatic kernel choice, 1 = force the register-staged MFMA GEMM (the ragged-shape kernel),
 * 10 + c = force LDS-DMA tile configuration c; 100 = automatic split-K for accumulate-only outputs, 101 = no split-K,
 * 100 + n = force n K-splits where legal */
int mafed_gemm_set_variant(int variant);
/* 720 = the persistent kernels walk their tiles in the static order (tile = round x grid + slot) whatever the call says, 721 = ticketed
 * order for every multi-round launch (a block's first tile is static, every further one is drawn from a per-XCD queue), 722 = per call
 * (MAFED_EPI_TICKETED; default).
 * mafed_gemm_get_variant reads the hooks back (which = 0: the tile-configuration variant, 7: the persistent-kernel mode 700 / 701 /
 * 710 + c, 72: 720 / 721, 73: launches that ran in ticketed order so far), so that a caller that changes one for a measurement can
 * restore what was set before */
int mafed_gemm_get_variant(int which);
/* 700 = never take the persistent ping-pong kernel, 701 = automatic (default), 710 + c = force its tile configuration c
 * (0: 144x256 tiles, 1: 128x256 tiles, 2: 256x256 tiles) wherever the shape tiles it.  mafed_gemm_pp_launches(): launches that took that kernel so far
 * (tests assert that a forced configuration really ran). */
int mafed_gemm_pp_launches(void);

/* test / tuning hook: 0 = automatic (one-block-per-head "resident" kernels when K/V fit in LDS), 1 = tiled kernels only */
int mafed_attn_set_variant(int variant);

/* test hook: the exact (non-MFMA) forward kernel on bf16 data -- on-GPU cross-check of the MFMA kernels */
int mafed_attn_fwd_exact_bf16(const void* qkv, int B, int S, int H, int D, int rot, const float* rot_cos, const float* rot_sin,
                              const int64_t* attention_mask, int T, void* out, float* lse, void* stream);

/* Tuning helper (not on the product path): occupies `blocks` CUs with one 512-thread block each (`lds_bytes` of LDS) for ~`cycles` shader
 * clocks -- a stand-in for a long-running collective kernel when measuring GEMMs with part of the chip taken (tools/contention_bench.py). */
int mafed_tune_occupy(int blocks, int lds_bytes, long long cycles, void* stream);

/* Tuning helper (not on the product path): `blocks` workgroups of `threads` threads stream `n_bytes` of `src` with 16-byte accesses, `unroll`
 * (2 / 4 / 8) loads in flight per thread; mode 0 reads, 1 copies to dst, 2 = four read + four written streams of n_bytes / 4 (AdamW-shaped).
 * Measures what a FEW CUs stream from HBM while the rest of the chip does something else (tools/stream_cu_bench.py). */
int mafed_tune_stream(const void* src, void* dst, long long n_bytes, int blocks, int threads, int unroll, int mode, void* stream);
/* The AdamW-shaped sweep of mafed_tune_stream, repeated over the buffer for `microseconds` of wall time (100 MHz device clock): a stand-in
 * for the kernels of ONE bucket's all-reduce -- `blocks` = RCCL channels -- on a single GPU (bench.py --emulate-collectives). */
int mafed_tune_stream_for(const void* src, void* dst, long long n_bytes, int blocks, int threads, double microseconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAFED_HIP_H */
