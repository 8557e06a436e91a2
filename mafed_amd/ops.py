"""Thin tensor-level wrappers over the C-ABI (one Python function per entry point of include/mafed_hip.h).

Every wrapper launches on torch's current HIP stream and never synchronises.  Tensors must live on the GPU;
there is deliberately no CPU implementation behind these names.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from mafed_amd import _lib
import contextlib
import threading

from mafed_amd._lib import BF16, EPI_GELU, EPI_GELU_BWD, EPI_NO_PERSISTENT, EPI_NONE, EPI_QUICK_GELU, EPI_RES1_BF16, EPI_TICKETED, F32, check  # noqa: F401


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _ptr(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise _lib.MafedHipError("mafed_amd ops need GPU tensors (no CPU fallback)")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of the caller's current HIP stream.  Through the two C entry points directly when this torch has them:
    ``torch.cuda.current_stream()`` builds a Stream object and resolves the device index through ``is_available()`` on every call
    (3 of the 15 ms the host spent enqueueing a 410M MAFED step, ~650 launches)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class _Fn:
    """Lazily bound entry points (one attribute lookup per call instead of load() + getattr)."""

    def __getattr__(self, name):
        fn = getattr(_lib.load(), "mafed_" + name)
        setattr(self, name, fn)
        return fn


_fn = _Fn()


class Workspace:
    """Grow-only scratch buffer handed to kernels that need one (no allocation inside the library)."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(1 << 20, dtype=torch.uint8, device=device)

    def get(self, nbytes: int) -> torch.Tensor:
        if self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


_workspaces = {}


def workspace(device) -> Workspace:
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = Workspace(device)
    return ws


_tls = threading.local()


@contextlib.contextmanager
def no_persistent_gemm(on: bool = True):
    """Inside this block the CALLING THREAD's ``gemm`` / ``gemm_grouped`` calls carry MAFED_EPI_NO_PERSISTENT: each of those launches keeps
    off the one-block-per-CU persistent kernels (e.g. a backward that runs beside collectives).  A per-call flag in the C-ABI, a
    thread-local here: nothing process-wide is touched, other threads / models / forced tuning variants are unaffected."""
    prev = getattr(_tls, "no_pp", 0)
    _tls.no_pp = EPI_NO_PERSISTENT if on else prev
    try:
        yield
    finally:
        _tls.no_pp = prev


@contextlib.contextmanager
def ticketed_gemm(on: bool = True):
    """Inside this block the calling thread's ``gemm`` / ``gemm_grouped`` calls carry MAFED_EPI_TICKETED: a launch that takes a persistent
    kernel over several rounds draws its tiles from per-XCD queues (a backward that runs beside collectives or other long-resident
    kernels).  Per call in the C-ABI, thread-local here, like :func:`no_persistent_gemm`."""
    prev = getattr(_tls, "no_pp", 0)
    _tls.no_pp = (prev | EPI_TICKETED) if on else prev
    try:
        yield
    finally:
        _tls.no_pp = prev


def gemm(A: torch.Tensor, B: torch.Tensor, transA: bool, transB: bool, out: Optional[torch.Tensor] = None,
         out_dtype: Optional[torch.dtype] = None, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_NONE,
         aux: Optional[torch.Tensor] = None, res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None,
         beta: float = 0.0, colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = op(A) @ op(B) with the fused epilogue of mafed_gemm.  A, B 2-D, same dtype (bf16 -> MFMA, f32 -> exact).
    ``colsum`` (fp32 [N]): += the column sums of the stored C (mafed_gemm_colsum)."""
    assert A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype
    assert A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    N = B.shape[0] if transB else B.shape[1]
    Kb = B.shape[1] if transB else B.shape[0]
    assert K == Kb, (A.shape, B.shape, transA, transB)
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    if res1 is not None and res1.dtype == torch.bfloat16:
        epilogue |= EPI_RES1_BF16
    epilogue |= getattr(_tls, "no_pp", 0)
    if colsum is None:
        rc = _fn.gemm(_dt(A), int(transA), int(transB), M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out),
                      out.stride(0), _dt(out), _ptr(bias), epilogue, _ptr(aux), _ptr(res1), _ptr(res2), float(beta), _stream())
    else:
        assert colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous()
        rc = _fn.gemm_colsum(_dt(A), int(transA), int(transB), M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out),
                             out.stride(0), _dt(out), _ptr(bias), epilogue, _ptr(aux), _ptr(res1), _ptr(res2), float(beta),
                             _ptr(colsum), _stream())
    if rc:
        check(rc, "mafed_gemm")
    return out


def gemm_grouped(problems, transA: bool, transB: bool) -> None:
    """Several independent ``C_i = op(A_i) @ op(B_i)`` as one persistent launch (mafed_gemm_grouped).  ``problems``: dicts with the
    keyword arguments of :func:`gemm` (``A``, ``B``, ``out`` required; ``bias``, ``epilogue``, ``aux``, ``res1``, ``res2``, ``beta``,
    ``colsum`` optional; ``sumsq`` = 16 fp32 slots that receive += the squares of the stored fp32 C).  All share the operand layouts, the input dtype and the output dtype."""
    n = len(problems)
    if n == 0:
        return
    arr = (_lib.GemmProblem * n)()
    in_dt = out_dt = None
    for i, q in enumerate(problems):
        A, B, out = q["A"], q["B"], q["out"]
        assert A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
        M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
        N = B.shape[0] if transB else B.shape[1]
        assert K == (B.shape[1] if transB else B.shape[0]) and out.shape == (M, N)
        in_dt = _dt(A) if in_dt is None else in_dt
        out_dt = _dt(out) if out_dt is None else out_dt
        assert _dt(A) == in_dt and _dt(out) == out_dt, "a grouped launch shares its input and output types"
        epi = q.get("epilogue", EPI_NONE) | getattr(_tls, "no_pp", 0)
        res1 = q.get("res1")
        if res1 is not None and res1.dtype == torch.bfloat16:
            epi |= EPI_RES1_BF16
        cs = q.get("colsum")
        assert cs is None or (cs.dtype == torch.float32 and cs.numel() == N and cs.is_contiguous())
        g = arr[i]
        g.M, g.N, g.K = M, N, K
        g.A, g.lda, g.B, g.ldb, g.C, g.ldc = _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0)
        g.bias, g.epilogue, g.aux = _ptr(q.get("bias")), epi, _ptr(q.get("aux"))
        g.res1, g.res2, g.beta, g.colsum = _ptr(res1), _ptr(q.get("res2")), float(q.get("beta", 0.0)), _ptr(cs)
        sq = q.get("sumsq")
        assert sq is None or (sq.dtype == torch.float32 and sq.numel() >= 16 and sq.is_contiguous() and out.dtype == torch.float32)
        g.sumsq = _ptr(sq)
    rc = _fn.gemm_grouped(in_dt, int(transA), int(transB), out_dt, arr, n, _stream())
    if rc:
        check(rc, "mafed_gemm_grouped")


def colsum_(X: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum_m X[m, n]"""
    assert X.dim() == 2 and X.stride(1) == 1 and out.dtype == torch.float32
    M, N = X.shape
    check(_lib.load().mafed_colsum(_ptr(X), _dt(X), M, N, X.stride(0), _ptr(out), 0, 0, _stream()), "mafed_colsum")


def layernorm_fwd(x: torch.Tensor, w1, b1, w2=None, b2=None, eps: float = 1e-5, out_dtype=torch.float32, save_stats: bool = True):
    rows, h = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous()
    y1 = torch.empty((rows, h), dtype=out_dtype, device=x.device)
    y2 = torch.empty_like(y1) if w2 is not None else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(_lib.load().mafed_layernorm_fwd(_ptr(x), rows, h, eps, _ptr(w1), _ptr(b1), _ptr(y1), _ptr(w2), _ptr(b2), _ptr(y2),
                                          _dt(y1), _ptr(mean), _ptr(rstd), _stream()), "mafed_layernorm_fwd")
    return y1, y2, mean, rstd


def layernorm_bwd(dy1, dy2, x, mean, rstd, w1, w2, dres, dw1, db1, dw2=None, db2=None, want_lp: bool = False,
                  teacher=None, attention_mask=None, S: int = 0, P: int = 0, inj_scale=None, inj_mul: float = 1.0,
                  dxsum_a=None, dxsum_b=None):
    rows, h = x.shape
    dx = torch.empty((rows, h), dtype=torch.float32, device=x.device)
    dx_lp = torch.empty((rows, h), dtype=dy1.dtype, device=x.device) if want_lp else None
    lib = _lib.load()
    nb = lib.mafed_layernorm_bwd_workspace_bytes(rows, h)
    ws = workspace(x.device).get(nb)
    check(lib.mafed_layernorm_bwd(_ptr(dy1), _ptr(dy2), _dt(dy1), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w1), _ptr(w2), rows, h,
                                  _ptr(dres), _ptr(dx), _ptr(dx_lp), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(teacher),
                                  _ptr(attention_mask), S, P, S - P, _ptr(inj_scale), float(inj_mul), _ptr(dxsum_a), _ptr(dxsum_b), _ptr(ws), ws.numel(),
                                  _stream()),
          "mafed_layernorm_bwd")
    return dx, dx_lp


def layernorm_bwd_rows(dy1, dy2, x, mean, rstd, w1, w2, dres, want_lp: bool = False, teacher=None, attention_mask=None, S: int = 0, P: int = 0,
                       inj_scale=None, inj_mul: float = 1.0, want_dxsum: bool = False):
    """Row kernel of the LayerNorm backward alone -> (dx, dx_lp, partials); ``partials`` is a private buffer that
    ``layernorm_bwd_params`` folds into the parameter gradients later (on another stream, ordered by the caller)."""
    rows, h = x.shape
    dx = torch.empty((rows, h), dtype=torch.float32, device=x.device)
    dx_lp = torch.empty((rows, h), dtype=dy1.dtype, device=x.device) if want_lp else None
    lib = _lib.load()
    ws = torch.empty(lib.mafed_layernorm_bwd_workspace_bytes(rows, h), dtype=torch.uint8, device=x.device)
    check(lib.mafed_layernorm_bwd_rows(_ptr(dy1), _ptr(dy2), _dt(dy1), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(w1), _ptr(w2), rows, h, _ptr(dres),
                                       _ptr(dx), _ptr(dx_lp), _ptr(teacher), _ptr(attention_mask), S, P, S - P, _ptr(inj_scale), float(inj_mul),
                                       1 if want_dxsum else 0, _ptr(ws), ws.numel(), _stream()), "mafed_layernorm_bwd_rows")
    return dx, dx_lp, ws


def layernorm_bwd_params(ws: torch.Tensor, rows: int, h: int, dw1, db1, dw2=None, db2=None, dxsum_a=None, dxsum_b=None) -> None:
    check(_lib.load().mafed_layernorm_bwd_params(rows, h, _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(dxsum_a), _ptr(dxsum_b), _ptr(ws),
                                                 ws.numel(), _stream()), "mafed_layernorm_bwd_params")


def attn_fwd(qkv: torch.Tensor, B: int, S: int, H: int, D: int, rot: int, cos, sin, attention_mask: torch.Tensor):
    T = attention_mask.shape[1]
    out = torch.empty((B * S, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    check(_lib.load().mafed_attn_fwd(_ptr(qkv), _dt(qkv), B, S, H, D, rot, _ptr(cos), _ptr(sin), _ptr(attention_mask), T, _ptr(out),
                                     _ptr(lse), _stream()), "mafed_attn_fwd")
    return out, lse


def attn_fwd_exact_bf16(qkv, B, S, H, D, rot, cos, sin, attention_mask):
    T = attention_mask.shape[1]
    out = torch.empty((B * S, H * D), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    check(_lib.load().mafed_attn_fwd_exact_bf16(_ptr(qkv), B, S, H, D, rot, _ptr(cos), _ptr(sin), _ptr(attention_mask), T, _ptr(out),
                                                _ptr(lse), _stream()), "mafed_attn_fwd_exact_bf16")
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, S, H, D, rot, cos, sin, attention_mask, colsum: Optional[torch.Tensor] = None):
    """dqkv; ``colsum`` (fp32 [3*H*D]) += the column sums of dqkv (query_key_value.bias gradient, mafed_attn_bwd_colsum)."""
    T = attention_mask.shape[1]
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    if colsum is None:
        check(_lib.load().mafed_attn_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _dt(qkv), B, S, H, D, rot, _ptr(cos), _ptr(sin),
                                         _ptr(attention_mask), T, _ptr(dqkv), _ptr(delta), _stream()), "mafed_attn_bwd")
    else:
        assert colsum.dtype == torch.float32 and colsum.numel() == 3 * H * D and colsum.is_contiguous()
        check(_lib.load().mafed_attn_bwd_colsum(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _dt(qkv), B, S, H, D, rot, _ptr(cos),
                                                _ptr(sin), _ptr(attention_mask), T, _ptr(dqkv), _ptr(delta), _ptr(colsum), _stream()),
              "mafed_attn_bwd_colsum")
    return dqkv


def attn_fwd_bidir(qkv: torch.Tensor, B: int, S: int, H: int, D: int, out: Optional[torch.Tensor] = None):
    """Bidirectional attention of the CLIP vision tower: qkv [>= B*S, H*3*D] -> out [>= B*S, H*D] (rows past B*S untouched)."""
    assert qkv.dim() == 2 and qkv.shape[1] == 3 * H * D and qkv.shape[0] >= B * S and qkv.is_contiguous()
    if out is None:
        out = torch.empty((B * S, H * D), dtype=qkv.dtype, device=qkv.device)
    assert out.dtype == qkv.dtype and out.shape[0] >= B * S and out.shape[1] == H * D and out.is_contiguous()
    lse = torch.empty((B, H, S), dtype=torch.float32, device=qkv.device)
    check(_lib.load().mafed_attn_fwd_bidir(_ptr(qkv), _dt(qkv), B, S, H, D, _ptr(out), _ptr(lse), _stream()), "mafed_attn_fwd_bidir")
    return out


def patchify(pixels: torch.Tensor, patch: int, rows_out: int, k_pad: int, out_dtype: torch.dtype) -> torch.Tensor:
    """im2col of the patch convolution: pixels [B,C,H,W] -> [rows_out, k_pad] (zero padding rows / columns)."""
    B, C, H, W = pixels.shape
    assert pixels.is_contiguous()
    out = torch.empty((rows_out, k_pad), dtype=out_dtype, device=pixels.device)
    check(_lib.load().mafed_patchify(_ptr(pixels), _dt(pixels), B, C, H, W, int(patch), int(rows_out), int(k_pad), _ptr(out), _dt(out), _stream()),
          "mafed_patchify")
    return out


def vit_assemble(patch_emb: torch.Tensor, class_embedding: torch.Tensor, position_embedding: torch.Tensor, B: int, num_patches: int, h: int,
                 out: torch.Tensor) -> torch.Tensor:
    """out[b, 0] = cls + pos[0]; out[b, 1 + p] = patch_emb[b * np + p] + pos[1 + p]   (fp32 rows of ``out`` [>= B*(np+1), h])"""
    assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= B * (num_patches + 1) and out.shape[1] == h
    assert class_embedding.dtype == torch.float32 and position_embedding.dtype == torch.float32 and position_embedding.is_contiguous()
    check(_lib.load().mafed_vit_assemble(_ptr(patch_emb), _dt(patch_emb), patch_emb.stride(0), _ptr(class_embedding), _ptr(position_embedding), B,
                                         num_patches, h, _ptr(out), _stream()), "mafed_vit_assemble")
    return out


def rotate_k_rows_(qkv: torch.Tensor, B: int, S: int, H: int, D: int, rot: int, cos, sin) -> None:
    """k part of a [B*S, H*3*D] (or [B,S,...]) qkv tensor rotated in place for each row's position (mafed_rotate_k_rows)."""
    assert qkv.is_contiguous() and qkv.numel() == B * S * H * 3 * D and cos.shape[0] >= S
    check(_lib.load().mafed_rotate_k_rows(_ptr(qkv), _dt(qkv), B, S, H, D, rot, _ptr(cos), _ptr(sin), _stream()), "mafed_rotate_k_rows")


def attn_decode(qkv_prefix: torch.Tensor, S0: int, qkv_new: torch.Tensor, t: int, B: int, H: int, D: int, rot: int, cos, sin,
                attention_mask: torch.Tensor, prerot: bool = False) -> torch.Tensor:
    """One decode step of attention: query = row t of ``qkv_new`` [B,cap,3*H*D], keys = the prefill's ``qkv_prefix``
    [B*S0, 3*H*D] followed by rows 0..t of ``qkv_new``.  -> [B, H*D]"""
    cap = qkv_new.shape[1]
    assert qkv_new.dim() == 3 and qkv_new.is_contiguous() and qkv_prefix.is_contiguous() and qkv_new.dtype == qkv_prefix.dtype
    assert cos.shape[0] >= S0 + t + 1
    out = torch.empty((B, H * D), dtype=qkv_new.dtype, device=qkv_new.device)
    fn = _lib.load().mafed_attn_decode_prerot if prerot else _lib.load().mafed_attn_decode
    check(fn(_ptr(qkv_prefix), S0, _ptr(qkv_new), cap, t, _dt(qkv_new), B, H, D, rot, _ptr(cos), _ptr(sin),
             _ptr(attention_mask), attention_mask.shape[1], _ptr(out), _stream()), "mafed_attn_decode")
    return out


def gemm_grouped_fuses_sumsq(shapes: Sequence[Tuple[int, int, int]], transA: bool, transB: bool) -> bool:
    """Would ``gemm_grouped`` run these (M, N, K) bf16 -> fp32 products as one persistent launch with the squares of C fused into its
    epilogue?  (host-side query, no launch)"""
    import ctypes
    n = len(shapes)
    arr = lambda k: (ctypes.c_int64 * n)(*[int(s[k]) for s in shapes])
    Ms, Ns, Ks = arr(0), arr(1), arr(2)
    return bool(_lib.load().mafed_gemm_grouped_fuses_sumsq(_lib.BF16, int(transA), int(transB), _lib.F32, ctypes.cast(Ms, ctypes.c_void_p),
                                                           ctypes.cast(Ns, ctypes.c_void_p), ctypes.cast(Ks, ctypes.c_void_p), n))


def decode_supported(M: int, h: int, n1: int) -> bool:
    """Shapes served by the fused decode layer kernels (``decode_ln_qkv_fc1`` / ``decode_out``)."""
    return bool(_lib.load().mafed_decode_supported(int(M), int(h), int(n1)))


def decode_ln_qkv_fc1(x: torch.Tensor, ln1_w, ln1_b, ln2_w, ln2_b, eps: float, wqkv: torch.Tensor, bqkv: torch.Tensor,
                      qkv_row: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, a_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode step, first launch of a layer: ``qkv_row[m] = LN1(x[m]) @ wqkv^T + bqkv`` (a strided [M, 3h] view: the cache row of this
    step) and ``a = gelu(LN2(x[m]) @ w1^T + b1)`` -> a [M, n1] bf16.  x fp32 [M, h]; weights bf16 [N, h]."""
    M, h = x.shape
    n1 = w1.shape[0]
    assert x.dtype == torch.float32 and x.is_contiguous() and wqkv.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16
    assert wqkv.shape == (3 * h, h) and w1.shape[1] == h and wqkv.is_contiguous() and w1.is_contiguous()
    assert qkv_row.shape == (M, 3 * h) and qkv_row.dtype == torch.bfloat16 and qkv_row.stride(1) == 1
    if a_out is None:
        a_out = torch.empty((M, n1), dtype=torch.bfloat16, device=x.device)
    check(_lib.load().mafed_decode_ln_qkv_fc1(_ptr(x), M, h, float(eps), _ptr(ln1_w), _ptr(ln1_b), _ptr(ln2_w), _ptr(ln2_b), _ptr(wqkv),
                                              _ptr(bqkv), _ptr(qkv_row), qkv_row.stride(0), _ptr(w1), _ptr(b1), n1, _ptr(a_out), _stream()),
          "mafed_decode_ln_qkv_fc1")
    return a_out


def decode_ln_linear(x: torch.Tensor, ln_w, ln_b, eps: float, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``LN(x) @ w^T (+ bias)`` for a decode step's M <= 64 rows: final LayerNorm + LM head as one launch.  x fp32 [M, h], w bf16 [N, h]
    -> bf16 [M, N]."""
    M, h = x.shape
    N = w.shape[0]
    assert x.dtype == torch.float32 and x.is_contiguous() and w.dtype == torch.bfloat16 and w.is_contiguous() and w.shape[1] == h
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    assert out.shape == (M, N) and out.dtype == torch.bfloat16 and out.stride(1) == 1
    check(_lib.load().mafed_decode_ln_linear(_ptr(x), M, h, float(eps), _ptr(ln_w), _ptr(ln_b), _ptr(w), _ptr(bias), N, _ptr(out), out.stride(0),
                                             _stream()), "mafed_decode_ln_linear")
    return out


def decode_out_workspace(M: int, h: int, device) -> torch.Tensor:
    """Zero-filled workspace of ``decode_out`` (partial tiles + arrival counters; one per stream, re-usable across calls)."""
    return torch.zeros(int(_lib.load().mafed_decode_out_workspace_bytes(int(M), int(h))), dtype=torch.uint8, device=device)


def decode_out(x: torch.Tensor, ao: torch.Tensor, act: torch.Tensor, wd: torch.Tensor, bd: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
               workspace: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode step, last launch of a layer: ``x + bd + b2 + ao @ wd^T + act @ w2^T`` -> fp32 [M, h] (``out`` may be ``x``)."""
    M, h = x.shape
    n1 = act.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous() and ao.shape == (M, h) and ao.is_contiguous() and act.is_contiguous()
    assert ao.dtype == torch.bfloat16 and act.dtype == torch.bfloat16 and wd.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
    assert wd.shape == (h, h) and w2.shape == (h, n1) and wd.is_contiguous() and w2.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().mafed_decode_out(_ptr(x), _ptr(out), M, h, n1, _ptr(ao), _ptr(act), _ptr(wd), _ptr(bd), _ptr(w2), _ptr(b2),
                                       _ptr(workspace), workspace.numel(), _stream()), "mafed_decode_out")
    return out


class DecodeFlow:
    """Buffers and the per-layer pointer table of the one-launch decode step (``mafed_decode_flow_step``) for one K/V cache."""

    def __init__(self, layer_ptrs: Sequence[Sequence[torch.Tensor]], M: int, h: int, n1: int, H: int, D: int, V: int, device):
        lib = _lib.load()
        self.L = len(layer_ptrs) - 1
        self.M, self.h, self.n1, self.H, self.D, self.V = M, h, n1, H, D, V
        tab = [[0 if t is None else _ptr(t) for t in rec] for rec in layer_ptrs]
        assert all(len(r) == 14 for r in tab)
        self._keep = layer_ptrs                                   # the table holds raw addresses: keep the tensors alive
        self.table = torch.tensor(tab, dtype=torch.int64).to(device)
        z = lambda *s, dt=torch.bfloat16: torch.zeros(*s, dtype=dt, device=device)
        self.x = z(32, h, dt=torch.float32)
        self.ln1, self.ln2, self.ao, self.act = z(32, h), z(32, h), z(32, h), z(32, n1)
        self.ws = torch.zeros(int(lib.mafed_decode_flow_workspace_bytes(h, n1)), dtype=torch.uint8, device=device)
        self.flags = torch.zeros(int(lib.mafed_decode_flow_flag_bytes(self.L)) // 4, dtype=torch.int32, device=device)
        self.logits = z(M, V)

    def step(self, S0: int, cap: int, t: int, rot: int, P: int, cos, sin, attention_mask: torch.Tensor, eps: float) -> torch.Tensor:
        """x (rows < M filled by the caller) -> logits [M, V] bf16 (a static buffer: consume before the next step)."""
        self.flags.zero_()
        check(_lib.load().mafed_decode_flow_step(_ptr(self.table), self.L, self.M, self.h, self.n1, self.H, self.D, S0, cap, t, rot, P,
                                                 attention_mask.shape[1], self.V, float(eps), _ptr(self.x), _ptr(self.ln1), _ptr(self.ln2),
                                                 _ptr(self.act), _ptr(self.ao), _ptr(self.ws), self.ws.numel(), _ptr(self.flags),
                                                 self.flags.numel() * 4, _ptr(cos), _ptr(sin), _ptr(attention_mask), _ptr(self.logits), _stream()),
              "mafed_decode_flow_step")
        return self.logits

    def timed_out(self) -> bool:
        """Host check (synchronises): did a hand-over of the last step time out?"""
        return bool(int(self.flags[-1].item()) != 0)


class DecodeAttnOut:
    """Second launch of every decode layer (``mafed_decode_attn_out``): pointer table, scratch rows and one counter slot per (layer, step)."""

    def __init__(self, layer_ptrs: Sequence[Sequence[torch.Tensor]], M: int, h: int, n1: int, H: int, D: int, cap: int, device):
        lib = _lib.load()
        self.L, self.M, self.h, self.n1, self.H, self.D, self.cap = len(layer_ptrs), M, h, n1, H, D, cap
        tab = [[0 if t is None else _ptr(t) for t in rec] for rec in layer_ptrs]
        assert all(len(r) == 14 for r in tab)
        self._keep = layer_ptrs                                   # raw addresses in the table: keep the tensors alive
        self.table = torch.tensor(tab, dtype=torch.int64).to(device)
        self.act = torch.zeros(32, n1, dtype=torch.bfloat16, device=device)      # first launch writes rows < M
        self.ao = torch.zeros(32, h, dtype=torch.bfloat16, device=device)
        self.ws = torch.zeros(int(lib.mafed_decode_flow_workspace_bytes(h, n1)), dtype=torch.uint8, device=device)
        self.flags = torch.zeros(cap, self.L, 256, dtype=torch.int32, device=device)

    def begin_step(self, t: int) -> None:
        self.flags[t].zero_()     # the step's counter slots (a launch leaves them non-zero)

    def run(self, i: int, t: int, x: torch.Tensor, S0: int, rot: int, P: int, cos, sin, attention_mask: torch.Tensor) -> None:
        assert x.dtype == torch.float32 and x.is_contiguous() and x.shape == (self.M, self.h)
        check(_lib.load().mafed_decode_attn_out(self.table.data_ptr() + i * 14 * 8, self.M, self.h, self.n1, self.H, self.D, S0, self.cap, t, rot, P,
                                                attention_mask.shape[1], _ptr(x), _ptr(self.act), _ptr(self.ao), _ptr(self.ws), self.ws.numel(),
                                                self.flags[t, i].data_ptr(), _ptr(cos), _ptr(sin), _ptr(attention_mask), _stream()),
              "mafed_decode_attn_out")

    def timed_out(self) -> bool:
        return bool(int(self.flags[:, :, 255].abs().sum().item()) != 0)


def decode_flow_supported(M: int, h: int, n1: int, H: int, D: int, V: int, nk: int) -> bool:
    return bool(_lib.load().mafed_decode_flow_supported(int(M), int(h), int(n1), int(H), int(D), int(V), int(nk)))


def embed_concat_fwd(image: torch.Tensor, embed_in: torch.Tensor, input_ids: torch.Tensor, B: int, P: int, T: int) -> torch.Tensor:
    V, h = embed_in.shape
    h0 = torch.empty((B * (P + T), h), dtype=torch.float32, device=embed_in.device)
    check(_lib.load().mafed_embed_concat_fwd(_ptr(image), _dt(image), _ptr(embed_in), _ptr(input_ids), B, P, T, h, V, _ptr(h0),
                                             _stream()), "mafed_embed_concat_fwd")
    return h0


def embed_concat_bwd(dh0, input_ids, B, P, T, h, V, d_embed_in: Optional[torch.Tensor], img_dtype) -> torch.Tensor:
    d_image = torch.empty((B * P, h), dtype=img_dtype, device=dh0.device)
    check(_lib.load().mafed_embed_concat_bwd(_ptr(dh0), _ptr(input_ids), B, P, T, h, V, _ptr(d_image), _dt(d_image),
                                             _ptr(d_embed_in), _stream()), "mafed_embed_concat_bwd")
    return d_image


def ce_fwd(logits: torch.Tensor, labels: torch.Tensor, poison: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits [B,T,V] (text positions), labels [B,T] -> (loss[1], lse[B,T]).  ``poison`` (int32 [1], device): a non-zero flag turns the
    loss into NaN (mafed_ce_fwd_guarded: the row-sparse head's overflow flag)."""
    B, T, V = logits.shape
    assert logits.is_contiguous() and labels.is_contiguous() and labels.dtype == torch.int64
    lse = torch.empty((B, T), dtype=torch.float32, device=logits.device)
    row_loss = torch.empty((B, T), dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    if poison is None:
        check(_lib.load().mafed_ce_fwd(_ptr(logits), _dt(logits), _ptr(labels), B, T, V, _ptr(lse), _ptr(row_loss), _ptr(loss), _stream()),
              "mafed_ce_fwd")
    else:
        assert poison.dtype == torch.int32 and poison.numel() == 1
        check(_lib.load().mafed_ce_fwd_guarded(_ptr(logits), _dt(logits), _ptr(labels), B, T, V, _ptr(lse), _ptr(row_loss), _ptr(loss),
                                               _ptr(poison), _stream()), "mafed_ce_fwd_guarded")
    return loss, lse


def ce_bwd(logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, gloss: torch.Tensor,
           out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dlogits (pass out=logits for the in-place form)."""
    B, T, V = logits.shape
    if out is None:
        out = torch.empty_like(logits)
    check(_lib.load().mafed_ce_bwd(_ptr(logits), _dt(logits), _ptr(labels), _ptr(lse), B, T, V, _ptr(gloss), _ptr(out), _stream()),
          "mafed_ce_bwd")
    return out


def distill_fwd(s: torch.Tensor, t: torch.Tensor, attention_mask: torch.Tensor, P: int, cosine: bool = False,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> out[4] = {lang_sum, vision_sum, n_lang, n_vision}"""
    B, S, h = s.shape
    assert s.dtype == torch.float32 and t.dtype == torch.float32 and s.is_contiguous() and t.is_contiguous()
    lib = _lib.load()
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=s.device)
    nb = lib.mafed_distill_workspace_bytes(B * S)
    ws = workspace(s.device).get(nb)
    check(lib.mafed_distill_fwd(_ptr(s), _ptr(t), _ptr(attention_mask), B, S, P, h, int(cosine), _ptr(out), _ptr(ws), ws.numel(),
                                _stream()), "mafed_distill_fwd")
    return out


def distill_bwd(s, t, attention_mask, P: int, coef: torch.Tensor, cosine: bool = False, out: Optional[torch.Tensor] = None,
                accumulate: bool = False) -> torch.Tensor:
    B, S, h = s.shape
    if out is None:
        out = torch.empty_like(s)
        accumulate = False
    check(_lib.load().mafed_distill_bwd(_ptr(s), _ptr(t), _ptr(attention_mask), B, S, P, h, int(cosine), _ptr(coef), _ptr(out),
                                        int(accumulate), _stream()), "mafed_distill_bwd")
    return out


def distill_combine(sums: torch.Tensor, layer_coeff: torch.Tensor, mode: int, lang_weight: float = 0.5,
                    lang_weight_vec: Optional[torch.Tensor] = None):
    """sums [nl,4] -> (loss [1], per_layer [nl], modality [nl,2], inject [nl,4]) in one launch (mafed_distill_combine)."""
    nl = sums.shape[0]
    assert sums.dtype == torch.float32 and sums.is_contiguous() and layer_coeff.dtype == torch.float32 and layer_coeff.numel() == nl
    dev = sums.device
    out = torch.empty(1 + nl + 2 * nl + 4 + 4 * nl, dtype=torch.float32, device=dev)  # one allocation; inject 16-byte aligned
    loss, per_layer, modality = out[0:1], out[1:1 + nl], out[1 + nl:1 + 3 * nl].view(nl, 2)
    o = (1 + 3 * nl + 3) // 4 * 4
    inject = out[o:o + 4 * nl].view(nl, 4)
    check(_lib.load().mafed_distill_combine(_ptr(sums), nl, _ptr(layer_coeff), int(mode), float(lang_weight), _ptr(lang_weight_vec), _ptr(loss),
                                            _ptr(per_layer), _ptr(modality), _ptr(inject), _stream()), "mafed_distill_combine")
    return loss, per_layer, modality, inject


def distill_cls_fwd(s, t) -> torch.Tensor:
    B, S, h = s.shape
    out = torch.empty(1, dtype=torch.float32, device=s.device)
    check(_lib.load().mafed_distill_cls_fwd(_ptr(s), _ptr(t), B, S, h, _ptr(out), _stream()), "mafed_distill_cls_fwd")
    return out


def distill_cls_bwd(s, t, coef) -> torch.Tensor:
    B, S, h = s.shape
    out = torch.empty_like(s)
    check(_lib.load().mafed_distill_cls_bwd(_ptr(s), _ptr(t), B, S, h, _ptr(coef), _ptr(out), 0, _stream()), "mafed_distill_cls_bwd")
    return out


def ewc_penalty_fwd(p: torch.Tensor, p_old: torch.Tensor, fisher: torch.Tensor, half_lambda: float,
                    out: Optional[torch.Tensor] = None, beta: float = 0.0) -> torch.Tensor:
    """out[0] = beta * out[0] + half_lambda * sum fisher * (p - p_old)^2 over flat fp32 buffers."""
    assert p.dtype == torch.float32 and p.is_contiguous() and p_old.shape == p.shape and fisher.shape == p.shape
    lib = _lib.load()
    if out is None:
        out = torch.zeros(1, dtype=torch.float32, device=p.device)
    ws = workspace(p.device).get(lib.mafed_ewc_workspace_bytes(p.numel()))
    check(lib.mafed_ewc_penalty_fwd(_ptr(p), _ptr(p_old), _ptr(fisher), p.numel(), float(half_lambda), float(beta), _ptr(out), _ptr(ws),
                                    ws.numel(), _stream()), "mafed_ewc_penalty_fwd")
    return out


def ewc_penalty_bwd_(p: torch.Tensor, p_old: torch.Tensor, fisher: torch.Tensor, lam: float, coef: torch.Tensor, grad: torch.Tensor) -> None:
    """grad += coef[0] * lam * fisher * (p - p_old)"""
    assert grad.dtype == torch.float32 and grad.shape == p.shape and coef.dtype == torch.float32
    check(_lib.load().mafed_ewc_penalty_bwd(_ptr(p), _ptr(p_old), _ptr(fisher), p.numel(), float(lam), _ptr(coef), _ptr(grad), _stream()),
          "mafed_ewc_penalty_bwd")


def gradnorm_clip(g: torch.Tensor, max_norm: float, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> out[2] = {||g||, clip scale}"""
    lib = _lib.load()
    if out2 is None:
        out2 = torch.empty(2, dtype=torch.float32, device=g.device)
    nb = lib.mafed_gradnorm_workspace_bytes(g.numel())
    ws = workspace(g.device).get(nb)
    check(lib.mafed_gradnorm_clip(_ptr(g), g.numel(), float(max_norm), _ptr(out2), _ptr(ws), ws.numel(), _stream()), "mafed_gradnorm_clip")
    return out2


def pad_text_rows(src: torch.Tensor, B: int, S: int, P: int, lp_dtype=None):
    """[B*(S-P), h] fp32 -> ([B*S, h] fp32 with zero image rows, the same in ``lp_dtype`` (bf16) or None)"""
    h = src.shape[-1]
    dst = torch.empty((B * S, h), dtype=torch.float32, device=src.device)
    lp = torch.empty((B * S, h), dtype=lp_dtype, device=src.device) if lp_dtype == torch.bfloat16 else None
    check(_lib.load().mafed_pad_text_rows(_ptr(src), B, S, P, h, _ptr(dst), _ptr(lp), _stream()), "mafed_pad_text_rows")
    return dst, lp


def label_rows(labels: torch.Tensor, Rc: int):
    """labels [B,T] int64 -> (row_of_slot [B*Rc] int32, slot_of_row [B*T] int32, labels_c [B,Rc] int64, overflow [1] int32)"""
    B, T = labels.shape
    dev = labels.device
    ros = torch.empty(B * Rc, dtype=torch.int32, device=dev)
    sor = torch.empty(B * T, dtype=torch.int32, device=dev)
    lc = torch.empty((B, Rc), dtype=torch.int64, device=dev)
    ov = torch.zeros(1, dtype=torch.int32, device=dev)
    check(_lib.load().mafed_label_rows(_ptr(labels), B, T, Rc, _ptr(ros), _ptr(sor), _ptr(lc), _ptr(ov), _stream()), "mafed_label_rows")
    return ros, sor, lc, ov


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[r] = src[idx[r]] (zeros where idx[r] < 0); src [n,h] contiguous fp32 / bf16, idx int32"""
    h = src.shape[-1]
    dst = torch.empty((idx.numel(), h), dtype=src.dtype, device=src.device)
    check(_lib.load().mafed_gather_rows(_ptr(src), _dt(src), _ptr(idx), idx.numel(), h, _ptr(dst), _stream()), "mafed_gather_rows")
    return dst


def gradnorm_blocks(n: int) -> int:
    return int(_lib.load().mafed_gradnorm_blocks(int(n)))


def gradnorm_partial(g: torch.Tensor, partial_out: torch.Tensor) -> None:
    """sum-of-squares partials of the (contiguous, 16-byte aligned) range ``g`` -> partial_out[:gradnorm_blocks(g.numel())]"""
    check(_lib.load().mafed_gradnorm_partial(_ptr(g), g.numel(), _ptr(partial_out), _stream()), "mafed_gradnorm_partial")


def gradnorm_finish(partials: torch.Tensor, max_norm: float, out2: torch.Tensor, advance=None, norm_log: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fold the per-range partials into out2 = {norm, clip scale}.  ``advance`` = (state, base_lr, warmup, total, beta1, beta2, hyper):
    the optimiser's schedule advance rides in the same launch (mafed_gradnorm_finish_advance); ``norm_log`` = a 1-element slot that
    also receives the norm."""
    if advance is None:
        check(_lib.load().mafed_gradnorm_finish(_ptr(partials), partials.numel(), float(max_norm), _ptr(out2), _stream()), "mafed_gradnorm_finish")
        if norm_log is not None:
            norm_log.copy_(out2[0:1])
    else:
        state, base_lr, warmup, total, b1, b2, hyper = advance
        check(_lib.load().mafed_gradnorm_finish_advance(_ptr(partials), partials.numel(), float(max_norm), _ptr(out2), _ptr(norm_log), _ptr(state),
                                                        float(base_lr), int(warmup), int(total), float(b1), float(b2), _ptr(hyper), _stream()),
              "mafed_gradnorm_finish_advance")
    return out2


def adamw_step_(p, g, m, v, lr_dev, beta1, beta2, eps, weight_decay, step, clip=None, grad_mul=1.0, p_bf16=None, zero_grad: bool = False,
                zero_n: Optional[int] = None) -> None:
    """``zero_grad``: the kernel also writes zeros over ``g`` (the next window's optimizer.zero_grad(), same pass); ``zero_n``: only over
    its first ``zero_n`` elements (mafed_adamw_step_partial_zero: the rest is overwritten by the next window's weight-gradient GEMMs)."""
    if zero_grad and zero_n is not None and zero_n < p.numel():
        check(_lib.load().mafed_adamw_step_partial_zero(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(lr_dev), beta1, beta2, eps, weight_decay,
                                                        int(step), _ptr(clip), float(grad_mul), _ptr(p_bf16), int(zero_n), _stream()), "mafed_adamw_step")
        return
    fn = _lib.load().mafed_adamw_step_zero_grad if zero_grad else _lib.load().mafed_adamw_step
    check(fn(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(lr_dev), beta1, beta2, eps, weight_decay,
             int(step), _ptr(clip), float(grad_mul), _ptr(p_bf16), _stream()), "mafed_adamw_step")


def optim_advance_(state: torch.Tensor, base_lr: float, warmup: int, total: int, beta1: float, beta2: float, hyper: torch.Tensor,
                   clip: Optional[torch.Tensor] = None) -> None:
    """``clip`` = the {norm, scale} pair of this step's clip: a skipped step (scale < 0: non-finite norm) does not advance the counter."""
    check(_lib.load().mafed_optim_advance_guarded(_ptr(state), float(base_lr), int(warmup), int(total), float(beta1), float(beta2),
                                                  _ptr(hyper), _ptr(clip), _stream()), "mafed_optim_advance")


def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert src.is_contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.load().mafed_cast(_ptr(src), _dt(src), _ptr(out), _dt(out), src.numel(), _stream()), "mafed_cast")
    return out


def gelu(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    check(_lib.load().mafed_gelu(_ptr(x), _ptr(y), _dt(x), x.numel(), _stream()), "mafed_gelu")
    return y
