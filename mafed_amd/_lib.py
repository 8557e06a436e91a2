"""ctypes binding of the C-ABI library ``libmafed_hip.so`` (include/mafed_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAFED_HIP_LIB") or os.path.join(_HERE, "libmafed_hip.so")  # override: A/B of two builds

F32, BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_QUICK_GELU = 0, 1, 2, 3
EPI_RES1_BF16 = 0x100
EPI_NO_PERSISTENT = 0x200
EPI_TICKETED = 0x400

_p, _i, _l, _f, _z, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_double

# name -> (restype, argtypes); mirrors include/mafed_hip.h one to one
SIGNATURES = {
    "mafed_version": (_i, []),
    "mafed_last_error_string": (C.c_char_p, []),
    "mafed_gemm": (_i, [_i, _i, _i, _l, _l, _l, _p, _l, _p, _l, _p, _l, _i, _p, _i, _p, _p, _p, _f, _p]),
    "mafed_gemm_colsum": (_i, [_i, _i, _i, _l, _l, _l, _p, _l, _p, _l, _p, _l, _i, _p, _i, _p, _p, _p, _f, _p, _p]),
    "mafed_gemm_grouped": (_i, [_i, _i, _i, _i, _p, _i, _p]),
    "mafed_gemm_set_variant": (_i, [_i]),
    "mafed_gemm_get_variant": (_i, [_i]),
    "mafed_gemm_pp_launches": (_i, []),
    "mafed_attn_decode": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "mafed_attn_decode_prerot": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "mafed_rotate_k_rows": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mafed_gemm_grouped_fuses_sumsq": (_i, [_i, _i, _i, _i, _p, _p, _p, _i]),
    "mafed_decode_supported": (_i, [_i, _i, _i]),
    "mafed_decode_ln_qkv_fc1": (_i, [_p, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _i, _p, _p]),
    "mafed_decode_ln_linear": (_i, [_p, _i, _i, _f, _p, _p, _p, _p, _l, _p, _l, _p]),
    "mafed_decode_out_workspace_bytes": (_z, [_i, _i]),
    "mafed_decode_set_trace": (_i, [_p]),
    "mafed_attn_decode_set_trace": (_i, [_p]),
    "mafed_decode_out": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "mafed_decode_flow_supported": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "mafed_decode_attn_out": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p, _p, _p, _p, _p]),
    "mafed_decode_flow_grid": (_z, [_i, _i, _i, _i, _i, _i]),
    "mafed_decode_flow_set_trace": (_i, [_p]),
    "mafed_decode_flow_flag_bytes": (_z, [_i]),
    "mafed_decode_flow_workspace_bytes": (_z, [_i, _i]),
    "mafed_decode_flow_step": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _z, _p, _z, _p, _p, _p, _p, _p]),
    "mafed_ewc_workspace_bytes": (_z, [_l]),
    "mafed_ewc_penalty_fwd": (_i, [_p, _p, _p, _l, _f, _f, _p, _p, _z, _p]),
    "mafed_ewc_penalty_bwd": (_i, [_p, _p, _p, _l, _f, _p, _p, _p]),
    "mafed_colsum_workspace_bytes": (_z, [_l, _l]),
    "mafed_colsum": (_i, [_p, _i, _l, _l, _l, _p, _p, _z, _p]),
    "mafed_layernorm_fwd": (_i, [_p, _l, _i, _f, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "mafed_layernorm_bwd_workspace_bytes": (_z, [_l, _i]),
    "mafed_layernorm_bwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _f, _p, _p, _p, _z, _p]),
    "mafed_layernorm_bwd_rows": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _f, _i, _p, _z, _p]),
    "mafed_layernorm_bwd_params": (_i, [_l, _i, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "mafed_attn_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "mafed_attn_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "mafed_attn_bwd_colsum": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p]),
    "mafed_attn_set_variant": (_i, [_i]),
    "mafed_attn_fwd_exact_bf16": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p]),
    "mafed_embed_concat_fwd": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _l, _p, _p]),
    "mafed_embed_concat_bwd": (_i, [_p, _p, _i, _i, _i, _i, _l, _p, _i, _p, _p]),
    "mafed_ce_fwd": (_i, [_p, _i, _p, _i, _i, _l, _p, _p, _p, _p]),
    "mafed_ce_fwd_guarded": (_i, [_p, _i, _p, _i, _i, _l, _p, _p, _p, _p, _p]),
    "mafed_ce_bwd": (_i, [_p, _i, _p, _p, _i, _i, _l, _p, _p, _p]),
    "mafed_distill_workspace_bytes": (_z, [_l]),
    "mafed_distill_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _z, _p]),
    "mafed_distill_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "mafed_distill_cls_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "mafed_distill_cls_bwd": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "mafed_gradnorm_workspace_bytes": (_z, [_l]),
    "mafed_gradnorm_clip": (_i, [_p, _l, _f, _p, _p, _z, _p]),
    "mafed_gradnorm_blocks": (_i, [_l]),
    "mafed_gradnorm_partial": (_i, [_p, _l, _p, _p]),
    "mafed_gradnorm_finish": (_i, [_p, _i, _f, _p, _p]),
    "mafed_sumsq_accumulate": (_i, [_p, _l, _p, _p]),
    "mafed_adamw_step": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _i, _p, _f, _p, _p]),
    "mafed_adamw_step_zero_grad": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _i, _p, _f, _p, _p]),
    "mafed_adamw_step_partial_zero": (_i, [_p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _i, _p, _f, _p, _l, _p]),
    "mafed_distill_combine": (_i, [_p, _i, _p, _i, _f, _p, _p, _p, _p, _p, _p]),
    "mafed_optim_advance": (_i, [_p, _d, _l, _l, _d, _d, _p, _p]),
    "mafed_optim_advance_guarded": (_i, [_p, _d, _l, _l, _d, _d, _p, _p, _p]),
    "mafed_tune_occupy": (_i, [_i, _i, C.c_longlong, _p]),
    "mafed_tune_stream": (_i, [_p, _p, C.c_longlong, _i, _i, _i, _i, _p]),
    "mafed_tune_stream_for": (_i, [_p, _p, C.c_longlong, _i, _i, _d, _p]),
    "mafed_gradnorm_finish_advance": (_i, [_p, _i, _f, _p, _p, _p, _d, _l, _l, _d, _d, _p, _p]),
    "mafed_cast": (_i, [_p, _i, _p, _i, _l, _p]),
    "mafed_pad_text_rows": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "mafed_label_rows": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "mafed_gather_rows": (_i, [_p, _i, _p, _l, _i, _p, _p]),
    "mafed_gelu": (_i, [_p, _p, _i, _l, _p]),
    "mafed_patchify": (_i, [_p, _i, _i, _i, _i, _i, _i, _l, _i, _p, _i, _p]),
    "mafed_vit_assemble": (_i, [_p, _i, _l, _p, _p, _i, _i, _i, _p, _p]),
    "mafed_attn_fwd_bidir": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mafed_prof_begin": (_i, [_i]),
    "mafed_prof_end": (_i, []),
    "mafed_prof_collect": (_i, [_p, _p, _p, _p, _i]),
    "mafed_prof_tag_name": (C.c_char_p, [_i]),
}



class GemmProblem(C.Structure):
    """``mafed_gemm_problem`` of include/mafed_hip.h (one entry of a grouped launch)."""
    _fields_ = [("M", _l), ("N", _l), ("K", _l), ("A", _p), ("lda", _l), ("B", _p), ("ldb", _l), ("C", _p), ("ldc", _l),
                ("bias", _p), ("epilogue", _i), ("aux", _p), ("res1", _p), ("res2", _p), ("beta", _f), ("colsum", _p), ("sumsq", _p)]


_lib: Optional[C.CDLL] = None


class MafedHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library and bind every exported entry point; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MafedHipError(
            f"{LIB_PATH} is missing: build it with `python -m mafed_amd.build` (or __graft_entry__.build()). "
            "mafed_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    loose = bool(os.environ.get("MAFED_HIP_LIB")) and os.environ.get("MAFED_HIP_LIB_LOOSE") == "1"   # tools: A/B against an older build of the library
    for name, (res, args) in SIGNATURES.items():
        if loose and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mafed_last_error_string()
        raise MafedHipError(f"{what or 'mafed call'} failed (rc={rc}): {msg.decode() if msg else ''}")
