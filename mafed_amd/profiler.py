"""Per-kernel execution times of the library's own launches (``mafed_prof_*`` of include/mafed_hip.h).

    with KernelProfile() as prof:
        trainer.step(batch, i)
    table = prof.summary()      # synchronises, then {tag: {"launches", "total_ms", "avg_us", "work", "rate", ...}}

While the profile is open every hot kernel is launched with a start/stop event pair (hipExtLaunchKernelGGL); a pair's elapsed
time is the dispatch's own execution time on the GPU -- the quantity ``rocprofv3 --kernel-trace --stats`` reports -- so the
numbers stay meaningful when several HIP streams share the chip (a stream-level event bracket would also count queueing).
``work`` is algorithmic: flops for the MFMA kernels, bytes for the HBM-bound ones.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch

from mafed_amd import _lib

# roofline each kernel tag is priced against: ("mfma", TFLOP/s peak) or ("hbm", GB/s peak) -- MI355X_MICROARCH.md chip table
PEAK_BF16_TFLOPS = 2500.0
PEAK_F32_MATRIX_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
MFMA_TAGS = {"gemm_bf16": PEAK_BF16_TFLOPS, "gemm_pp": PEAK_BF16_TFLOPS, "gemm_skinny": PEAK_BF16_TFLOPS, "attn_fwd": PEAK_BF16_TFLOPS, "attn_bwd_dq": PEAK_BF16_TFLOPS,
             "attn_bwd_dkv": PEAK_BF16_TFLOPS, "gemm_f32": PEAK_F32_MATRIX_TFLOPS}


class KernelProfile:
    def __init__(self, max_records: int = 1 << 16):
        self.max_records = int(max_records)
        self._records = None
        self.starts_ms: List[float] = []  # start of every record relative to the first (same order as records())

    def __enter__(self):
        _lib.check(_lib.load().mafed_prof_begin(self.max_records), "mafed_prof_begin")
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.load().mafed_prof_end(), "mafed_prof_end")
        return False

    def records(self) -> List[Tuple[str, float, float]]:
        """[(tag, work, ms)] in launch order; synchronises the device first."""
        if self._records is None:
            torch.cuda.synchronize()
            lib = _lib.load()
            n = lib.mafed_prof_collect(None, None, None, None, 0)
            tags, work, ms, st = (C.c_int * n)(), (C.c_double * n)(), (C.c_float * n)(), (C.c_float * n)()
            lib.mafed_prof_collect(C.cast(tags, C.c_void_p), C.cast(work, C.c_void_p), C.cast(ms, C.c_void_p), C.cast(st, C.c_void_p), n)
            self.starts_ms = [float(st[i]) for i in range(n)]
            names = {}
            out = []
            for i in range(n):
                t = tags[i]
                if t not in names:
                    names[t] = lib.mafed_prof_tag_name(t).decode()
                out.append((names[t], float(work[i]), float(ms[i])))
            self._records = out
        return self._records

    def summary(self) -> Dict[str, Dict[str, float]]:
        agg: Dict[str, Dict[str, float]] = {}
        for tag, work, ms in self.records():
            if ms < 0:
                continue
            a = agg.setdefault(tag, {"launches": 0, "total_ms": 0.0, "work": 0.0})
            a["launches"] += 1
            a["total_ms"] += ms
            a["work"] += work
        for tag, a in agg.items():
            a["avg_us"] = a["total_ms"] * 1e3 / max(1, a["launches"])
            sec = a["total_ms"] * 1e-3
            if tag in MFMA_TAGS:
                a["bound"], a["unit"], a["peak"] = "mfma", "TFLOP/s", MFMA_TAGS[tag]
                a["achieved"] = a["work"] / sec / 1e12 if sec > 0 else 0.0
            else:
                a["bound"], a["unit"], a["peak"] = "hbm", "GB/s", PEAK_HBM_GBS
                a["achieved"] = a["work"] / sec / 1e9 if sec > 0 else 0.0
            a["frac"] = a["achieved"] / a["peak"]
        return agg
