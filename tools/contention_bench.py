"""How the GEMMs of the step behave when part of the chip is taken by another stream's kernel (under data parallelism RCCL's all-reduce
kernels hold one workgroup per channel for milliseconds while the backward's GEMMs run; in the single-GPU step the HBM-bound passes of
the side streams do the same for tens of microseconds).  Two occupiers on a side stream:
  idle n   : mafed_tune_occupy -- n CUs, one sleeping 512-thread block with 96 KiB of LDS each
  stream n : mafed_tune_stream -- n workgroups of 1024 threads streaming four read + four written fp32 streams (AdamW-shaped, what a
             collective's copy / reduce loop looks like to the memory system)
against three GEMM modes: the 128 x 128 kernels (variant 700), the persistent kernels in static tile order (701 + 720) and in ticketed
order (701 + 721)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

dev = "cuda"
lib = _lib.load()
M = 9216
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(s, device=dev, generator=g).to(BF)
SHAPES = [("qkv", False, True, M, 3072, 1024), ("dense", False, True, M, 1024, 1024), ("fc1", False, True, M, 4096, 1024),
          ("dfc1", False, False, M, 1024, 4096), ("dqkv", False, False, M, 1024, 3072)]
side = torch.cuda.Stream()
OCC = [("idle", int(v)) for v in os.environ.get("CONTENTION_OCC", "8,32").split(",") if v] + \
      [("stream", int(v)) for v in os.environ.get("CONTENTION_STREAM", "16,32").split(",") if v]
nbuf = 1 << 30
sbuf = torch.empty(nbuf, dtype=torch.uint8, device=dev)
sbuf.view(torch.float32).normal_()
dbuf = torch.empty_like(sbuf)
MODES = [("128x128", 700, 721), ("pp static", 701, 720), ("pp ticket", 701, 721)]


def timed(fn, occ, reps=8):
    torch.cuda.synchronize()
    if occ is not None:
        kind, n = occ
        with torch.cuda.stream(side):
            if kind == "idle":
                _lib.check(lib.mafed_tune_occupy(n, 96 * 1024, int(8e6), side.cuda_stream), "occupy")   # ~4 ms at 2 GHz
            else:
                for _ in range(3):   # 3 x 2 GiB moved by n workgroups: several ms
                    _lib.check(lib.mafed_tune_stream(sbuf.data_ptr(), dbuf.data_ptr(), nbuf, n, 1024, 4, 2, side.cuda_stream), "stream")
        torch.cuda._sleep(300000)   # let the occupier's blocks take their CUs first (~0.15 ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def set_mode(v, tk):
    lib.mafed_gemm_set_variant(v)
    lib.mafed_gemm_set_variant(tk)


print("# us per product; (x) = against the same mode with a free chip", flush=True)
for name, tA, tB, m, n, k in SHAPES:
    A = rn(*((k, m) if tA else (m, k)))
    B = rn(*((n, k) if tB else (k, n)))
    out = torch.empty((m, n), dtype=BF, device=dev)
    fn = lambda: ops.gemm(A, B, tA, tB, out=out)
    row = f"{name:6s}"
    for label, v, tk in MODES:
        set_mode(v, tk)
        fn()
        base = min(timed(fn, None) for _ in range(3))
        row += f" | {label}: free {base:6.1f}"
        for occ in OCC:
            t = min(timed(fn, occ) for _ in range(2))
            row += f", {occ[0]} {occ[1]}: {t:6.1f} ({t / base:4.2f}x)"
    print(row, flush=True)
set_mode(701, 721)
# grouped weight gradients of two layers
probs = []
for l in range(2):
    dy, du, dqkv = rn(M, 1024), rn(M, 4096), rn(M, 3072)
    a, ao, ln2, ln1 = rn(M, 4096), rn(M, 1024), rn(M, 1024), rn(M, 1024)
    for dY, X in ((dy, a), (du, ln2), (dqkv, ln1), (dy, ao)):
        probs.append(dict(A=dY, B=X, out=torch.zeros(dY.shape[1], X.shape[1], device=dev), beta=1.0))
grp = lambda: ops.gemm_grouped(probs, True, False)
each = lambda: [ops.gemm(p["A"], p["B"], True, False, out=p["out"], beta=1.0) for p in probs]
row = "dW x8 "
for label, v, tk, fn in (("one launch each, 128x128", 700, 721, each), ("grouped pp static", 701, 720, grp), ("grouped pp ticket", 701, 721, grp)):
    set_mode(v, tk)
    fn()
    b0 = min(timed(fn, None, 3) for _ in range(3))
    row += f" | {label}: free {b0:6.1f}"
    for occ in OCC:
        t = min(timed(fn, occ, 3) for _ in range(2))
        row += f", {occ[0]} {occ[1]}: {t:6.1f} ({t / b0:4.2f}x)"
print(row, flush=True)
set_mode(701, 721)
