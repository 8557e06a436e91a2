"""How the GEMMs of the step behave when part of the chip is taken by a long-running kernel (the situation under data parallelism:
RCCL's all-reduce kernels hold one workgroup per channel for milliseconds while the backward's GEMMs run).  An occupier kernel
(mafed_tune_occupy: `n` CUs, one 512-thread block with 96 KiB of LDS each, ~3 ms) runs on a side stream; the GEMM is timed on the main
stream meanwhile, persistent kernels (variant 701) against the 128 x 128 kernels (700)."""
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
          ("dfc1", False, False, M, 1024, 4096), ("dao", False, False, M, 1024, 1024)]
side = torch.cuda.Stream()
# negative entries: that many LIGHT blocks (256 threads, a handful of registers, no LDS) -- can small kernels share CUs with a persistent block?
OCC = [int(v) for v in os.environ.get("CONTENTION_OCC", "8,16,32").split(",")]


def timed(fn, occupy, reps=8):
    torch.cuda.synchronize()
    if occupy:
        with torch.cuda.stream(side):
            _lib.check(lib.mafed_tune_occupy(occupy, 96 * 1024 if occupy > 0 else 0, int(6e6), side.cuda_stream), "occupy")   # ~3 ms at 2 GHz
        torch.cuda._sleep(200000)   # let the occupier's blocks take their CUs first (~0.1 ms)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, tA, tB, m, n, k in SHAPES:
    A = rn(*((k, m) if tA else (m, k)))
    B = rn(*((n, k) if tB else (k, n)))
    out = torch.empty((m, n), dtype=BF, device=dev)
    row = f"{name:6s}"
    for v in (700, 701):
        lib.mafed_gemm_set_variant(v)
        fn = lambda: ops.gemm(A, B, tA, tB, out=out)
        fn()
        base = min(timed(fn, 0) for _ in range(3))
        row += f" | v{v}: free {base:6.1f} us"
        for occ in OCC:
            t = min(timed(fn, occ) for _ in range(3))
            row += f", {occ} CUs taken {t:6.1f} ({t / base:4.2f}x)"
    print(row, flush=True)
lib.mafed_gemm_set_variant(701)
# grouped weight gradients of two layers
probs = []
for l in range(2):
    dy, du, dqkv = rn(M, 1024), rn(M, 4096), rn(M, 3072)
    a, ao, ln2, ln1 = rn(M, 4096), rn(M, 1024), rn(M, 1024), rn(M, 1024)
    for dY, X in ((dy, a), (du, ln2), (dqkv, ln1), (dy, ao)):
        probs.append(dict(A=dY, B=X, out=torch.zeros(dY.shape[1], X.shape[1], device=dev), beta=1.0))
grp = lambda: ops.gemm_grouped(probs, True, False)
lib.mafed_gemm_set_variant(700)
each = lambda: [ops.gemm(p["A"], p["B"], True, False, out=p["out"], beta=1.0) for p in probs]
each()
row = "dW x8 "
b0 = min(timed(each, 0, 3) for _ in range(3))
row += f" | one launch each (128 x 128 kernels): free {b0:6.1f} us"
for occ in (8, 16, 32):
    t = min(timed(each, occ, 3) for _ in range(3))
    row += f", {occ} taken {t:6.1f} ({t / b0:4.2f}x)"
lib.mafed_gemm_set_variant(701)
grp()
b1 = min(timed(grp, 0, 3) for _ in range(3))
row += f" | grouped persistent: free {b1:6.1f} us"
for occ in (8, 16, 32):
    t = min(timed(grp, occ, 3) for _ in range(3))
    row += f", {occ} taken {t:6.1f} ({t / b1:4.2f}x)"
print(row, flush=True)
