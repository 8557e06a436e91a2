"""Kernel execution time (in-library start/stop events per launch) beside the event-bracketed loop time (kernel + launch-to-launch gap)
for the step's GEMM shapes: separates a slower kernel from a slower hand-over when two builds of the library are compared
(MAFED_HIP_LIB=... MAFED_HIP_LIB_LOOSE=1)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import _lib, ops
from mafed_amd.profiler import KernelProfile

dev = "cuda"
lib = _lib.load()
for pv in os.environ.get("GEMM_BENCH_PRE", "").split(","):
    if pv:
        lib.mafed_gemm_set_variant(int(pv))
M = 9216
g = torch.Generator(device=dev).manual_seed(0)
SHAPES = [("qkv", False, True, M, 3072, 1024), ("fc1", False, True, M, 4096, 1024), ("dfc1", False, False, M, 1024, 4096), ("dao", False, False, M, 1024, 1024)]
tag = os.environ.get("TAG", "")
for name, tA, tB, m, n, k in SHAPES:
    A = torch.randn((k, m) if tA else (m, k), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((n, k) if tB else (k, n), device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm(A, B, tA, tB, out=out)
    loop = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(A, B, tA, tB, out=out)
        e1.record()
        torch.cuda.synchronize()
        loop.append(e0.elapsed_time(e1) / 20 * 1e3)
    with KernelProfile() as kp:
        for _ in range(40):
            ops.gemm(A, B, tA, tB, out=out)
        torch.cuda.synchronize()
    ks = sorted(ms * 1e3 for _, _, ms in kp.records())
    print(f"{tag:12s} {name:5s} loop (kernel + gap) min {min(loop):6.1f} us | kernel alone: min {ks[0]:6.1f} median {ks[len(ks) // 2]:6.1f} us", flush=True)
