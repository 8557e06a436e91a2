"""Decode-path skinny products (M = 32 rows) of the 410M model in isolation (run on the GPU box): python3 tools/skinny_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
M = 32
SH = [("qkv", 3072, 1024, torch.bfloat16, {}), ("dense", 1024, 1024, torch.bfloat16, {}), ("fc1", 4096, 1024, torch.bfloat16, {"gelu": True}),
      ("fc2", 1024, 4096, torch.float32, {"res": True}), ("head", 50304, 1024, torch.bfloat16, {})]
for name, N, K, od, opt in SH:
    X = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    Ws = [torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(24)]   # 24 layers' weights: every call streams cold weights
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias)
    if opt.get("gelu"):
        kw["epilogue"] = ops.EPI_GELU
    if opt.get("res"):
        kw["res1"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
        kw["res2"] = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, dtype=od, device=dev)
    res = []
    for ns, wide in ((0, 699), (16, 600), (16, 601), (4, 600), (4, 601)):
        lib.mafed_gemm_set_variant(500 + ns)
        lib.mafed_gemm_set_variant(wide)
        for W in Ws[:3]:
            ops.gemm(X, W, False, True, out=out, **kw)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for W in Ws:
                ops.gemm(X, W, False, True, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / len(Ws))
        res.append(f"ns{ns or 'A'}/{'w' if wide == 601 else ('a' if wide == 699 else 'n')} {best * 1e3:5.1f}")
    lib.mafed_gemm_set_variant(500)
    lib.mafed_gemm_set_variant(699)
    print(f"{name:6s} N={N:6d} K={K:5d} ({2.0 * N * K / 1e6:6.1f} MB of W): " + "  ".join(res) + " us", flush=True)
