"""Micro-benchmark of the bf16 MFMA GEMM on the shapes of the 410M MAFED step (run on the GPU box).
Interleaved rounds in one process (cdna_hip_programming rule 24), random data (rule 25)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

dev = "cuda"
M = int(os.environ.get("GEMM_BENCH_M", "9216"))   # rows = batch x (image + text tokens): 9216 = 32 x 288, 4608 = 16 x 288
SHAPES = [  # (name, transA, transB, M, N, K, out_dtype)
    ("qkv   NT", False, True, M, 3072, 1024, torch.bfloat16),
    ("dense NT", False, True, M, 1024, 1024, torch.bfloat16),
    ("fc1   NT", False, True, M, 4096, 1024, torch.bfloat16),
    ("fc2   NT", False, True, M, 1024, 4096, torch.float32),
    ("dfc2  NN", False, False, M, 4096, 1024, torch.bfloat16),
    ("dfc1  NN", False, False, M, 1024, 4096, torch.bfloat16),
    ("dqkv  NN", False, False, M, 1024, 3072, torch.bfloat16),
    ("dao   NN", False, False, M, 1024, 1024, torch.bfloat16),
    ("wfc2  TN", True, False, 1024, 4096, M, torch.float32),
    ("wfc1  TN", True, False, 4096, 1024, M, torch.float32),
    ("wqkv  TN", True, False, 3072, 1024, M, torch.float32),
    ("wdns  TN", True, False, 1024, 1024, M, torch.float32),
    ("head  NT", False, True, 1024, 50304, 1024, torch.bfloat16),
    ("dlnf  NN", False, False, 1024, 1024, 50304, torch.float32),   # LM-head dX: accumulate-only fp32 output, K split (as model.py issues it)
    ("proj  NT", False, True, 8192, 1024, 1024, torch.bfloat16),
]
EPI = os.environ.get("GEMM_BENCH_EPI", "0") == "1"   # fc1 with bias + GELU + saved pre-activation, dfc2 with GELU'
LIBREF = os.environ.get("GEMM_BENCH_LIB", "0") == "1"  # add a torch.matmul (hipBLASLt/rocBLAS) column as a yardstick
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
lib = _lib.load()
for pv in os.environ.get("GEMM_BENCH_PRE", "").split(","):
    if pv:
        lib.mafed_gemm_set_variant(int(pv))   # persistent knobs (split / group / desync ...), applied before the sweep
g = torch.Generator(device=dev).manual_seed(0)
res = {}
ONLY = [x for x in os.environ.get("GEMM_BENCH_ONLY", "").split(",") if x]
for name, tA, tB, m, n, k, od in SHAPES:
    if ONLY and name.split()[0] not in ONLY:
        continue
    A = torch.randn((k, m) if tA else (m, k), device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn((n, k) if tB else (k, n), device=dev, generator=g).to(torch.bfloat16)
    out = torch.zeros((m, n), dtype=od, device=dev)
    ref = None
    beta = 1.0 if ((tA or name.startswith("dlnf")) and od == torch.float32) else 0.0   # weight-gradient GEMMs accumulate into the gradient buffer
    kw = {}
    if EPI and name.startswith("fc1"):
        kw = dict(bias=torch.randn(n, device=dev), epilogue=ops.EPI_GELU, aux=torch.empty((m, n), dtype=od, device=dev))
    elif EPI and name.startswith("dfc2"):
        kw = dict(epilogue=ops.EPI_GELU_BWD, aux=torch.randn((m, n), device=dev).to(od))
        if os.environ.get("GEMM_BENCH_NOCOLSUM", "0") != "1":
            kw["colsum"] = torch.zeros(n, device=dev)
    elif EPI and name.startswith("fc2"):
        kw = dict(bias=torch.randn(n, device=dev), res1=torch.randn((m, n), device=dev).to(torch.bfloat16), res2=torch.randn((m, n), device=dev))
    elif EPI and (name.startswith("qkv") or name.startswith("dense")):
        kw = dict(bias=torch.randn(n, device=dev))
    for v in variants:
        lib.mafed_gemm_set_variant(v)
        out.zero_()
        ops.gemm(A, B, tA, tB, out=out, beta=beta, **kw)
        if ref is None:
            ref = out.float().clone()
        elif (v < 30 or v >= 100) and os.environ.get("GEMM_BENCH_NOCHECK", "0") != "1":
            err = (out.float() - ref).abs().max().item()
            assert err <= 1e-2 * ref.abs().max().item(), (name, v, err)
    times = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            lib.mafed_gemm_set_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(A, B, tA, tB, out=out, beta=beta, **kw)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * m * n * k
    lib_col = ""
    if LIBREF:   # yardstick only: the vendor library on the same operands (plain product, no epilogue, bf16 output)
        At, Bt = (A.t() if tA else A), (B.t() if tB else B)
        o2 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        tl = []
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                torch.matmul(At, Bt, out=o2)
            e1.record()
            torch.cuda.synchronize()
            tl.append(e0.elapsed_time(e1) / 10)
        lib_col = f"  lib: {min(tl)*1e3:7.1f} us {fl/min(tl)/1e9:7.1f} TF"
    line = f"{name} {m}x{n}x{k}: " + "  ".join(f"v{v}: {min(t)*1e3:7.1f} us {fl/min(t)/1e9:7.1f} TF" for v, t in times.items())
    print(line + lib_col, flush=True)
lib.mafed_gemm_set_variant(0)
lib.mafed_gemm_set_variant(100)
lib.mafed_gemm_set_variant(701)
