"""Run ONE product of the 410M step (by name, with the epilogue the step gives it) N times under the automatic dispatch -- the
target of the per-shape rocprofv3 --pmc passes (tools/pmc_traffic.sh).  `wgrp2` = the grouped weight-gradient launch of two layers."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

M = 9216
BF, F32 = torch.bfloat16, torch.float32
SHAPES = {  # name: (transA, transB, M, N, K, out dtype)
    "qkv": (False, True, M, 3072, 1024, BF), "dense": (False, True, M, 1024, 1024, BF), "fc1": (False, True, M, 4096, 1024, BF),
    "fc2": (False, True, M, 1024, 4096, F32), "dfc2": (False, False, M, 4096, 1024, BF), "dfc1": (False, False, M, 1024, 4096, BF),
    "dqkv": (False, False, M, 1024, 3072, BF), "dao": (False, False, M, 1024, 1024, BF),
}
name, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(s, device=dev, generator=g)
lib = _lib.load()
lib.mafed_gemm_set_variant(int(os.environ.get("GEMM_VARIANT", "701")))
if name.startswith("wgrp"):
    L = int(name[4:])
    probs = []
    for l in range(L):
        dy, du, dqkv = rn(M, 1024).to(BF), rn(M, 4096).to(BF), rn(M, 3072).to(BF)
        a, ao, ln2, ln1 = rn(M, 4096).to(BF), rn(M, 1024).to(BF), rn(M, 1024).to(BF), rn(M, 1024).to(BF)
        for dY, X in ((dy, a), (du, ln2), (dqkv, ln1), (dy, ao)):
            probs.append(dict(A=dY, B=X, out=torch.zeros(dY.shape[1], X.shape[1], device=dev), beta=1.0))
    fn = lambda: ops.gemm_grouped(probs, True, False)
else:
    tA, tB, m, n, k, od = SHAPES[name]
    A = rn(*((k, m) if tA else (m, k))).to(BF)
    B = rn(*((n, k) if tB else (k, n))).to(BF)
    out = torch.zeros((m, n), dtype=od, device=dev)
    kw = {}
    if name == "fc1":
        kw = dict(bias=rn(n), epilogue=ops.EPI_GELU, aux=torch.empty((m, n), dtype=od, device=dev))
    elif name == "dfc2":
        kw = dict(epilogue=ops.EPI_GELU_BWD, aux=rn(m, n).to(od), colsum=torch.zeros(n, device=dev))
    elif name == "fc2":
        kw = dict(bias=rn(n), res1=rn(m, n).to(BF), res2=rn(m, n))
    elif name in ("qkv", "dense"):
        kw = dict(bias=rn(n))
    fn = lambda: ops.gemm(A, B, tA, tB, out=out, **kw)
n0 = lib.mafed_gemm_pp_launches()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print(f"{name}: {reps} launches, {lib.mafed_gemm_pp_launches() - n0} on the persistent kernel")
