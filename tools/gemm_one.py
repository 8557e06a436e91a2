"""Run one bf16 GEMM shape with one kernel variant N times (for rocprofv3 --pmc runs)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

variant, tA, tB, m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((k, m) if tA else (m, k), device="cuda", generator=g).to(torch.bfloat16)
B = torch.randn((n, k) if tB else (k, n), device="cuda", generator=g).to(torch.bfloat16)
out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
_lib.load().mafed_gemm_set_variant(variant)
for _ in range(reps):
    ops.gemm(A, B, bool(tA), bool(tB), out=out)
torch.cuda.synchronize()
print("done")
