"""Grouped weight-gradient launch of the ping-pong kernel: the dW += dY^T.X products of `L` layers (wfc2, wfc1, wqkv, wdense each)
as ONE persistent launch (mafed_gemm_grouped) against the same products one launch each (old path / new path)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

dev = "cuda"
M = 9216
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lib = _lib.load()
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(s, device=dev, generator=g).to(torch.bfloat16)
probs = []
for l in range(L):
    dy, du, dqkv = rn(M, 1024), rn(M, 4096), rn(M, 3072)
    a, ao, ln2, ln1 = rn(M, 4096), rn(M, 1024), rn(M, 1024), rn(M, 1024)
    for dY, X in ((dy, a), (du, ln2), (dqkv, ln1), (dy, ao)):
        probs.append(dict(A=dY, B=X, out=torch.zeros(dY.shape[1], X.shape[1], device=dev), beta=1.0))
fl = sum(2.0 * M * p["out"].numel() for p in probs)


def run_each():
    for p in probs:
        ops.gemm(p["A"], p["B"], True, False, out=p["out"], beta=1.0)


def run_grouped():
    ops.gemm_grouped(probs, True, False)


def timeit(fn, n=5):
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


lib.mafed_gemm_set_variant(700)
run_each(); ref = [p["out"].clone() for p in probs]
for p in probs: p["out"].zero_()
t_old = timeit(run_each)
lib.mafed_gemm_set_variant(701)
for p in probs: p["out"].zero_()
n0 = lib.mafed_gemm_pp_launches()
run_grouped()
assert lib.mafed_gemm_pp_launches() == n0 + 1, "grouped launch did not take the ping-pong kernel"
for p, r in zip(probs, ref):
    err = (p["out"] - r).abs().max().item()
    assert err <= 2e-2 * r.abs().max().item(), err
t_grp = timeit(run_grouped)
t_new_each = timeit(run_each)
print(f"{L} layers of dW ({len(probs)} products, {fl / 1e9:.0f} GFLOP): one launch each, old kernel {t_old * 1e3:.1f} us {fl / t_old / 1e9:.0f} TF | "
      f"one launch each, automatic {t_new_each * 1e3:.1f} us {fl / t_new_each / 1e9:.0f} TF | grouped {t_grp * 1e3:.1f} us {fl / t_grp / 1e9:.0f} TF")
