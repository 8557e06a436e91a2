"""HBM roofline of the EWC penalty kernels on the 410M flat buffer (run on the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops

n = 407_000_000 if len(sys.argv) < 2 else int(sys.argv[1])
dev = "cuda"
p, q = torch.randn(n, device=dev), torch.randn(n, device=dev)
f, grad = torch.rand(n, device=dev), torch.zeros(n, device=dev)
coef = torch.ones(1, device=dev)
out = torch.zeros(1, device=dev)


def timed(fn, reps=10):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


tf = timed(lambda: ops.ewc_penalty_fwd(p, q, f, 0.5, out=out))
tb = timed(lambda: ops.ewc_penalty_bwd_(p, q, f, 1.0, coef, grad))
print(f"ewc_penalty_fwd: {tf:.3f} ms  {12 * n / tf / 1e9:.2f} TB/s ({12 * n / 1e9:.2f} GB)   "
      f"ewc_penalty_bwd: {tb:.3f} ms  {20 * n / tb / 1e9:.2f} TB/s ({20 * n / 1e9:.2f} GB)")
