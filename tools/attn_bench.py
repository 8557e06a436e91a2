"""Isolated timing of the attention kernels at the 410M MAFED step shape (run on the GPU box; wrap in
`rocprofv3 --kernel-trace --stats` to split the backward into its dQ and dK/dV kernels)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops

B, H, D, P, T = 32, 16, 64, 256, 32
if len(sys.argv) > 1:
    B, H, D, P, T = [int(v) for v in sys.argv[1].split(",")]
S, rot = P + T, D // 4
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((B * S, 3 * H * D), device=dev, generator=g).to(torch.bfloat16)
dout = torch.randn((B * S, H * D), device=dev, generator=g).to(torch.bfloat16)
am = torch.ones((B, T), dtype=torch.int64, device=dev)
for b in range(B):
    am[b, : (b * 5) % (T - 1)] = 0  # left padding
inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32, device=dev) / rot))
ang = torch.arange(S, dtype=torch.float32, device=dev)[:, None] * inv[None, :]
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()


def timed(fn, n=20, rounds=5):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


out, lse = ops.attn_fwd(qkv, B, S, H, D, rot, cos, sin, am)
t_f = timed(lambda: ops.attn_fwd(qkv, B, S, H, D, rot, cos, sin, am))
t_b = timed(lambda: ops.attn_bwd(qkv, out, dout, lse, B, S, H, D, rot, cos, sin, am))
fl = 4.0 * B * H * S * S * D
print(f"attn B={B} H={H} S={S} D={D}: fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.1f} TF dense-equiv)   bwd {t_b:7.1f} us ({2.5 * fl / t_b / 1e6:6.1f} TF)", flush=True)
# kernel execution times (start/stop events on each launch: not limited by the host's ~10 us per call)
from mafed_amd.profiler import KernelProfile
with KernelProfile() as kp:
    for _ in range(20):
        ops.attn_fwd(qkv, B, S, H, D, rot, cos, sin, am)
        ops.attn_bwd(qkv, out, dout, lse, B, S, H, D, rot, cos, sin, am)
for tag, a in kp.summary().items():
    print(f"   {tag:14s} {a['launches']:3d} launches  avg {a['avg_us']:7.2f} us", flush=True)
