"""Isolated timing of the fused dual-LayerNorm backward at the 410M MAFED step shape (run on the GPU box)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops

rows, h, B, S, P = 9216, 1024, 32, 288, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
x, dres, teacher = rn(rows, h), rn(rows, h), rn(rows, h)
dy1, dy2 = rn(rows, h).to(torch.bfloat16), rn(rows, h).to(torch.bfloat16)
w1, w2 = rn(h), rn(h)
mean, rstd = x.mean(1), 1.0 / x.std(1)
am = torch.ones(B, S - P, dtype=torch.int64, device=dev)
scale = torch.tensor([0.1, 0.2, 0.0, 0.0], device=dev)
grads = [torch.zeros(h, device=dev) for _ in range(6)]


def run(inject):
    kw = dict(teacher=teacher, attention_mask=am, S=S, P=P, inj_scale=scale, inj_mul=2.0 / h) if inject else {}
    return ops.layernorm_bwd(dy1, dy2, x, mean, rstd, w1, w2, dres, grads[0], grads[1], grads[2], grads[3], want_lp=True,
                             dxsum_a=grads[4], dxsum_b=grads[5], **kw)


for inject in (False, True):
    run(inject)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(inject)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    mb = rows * h * (4 + 2 + 2 + 4 + 4 + 2 + (4 if inject else 0)) / 1e6
    print(f"layernorm_bwd dual{' + injection' if inject else ''}: {best * 1e3:6.1f} us  ({mb / best / 1e3:5.2f} TB/s algorithmic)", flush=True)


b1, b2 = rn(h), rn(h)
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.layernorm_fwd(x, w1, b1, w2, b2, 1e-5, torch.bfloat16, save_stats=True)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"layernorm_fwd dual: {best * 1e3:6.1f} us  ({rows * h * (4 + 2 + 2) / 1e6 / best / 1e3:5.2f} TB/s algorithmic)", flush=True)
