"""Two dependent-free products of one layer as ONE grouped persistent launch against two launches (410M shapes, rows = 32 x 288):
forward [dense | fc1] (dense is one round of 256 tiles on its own), backward [dao | dfc2] and [dfc1 | dqkv].  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops
from mafed_amd._lib import EPI_GELU, EPI_GELU_BWD

dev = "cuda"
rows, h, f = 32 * 288, 1024, 4096
g = torch.Generator().manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(torch.bfloat16).to(dev)
f32 = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(dev)
NL = 6   # distinct operand sets (cold-ish operands like in the step)
S = [dict(ao=bf(rows, h), ln2=bf(rows, h), ln1=bf(rows, h), Wd=bf(h, h), W1=bf(f, h), Wqkv=bf(3 * h, h), W2=bf(h, f), bd=f32(h), b1=f32(f), bq=f32(3 * h),
          dx=bf(rows, h), u=bf(rows, f), du=bf(rows, f), dqkv=bf(rows, 3 * h)) for _ in range(NL)]
out = dict(attn=torch.empty(rows, h, dtype=torch.bfloat16, device=dev), a=torch.empty(rows, f, dtype=torch.bfloat16, device=dev),
           uo=torch.empty(rows, f, dtype=torch.bfloat16, device=dev), qkv=torch.empty(rows, 3 * h, dtype=torch.bfloat16, device=dev),
           dao=torch.empty(rows, h, dtype=torch.bfloat16, device=dev), duo=torch.empty(rows, f, dtype=torch.bfloat16, device=dev),
           dl2=torch.empty(rows, h, dtype=torch.bfloat16, device=dev), dl1=torch.empty(rows, h, dtype=torch.bfloat16, device=dev))


def timed(name, body, flops):
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(4):
            body()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * NL))
    print(f"{name:62s} {best:7.1f} us  {flops / best / 1e6:7.1f} TFLOP/s", flush=True)
    return best


def fwd_sep():
    for s in S:
        ops.gemm(s["ao"], s["Wd"], False, True, bias=s["bd"], out=out["attn"])
        ops.gemm(s["ln2"], s["W1"], False, True, bias=s["b1"], epilogue=EPI_GELU, aux=out["uo"], out=out["a"])


def fwd_grp():
    for s in S:
        ops.gemm_grouped([dict(A=s["ao"], B=s["Wd"], out=out["attn"], bias=s["bd"]),
                          dict(A=s["ln2"], B=s["W1"], out=out["a"], bias=s["b1"], epilogue=EPI_GELU, aux=out["uo"])], False, True)


def fwd_qkv_fc1_sep():
    for s in S:
        ops.gemm(s["ln1"], s["Wqkv"], False, True, bias=s["bq"], out=out["qkv"])
        ops.gemm(s["ln2"], s["W1"], False, True, bias=s["b1"], epilogue=EPI_GELU, aux=out["uo"], out=out["a"])


def fwd_qkv_fc1_grp():
    for s in S:
        ops.gemm_grouped([dict(A=s["ln1"], B=s["Wqkv"], out=out["qkv"], bias=s["bq"]),
                          dict(A=s["ln2"], B=s["W1"], out=out["a"], bias=s["b1"], epilogue=EPI_GELU, aux=out["uo"])], False, True)


def bwd1_sep():
    for s in S:
        ops.gemm(s["dx"], s["Wd"], False, False, out=out["dao"])
        ops.gemm(s["dx"], s["W2"], False, False, epilogue=EPI_GELU_BWD, aux=s["u"], out=out["duo"])


def bwd1_grp():
    for s in S:
        ops.gemm_grouped([dict(A=s["dx"], B=s["Wd"], out=out["dao"]),
                          dict(A=s["dx"], B=s["W2"], out=out["duo"], epilogue=EPI_GELU_BWD, aux=s["u"])], False, False)


def bwd2_sep():
    for s in S:
        ops.gemm(s["du"], s["W1"], False, False, out=out["dl2"])
        ops.gemm(s["dqkv"], s["Wqkv"], False, False, out=out["dl1"])


def bwd2_grp():
    for s in S:
        ops.gemm_grouped([dict(A=s["du"], B=s["W1"], out=out["dl2"]), dict(A=s["dqkv"], B=s["Wqkv"], out=out["dl1"])], False, False)


F = lambda n, k: 2.0 * rows * n * k
timed("forward  dense, fc1 (GELU + pre-activation): two launches", fwd_sep, F(h, h) + F(f, h))
timed("forward  [dense | fc1] one grouped launch", fwd_grp, F(h, h) + F(f, h))
timed("forward  qkv, fc1: two launches", fwd_qkv_fc1_sep, F(3 * h, h) + F(f, h))
timed("forward  [qkv | fc1] one grouped launch", fwd_qkv_fc1_grp, F(3 * h, h) + F(f, h))
timed("backward dao, dfc2 (GELU'): two launches", bwd1_sep, F(h, h) + F(f, h))
timed("backward [dao | dfc2] one grouped launch", bwd1_grp, F(h, h) + F(f, h))
timed("backward dfc1, dqkv: two launches", bwd2_sep, F(h, f) + F(h, 3 * h))
timed("backward [dfc1 | dqkv] one grouped launch", bwd2_grp, F(h, f) + F(h, 3 * h))
from mafed_amd import _lib
print("persistent launches so far:", _lib.load().mafed_gemm_pp_launches())
