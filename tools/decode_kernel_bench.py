"""Isolated timings of the decode-layer kernels at the headline shape (410M: h = 1024, 16 heads x 64, B = 32, 288 + t keys), each
replayed from a hipGraph of back-to-back launches (run on the GPU box).  Usage: decode_kernel_bench.py [B]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib
from mafed_amd._lib import EPI_GELU

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
h, n1, H, D, S0, cap, t = 1024, 4096, 16, 64, 288, 10, 5
dev = "cuda"
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)
NL = 24   # distinct weight sets, so that no launch finds its weights in the L2 / MALL of the previous one (24 x 29.5 MB = 0.7 GB)
x = r(B, h).to(dev)
ln = [(1 + 0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev), (1 + 0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev)]
W = [dict(wqkv=(r(3 * h, h) / 32).bfloat16().to(dev), w1=(r(n1, h) / 32).bfloat16().to(dev), wd=(r(h, h) / 32).bfloat16().to(dev),
          w2=(r(h, n1) / 64).bfloat16().to(dev)) for _ in range(NL)]
bqkv, b1, bd, b2 = (0.1 * r(3 * h)).to(dev), (0.1 * r(n1)).to(dev), (0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev)
prefix = [r(B * S0, 3 * h).bfloat16().to(dev) for _ in range(NL)]
new = [torch.zeros(B, cap, 3 * h, dtype=torch.bfloat16, device=dev) for _ in range(NL)]
am = torch.ones(B, 32, dtype=torch.int64, device=dev)
rot = D // 4
inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
ang = torch.arange(S0 + cap, dtype=torch.float32)[:, None] * inv[None, :]
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
ws = ops.decode_out_workspace(B, h, dev)
ao = r(B, h).bfloat16().to(dev)
act = r(B, n1).bfloat16().to(dev)
lib = _lib.load()


def timed(name, body, per=NL, bytes_per=None):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / per)
    extra = f"  {bytes_per / best / 1e6:.2f} TB/s" if bytes_per else ""
    print(f"{name:58s} {best:7.2f} us per launch{extra}", flush=True)
    return best


def A():
    for i in range(NL):
        ops.decode_ln_qkv_fc1(x, ln[0], ln[1], ln[2], ln[3], 1e-5, W[i]["wqkv"], bqkv, new[i][:, t, :], W[i]["w1"], b1)


def A6():
    for i in range(NL):
        l1, l2, _, _ = ops.layernorm_fwd(x, ln[0], ln[1], ln[2], ln[3], 1e-5, torch.bfloat16, save_stats=False)
        ops.gemm(l1, W[i]["wqkv"], False, True, bias=bqkv, out=new[i][:, t, :])
        ops.gemm(l2, W[i]["w1"], False, True, bias=b1, epilogue=EPI_GELU)


def Bk(prerot=True):
    for i in range(NL):
        ops.attn_decode(prefix[i], S0, new[i], t, B, H, D, rot, cos, sin, am, prerot=prerot)


def C():
    for i in range(NL):
        ops.decode_out(x, ao, act, W[i]["wd"], bd, W[i]["w2"], b2, ws)


def C6():
    for i in range(NL):
        attn = ops.gemm(ao, W[i]["wd"], False, True, bias=bd, out_dtype=torch.bfloat16)
        ops.gemm(act, W[i]["w2"], False, True, bias=b2, out_dtype=torch.float32, res1=attn, res2=x)


def layer():
    xx = x
    for i in range(NL):
        a = ops.decode_ln_qkv_fc1(xx, ln[0], ln[1], ln[2], ln[3], 1e-5, W[i]["wqkv"], bqkv, new[i][:, t, :], W[i]["w1"], b1)
        o = ops.attn_decode(prefix[i], S0, new[i], t, B, H, D, rot, cos, sin, am, prerot=True)
        xx = ops.decode_out(xx, o, a, W[i]["wd"], bd, W[i]["w2"], b2, ws)


print(f"B = {B}, h = {h}, {S0 + t + 1} keys; one launch per layer's weights, {NL} layers per replay")
wa, wc, kv = 2.0 * 7 * h * h, 2.0 * 5 * h * h, B * (S0 + t + 1) * 2 * h * 2.0
timed("A  decode_ln_qkv_fc1 (LN1 | LN2 + QKV + fc1 / GELU), through LDS", A, bytes_per=wa)
timed("   LayerNorm + QKV + fc1 as three launches", A6, bytes_per=wa)
timed("B  attention, all rows in flight", Bk, bytes_per=kv)
lib.mafed_gemm_set_variant(740)
timed("   attention, online softmax form", Bk, bytes_per=kv)
lib.mafed_gemm_set_variant(741)
timed("C  decode_out (dense + fc2 + residuals), operands through LDS", C, bytes_per=wc)
lib.mafed_gemm_set_variant(760)
timed("   decode_out, fragments straight from global memory", C, bytes_per=wc)
timed("   decode_ln_qkv_fc1, fragments straight from global memory", A, bytes_per=wa)
lib.mafed_gemm_set_variant(761)
timed("   dense, fc2 as two launches", C6, bytes_per=wc)
timed("A + B + C: one layer", layer, bytes_per=wa + wc + kv)
V = 50304
wout = [(r(V, h) / 32).bfloat16().to(dev) for _ in range(4)]


def head():
    for i in range(8):
        ops.decode_ln_linear(x, ln[0], ln[1], 1e-5, wout[i % 4])


def head2():
    for i in range(8):
        l1, _, _, _ = ops.layernorm_fwd(x, ln[0], ln[1], None, None, 1e-5, torch.bfloat16, save_stats=False)
        ops.gemm(l1, wout[i % 4], False, True)


timed("final LayerNorm + LM head, one launch", head, per=8, bytes_per=2.0 * V * h)
timed("   LayerNorm, head as two launches", head2, per=8, bytes_per=2.0 * V * h)
one = torch.zeros(64, device=dev)


def empty():
    for i in range(72):
        one.add_(1.0)


timed("(a 64-element add_: the floor of a launch in a graph chain)", empty, per=72)
