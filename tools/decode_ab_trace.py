"""Stamps of one launch each of the decode layer's first two kernels (410M: h = 1024, 16 heads x 64, B = 32, 288 + 5 keys), operands
cold in the L2 (seven other layers' operands touched in between).  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mafed_amd import ops, _lib

dev = "cuda"
B, h, n1, H, D, S0, cap, t = 32, 1024, 4096, 16, 64, 288, 10, 5
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)
NL = 8
x = r(B, h).to(dev)
ln = [(1 + 0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev), (1 + 0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev)]
W = [dict(wqkv=(r(3 * h, h) / 32).bfloat16().to(dev), w1=(r(n1, h) / 32).bfloat16().to(dev)) for _ in range(NL)]
bqkv, b1 = (0.1 * r(3 * h)).to(dev), (0.1 * r(n1)).to(dev)
prefix = [r(B * S0, 3 * h).bfloat16().to(dev) for _ in range(NL)]
new = [torch.zeros(B, cap, 3 * h, dtype=torch.bfloat16, device=dev) for _ in range(NL)]
am = torch.ones(B, 32, dtype=torch.int64, device=dev)
rot = D // 4
inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
ang = torch.arange(S0 + cap, dtype=torch.float32)[:, None] * inv[None, :]
cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
lib = _lib.load()


def A(i):
    ops.decode_ln_qkv_fc1(x, ln[0], ln[1], ln[2], ln[3], 1e-5, W[i]["wqkv"], bqkv, new[i][:, t, :], W[i]["w1"], b1)


def Bk(i):
    ops.attn_decode(prefix[i], S0, new[i], t, B, H, D, rot, cos, sin, am, prerot=True)


def show(title, tr, names):
    tr = tr.astype(np.float64) / 100.0
    t0 = tr[:, 0].min()
    print(title)
    for k, nm in enumerate(names):
        col = tr[:, k] - t0
        print(f"   {nm:28s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")


for i in range(NL):
    A(i); Bk(i)
torch.cuda.synchronize()
nA = (3 * h + n1) // 32
buf = torch.zeros(nA, 8, dtype=torch.int64, device=dev)
for i in range(1, NL):
    A(i)
torch.cuda.synchronize()
lib.mafed_decode_set_trace(buf.data_ptr())
A(0)
torch.cuda.synchronize()
lib.mafed_decode_set_trace(0)
show(f"A  decode_ln_qkv_fc1, {nA} workgroups", buf.cpu().numpy(),
     ["entered", "x rows in registers", "row statistics done", "normalised rows in LDS", "weight slab landed", "barrier passed", "MFMAs issued", "left (stores out)"])
buf2 = torch.zeros(B * H, 8, dtype=torch.int64, device=dev)
for i in range(1, NL):
    Bk(i)
torch.cuda.synchronize()
lib.mafed_attn_decode_set_trace(buf2.data_ptr())
Bk(0)
torch.cuda.synchronize()
lib.mafed_attn_decode_set_trace(0)
show(f"B  attention (all rows in flight), {B * H} workgroups", buf2.cpu().numpy()[:, :5],
     ["entered", "q | k row rotated", "scores done (K rows in)", "V sum + shuffles done", "left (store out)"])
