"""Stamps of one mafed_decode_out launch (410M: h = 1024, n1 = 4096, M = 32; 32 column groups x 10 K-slices).  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mafed_amd import ops, _lib

dev = "cuda"
M, h, n1 = 32, 1024, 4096
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)
x = r(M, h).to(dev)
W = [dict(wd=(r(h, h) / 32).bfloat16().to(dev), w2=(r(h, n1) / 64).bfloat16().to(dev)) for _ in range(8)]
bd, b2 = (0.1 * r(h)).to(dev), (0.1 * r(h)).to(dev)
ao, act = r(M, h).bfloat16().to(dev), r(M, n1).bfloat16().to(dev)
ws = ops.decode_out_workspace(M, h, dev)
lib = _lib.load()
for w in W:
    ops.decode_out(x, ao, act, w["wd"], bd, w["w2"], b2, ws)
torch.cuda.synchronize()
buf = torch.zeros(32 * 10, 8, dtype=torch.int64, device=dev)
lib.mafed_decode_set_trace(buf.data_ptr())
ops.decode_out(x, ao, act, W[0]["wd"], bd, W[0]["w2"], b2, ws)
torch.cuda.synchronize()
lib.mafed_decode_set_trace(0)
tr = buf.cpu().numpy().astype(np.float64) / 100.0
t0 = tr[:, 0].min()
names = ["entered", "operands in LDS", "K loop done", "partial tile out", "counter bumped", "left"]
for k, nm in enumerate(names):
    col = tr[:, k] - t0
    print(f"{nm:18s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
last = tr[:, 5] - tr[:, 4]
print(f"after the counter: median {np.median(last):.2f}, the 32 reducing workgroups {np.sort(last)[-32:].mean():.2f} us")
