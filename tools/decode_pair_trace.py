"""Per-workgroup stamps of ONE launch of decode_attn_out_kernel (layer 12 of a 410M step, B = 32): fc2 K-slices, attention slices, dense
K-slices.  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM, _lib, ops
from mafed_amd.model import _DecodeCache

B, P, T, NEW = 32, 256, 32, 10
cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=1234)
model.pair_decode = True
g = torch.Generator().manual_seed(0)
ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).cuda()
am = torch.ones(B, T, dtype=torch.int64).cuda()
feats = torch.randn(B, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).cuda()
st = model._engine_forward(feats, ids, am, None, False, train=False, keep_qkv=True)
cache = _DecodeCache(model, [l["qkv"] for l in st["layers"]], B, st["S"], NEW, am)
pr = cache.pair
assert pr is not None
tok = ids[:, -1].contiguous()
for t in range(3):
    model._engine_decode_step(tok, t, cache)
torch.cuda.synchronize()
lib = _lib.load()
h, n1, H = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads
nCa, nB, nCo = (h // 32) * (n1 // 512), B * H, (h // 32) * (h // 512)
grid = nCa + nB + nCo
buf = torch.zeros(grid, 4, dtype=torch.int64, device="cuda")
x = torch.randn(B, h, device="cuda")
cos, sin = model.rotary_tables(st["S"] + cache.cap)
pr.begin_step(3)
# flush the caches with an unrelated pass so that the layer's operands come from HBM as in the step
junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda").fill_(1.0)
torch.cuda.synchronize()
lib.mafed_decode_flow_set_trace(buf.data_ptr())
pr.run(12, 3, x, st["S"], cfg.rotary_ndims, P, cos, sin, am)
torch.cuda.synchronize()
lib.mafed_decode_flow_set_trace(0)
tr = buf.cpu().numpy().astype(np.float64) / 100.0
tr[:, :3] -= tr[:, 0].min()
print(f"grid {grid}; launch spans {tr[:, 2].max():.1f} us from first dispatch to last done")
o = 0
for name, n in (("fc2 K-slices", nCa), ("attention", nB), ("dense K-slices", nCo)):
    blk = tr[o:o + n]
    f = lambda c: f"{np.min(blk[:, c]):7.1f} {np.median(blk[:, c]):7.1f} {np.max(blk[:, c]):7.1f}"
    wait = blk[:, 1] > 0
    print(f"   {name:15s} n {n:4d} | dispatched {f(0)} | done {f(2)} | duration median {np.median(blk[:, 2] - blk[:, 0]):6.2f} max {np.max(blk[:, 2] - blk[:, 0]):6.2f}"
          + (f" | wait over {np.min(blk[wait, 1]):7.1f} {np.median(blk[wait, 1]):7.1f} {np.max(blk[wait, 1]):7.1f}" if wait.any() else ""))
    o += n
