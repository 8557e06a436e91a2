"""Where the one-launch decode step (csrc/decode_flow.hip) spends a layer: per-workgroup wall-clock stamps {dispatched, wait over, done},
summarised per role for a few layers in the middle of the stack (410M, B = 32, 256 + 32 tokens).  Run on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM, _lib
from mafed_amd.model import _DecodeCache

B, P, T, NEW = 32, 256, 32, 10
cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=1234)
model.flow_decode = True
g = torch.Generator().manual_seed(0)
ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).cuda()
am = torch.ones(B, T, dtype=torch.int64).cuda()
feats = torch.randn(B, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).cuda()
st = model._engine_forward(feats, ids, am, None, False, train=False, keep_qkv=True)
cache = _DecodeCache(model, [l["qkv"] for l in st["layers"]], B, st["S"], NEW, am)
assert cache.flow is not None
tok = ids[:, -1].contiguous()
for t in range(3):
    model._engine_decode_step(tok, t, cache)
torch.cuda.synchronize()
lib = _lib.load()
L, h, n1, H, V = cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.vocab_size
grid = int(lib.mafed_decode_flow_grid(L, B, h, n1, H, V))
buf = torch.zeros(grid, 4, dtype=torch.int64, device="cuda")
lib.mafed_decode_flow_set_trace(buf.data_ptr())
model._engine_decode_step(tok, 3, cache)
torch.cuda.synchronize()
lib.mafed_decode_flow_set_trace(0)
assert not cache.flow.timed_out()
tr = buf.cpu().numpy().astype(np.float64) / 100.0   # 100 MHz wall clock -> us
t00 = tr[:, 0].min()
tr[:, :3] -= t00
nL, nAq, nAf, nB = 8, 3 * h // 16, n1 // 16, B * H
nCa, nCo = (h // 32) * (n1 // 512), (h // 32) * (h // 512)
per = nL + nAq + nAf + nB + nCa + nCo
roles = [("LN rows", nL), ("qkv strips", nAq), ("fc1 strips", nAf), ("attention", nB), ("fc2 K-slices", nCa), ("dense K-slices", nCo)]
print(f"grid {grid} workgroups, {per} per layer; step {tr[:, 2].max():.1f} us from first dispatch to last done")
for layer in (1, 11, 12, 22):
    base = layer * per
    t_first = tr[base:base + per, 0].min()
    print(f"layer {layer}: first dispatch at {t_first:8.1f} us; times below relative to it (min / median / max)")
    o = base
    for name, n in roles:
        blk = tr[o:o + n]
        f = lambda c: f"{np.min(blk[:, c]) - t_first:7.1f} {np.median(blk[:, c]) - t_first:7.1f} {np.max(blk[:, c]) - t_first:7.1f}"
        own = blk[:, 2] - blk[:, 1]
        print(f"   {name:15s} n {n:4d} | dispatched {f(0)} | wait over {f(1)} | done {f(2)} | work after the wait {np.median(own):5.2f} (max {own.max():5.2f})")
        o += n
lay_done = [tr[l * per:(l + 1) * per, 2].max() for l in range(L)]
print("layer completion intervals (us):", " ".join(f"{lay_done[l] - lay_done[l - 1]:.1f}" for l in range(1, L)))
hb = tr[L * per:]
print(f"head: dispatched {hb[:, 0].min():.1f} .. {hb[:, 0].max():.1f}, done {hb[:, 2].max():.1f}")
