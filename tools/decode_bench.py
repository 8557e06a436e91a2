"""Validation decode at the headline shape (VLPythia-410M, B = 32, 256 image + 32 text tokens, 10 new tokens, bf16):
the reference's use_cache=False recompute against the KV-cached path (run on the GPU box)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM

B, P, T, NEW = 32, 256, 32, 10
model_name = sys.argv[1] if len(sys.argv) > 1 else "410m"
cfg = VLPythiaConfig.preset(model_name, num_vision_tokens=P)
model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=1234)
g = torch.Generator().manual_seed(0)
ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).cuda()
am = torch.ones(B, T, dtype=torch.int64).cuda()
feats = torch.randn(B, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).cuda()


def run(use_cache, reps=5, use_graph=False):
    kw = dict(input_ids=ids, attention_mask=am, patch_embeddings=feats, max_new_tokens=NEW, use_cache=use_cache, eos_token_id=None,
              use_graph=use_graph)
    out = model.generate(**kw)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = model.generate(**kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


t_u, o_u = run(False)
legs = {}
for name, fused, flow, pair in (("six launches per layer", False, False, False),
                                ("one launch per step (opt-in)", True, True, False), ("two launches per layer (opt-in)", True, False, True),
                                ("three launches per layer", True, False, False)):
    model._decode_graphs = {}
    model.fused_decode, model.flow_decode, model.pair_decode = fused, flow, pair
    t_eager, o_eager = run(True)
    t_graph, o_graph = run(True, use_graph=True)
    assert torch.equal(o_eager, o_graph), f"{name}: graph replay and eager launches must produce the same tokens"
    legs[name] = (t_eager, t_graph, o_graph)
t_e, t_c, o_c = legs["three launches per layer"]
same = float((o_u == o_c).float().mean())
n_params = sum(p.numel() for p in model.parameters())
wbytes = 2.0 * (n_params - cfg.vocab_size * cfg.hidden_size)  # bf16 weights streamed per decode step (all but embed_in)
print(f"{model_name}: generate B={B} {P}+{T} tokens, {NEW} new: recompute {t_u * 1e3:.1f} ms ({B / t_u:.0f} ex/s), "
      f"KV-cached eager {t_e * 1e3:.1f} ms, KV-cached + hipGraph {t_c * 1e3:.1f} ms ({B / t_c:.0f} ex/s), speed-up {t_u / t_c:.2f}x, "
      f"tokens equal {same:.3f}")
for name, (te, tg, og) in legs.items():
    print(f"   {name}: eager {te * 1e3:.1f} ms, hipGraph {tg * 1e3:.1f} ms, tokens equal to the default path {float((og == o_c).float().mean()):.3f}")
# decode-step roofline: the (NEW - 1) cached steps stream the weights once each and every layer's K/V slice once
from mafed_amd.model import _DecodeCache
kv_bytes = 0.0


def step_ms(fused: bool):
    """One cached step, timed inside a hipGraph replay of NEW - 1 steps (eager launches of 75 - 150 small kernels are host-bound)."""
    global kv_bytes
    st = model._engine_forward(feats, ids, am, None, False, train=False, keep_qkv=True)
    cache = _DecodeCache(model, [l["qkv"] for l in st["layers"]], B, st["S"], NEW, am, fused=fused)
    kv_bytes = cfg.num_hidden_layers * B * (st["S"] + NEW / 2) * 2 * cfg.hidden_size * 2.0
    tok = ids[:, -1].contiguous()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for t in range(NEW - 1):
            model._engine_decode_step(tok, t, cache)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for t in range(NEW - 1):
            out = model._engine_decode_step(tok, t, cache)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (NEW - 1))
    return best, cache.fused


for name, fused, flow, pair in (("three launches per layer", True, False, False), ("two launches per layer, opt-in", True, False, True),
                                ("one launch per step, opt-in", True, True, False), ("six launches per layer", False, False, False)):
    model.fused_decode, model.flow_decode, model.pair_decode = fused, flow, pair
    ms, was_fused = step_ms(fused)
    tot = wbytes + kv_bytes
    print(f"decode step ({name}): {ms:.3f} ms; weights {wbytes / 1e9:.2f} GB + K/V {kv_bytes / 1e9:.2f} GB "
          f"per step -> {tot / ms / 1e9:.2f} TB/s ({tot / ms / 1e9 / 8.0 * 100:.1f} % of 8 TB/s; weights alone {wbytes / ms / 1e9 / 8.0 * 100:.1f} %)")
model.fused_decode, model.flow_decode, model.pair_decode = True, False, False
