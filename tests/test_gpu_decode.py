"""GPU parity of the KV-cached greedy decode: ``model.generate`` with and without the cache against
the golden tokens / last-position logits produced by the reference model's own forward inside the greedy-search loop."""
import pytest
import torch

from tests.helpers import DECODE_CASES, decode_setup
from tests.test_gpu_model import DEV, build_model, close, to_dev

pytestmark = pytest.mark.gpu


def _attn_decode_reference(qkv_all, B, S, H, D, rot, cos, sin, am, ops):
    """Last row of the full (training) attention forward over S keys = what the decode kernel must produce for position S-1."""
    out, _ = ops.attn_fwd(qkv_all.reshape(B * S, -1), B, S, H, D, rot, cos, sin, am)
    return out.view(B, S, H * D)[:, -1, :]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,P,T,H,D,cap,t", [(2, 8, 6, 2, 64, 4, 0), (3, 40, 13, 2, 64, 5, 4), (2, 8, 6, 1, 128, 3, 2), (1, 5, 3, 1, 256, 2, 1)])
def test_attn_decode_kernel_equals_last_row_of_full_attention(dt, B, P, T, H, D, cap, t):
    from mafed_amd import ops
    g = torch.Generator().manual_seed(B * 100 + t)
    S0 = P + T
    S = S0 + t + 1
    qkv = torch.randn(B, S, 3 * H * D, generator=g).to(dt).to(DEV)
    am = torch.ones(B, T, dtype=torch.int64)
    for b in range(1, B):
        am[b, : (2 * b) % T] = 0
    am = am.to(DEV)
    rot = D // 4
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot, 2, dtype=torch.float32) / rot))
    ang = torch.arange(S0 + cap, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    # full attention over prompt + generated tokens: the mask of the generated keys is 1
    am_full = torch.cat([am, torch.ones(B, t + 1, dtype=torch.int64, device=DEV)], dim=1).contiguous()
    want = _attn_decode_reference(qkv, B, S, H, D, rot, cos[:S].contiguous(), sin[:S].contiguous(), am_full, ops)
    prefix = qkv[:, :S0, :].contiguous().view(B * S0, -1)
    new = torch.zeros(B, cap, 3 * H * D, dtype=dt, device=DEV)
    new[:, : t + 1, :] = qkv[:, S0:, :]
    got = ops.attn_decode(prefix, S0, new, t, B, H, D, rot, cos, sin, am)
    close(got.float(), want.float(), 1e-5 if dt == torch.float32 else 2e-2, "decode attention vs last row of the full forward")


@pytest.mark.parametrize("use_cache", [True, False])
@pytest.mark.parametrize("case", list(DECODE_CASES))
def test_generate_fp32_matches_reference_golden(case, use_cache):
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup(case)
    model = build_model(cfg, sd)
    b = to_dev(batch)
    out, steps = model.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], patch_embeddings=b["patch_embeddings"],
                                max_new_tokens=max_new, use_cache=use_cache, eos_token_id=eos, pad_token_id=eos, return_step_logits=True)
    assert out.shape == tokens.shape, (out.shape, tokens.shape)
    assert torch.equal(out.cpu(), tokens), (out.cpu(), tokens)
    close(steps, step_logits, 1e-3, "last-position logits of every step")


@pytest.mark.parametrize("case", ["t64", "m64", "t128"])
def test_generate_bf16_cached_equals_uncached_and_tracks_fp32(case):
    """bf16 (MFMA) mode: the cached and the recompute-everything paths pick the same tokens wherever the fp32 top-2 gap is
    far above bf16 noise, and their logits agree at bf16 level with the fp32 golden ones."""
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup(case)
    model = build_model(cfg, sd, dtype=torch.bfloat16)
    b = to_dev(batch)
    kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], patch_embeddings=b["patch_embeddings"].to(torch.bfloat16),
              max_new_tokens=max_new, eos_token_id=eos, return_step_logits=True)
    out_c, st_c = model.generate(use_cache=True, **kw)
    out_u, st_u = model.generate(use_cache=False, **kw)
    T = b["input_ids"].shape[1]
    scale = float(step_logits.abs().max())
    # compare step by step while the two runs (and the golden) still share the prefix
    for i in range(out_c.shape[1] - T):
        close(st_c[i], st_u[i], 3e-2, f"step {i}: cached vs recomputed logits")
        close(st_c[i], step_logits[i], 5e-2, f"step {i}: bf16 vs fp32 golden logits")
        if float(gaps[i].min()) < 0.05 * scale:
            break  # a near-tie may legitimately flip under bf16: later steps see different prefixes
        assert torch.equal(out_c[:, T + i], out_u[:, T + i]) and torch.equal(out_c[:, T + i].cpu(), tokens[:, T + i])


@pytest.mark.parametrize("case", ["t64", "t64_eos", "t128"])
def test_generate_graph_replay_equals_eager(case):
    """use_graph=True: the decode steps replayed from one hipGraph (static K/V cache written by the prefill) give the golden
    tokens, twice in a row (second call = pure replay with new inputs)."""
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup(case)
    model = build_model(cfg, sd)
    b = to_dev(batch)
    kw = dict(attention_mask=b["attention_mask"], patch_embeddings=b["patch_embeddings"], max_new_tokens=max_new, eos_token_id=eos,
              pad_token_id=eos, use_cache=True, use_graph=True)
    out1 = model.generate(input_ids=b["input_ids"], **kw)
    assert torch.equal(out1.cpu(), tokens)
    # same shapes, different prompt: rows rolled by one -> outputs roll with them
    rolled = {k: torch.roll(v, 1, dims=0) for k, v in b.items()}
    out2 = model.generate(input_ids=rolled["input_ids"], **{**kw, "attention_mask": rolled["attention_mask"], "patch_embeddings": rolled["patch_embeddings"]})
    if eos is None:
        assert torch.equal(out2.cpu(), torch.roll(tokens, 1, dims=0))
    assert len(model._decode_graphs) == 1


def test_generate_call_signature_of_the_reference_validation_step():
    """mafed/model/vqa_cont_learner.py:260-267: pixel_values features + use_cache=False + pad_token_id=eos."""
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup("t64")
    model = build_model(cfg, sd)
    b = to_dev(batch)
    pv = torch.cat([torch.zeros(b["patch_embeddings"].shape[0], 1, cfg.vision_hidden_size, device=DEV), b["patch_embeddings"]], dim=1)
    out = model.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], pixel_values=pv, max_new_tokens=10,
                         use_cache=False, pad_token_id=0, eos_token_id=None)
    assert torch.equal(out.cpu(), tokens)
    with pytest.raises(NotImplementedError):
        model.generate(input_ids=b["input_ids"], pixel_values=pv, do_sample=True)
