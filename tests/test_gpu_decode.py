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


def _prerot_case(B, P, T, H, D, cap, t, dt, seed=0):
    from mafed_amd import ops
    g = torch.Generator().manual_seed(seed + B * 100 + t)
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
    am_full = torch.cat([am, torch.ones(B, t + 1, dtype=torch.int64, device=DEV)], dim=1).contiguous()
    want = _attn_decode_reference(qkv, B, S, H, D, rot, cos[:S].contiguous(), sin[:S].contiguous(), am_full, ops)
    # the cache as the decode loop leaves it before step t: prefix and rows < t rotated, row t as the QKV product wrote it
    rotated = qkv.clone().view(B * S, -1)
    ops.rotate_k_rows_(rotated, B, S, H, D, rot, cos, sin)
    rotated = rotated.view(B, S, -1)
    prefix = rotated[:, :S0, :].contiguous().view(B * S0, -1)
    new = torch.zeros(B, cap, 3 * H * D, dtype=dt, device=DEV)
    new[:, :t, :] = rotated[:, S0:S0 + t, :]
    new[:, t, :] = qkv[:, S0 + t, :]
    return ops, prefix, S0, new, rot, cos, sin, am, want, rotated


@pytest.mark.parametrize("flat", [0, 1])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,P,T,H,D,cap,t", [(2, 8, 6, 2, 64, 4, 0), (3, 40, 13, 2, 64, 5, 4), (2, 100, 20, 2, 128, 3, 2), (32, 256, 32, 16, 64, 10, 8),
                                             (2, 300, 40, 2, 64, 10, 9), (2, 500, 64, 1, 64, 4, 3), (1, 5, 3, 1, 256, 2, 1)])
def test_attn_decode_prerotated_cache(flat, dt, B, P, T, H, D, cap, t):
    """Pre-rotated K cache (round 4): the step equals the last row of the full attention forward, in the online form and in the
    all-rows-in-flight form (bf16, D in {64, 128}, up to 24 row groups; longer slices fall back to the online form), and leaves row t
    rotated in the cache."""
    from mafed_amd import _lib
    ops, prefix, S0, new, rot, cos, sin, am, want, rotated = _prerot_case(B, P, T, H, D, cap, t, dt)
    lib = _lib.load()
    assert lib.mafed_gemm_set_variant(740 + flat) == 0
    try:
        got = ops.attn_decode(prefix, S0, new, t, B, H, D, rot, cos, sin, am, prerot=True)
    finally:
        lib.mafed_gemm_set_variant(741)
    close(got.float(), want.float(), 1e-5 if dt == torch.float32 else 2e-2, "decode attention (pre-rotated cache) vs last row of the full forward")
    HD = H * D
    k_t = new[:, t, :].view(B, H, 3, D)[:, :, 1, :].float()
    k_want = rotated[:, S0 + t, :].view(B, H, 3, D)[:, :, 1, :].float()
    close(k_t, k_want, 1e-6 if dt == torch.float32 else 1e-2, "row t of the cache after the step: rotated key")
    assert torch.equal(new[:, t, :].view(B, H, 3, D)[:, :, 2, :], rotated[:, S0 + t, :].view(B, H, 3, D)[:, :, 2, :]) and HD > 0


def _decode_layer_operands(M, h, n1, seed=0):
    g = torch.Generator().manual_seed(seed + M + h)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    x = (r(M, h) * 1.5 + 0.3).to(DEV)
    ln = [(1.0 + 0.1 * r(h)).to(DEV), (0.1 * r(h)).to(DEV), (1.0 + 0.1 * r(h)).to(DEV), (0.1 * r(h)).to(DEV)]
    wqkv, bqkv = r(3 * h, h, scale=h ** -0.5).to(torch.bfloat16).to(DEV), (0.1 * r(3 * h)).to(DEV)
    w1, b1 = r(n1, h, scale=h ** -0.5).to(torch.bfloat16).to(DEV), (0.1 * r(n1)).to(DEV)
    wd, bd = r(h, h, scale=h ** -0.5).to(torch.bfloat16).to(DEV), (0.1 * r(h)).to(DEV)
    w2, b2 = r(h, n1, scale=n1 ** -0.5).to(torch.bfloat16).to(DEV), (0.1 * r(h)).to(DEV)
    return x, ln, wqkv, bqkv, w1, b1, wd, bd, w2, b2


DECODE_LAYER_SHAPES = [(32, 1024, 4096), (5, 768, 3072), (64, 1024, 4096), (17, 2048, 8192), (48, 768, 3072), (1, 1024, 4096), (16, 1024, 4096),
                       (19, 1024, 4096), (31, 1024, 2048)]


@pytest.fixture(params=[1, 0], ids=["lds", "direct"])
def decode_operand_path(request):
    """Both operand paths of the decode layer kernels: full-line loads through LDS (h = 1024, M <= 32; other shapes take the direct form
    either way) and MFMA fragments straight from global memory."""
    from mafed_amd import _lib
    lib = _lib.load()
    assert lib.mafed_gemm_set_variant(760 + request.param) == 0
    yield request.param
    lib.mafed_gemm_set_variant(761)


@pytest.mark.parametrize("M,h,n1", DECODE_LAYER_SHAPES)
def test_decode_ln_qkv_fc1_equals_the_separate_launches(M, h, n1, decode_operand_path):
    """csrc/decode.hip, launch A: both LayerNorms + QKV + fc1/GELU against mafed_layernorm_fwd + three mafed_gemm calls."""
    from mafed_amd import ops
    from mafed_amd._lib import EPI_GELU
    x, ln, wqkv, bqkv, w1, b1, *_ = _decode_layer_operands(M, h, n1)
    assert ops.decode_supported(M, h, n1)
    cap = 3
    cache = torch.zeros(M, cap, 3 * h, dtype=torch.bfloat16, device=DEV)
    a = ops.decode_ln_qkv_fc1(x, ln[0], ln[1], ln[2], ln[3], 1e-5, wqkv, bqkv, cache[:, 1, :], w1, b1)
    ln1, ln2, _, _ = ops.layernorm_fwd(x, ln[0], ln[1], ln[2], ln[3], 1e-5, torch.bfloat16, save_stats=False)
    qkv_want = ops.gemm(ln1, wqkv, False, True, bias=bqkv)
    a_want = ops.gemm(ln2, w1, False, True, bias=b1, epilogue=EPI_GELU)
    close(cache[:, 1, :].float(), qkv_want.float(), 1e-2, "qkv row")
    close(a.float(), a_want.float(), 1e-2, "gelu(fc1) row")
    assert float(cache[:, 0, :].abs().max()) == 0.0 and float(cache[:, 2, :].abs().max()) == 0.0   # neighbours of the cache row untouched
    # fp32 reference of the same op (torch): bf16-level agreement
    lnr = torch.nn.functional.layer_norm(x, (h,), ln[0], ln[1], 1e-5)
    close(cache[:, 1, :].float(), lnr @ wqkv.float().t() + bqkv, 2e-2, "qkv row vs fp32 torch")
    ln2r = torch.nn.functional.layer_norm(x, (h,), ln[2], ln[3], 1e-5)
    close(a.float(), torch.nn.functional.gelu(ln2r @ w1.float().t() + b1), 2e-2, "gelu(fc1) row vs fp32 torch")


@pytest.mark.parametrize("M,h,n1", DECODE_LAYER_SHAPES)
def test_decode_out_equals_the_separate_launches_and_is_deterministic(M, h, n1, decode_operand_path):
    """csrc/decode.hip, launch C: x + dense(ao) + fc2(a) as one product over the concatenated K, split over blocks and added in slice
    order: same bits on every run, in place or not, and the workspace's counters are back at zero after every launch."""
    from mafed_amd import ops
    x, ln, wqkv, bqkv, w1, b1, wd, bd, w2, b2 = _decode_layer_operands(M, h, n1, seed=7)
    g = torch.Generator().manual_seed(3)
    ao = torch.randn(M, h, generator=g).to(torch.bfloat16).to(DEV)
    act = torch.randn(M, n1, generator=g).to(torch.bfloat16).to(DEV)
    ws = ops.decode_out_workspace(M, h, DEV)
    outs = [ops.decode_out(x, ao, act, wd, bd, w2, b2, ws) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    x2 = x.clone()
    assert ops.decode_out(x2, ao, act, wd, bd, w2, b2, ws, out=x2) is x2 and torch.equal(x2, outs[0])
    n_cnt = h // 32
    assert int(ws[-4 * n_cnt:].view(torch.int32).abs().sum()) == 0
    want = x + (ao.float() @ wd.float().t() + bd) + (act.float() @ w2.float().t() + b2)
    close(outs[0], want, 2e-3, "x + dense + fc2 vs fp32 torch")
    attn = ops.gemm(ao, wd, False, True, bias=bd, out_dtype=torch.bfloat16)
    got_unfused = ops.gemm(act, w2, False, True, bias=b2, out_dtype=torch.float32, res1=attn, res2=x)
    close(outs[0], got_unfused, 1e-2, "vs the two separate launches (which round the attention branch to bf16)")


@pytest.mark.parametrize("case", ["t128"])
def test_generate_bf16_fused_decode_layers_equal_the_separate_launches(case):
    """The three-launch decode layer picks the same tokens as the six-launch one wherever the fp32 top-2 gap is above bf16 noise
    (h = 256 fixtures: the smallest width the fused kernels serve)."""
    from mafed_amd import ops
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup(case)
    model = build_model(cfg, sd, dtype=torch.bfloat16)
    b = to_dev(batch)
    assert ops.decode_supported(b["input_ids"].shape[0], cfg.hidden_size, cfg.intermediate_size)
    kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], patch_embeddings=b["patch_embeddings"].to(torch.bfloat16),
              max_new_tokens=max_new, eos_token_id=eos, return_step_logits=True, use_cache=True)
    model.fused_decode = True
    out_f, st_f = model.generate(**kw)
    model.fused_decode = False
    out_s, st_s = model.generate(**kw)
    T = b["input_ids"].shape[1]
    scale = float(step_logits.abs().max())
    for i in range(out_f.shape[1] - T):
        close(st_f[i], st_s[i], 3e-2, f"step {i}: fused vs separate launches")
        close(st_f[i], step_logits[i], 5e-2, f"step {i}: fused bf16 vs fp32 golden logits")
        if float(gaps[i].min()) < 0.05 * scale:
            break
        assert torch.equal(out_f[:, T + i], out_s[:, T + i]) and torch.equal(out_f[:, T + i].cpu(), tokens[:, T + i])


def test_generate_410m_fused_decode_tracks_the_separate_launches():
    """Full-width layers (VLPythia-410M, random weights, B = 8): step logits of the fused decode layers against the six-launch path,
    eager and replayed from the hipGraph."""
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig.preset("410m", num_vision_tokens=64)
    model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=11)
    g = torch.Generator().manual_seed(5)
    B, T = 8, 12
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).to(DEV)
    am = torch.ones(B, T, dtype=torch.int64)
    am[1, :3] = 0
    am[5, :7] = 0
    am = am.to(DEV)
    feats = torch.randn(B, 64, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).to(DEV)
    kw = dict(input_ids=ids, attention_mask=am, patch_embeddings=feats, max_new_tokens=4, eos_token_id=None, use_cache=True)
    model.fused_decode = True
    out_f, st_f = model.generate(return_step_logits=True, **kw)
    out_g = model.generate(use_graph=True, **kw)
    model.fused_decode = False
    out_s, st_s = model.generate(return_step_logits=True, **kw)
    assert torch.equal(out_f, out_g), "graph replay and eager launches of the fused layers pick the same tokens"
    # step 1 logits come from the first cached step on identical prefixes (step 0 is the prefill)
    close(st_f[0], st_s[0], 1e-6, "prefill logits")
    if torch.equal(out_f[:, T], out_s[:, T]):
        close(st_f[1], st_s[1], 3e-2, "first cached step: fused vs separate launches")


@pytest.mark.parametrize("M,h,N", [(32, 1024, 50304), (8, 1024, 50304), (40, 1024, 50304), (3, 768, 50304), (32, 1024, 256), (2, 256, 96), (16, 2048, 50304)])
def test_decode_ln_linear_equals_layernorm_then_head(M, h, N):
    """Final LayerNorm + LM head as one launch: the persistent strip kernel for a big vocabulary at h = 1024, M <= 32 (rows normalised once
    per CU, weight strips streamed through LDS), the one-slab-per-block forms else."""
    from mafed_amd import ops
    g = torch.Generator().manual_seed(M + h + N)
    x = (torch.randn(M, h, generator=g) * 2.0 - 0.5).to(DEV)
    lw, lb = (1.0 + 0.1 * torch.randn(h, generator=g)).to(DEV), (0.1 * torch.randn(h, generator=g)).to(DEV)
    w = (torch.randn(N, h, generator=g) * h ** -0.5).to(torch.bfloat16).to(DEV)
    got = ops.decode_ln_linear(x, lw, lb, 1e-5, w)
    lnf, _, _, _ = ops.layernorm_fwd(x, lw, lb, None, None, 1e-5, torch.bfloat16, save_stats=False)
    want = ops.gemm(lnf, w, False, True)
    close(got.float(), want.float(), 1e-2, "LN + head vs the two launches")
    close(got.float(), torch.nn.functional.layer_norm(x, (h,), lw, lb, 1e-5) @ w.float().t(), 2e-2, "LN + head vs fp32 torch")
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    got_b = ops.decode_ln_linear(x, lw, lb, 1e-5, w, bias=bias)
    close(got_b.float(), got.float() + bias, 1e-2, "bias operand")


# ---- the decode step as ONE launch (csrc/decode_flow.hip) ---------------------------------------------------------------------------
def _flow_model(layers=3, P=64, seed=21):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig(hidden_size=1024, num_hidden_layers=layers, num_attention_heads=16, intermediate_size=4096, num_vision_tokens=P)
    return cfg, VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=seed)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,T,P", [(32, 12, 64), (5, 9, 40), (17, 32, 256)])
def test_decode_flow_step_equals_the_three_launch_layers(B, T, P):
    """One launch per step against the three-launch layers on the same cache state: logits of three consecutive steps, the rows the steps
    append to every layer's K/V cache, no hand-over time-out, and the same bits when the step is repeated on a restored cache."""
    from mafed_amd.model import _DecodeCache
    cfg, model = _flow_model(P=P)
    g = torch.Generator().manual_seed(B)
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).to(DEV)
    am = torch.ones(B, T, dtype=torch.int64)
    if B > 1:
        am[1, : T // 3] = 0
        am[B - 1, : T // 2] = 0
    am = am.to(DEV)
    feats = torch.randn(B, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).to(DEV)
    NEW = 4

    def caches():
        st = model._engine_forward(feats, ids, am, None, False, train=False, keep_qkv=True)
        pre = [l["qkv"] for l in st["layers"]]
        model.flow_decode = True
        c_flow = _DecodeCache(model, [p.clone() for p in pre], B, st["S"], NEW, am)
        model.flow_decode = False
        c_ref = _DecodeCache(model, [p.clone() for p in pre], B, st["S"], NEW, am)
        model.flow_decode = True
        return c_flow, c_ref

    c_flow, c_ref = caches()
    assert c_flow.flow is not None and c_ref.flow is None and c_ref.fused
    tok = ids[:, -1].contiguous()
    for t in range(3):
        lf = model._engine_decode_step(tok, t, c_flow).float().clone()
        assert not c_flow.flow.timed_out(), f"step {t}: a hand-over timed out"
        lr = model._engine_decode_step(tok, t, c_ref).float()
        close(lf, lr, 2e-2, f"step {t}: logits, one launch vs three per layer")
        for i in range(cfg.num_hidden_layers):
            close(c_flow.new[i][:, t, :].float(), c_ref.new[i][:, t, :].float(), 2e-2, f"step {t}, layer {i}: appended cache row")
        assert torch.equal(lf.argmax(-1), lr.argmax(-1)) or float((lf.argmax(-1) == lr.argmax(-1)).float().mean()) > 0.9
        tok = lr.argmax(-1)
    # bit-identical from run to run (fixed summation orders, no float atomics): redo step 2 on both
    again = model._engine_decode_step(tok * 0 + ids[:, 0], 2, c_flow).float().clone()
    again2 = model._engine_decode_step(tok * 0 + ids[:, 0], 2, c_flow).float()
    assert torch.equal(again, again2)


@pytest.mark.timeout(300)
def test_generate_410m_one_launch_decode_tracks_the_three_launch_layers():
    """Full 24-layer stack, B = 8: generate() through the one-launch step (eager and replayed from the hipGraph) against the three-launch
    layers."""
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig.preset("410m", num_vision_tokens=64)
    model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device="cuda", seed=11)
    g = torch.Generator().manual_seed(5)
    B, T = 8, 12
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).to(DEV)
    am = torch.ones(B, T, dtype=torch.int64)
    am[1, :3] = 0
    am = am.to(DEV)
    feats = torch.randn(B, 64, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).to(DEV)
    kw = dict(input_ids=ids, attention_mask=am, patch_embeddings=feats, max_new_tokens=4, eos_token_id=None, use_cache=True)
    model.flow_decode = True
    out_f, st_f = model.generate(return_step_logits=True, **kw)
    out_g = model.generate(use_graph=True, **kw)
    model.flow_decode = False
    out_s, st_s = model.generate(return_step_logits=True, **kw)
    assert torch.equal(out_f, out_g), "graph replay and eager launches of the one-launch step pick the same tokens"
    close(st_f[0], st_s[0], 1e-6, "prefill logits")
    if torch.equal(out_f[:, T], out_s[:, T]):
        close(st_f[1], st_s[1], 3e-2, "first cached step: one launch vs three per layer")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,T,P", [(32, 12, 64), (5, 9, 40), (17, 32, 256)])
def test_decode_two_launch_layers_equal_the_three_launch_layers(B, T, P):
    """Attention + dense + fc2 + residuals as one grid behind the strip launch (mafed_decode_attn_out) against the three-launch layers on
    the same cache state: logits of three steps, appended cache rows, no time-out, same bits when a step is repeated."""
    from mafed_amd.model import _DecodeCache
    cfg, model = _flow_model(P=P)
    g = torch.Generator().manual_seed(B + 1)
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g).to(DEV)
    am = torch.ones(B, T, dtype=torch.int64)
    if B > 1:
        am[1, : T // 3] = 0
        am[B - 1, : T // 2] = 0
    am = am.to(DEV)
    feats = torch.randn(B, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16).to(DEV)
    st = model._engine_forward(feats, ids, am, None, False, train=False, keep_qkv=True)
    pre = [l["qkv"] for l in st["layers"]]
    model.flow_decode, model.pair_decode = False, True
    c_pair = _DecodeCache(model, [p.clone() for p in pre], B, st["S"], 4, am)
    model.pair_decode = False
    c_ref = _DecodeCache(model, [p.clone() for p in pre], B, st["S"], 4, am)
    model.pair_decode = True
    assert c_pair.pair is not None and c_ref.pair is None and c_ref.fused and c_ref.flow is None
    tok = ids[:, -1].contiguous()
    for t in range(3):
        lp = model._engine_decode_step(tok, t, c_pair).float().clone()
        assert not c_pair.pair.timed_out(), f"step {t}: a hand-over timed out"
        lr = model._engine_decode_step(tok, t, c_ref).float()
        close(lp, lr, 2e-2, f"step {t}: logits, two launches vs three per layer")
        for i in range(cfg.num_hidden_layers):
            close(c_pair.new[i][:, t, :].float(), c_ref.new[i][:, t, :].float(), 2e-2, f"step {t}, layer {i}: appended cache row")
        tok = lr.argmax(-1)
    again = model._engine_decode_step(ids[:, 0].contiguous(), 2, c_pair).float().clone()
    again2 = model._engine_decode_step(ids[:, 0].contiguous(), 2, c_pair).float()
    assert torch.equal(again, again2)
