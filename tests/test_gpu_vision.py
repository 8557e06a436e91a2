"""The native frozen CLIP vision tower through the HIP kernels, against fixtures produced by
``transformers.CLIPVisionModel`` through the reference's own ``get_patch_embeddings`` / ``forward`` (oracle/gen_golden.py::
gen_clip_fixture), and against the oracle restatement at the real CLIP-L/14 geometry."""
import types

import numpy as np
import pytest
import torch

from oracle import clip_vit_ref as C
from oracle import vlpythia_ref as R
from tests.helpers import clip_setup

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, tol, what=""):
    a = np.asarray(a.detach().cpu().double() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu().double() if isinstance(b, torch.Tensor) else b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.1e}*{scale:.3g}"


def build_tower(cc, csd, dtype):
    from mafed_amd.vision import ClipVisionConfig, ClipVisionTower
    cfg = ClipVisionConfig(hidden_size=cc.hidden_size, num_hidden_layers=cc.num_hidden_layers, num_attention_heads=cc.num_attention_heads,
                           intermediate_size=cc.intermediate_size, image_size=cc.image_size, patch_size=cc.patch_size)
    t = ClipVisionTower(cfg, compute_dtype=dtype, device=DEV)
    t.load_state_dict(csd, strict=True)
    assert all(not p.requires_grad for p in t.parameters())
    return t


def build_lm(cfg, sd, dtype, tower):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    mc = VLPythiaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        vision_hidden_size=cfg.vision_hidden_size, num_vision_tokens=cfg.num_vision_tokens)
    m = VLPythiaForCausalLM(mc, compute_dtype=dtype, device=DEV, vision_encoder=tower)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("vision_encoder.vision_model.") for k in missing), (missing, unexpected)
    return m


@pytest.mark.parametrize("name", ["c17", "c50"])
def test_tower_matches_reference_fixture_fp32(name):
    cc, csd, pixels, cfg, sd, batch, g = clip_setup(name)
    tower = build_tower(cc, csd, torch.float32)
    hs = tower(pixels.to(DEV))
    assert hs.shape == (pixels.shape[0], cc.num_patches + 1, cc.hidden_size) and hs.dtype == torch.float32
    close(hs[:, 1:], g["features"], 1e-3, "hidden_states[-2][:, 1:]")
    close(hs[:, 0], g["penultimate_cls"], 1e-3, "class-token row")
    close(tower.patch_features(pixels.to(DEV)), g["features"], 1e-3, "patch_features")
    # unprefixed names (transformers 5.x registration) load into the same tensors
    t2 = build_tower(cc, {k[len("vision_model."):]: v for k, v in csd.items()}, torch.float32)
    assert torch.equal(t2(pixels.to(DEV)), hs)
    with pytest.raises(ValueError):
        tower(torch.zeros(1, 3, cc.image_size + 14, cc.image_size, device=DEV))


@pytest.mark.parametrize("name", ["c17", "c50"])
def test_model_accepts_pixel_values_end_to_end(name):
    """model(**batch) with pixel_values [B,3,H,W]: tower -> feature_select -> projector -> decoder -> loss, vs the reference."""
    cc, csd, pixels, cfg, sd, batch, g = clip_setup(name)
    model = build_lm(cfg, sd, torch.float32, build_tower(cc, csd, torch.float32))
    T = batch["input_ids"].shape[1]
    b = {k: v.to(DEV) for k, v in batch.items() if k != "patch_embeddings"}
    out = model(**b, pixel_values=pixels.to(DEV), output_hidden_states=True, return_dict=True)
    close(out.loss, float(g["loss"]), 1e-3, "loss")
    close(out.logits, g["logits_text"], 1e-3, "logits (text positions)")
    close(out.hidden_states[0], g["lm_hidden0"], 1e-3, "LM hidden 0")
    # state-dict contract: the tower's tensors sit under vision_encoder.vision_model.* like the reference's CLIPVisionModel
    names = [k for k in model.state_dict() if k.startswith("vision_encoder.")]
    assert len(names) == len(C.param_shapes(cc)) and names[0] == "vision_encoder.vision_model.embeddings.class_embedding"
    assert all(not p.requires_grad for p in model.vision_encoder.parameters())   # what vqa_cont_learner.py:202-203 freezes
    out.loss.backward()   # the trainable part still trains
    assert float(model.flat_grads.abs().sum()) > 0


def test_bf16_tower_tracks_fp32_and_replay_encodes_once():
    from mafed_amd import FeatureDistillation
    cc, csd, pixels, cfg, sd, batch, g = clip_setup("c50")
    t16 = build_tower(cc, csd, torch.bfloat16)
    f16 = t16.patch_features(pixels.to(DEV))
    close(f16, g["features"], 4e-2, "bf16 tower vs reference features")
    # MAFED replay on an IMAGE memory batch: student and teacher share one pass through the frozen tower
    model = build_lm(cfg, sd, torch.float32, build_tower(cc, csd, torch.float32))
    calls = []
    orig = model.vision_encoder.hidden_state
    model.vision_encoder.hidden_state = lambda pv: (calls.append(1), orig(pv))[1]
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=pixels.shape[0], seed=1, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None)
    fd._update_model(model)
    assert fd.past_model.vision_encoder is model.vision_encoder      # the frozen tower is shared, not copied
    fd.past_model.flat_params.mul_(1.01)
    fd.past_model._shadow_dirty = True
    fd.task_id = 1
    fd.num_vision_tokens = cc.num_patches
    mem = {k: v.to(DEV) for k, v in batch.items() if k != "patch_embeddings"}
    mem["pixel_values"] = pixels.to(DEV)
    fd.mem_dataloader = [mem]
    loss, n = fd.replay(model)
    loss.backward()
    assert len(calls) == 1, f"the tower ran {len(calls)} times for one replay step"
    # same numbers as feeding the oracle's features to both models
    feats = C.patch_features(csd, pixels, cc)
    tsd = {k: v * 1.01 for k, v in sd.items()}
    ref_loss, _, _ = R.mafed_replay_loss({k: v.clone() for k, v in sd.items()}, tsd, dict(batch, patch_embeddings=feats), cfg,
                                         R.DistillSpec(modality="balanced", layer_strategy="discounted", gamma=0.5), task_id=1)
    close(loss, float(ref_loss), 1e-3, "replay loss on image inputs")


def test_replay_orders_the_teacher_behind_its_own_tower_pass():
    """A loader that hands out ``last_ready_event`` AND image batches: the event covers the loader's gather, not the features
    replay() computes on the caller's stream afterwards -- the teacher's side stream must wait for those (it used
    to wait on the loader's event only and could read ``patch_embeddings`` before the tower had written them)."""
    from mafed_amd import FeatureDistillation
    cc, csd, pixels, cfg, sd, batch, g = clip_setup("c50")
    model = build_lm(cfg, sd, torch.float32, build_tower(cc, csd, torch.float32))
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=pixels.shape[0], seed=1, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None)
    fd._update_model(model)
    fd.past_model.flat_params.mul_(1.01)
    fd.past_model._shadow_dirty = True
    fd.task_id = 1
    fd.num_vision_tokens = cc.num_patches
    mem = {k: v.to(DEV) for k, v in batch.items() if k != "patch_embeddings"}
    mem["pixel_values"] = pixels.to(DEV)

    class EventLoader:      # the HBMReplayBuffer protocol: an event that marks the handed-out batch as gathered
        def __init__(self):
            self.last_ready_event = None
        def __iter__(self):
            return self
        def __next__(self):
            self.last_ready_event = torch.cuda.current_stream().record_event()
            return dict(mem)

    fd.mem_dataloader = [dict(mem)]
    ref, _ = fd.replay(model)
    ref = float(ref)
    fd.mem_dataloader = EventLoader()
    big = torch.randn(8192, 8192, device=DEV)
    for _ in range(3):
        torch.cuda.synchronize()
        # a long queue on the caller's stream between the loader's event and the tower pass: a teacher ordered behind the loader's
        # event alone would run far ahead of the features
        for _ in range(6):
            big @ big
        got, _ = fd.replay(model)
        assert float(got) == pytest.approx(ref, rel=1e-6), "the teacher read features the tower had not written yet"


def test_bidirectional_attention_at_clip_l_geometry():
    """S = 257 (16 x 16 patches + class token), D = 64: the resident bf16 MFMA kernel and the exact fp32 kernel vs torch."""
    from mafed_amd import ops
    B, S, H, D = 2, 257, 4, 64
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(B, S, H, 3, D, generator=g)
    q, k, v = (qkv[:, :, :, i].transpose(1, 2).double() for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * S, H * D)
    o32 = ops.attn_fwd_bidir(qkv.to(DEV).view(B * S, -1), B, S, H, D)
    close(o32, ref, 2e-5, "bidirectional attention, exact fp32 kernel")
    qb = qkv.to(torch.bfloat16)
    qd, kd, vd = (qb[:, :, :, i].transpose(1, 2).double() for i in range(3))
    refb = (torch.softmax(qd @ kd.transpose(-1, -2) * D ** -0.5, dim=-1) @ vd).transpose(1, 2).reshape(B * S, H * D)
    o16 = ops.attn_fwd_bidir(qb.to(DEV).view(B * S, -1), B, S, H, D)
    close(o16.float(), refb, 2e-2, "bidirectional attention, bf16 MFMA kernel")
    # an S that is a multiple of 64 (no ragged key tile) and a short one
    for S2 in (128, 40):
        x = torch.randn(1, S2, 2, 3, D, generator=g).to(torch.bfloat16)
        a, b_, c = (x[:, :, :, i].transpose(1, 2).double() for i in range(3))
        r = (torch.softmax(a @ b_.transpose(-1, -2) * D ** -0.5, dim=-1) @ c).transpose(1, 2).reshape(S2, 2 * D)
        close(ops.attn_fwd_bidir(x.to(DEV).view(S2, -1), 1, S2, 2, D).float(), r, 2e-2, f"bidirectional attention S={S2}")


@pytest.mark.timeout(600)
def test_clip_l14_geometry_against_oracle():
    """openai/clip-vit-large-patch14 dimensions (1024 / 16 heads / 4096, 224 px -> 257 tokens) with 3 layers kept so that the fp32
    CPU oracle finishes in seconds: the bf16 MFMA tower and the fp32 kernels against the restatement."""
    cc = C.ClipVisionRefConfig(hidden_size=1024, num_hidden_layers=4, num_attention_heads=16, intermediate_size=4096, image_size=224, patch_size=14)
    csd = C.init_weights(cc, seed=3, std=0.02)
    pixels = C.make_pixels(cc, 3, seed=4)
    torch.set_num_threads(8)
    ref = C.patch_features(csd, pixels, cc)
    f32 = build_tower(cc, csd, torch.float32).patch_features(pixels.to(DEV))
    close(f32, ref, 1e-3, "CLIP-L geometry, fp32 kernels")
    f16 = build_tower(cc, csd, torch.bfloat16).patch_features(pixels.to(DEV))
    rel = float((f16.cpu().double() - ref.double()).norm() / ref.double().norm())
    assert rel <= 2e-2, f"CLIP-L geometry, bf16 tower: relative error {rel:.3e}"
