"""GPU parity of the whole hot path (model forward/backward, MAFED replay step, Trainer.step sequence) through the
HIP kernels, against the golden vectors captured from the reference classes and against the oracle."""
import types

import numpy as np
import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import G3_VARIANTS, TINY, g3_spec, golden_setup, load_golden, tiny_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3  # the north star's fp32 gate; most quantities land near 1e-5


def close(a, b, tol=TOL, what=""):
    a = np.asarray(a.detach().cpu().double() if isinstance(a, torch.Tensor) else a, np.float64)
    b = np.asarray(b.detach().cpu().double() if isinstance(b, torch.Tensor) else b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.1e}*{scale:.3g}"


def build_model(cfg: R.RefConfig, sd, dtype=torch.float32):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    mc = VLPythiaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        vision_hidden_size=cfg.vision_hidden_size, num_vision_tokens=cfg.num_vision_tokens)
    m = VLPythiaForCausalLM(mc, compute_dtype=dtype, device=DEV)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    return m


def to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def grad_norms(model, cfg):
    names = [k for k, _ in R.param_shapes(cfg)]
    return names, np.array([float(model._g(k).norm()) for k in names])


def check_named_grads(model, g, prefix, tol):
    for key in g.files:
        if not key.startswith(prefix + "grad/"):
            continue
        pname = key[len(prefix + "grad/"):]
        if pname.endswith("/rows4"):
            close(model._g(pname[:-6])[:4], g[key], tol, key)
        elif pname.endswith("/rowsum"):
            close(model._g(pname[:-7]).sum(-1), g[key], tol, key)
        elif pname.endswith("/colsum"):
            close(model._g(pname[:-7]).sum(0), g[key], tol, key)
        else:
            close(model._g(pname), g[key], tol, key)


@pytest.mark.parametrize("name", list(TINY))
def test_forward_backward_vs_reference_golden(name):
    cfg, sd, tsd, batch, g = golden_setup(name)
    model = build_model(cfg, sd)
    T = batch["input_ids"].shape[1]
    out = model(**to_dev(batch), output_hidden_states=True, return_dict=True)
    close(out.loss, float(g["g1/loss"]), TOL, "loss")
    close(out.logits, g["g1/logits_text"], TOL, "logits (text positions)")
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    for i, hs in enumerate(out.hidden_states):
        close(hs, g[f"g1/hidden/{i}"], TOL, f"hidden {i}")
    model.zero_grad()
    out.loss.backward()
    names, norms = grad_norms(model, cfg)
    close(norms, g["g2/grad_norms"], TOL, "per-parameter grad norms")
    close(float(np.sqrt((norms ** 2).sum())), float(g["g2/grad_norm_total"]), TOL, "global grad norm")
    check_named_grads(model, g, "g2/", TOL)
    # gradients accumulate across micro-batches (Lightning accumulate_grad_batches)
    out2 = model(**to_dev(batch), return_dict=True)
    out2.loss.backward()
    _, norms2 = grad_norms(model, cfg)
    close(norms2, 2 * g["g2/grad_norms"], TOL, "accumulated grads")


def make_fd(cfg, vname, g, teacher_model, batch, B):
    from mafed_amd import FeatureDistillation
    spec = g3_spec(vname, cfg, g)
    opts = types.SimpleNamespace(tasks=["a", "b", "c"], batch_size=B, seed=42, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=100, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_coeff=spec.distillation_coeff, replay_coeff=spec.replay_coeff,
                             distillation_modality_weighing_strategy=spec.modality,
                             distillation_layer_weighing_strategy=spec.layer_strategy, gamma=spec.gamma,
                             distillation_layer=spec.distillation_layer, distillation_loss=spec.loss, cls_distillation=spec.cls)
    fd._update_model(teacher_model)
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    if spec.modality == "adaptive":
        fd.loss_weights.lang_coeff = spec.lang_coeff.to(DEV)
    return fd, spec


@pytest.mark.parametrize("name", ["t64", "m64", "t128", "t256"])
@pytest.mark.parametrize("vname", list(G3_VARIANTS))
def test_mafed_replay_vs_reference_golden(name, vname):
    cfg, sd, tsd, batch, g = golden_setup(name)
    pre = f"g3/{vname}/"
    if pre + "loss" not in g.files:
        pytest.skip("variant not generated for this config")
    model = build_model(cfg, sd)
    teacher = build_model(cfg, tsd)
    fd, spec = make_fd(cfg, vname, g, teacher, batch, batch["input_ids"].shape[0])
    mem = to_dev(batch)
    fd.mem_dataloader = [mem]
    model.zero_grad()
    loss, n_ex = fd.replay(model)
    assert n_ex == int(g[pre + "n_ex"])
    assert "labels" not in mem  # the reference pops labels from the caller's dict (distillation.py:221)
    close(loss, float(g[pre + "loss"]), TOL, "replay loss")
    assert fd.loss_weights.get_distillation_layers() == list(g[pre + "layers"])
    if not spec.cls:
        close(fd.last_modality_losses.reshape(-1), g[pre + "per_call_losses"], TOL, "per-layer lang/vision losses")
    loss.backward()
    names, norms = grad_norms(model, cfg)
    close(norms, g[pre + "grad_norms"], TOL, "grad norms")
    close(float(np.sqrt((norms ** 2).sum())), float(g[pre + "grad_norm_total"]), TOL, "global grad norm")
    check_named_grads(model, g, pre, TOL)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("replay_coeff", [0.7, 0.0])
@pytest.mark.parametrize("vname", ["equal_discounted_g09_mse", "balanced_discounted_g05_cosine"])
def test_mafed_fused_and_generic_distillation_paths(fused, replay_coeff, vname, monkeypatch):
    """Distillation gradient (MSE, and the cosine distance of distillation.py:226-235) injected inside the LayerNorm-backward
    kernels (fused) vs materialised per layer through plain autograd (generic; what the reference's own plugin code would
    exercise), with and without the replay CE term.  The fused path must not materialise a per-layer gradient tensor."""
    from mafed_amd import ops
    cfg, sd, tsd, batch, g = golden_setup("m64")
    model, teacher = build_model(cfg, sd), build_model(cfg, tsd)
    fd, spec = make_fd(cfg, vname, g, teacher, batch, batch["input_ids"].shape[0])
    fd.replay_coeff = replay_coeff
    fd.fused_distill = fused
    fd.mem_dataloader = [to_dev(batch)]
    calls = []
    real = ops.distill_bwd
    monkeypatch.setattr(ops, "distill_bwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    model.zero_grad()
    loss, _ = fd.replay(model)
    loss.backward()
    n_distilled = len(fd.loss_weights.get_distillation_layers())
    # (with no CE term the deepest distilled state starts the gradient chain: that one row tensor is the chain's seed, not an extra pass)
    assert len(calls) == ((1 if replay_coeff == 0.0 else 0) if fused else n_distilled), (fused, len(calls))
    spec.replay_coeff = replay_coeff
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref_loss, _, _ = R.mafed_replay_loss(params, tsd, batch, cfg, spec, task_id=1)
    ref_loss.backward()
    close(loss, float(ref_loss), TOL, "loss")
    names, norms = grad_norms(model, cfg)
    refn = np.array([float(params[k].grad.norm()) if params[k].grad is not None else 0.0 for k in names])
    close(norms, refn, TOL, "grad norms")


def test_layer_strategy_errors_match_reference():
    from mafed_amd import FeatureDistillation
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=2, seed=1, pin_mem=False, accumulate_grad_batches=1)
    with pytest.raises(AssertionError):
        FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", distillation_layer_weighing_strategy="single",
                            distillation_layer=None, num_hidden_layers=11)
    with pytest.raises(AssertionError):
        FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", distillation_layer_weighing_strategy="cumulative",
                            distillation_layer=11, num_hidden_layers=11)  # out of range -> None -> assertion


@pytest.mark.parametrize("pipeline", [False, True])
def test_trainer_sequence_vs_reference_golden(pipeline):
    """8 micro-batches, task 1, replay_interval 4, accumulate 4: (branch, loss, grad-norm, lr, parameter checksum).
    pipeline=True: AdamW + gradient zeroing chunk by chunk on their own stream, the next forward waiting layer by layer
    (Trainer(pipeline_optimizer=True), what bench.py runs); parameters are read after Trainer.join()."""
    from mafed_amd import FeatureDistillation, Trainer
    g = load_golden("trainer_t64.npz")
    name, seed = "t64", int(g["meta/seed"])
    cfg = tiny_cfg(name)
    t = TINY[name]
    sd = R.init_weights(cfg, seed=seed, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=seed + 100, std=5e-3)
    model, teacher = build_model(cfg, sd), build_model(cfg, tsd)
    opts = types.SimpleNamespace(tasks=["a", "b", "c"], batch_size=t["B"], seed=42, pin_mem=False, accumulate_grad_batches=4)
    fd = FeatureDistillation(memory_size=100, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
    fd._update_model(teacher)
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    conf = types.SimpleNamespace(accumulate_grad_batches=4, replay_interval=4, grad_norm=2.0, learning_rate=float(g["meta/lr"]),
                                 betas=(0.9, 0.98), weight_decay=0.01, optim="adamw", warmup_steps=int(g["meta/warmup"]),
                                 total_steps=int(g["meta/total_steps"]))
    tr = Trainer(model, fd, conf, task_id=1, pipeline_optimizer=pipeline)
    assert tr.pipeline_optimizer == pipeline
    branches, losses, gns, lrs, sums = [], [], [], [], []
    for bi in range(8):
        batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 10 + bi, pad=True, n_answer=3)
        mem = R.make_batch(cfg, t["B"], t["T"], seed=seed + 50 + bi, pad=True, n_answer=3)
        fd.mem_dataloader = [to_dev(mem)]
        rec = tr.step(to_dev(batch), bi)
        branches.append(int(rec["branch"] == "replay"))
        losses.append(float(rec["loss"]))
        if rec["stepped"]:
            gns.append(float(rec["grad_norm"]))
            lrs.append(rec["lr"])
            if pipeline and bi == 3:
                assert model._param_events is not None  # left for the next forward to consume chunk by chunk
            else:
                tr.join()
                sums.append(float(sum(p.detach().double().sum() for p in model.parameters())))
                assert float(model.flat_grads.abs().max()) == 0.0
    assert branches == list(g["seq/branch"].astype(int))
    close(np.array(losses), g["seq/loss"], TOL, "loss sequence")
    close(np.array(gns), g["seq/grad_norm"], TOL, "grad-norm sequence")
    close(np.array(lrs), g["seq/lr"], 1e-9, "lr sequence")
    close(np.array(sums), g["seq/checksum"][-len(sums):], 1e-5, "parameter checksum after each optimiser step")
    tr.join()
    close(model._p("gpt_neox.final_layer_norm.weight"), g["final/gpt_neox.final_layer_norm.weight"], 1e-4, "final LN weight")


def test_adaptive_weights_pass_vs_reference_golden():
    """compute_adaptive_weights (between-task pass of MAFED-A) from one tapped backward sweep per batch."""
    from mafed_amd.methods import DistillationWeights
    for name in ("t64", "m64"):
        cfg, sd, tsd, batch, g = golden_setup(name)
        model = build_model(cfg, sd)
        b2 = {k: torch.from_numpy(g["g7/batch2/" + k]) for k in ("input_ids", "attention_mask", "labels", "patch_embeddings")}
        dw = DistillationWeights("adaptive", "discounted", gamma=0.5, num_hidden_layers=cfg.num_hidden_layers - 1,
                                 distillation_layer=None, num_vision_tokens=cfg.num_vision_tokens)
        imp = dw.compute_adaptive_weights(model, [to_dev(batch), to_dev(b2)])
        close(imp, g["g7/lang_importances"], TOL, "adaptive lang importances")


@pytest.mark.parametrize("name", ["t64", "m64", "t128", "t256"])
def test_bf16_mode_tracks_oracle_autocast(name):
    """Perf-path numerics (bf16 MFMA GEMMs/attention, fp32 residual stream) against the oracle run under bf16
    autocast -- what the reference's Lightning precision="bf16" computes.  Loose tolerance: both sides round to bf16."""
    cfg, sd, tsd, batch, g = golden_setup(name)
    model = build_model(cfg, sd, torch.bfloat16)
    out = model(**to_dev(batch), output_hidden_states=True, return_dict=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = R.forward(params, batch, cfg, autocast_bf16=True)
    assert out.hidden_states[0].dtype == torch.float32  # the residual stream stays fp32 (SURVEY.md section 8 numerics)
    close(out.loss, float(ref.loss), 3e-2, "bf16 loss")
    for i in range(cfg.num_hidden_layers):
        close(out.hidden_states[i], ref.hidden_states[i].detach().float(), 3e-2, f"bf16 hidden {i}")
    model.zero_grad()
    out.loss.backward()
    ref.loss.backward()
    names, norms = grad_norms(model, cfg)
    refn = np.array([float(params[k].grad.norm()) for k in names])
    tot, rtot = np.sqrt((norms ** 2).sum()), np.sqrt((refn ** 2).sum())
    assert abs(tot - rtot) <= 5e-2 * rtot, (tot, rtot)


def test_config1_shape_against_oracle():
    """BASELINE config[0]: VLPythia-160M shapes, naive step, B=4, 64 image + 16 text tokens, fp32 -- the HIP path
    against the oracle at real width/depth (scalars only)."""
    cfg = R.preset("160m", num_vision_tokens=64)
    sd = R.init_weights(cfg, seed=1234)
    batch = R.make_batch(cfg, 4, 16, seed=1235, pad=True)
    model = build_model(cfg, sd)
    out = model(**to_dev(batch), return_dict=True)
    model.zero_grad()
    out.loss.backward()
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = R.forward(params, batch, cfg)
    ref.loss.backward()
    close(out.loss, float(ref.loss), TOL, "loss")
    close(out.logits, ref.logits[:, -16:].detach(), TOL, "logits")
    names, norms = grad_norms(model, cfg)
    refn = np.array([float(params[k].grad.norm()) for k in names])
    close(norms, refn, TOL, "per-parameter grad norms")
    close(float(np.sqrt((norms ** 2).sum())), float(np.sqrt((refn ** 2).sum())), TOL, "global grad norm")


def test_deepcopy_teacher_and_state_dict_contract():
    import copy
    cfg, sd, tsd, batch, g = golden_setup("t64")
    model = build_model(cfg, sd)
    names = [k for k, _ in R.param_shapes(cfg)]
    got = dict(model.state_dict())
    assert list(got) == names  # registration order and names of the reference state dict
    t = copy.deepcopy(model)
    t.eval()
    assert t.flat_params.data_ptr() != model.flat_params.data_ptr()
    assert torch.equal(t.flat_params, model.flat_params)
    hs = t.hidden_states_upto(batch["input_ids"].to(DEV), batch["attention_mask"].to(DEV), patch_embeddings=batch["patch_embeddings"].to(DEV), n_hidden=2)
    assert len(hs) == 2
    close(hs[1], g["g1/hidden/1"], TOL, "teacher fast path hidden 1")
    assert list(model.vision_encoder.parameters()) == []
