"""TEST INFRASTRUCTURE ONLY -- golden-vector generator.  Runs ONLY in the authoring container.

Imports the reference's own classes from /root/reference (read-only) on CPU, feeds them the build's
deterministic weights / synthetic batches (oracle/vlpythia_ref.py) and writes small ``.npz`` fixtures to
``tests/golden/``.  The reference never travels to the GPU box; these fixtures do.

Third-party packages the reference imports but the image lacks (timm, wandb, pytorch_lightning, toolz,
torchmetrics, overrides, torchvision) are satisfied by empty in-process module objects so that the
reference's *own* code for this path (VLCLIPGPTNeoXForCausalLM, compute_loss, FeatureDistillation,
DistillationWeights, AdamW, get_linear_schedule_with_warmup) runs verbatim; the frozen vision encoder is
replaced by an identity feature module (the encoder is the path's input boundary).  Decoder
arithmetic comes from the installed ``transformers`` GPT-NeoX (5.15.0; upstream pin 4.37.1).

Usage:  PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import copy
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import vlpythia_ref as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, n):
        return _Any()


def import_reference():
    import transformers.modeling_utils as mu
    from transformers import CLIPImageProcessor, CLIPVisionModel  # noqa: F401  (resolve lazies first)

    if not hasattr(mu, "load_sharded_checkpoint"):
        mu.load_sharded_checkpoint = None
    t = _stub("timm", __version__="0.9.16")
    t.models = _stub("timm.models", Eva=type("Eva", (object,), {}))
    t.data = _stub("timm.data")
    _stub("wandb", log=lambda *a, **k: None, run=None)
    _stub("wandb.wandb_run", Run=object)
    _stub("pytorch_lightning", LightningModule=type("LightningModule", (object,), {}),
          LightningDataModule=type("LightningDataModule", (object,), {}), Trainer=_Any, seed_everything=lambda s: None)
    _stub("pytorch_lightning.loggers", WandbLogger=type("WandbLogger", (object,), {}))
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.rank_zero", rank_zero_only=lambda f: f, rank_zero_warn=lambda *a, **k: None)
    _stub("pytorch_lightning.callbacks", EarlyStopping=_Any, ModelCheckpoint=_Any, RichProgressBar=_Any)
    _stub("toolz")
    _stub("toolz.sandbox", unzip=lambda seq: zip(*seq))
    _stub("torchmetrics", Metric=type("Metric", (object,), {}))
    _stub("overrides", overrides=lambda **k: (lambda f: f))
    _stub("torchvision")
    _stub("torchvision.transforms", Compose=type("Compose", (object,), {}))
    import mafed.model.vl_pythia as vp
    from mafed.methods.distillation import FeatureDistillation
    from mafed.methods.distillation_loss_weights import DistillationWeights
    from mafed.optim.adamw import AdamW
    from mafed.optim.sched import get_linear_schedule_with_warmup
    return vp, FeatureDistillation, DistillationWeights, AdamW, get_linear_schedule_with_warmup


TINY = {
    # name: dict(h, H, L, V, P, T, B, Dv)   head dims 64 / 128 / 256 mirror 160M-410M / 1.4B / 1B
    "t64": dict(h=128, H=2, L=3, V=512, P=8, T=6, B=3, Dv=32),
    "t128": dict(h=256, H=2, L=2, V=256, P=8, T=6, B=2, Dv=32),
    "t256": dict(h=256, H=1, L=2, V=256, P=8, T=6, B=2, Dv=32),
    # wider than one wave tile / more tokens than one kv tile, left padding, D=64
    "m64": dict(h=128, H=2, L=4, V=600, P=40, T=24, B=3, Dv=48),
}


def tiny_cfg(name):
    t = TINY[name]
    return R.RefConfig(vocab_size=t["V"], hidden_size=t["h"], num_hidden_layers=t["L"], num_attention_heads=t["H"],
                       intermediate_size=4 * t["h"], vision_hidden_size=t["Dv"], num_vision_tokens=t["P"])


def build_ref_model(vp, cfg: R.RefConfig, sd):
    from transformers import GPTNeoXConfig

    class FakeVision(torch.nn.Module):
        def __init__(self, dv):
            super().__init__()
            self.num_features = dv

        def forward_features(self, x):
            return x

    vp.build_vision_encoder = lambda name: FakeVision(cfg.vision_hidden_size)
    hc = GPTNeoXConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                       num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                       rotary_pct=cfg.rotary_pct, rotary_emb_base=cfg.rotary_emb_base, max_position_embeddings=2048,
                       layer_norm_eps=cfg.layer_norm_eps, tie_word_embeddings=False, hidden_dropout=0.0,
                       attention_dropout=0.0, use_parallel_residual=True, attention_bias=True)
    hc.vision_encoder_name = "fake"
    hc.select_layer = -2
    hc.select_feature = "patch"
    hc._attn_implementation = "eager"
    model = vp.VLCLIPGPTNeoXForCausalLM(hc)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "vision_encoder" in k for k in missing), missing
    # the state-dict contract: every trainable name/shape the build generates exists upstream
    ref_names = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for k, shp in R.param_shapes(cfg):
        assert ref_names[k] == shp, (k, ref_names.get(k), shp)
    return model


def ref_batch(batch, drop_labels=False):
    """Reference batch dict: pixel_values = [B,1+P,Dv] features (CLS row dropped by feature_select)."""
    f = batch["patch_embeddings"]
    pv = torch.cat([torch.zeros(f.shape[0], 1, f.shape[2]), f], dim=1)
    out = {"input_ids": batch["input_ids"].clone(), "attention_mask": batch["attention_mask"].clone(), "pixel_values": pv}
    if not drop_labels:
        out["labels"] = batch["labels"].clone()
    return out


def np_(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def grads_summary(model, cfg):
    names = [k for k, _ in R.param_shapes(cfg)]
    named = dict(model.named_parameters())
    norms = np.array([float(named[k].grad.norm()) if named[k].grad is not None else 0.0 for k in names], np.float64)
    total = float(torch.linalg.vector_norm(torch.stack([named[k].grad.norm() for k in names if named[k].grad is not None])))
    keep = ["gpt_neox.final_layer_norm.weight", "gpt_neox.layers.0.attention.query_key_value.bias",
            "gpt_neox.layers.0.input_layernorm.weight", "gpt_neox.layers.0.post_attention_layernorm.bias",
            "gpt_neox.layers.1.mlp.dense_4h_to_h.bias", "gpt_neox.layers.1.attention.dense.weight",
            "gpt_neox.layers.0.mlp.dense_h_to_4h.weight", "vision_embed_tokens.0.bias", "vision_embed_tokens.2.weight"]
    full = {}
    for k in keep:  # 1-D grads in full; matrices as first 4 rows + row/column sums (keeps fixtures small)
        g = named[k].grad
        if g.dim() == 1:
            full["grad/" + k] = np_(g)
        else:
            full["grad/" + k + "/rows4"] = np_(g[:4])
            full["grad/" + k + "/rowsum"] = np_(g.sum(-1))
            full["grad/" + k + "/colsum"] = np_(g.sum(0))
    # embedding rows that were actually used + head rows of the label ids
    full["grad/gpt_neox.embed_in.weight/rowsum"] = np_(named["gpt_neox.embed_in.weight"].grad.sum(-1))
    full["grad/embed_out.weight/rowsum"] = np_(named["embed_out.weight"].grad.sum(-1))
    return names, norms, total, full


def gen_model_fixture(name, refs, seed=7):
    vp, FD, DW, AdamW, sched = refs
    cfg = tiny_cfg(name)
    t = TINY[name]
    sd = R.init_weights(cfg, seed=seed, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=seed + 100, std=5e-3)
    batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 1, pad=True, n_answer=3)
    out = {"meta/seed": seed, "meta/name": name}
    for k, v in batch.items():
        out["batch/" + k] = np_(v)
    out["meta/weight_checksum"] = float(sum(v.double().abs().sum() for v in sd.values()))

    # --- G1/G2: naive forward + backward -------------------------------------------------------
    model = build_ref_model(vp, cfg, sd)
    model.train()
    o = model(**ref_batch(batch), output_hidden_states=True, return_dict=True)
    assert len(o.hidden_states) == cfg.num_hidden_layers + 1
    out["g1/loss"] = float(o.loss)
    out["g1/logits_text"] = np_(o.logits[:, -t["T"]:])
    for i, hsi in enumerate(o.hidden_states):
        out[f"g1/hidden/{i}"] = np_(hsi)
    o.loss.backward()
    names, norms, total, full = grads_summary(model, cfg)
    out["g2/grad_norms"] = norms
    out["g2/grad_norm_total"] = total
    out.update({"g2/" + k: v for k, v in full.items()})

    # --- G3: MAFED replay step variants ---------------------------------------------------------
    teacher_src = build_ref_model(vp, cfg, tsd)
    opts = types.SimpleNamespace(tasks=["a", "b", "c"], batch_size=t["B"], seed=42, pin_mem=False, accumulate_grad_batches=1)
    nh = cfg.num_hidden_layers - 1
    variants = [
        ("balanced_discounted_g05_mse", dict(distillation_modality_weighing_strategy="balanced",
                                             distillation_layer_weighing_strategy="discounted", gamma=0.5, distillation_layer=None)),
        ("equal_discounted_g09_mse", dict(distillation_modality_weighing_strategy="equal",
                                          distillation_layer_weighing_strategy="discounted", gamma=0.9, distillation_layer=None)),
        ("equal_equal_mse", dict(distillation_modality_weighing_strategy="equal",
                                 distillation_layer_weighing_strategy="equal", distillation_layer=None)),
        ("balanced_single_mse", dict(distillation_modality_weighing_strategy="balanced",
                                     distillation_layer_weighing_strategy="single", distillation_layer=min(1, nh - 1))),
        ("balanced_discounted_g05_cosine", dict(distillation_modality_weighing_strategy="balanced",
                                                distillation_layer_weighing_strategy="discounted", gamma=0.5,
                                                distillation_layer=None, distillation_loss="cosine")),
        ("cls_cosine", dict(distillation_modality_weighing_strategy="balanced",
                            distillation_layer_weighing_strategy="discounted", gamma=0.5, distillation_layer=None,
                            distillation_loss="cosine", cls_distillation=True)),
        ("adaptive_discounted_g05_mse", dict(distillation_modality_weighing_strategy="adaptive",
                                             distillation_layer_weighing_strategy="discounted", gamma=0.5, distillation_layer=None)),
    ]
    if nh >= 2:
        variants.append(("balanced_cumulative_mse", dict(distillation_modality_weighing_strategy="balanced",
                                                         distillation_layer_weighing_strategy="cumulative",
                                                         distillation_layer=nh - 1)))
    adaptive_vec = torch.linspace(0.3, 0.8, nh)
    for vname, kw in variants:
        model = build_ref_model(vp, cfg, sd)
        model.train()
        fd = FD(memory_size=100, opts=opts, model_type="vlpythia", num_hidden_layers=nh, distillation_coeff=1.5,
                replay_coeff=0.7, **kw)
        fd._update_model(teacher_src)
        fd.task_id = 1
        fd.num_vision_tokens = cfg.num_vision_tokens
        fd.loss_weights.num_vision_tokens = cfg.num_vision_tokens
        if kw["distillation_modality_weighing_strategy"] == "adaptive":
            fd.loss_weights.lang_coeff = adaptive_vec.clone()
            out["g3/adaptive_lang_coeff"] = np_(adaptive_vec)
        captured = {}
        orig = fd._compute_distillation_loss

        def spy(hidden_states, past_hidden_states, mask, _orig=orig, _c=captured):
            v = _orig(hidden_states=hidden_states, past_hidden_states=past_hidden_states, mask=mask)
            _c.setdefault("vals", []).append(float(v))
            return v

        fd._compute_distillation_loss = spy
        rb = ref_batch(batch)
        fd.mem_dataloader = [rb]
        loss, n_ex = fd.replay(model)
        assert "labels" not in rb and ("lang_masks" in rb or kw.get("cls_distillation"))  # side effects, SURVEY A13/A15
        loss.backward()
        _, norms, total, full = grads_summary(model, cfg)
        pre = f"g3/{vname}/"
        out[pre + "loss"] = float(loss)
        out[pre + "n_ex"] = int(n_ex)
        out[pre + "per_call_losses"] = np.array(captured.get("vals", []), np.float64)  # [lang_l0, vis_l0, lang_l1, ...]
        out[pre + "layers"] = np.array(fd.loss_weights.get_distillation_layers(), np.int64)
        lc = fd.loss_weights.layer_coeffs
        out[pre + "layer_coeffs"] = np_(lc) if lc is not None else np.array([1.0], np.float32)
        out[pre + "grad_norms"] = norms
        out[pre + "grad_norm_total"] = total
        for k in ("grad/gpt_neox.layers.0.input_layernorm.weight", "grad/vision_embed_tokens.0.bias",
                  "grad/gpt_neox.layers.1.attention.dense.weight/rows4"):
            out[pre + k] = full[k]

    # --- G7: adaptive modality weights pass (between-task; section 8f-2) ---------------------------
    model = build_ref_model(vp, cfg, sd)
    dw = DW(distillation_modality_weighing_strategy="adaptive", distillation_layer_weighing_strategy="discounted",
            gamma=0.5, num_hidden_layers=nh, distillation_layer=None, num_vision_tokens=cfg.num_vision_tokens)
    b2 = R.make_batch(cfg, t["B"], t["T"], seed=seed + 2, pad=True, n_answer=3)
    out["g7/lang_importances"] = np_(dw.compute_adaptive_weights(model, [ref_batch(batch), ref_batch(b2)]))
    for k, v in b2.items():
        out["g7/batch2/" + k] = np_(v)

    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    print(name, "loss", out["g1/loss"], "gn", out["g2/grad_norm_total"])


def gen_trainer_fixture(refs, name="t64", seed=11):
    """8 micro-batches, task 1, replay_interval=4, accumulate=4 (scripts/run_seed42.sh:59,68-69), hook order of
    SURVEY.md 8b(3): training_step -> loss/accum -> backward -> [clip 2.0 -> AdamW -> LambdaLR] on window end."""
    vp, FD, DW, AdamW, sched = refs
    cfg = tiny_cfg(name)
    t = TINY[name]
    sd = R.init_weights(cfg, seed=seed, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=seed + 100, std=5e-3)
    model = build_ref_model(vp, cfg, sd)
    model.train()
    teacher_src = build_ref_model(vp, cfg, tsd)
    opts = types.SimpleNamespace(tasks=["a", "b", "c"], batch_size=t["B"], seed=42, pin_mem=False, accumulate_grad_batches=4)
    fd = FD(memory_size=100, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
            distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
            gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
    fd._update_model(teacher_src)
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    # optimiser exactly as configure_optimizers builds it (vqa_cont_learner.py:71-128)
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight", "vqa_output_distill_loss_params"]
    named = [(n, p) for n, p in model.named_parameters()]
    lr, wd, lr_mul = 1e-3, 0.01, 10.0
    groups = [
        {"params": [p for n, p in named if "vqa_output" in n and not any(nd in n for nd in no_decay)], "lr": lr_mul * lr, "weight_decay": wd},
        {"params": [p for n, p in named if "vqa_output" in n and any(nd in n for nd in no_decay)], "lr": lr_mul * lr, "weight_decay": 0.0},
        {"params": [p for n, p in named if "vqa_output" not in n and not any(nd in n for nd in no_decay)], "lr": lr, "weight_decay": wd},
        {"params": [p for n, p in named if "vqa_output" not in n and any(nd in n for nd in no_decay)], "lr": lr, "weight_decay": 0.0},
    ]
    opt = AdamW(groups, lr=lr, betas=(0.9, 0.98))
    total_steps, warm = 20, 2
    sch = sched(opt, warm, total_steps, last_epoch=-1)
    accum, interval = 4, 4
    recs = {"branch": [], "loss": [], "grad_norm": [], "lr": [], "checksum": []}
    out = {"meta/seed": seed, "meta/name": name, "meta/lr": lr, "meta/total_steps": total_steps, "meta/warmup": warm}
    for bi in range(8):
        batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 10 + bi, pad=True, n_answer=3)
        mem = R.make_batch(cfg, t["B"], t["T"], seed=seed + 50 + bi, pad=True, n_answer=3)
        loss = None
        if (bi + 1) % interval == 0:
            fd.mem_dataloader = [ref_batch(mem)]
            loss, _ = fd.replay(model)
            recs["branch"].append(1)
        if loss is None:
            loss = model(**ref_batch(batch), compute_loss=True, return_dict=True).loss
            loss = fd.compute_loss(model, loss, batch=batch)
            recs["branch"].append(0)
        recs["loss"].append(float(loss))
        (loss / accum).backward()
        if (bi + 1) % accum == 0:
            fd.update_after_backward(model=model)
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
            recs["grad_norm"].append(float(gn))
            recs["lr"].append(opt.param_groups[2]["lr"])
            opt.step()
            sch.step()
            opt.zero_grad()
            recs["checksum"].append(float(sum(p.detach().double().sum() for n, p in model.named_parameters())))
        fd.update_after_step(model=model, batch_idx=bi)
    for k, v in recs.items():
        out["seq/" + k] = np.array(v, np.float64)
    fin = dict(model.named_parameters())
    for k in ("gpt_neox.final_layer_norm.weight", "gpt_neox.layers.0.attention.query_key_value.bias"):
        out["final/" + k] = np_(fin[k])
    for k in ("gpt_neox.layers.1.attention.dense.weight", "vision_embed_tokens.2.weight"):
        out["final/" + k + "/rows4"] = np_(fin[k][:4])
    np.savez_compressed(os.path.join(OUT, f"trainer_{name}.npz"), **out)
    print("trainer", recs["loss"], recs["grad_norm"], recs["lr"])


def gen_optim_fixture(refs):
    """G4 (layer coefficient vectors) + G5 (AdamW / LambdaLR known answers)."""
    vp, FD, DW, AdamW, sched = refs
    out = {}
    for nh in (11, 15, 23):
        for g in (0.5, 0.8, 0.9):
            dw = DW(distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                    gamma=g, num_hidden_layers=nh, distillation_layer=None)
            out[f"g4/discounted/nh{nh}/g{g}"] = np_(dw.layer_coeffs)
        dw = DW(distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="equal",
                num_hidden_layers=nh, distillation_layer=None)
        out[f"g4/equal/nh{nh}"] = np_(dw.layer_coeffs)
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal((37, 19)).astype(np.float32)
    b0 = rng.standard_normal((19,)).astype(np.float32)
    gs = [(rng.standard_normal((37, 19)).astype(np.float32), rng.standard_normal((19,)).astype(np.float32)) for _ in range(3)]
    p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    b = torch.nn.Parameter(torch.from_numpy(b0.copy()))
    opt = AdamW([{"params": [p], "weight_decay": 0.01}, {"params": [b], "weight_decay": 0.0}], lr=5e-3, betas=(0.9, 0.98))
    sch = sched(opt, 2, 10, last_epoch=-1)
    out["g5/p0"], out["g5/b0"] = p0, b0
    lrs = []
    for i, (gp, gb) in enumerate(gs):
        p.grad, b.grad = torch.from_numpy(gp.copy()), torch.from_numpy(gb.copy())
        gn = torch.nn.utils.clip_grad_norm_([p, b], 2.0)
        out[f"g5/step{i}/gp"], out[f"g5/step{i}/gb"] = gp, gb
        out[f"g5/step{i}/grad_norm"] = float(gn)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        out[f"g5/step{i}/p"], out[f"g5/step{i}/b"] = np_(p.data.clone()), np_(b.data.clone())
    out["g5/lrs"] = np.array(lrs, np.float64)
    opt2 = AdamW([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    s2 = sched(opt2, 3, 12, last_epoch=-1)
    lam = []
    for _ in range(14):
        lam.append(opt2.param_groups[0]["lr"])
        opt2.step()
        s2.step()
    out["g5/lambda_w3_t12"] = np.array(lam, np.float64)
    np.savez_compressed(os.path.join(OUT, "optim.npz"), **out)
    print("optim fixture ok")


def gen_ewc_fixture(refs, name="t64", seed=21, reg_lambda=10000.0):
    """the reference's own ``EWC`` (mafed/methods/ewc.py) -- two between-task importance passes (the
    second exercising the online decay), the per-step quadratic penalty and the gradient of CE + penalty."""
    from mafed.methods.ewc import EWC
    import mafed.methods.ewc as ewc_mod
    ewc_mod.tqdm = lambda it, **k: it
    vp = refs[0]
    cfg = tiny_cfg(name)
    t = TINY[name]
    sd0 = R.init_weights(cfg, seed=seed)
    model = build_ref_model(vp, cfg, sd0)
    names = [k for k, _ in R.param_shapes(cfg)]
    loaders = [[ref_batch(R.make_batch(cfg, t["B"], t["T"], seed=seed + 10 * r + i, pad=True)) for i in range(2)] for r in range(2)]
    ewc = EWC(reg_lambda=reg_lambda, online=True, online_factor=0.95)
    out = {"reg_lambda": np.float64(reg_lambda), "online_factor": np.float64(0.95), "seed": np.int64(seed)}

    def summarise(tag):
        f = ewc.fisher[0]
        out[tag + "/sum"] = np.array([float(f[k].double().sum()) for k in names], np.float64)
        out[tag + "/max"] = np.array([float(f[k].max()) for k in names], np.float64)
        for k in ("gpt_neox.layers.0.input_layernorm.weight", "gpt_neox.layers.1.mlp.dense_4h_to_h.bias", "vision_embed_tokens.0.bias"):
            out[tag + "/full/" + k] = np_(f[k])
        out[tag + "/rows4/gpt_neox.layers.0.mlp.dense_h_to_4h.weight"] = np_(f["gpt_neox.layers.0.mlp.dense_h_to_4h.weight"][:4])

    # task 0 -> 1: importances at sd0 (CPU bf16 autocast inside the reference, ewc.py:86)
    ewc.update(model=model, dataloader=loaders[0])
    summarise("fisher1")
    assert ewc.task_id == 1
    # "training" on task 1 = a deterministic perturbation; then one step's loss and gradient with the penalty
    sd1 = R.perturb(sd0, seed=seed + 1, std=2e-3)
    model.load_state_dict(sd1, strict=False)
    batch = ref_batch(R.make_batch(cfg, t["B"], t["T"], seed=seed + 5, pad=True))
    model.zero_grad()
    ce = model(**batch, compute_loss=True, return_dict=True).loss
    total = ewc.compute_loss(model, ce.clone())
    total.backward()
    out["step/ce"] = np.float64(float(ce.detach()))
    out["step/total"] = np.float64(float(total.detach()))
    _, norms, gtot, full = grads_summary(model, cfg)
    out["step/grad_norms"] = norms
    out["step/grad_total"] = np.float64(gtot)
    for k, v in full.items():
        out["step/" + k] = v
    # the same step with a synthetic Fisher diagonal (|N(0, 0.02)| from the build's generator) installed in the reference
    # object: pins compute_regularization and its gradient exactly, free of the bf16 noise of the importance pass
    real_fisher = ewc.fisher[0]
    syn = {k: v.abs() for k, v in R.init_weights(cfg, seed=seed + 7).items()}
    ewc.fisher[0] = {k: syn[k].clone() for k in real_fisher}
    model.zero_grad()
    ce2 = model(**batch, compute_loss=True, return_dict=True).loss
    total2 = ewc.compute_loss(model, ce2.clone())
    total2.backward()
    out["step_syn/ce"] = np.float64(float(ce2.detach()))
    out["step_syn/total"] = np.float64(float(total2.detach()))
    _, norms2, gtot2, full2 = grads_summary(model, cfg)
    out["step_syn/grad_norms"] = norms2
    out["step_syn/grad_total"] = np.float64(gtot2)
    for k, v in full2.items():
        out["step_syn/" + k] = v
    ewc.fisher[0] = real_fisher
    # task 1 -> 2: task_id is still 1 when update() runs, so the importances are overwritten (ewc.py:56-57); anchor := sd1
    model.zero_grad()
    ewc.update(model=model, dataloader=loaders[1])
    summarise("fisher2")
    assert ewc.task_id == 2
    # task 2 -> 3: the online accumulation proper, fisher = new + 0.95 * old (ewc.py:58-61), at weights sd2
    sd2 = R.perturb(sd1, seed=seed + 2, std=2e-3)
    model.load_state_dict(sd2, strict=False)
    model.zero_grad()
    ewc.update(model=model, dataloader=loaders[0])
    summarise("fisher3")
    assert ewc.task_id == 3
    np.savez_compressed(os.path.join(OUT, f"ewc_{name}.npz"), **out)
    print("ewc fixture ok: ce %.5f total %.5f (synthetic Fisher: %.5f)" % (float(ce), float(total), float(total2)))


def gen_decode_fixture(refs, seed=31):
    """SURVEY.md section 8f-3.  The reference's ``generate`` comes from HF's GenerationMixin, which the installed transformers
    5.x no longer mixes into PreTrainedModel (the reference pins 4.37.1), so the fixture is built from what that call
    computes step by step: the reference model's OWN forward on the growing sequence (use_cache=False re-runs the full
    prefix per token) inside the published greedy-search loop (oracle generate_greedy with logits_fn = reference forward)."""
    vp = refs[0]
    out = {"seed": np.int64(seed)}
    # (case, tiny config, eos): with random weights no fixed id ever wins the argmax, so the early-stop case takes as eos the
    # token row 0 emits at its third step -- row 0 then finishes early and keeps emitting the pad id while the others go on
    picked = {}
    for case, name, eos in (("t64", "t64", None), ("t64_eos", "t64", "pick"), ("t64_row0", "t64", "pick"), ("m64", "m64", None),
                            ("t128", "t128", None)):
        cfg = tiny_cfg(name)
        t = TINY[name]
        sd = R.init_weights(cfg, seed=seed)
        model = build_ref_model(vp, cfg, sd)
        model.eval()
        batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 1, pad=True)
        if case.endswith("_row0"):  # a single row: generation ends as soon as that row emits eos (output shorter than max_new)
            batch = {k: v[:1].clone() for k, v in batch.items()}
        pv = ref_batch(batch, drop_labels=True)["pixel_values"]

        def ref_logits(ids, am):
            with torch.no_grad():
                return model(input_ids=ids, attention_mask=am, pixel_values=pv, return_dict=True).logits

        if eos == "pick":
            eos = int(picked[name][0, t["T"] + 2])
        n_new = 10 if name != "m64" else 6
        ids, steps = R.generate_greedy(sd, batch, cfg, max_new_tokens=n_new, eos_token_id=eos, logits_fn=ref_logits)
        picked.setdefault(name, ids)
        # the oracle's own forward inside the same loop must agree (pins the restatement, not only the loop)
        ids2, steps2 = R.generate_greedy(sd, batch, cfg, max_new_tokens=n_new, eos_token_id=eos)
        assert torch.equal(ids, ids2), (case, ids, ids2)
        assert float((steps - steps2).abs().max()) < 2e-5 * max(1.0, float(steps.abs().max()))
        out[f"{case}/tokens"] = np_(ids)
        out[f"{case}/step_logits"] = np_(steps)
        out[f"{case}/eos"] = np.int64(-1 if eos is None else eos)
        out[f"{case}/max_new"] = np.int64(n_new)
        top2 = steps.topk(2, dim=-1).values
        out[f"{case}/top2_gap"] = np_(top2[..., 0] - top2[..., 1])
        print("decode fixture", case, "eos", eos, "generated", ids.shape[1] - t["T"], "tokens; min top-2 gap %.3e" % float((top2[..., 0] - top2[..., 1]).min()),
              "| row 0:", ids[0, t["T"]:].tolist())
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **out)


CLIP_TINY = {
    # name: CLIP tower dims + the LM config it feeds (vision_hidden_size = the tower's hidden size, P = its patch count)
    "c17": dict(hidden=128, heads=2, layers=3, ff=512, image=56, patch=14, B=2, lm=dict(h=128, H=2, L=2, V=384, T=6)),
    "c50": dict(hidden=192, heads=3, layers=4, ff=640, image=98, patch=14, B=3, lm=dict(h=128, H=2, L=2, V=384, T=7)),
}


def clip_cfg(name):
    from oracle import clip_vit_ref as C
    t = CLIP_TINY[name]
    return C.ClipVisionRefConfig(hidden_size=t["hidden"], num_hidden_layers=t["layers"], num_attention_heads=t["heads"],
                                 intermediate_size=t["ff"], image_size=t["image"], patch_size=t["patch"])


def gen_clip_fixture(refs, seed=41):
    """Frozen CLIP tower: ``transformers.CLIPVisionModel`` built offline from a ``CLIPVisionConfig`` with the
    build's deterministic weights, called exactly as the reference calls it -- through the reference's own
    ``VLCLIPGPTNeoXForCausalLM.get_patch_embeddings`` / ``forward`` with ``pixel_values`` [B,3,H,W]."""
    from transformers import CLIPVisionConfig, CLIPVisionModel, GPTNeoXConfig
    from oracle import clip_vit_ref as C
    vp = refs[0]
    for name, t in CLIP_TINY.items():
        cc = clip_cfg(name)
        csd = C.init_weights(cc, seed=seed)
        hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=cc.hidden_size, intermediate_size=cc.intermediate_size,
                                              num_hidden_layers=cc.num_hidden_layers, num_attention_heads=cc.num_attention_heads,
                                              image_size=cc.image_size, patch_size=cc.patch_size, num_channels=3, hidden_act="quick_gelu",
                                              layer_norm_eps=cc.layer_norm_eps, attention_dropout=0.0))
        hf.config._attn_implementation = "eager"
        have = {k: tuple(v.shape) for k, v in hf.state_dict().items()}
        # checkpoint names carry the ``vision_model.`` prefix (transformers 4.37.1, the hub files); the installed 5.x class
        # registers the same tensors without it
        strip = not any(k.startswith("vision_model.") for k in have)
        fix = (lambda k: k[len("vision_model."):]) if strip else (lambda k: k)
        for k, shp in C.param_shapes(cc):   # the tower's state-dict contract
            assert have[fix(k)] == shp, (k, have.get(fix(k)), shp)
        # the LM around it, through the reference class (CLIP branch of build_vision_encoder, vl_pythia.py:196-198)
        lm = t["lm"]
        cfg = R.RefConfig(vocab_size=lm["V"], hidden_size=lm["h"], num_hidden_layers=lm["L"], num_attention_heads=lm["H"],
                          intermediate_size=4 * lm["h"], vision_hidden_size=cc.hidden_size, num_vision_tokens=cc.num_patches)
        sd = R.init_weights(cfg, seed=seed + 2, bias_std=0.02, ln_jitter=0.05)
        vp.build_vision_encoder = lambda nm: hf
        hc = GPTNeoXConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                           num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                           rotary_pct=cfg.rotary_pct, rotary_emb_base=cfg.rotary_emb_base, max_position_embeddings=2048,
                           layer_norm_eps=cfg.layer_norm_eps, tie_word_embeddings=False, hidden_dropout=0.0,
                           attention_dropout=0.0, use_parallel_residual=True, attention_bias=True)
        hc.vision_encoder_name = "openai/clip-tiny"
        hc.select_layer = -2
        hc.select_feature = "patch"
        hc._attn_implementation = "eager"
        # (VLCLIPGPTNeoXForCausalLM reads vision_encoder.config.hidden_size for CLIP towers, vl_pythia.py:219-224)
        model = vp.VLCLIPGPTNeoXForCausalLM(hc)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        # (post_init() re-initialises every sub-module, the tower included: load the tower's weights afterwards)
        missing, unexpected = hf.load_state_dict({fix(k): v for k, v in csd.items()}, strict=False)
        assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
        assert model.vision_encoder is hf
        model.eval()
        pixels = C.make_pixels(cc, t["B"], seed + 1)
        with torch.no_grad():
            outs = hf(pixels, output_hidden_states=True)
        assert len(outs.hidden_states) == cc.num_hidden_layers + 1
        batch = R.make_batch(cfg, t["B"], lm["T"], seed=seed + 3, pad=True, n_answer=3)
        with torch.no_grad():
            feats = model.get_patch_embeddings(pixels)
            out = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], pixel_values=pixels,
                        output_hidden_states=True, return_dict=True)
        assert feats.shape == (t["B"], cc.num_patches, cc.hidden_size)
        assert torch.equal(feats, outs.hidden_states[-2][:, 1:])
        # restatement check while the reference is at hand
        mine = C.patch_features(csd, pixels, cc)
        err = float((mine - feats).abs().max())
        assert err < 2e-5, err
        g = {"seed": np.array(seed), "hidden0": np_(outs.hidden_states[0]), "features": np_(feats),
             "penultimate_cls": np_(outs.hidden_states[-2][:, 0]), "loss": np_(out.loss), "logits_text": np_(out.logits[:, -lm["T"]:]),
             "lm_hidden0": np_(out.hidden_states[0]), "weight_checksum": np.array(float(sum(v.double().abs().sum() for v in csd.values())))}
        np.savez_compressed(os.path.join(OUT, f"clip_{name}.npz"), **g)
        print("clip fixture", name, "S", cc.num_patches + 1, "features", tuple(feats.shape), "restatement err %.2e" % err, "loss %.5f" % float(out.loss))


def main():
    os.makedirs(OUT, exist_ok=True)
    import logging
    import transformers
    transformers.logging.set_verbosity_error()
    logging.disable(logging.INFO)
    torch.manual_seed(0)
    torch.set_num_threads(4)
    refs = import_reference()
    for name in TINY:
        gen_model_fixture(name, refs)
    gen_trainer_fixture(refs)
    gen_optim_fixture(refs)
    gen_ewc_fixture(refs)
    gen_decode_fixture(refs)
    gen_clip_fixture(refs)


if __name__ == "__main__":
    main()
