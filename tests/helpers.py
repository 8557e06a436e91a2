"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from oracle import vlpythia_ref as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY = {
    "t64": dict(h=128, H=2, L=3, V=512, P=8, T=6, B=3, Dv=32),
    "t128": dict(h=256, H=2, L=2, V=256, P=8, T=6, B=2, Dv=32),
    "t256": dict(h=256, H=1, L=2, V=256, P=8, T=6, B=2, Dv=32),
    "m64": dict(h=128, H=2, L=4, V=600, P=40, T=24, B=3, Dv=48),
}


def tiny_cfg(name):
    t = TINY[name]
    return R.RefConfig(vocab_size=t["V"], hidden_size=t["h"], num_hidden_layers=t["L"], num_attention_heads=t["H"],
                       intermediate_size=4 * t["h"], vision_hidden_size=t["Dv"], num_vision_tokens=t["P"])


def load_golden(fname):
    return np.load(os.path.join(GOLDEN, fname), allow_pickle=False)


def golden_setup(name):
    """(cfg, student weights, teacher weights, batch, golden) exactly as oracle/gen_golden.py built them."""
    g = load_golden(f"model_{name}.npz")
    cfg = tiny_cfg(name)
    seed = int(g["meta/seed"])
    sd = R.init_weights(cfg, seed=seed, bias_std=0.02, ln_jitter=0.05)
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - float(g["meta/weight_checksum"])) < 1e-6 * chk, "deterministic weight generator drifted"
    tsd = R.perturb(sd, seed=seed + 100, std=5e-3)
    batch = {k: torch.from_numpy(g["batch/" + k]) for k in ("input_ids", "attention_mask", "labels", "patch_embeddings")}
    return cfg, sd, tsd, batch, g


G3_VARIANTS = {
    "balanced_discounted_g05_mse": dict(modality="balanced", layer_strategy="discounted", gamma=0.5),
    "equal_discounted_g09_mse": dict(modality="equal", layer_strategy="discounted", gamma=0.9),
    "equal_equal_mse": dict(modality="equal", layer_strategy="equal"),
    "balanced_single_mse": dict(modality="balanced", layer_strategy="single", distillation_layer="min1"),
    "balanced_discounted_g05_cosine": dict(modality="balanced", layer_strategy="discounted", gamma=0.5, loss="cosine"),
    "cls_cosine": dict(modality="balanced", layer_strategy="discounted", gamma=0.5, loss="cosine", cls=True),
    "adaptive_discounted_g05_mse": dict(modality="adaptive", layer_strategy="discounted", gamma=0.5),
    "balanced_cumulative_mse": dict(modality="balanced", layer_strategy="cumulative", distillation_layer="nh-1", gamma=0.8),
}


def g3_spec(vname, cfg, g):
    kw = dict(G3_VARIANTS[vname])
    nh = cfg.num_hidden_layers - 1
    if kw.get("distillation_layer") == "min1":
        kw["distillation_layer"] = min(1, nh - 1)
    elif kw.get("distillation_layer") == "nh-1":
        kw["distillation_layer"] = nh - 1
    if kw["modality"] == "adaptive":
        kw["lang_coeff"] = torch.from_numpy(g["g3/adaptive_lang_coeff"])
    return R.DistillSpec(distillation_coeff=1.5, replay_coeff=0.7, **kw)


def ewc_setup(name="t64"):
    """Inputs of oracle/gen_golden.py::gen_ewc_fixture: (cfg, golden, anchor weights sd0, task-1 weights sd1, the two importance
    loaders, the step batch, the synthetic Fisher diagonal)."""
    g = load_golden(f"ewc_{name}.npz")
    cfg = tiny_cfg(name)
    t = TINY[name]
    seed = int(g["seed"])
    sd0 = R.init_weights(cfg, seed=seed)
    loaders = [[R.make_batch(cfg, t["B"], t["T"], seed=seed + 10 * r + i, pad=True) for i in range(2)] for r in range(2)]
    sd1 = R.perturb(sd0, seed=seed + 1, std=2e-3)
    batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 5, pad=True)
    syn = {k: v.abs() for k, v in R.init_weights(cfg, seed=seed + 7).items()}
    return cfg, g, sd0, sd1, loaders, batch, syn


DECODE_CASES = {"t64": "t64", "t64_eos": "t64", "t64_row0": "t64", "m64": "m64", "t128": "t128"}


def decode_setup(case):
    """Inputs of oracle/gen_golden.py::gen_decode_fixture: (cfg, weights, batch, eos or None, max_new, golden tokens, step logits, top-2 gaps)."""
    g = load_golden("decode.npz")
    name = DECODE_CASES[case]
    cfg, t = tiny_cfg(name), TINY[name]
    seed = int(g["seed"])
    sd = R.init_weights(cfg, seed=seed)
    batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 1, pad=True)
    if case.endswith("_row0"):
        batch = {k: v[:1].clone() for k, v in batch.items()}
    eos = int(g[f"{case}/eos"])
    return (cfg, sd, batch, None if eos < 0 else eos, int(g[f"{case}/max_new"]), torch.from_numpy(g[f"{case}/tokens"]),
            torch.from_numpy(g[f"{case}/step_logits"]), torch.from_numpy(g[f"{case}/top2_gap"]))


CLIP_TINY = {  # mirrors oracle/gen_golden.py::CLIP_TINY
    "c17": dict(hidden=128, heads=2, layers=3, ff=512, image=56, patch=14, B=2, lm=dict(h=128, H=2, L=2, V=384, T=6)),
    "c50": dict(hidden=192, heads=3, layers=4, ff=640, image=98, patch=14, B=3, lm=dict(h=128, H=2, L=2, V=384, T=7)),
}


def clip_setup(name):
    """Inputs of oracle/gen_golden.py::gen_clip_fixture: (tower cfg, tower weights, pixels, LM cfg, LM weights, text batch, golden)."""
    from oracle import clip_vit_ref as C
    g = load_golden(f"clip_{name}.npz")
    t = CLIP_TINY[name]
    seed = int(g["seed"])
    cc = C.ClipVisionRefConfig(hidden_size=t["hidden"], num_hidden_layers=t["layers"], num_attention_heads=t["heads"],
                               intermediate_size=t["ff"], image_size=t["image"], patch_size=t["patch"])
    csd = C.init_weights(cc, seed=seed)
    chk = float(sum(v.double().abs().sum() for v in csd.values()))
    assert abs(chk - float(g["weight_checksum"])) < 1e-6 * chk, "deterministic CLIP weight generator drifted"
    pixels = C.make_pixels(cc, t["B"], seed + 1)
    lm = t["lm"]
    cfg = R.RefConfig(vocab_size=lm["V"], hidden_size=lm["h"], num_hidden_layers=lm["L"], num_attention_heads=lm["H"],
                      intermediate_size=4 * lm["h"], vision_hidden_size=cc.hidden_size, num_vision_tokens=cc.num_patches)
    sd = R.init_weights(cfg, seed=seed + 2, bias_std=0.02, ln_jitter=0.05)
    batch = R.make_batch(cfg, t["B"], lm["T"], seed=seed + 3, pad=True, n_answer=3)
    return cc, csd, pixels, cfg, sd, batch, g
