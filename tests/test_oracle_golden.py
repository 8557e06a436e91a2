"""CPU: the oracle (oracle/vlpythia_ref.py) against the golden vectors captured from the reference classes."""
import numpy as np
import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import DECODE_CASES, G3_VARIANTS, TINY, decode_setup, ewc_setup, g3_spec, golden_setup, load_golden

TOL = 2e-5


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    assert err <= tol * scale, f"max err {err} (scale {scale})"


def grads_of(params, cfg):
    names = [k for k, _ in R.param_shapes(cfg)]
    norms = np.array([float(params[k].grad.norm()) if params[k].grad is not None else 0.0 for k in names])
    total = float(np.sqrt((norms ** 2).sum()))
    return names, norms, total


@pytest.mark.parametrize("name", list(TINY))
def test_forward_and_naive_grads(name):
    cfg, sd, tsd, batch, g = golden_setup(name)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = R.forward(params, batch, cfg)
    T = batch["input_ids"].shape[1]
    close(float(out.loss), float(g["g1/loss"]))
    close(out.logits[:, -T:].detach().numpy(), g["g1/logits_text"])
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    for i, hs in enumerate(out.hidden_states):
        close(hs.detach().numpy(), g[f"g1/hidden/{i}"])
    out.loss.backward()
    names, norms, total = grads_of(params, cfg)
    close(norms, g["g2/grad_norms"], 1e-4)
    close(total, float(g["g2/grad_norm_total"]), 1e-4)
    for key in g.files:
        if not key.startswith("g2/grad/"):
            continue
        pname = key[len("g2/grad/"):]
        if pname.endswith("/rows4"):
            close(params[pname[:-6]].grad[:4].numpy(), g[key], 1e-4)
        elif pname.endswith("/rowsum"):
            close(params[pname[:-7]].grad.sum(-1).numpy(), g[key], 1e-4)
        elif pname.endswith("/colsum"):
            close(params[pname[:-7]].grad.sum(0).numpy(), g[key], 1e-4)
        else:
            close(params[pname].grad.numpy(), g[key], 1e-4)


@pytest.mark.parametrize("name", list(TINY))
@pytest.mark.parametrize("vname", list(G3_VARIANTS))
def test_mafed_replay(name, vname):
    cfg, sd, tsd, batch, g = golden_setup(name)
    pre = f"g3/{vname}/"
    if pre + "loss" not in g.files:
        pytest.skip("variant not generated for this config")
    spec = g3_spec(vname, cfg, g)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, out, per_layer = R.mafed_replay_loss(params, tsd, batch, cfg, spec, task_id=1)
    close(float(loss), float(g[pre + "loss"]))
    layers = list(per_layer)
    assert layers == list(g[pre + "layers"])
    if not spec.cls:
        flat = []
        for l in layers:
            flat += [float(per_layer[l]["lang"]), float(per_layer[l]["vision"])]
        close(flat, g[pre + "per_call_losses"])
    loss.backward()
    names, norms, total = grads_of(params, cfg)
    close(norms, g[pre + "grad_norms"], 1e-4)
    close(total, float(g[pre + "grad_norm_total"]), 1e-4)
    close(params["gpt_neox.layers.0.input_layernorm.weight"].grad.numpy(),
          g[pre + "grad/gpt_neox.layers.0.input_layernorm.weight"], 1e-4)


def test_layer_coeffs_and_optim():
    g = load_golden("optim.npz")
    for nh in (11, 15, 23):
        for gam in (0.5, 0.8, 0.9):
            layers, c = R.layer_coeffs("discounted", nh, gam, None)
            assert layers == list(range(nh))
            close(c.numpy(), g[f"g4/discounted/nh{nh}/g{gam}"], 1e-6)
        _, c = R.layer_coeffs("equal", nh, 0.5, None)
        close(c.numpy(), g[f"g4/equal/nh{nh}"], 1e-6)
    with pytest.raises(AssertionError):
        R.layer_coeffs("single", 11, 0.5, None)
    with pytest.raises(AssertionError):
        R.layer_coeffs("single", 11, 0.5, 11)  # out of range -> None -> assertion (distillation.py:61-64)
    # AdamW / clip / schedule known answers
    p = torch.from_numpy(g["g5/p0"].copy())
    b = torch.from_numpy(g["g5/b0"].copy())
    mp, vp_, mb, vb = (torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(b), torch.zeros_like(b))
    for i in range(3):
        gp, gb = torch.from_numpy(g[f"g5/step{i}/gp"]), torch.from_numpy(g[f"g5/step{i}/gb"])
        total, scale = R.clip_grad_norm([gp, gb], 2.0)
        close(float(total), float(g[f"g5/step{i}/grad_norm"]), 1e-6)
        lr = 5e-3 * R.lr_lambda(i, 2, 10)
        close(lr, g["g5/lrs"][i], 1e-9)
        R.adamw_step(p, gp * scale, mp, vp_, i + 1, lr, 0.9, 0.98, 1e-6, 0.01)
        R.adamw_step(b, gb * scale, mb, vb, i + 1, lr, 0.9, 0.98, 1e-6, 0.0)
        close(p.numpy(), g[f"g5/step{i}/p"], 1e-6)
        close(b.numpy(), g[f"g5/step{i}/b"], 1e-6)
    close([R.lr_lambda(s, 3, 12) for s in range(14)], g["g5/lambda_w3_t12"], 1e-9)


def test_trainer_sequence():
    """RefTrainer reproduces the reference's (branch, loss, grad-norm, lr, checksum) script (SURVEY 8c last row)."""
    from tests.helpers import tiny_cfg
    g = load_golden("trainer_t64.npz")
    name, seed = "t64", int(g["meta/seed"])
    cfg = tiny_cfg(name)
    t = TINY[name]
    sd = R.init_weights(cfg, seed=seed, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=seed + 100, std=5e-3)
    tr = R.RefTrainer(cfg, sd, lr=float(g["meta/lr"]), accumulate=4, replay_interval=4, warmup_steps=int(g["meta/warmup"]),
                      total_steps=int(g["meta/total_steps"]), task_id=1, teacher_sd=tsd,
                      spec=R.DistillSpec(modality="balanced", layer_strategy="discounted", gamma=0.5))
    for bi in range(8):
        batch = R.make_batch(cfg, t["B"], t["T"], seed=seed + 10 + bi, pad=True, n_answer=3)
        mem = R.make_batch(cfg, t["B"], t["T"], seed=seed + 50 + bi, pad=True, n_answer=3)
        tr.step(batch, bi, mem)
    assert [int(r["branch"] == "replay") for r in tr.log] == list(g["seq/branch"].astype(int))
    close([r["loss"] for r in tr.log], g["seq/loss"], 1e-5)
    close([r["grad_norm"] for r in tr.log if "grad_norm" in r], g["seq/grad_norm"], 1e-4)
    close([r["lr"] for r in tr.log if "lr" in r], g["seq/lr"], 1e-9)
    close([r["param_checksum"] for r in tr.log if "param_checksum" in r], g["seq/checksum"], 1e-6)
    close(tr.params["gpt_neox.final_layer_norm.weight"].detach().numpy(), g["final/gpt_neox.final_layer_norm.weight"], 1e-5)


# ---------------------------------------------------------------------------------------------------------------
# online EWC against the reference's own EWC class
# ---------------------------------------------------------------------------------------------------------------
def _fisher_checks(f, g, tag, tol):
    names = list(f)
    close(np.array([float(f[k].double().sum()) for k in names]), g[tag + "/sum"], tol)
    for key in g.files:
        if key.startswith(tag + "/full/"):
            close(f[key[len(tag) + 6:]].numpy(), g[key], tol)
        elif key.startswith(tag + "/rows4/"):
            close(f[key[len(tag) + 7:]][:4].numpy(), g[key], tol)


def test_ewc_importances_and_online_update():
    """The importance pass runs under CPU bf16 autocast inside the reference (ewc.py:84-86): the oracle follows it through
    its own autocast forward, so agreement is at bf16 level (the two forwards round in slightly different places)."""
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    f1 = R.ewc_importances(sd0, loaders[0], cfg, autocast_bf16=True)
    _fisher_checks(f1, g, "fisher1", 2e-2)
    of = float(g["online_factor"])
    f1 = R.ewc_online_update(None, f1, task_id=0, online_factor=of)
    # second update: task_id is 1 when it runs -> overwritten, not accumulated (ewc.py:56-57)
    f2 = R.ewc_online_update(f1, R.ewc_importances(sd1, loaders[1], cfg, autocast_bf16=True), task_id=1, online_factor=of)
    _fisher_checks(f2, g, "fisher2", 2e-2)
    sd2 = R.perturb(sd1, seed=int(g["seed"]) + 2, std=2e-3)
    f3 = R.ewc_online_update(f2, R.ewc_importances(sd2, loaders[0], cfg, autocast_bf16=True), task_id=2, online_factor=of)
    _fisher_checks(f3, g, "fisher3", 2e-2)
    # fp32 importances differ from the autocast ones by far less than the terms themselves (sanity of the tolerance above)
    f32 = R.ewc_importances(sd0, loaders[0], cfg, autocast_bf16=False)
    tot = lambda d: sum(float(v.double().sum()) for v in d.values())
    assert abs(tot(f32) - tot(f1)) <= 5e-2 * tot(f32)


def test_ewc_penalty_step_exact_with_synthetic_fisher():
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    lam = float(g["reg_lambda"])
    params = {k: v.clone().requires_grad_(True) for k, v in sd1.items()}
    ce = R.forward(params, batch, cfg).loss
    total = ce + R.ewc_penalty(params, sd0, syn, lam)
    total.backward()
    close(float(ce), float(g["step_syn/ce"]))
    close(float(total), float(g["step_syn/total"]), 1e-5)
    names, norms, gtot = grads_of(params, cfg)
    close(norms, g["step_syn/grad_norms"], 1e-5)
    close(gtot, float(g["step_syn/grad_total"]), 1e-5)
    for key in g.files:
        if key.startswith("step_syn/grad/") and not key.endswith(("/rows4", "/rowsum", "/colsum")):
            close(params[key[len("step_syn/grad/"):]].grad.numpy(), g[key], 1e-5)


def test_ewc_penalty_step_with_reference_fisher():
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    lam = float(g["reg_lambda"])
    f1 = R.ewc_importances(sd0, loaders[0], cfg, autocast_bf16=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd1.items()}
    ce = R.forward(params, batch, cfg).loss
    total = ce + R.ewc_penalty(params, sd0, f1, lam)
    close(float(ce), float(g["step/ce"]))
    close(float(total), float(g["step/total"]), 5e-2)


# ---------------------------------------------------------------------------------------------------------------
# greedy validation decode against the reference model's own forward in the greedy loop
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", list(DECODE_CASES))
def test_greedy_decode(case):
    cfg, sd, batch, eos, max_new, tokens, step_logits, gaps = decode_setup(case)
    ids, steps = R.generate_greedy(sd, batch, cfg, max_new_tokens=max_new, eos_token_id=eos)
    assert torch.equal(ids, tokens)
    close(steps.numpy(), step_logits.numpy())
    T = batch["input_ids"].shape[1]
    if case == "t64_row0":
        assert ids.shape[1] - T < max_new and int(ids[0, -1]) == eos  # stopped as soon as the only row emitted eos
    if case == "t64_eos":
        first = (ids[0, T:] == eos).nonzero()[0, 0]
        assert bool((ids[0, T + int(first):] == eos).all())          # a finished row keeps emitting pad (= eos)


@pytest.mark.parametrize("name", ["c17", "c50"])
def test_clip_tower_restatement(name):
    """oracle/clip_vit_ref.py against what transformers.CLIPVisionModel produced through the reference's own
    get_patch_embeddings / forward (fixtures of oracle/gen_golden.py::gen_clip_fixture)."""
    from oracle import clip_vit_ref as C
    from tests.helpers import clip_setup
    cc, csd, pixels, cfg, sd, batch, g = clip_setup(name)
    hs = C.hidden_states(csd, pixels, cc, n_layers=cc.layers_run)
    assert len(hs) == cc.num_hidden_layers  # hidden_states[-2] of L + 1 entries
    close(hs[0], g["hidden0"], 2e-5)
    close(hs[-1][:, 0], g["penultimate_cls"], 2e-5)
    feats = C.patch_features(csd, pixels, cc)
    close(feats, g["features"], 2e-5)
    # the LM on top of those features (pixel_values -> tower -> projector -> decoder -> loss)
    out = R.forward(sd, dict(batch, patch_embeddings=feats), cfg)
    close(out.loss, float(g["loss"]), 2e-5)
    T = batch["input_ids"].shape[1]
    close(out.logits[:, -T:], g["logits_text"], 2e-5)
    close(out.hidden_states[0], g["lm_hidden0"], 2e-5)
