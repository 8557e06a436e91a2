"""GPU parity of the first "next" row: online EWC through the HIP penalty kernels and the native
model, against the golden vectors captured from the reference's own ``EWC`` class (tests/golden/ewc_t64.npz) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import ewc_setup
from tests.test_gpu_model import DEV, build_model, check_named_grads, close, grad_norms, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 7, 4096, 1000003])
def test_ewc_penalty_kernels(n):
    from mafed_amd import ops
    g = torch.Generator().manual_seed(n)
    p, q = torch.randn(n, generator=g), torch.randn(n, generator=g)
    f = torch.rand(n, generator=g)
    grad0 = torch.randn(n, generator=g)
    lam = 3.5
    want = 0.5 * lam * (f.double() * (p.double() - q.double()) ** 2).sum()
    out = ops.ewc_penalty_fwd(p.to(DEV), q.to(DEV), f.to(DEV), 0.5 * lam)
    close(out.reshape(()), want, 2e-6, "penalty")
    out2 = ops.ewc_penalty_fwd(p.to(DEV), q.to(DEV), f.to(DEV), 0.5 * lam, out=out.clone(), beta=1.0)  # chained per-task terms
    close(out2.reshape(()), 2 * want, 2e-6, "penalty (beta = 1)")
    grad = grad0.to(DEV)
    coef = torch.tensor([0.25], device=DEV)
    ops.ewc_penalty_bwd_(p.to(DEV), q.to(DEV), f.to(DEV), lam, coef, grad)
    close(grad, grad0.double() + 0.25 * lam * f.double() * (p.double() - q.double()), 1e-6, "penalty gradient")


def _flat_of(model_like, sd):
    """Flat fp32 tensor laid out like ``flat_params`` holding the values of a state dict."""
    m = build_model(model_like, sd)
    return m.flat_params.detach().clone()


def test_ewc_step_with_synthetic_fisher_matches_reference_golden():
    """compute_regularization + its gradient, exact: the Fisher diagonal is the synthetic one the reference object was
    given when the fixture was made (no bf16 importance pass involved)."""
    from mafed_amd import CLMethod
    from mafed_amd.methods.ewc import _FlatDict
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    model = build_model(cfg, sd1)
    ewc = CLMethod["ewc"](reg_lambda=float(g["reg_lambda"]))
    ewc.fisher[0] = _FlatDict(model, _flat_of(cfg, syn))
    ewc.old_params[0] = _FlatDict(model, _flat_of(cfg, sd0))
    ewc.task_id = 1
    model.zero_grad()
    ce = model(**to_dev(batch), compute_loss=True, return_dict=True).loss
    total = ewc.compute_loss(model, ce, batch=batch)
    total.backward()
    close(ce, float(g["step_syn/ce"]), 1e-4, "ce")
    close(total, float(g["step_syn/total"]), 1e-4, "ce + penalty")
    names, norms = grad_norms(model, cfg)
    close(norms, g["step_syn/grad_norms"], 1e-4, "per-parameter grad norms")
    close(float(np.sqrt((norms ** 2).sum())), float(g["step_syn/grad_total"]), 1e-4, "global grad norm")
    check_named_grads(model, g, "step_syn/", 1e-4)
    # task 0: no penalty at all (ewc.py:118-119)
    ewc.task_id = 0
    assert float(ewc.compute_loss(model, ce.detach())) == float(ce.detach())


def _check_fisher(fd, g, tag, tol, names):
    sums = np.array([float(fd[k].double().sum()) for k in names])
    ref = g[tag + "/sum"]
    err = np.abs(sums - ref).max()
    assert err <= tol * np.abs(ref).max(), f"{tag}: per-parameter Fisher sums, max err {err:.3e} vs scale {np.abs(ref).max():.3e}"
    for key in g.files:
        if key.startswith(tag + "/full/"):
            close(fd[key[len(tag) + 6:]], g[key], tol, key)
        elif key.startswith(tag + "/rows4/"):
            close(fd[key[len(tag) + 7:]][:4], g[key], tol, key)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-2), (torch.bfloat16, 6e-2)])
def test_ewc_importance_passes_and_online_update_vs_reference_golden(dtype, tol):
    """update() three times (overwrite, overwrite, decayed accumulation -- ewc.py:52-62) through the native model.  The
    reference ran its importance pass under CPU bf16 autocast: the fp32 kernels sit within the autocast noise of it, the bf16
    MFMA mode rounds in other places again."""
    from mafed_amd import CLMethod
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    names = [k for k, _ in R.param_shapes(cfg)]
    model = build_model(cfg, sd0, dtype=dtype)
    ewc = CLMethod["ewc"](reg_lambda=float(g["reg_lambda"]), online=True, online_factor=float(g["online_factor"]))
    ewc.update(model=model, dataloader=[to_dev(b) for b in loaders[0]])
    assert ewc.task_id == 1 and float(model.flat_grads.abs().max()) == 0.0
    _check_fisher(ewc.fisher[0], g, "fisher1", tol, names)
    assert torch.equal(ewc.old_params[0].flat, model.flat_params)
    model.load_state_dict(sd1)
    ewc.update(model=model, dataloader=[to_dev(b) for b in loaders[1]])
    _check_fisher(ewc.fisher[0], g, "fisher2", tol, names)
    sd2 = R.perturb(sd1, seed=int(g["seed"]) + 2, std=2e-3)
    model.load_state_dict(sd2)
    ewc.update(model=model, dataloader=[to_dev(b) for b in loaders[0]])
    assert ewc.task_id == 3
    _check_fisher(ewc.fisher[0], g, "fisher3", tol, names)
    if dtype == torch.float32:
        # and tightly against the oracle's fp32 importances (same arithmetic, no autocast on either side)
        f32 = R.ewc_importances(sd0, loaders[0], cfg, autocast_bf16=False)
        m2 = build_model(cfg, sd0)
        e2 = CLMethod["ewc"](reg_lambda=1.0)
        got = e2.compute_importances(m2, [to_dev(b) for b in loaders[0]])
        for k in names:
            close(got[k], f32[k], 1e-3, "fp32 importances " + k)


def test_ewc_step_with_reference_fisher_and_trainer_hook_order():
    """End to end at bf16-importance level: update() then one Trainer.step() whose loss is CE + penalty."""
    import types
    from mafed_amd import CLMethod, Trainer
    cfg, g, sd0, sd1, loaders, batch, syn = ewc_setup()
    model = build_model(cfg, sd0)
    ewc = CLMethod["ewc"](reg_lambda=float(g["reg_lambda"]))
    ewc.update(model=model, dataloader=[to_dev(b) for b in loaders[0]])
    model.load_state_dict(sd1)
    ce = model(**to_dev(batch), compute_loss=True, return_dict=True).loss
    total = ewc.compute_loss(model, ce, batch=batch)
    close(total, float(g["step/total"]), 2e-2, "ce + penalty (reference Fisher)")
    conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=4, grad_norm=2.0, learning_rate=5e-5, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_perc=0.1)
    tr = Trainer(model, ewc, conf, task_id=1, n_batches_per_epoch=10, ddp=False)
    before = model.flat_params.clone()
    rec = tr.step(to_dev(batch), 0)
    assert rec["branch"] == "task" and rec["stepped"]
    close(rec["loss"], float(total), 1e-5, "Trainer.step loss = CE + EWC penalty")
    assert torch.equal(before, model.flat_params)  # warm-up step 0 has lr = 0
    rec = tr.step(to_dev(batch), 1)
    assert rec["stepped"] and not torch.equal(before, model.flat_params)
