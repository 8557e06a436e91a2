"""Overwrite-mode weight gradients: on the first micro-batch of an accumulation window the grouped weight-gradient
GEMMs WRITE the layers' matrix gradients (beta = 0) and the AdamW pass does not zero them (mafed_adamw_step_partial_zero); later
micro-batches accumulate.  The steps must equal the classic "zero, then accumulate" steps -- in both optimiser pipeline modes, with
accumulate_grad_batches 1 and 4, when a layer's backward is skipped, and when a foreign backward accumulates into un-zeroed matrices
(reference semantics: optimizer.zero_grad() + AdamW.step, mafed/optim/adamw.py:50-113, mafed/train.py:288)."""
import types

import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import tiny_cfg
from tests.test_gpu_replay import _conf, _dataset, _model

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mafed(cfg, teacher, mem_batches, P):
    from mafed_amd import FeatureDistillation
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=4, seed=1, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=16, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None)
    fd._update_model(teacher)
    fd.task_id = 1
    fd.num_vision_tokens = P
    return fd


def _steps(name, overwrite, pipeline, accumulate, n_micro, replay_interval=1):
    from mafed_amd import Trainer
    cfg = tiny_cfg(name)
    sd = R.init_weights(cfg, seed=3, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=4, std=5e-3)
    student, teacher = _model(cfg, sd, torch.bfloat16), _model(cfg, tsd, torch.bfloat16)
    student.dw_group_layers = 2
    data = _dataset(cfg, 4 * n_micro, 6, seed=50)
    batches = [{k: v[4 * i: 4 * i + 4].to(DEV) for k, v in data.items()} for i in range(n_micro)]
    fd = _mafed(cfg, teacher, batches, cfg.num_vision_tokens)
    conf = _conf(lr=1e-3, accumulate=accumulate)
    conf.replay_interval = replay_interval
    tr = Trainer(student, fd, conf, task_id=1, pipeline_optimizer=pipeline, overwrite_weight_grads=overwrite)
    assert tr._overwrite_ok() == overwrite
    losses, gns = [], []
    for i in range(n_micro):
        fd.mem_dataloader = [dict(batches[i])]
        rec = tr.step(dict(batches[(i + 1) % n_micro]), i)
        losses.append(float(rec["loss"]))
        if rec["stepped"]:
            gns.append(float(rec["grad_norm"]))
    tr.join()
    torch.cuda.synchronize()
    return losses, gns, student.flat_params.clone(), student


@pytest.mark.parametrize("pipeline", [False, True])
@pytest.mark.parametrize("accumulate,replay_interval", [(1, 1), (4, 4), (2, 1)])
def test_overwrite_equals_zero_then_accumulate(pipeline, accumulate, replay_interval):
    a = _steps("t64", True, pipeline, accumulate, 8, replay_interval)
    b = _steps("t64", False, pipeline, accumulate, 8, replay_interval)
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), f"micro-batch {i}: loss {x} vs {y}"
    assert len(a[1]) == 8 // accumulate
    for i, (x, y) in enumerate(zip(a[1], b[1])):
        assert abs(x - y) <= 2e-3 * max(1.0, abs(y)), f"step {i}: grad norm {x} vs {y}"
    d = float((a[2] - b[2]).abs().max())
    assert d <= 2e-5, f"parameters differ by {d} after {8 // accumulate} optimiser steps"
    assert a[3]._dw_stale and not b[3]._dw_stale   # the overwrite run really left the matrices to the next sweep


def test_a_foreign_backward_after_an_overwrite_step_starts_from_zero():
    """After an overwrite-mode optimiser step the matrices' gradient buffer holds the last gradient; a backward that accumulates
    (here: a plain loss.backward() outside the Trainer) must not add to it."""
    _, _, _, student = _steps("t64", True, True, 1, 2)
    cfg = tiny_cfg("t64")
    batch = {k: v.to(DEV) for k, v in R.make_batch(cfg, 4, 6, seed=77, pad=True, n_answer=3).items()}
    assert student._dw_stale
    out = student(**batch, return_dict=True)
    out.loss.backward()
    g1 = student.flat_grads.clone()
    student.zero_grad()
    out = student(**batch, return_dict=True)
    out.loss.backward()
    torch.cuda.synchronize()
    lo, hi = student.layer_matrix_range(0)
    ref = student.flat_grads[lo:hi]
    assert float(ref.abs().max()) > 0
    assert float((g1[lo:hi] - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


def test_partial_zero_adamw_kernel():
    """mafed_adamw_step_partial_zero: same update as mafed_adamw_step, g zeroed on [0, zero_n) only; a skipped step (clip scale < 0)
    still zeroes that part and nothing else."""
    from mafed_amd import ops
    n, zn = 4096 + 8, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    p0 = torch.randn(n, device=DEV, generator=g)
    gr = torch.randn(n, device=DEV, generator=g)
    m0, v0 = torch.rand(n, device=DEV, generator=g) * 0.1, torch.rand(n, device=DEV, generator=g) * 0.01
    lr_dev = torch.tensor([1e-2, 0.1, 0.3], device=DEV)
    clip = torch.tensor([3.0, 0.5], device=DEV)

    def run(zero_n, clip_t):
        p, gg, m, v = p0.clone(), gr.clone(), m0.clone(), v0.clone()
        sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
        ops.adamw_step_(p, gg, m, v, lr_dev, 0.9, 0.98, 1e-6, 0.01, 0, clip_t, 1.0, sh, zero_grad=zero_n is not None, zero_n=zero_n)
        return p, gg, m, v, sh

    ref = run(None, clip)
    out = run(zn, clip)
    for a, b in zip(ref[:1] + ref[2:], out[:1] + out[2:]):
        assert torch.equal(a, b)
    assert float(out[1][:zn].abs().max()) == 0.0 and torch.equal(out[1][zn:], gr[zn:])
    skip = run(zn, torch.tensor([float("nan"), -1.0], device=DEV))
    assert torch.equal(skip[0], p0) and torch.equal(skip[2], m0) and torch.equal(skip[3], v0)
    assert float(skip[1][:zn].abs().max()) == 0.0 and torch.equal(skip[1][zn:], gr[zn:])


def test_non_finite_gradient_norm_skips_the_update():
    """a NaN loss (e.g. the row-sparse head's overflow poison) must not reach the parameters: the norm kernels publish clip
    scale -1, AdamW leaves p / m / v / shadow alone, the step counter does not advance, the gradient buffer is zeroed for the next
    window -- and the host sees the non-finite norm in the step's record."""
    from mafed_amd import Trainer
    from mafed_amd.methods import Naive
    cfg = tiny_cfg("t64")
    sd = R.init_weights(cfg, seed=3, bias_std=0.02, ln_jitter=0.05)
    for pipeline in (False, True):
        student = _model(cfg, sd, torch.bfloat16)
        tr = Trainer(student, Naive(), _conf(lr=1e-3), task_id=0, pipeline_optimizer=pipeline)
        batch = {k: v.to(DEV) for k, v in R.make_batch(cfg, 4, 6, seed=5, pad=True, n_answer=3).items()}
        tr.step(dict(batch), 0)
        tr.join()
        torch.cuda.synchronize()
        p1, m1, t1 = student.flat_params.clone(), tr.optimizer.exp_avg.clone(), int(tr.optimizer.state_dev)
        assert t1 == 1
        hook = student.register_forward_hook(lambda mod, a, out: setattr(out, "loss", out.loss * float("nan")) or out)
        rec = tr.step(dict(batch), 1)
        hook.remove()
        tr.join()
        torch.cuda.synchronize()
        assert not torch.isfinite(rec["grad_norm"]).item()
        assert torch.equal(student.flat_params, p1) and torch.equal(tr.optimizer.exp_avg, m1)
        assert int(tr.optimizer.state_dev) == 1, "the device step counter advanced on a skipped step"
        lo, hi = student.layer_matrix_range(0)
        bias_seg = student.flat_grads[student.decay_split():]
        assert float(bias_seg.abs().max()) == 0.0      # zeroed for the next window (the matrices are overwritten by it instead)
        rec = tr.step(dict(batch), 2)                  # and training goes on
        tr.join()
        torch.cuda.synchronize()
        assert torch.isfinite(rec["grad_norm"]).item() and int(tr.optimizer.state_dev) == 2
        assert torch.isfinite(student.flat_params).all()


# ---- squares of the weight gradients from the GEMM epilogue (mafed_gemm_problem.sumsq) ---------------------------------------------
@pytest.mark.parametrize("shape", [(1024, 1024, 4608), (3072, 1024, 2304), (192, 64, 96), (2048, 2048, 2304), (768, 2304, 2304)])
@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_grouped_gemm_leaves_the_squares_of_c(shape, beta):
    """sum of the 16 slots += sum C^2 of the stored C: fused into the persistent weight-gradient kernel where the shapes tile it
    (first two), a pass over C behind the product otherwise (third; and the 256 x 256-tile kernel's shapes, h = 2048 / 768, where the
    group must still go out as ONE persistent launch -- the regression this guards sent it back to one launch per product)."""
    from mafed_amd import ops, _lib
    M, N, K = shape
    launches0 = _lib.load().mafed_gemm_pp_launches()
    g = torch.Generator(device=DEV).manual_seed(5)
    probs = []
    big = M * N >= 768 * 2304      # the 256 x 256-tile kernel's shapes: eight products fill the chip (one persistent launch)
    for i in range(8 if big else 2):
        A = (torch.randn(K, M, device=DEV, generator=g) * 0.3).to(torch.bfloat16)
        B = (torch.randn(K, N, device=DEV, generator=g) * 0.3).to(torch.bfloat16)
        out = torch.randn(M, N, device=DEV, generator=g)
        probs.append(dict(A=A, B=B, out=out, beta=beta, sumsq=torch.full((16,), 0.5 * i, device=DEV)))
    ops.gemm_grouped(probs, True, False)
    torch.cuda.synchronize()
    if big and shape != (3072, 1024, 2304):
        assert _lib.load().mafed_gemm_pp_launches() - launches0 == 1, "the group left the persistent path"
    for i, q in enumerate(probs):
        want = float((q["out"].double() ** 2).sum()) + 16 * 0.5 * i
        got = float(q["sumsq"].double().sum())
        assert abs(got - want) <= 1e-5 * want, (shape, beta, i, got, want)


@pytest.mark.timeout(600)
def test_fused_norm_squares_give_the_same_clip_norm_at_410m():
    """Trainer(fused_norm_squares) at the benchmark's size and configuration: the norm the clip sees == the norm of the range partials
    == the one-pass norm, step by step."""
    from mafed_amd import FeatureDistillation, Trainer, VLPythiaConfig, VLPythiaForCausalLM
    from mafed_amd.methods import HBMReplayBuffer
    B, P, T = 32, 256, 32
    cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)
    gcpu = torch.Generator().manual_seed(1235)
    ids = torch.randint(1, cfg.vocab_size, (64, T), generator=gcpu)
    labels = torch.full((64, T), -100, dtype=torch.int64)
    labels[:, -4:] = ids[:, -4:]
    feats = torch.randn(64, P, cfg.vision_hidden_size, generator=gcpu).to(torch.bfloat16)
    samples = {"input_ids": ids, "attention_mask": torch.ones(64, T, dtype=torch.int64), "labels": labels, "patch_embeddings": feats}
    norms = {}
    for mode in ("fused", "partials", "onepass"):
        student = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=DEV, seed=1234)
        opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
        fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                                 distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                                 gamma=0.5, distillation_layer=None)
        fd._update_model(student)
        gen = torch.Generator(device=DEV).manual_seed(1237)
        fd.past_model.flat_params.add_(torch.randn(fd.past_model.flat_params.shape, generator=gen, device=DEV) * 1e-3)
        fd.past_model._shadow_dirty = True
        fd.task_id, fd.num_vision_tokens = 1, P
        mem = HBMReplayBuffer(B, torch.device(DEV), seed=9)
        mem.add(samples)
        fd.mem_dataloader = mem
        tr = Trainer(student, fd, _conf(lr=5e-5), task_id=1, pipeline_optimizer=True, incremental_norm=mode != "onepass")
        tr.fused_norm_squares = mode == "fused"
        task_batch = mem.sample()
        norms[mode] = [float(tr.step(task_batch, i)["grad_norm"]) for i in range(3)]
        tr.join()
        torch.cuda.synchronize()
        if mode == "fused":
            assert student.dw_sumsq is not None and student._dw_sumsq_used == student._bw_serial
        del tr, fd, student, mem
        torch.cuda.empty_cache()
    for a, b, c in zip(norms["fused"], norms["partials"], norms["onepass"]):
        assert abs(a - c) <= 2e-3 * c and abs(b - c) <= 2e-3 * c, norms   # (2e-3: the run-to-run noise of a bf16 backward's atomics)
    assert abs(norms["fused"][0] - norms["partials"][0]) <= 2e-3 * norms["partials"][0]


def test_fused_squares_are_only_promised_where_the_launch_emits_them():
    """Host query behind model._dw_group_fuses_squares: the 410M group (two layers, 128 x 256 tiles) fuses the squares; the h = 768 / 2048
    groups run on the 256 x 256-tile kernel, which does not -- there the norm hook keeps its range pass (the regression this guards cost the
    160M / 1.4B-shape steps 15 - 25 %)."""
    from mafed_amd import ops
    rows = 32 * 288
    layer = lambda h: [(3 * h, h, rows), (h, h, rows), (4 * h, h, rows), (h, 4 * h, rows)]
    assert ops.gemm_grouped_fuses_sumsq(layer(1024) * 2, True, False)
    assert not ops.gemm_grouped_fuses_sumsq(layer(2048) * 2, True, False)
    assert not ops.gemm_grouped_fuses_sumsq(layer(768) * 2, True, False)
    assert not ops.gemm_grouped_fuses_sumsq(layer(1024)[:1], True, False)      # one small product does not fill the chip
