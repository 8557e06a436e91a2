"""Parity at BASELINE.json's FULL sizes -- configs[1] VLPythia-160M, configs[2] VLPythia-410M (the configuration the headline
metric is quoted on) and the single-GPU shape of configs[4] (Pythia-1.4B dims: h 2048, 16 heads of 128, the D = 128 resident
attention kernels, 2048-wide LayerNorm rows), each with B = 32, 256 image + 32 text tokens, bf16 -- where the fp32 CPU oracle
would need minutes per step: size-independent properties of the MAFED step instead of element-wise comparison.

  * teacher == student  =>  every per-layer, per-modality MSE is exactly 0 (both forwards run the same kernels on the
    same bits) and the step's loss is the replay CE alone
  * batch linearity: with equal token counts per sample the full-batch gradient is the mean of the two half-batch
    gradients (CE is normalised per sample, the masked MSE per token)
  * bf16 (MFMA) mode against the exact-fp32 kernels of the same library on the same weights and batch
  * clip + AdamW invariants: post-clip global norm <= max_norm, lr = 0 leaves the parameters untouched, the bf16 shadow
    weights are the rounded fp32 weights
  * left-padding a sample changes nothing at its valid positions
"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, P, T = 32, 256, 32


PRESETS = ["160m", "410m", "1.4b"]


def _setup(dtype=torch.bfloat16, perturb_teacher=1e-3, seed=1234, preset="410m"):
    from mafed_amd import FeatureDistillation, VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig.preset(preset, num_vision_tokens=P)
    student = VLPythiaForCausalLM(cfg, compute_dtype=dtype, device=DEV, seed=seed)
    opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
    fd._update_model(student)
    if perturb_teacher:
        g = torch.Generator(device=DEV).manual_seed(1237)
        fd.past_model.flat_params.add_(torch.randn(fd.past_model.flat_params.shape, generator=g, device=DEV) * perturb_teacher)
        fd.past_model._shadow_dirty = True
    fd.task_id = 1
    fd.num_vision_tokens = P
    return cfg, student, fd


def _batch(cfg, n=B, seed=1235, pad=None):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.vocab_size, (n, T), generator=g)
    labels = torch.full((n, T), -100, dtype=torch.int64)
    labels[:, -4:] = ids[:, -4:]
    am = torch.ones(n, T, dtype=torch.int64)
    if pad is not None:
        for b, k in pad.items():
            am[b, :k] = 0
            ids[b, :k] = 0
            labels[b, :k] = -100
    feats = torch.randn(n, P, cfg.vision_hidden_size, generator=g).to(torch.bfloat16)
    return {"input_ids": ids.to(DEV), "attention_mask": am.to(DEV), "labels": labels.to(DEV), "patch_embeddings": feats.to(DEV)}


def _replay_grads(student, fd, batch):
    fd.mem_dataloader = [dict(batch)]
    student.zero_grad()
    loss, n = fd.replay(student)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), student.flat_grads.clone()


@pytest.fixture(autouse=True)
def _release_hbm():
    yield
    import gc
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("preset", PRESETS)
def test_teacher_equal_to_student_gives_zero_distillation(preset):
    cfg, student, fd = _setup(perturb_teacher=0.0, preset=preset)
    batch = _batch(cfg)
    loss, grads = _replay_grads(student, fd, batch)
    assert float(fd.last_layer_losses.abs().max()) == 0.0, "identical teacher: every per-layer loss must be exactly 0"
    ce = float(student(**batch, compute_loss=True, return_dict=True).loss)
    assert abs(loss - ce) <= 1e-6 * max(1.0, abs(ce)), (loss, ce)
    # and the gradient is the CE gradient alone
    student.zero_grad()
    student(**batch, compute_loss=True, return_dict=True).loss.backward()
    torch.cuda.synchronize()
    ref = student.flat_grads
    # run-to-run noise floor of a bf16 backward is ~2e-3 in relative norm: the split-K atomics of the LM-head dX differ in
    # the last fp32 bit, which flips single bf16 roundings of the gradient stream 24 layers deep
    rel = float((grads - ref).norm() / ref.norm())
    assert rel <= 1e-2, f"|g(replay, identical teacher) - g(CE)| / |g| = {rel:.3e}"


@pytest.mark.parametrize("preset", PRESETS)
def test_full_batch_gradient_is_mean_of_half_batches(preset):
    cfg, student, fd = _setup(preset=preset)
    batch = _batch(cfg)
    loss, g_full = _replay_grads(student, fd, batch)
    halves = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        hb = {k: v[sl].contiguous() for k, v in batch.items()}
        halves.append(_replay_grads(student, fd, hb))
    loss_h = 0.5 * (halves[0][0] + halves[1][0])
    assert abs(loss - loss_h) <= 2e-3 * abs(loss), (loss, loss_h)
    g_mean = 0.5 * (halves[0][1] + halves[1][1])
    num = float((g_full - g_mean).norm())
    den = float(g_full.norm())
    assert num <= 2e-2 * den, f"batch linearity: |g32 - mean(g16, g16)| / |g32| = {num / den:.3e}"


@pytest.mark.timeout(900)
@pytest.mark.parametrize("preset", PRESETS)
def test_bf16_step_tracks_exact_fp32_kernels_at_full_size(preset):
    cfg, s16, fd16 = _setup(dtype=torch.bfloat16, preset=preset)
    batch = _batch(cfg)
    l16, g16 = _replay_grads(s16, fd16, batch)
    per16 = fd16.last_layer_losses.clone()
    del s16, fd16
    torch.cuda.empty_cache()
    cfg, s32, fd32 = _setup(dtype=torch.float32, preset=preset)
    l32, g32 = _replay_grads(s32, fd32, {k: (v.float() if v.is_floating_point() else v) for k, v in batch.items()})
    per32 = fd32.last_layer_losses
    assert abs(l16 - l32) <= 1e-2 * abs(l32), (l16, l32)
    rel = float(((per16 - per32).abs() / per32.abs().clamp_min(1e-12)).max())
    assert rel <= 5e-2, f"per-layer distillation losses bf16 vs fp32: {rel:.3e}"
    gn16, gn32 = float(g16.norm()), float(g32.norm())
    assert abs(gn16 - gn32) <= 3e-2 * gn32, (gn16, gn32)
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    assert cos >= 0.995, f"gradient direction bf16 vs fp32: cos = {cos:.5f}"


def test_clip_and_adamw_invariants_at_full_size():
    from mafed_amd import Trainer
    cfg, student, fd = _setup()
    batch = _batch(cfg)
    fd.mem_dataloader = [dict(batch)]
    conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=0.0, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_perc=0.1)
    tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=100, ddp=False)
    before = student.flat_params.clone()
    # one step with the gradients kept for inspection: reproduce the clip by hand
    student.zero_grad()
    loss, _ = fd.replay(student)
    loss.backward()
    gn = float(student.flat_grads.norm())
    out = tr.optimizer.clip_grad_norm_(2.0)   # -> global norm; the clip scale stays on the device for the AdamW kernel
    torch.cuda.synchronize()
    assert abs(float(out) - gn) <= 1e-4 * gn, (float(out), gn)
    scale = float(tr.optimizer.clip_out[1])
    assert scale <= 1.0 and scale * gn <= 2.0 * (1 + 1e-4), (scale, gn)
    assert gn <= 2.0 or abs(scale - 2.0 / (gn + 1e-6)) <= 1e-5, (scale, gn)
    tr.optimizer.advance()
    tr.optimizer.apply()
    torch.cuda.synchronize()
    assert torch.equal(student.flat_params, before), "lr = 0 (warm-up step 0) must leave the fp32 parameters untouched"
    w = student._p("gpt_neox.layers.3.mlp.dense_h_to_4h.weight")
    wl = student._w("gpt_neox.layers.3.mlp.dense_h_to_4h.weight")
    assert torch.equal(wl, w.to(torch.bfloat16)), "bf16 shadow weights = RNE(fp32 weights)"


def test_left_padding_is_invisible_at_valid_positions():
    cfg, student, fd = _setup()
    batch = _batch(cfg, n=4)
    k = 7
    padded = {kk: v.clone() for kk, v in batch.items()}
    # sample 1: shift its text right by k and left-pad; the last T-k tokens of the original become the valid suffix
    for key, fill in (("input_ids", 0), ("labels", -100), ("attention_mask", 0)):
        padded[key][1, k:] = batch[key][1, : T - k]
        padded[key][1, :k] = fill
    with torch.no_grad():
        a = student(**batch, output_hidden_states=True, return_dict=True)
        b = student(**padded, output_hidden_states=True, return_dict=True)
    torch.cuda.synchronize()
    # untouched samples are bit-identical; the shifted sample's image rows are identical (text sits after them, causal)
    for l in (0, 5, cfg.num_hidden_layers - 1):
        assert torch.equal(a.hidden_states[l][0], b.hidden_states[l][0])
        assert torch.equal(a.hidden_states[l][1, :P], b.hidden_states[l][1, :P])
    # its text rows differ only through the rotary position (k later): same tokens, same causal context => close, not equal
    assert b.logits.shape == a.logits.shape
    assert torch.isfinite(b.logits).all()


def test_configs4_step_fed_by_the_resident_replay_memory():
    """BASELINE.json configs[4] on one GPU: VLPythia-1.4B + MAFED with the experience-replay memory (4000 samples, bf16 patch features
    resident in HBM, gathered one draw ahead on the loader stream) feeding Trainer.step().  The buffer's draws are a seeded permutation:
    the batches it hands out are exactly the rows of that permutation (integer comparison), distinct per step, and the same steps fed
    with hand-gathered batches of the same indices give the same first loss bit for bit and the same trajectory up to the run-to-run
    noise of the fp32 atomics (the prefetch stream, its events and the attached label-row hint change nothing)."""
    from mafed_amd import Trainer
    from mafed_amd.methods import HBMReplayBuffer
    n_mem, seed, steps = 4000, 77, 3

    def samples(cfg):
        ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=torch.Generator().manual_seed(5))
        labels = torch.full((n_mem, T), -100, dtype=torch.int64)
        labels[:, -4:] = ids[:, -4:]
        feats = torch.empty(n_mem, P, cfg.vision_hidden_size, dtype=torch.bfloat16, device=DEV)
        gd = torch.Generator(device=DEV).manual_seed(6)
        for lo in range(0, n_mem, 500):
            feats[lo:lo + 500] = torch.randn(500, P, cfg.vision_hidden_size, generator=gd, device=DEV).to(torch.bfloat16)
        return {"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels, "patch_embeddings": feats}

    def run(use_buffer):
        cfg, student, fd = _setup(preset="1.4b")
        data = samples(cfg)
        conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=2e-6, betas=(0.9, 0.98),
                                     weight_decay=0.01, optim="adamw", warmup_steps=0, total_steps=100)
        mem = HBMReplayBuffer(B, DEV, seed=seed)
        mem.add(data)
        assert len(mem) == n_mem and mem.max_label_rows == 4
        resident = sum(v.numel() * v.element_size() for v in mem.data.values())
        assert resident >= n_mem * P * cfg.vision_hidden_size * 2
        handed = []
        if use_buffer:
            class Spy:   # the loader protocol of HBMReplayBuffer, recording what was handed out
                @property
                def last_ready_event(self):
                    return mem.last_ready_event
                max_label_rows, attach_label_hint = mem.max_label_rows, mem.attach_label_hint
                def __iter__(self):
                    return self
                def __next__(self):
                    b = mem.sample()
                    handed.append(b["input_ids"])
                    return b
            fd.mem_dataloader = Spy()
        else:
            gen = torch.Generator().manual_seed(seed)
            idxs = [torch.randperm(n_mem, generator=gen)[:B] for _ in range(steps + 1)]   # (+1: the buffer gathers one draw ahead)
            batches = []
            for ix in idxs:
                b = {k: v.to(DEV).index_select(0, ix.to(DEV)) for k, v in data.items()}
                b["max_label_rows"] = 4
                batches.append(b)
            it = iter(batches)

            class Feed:
                def __iter__(self):
                    return self
                def __next__(self):
                    return dict(next(it))
            fd.mem_dataloader = Feed()
        tr = Trainer(student, fd, conf, task_id=1, n_batches_per_epoch=100, pipeline_optimizer=True)
        task = {k: v[:B].to(DEV) for k, v in data.items()}
        losses = []
        for i in range(steps):
            rec = tr.step(task, i)
            losses.append(float(rec["loss"]))
        tr.join()
        torch.cuda.synchronize()
        chk = float(student.flat_params.double().sum())
        handed = [h.cpu() for h in handed]
        del tr, student, fd, mem
        return losses, chk, handed, data["input_ids"]

    la, ca, handed, all_ids = run(True)
    gen = torch.Generator().manual_seed(seed)
    want = [all_ids[torch.randperm(n_mem, generator=gen)[:B]] for _ in range(steps)]
    assert len(handed) == steps
    for h, w in zip(handed, want):
        assert torch.equal(h, w), "the buffer handed out other rows than its seeded permutation"
    assert not torch.equal(handed[0], handed[1]) and not torch.equal(handed[1], handed[2])
    import gc
    gc.collect(); torch.cuda.empty_cache()
    lb, cb, _, _ = run(False)
    assert all(l == l and l < 20.0 for l in la), la
    assert la[0] == lb[0], (la, lb)                       # same weights, same batch: the same forward bit for bit
    assert all(abs(a - b) <= 2e-4 * abs(b) for a, b in zip(la, lb)), (la, lb)
    assert abs(ca - cb) <= 2e-6 * abs(cb)   # (sum over 1.4 G parameters after three AdamW steps: atomics noise in the gradients)


def test_contended_backward_keeps_off_the_persistent_kernels():
    """Under data parallelism the Trainer flags the backward that runs beside RCCL's collectives (``model.contended_backward``): its
    GEMMs then take the 128 x 128 kernels (many small blocks) instead of the one-block-per-CU persistent ones, whose static tile
    schedule takes 1.7x as long with a few CUs occupied (tools/contention_bench.py).  Same gradients either way."""
    from mafed_amd import _lib
    lib = _lib.load()
    cfg, student, fd = _setup(preset="410m")
    batch = _batch(cfg)
    n0 = lib.mafed_gemm_pp_launches()
    loss_a, grads_a = _replay_grads(student, fd, batch)
    n_plain = lib.mafed_gemm_pp_launches() - n0
    assert n_plain > 150, "forward and backward of the 410M step run on the persistent kernels"
    student.contended_backward = True
    n1 = lib.mafed_gemm_pp_launches()
    loss_b, grads_b = _replay_grads(student, fd, batch)
    n_cont = lib.mafed_gemm_pp_launches() - n1
    student.contended_backward = False
    assert loss_a == loss_b                                   # the forward is the same launches
    assert 0 < n_cont < n_plain - 80, (n_plain, n_cont)       # forward (student + teacher) still persistent, the backward is not
    den = float(grads_a.norm())
    assert float((grads_a - grads_b).norm()) <= 1e-2 * den    # two tilings of the same bf16 products (measured 2.1e-3: bf16 roundings of dX flip)
    # the switch is scoped to the backward: the next forward is back on the persistent kernels
    n2 = lib.mafed_gemm_pp_launches()
    fd.mem_dataloader = [dict(batch)]
    fd.replay(student)
    assert lib.mafed_gemm_pp_launches() - n2 > 80


def test_contended_backward_ticketed_keeps_the_persistent_kernels_and_a_forced_variant():
    """``contended_backward = "ticketed"`` (the data-parallel default of round 4): the backward stays on the persistent kernels, its
    multi-round launches run in ticketed tile order (MAFED_EPI_TICKETED, per call), the forward does not; gradients equal the plain
    backward's.  ADVICE r3: the choice is per call -- a variant the caller forced before (here 720 = static order everywhere, and the
    tile-configuration variant) is what the hooks still say afterwards."""
    from mafed_amd import _lib
    lib = _lib.load()
    cfg, student, fd = _setup(preset="410m")
    batch = _batch(cfg)
    loss_a, grads_a = _replay_grads(student, fd, batch)
    t0, p0 = lib.mafed_gemm_get_variant(73), lib.mafed_gemm_pp_launches()
    student.contended_backward = "ticketed"
    loss_b, grads_b = _replay_grads(student, fd, batch)
    n_tk, n_pp = lib.mafed_gemm_get_variant(73) - t0, lib.mafed_gemm_pp_launches() - p0
    assert loss_a == loss_b
    assert n_pp > 150 and 30 < n_tk < 60, (n_pp, n_tk)          # the multi-round launches of the backward (24 dX of fc2 + 12 grouped dW + head), none of the forward
    den = float(grads_a.norm())
    assert float((grads_a - grads_b).norm()) <= 3e-3 * den    # same kernels' arithmetic, another tile order (atomics noise only)
    # a forced mode survives the contended backward (no process-wide switch is flipped and restored to a default)
    lib.mafed_gemm_set_variant(720)
    try:
        t1 = lib.mafed_gemm_get_variant(73)
        _replay_grads(student, fd, batch)
        assert lib.mafed_gemm_get_variant(73) == t1, "720 = static order everywhere, whatever the call asks for"
        assert lib.mafed_gemm_get_variant(72) == 720 and lib.mafed_gemm_get_variant(7) == 701
        student.contended_backward = "128x128"
        _replay_grads(student, fd, batch)
        assert lib.mafed_gemm_get_variant(72) == 720 and lib.mafed_gemm_get_variant(7) == 701
    finally:
        lib.mafed_gemm_set_variant(722)
        student.contended_backward = False
