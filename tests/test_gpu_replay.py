"""The replay memory on the GPU, in the configuration bench.py times: ``HBMReplayBuffer`` gathering one draw ahead on its loader
stream, ``last_ready_event`` ordering the frozen-teacher forward, the per-layer early distillation sums on the teacher's stream,
``Trainer(pipeline_optimizer=True)`` -- checked against the SAME optimiser steps fed from a plain list on one stream.

  * ``CLMethod["replay"]`` (ER, mafed/methods/replay.py:68-72): plain CE on a memory batch
  * ``CLMethod["featdistill"]`` (MAFED) with the memory built by ``_update_memory`` (mafed/methods/distillation.py:182-209)

A twin buffer with the same seed reproduces the draws (a fresh permutation's first batch per call), so every step of the two
runs sees identical samples; any stream-ordering bug on the buffered path shows up as a mismatch in loss / grad-norm / parameters.
"""
import types

import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import TINY, tiny_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, sd, dtype):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    mc = VLPythiaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        vision_hidden_size=cfg.vision_hidden_size, num_vision_tokens=cfg.num_vision_tokens)
    m = VLPythiaForCausalLM(mc, compute_dtype=dtype, device=DEV)
    m.load_state_dict(sd, strict=True)
    return m


def _dataset(cfg, n, T, seed):
    """A finished task's collated samples (what ``update(dataset=...)`` receives): left-padded text, labels on the answer."""
    parts = [R.make_batch(cfg, 4, T, seed=seed + i, pad=True, n_answer=3) for i in range(n // 4)]
    return {k: torch.cat([p[k] for p in parts], 0) for k in ("input_ids", "attention_mask", "labels", "patch_embeddings")}


def _conf(lr=1e-3, accumulate=1):
    return types.SimpleNamespace(accumulate_grad_batches=accumulate, replay_interval=1, grad_norm=2.0, learning_rate=lr, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_steps=0, total_steps=100)


def _run(method, model, loader, n_steps, pipeline, task_batch):
    """``pipeline`` = the bench configuration: pipelined optimiser AND the clip norm from per-range partials launched by the
    backward's hooks; the comparison run uses the one-pass norm at the start of the optimiser step."""
    from mafed_amd import Trainer, ops
    tr = Trainer(model, method, _conf(), task_id=1, pipeline_optimizer=pipeline, incremental_norm=pipeline)
    calls = {"finish": 0, "onepass": 0}
    fin, one = ops.gradnorm_finish, ops.gradnorm_clip
    ops.gradnorm_finish = lambda *a, **k: (calls.__setitem__("finish", calls["finish"] + 1), fin(*a, **k))[1]
    ops.gradnorm_clip = lambda *a, **k: (calls.__setitem__("onepass", calls["onepass"] + 1), one(*a, **k))[1]
    try:
        return _run_steps(tr, method, model, loader, n_steps, task_batch, calls, pipeline)
    finally:
        ops.gradnorm_finish, ops.gradnorm_clip = fin, one


def _run_steps(tr, method, model, loader, n_steps, task_batch, calls, incremental):
    losses, gns = [], []
    for i in range(n_steps):
        if isinstance(loader, list):
            method.mem_dataloader = [dict(loader[i])]
        rec = tr.step(task_batch, i)
        assert rec["branch"] == "replay" and rec["stepped"]
        losses.append(rec["loss"])
        gns.append(rec["grad_norm"])
    tr.join()
    torch.cuda.synchronize()
    assert calls == ({"finish": n_steps, "onepass": 0} if incremental else {"finish": 0, "onepass": n_steps}), calls
    return [float(x) for x in losses], [float(x) for x in gns], model.flat_params.clone()


def _compare(a, b, tol_loss, tol_gn, tol_p):
    la, ga, pa = a
    lb, gb, pb = b
    for i, (x, y) in enumerate(zip(la, lb)):
        assert abs(x - y) <= tol_loss * max(1.0, abs(y)), f"step {i}: loss {x} vs {y}"
    for i, (x, y) in enumerate(zip(ga, gb)):
        assert abs(x - y) <= tol_gn * max(1.0, abs(y)), f"step {i}: grad norm {x} vs {y}"
    rel = float((pa - pb).norm() / pb.norm())
    assert rel <= tol_p, f"parameters after {len(la)} steps: relative difference {rel:.3e}"


@pytest.mark.parametrize("dtype,name", [(torch.float32, "t64"), (torch.bfloat16, "m64")])
def test_er_with_hbm_buffer_matches_list_fed_steps(dtype, name):
    from mafed_amd import CLMethod
    from mafed_amd.methods import HBMReplayBuffer
    cfg, t = tiny_cfg(name), TINY[name]
    sd = R.init_weights(cfg, seed=11, bias_std=0.02, ln_jitter=0.05)
    B, n_steps = t["B"], 5
    data = _dataset(cfg, 16, t["T"], seed=300)
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=B, seed=77, pin_mem=False, accumulate_grad_batches=1)
    task_batch = {k: v[:B].to(DEV) for k, v in data.items()}

    def make():
        er = CLMethod["replay"](opts=opts, memory_size=12, model_type="vlpythia")
        model = _model(cfg, sd, dtype)
        er.update(dataset=data, model=model)
        assert er.task_id == 1 and isinstance(er.mem_dataloader, HBMReplayBuffer) and len(er.mem_dataloader) == 12
        return er, model

    er1, m1 = make()
    a = _run(er1, m1, er1.mem_dataloader, n_steps, pipeline=True, task_batch=task_batch)
    er2, m2 = make()
    twin = er2.mem_dataloader  # same seed, same samples: its draws are the batches run 1 consumed
    batches = [twin._draw() for _ in range(n_steps)]
    torch.cuda.synchronize()
    b = _run(er2, m2, batches, n_steps, pipeline=False, task_batch=task_batch)
    tol = (1e-6, 1e-5, 1e-6) if dtype == torch.float32 else (2e-3, 2e-2, 2e-3)
    _compare(a, b, *tol)


def _fd_setup(cfg, sd, tsd, dtype, B, data, seed=91):
    from mafed_amd import CLMethod
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=B, seed=seed, pin_mem=False, accumulate_grad_batches=1)
    fd = CLMethod["featdistill"](memory_size=12, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                                 distillation_modality_weighing_strategy="equal", distillation_layer_weighing_strategy="discounted",
                                 gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
    fd._update_model(_model(cfg, tsd, dtype))
    fd._update_memory(data)          # -> HBMReplayBuffer on the teacher's device
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    return fd, _model(cfg, sd, dtype)


@pytest.mark.parametrize("dtype,name", [(torch.float32, "t64"), (torch.bfloat16, "m64")])
def test_mafed_with_hbm_buffer_matches_list_fed_steps(dtype, name):
    from mafed_amd.methods import HBMReplayBuffer
    cfg, t = tiny_cfg(name), TINY[name]
    sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=22, std=5e-3)
    B, n_steps = t["B"], 5
    data = _dataset(cfg, 16, t["T"], seed=400)
    task_batch = {k: v[:B].to(DEV) for k, v in data.items()}
    fd1, m1 = _fd_setup(cfg, sd, tsd, dtype, B, data)
    assert isinstance(fd1.mem_dataloader, HBMReplayBuffer) and len(fd1.mem_dataloader) == 12
    a = _run(fd1, m1, fd1.mem_dataloader, n_steps, pipeline=True, task_batch=task_batch)
    assert fd1.mem_dataloader.last_ready_event is not None      # the loader-stream path really ran
    per_layer_a = fd1.last_layer_losses.clone()
    fd2, m2 = _fd_setup(cfg, sd, tsd, dtype, B, data)
    batches = [fd2.mem_dataloader._draw() for _ in range(n_steps)]
    torch.cuda.synchronize()
    fd2.overlap_teacher = False     # one stream: teacher forward, sums and student in program order
    m2.overlap_param_grads = False
    b = _run(fd2, m2, batches, n_steps, pipeline=False, task_batch=task_batch)
    tol = (1e-6, 1e-5, 1e-6) if dtype == torch.float32 else (2e-3, 2e-2, 2e-3)
    _compare(a, b, *tol)
    rel = float(((per_layer_a - fd2.last_layer_losses).abs() / fd2.last_layer_losses.abs().clamp_min(1e-12)).max())
    assert rel <= (1e-5 if dtype == torch.float32 else 2e-2), f"per-layer distillation losses of the last step: {rel:.3e}"


@pytest.mark.timeout(600)
def test_mafed_with_hbm_buffer_at_bench_size():
    """The bench configuration itself (VLPythia-410M, B = 32, 256 + 32 tokens, bf16, pipelined optimiser, buffer prefetch,
    teacher stream, early sums) against the single-stream list-fed run: three optimiser steps."""
    from mafed_amd import FeatureDistillation, VLPythiaConfig, VLPythiaForCausalLM
    from mafed_amd.methods import HBMReplayBuffer
    B, P, T, n_steps = 32, 256, 32, 3
    cfg = VLPythiaConfig.preset("410m", num_vision_tokens=P)

    def make():
        student = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=DEV, seed=1234)
        opts = types.SimpleNamespace(tasks=["t0", "t1"], batch_size=B, seed=1236, pin_mem=False, accumulate_grad_batches=1)
        fd = FeatureDistillation(memory_size=4000, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                                 distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                                 gamma=0.5, distillation_layer=None, distillation_coeff=1.0, replay_coeff=1.0)
        fd._update_model(student)
        g = torch.Generator(device=DEV).manual_seed(1237)
        fd.past_model.flat_params.add_(torch.randn(fd.past_model.flat_params.shape, generator=g, device=DEV) * 1e-3)
        fd.past_model._shadow_dirty = True
        fd.task_id = 1
        fd.num_vision_tokens = P
        gc = torch.Generator().manual_seed(1235)
        n_mem = 4 * B
        ids = torch.randint(1, cfg.vocab_size, (n_mem, T), generator=gc)
        labels = torch.full((n_mem, T), -100, dtype=torch.int64)
        labels[:, -4:] = ids[:, -4:]
        mem = HBMReplayBuffer(B, DEV, seed=1236)
        mem.add({"input_ids": ids, "attention_mask": torch.ones(n_mem, T, dtype=torch.int64), "labels": labels,
                 "patch_embeddings": torch.randn(n_mem, P, cfg.vision_hidden_size, generator=gc)})
        fd.mem_dataloader = mem
        return fd, student, mem

    fd1, m1, mem1 = make()
    task_batch = mem1._draw()   # dropped by a replay step; consumes one draw on both sides
    a = _run(fd1, m1, mem1, n_steps, pipeline=True, task_batch=task_batch)
    del fd1, m1, mem1
    torch.cuda.empty_cache()
    fd2, m2, mem2 = make()
    mem2._draw()
    batches = [mem2._draw() for _ in range(n_steps)]
    fd2.overlap_teacher = False
    m2.overlap_param_grads = False
    b = _run(fd2, m2, batches, n_steps, pipeline=False, task_batch=task_batch)
    # bf16 backward noise floor (fp32 split-K atomics): 2e-3 relative gradient norm per step
    _compare(a, b, 2e-3, 2e-2, 1e-3)


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2)])
def test_teacher_cache_steps_are_bit_identical_to_the_teacher_forward(rank, world):
    """``build_teacher_cache``: the frozen teacher's distilled hidden states of the whole replay memory, written once by the same kernels
    on the same batch shapes; a replay step then gathers its rows instead of running the teacher.  The gathered states equal a fresh
    teacher forward on the same samples bit for bit (any batch, including the ragged tail of the memory); losses, gradient norms and
    parameters of cached steps equal the uncached run (to the run-to-run noise of the backward's fp32 atomics), the teacher forward does
    not run, and changing the teacher drops the cache."""
    from mafed_amd import FeatureDistillation, Trainer
    from mafed_amd.methods import HBMReplayBuffer
    cfg = tiny_cfg("m64")
    t = TINY["m64"]
    B, T = t["B"], t["T"]
    n_mem = 5 * B + 3                       # a ragged tail: the fill re-runs the last full batch
    sd = R.init_weights(cfg, seed=31, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=32, std=5e-3)
    data = R.make_batch(cfg, n_mem, T, seed=33, pad=True, n_answer=3)
    data["patch_embeddings"] = data["patch_embeddings"].to(torch.bfloat16).float()   # the buffer stores bf16 features

    def run(cached):
        model, teacher = _model(cfg, sd, torch.float32), _model(cfg, tsd, torch.float32)
        opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=B, seed=3, pin_mem=False, accumulate_grad_batches=1)
        fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                                 distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                                 gamma=0.5, distillation_layer=None)
        fd._update_model(teacher)
        fd.task_id = 1
        fd.num_vision_tokens = cfg.num_vision_tokens
        mem = HBMReplayBuffer(B, DEV, seed=9, rank=rank, world_size=world)   # (rank 1 of 2: the cache covers this rank's shard only)
        mem.add(data)
        fd.mem_dataloader = mem
        lo, hi = mem.shard()
        calls = []
        if cached:
            info = fd.build_teacher_cache()
            assert info["samples"] == hi - lo and mem.attach_index
            layers = fd.loss_weights.get_distillation_layers()
            for idx in (torch.arange(lo, lo + B), torch.arange(hi - B, hi), torch.tensor([lo, hi - 1, lo + 7, lo + B, lo + 2 * B + 1, lo + 3, hi - 2, lo + 11][:B])):
                idx = idx.to(DEV)
                fd._mem_index = idx
                got = fd._cached_teacher_states(max(layers) + 1)
                want = fd.past_model.hidden_states_upto(mem.data["input_ids"][idx], mem.data["attention_mask"][idx],
                                                        patch_embeddings=mem.data["patch_embeddings"][idx], n_hidden=max(layers) + 1)
                for l in layers:
                    assert torch.equal(got[l], want[l].view_as(got[l])), f"cached teacher state of layer {l} differs from the forward"
            fd._mem_index = None
            orig = fd.past_model.hidden_states_upto
            fd.past_model.hidden_states_upto = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        tr = Trainer(model, fd, _conf(lr=1e-3), task_id=1, pipeline_optimizer=True)
        task = {k: v[:B].to(DEV) for k, v in data.items()}
        losses, gns = [], []
        for i in range(4):
            rec = tr.step(task, i)
            losses.append(rec["loss"]); gns.append(rec["grad_norm"])
        tr.join()
        torch.cuda.synchronize()
        assert not calls, "the teacher forward ran although its states are cached"
        out = ([float(x) for x in losses], [float(x) for x in gns], model.flat_params.clone())
        if cached:
            fd._update_model(model)
            assert fd._tcache is None and not mem.attach_index, "a new teacher must drop the cached states"
        return out

    la, ga, pa = run(False)
    lb, gb, pb = run(True)
    assert la[0] == lb[0] and ga[0] == gb[0], (la, lb, ga, gb)     # same weights, same batch, same teacher bits
    assert all(abs(a - b) <= 1e-6 * abs(a) for a, b in zip(la, lb)) and all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(ga, gb)), (la, lb, ga, gb)
    assert float((pa - pb).abs().max()) <= 1e-6
    assert len(set(la)) == 4
