"""GPU, two or four ranks sharing the box's one GPU over gloo (RCCL refuses two ranks on one device): the data-parallel step --
hand-scheduled backward on two streams, bucket hooks, side-stream gradient mean (all-reduce, or reduce-scatter + all-gather,
fp32 or bf16 buckets), clip, AdamW -- must equal a single process that averages the ranks' gradients itself."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import vlpythia_ref as R
from tests.helpers import TINY, tiny_cfg

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(cfg, sd, dtype=torch.float32):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    mc = VLPythiaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        vision_hidden_size=cfg.vision_hidden_size, num_vision_tokens=cfg.num_vision_tokens)
    m = VLPythiaForCausalLM(mc, compute_dtype=dtype, device="cuda")
    m.load_state_dict(sd, strict=True)
    return m


def _make(cfg, sd, tsd, ddp, rank_batches, pipeline=False, mode="all_reduce", grad_dtype=None):
    from mafed_amd import FeatureDistillation, Trainer
    t = TINY["m64"]
    model, teacher = _build(cfg, sd), _build(cfg, tsd)
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=t["B"], seed=3, pin_mem=False, accumulate_grad_batches=2)
    fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy="balanced", distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None)
    fd._update_model(teacher)
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    conf = types.SimpleNamespace(accumulate_grad_batches=2, replay_interval=2, grad_norm=2.0, learning_rate=1e-3, betas=(0.9, 0.98),
                                 weight_decay=0.01, optim="adamw", warmup_steps=1, total_steps=10)
    tr = Trainer(model, fd, conf, task_id=1, ddp=ddp, bucket_mb=0.05, pipeline_optimizer=pipeline, reduce_mode=mode, grad_dtype=grad_dtype)
    return model, fd, tr


def _batches(cfg, rank):
    t = TINY["m64"]
    return [{k: v.cuda() for k, v in R.make_batch(cfg, t["B"], t["T"], seed=100 + 10 * rank + i, pad=True, n_answer=3).items()} for i in range(4)]


def _worker(rank, world, port, q, pipeline=False, mode="all_reduce", grad_dtype=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = tiny_cfg("m64")
        sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
        tsd = R.perturb(sd, seed=22, std=5e-3)
        model, fd, tr = _make(cfg, sd, tsd, True, None, pipeline=pipeline, mode=mode, grad_dtype=grad_dtype)
        assert tr.reducer.world == world and tr.reducer.mode == mode
        bs = _batches(cfg, rank)
        gns = []
        for i in range(4):  # accumulate 2, replay every 2nd micro-batch: two optimiser steps
            fd.mem_dataloader = [dict(bs[(i + 1) % 4])]
            rec = tr.step(dict(bs[i]), i)
            if rec["stepped"]:
                gns.append(float(rec["grad_norm"]))
        tr.join()
        torch.cuda.synchronize()
        q.put((rank, gns, model.flat_params.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,pipeline,mode,grad_dtype", [
    (2, False, "all_reduce", None), (2, True, "all_reduce", None),
    (2, True, "reduce_scatter", None), (2, True, "reduce_scatter", torch.bfloat16),
    (4, True, "all_reduce", None), (4, True, "reduce_scatter", torch.bfloat16)])
def test_ddp_step_equals_mean_of_rank_gradients(world, pipeline, mode, grad_dtype):
    """pipeline=True: the ranks run Trainer(pipeline_optimizer=True) (AdamW chunks on their own stream behind the bucket
    collectives, next forward waiting per layer) -- what bench.py launches at N > 1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, pipeline, mode, grad_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=400) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(30)
    assert len(res) == world
    # replicas stay identical
    for r in range(1, world):
        assert np.array_equal(res[0][2], res[r][2])
        assert res[0][1] == res[r][1]
    # single-process emulation: run every rank's micro-batches, average gradients by hand before the optimiser step
    cfg = tiny_cfg("m64")
    sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=22, std=5e-3)
    model, fd, tr = _make(cfg, sd, tsd, False, None)
    bs = [_batches(cfg, r) for r in range(world)]
    gns = []
    for step in range(2):
        acc = torch.zeros_like(model.flat_grads)
        for rank in range(world):
            model.flat_grads.zero_()
            for j in range(2):
                i = 2 * step + j
                fd.mem_dataloader = [dict(bs[rank][(i + 1) % 4])]
                loss, _ = tr.training_step(dict(bs[rank][i]), i)
                (loss / 2).backward()
            acc += model.flat_grads
        model.flat_grads.copy_(acc / world)
        gns.append(float(tr.optimizer.clip_grad_norm_(2.0)))
        tr.optimizer.step()
        tr.scheduler.step()
        tr.optimizer.zero_grad()
    torch.cuda.synchronize()
    if grad_dtype is None:
        np.testing.assert_allclose(res[0][1], gns, rtol=1e-5)
        np.testing.assert_allclose(res[0][2], model.flat_params.detach().cpu().numpy(), rtol=0, atol=2e-6)
    else:  # bf16 buckets: every gradient element carries a 2^-9 relative rounding; AdamW's normalised update keeps it bounded by lr
        np.testing.assert_allclose(res[0][1], gns, rtol=1e-2)
        np.testing.assert_allclose(res[0][2], model.flat_params.detach().cpu().numpy(), rtol=0, atol=2.5e-3)


def _rccl_worker(port, q, mode, grad_dtype):
    """ONE rank on the box's one GPU over the real RCCL backend ("nccl"): the collectives the N > 1 run issues -- ReduceOp.AVG
    all-reduce, reduce-scatter + all-gather, bf16 staging, side-stream ordering -- go through RCCL itself; with a single rank the
    mean is the identity, so the step must equal the plain single-process step."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = tiny_cfg("m64")
        sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
        tsd = R.perturb(sd, seed=22, std=5e-3)
        out = []
        for ddp in (True, False):
            model, fd, tr = _make(cfg, sd, tsd, ddp, None, pipeline=True, mode=mode, grad_dtype=grad_dtype)
            if ddp:
                assert tr.reducer._nccl and tr.reducer._avg and tr.reducer.world == 1
                tr.reducer.force = True
            bs = _batches(cfg, 0)
            gns = []
            for i in range(4):
                fd.mem_dataloader = [dict(bs[(i + 1) % 4])]
                rec = tr.step(dict(bs[i]), i)
                if rec["stepped"]:
                    gns.append(float(rec["grad_norm"]))
            tr.join()
            torch.cuda.synchronize()
            if ddp:
                assert tr.reducer.bytes_per_step == model.flat_grads.numel() * (2 if grad_dtype is not None else 4)
            out.append((gns, model.flat_params.detach().cpu().numpy()))
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,grad_dtype", [("all_reduce", None), ("reduce_scatter", None), ("all_reduce", torch.bfloat16), ("reduce_scatter", torch.bfloat16)])
def test_single_rank_rccl_collectives_are_the_identity(mode, grad_dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q, mode, grad_dtype))
    p.start()
    (g_ddp, p_ddp), (g_ref, p_ref) = q.get(timeout=400)
    p.join(60)
    if grad_dtype is None:
        np.testing.assert_allclose(g_ddp, g_ref, rtol=1e-6)
        # two runs of the same step differ in the last bit of a few atomically accumulated gradient elements (token-embedding rows);
        # where such an element is ~0, Adam's first steps turn the sign of the noise into a +-lr move: a handful of outliers, no bias
        diff = np.abs(p_ddp - p_ref)
        assert (diff > 1e-7).mean() < 1e-4 and diff.max() <= 2.5e-3, ((diff > 1e-7).sum(), diff.max())
        assert np.linalg.norm(p_ddp - p_ref) <= 1e-4 * np.linalg.norm(p_ref)
    else:
        np.testing.assert_allclose(g_ddp, g_ref, rtol=1e-2)
        np.testing.assert_allclose(p_ddp, p_ref, rtol=0, atol=2.5e-3)


def _exact_worker(rank, world, port, q, exact, modality):
    import torch.distributed as dist
    from mafed_amd import FeatureDistillation
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = tiny_cfg("m64")
        sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
        tsd = R.perturb(sd, seed=22, std=5e-3)
        model, teacher = _build(cfg, sd), _build(cfg, tsd)
        fd = _exact_fd(cfg, teacher, modality, exact)
        fd.mem_dataloader = [_exact_batch(cfg, rank)]
        loss, _ = fd.replay(model)
        loss.backward()
        torch.cuda.synchronize()
        q.put((rank, float(loss.detach()), model.flat_grads.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


def _exact_fd(cfg, teacher, modality, exact):
    from mafed_amd import FeatureDistillation
    t = TINY["m64"]
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=t["B"], seed=3, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", num_hidden_layers=cfg.num_hidden_layers - 1,
                             distillation_modality_weighing_strategy=modality, distillation_layer_weighing_strategy="discounted",
                             gamma=0.5, distillation_layer=None, replay_coeff=0.0, exact_normaliser=exact)
    fd._update_model(teacher)
    fd.task_id = 1
    fd.num_vision_tokens = cfg.num_vision_tokens
    return fd


def _exact_batch(cfg, rank):
    t = TINY["m64"]
    b = R.make_batch(cfg, t["B"], t["T"], seed=300 + rank, pad=True, n_answer=3)
    if rank == 1:   # make the ranks' valid-text-token counts clearly different
        b["attention_mask"][:, : t["T"] // 2] = 0
        b["input_ids"][:, : t["T"] // 2] = 0
    return {k: v.cuda() for k, v in b.items()}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("modality", ["balanced", "equal"])
def test_exact_normaliser_equals_one_process_with_the_concatenated_batch(modality):
    """with ``exact_normaliser`` the rank MEAN of the distillation losses / gradients is the loss / gradient of one
    process holding both ranks' batches (masked means over the global token counts); without it (the reference's behaviour:
    distillation.py:248 normalises by the local counts) it is not, when the ranks' pad counts differ."""
    world = 2
    ctx = mp.get_context("spawn")
    out = {}
    for exact in (True, False):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_exact_worker, args=(r, world, port, q, exact, modality)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=400) for _ in range(world)], key=lambda x: x[0])
        for p in procs:
            p.join(30)
        out[exact] = (np.mean([r[1] for r in res]), np.mean([r[2] for r in res], axis=0))
    cfg = tiny_cfg("m64")
    sd = R.init_weights(cfg, seed=21, bias_std=0.02, ln_jitter=0.05)
    tsd = R.perturb(sd, seed=22, std=5e-3)
    model, teacher = _build(cfg, sd), _build(cfg, tsd)
    fd = _exact_fd(cfg, teacher, modality, False)
    b0, b1 = _exact_batch(cfg, 0), _exact_batch(cfg, 1)
    fd.mem_dataloader = [{k: torch.cat([b0[k], b1[k]], dim=0) for k in b0}]
    loss, _ = fd.replay(model)
    loss.backward()
    ref_loss, ref_grad = float(loss), model.flat_grads.detach().cpu().numpy()
    scale = np.abs(ref_grad).max()
    assert abs(out[True][0] - ref_loss) <= 1e-5 * abs(ref_loss)
    np.testing.assert_allclose(out[True][1], ref_grad, rtol=0, atol=2e-5 * scale)
    # negative control: the per-rank means (reference behaviour) are measurably NOT the concatenated-batch loss on these batches
    assert abs(out[False][0] - ref_loss) > max(2e-4 * abs(ref_loss), 20 * abs(out[True][0] - ref_loss)), \
        "the test batches must make the local and the global normaliser differ"
