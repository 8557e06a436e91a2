"""Row-sparse LM head (training): batches that carry ``max_label_rows`` run the head GEMM, the cross-entropy and the head's two gradient
GEMMs on the labelled rows only.  Loss and every parameter gradient must equal the dense head's (the dropped rows have no label: zero
loss weight, zero dlogits), the kernels that list / move the rows are checked against torch, and the replay buffer must attach the hint."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _batch(cfg, B, T, n_ans, seed, ragged=False):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g)
    am = torch.ones(B, T, dtype=torch.int64)
    labels = torch.full((B, T), -100, dtype=torch.int64)
    for b in range(B):
        k = n_ans if not ragged else int(torch.randint(0, n_ans + 1, (1,), generator=g))
        if k:
            labels[b, -k:] = ids[b, -k:]
        pad = int(torch.randint(0, 3, (1,), generator=g)) if ragged else 0
        am[b, :pad] = 0
    feats = torch.randn(B, cfg.num_vision_tokens, cfg.vision_hidden_size, generator=g)
    return {"input_ids": ids.to(DEV), "attention_mask": am.to(DEV), "labels": labels.to(DEV), "patch_embeddings": feats.to(DEV)}


def test_label_rows_and_gather_kernels():
    from mafed_amd import ops
    g = torch.Generator().manual_seed(1)
    B, T, Rc = 9, 12, 5
    labels = torch.full((B, T), -100, dtype=torch.int64)
    for b in range(B):
        for t in torch.randperm(T, generator=g)[: b % 5].tolist():
            labels[b, t] = 7 + t
    ros, sor, lc, ov = ops.label_rows(labels.to(DEV), Rc)
    ros, sor, lc = ros.cpu().view(B, Rc), sor.cpu().view(B, T), lc.cpu()
    assert int(ov) == 0
    for b in range(B):
        want = [t for t in range(T - 1) if labels[b, t + 1] != -100]
        assert ros[b, : len(want)].tolist() == [b * T + t for t in want] and (ros[b, len(want):] == -1).all()
        assert lc[b, 0] == -100 and lc[b, 1: len(want) + 1].tolist() == [int(labels[b, t + 1]) for t in want] and (lc[b, len(want) + 1:] == -100).all()
        for n, t in enumerate(want):
            assert sor[b, t] == b * Rc + n
        assert (sor[b][[t for t in range(T) if t not in want]] == -1).all()
    # a bound that is too small raises the device flag
    assert int(ops.label_rows(labels.to(DEV), 3)[3]) == 1
    for dt, h in ((torch.float32, 20), (torch.bfloat16, 24)):
        src = torch.randn(B * T, h, generator=g).to(dt).to(DEV)
        out = ops.gather_rows(src, ros.view(-1).to(DEV))
        ref = torch.where(ros.view(-1, 1) >= 0, src.cpu()[ros.view(-1).clamp_min(0)], torch.zeros(1, dtype=dt))
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype,B,T,n_ans,ragged", [(torch.float32, 6, 12, 3, True), (torch.float32, 4, 10, 2, False),
                                                     (torch.bfloat16, 32, 16, 3, True), (torch.bfloat16, 64, 16, 1, False)])
def test_sparse_head_equals_dense_head(dtype, B, T, n_ans, ragged):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         vision_hidden_size=32, num_vision_tokens=8)
    model = VLPythiaForCausalLM(cfg, compute_dtype=dtype, device=DEV, seed=5)
    batch = _batch(cfg, B, T, n_ans, seed=B + T, ragged=ragged)

    def run(hint):
        model.zero_grad()
        kw = {"max_label_rows": hint} if hint is not None else {}
        out = model(**batch, **kw, return_dict=True)
        out.loss.backward()
        torch.cuda.synchronize()
        return out, float(out.loss.detach()), model.flat_grads.clone()

    dense, l0, g0 = run(None)
    sparse, l1, g1 = run(n_ans)
    assert dense.logits is not None and sparse.logits is None, "the compact logits are internal"
    assert int(model.last_label_overflow) == 0
    tol = 1e-6 if dtype == torch.float32 else 2e-3
    assert abs(l1 - l0) <= tol * max(1.0, abs(l0)), (l0, l1)
    rel = float((g1 - g0).norm() / g0.norm())
    assert rel <= (1e-5 if dtype == torch.float32 else 1e-2), f"gradients: relative difference {rel:.3e}"
    # switched off on the model: the hint is ignored
    model.sparse_lm_head = False
    assert run(n_ans)[0].logits is not None
    model.sparse_lm_head = True


@pytest.mark.parametrize("dtype,B", [(torch.float32, 6), (torch.bfloat16, 32)])
def test_too_small_a_hint_poisons_the_loss(dtype, B):
    """A caller-supplied max_label_rows below the true count drops labelled rows: the device overflow flag turns the loss into NaN
    (no host synchronisation on the step) instead of training on a wrong loss; a hint of 0 still builds a valid compact problem."""
    import math
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = VLPythiaConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         vision_hidden_size=32, num_vision_tokens=8)
    model = VLPythiaForCausalLM(cfg, compute_dtype=dtype, device=DEV, seed=5)
    n_lab, T = 5, 16          # 5 labelled positions per sample; bf16 rounds the slots per sample up to whole 128-row tiles (B = 32: 4 or 8)
    batch = _batch(cfg, B, T, n_lab, seed=2, ragged=False)
    good = model(**batch, max_label_rows=n_lab, return_dict=True)
    assert math.isfinite(float(good.loss)) and int(model.last_label_overflow) == 0
    ran_sparse = 0
    for hint in (2, 1, 0):
        need = max(2, hint + 1)
        slots = need if dtype == torch.float32 else next(r for r in range(need, T + 1) if (B * r) % 128 == 0)
        assert slots - 1 < n_lab, "the case must overflow"
        bad = model(**batch, max_label_rows=hint, return_dict=True)
        ran_sparse += 1
        assert int(model.last_label_overflow) == 1
        assert math.isnan(float(bad.loss)), f"hint {hint}: overflow must poison the loss"
    assert ran_sparse == 3
    # a hint that is too LARGE only costs rows: exact loss, no flag
    roomy = model(**batch, max_label_rows=n_lab + 2, return_dict=True)
    assert int(model.last_label_overflow) == 0 and abs(float(roomy.loss) - float(good.loss)) <= 1e-5 * abs(float(good.loss))


def test_replay_buffer_attaches_the_hint_and_trainer_uses_it():
    import types
    from mafed_amd import CLMethod, Trainer, VLPythiaConfig, VLPythiaForCausalLM
    from mafed_amd.methods import HBMReplayBuffer
    cfg = VLPythiaConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         vision_hidden_size=32, num_vision_tokens=8)
    B, T = 32, 16
    data = {k: v.cpu() for k, v in _batch(cfg, 64, T, 3, seed=3, ragged=True).items()}
    opts = types.SimpleNamespace(tasks=["a", "b"], batch_size=B, seed=7, pin_mem=False, accumulate_grad_batches=1)
    conf = types.SimpleNamespace(accumulate_grad_batches=1, replay_interval=1, grad_norm=2.0, learning_rate=1e-3, betas=(0.9, 0.98), weight_decay=0.01,
                                 optim="adamw", warmup_steps=0, total_steps=100)
    res = []
    for sparse in (True, False):
        model = VLPythiaForCausalLM(cfg, compute_dtype=torch.bfloat16, device=DEV, seed=9)
        model.sparse_lm_head = sparse
        er = CLMethod["replay"](opts=opts, memory_size=64, model_type="vlpythia")
        er.update(dataset=data, model=model)
        buf = er.mem_dataloader
        assert isinstance(buf, HBMReplayBuffer) and buf.max_label_rows == 3
        assert buf._draw()["max_label_rows"] == 3
        buf.gen.manual_seed(11)
        buf._next = None
        tr = Trainer(model, er, conf, task_id=1)
        task = {k: v[:B].to(DEV) for k, v in data.items()}
        losses = [float(tr.step(task, i)["loss"]) for i in range(3)]
        tr.join()
        torch.cuda.synchronize()
        res.append((losses, model.flat_params.clone()))
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 3e-3 * max(1.0, abs(b)), (res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).norm() / res[1][1].norm()) <= 2e-3
