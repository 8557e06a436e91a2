"""CPU: host-side mirror of the reference plugin surface (registry, strategies, weights, schedule, state dict)."""
import copy
import types

import numpy as np
import pytest
import torch

from oracle import vlpythia_ref as R
from tests.helpers import load_golden, tiny_cfg


def small_model(device="cpu"):
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
    cfg = tiny_cfg("t64")
    mc = VLPythiaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                        num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                        vision_hidden_size=cfg.vision_hidden_size, num_vision_tokens=cfg.num_vision_tokens)
    return cfg, VLPythiaForCausalLM(mc, compute_dtype=torch.float32, device=device)


def test_registry_and_protocol():
    from mafed_amd import CLMethod, CLStrategy, ER, FeatureDistillation, Naive, model_architecture
    assert set(CLMethod) == {"naive", "ewc", "replay", "featdistill"}   # the reference's registry (mafed/methods/__init__.py:6-11)
    assert CLMethod["featdistill"] is FeatureDistillation and CLMethod["replay"] is ER and CLMethod["naive"] is Naive
    assert "vlpythia" in model_architecture
    base = CLStrategy(opts=types.SimpleNamespace(accumulate_grad_batches=4))
    assert base.update_freq == 4 and base.task_id == 0
    assert base.replay(None) == (None, 0)
    with pytest.raises(NotImplementedError):
        base.compute_loss(None, 1.0)
    assert [base._is_batch_after_step(i) for i in range(4)] == [False, False, False, True]
    n = Naive()
    assert n.update_freq == 1 and n.compute_loss(None, 3.5) == 3.5
    n.update(None)
    assert n.task_id == 1


def test_distillation_weights_match_reference_vectors():
    from mafed_amd.methods import DistillationWeights
    g = load_golden("optim.npz")
    for nh in (11, 15, 23):
        for gam in (0.5, 0.8, 0.9):
            dw = DistillationWeights("balanced", "discounted", gamma=gam, num_hidden_layers=nh, distillation_layer=None)
            assert dw.get_distillation_layers() == list(range(nh))
            np.testing.assert_allclose(dw.layer_coeffs.numpy(), g[f"g4/discounted/nh{nh}/g{gam}"], rtol=1e-6)
            np.testing.assert_allclose(dw.layer_coeff_vector("cpu").numpy(), g[f"g4/discounted/nh{nh}/g{gam}"], rtol=1e-6)
        dw = DistillationWeights("balanced", "equal", num_hidden_layers=nh, distillation_layer=None)
        np.testing.assert_allclose(dw.layer_coeffs.numpy(), g[f"g4/equal/nh{nh}"], rtol=1e-6)
    dw = DistillationWeights("equal", "single", num_hidden_layers=11, distillation_layer=3)
    assert dw.get_distillation_layers() == [3] and dw.get_layer_loss_weight(3) == 1.0
    dw = DistillationWeights("equal", "discounted", num_hidden_layers=11, distillation_layer=3)  # a concrete layer wins
    assert dw.get_distillation_layers() == [3]
    dw = DistillationWeights("equal", "cumulative", gamma=0.8, num_hidden_layers=11, distillation_layer=4)
    assert dw.get_distillation_layers() == [0, 1, 2, 3]
    with pytest.raises(AssertionError):
        DistillationWeights("equal", "single", distillation_layer=None)
    with pytest.raises(AssertionError):
        DistillationWeights("equal", "cumulative", distillation_layer=None)
    with pytest.raises(NotImplementedError):
        DistillationWeights("nonsense", "equal", distillation_layer=None).get_modality_loss_weights({}, 0)
    # modality weights: equal = token shares, balanced = 0.5/0.5, adaptive = per-layer vector
    lang = torch.zeros(2, 10, dtype=torch.int64)
    lang[:, 6:] = 1
    img = torch.zeros(2, 10, dtype=torch.int64)
    img[:, :6] = 1
    lw, vw = DistillationWeights("equal", "equal", distillation_layer=None).get_modality_loss_weights({"lang_masks": lang, "image_masks": img}, 0)
    assert abs(float(lw) - 0.4) < 1e-6 and abs(float(vw) - 0.6) < 1e-6
    dwv = DistillationWeights("equal", "equal", num_hidden_layers=3, distillation_layer=None)
    a, b = dwv.modality_weight_vectors(torch.tensor(8.0), torch.tensor(12.0), [0, 1, 2], "cpu")
    assert torch.allclose(a, torch.full((3,), 0.4)) and torch.allclose(b, torch.full((3,), 0.6))
    dwa = DistillationWeights("adaptive", "equal", num_hidden_layers=3, distillation_layer=None)
    dwa.lang_coeff = torch.tensor([0.2, 0.5, 0.9])
    a, b = dwa.modality_weight_vectors(torch.tensor(8.0), torch.tensor(12.0), [0, 2], "cpu")
    assert torch.allclose(a, torch.tensor([0.2, 0.9]))
    # running mean over tasks of the adaptive importances (distillation_loss_weights.py:62-69)
    dwa.compute_adaptive_weights = lambda m, d: torch.tensor([0.4, 0.4, 0.4])
    dwa.update_weights(None, None, task_id=1)
    assert torch.allclose(dwa.lang_coeff, torch.tensor([0.3, 0.45, 0.65]))


def test_schedule_and_warmup():
    from mafed_amd.optim import compute_warmup, lr_lambda
    g = load_golden("optim.npz")
    np.testing.assert_allclose([lr_lambda(s, 3, 12) for s in range(14)], g["g5/lambda_w3_t12"], rtol=0, atol=1e-12)
    assert compute_warmup(1000, 4, 0.1) == (250 * 60, 1500)  # the 60 is hard-coded upstream
    assert compute_warmup(1001, 4, 0.1, warmup_steps=7) == (251 * 60, 7)
    assert R.compute_warmup(1000, 4, 0.1) == (15000, 1500)


def test_state_dict_contract_groups_and_deepcopy():
    from mafed_amd.model import is_no_decay
    cfg, model = small_model()
    names = [k for k, _ in R.param_shapes(cfg)]
    sd = model.state_dict()
    assert list(sd) == names
    for k, shp in R.param_shapes(cfg):
        assert tuple(sd[k].shape) == shp
    # reference grouping: only 'bias' names escape decay; LayerNorm weights are decayed
    for k in names:
        assert is_no_decay(k) == (R.param_group_of(k) % 2 == 1)
    n_decay = model.decay_split()
    for k in names:
        o, n, _ = model._offsets[k]
        assert (o + n <= n_decay) == (not is_no_decay(k))
        assert o % 64 == 0
    # parameters and their .grad are views of the flat buffers
    p = dict(model.named_parameters())["gpt_neox.layers.1.attention.dense.weight"]
    with torch.no_grad():
        p.add_(1.0)
    o, n, shp = model._offsets["gpt_neox.layers.1.attention.dense.weight"]
    assert torch.equal(model.flat_params[o:o + n].view(shp), p.data)
    assert p.grad.data_ptr() == model.flat_grads[o:o + n].data_ptr()
    # load a reference-shaped state dict, deepcopy = teacher snapshot with its own storage
    ref_sd = R.init_weights(cfg, seed=3, bias_std=0.02, ln_jitter=0.05)
    model.load_state_dict(ref_sd, strict=True)
    assert model._shadow_dirty
    t = copy.deepcopy(model)
    assert t is not model and t.flat_params.data_ptr() != model.flat_params.data_ptr()
    for k in names:
        assert torch.equal(t.state_dict()[k], ref_sd[k])
    assert list(model.vision_encoder.parameters()) == []
    # Lightning checkpoints prefix names with "model." (utils/checkpoint.py:19-21)
    pref = {"model." + k: v for k, v in ref_sd.items()}
    model.load_state_dict({k[len("model."):]: v for k, v in pref.items()}, strict=True)


def test_no_cpu_fallback():
    cfg, model = small_model()
    batch = R.make_batch(cfg, 2, 6, seed=1)
    with pytest.raises(RuntimeError):
        model(**batch)


def test_hbm_replay_buffer_and_memory_update():
    from mafed_amd import FeatureDistillation
    from mafed_amd.methods import HBMReplayBuffer
    cfg = tiny_cfg("t64")
    data = R.make_batch(cfg, 20, 6, seed=2)
    buf = HBMReplayBuffer(batch_size=4, device="cpu", seed=0)
    buf.add(data)
    shorter = R.make_batch(cfg, 5, 4, seed=3)
    buf.add(shorter)  # shorter text gets LEFT padded (pad id 0 / mask 0 / label -100)
    assert len(buf) == 25 and buf.data["input_ids"].shape[1] == 6
    assert (buf.data["attention_mask"][20:, :2] == 0).all() and (buf.data["labels"][20:, :2] == -100).all()
    b = next(iter(buf))
    assert b["input_ids"].shape == (4, 6) and b["patch_embeddings"].dtype == torch.bfloat16
    # rank shards are disjoint
    r0 = HBMReplayBuffer(4, "cpu", seed=0, rank=0, world_size=2)
    r1 = HBMReplayBuffer(4, "cpu", seed=0, rank=1, world_size=2)
    r0.add(data); r1.add(data)
    i0 = {tuple(x.tolist()) for x in r0.sample()["input_ids"]}
    i1 = {tuple(x.tolist()) for x in r1.sample()["input_ids"]}
    first_half = {tuple(x.tolist()) for x in data["input_ids"][:10]}
    assert i0 <= first_half and not (i1 & first_half)
    # plugin: memory_per_task, rng stream, task counter (distillation.py:36-37, 75-79, 182-190)
    opts = types.SimpleNamespace(tasks=["a", "b", "c"], batch_size=4, seed=42, pin_mem=False, accumulate_grad_batches=1)
    fd = FeatureDistillation(memory_size=10, opts=opts, model_type="vlpythia", distillation_layer_weighing_strategy="discounted",
                             distillation_layer=None, num_hidden_layers=2)
    assert fd.memory_per_task == 5 and fd.num_vision_tokens == 256
    _, model = small_model()
    fd.update(dataset=data, model=model, dataloader=None)
    assert fd.task_id == 1 and len(fd.mem_dataloader) == 5 and fd.past_model is not model
    assert not any(p.requires_grad for p in fd.past_model.parameters())
    expect = np.random.default_rng(42).choice(np.arange(20), 5, replace=False)
    got = fd.datasets[0]["input_ids"]
    assert torch.equal(got, data["input_ids"][torch.as_tensor(np.sort(expect))])


def test_ewc_foreign_module_path_matches_oracle():
    """CLMethod["ewc"] on a plain nn.Module (no flat buffers, CPU): importances, online update and penalty through the
    multi-tensor route against the oracle formulas."""
    import torch
    from mafed_amd import CLMethod
    from oracle import vlpythia_ref as R

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a = torch.nn.Linear(5, 7)
            self.b = torch.nn.Linear(7, 3)

        def forward(self, input_ids, compute_loss=True, return_dict=True, **kw):
            x = torch.nn.functional.one_hot(input_ids % 5, 5).float().mean(1)
            out = self.b(torch.tanh(self.a(x)))
            return type("O", (), {"loss": out.pow(2).mean()})()

    m = Tiny()
    loader = [{"input_ids": torch.arange(12).reshape(4, 3) + i} for i in range(3)]
    ewc = CLMethod["ewc"](reg_lambda=7.0, online=True, online_factor=0.9)
    ewc.update(model=m, dataloader=loader)
    want = {k: torch.zeros_like(p) for k, p in m.named_parameters()}
    for b in loader:
        m.zero_grad()
        (4 * m(**b).loss).backward()
        for k, p in m.named_parameters():
            want[k] += p.grad.pow(2)
    for k in want:
        assert torch.allclose(ewc.fisher[0][k], want[k] / 12.0, rtol=1e-6, atol=1e-9)
    old = {k: p.detach().clone() for k, p in m.named_parameters()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.01)
    base = torch.tensor(1.5)
    got = ewc.compute_loss(m, base)
    ref = base + R.ewc_penalty(dict(m.named_parameters()), old, ewc.fisher[0], 7.0)
    assert torch.allclose(got, ref, rtol=1e-6)
    got.backward()
    assert all(p.grad is not None for p in m.parameters())
    ewc.update(model=m, dataloader=loader)   # task_id 1: overwrite
    f1 = {k: v.clone() for k, v in ewc.fisher[0].items()}
    ewc.update(model=m, dataloader=loader)   # task_id 2: new + 0.9 * old
    for k in f1:
        assert torch.allclose(ewc.fisher[0][k], f1[k] * 1.9, rtol=1e-5)


def test_clip_norm_plan_tiles_the_gradient_buffer():
    """FlatAdamW._norm_plan: the per-hook ranges of the incremental clip norm (LM head, layers L-1 .. 0, embeddings / projector / biases)
    must tile the flat gradient buffer exactly once -- otherwise the norm would miss or double-count elements -- with 16-byte aligned
    starts and consecutive partial slots."""
    from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM, ops
    from mafed_amd.optim import FlatAdamW
    for L, h in ((5, 32), (3, 64)):
        cfg = VLPythiaConfig(vocab_size=64, hidden_size=h, num_hidden_layers=L, num_attention_heads=2, intermediate_size=4 * h,
                             vision_hidden_size=16, num_vision_tokens=4)
        model = VLPythiaForCausalLM(cfg, compute_dtype=torch.float32, device="cpu")
        opt = FlatAdamW.__new__(FlatAdamW)     # no device state: only the plan is under test
        opt.model = model
        plan = opt._norm_plan()
        assert plan is not None and sorted(plan) == [-1] + list(range(L)) + [L]
        ranges = sorted((lo, hi, slot) for rs in plan.values() for lo, hi, slot in rs)
        assert ranges[0][0] == 0 and ranges[-1][1] == model.flat_grads.numel()
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])), "ranges must be contiguous and disjoint"
        assert all(lo % 4 == 0 for lo, _, _ in ranges)
        order = [r for t in [L] + list(range(L - 1, -1, -1)) + [-1] for r in plan[t]]
        slot = 0
        for lo, hi, s in order:
            assert s == slot
            slot += ops.gradnorm_blocks(hi - lo)
        # (behind the range partials: 16 slots per layer weight matrix for the squares the weight-gradient epilogues leave)
        assert slot == opt._norm_dw_lo and opt._norm_dw_n == 16 * 4 * L
        assert slot + opt._norm_dw_n == opt._norm_slots == opt._norm_partials.numel()


def test_clip_tower_row_padding_minimises_tile_rounds():
    from mafed_amd.vision import _pad_rows
    assert _pad_rows(32 * 257, (1024, 3072, 4096)) == 8352      # 58 x 144 rows: 1 + 3 + 4 rounds of 512 tiles (65 x 128: 2 + 4 + 5)
    assert _pad_rows(8192, (1024, 3072, 4096)) == 8192          # already a multiple of 128 that fills whole rounds
    for rows in (1, 100, 257, 5000):
        p = _pad_rows(rows, (1024, 3072, 4096))
        assert p >= rows and (p % 128 == 0 or p % 144 == 0)


def test_every_ops_attribute_the_package_uses_exists():
    """The host code reaches the kernels through ``mafed_amd.ops``; a wrapper lost in an edit only shows on the GPU box otherwise."""
    import glob
    import os
    import re
    from mafed_amd import ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "mafed_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(root, "tests", "*.py")) + \
        glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    missing = set()
    for f in files:
        for m in re.finditer(r"\bops\.([A-Za-z_][A-Za-z_0-9]*)", open(f).read()):
            if not hasattr(ops, m.group(1)):
                missing.add((os.path.basename(f), m.group(1)))
    assert not missing, missing


def test_runtime_env_is_opt_in_and_respects_the_environment(monkeypatch):
    """Importing the package sets no HIP runtime option; apply_recommended_runtime_env() sets the tuned ones unless exported already,
    explicit overrides win, and a call after the runtime initialised warns instead of pretending."""
    import warnings
    import mafed_amd
    import importlib
    RE = importlib.import_module("mafed_amd.runtime_env")   # (the package re-exports a function of the same name)
    monkeypatch.delenv("HIP_FORCE_DEV_KERNARG", raising=False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")
    monkeypatch.setattr(RE, "_hip_initialised", lambda: False)
    env = mafed_amd.apply_recommended_runtime_env()
    assert env["HIP_FORCE_DEV_KERNARG"] == "1" and env["GPU_MAX_HW_QUEUES"] == "6"        # an exported value wins over the recommendation
    env = mafed_amd.apply_recommended_runtime_env({"GPU_MAX_HW_QUEUES": "3"})
    assert env["GPU_MAX_HW_QUEUES"] == "3"                                                # an explicit override wins over the environment
    monkeypatch.delenv("HIP_FORCE_DEV_KERNARG", raising=False)
    monkeypatch.setattr(RE, "_hip_initialised", lambda: True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        env = mafed_amd.apply_recommended_runtime_env()
    assert env["HIP_FORCE_DEV_KERNARG"] is None and any("no effect" in str(x.message) for x in w)


def test_no_persistent_gemm_flag_is_per_thread_and_restores():
    """the data-parallel backward's kernel choice is a per-call flag (MAFED_EPI_NO_PERSISTENT), thread-local on the host --
    not a process-wide switch that a forced tuning variant or another thread's launches would see."""
    import threading
    from mafed_amd import ops
    from mafed_amd._lib import EPI_NO_PERSISTENT
    seen = {}

    def other():
        seen["other"] = getattr(ops._tls, "no_pp", 0)

    assert getattr(ops._tls, "no_pp", 0) == 0
    with ops.no_persistent_gemm():
        assert ops._tls.no_pp == EPI_NO_PERSISTENT
        t = threading.Thread(target=other)
        t.start()
        t.join()
        with ops.no_persistent_gemm(False):      # nested "leave it as it is"
            assert ops._tls.no_pp == EPI_NO_PERSISTENT
        try:
            with ops.no_persistent_gemm():
                raise ValueError
        except ValueError:
            pass
        assert ops._tls.no_pp == EPI_NO_PERSISTENT
    assert ops._tls.no_pp == 0 and seen["other"] == 0


def test_replay_buffer_refuses_an_empty_rank_shard():
    """with fewer samples than ranks a shard is empty; drawing index `lo` anyway read another rank's sample (or past the end)."""
    import torch
    from mafed_amd.methods import HBMReplayBuffer
    data = {"input_ids": torch.arange(6).view(3, 2), "attention_mask": torch.ones(3, 2, dtype=torch.int64),
            "labels": torch.arange(6).view(3, 2), "patch_embeddings": torch.zeros(3, 4, 8)}
    ok = HBMReplayBuffer(2, torch.device("cpu"), seed=0, rank=1, world_size=3)
    ok.add(data)
    assert ok._draw()["input_ids"].shape[0] == 1
    empty = HBMReplayBuffer(2, torch.device("cpu"), seed=0, rank=0, world_size=5)    # (3 * 0) // 5 == (3 * 1) // 5 == 0: rank 0's shard is empty
    empty.add(data)
    with pytest.raises(RuntimeError, match="empty shard"):
        empty._draw()


def test_fused_squares_query_is_host_logic():
    """mafed_gemm_grouped_fuses_sumsq needs no GPU (it only runs the dispatcher's pick): the 410M weight-gradient group fuses its squares,
    the h = 768 / 2048 groups (256 x 256-tile kernel) and a lone small product do not -- the model then keeps the norm hook's range pass
    (the regression this guards cost the 160M / 1.4B-shape steps 15 - 27 % for most of round 4)."""
    import ctypes
    from mafed_amd import _lib
    lib = _lib.load()
    rows = 32 * 288

    def fuses(shapes):
        n = len(shapes)
        arr = lambda k: (ctypes.c_int64 * n)(*[int(s[k]) for s in shapes])
        Ms, Ns, Ks = arr(0), arr(1), arr(2)
        return bool(lib.mafed_gemm_grouped_fuses_sumsq(_lib.BF16, 1, 0, _lib.F32, ctypes.cast(Ms, ctypes.c_void_p), ctypes.cast(Ns, ctypes.c_void_p),
                                                       ctypes.cast(Ks, ctypes.c_void_p), n))

    layer = lambda h: [(3 * h, h, rows), (h, h, rows), (4 * h, h, rows), (h, 4 * h, rows)]
    assert fuses(layer(1024) * 2)
    assert not fuses(layer(2048) * 2)
    assert not fuses(layer(768) * 2)
    assert not fuses(layer(1024)[1:2])
