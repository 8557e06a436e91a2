"""CPU, world_size 2 over gloo: the data-parallel gradient bucket reducer (no reference counterpart; README.md:47)."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class FakeModel:
    """Only what GradReducer touches: flat_grads, offsets in the model's layout, the hook slot."""

    def __init__(self):
        from mafed_amd import VLPythiaConfig, VLPythiaForCausalLM
        cfg = VLPythiaConfig(vocab_size=64, hidden_size=32, num_hidden_layers=5, num_attention_heads=2, intermediate_size=128,
                             vision_hidden_size=16, num_vision_tokens=4)
        real = VLPythiaForCausalLM(cfg, compute_dtype=torch.float32, device="cpu")
        self.config, self._offsets = cfg, real._offsets
        self._n_decay = real.decay_split()
        self.flat_grads = torch.zeros_like(real.flat_grads)
        self.grad_ready_hook = None

    def decay_split(self):
        return self._n_decay


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


VARIANTS = [  # (mode, grad_dtype, average)
    ("all_reduce", None, False), ("all_reduce", None, True),
    ("reduce_scatter", None, False), ("reduce_scatter", None, True),
    ("all_reduce", torch.bfloat16, False), ("reduce_scatter", torch.bfloat16, True),
]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mafed_amd.dist import GradReducer
        out = []
        for mode, gdt, avg in VARIANTS:
            m = FakeModel()
            red = GradReducer(m, bucket_mb=0.02, grad_dtype=gdt, mode=mode, average=avg)  # several buckets even for the toy model
            n = m.flat_grads.numel()
            L = m.config.num_hidden_layers
            # (values exactly representable in bf16 so that the bf16 buckets can be checked for equality too)
            base = (torch.arange(n, dtype=torch.float32) % 61) * 0.25
            # window 1: reducer disabled (non-final micro-batch of an accumulation window) -> grads stay local
            red.enabled = False
            m.flat_grads.copy_(base * (rank + 1))
            for trig in [L] + list(range(L - 1, -1, -1)) + [-1]:
                m.grad_ready_hook(trig)
            red.wait()
            ok_local = torch.equal(m.flat_grads, base * (rank + 1))
            # window end: hooks fire in backward order; every element is averaged exactly once
            red.enabled = True
            red.begin_window()
            for trig in [L] + list(range(L - 1, -1, -1)) + [-1]:
                m.grad_ready_hook(trig)
            red.wait()
            mean = base * (sum(range(1, world + 1)) / world)
            ok_mean = torch.allclose(m.flat_grads, mean, rtol=(2e-2 if gdt is not None else 1e-6), atol=0)
            ok_bytes = red.bytes_per_step == n * (2 if gdt is not None else 4)
            # buckets tile the flat buffer without overlap
            cover = torch.zeros(n)
            for _, (lo, hi) in red.buckets:
                cover[lo:hi] += 1
            out.append((mode, str(gdt), avg, ok_local, ok_mean, ok_bytes, bool((cover == 1).all()), len(red.buckets)))
        # the SUM / world and the AVG forms give the same fp32 result bit for bit on every rank (power-of-two world)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_gradient_mean(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert len(res) == world
    for rank, out in res:
        assert len(out) == len(VARIANTS)
        for mode, gdt, avg, ok_local, ok_mean, ok_bytes, ok_cover, nb in out:
            tag = f"rank {rank} / {mode} / {gdt} / avg={avg}"
            assert ok_local, f"{tag}: gradients changed although the reducer was disabled"
            assert ok_mean, f"{tag}: gradients are not the rank mean"
            assert ok_bytes, f"{tag}: payload accounting"
            assert ok_cover, f"{tag}: buckets do not tile the flat gradient buffer exactly once"
            assert nb >= 4


def test_reducer_rejects_unknown_modes():
    from mafed_amd.dist import GradReducer
    m = FakeModel()
    with pytest.raises(ValueError):
        GradReducer(m, mode="ring")
    with pytest.raises(ValueError):
        GradReducer(m, grad_dtype=torch.float16)


# ---- teacher broadcast, replay-memory rank shards, exact normaliser (CPU tensors over gloo) ------------------------------------
def _aux_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mafed_amd.dist import broadcast_teacher
        from mafed_amd.methods import HBMReplayBuffer
        from mafed_amd.methods.distillation import global_token_counts
        # (1) every replica's frozen teacher := rank 0's, and the bf16 shadow is marked stale
        teacher = types.SimpleNamespace(flat_params=torch.full((1000,), float(rank + 1)), _shadow_dirty=False)
        broadcast_teacher(teacher)
        ok_bcast = bool((teacher.flat_params == 1.0).all()) and teacher._shadow_dirty
        # (2) rank shards of the replay memory: every draw stays inside [n r / W, n (r + 1) / W)
        n = 37
        mem = HBMReplayBuffer(4, "cpu", seed=5 + rank, rank=rank, world_size=world)
        mem.add({"input_ids": torch.arange(n).view(n, 1).repeat(1, 3), "attention_mask": torch.ones(n, 3, dtype=torch.int64),
                 "labels": torch.full((n, 3), -100), "patch_embeddings": torch.zeros(n, 2, 4)})
        seen = set()
        for _ in range(200):
            seen.update(int(i) for i in mem.sample()["input_ids"][:, 0])
        # (3) exact normaliser: counts become the rank mean, sums stay local
        sums = torch.tensor([[1.0 + rank, 2.0, 10.0 + 4 * rank, 64.0], [3.0, 4.0 + rank, 10.0 + 4 * rank, 64.0]])
        gl = global_token_counts(sums)
        q.put((rank, ok_bcast, sorted(seen), gl.tolist(), sums.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_teacher_broadcast_memory_shards_and_global_counts(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aux_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(30)
    n = 37
    union = []
    for rank, ok_bcast, seen, gl, sums in res:
        assert ok_bcast, f"rank {rank}: teacher not equal to rank 0's after the broadcast"
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        assert seen == list(range(lo, hi)), f"rank {rank}: draws {seen[:3]}..{seen[-3:]} outside / not covering its shard [{lo}, {hi})"
        union += seen
        mean_lang = sum(10.0 + 4 * r for r in range(world)) / world
        for row_g, row_s in zip(gl, sums):
            assert row_g[:2] == row_s[:2] and row_g[2] == pytest.approx(mean_lang) and row_g[3] == 64.0
    assert sorted(union) == list(range(n)), "the rank shards are disjoint and cover the memory"
    # the point of the mean count: rank-mean of S_r / n_mean == sum S_r / sum n_r
    S = [1.0 + r for r in range(world)]
    N = [10.0 + 4 * r for r in range(world)]
    nm = sum(N) / world
    assert sum(s / nm for s in S) / world == pytest.approx(sum(S) / sum(N))


def test_global_counts_are_the_identity_without_a_process_group():
    from mafed_amd.methods.distillation import global_token_counts
    s = torch.rand(3, 4)
    assert global_token_counts(s) is s


# ---- validation metrics under data parallelism (mafed/utils/eval_utils.py:89-90,135-137) -------------------------------------------
def _metric_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mafed_amd.dist import generative_accuracy, reduce_validation_metrics
        # rank r saw 100 + r examples, loss sum 1.5 * (r + 1), score sum 10 * (r + 1); one rank holds more than 2^24 examples
        n_ex = 100 + rank + (2 ** 24 + 1 if rank == 0 else 0)
        out = reduce_validation_metrics(n_ex, 1.5 * (rank + 1), 10.0 * (rank + 1))
        acc = generative_accuracy(0.3 * (rank + 1), 10 * (rank + 1))
        q.put((rank, out, acc))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_validation_metric_sums_over_ranks(world):
    """The reference sums (n_ex, val_loss, tot_score) over the ranks before dividing (eval_utils.py:135-137); the torchmetrics states
    of VQAGenerativeAccuracy reduce by sum as well (:89-90).  Every rank gets the global sums, exactly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_metric_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(30)
    tri = world * (world + 1) / 2
    want = (sum(100 + r for r in range(world)) + 2 ** 24 + 1, 1.5 * tri, 10.0 * tri)
    for rank, out, acc in res:
        assert out[0] == want[0], (rank, out)          # beyond 2^24: would round in fp32
        assert abs(out[1] - want[1]) < 1e-9 and abs(out[2] - want[2]) < 1e-9
        assert abs(acc - (0.3 * tri) / (10 * tri)) < 1e-12


def test_validation_metric_reduction_is_the_identity_in_one_process():
    from mafed_amd.dist import generative_accuracy, reduce_validation_metrics
    assert reduce_validation_metrics(7, 2.5, 3.0) == (7.0, 2.5, 3.0)
    assert generative_accuracy(3.0, 10) == 0.3
