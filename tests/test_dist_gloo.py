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
