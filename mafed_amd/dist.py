"""Data-parallel gradient exchange (no reference counterpart: upstream DDP is "not tested", README.md:47).

One process per GPU; replicas hold the full student, teacher and optimiser state (410M: ~8 GB of 288 GB).  The only
data-path collective is the gradient mean, once per optimiser step: the flat fp32 gradient buffer is cut into
contiguous buckets that follow the order in which the hand-scheduled backward finishes them (LM head first, then
layers L-1 .. 0, then embeddings / projector / all biases); each bucket is all-reduced over RCCL on a side HIP stream
as soon as its last layer is done, overlapping the rest of backward.  Gradients are identical on every rank afterwards,
so the global-norm clip needs no further collective.  Teacher weights are broadcast once per task.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def layer_ranges(model) -> Tuple[List[Tuple[int, int]], Tuple[int, int], List[Tuple[int, int]]]:
    """Contiguous flat ranges of the DECAYED segment: per-layer ranges, the head range (final LN weight + embed_out),
    and the tail ranges reduced last (embed_in, projector, the whole non-decayed segment)."""
    offs = model._offsets
    L = model.config.num_hidden_layers

    def span(names):
        lo = min(offs[n][0] for n in names)
        hi = max(offs[n][0] + (offs[n][1] + 63) // 64 * 64 for n in names)
        return lo, hi

    per_layer = []
    for i in range(L):
        pre = f"gpt_neox.layers.{i}."
        per_layer.append(span([n for n in offs if n.startswith(pre) and "bias" not in n]))
    head = span(["gpt_neox.final_layer_norm.weight", "embed_out.weight"])
    tail = [span(["gpt_neox.embed_in.weight"]), span(["vision_embed_tokens.0.weight", "vision_embed_tokens.2.weight"]),
            (model.decay_split(), model.flat_grads.numel())]
    return per_layer, head, tail


class GradReducer:
    """Bucketed, backward-overlapped all-reduce(mean) of ``flat_grads``.  Works on any object exposing
    ``flat_grads``, ``grad_ready_hook``, ``_offsets``, ``config.num_hidden_layers`` and ``decay_split()`` (the CPU/gloo
    tests drive it with a stand-in), so the bucket logic is testable without a GPU."""

    def __init__(self, model, process_group=None, bucket_mb: float = 64.0):
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.enabled = True
        per_layer, head, tail = layer_ranges(model)
        L = len(per_layer)
        cap = int(bucket_mb * (1 << 20) / 4)
        # buckets of consecutive layers, walking down from the last layer; trigger = lowest layer index of the bucket
        self.buckets: List[Tuple[int, Tuple[int, int]]] = [(L, head)]  # (trigger id, flat range)
        hi_layer = L - 1
        while hi_layer >= 0:
            lo_layer = hi_layer
            size = per_layer[hi_layer][1] - per_layer[hi_layer][0]
            while lo_layer - 1 >= 0 and size < cap:
                lo_layer -= 1
                size += per_layer[lo_layer][1] - per_layer[lo_layer][0]
            self.buckets.append((lo_layer, (per_layer[lo_layer][0], per_layer[hi_layer][1])))
            hi_layer = lo_layer - 1
        for r in tail:
            self.buckets.append((-1, r))
        self._by_trigger = {}
        for trig, rng in self.buckets:
            self._by_trigger.setdefault(trig, []).append(rng)
        self._works = []
        self._use_cuda = model.flat_grads.is_cuda
        self._side = torch.cuda.Stream(device=model.flat_grads.device) if self._use_cuda else None
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"
        model.grad_ready_hook = self._on_ready

    def covered(self) -> int:
        return sum(hi - lo for _, (lo, hi) in self.buckets)

    def _on_ready(self, trigger: int) -> None:
        if not self.enabled or self.world == 1:
            return
        for lo, hi in self._by_trigger.get(trigger, ()):
            if hi <= lo:
                continue
            buf = self.model.flat_grads[lo:hi]
            if self._use_cuda:
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    w = dist.all_reduce(buf, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._works.append((w, buf))

    def wait(self) -> None:
        """Call before clipping / the optimiser step: the compute stream waits for every bucket."""
        for w, buf in self._works:
            w.wait()
            if not self._avg:
                buf.div_(self.world)
        if self._use_cuda and self._works:
            torch.cuda.current_stream().wait_stream(self._side)
        self._works = []


def broadcast_teacher(model, src: int = 0, process_group=None) -> None:
    """Once per task: every replica's frozen teacher := rank ``src``'s (1.63 GB at 410M)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(model.flat_params, src=src, group=process_group)
        model._shadow_dirty = True


def init_from_env(backend: Optional[str] = None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* rendezvous as launched by torch.distributed.run."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # MAFED_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks on ONE GPU (RCCL needs a device per rank)
        backend = backend or os.environ.get("MAFED_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if backend != "nccl":
                local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
