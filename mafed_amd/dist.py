"""Data-parallel gradient exchange (no reference counterpart: upstream DDP is "not tested", README.md:47).

One process per GPU; replicas hold the full student, teacher and optimiser state (410M: ~8 GB of 288 GB).  The only
data-path collective is the gradient mean, once per optimiser step: the flat fp32 gradient buffer is cut into
contiguous buckets that follow the order in which the hand-scheduled backward finishes them (LM head first, then
layers L-1 .. 0, then embeddings / projector / all biases); each bucket is all-reduced over RCCL on a side HIP stream
as soon as its last layer is done, overlapping the rest of backward -- as one all-reduce, or as reduce-scatter + all-gather
(all seven xGMI links of a GPU busy at once), in fp32 or through bf16 staging buffers.  Gradients are identical on every
rank afterwards, so the global-norm clip needs no further collective.  Teacher weights are broadcast once per task.
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
    """Bucketed, backward-overlapped gradient mean over ``flat_grads``.  Works on any object exposing ``flat_grads``,
    ``grad_ready_hook``, ``_offsets``, ``config.num_hidden_layers`` and ``decay_split()`` (the CPU/gloo tests drive it with a
    stand-in), so the bucket logic is testable without a GPU.

    ``mode``        "all_reduce": one ``all_reduce`` per bucket.
                    "reduce_scatter": ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` per bucket -- on the MI355X
                    node's point-to-point xGMI mesh (7 links x ~153 GB/s per GPU) a direct exchange keeps all seven links busy
                    (1/8 of the bucket per peer per phase), where a single ring is bound by one link per hop (SURVEY.md section 5:
                    ~2.7 ms vs ~18.6 ms for the 1.63 GB of fp32 gradients at 410M).
    ``grad_dtype``  None: buckets travel as fp32.  torch.bfloat16: each bucket is cast into a bf16 staging buffer, reduced in
                    bf16 and cast back (half the bytes on the links; the sum of <= 8 bf16 values per element).
    ``average``     "auto": ``ReduceOp.AVG`` on RCCL, ``SUM`` followed by a division elsewhere; True / False force one form
                    (both are covered by tests/test_dist_gloo.py: gloo implements AVG too)."""

    def __init__(self, model, process_group=None, bucket_mb: float = 64.0, grad_dtype: Optional[torch.dtype] = None,
                 mode: str = "all_reduce", average="auto"):
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"unknown reduce mode {mode!r}")
        if grad_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError(f"unsupported gradient bucket dtype {grad_dtype}")
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.mode = mode
        self.grad_dtype = torch.bfloat16 if grad_dtype == torch.bfloat16 else None
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
        if mode == "reduce_scatter":
            for _, (lo, hi) in self.buckets:  # every tensor starts on a 64-element boundary, so 2 / 4 / 8 ranks divide a bucket
                if (hi - lo) % max(1, self.world):
                    raise ValueError(f"bucket [{lo}, {hi}) is not divisible by the world size {self.world}")
        self._works = []
        self._staging = {}  # (lo, hi) -> bf16 staging buffer / fp32 shard buffer, allocated once
        self._use_cuda = model.flat_grads.is_cuda
        self._side = torch.cuda.Stream(device=model.flat_grads.device) if self._use_cuda else None
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._nccl = backend == "nccl"
        self._avg = self._nccl if average == "auto" else bool(average)
        self.bytes_per_step = 0  # payload handed to the collectives in the last window (diagnostics / bench line)
        self.force = False       # tests: issue the collectives even with a single rank (a one-rank RCCL group on a one-GPU box)
        # diagnostics (bench.py --gpus N): with ``time_wait`` every wait() brackets the compute stream's wait for the side stream with two
        # timing events; exposed_wait_ms() = how long the compute stream stood still for the gradient exchange (0 = fully hidden)
        self.time_wait = False
        self._wait_events = []
        model.grad_ready_hook = self._on_ready

    def covered(self) -> int:
        return sum(hi - lo for _, (lo, hi) in self.buckets)

    def _buf(self, key, n, dtype, device):
        b = self._staging.get(key)
        if b is None:
            b = self._staging[key] = torch.empty(n, dtype=dtype, device=device)
        return b

    def _cast(self, src, dst):
        if src.is_cuda:
            from mafed_amd import ops
            ops.cast(src, dst.dtype, out=dst)
        else:
            dst.copy_(src)

    def _reduce_bucket(self, lo: int, hi: int):
        """Issue the collective(s) of one bucket; returns (works, finish) where finish() runs once they are complete."""
        g32 = self.model.flat_grads[lo:hi]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.grad_dtype is not None:
            buf = self._buf(("lp", lo, hi), hi - lo, self.grad_dtype, g32.device)
            self._cast(g32, buf)
        else:
            buf = g32
        self.bytes_per_step += buf.numel() * buf.element_size()
        works = []
        if self.mode == "all_reduce":
            works.append(dist.all_reduce(buf, op=op, group=self.pg, async_op=True))
        else:
            n = (hi - lo) // self.world
            shard = self._buf(("shard", lo, hi), n, buf.dtype, buf.device)
            w = dist.reduce_scatter_tensor(shard, buf, op=op, group=self.pg, async_op=True)
            if not self._nccl:
                w.wait()  # RCCL runs a group's collectives in issue order on its own stream; gloo's worker threads do not
            works.append(w)
            works.append(dist.all_gather_into_tensor(buf, shard, group=self.pg, async_op=True))

        def finish():
            if self.grad_dtype is not None:
                self._cast(buf, g32)
            if not self._avg:
                g32.div_(self.world)
        return works, finish

    def _on_ready(self, trigger: int) -> None:
        if not self.enabled or (self.world == 1 and not self.force):
            return
        for lo, hi in self._by_trigger.get(trigger, ()):
            if hi <= lo:
                continue
            if self._use_cuda:
                self._side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._side):
                    self._works.append(self._reduce_bucket(lo, hi))
            else:
                self._works.append(self._reduce_bucket(lo, hi))

    def wait(self) -> None:
        """Call before clipping / the optimiser step: the compute stream waits for every bucket (and its cast back)."""
        if self._use_cuda and self._works:
            with torch.cuda.stream(self._side):
                for works, finish in self._works:
                    for w in works:
                        w.wait()
                    finish()
            cur = torch.cuda.current_stream()
            if self.time_wait:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self._side)
                e1.record(cur)
                self._wait_events.append((e0, e1))
            else:
                cur.wait_stream(self._side)
        else:
            for works, finish in self._works:
                for w in works:
                    w.wait()
                finish()
        self._works = []

    def begin_window(self) -> None:
        self.bytes_per_step = 0

    def exposed_wait_ms(self, reset: bool = True) -> List[float]:
        """Per timed wait(): milliseconds the compute stream was blocked behind the bucket collectives (synchronises the device)."""
        if not self._wait_events:
            return []
        torch.cuda.synchronize()
        out = [float(e0.elapsed_time(e1)) for e0, e1 in self._wait_events]
        if reset:
            self._wait_events = []
        return out


class EmulatedReducer(GradReducer):
    """ONE-GPU stand-in for the gradient exchange of an N-GPU run (``bench.py --emulate-collectives``; no multi-GPU node was ever available
    to this build): same buckets, same hooks, same side stream -- but instead of a collective each bucket launches
    ``mafed_tune_stream_for``: ``channels`` workgroups that stream memory for the time an all-reduce of that bucket would take when the
    whole 1.63 GB exchange takes ``allreduce_ms`` (~2.7 ms with all seven xGMI links busy, ~18.6 ms for a single
    ring).  Gradients are left untouched (one rank's mean is itself).  What this prices: the CUs and the memory traffic the collective's
    kernels take from the backward that runs beside them; what it cannot: link contention, rank skew, RCCL's own launch costs."""

    def __init__(self, model, allreduce_ms: float, channels: int = 16, world: int = 8, bucket_mb: float = 64.0, buffer_mb: int = 256):
        super().__init__(model, None, bucket_mb)
        self.world = int(world)          # (what Trainer looks at to decide how the backward beside the collectives runs)
        self.allreduce_ms, self.channels = float(allreduce_ms), int(channels)
        dev = model.flat_grads.device
        self._src = torch.empty(buffer_mb << 20, dtype=torch.uint8, device=dev)
        self._src.view(torch.float32).normal_()
        self._dst = torch.empty_like(self._src)
        self._total = float(sum(hi - lo for _, (lo, hi) in self.buckets))

    def _reduce_bucket(self, lo: int, hi: int):
        from mafed_amd import _lib
        us = self.allreduce_ms * 1e3 * (hi - lo) / self._total
        self.bytes_per_step += (hi - lo) * 4
        st = torch.cuda.current_stream()
        _lib.check(_lib.load().mafed_tune_stream_for(self._src.data_ptr(), self._dst.data_ptr(), self._src.numel(), self.channels, 1024, us,
                                                     st.cuda_stream), "tune_stream_for")
        return [], (lambda: None)


def broadcast_teacher(model, src: int = 0, process_group=None) -> None:
    """Once per task: every replica's frozen teacher := rank ``src``'s (1.63 GB at 410M)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(model.flat_params, src=src, group=process_group)
        model._shadow_dirty = True


def reduce_validation_metrics(n_ex, val_loss, tot_score, process_group=None, device=None):
    """Sum of the validation counters over the ranks, as the reference does at the end of its validation loops
    (mafed/utils/eval_utils.py:135-137: ``dist.all_reduce(torch.tensor([n_ex, val_loss, tot_score]))``; the torchmetrics states of
    ``VQAGenerativeAccuracy`` -- ``accuracy`` and ``total``, ``dist_reduce_fx="sum"``, eval_utils.py:89-90 -- reduce the same way).
    Every rank calls it with its own partial sums and gets the global ones back as Python numbers; a single process returns its input.
    One collective of three floats (fp64 on the wire: example counts beyond 2^24 stay exact)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) < 2:
        return float(n_ex), float(val_loss), float(tot_score)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(process_group) == "nccl" else torch.device("cpu")
    t = torch.tensor([float(n_ex), float(val_loss), float(tot_score)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    n, l, s = t.tolist()
    return n, l, s


def generative_accuracy(accuracy_sum, total, process_group=None, device=None) -> float:
    """``VQAGenerativeAccuracy.compute()`` under data parallelism (eval_utils.py:84-107): sum(accuracy) / sum(total) over the ranks."""
    n, _, s = reduce_validation_metrics(total, 0.0, accuracy_sum, process_group, device)
    return s / n if n > 0 else float("nan")


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
