"""Replay memory for the MI355X path.

The reference keeps ``Subset``s of the task dataset behind a DataLoader + PrefetchLoader and calls
``next(iter(loader))`` on every replay step (mafed/methods/distillation.py:85, 182-209) -- two worker processes are
re-spawned per call.  Here the memory holds pre-encoded samples (token ids, masks, labels and the frozen encoder's
patch features in bf16) resident in HBM: 0.5 MB per 256x1024 sample, 2 GB for the reference's 4000-sample budget
(mafed/train.py:414) out of 288 GB.  ``next(iter(buffer))`` keeps working, so the plugin code reads like upstream.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import torch


class HBMReplayBuffer:
    """Device-resident replay memory; iterating yields one random batch (fresh permutation's first batch, like
    ``RandomSampler`` behind ``next(iter(...))``).  Under torch.distributed each rank draws from its own shard."""

    KEYS = ("input_ids", "attention_mask", "labels", "patch_embeddings")

    def __init__(self, batch_size: int, device, seed: int = 0, rank: int = 0, world_size: int = 1,
                 feature_dtype: torch.dtype = torch.bfloat16):
        self.batch_size = batch_size
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self.feature_dtype = feature_dtype
        self.data: Dict[str, Optional[torch.Tensor]] = {k: None for k in self.KEYS}
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        # GPU: the next batch is gathered one draw ahead on a loader stream (a DataLoader's prefetch), so that consumers
        # that run beside the caller's stream -- the frozen-teacher forward -- can start on it before the caller's stream
        # has drained the previous step's optimiser.  ``last_ready_event`` marks the batch handed out last.
        self._next = None
        self._stream = None
        self.last_ready_event = None
        self.max_label_rows: Optional[int] = None
        self.attach_label_hint = True   # batches carry ``max_label_rows`` (row-sparse LM head in training)
        self.attach_index = False       # batches carry ``memory_index`` [B] (device int64): which stored samples they are (teacher cache)

    def __len__(self) -> int:
        t = self.data["input_ids"]
        return 0 if t is None else t.shape[0]

    def add(self, samples: Dict[str, torch.Tensor]) -> None:
        """Append already-collated samples (text tensors must share T, i.e. be left-padded to one length)."""
        for k in self.KEYS:
            v = samples[k].to(self.device)
            if k == "patch_embeddings":
                v = v.to(self.feature_dtype)
            cur = self.data[k]
            if cur is not None and k != "patch_embeddings" and cur.shape[1] != v.shape[1]:
                # left-pad the shorter side so that the buffer stays rectangular (pad id 0 / mask 0 / label -100)
                fill = -100 if k == "labels" else 0
                T = max(cur.shape[1], v.shape[1])
                cur = torch.nn.functional.pad(cur, (T - cur.shape[1], 0), value=fill)
                v = torch.nn.functional.pad(v, (T - v.shape[1], 0), value=fill)
            self.data[k] = v if cur is None else torch.cat([cur, v], dim=0)
        self._next = None  # a batch gathered ahead of time no longer reflects the buffer
        # largest number of labelled (shifted) positions of any stored sample: a host-side int that rides along with every batch
        # (``max_label_rows``) and lets the model run its LM head on the labelled rows only -- known here without touching the GPU
        # inside the step (add() runs between tasks)
        lab = self.data["labels"]
        self.max_label_rows = int((lab[:, 1:] != -100).sum(dim=1).max().item()) if lab is not None and lab.numel() else None

    def sample(self) -> Dict[str, torch.Tensor]:
        if self.device.type != "cuda":
            return self._draw()
        cur = torch.cuda.current_stream(self.device)
        if self._next is None:
            self._prefetch(cur)
        batch, ev = self._next
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)  # allocated on the loader stream, consumed here
        self.last_ready_event = ev
        self._prefetch(cur)
        return batch

    def _prefetch(self, cur) -> None:
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        self._stream.wait_stream(cur)  # the buffer contents (add) were written on the caller's stream
        with torch.cuda.stream(self._stream):
            batch = self._draw()
            self._next = (batch, self._stream.record_event())

    def shard(self):
        """[lo, hi): the stored samples this rank draws from."""
        n = len(self)
        return (n * self.rank) // self.world_size, (n * (self.rank + 1)) // self.world_size

    def _draw(self) -> Dict[str, torch.Tensor]:
        n = len(self)
        if n == 0:
            raise RuntimeError("replay memory is empty")
        lo, hi = (n * self.rank) // self.world_size, (n * (self.rank + 1)) // self.world_size
        if hi <= lo:
            # fewer samples than ranks: this rank's shard is empty (drawing index `lo` anyway would read another rank's sample, or past the
            # end of the buffer on the last rank)
            raise RuntimeError(f"replay memory: rank {self.rank} of {self.world_size} has an empty shard ({n} samples in the memory)")
        m = hi - lo
        idx = torch.randperm(m, generator=self.gen)[: self.batch_size] + lo
        if self.device.type == "cuda":
            # pinned + non_blocking: a pageable .to(device) synchronises the stream, i.e. the host could not start
            # enqueueing step n+1 before the GPU had finished step n (the host allocator holds the pinned block until
            # the copy has run)
            idx = idx.pin_memory().to(self.device, non_blocking=True)
        else:
            idx = idx.to(self.device)
        out = {k: v.index_select(0, idx) for k, v in self.data.items()}
        if self.attach_index:
            out["memory_index"] = idx
        if getattr(self, "max_label_rows", None) is not None and self.attach_label_hint:
            out["max_label_rows"] = self.max_label_rows
        return out

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        while True:
            yield self.sample()
