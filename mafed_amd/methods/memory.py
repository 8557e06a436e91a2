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

    def sample(self) -> Dict[str, torch.Tensor]:
        n = len(self)
        if n == 0:
            raise RuntimeError("replay memory is empty")
        lo, hi = (n * self.rank) // self.world_size, (n * (self.rank + 1)) // self.world_size
        m = max(1, hi - lo)
        idx = (torch.randperm(m, generator=self.gen)[: self.batch_size] + lo).to(self.device)
        return {k: v.index_select(0, idx) for k, v in self.data.items()}

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        while True:
            yield self.sample()
