"""Experience replay (reference: mafed/methods/replay.py): a random memory subset per task, plain CE on a memory batch."""
from __future__ import annotations

import numpy as np
import torch

from mafed_amd.methods.base import CLStrategy
from mafed_amd.methods.memory import HBMReplayBuffer


class ER(CLStrategy):
    grads_only_through_model = True  # every parameter gradient of a step comes out of the model's own backward (Trainer: incremental clip norm)
    def __init__(self, opts, memory_size, model_type, **kwargs):
        super().__init__(opts=opts, **{k: v for k, v in kwargs.items() if k in ("reg_lambda", "mask", "scaler")})
        self.memory_size = memory_size
        self.memory_per_task = int(memory_size / (len(opts.tasks) - 1))
        self.datasets = []
        self.rng = np.random.default_rng(opts.seed)
        self.batch_size = opts.batch_size
        self.seed = 1
        self.model_type = model_type
        self.opts = opts
        self.mem_dataloader = None

    def update(self, dataset, model=None, **kwargs):
        self.task_id += 1
        n = len(dataset["input_ids"]) if isinstance(dataset, dict) else len(dataset)
        k = min(self.memory_per_task, n)
        idx = self.rng.choice(np.arange(n), k, replace=False)
        if isinstance(dataset, dict):
            sel = torch.as_tensor(np.sort(idx))
            samples = {key: dataset[key][sel] for key in HBMReplayBuffer.KEYS}
        else:
            items = [dataset[int(i)] for i in idx]
            samples = {key: torch.stack([torch.as_tensor(it[key]) for it in items]) for key in HBMReplayBuffer.KEYS}
        self.datasets.append(samples)
        if self.mem_dataloader is None:
            import torch.distributed as dist
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
            dev = model.flat_params.device if model is not None and hasattr(model, "flat_params") else samples["input_ids"].device
            self.mem_dataloader = HBMReplayBuffer(self.batch_size, dev, seed=self.opts.seed, rank=rank, world_size=world)
        self.mem_dataloader.add(samples)

    def update_mask(self, mask=None):
        self.mask = mask

    def compute_loss(self, model, loss, **kwargs):
        return loss

    def replay(self, model):
        batch = next(iter(self.mem_dataloader))
        n_ex = batch["input_ids"].size(0)
        loss = model(**batch, compute_loss=True, return_dict=True).loss
        return loss, n_ex
