"""Step driver: ``Trainer.step()`` reproduces PyTorch-Lightning's automatic-optimisation order around
``VLPythiaVQACLearner.training_step`` (reference: mafed/model/vqa_cont_learner.py:209-254, mafed/train.py:284-301;
hook order of SURVEY.md section 8b(3)):

    training_step -> loss / accumulate_grad_batches -> backward ->
      [last micro-batch of the window: on_before_optimizer_step (cl_method.update_after_backward) ->
       clip_grad_norm(grad_norm) -> optimizer.step -> lr_scheduler.step -> zero_grad] ->
    on_train_batch_end (cl_method.update_after_step)

Nothing in a step synchronises with the host: loss / grad-norm come back as device tensors.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Any, Dict, Optional

import torch

from mafed_amd.dist import GradReducer
from mafed_amd.optim import FlatAdamW, compute_warmup, get_linear_schedule_with_warmup


class Trainer:
    def __init__(self, model, cl_method, config: Optional[Any] = None, task_id: int = 0, n_batches_per_epoch: int = 1000,
                 process_group=None, ddp: bool = False, bucket_mb: float = 64.0):
        cfg = config if config is not None else SimpleNamespace()
        self.config = cfg
        self.model = model
        self.cl_method = cl_method
        self.task_id = task_id
        self.accumulate = int(getattr(cfg, "accumulate_grad_batches", 1) or 1)
        self.replay_interval = int(getattr(cfg, "replay_interval", 4))
        self.grad_norm = float(getattr(cfg, "grad_norm", 2.0))
        optim_name = getattr(cfg, "optim", "adamw")
        if optim_name != "adamw":
            raise ValueError("invalid optimizer")  # vqa_cont_learner.py:111-112; only the scripts' AdamW is on the path
        self.optimizer = FlatAdamW(model, lr=float(getattr(cfg, "learning_rate", 5e-5)), betas=tuple(getattr(cfg, "betas", (0.9, 0.98))),
                                   weight_decay=float(getattr(cfg, "weight_decay", 0.01)))
        total, warm = compute_warmup(n_batches_per_epoch, self.accumulate, float(getattr(cfg, "warmup_perc", 0.1)),
                                     getattr(cfg, "warmup_steps", None))
        total = int(getattr(cfg, "total_steps", total))
        self.scheduler = get_linear_schedule_with_warmup(self.optimizer, warm, total, last_epoch=-1)
        self.reducer = GradReducer(model, process_group, bucket_mb) if ddp else None
        self.global_step = 0
        self.optimizer.zero_grad()
        self.on_train_start()

    # ---- Lightning hooks kept by name -----------------------------------------------------------------------------------
    def on_train_start(self):
        self.cl_method.num_training_steps = getattr(self.scheduler, "total_steps", None)

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int):
        """Replay / MAFED step iff task_id > 0 and (batch_idx+1) % replay_interval == 0 -- the current-task batch is
        then dropped; otherwise plain CE through cl_method.compute_loss."""
        loss = None
        branch = "task"
        if self.task_id > 0 and (batch_idx + 1) % self.replay_interval == 0:
            loss, _ = self.cl_method.replay(self.model)
            if loss is not None:
                branch = "replay"
        if loss is None:
            loss = self.model(**batch, compute_loss=True, return_dict=True).loss
            loss = self.cl_method.compute_loss(self.model, loss, batch=batch)
        return loss, branch

    def step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, Any]:
        window_end = (batch_idx + 1) % self.accumulate == 0
        if self.reducer is not None:
            self.reducer.enabled = window_end  # all-reduce only on the last micro-batch of an accumulation window
        loss, branch = self.training_step(batch, batch_idx)
        (loss / self.accumulate if self.accumulate != 1 else loss).backward()
        rec: Dict[str, Any] = {"loss": loss.detach(), "branch": branch, "stepped": False}
        if window_end:
            self.cl_method.update_after_backward(model=self.model)  # on_before_optimizer_step
            if self.reducer is not None:
                self.reducer.wait()
            rec["lr"] = self.optimizer.param_groups[0]["lr"]
            if self.grad_norm and self.grad_norm > 0:
                rec["grad_norm"] = self.optimizer.clip_grad_norm_(self.grad_norm).clone()
            self.optimizer.step()
            self.scheduler.step()
            self.optimizer.zero_grad()
            self.global_step += 1
            rec["stepped"] = True
        self.cl_method.update_after_step(model=self.model, batch_idx=batch_idx)  # on_train_batch_end
        return rec
