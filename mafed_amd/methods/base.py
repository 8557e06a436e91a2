"""CL-method plugin protocol of the training path (reference: mafed/methods/base.py:1-57).

Same names, argument meaning and error behaviour as the reference so that its task-sequencing loop
(mafed/train.py:116-213) and step driver (mafed/model/vqa_cont_learner.py:209-254) can call these objects unchanged.

Call order seen by a plugin:

    between tasks   update(model=, dataset=, dataloader=, scaler=)          train.py:206-213
                    update_after_new_task(...)                              after the new task's learner exists
    every batch     replay(model) -> (loss | None, n_examples)              only on replay steps of task > 0
                    compute_loss(model, loss, batch=batch) -> loss          on every other step
                    [backward]
                    update_after_backward(model=)                           Lightning's on_before_optimizer_step
                    [clip, optimiser step, scheduler step]
                    update_after_step(model=, batch_idx=)                   Lightning's on_train_batch_end
"""
from __future__ import annotations

from typing import Any, Optional, Tuple


class CLStrategy:
    """Base plugin.  Every hook is a no-op, ``compute_loss`` is abstract and ``replay`` reports "nothing replayed"."""

    def __init__(self, reg_lambda: float = 1.0, mask: Any = None, scaler: Any = None, **kwargs):
        self.reg_lambda = reg_lambda      # weight of a regularisation term (EWC); unused by MAFED
        self.mask = mask
        self.scaler = scaler              # threaded through by the reference, never used
        self.task_id = 0                  # number of update() calls so far = index of the task being learnt
        accum = getattr(kwargs.get("opts"), "accumulate_grad_batches", None)
        self.update_freq = accum or 1     # optimiser steps happen every update_freq micro-batches

    # ---- between tasks -------------------------------------------------------------------------------------------------
    def update(self, model, **kwargs) -> None:
        self.task_id += 1

    def update_after_new_task(self, **kwargs) -> None:
        return None

    # ---- inside a step -------------------------------------------------------------------------------------------------
    def compute_loss(self, model, loss, **kwargs):
        raise NotImplementedError

    def replay(self, model, **kwargs) -> Tuple[Optional[Any], int]:
        return None, 0

    def update_after_backward(self, **kwargs) -> None:
        return None

    def update_after_step(self, **kwargs) -> None:
        return None

    def _is_batch_after_step(self, batch_idx: int = 0) -> bool:
        """True on the micro-batch that closes an accumulation window (the one followed by an optimiser step)."""
        return (batch_idx + 1) % self.update_freq == 0


class Naive(CLStrategy):
    """Plain fine-tuning: the task loss is returned untouched (mafed/methods/base.py:50-57)."""
    grads_only_through_model = True  # every parameter gradient of a step comes out of the model's own backward (Trainer: incremental clip norm)

    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def compute_loss(self, model, loss, **kwargs):
        return loss
