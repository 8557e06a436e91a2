"""CL-method plugin protocol of the training path (reference: mafed/methods/base.py:1-57).

Same names, argument meaning and error behaviour as the reference so that its task-sequencing loop
(mafed/train.py:116-213) and step driver (mafed/model/vqa_cont_learner.py:209-254) can call these objects unchanged.
"""
from __future__ import annotations


class CLStrategy:
    """Base plugin: hooks are no-ops, ``compute_loss`` is abstract, ``replay`` returns ``(None, 0)``."""

    def __init__(self, reg_lambda=1.0, mask=None, scaler=None, **kwargs):
        self.task_id = 0
        self.reg_lambda = reg_lambda
        self.mask = mask
        self.scaler = scaler  # threaded through by the reference, never used
        opts = kwargs.get("opts")
        accum = getattr(opts, "accumulate_grad_batches", None) if opts is not None else None
        self.update_freq = accum if accum else 1

    # between tasks -------------------------------------------------------------------------------------------------
    def update(self, model, **kwargs):
        self.task_id += 1

    def update_after_new_task(self, **kwargs):
        return None

    # inside a step ---------------------------------------------------------------------------------------------------
    def update_after_backward(self, **kwargs):
        return None

    def update_after_step(self, **kwargs):
        return None

    def compute_loss(self, model, loss, **kwargs):
        raise NotImplementedError

    def replay(self, model, **kwargs):
        return None, 0

    def _is_batch_after_step(self, batch_idx=0):
        return (batch_idx + 1) % self.update_freq == 0


class Naive(CLStrategy):
    """Plain fine-tuning: the task loss is returned untouched (mafed/methods/base.py:50-57)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def compute_loss(self, model, loss, **kwargs):
        return loss
