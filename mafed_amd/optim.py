"""Optimiser side of the step (reference: mafed/optim/adamw.py, mafed/optim/sched.py,
mafed/model/vqa_cont_learner.py:58-128) on the model's flat parameter / gradient buffers.

``FlatAdamW`` = HF-style AdamW (eps added to sqrt(v) un-corrected, bias correction folded into the step size,
decoupled decay applied after the update with the scheduled lr) as ONE kernel launch per weight-decay segment, with the
global-norm clip scale (Lightning ``gradient_clip_val``, mafed/train.py:288) read from device memory -- the step never
synchronises with the host.  The bf16 shadow weights used by the MFMA GEMMs are written by the same kernel.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from mafed_amd import ops


def lr_lambda(current_step: int, warmup_steps: int, total_steps: int) -> float:
    """get_linear_schedule_with_warmup's multiplier (mafed/optim/sched.py:34-48)."""
    if current_step < warmup_steps:
        return float(current_step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - current_step) / float(max(1, total_steps - warmup_steps)))


def compute_warmup(n_batches: int, accumulate_grad_batches: int, warmup_perc: float, warmup_steps: Optional[int] = None) -> Tuple[int, int]:
    """BaseModule.compute_warmup (vqa_cont_learner.py:58-69): the horizon is ceil(len(dl)/accum) * 60 -- the 60 is
    hard-coded upstream -- and warm-up is ``warmup_perc`` of it unless ``warmup_steps`` is configured."""
    total = math.ceil(n_batches / accumulate_grad_batches) * 60
    return total, int(warmup_steps if warmup_steps is not None else warmup_perc * total)


class FlatAdamW:
    """AdamW over ``model.flat_params`` with the reference's two effective parameter groups:
    names without ``bias`` are decayed (LayerNorm weights included), names with ``bias`` are not.
    (The ``vqa_output`` lr_mul groups of configure_optimizers are empty for VLPythia.)"""

    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-6, weight_decay: float = 0.0,
                 correct_bias: bool = True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        if not correct_bias:
            raise NotImplementedError("correct_bias=False is not used on the MAFED path")
        self.model = model
        self.base_lr = lr
        self.betas, self.eps, self.weight_decay = tuple(betas), eps, weight_decay
        dev = model.flat_params.device
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self.lr_dev = torch.full((1,), lr, dtype=torch.float32, device=dev)
        self.clip_out = torch.ones(2, dtype=torch.float32, device=dev)  # {grad norm, clip scale}
        self.step_count = 0
        n_decay = model.decay_split()
        self.param_groups = [{"lr": lr, "initial_lr": lr, "weight_decay": weight_decay, "range": (0, n_decay)},
                             {"lr": lr, "initial_lr": lr, "weight_decay": 0.0, "range": (n_decay, model.flat_params.numel())}]

    def set_lr(self, lr: float) -> None:
        for g in self.param_groups:
            g["lr"] = lr
        self.lr_dev.fill_(lr)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.model.flat_grads.zero_()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Global L2 norm + clip scale on the device; the scale is applied inside the AdamW kernel."""
        ops.gradnorm_clip(self.model.flat_grads, max_norm, self.clip_out)
        self._clip_pending = True
        return self.clip_out[0]

    def step(self, grad_mul: float = 1.0) -> None:
        m = self.model
        self.step_count += 1
        clip = self.clip_out if getattr(self, "_clip_pending", False) else None
        for grp in self.param_groups:
            lo, hi = grp["range"]
            if hi <= lo:
                continue
            shadow = m.flat_shadow[lo:hi] if m.flat_shadow is not None else None
            ops.adamw_step_(m.flat_params[lo:hi], m.flat_grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr_dev,
                            self.betas[0], self.betas[1], self.eps, grp["weight_decay"], self.step_count, clip, grad_mul, shadow)
        self._clip_pending = False
        if m.flat_shadow is not None:
            m._shadow_dirty = False

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.set_lr(float(sd["lr"]))


class LinearWarmupSchedule:
    """LambdaLR(get_linear_schedule_with_warmup) equivalent for FlatAdamW: lr is set at construction (epoch 0) and
    after every ``step()``."""

    def __init__(self, optimizer: FlatAdamW, warmup_steps: int, total_steps: int, last_epoch: int = -1):
        self.optimizer, self.warmup_steps, self.total_steps = optimizer, warmup_steps, total_steps
        self.last_epoch = last_epoch
        self.step()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self) -> None:
        self.last_epoch += 1
        self.optimizer.set_lr(self.optimizer.base_lr * lr_lambda(self.last_epoch, self.warmup_steps, self.total_steps))


def get_linear_schedule_with_warmup(optimizer: FlatAdamW, warmup_steps: int, total_steps: int, last_epoch: int = -1) -> LinearWarmupSchedule:
    return LinearWarmupSchedule(optimizer, warmup_steps, total_steps, last_epoch)
