"""Online EWC on the flat parameter buffers (reference: mafed/methods/ewc.py).

Same constructor, ``update`` / ``compute_importances`` / ``compute_regularization`` / ``compute_loss`` surface and the same
arithmetic as the reference class.  For the native model the Fisher diagonal and the anchor weights are flat fp32 tensors
laid out like ``flat_params`` (``self.fisher[t][name]`` / ``self.old_params[t][name]`` are views into them, so the
reference's dict access keeps working) and the per-step penalty is two HBM-bound HIP passes (``mafed_ewc_penalty_fwd/bwd``)
instead of ~6 torch kernels per parameter tensor; a foreign ``nn.Module`` takes the reference's per-parameter loop.
"""
from __future__ import annotations

from typing import Dict

import torch

from mafed_amd import ops
from mafed_amd.methods.base import CLStrategy


class _FlatDict(dict):
    """name -> view into one flat tensor (``.flat``)."""

    def __init__(self, model, flat: torch.Tensor):
        super().__init__({name: flat[o:o + n].view(shape) for name, (o, n, shape) in model._offsets.items()})
        self.flat = flat


class _EwcPenaltyFn(torch.autograd.Function):
    """penalty = 0.5 * lambda * sum F (p - p*)^2; backward adds dL * lambda * F (p - p*) straight into ``flat_grads``."""

    @staticmethod
    def forward(ctx, anchor, model, old_flat, fisher_flat, reg_lambda):
        out = ops.ewc_penalty_fwd(model.flat_params, old_flat, fisher_flat, 0.5 * reg_lambda)
        ctx.model, ctx.old, ctx.fisher, ctx.lam = model, old_flat, fisher_flat, float(reg_lambda)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        m = ctx.model
        ops.ewc_penalty_bwd_(m.flat_params, ctx.old, ctx.fisher, ctx.lam, g.reshape(1).to(torch.float32).contiguous(), m.flat_grads)
        return None, None, None, None, None


class EWC(CLStrategy):
    """Online EWC regulariser (mafed/methods/ewc.py:17-127)."""

    def __init__(self, reg_lambda=1.0, online=True, online_factor=0.95, soft_targets=True, soft_targets_thres=0.1, **kwargs):
        super().__init__(reg_lambda, **{k: v for k, v in kwargs.items() if k in ("mask", "scaler", "opts")})
        self.fisher: Dict[int, Dict[str, torch.Tensor]] = {}
        self.old_params: Dict[int, Dict[str, torch.Tensor]] = {}
        self.online = online
        self.online_factor = online_factor
        self.soft_targets = soft_targets
        self.soft_targets_thres = soft_targets_thres

    # between tasks -------------------------------------------------------------------------------------------------
    def update(self, model, dataloader, **kwargs):
        wimportances = self.compute_importances(model, dataloader)
        native = hasattr(model, "flat_params")
        prev_w = _FlatDict(model, model.flat_params.detach().clone()) if native else {k: p.data.clone() for k, p in model.named_parameters()}
        if self.online:
            if self.task_id <= 1:
                self.fisher[0] = wimportances
            elif native:
                wimportances.flat.add_(self.fisher[0].flat, alpha=self.online_factor)  # new + factor * old (ewc.py:60-61)
                self.fisher[0] = wimportances
            else:
                prev = self.fisher[0]
                self.fisher[0] = {k: v.add(prev[k], alpha=self.online_factor) for k, v in wimportances.items()}
            self.old_params[0] = prev_w
        else:
            self.fisher[self.task_id] = wimportances
            self.old_params[self.task_id] = prev_w
        self.task_id += 1

    def compute_importances(self, model, dataloader):
        """Fisher diagonal: sum over batches of grad(batch_size * CE)^2 / number of samples (ewc.py:70-103)."""
        model.train()
        native = hasattr(model, "flat_params")
        if native:
            acc = torch.zeros_like(model.flat_params)
        else:  # foreign nn.Module: one multi-tensor op per batch over the trainable tensors
            names, params = zip(*[(k, p) for k, p in model.named_parameters() if p.requires_grad])
            acc = [torch.zeros_like(p) for p in params]
        seen = 0.0
        for batch in dataloader:
            model.zero_grad()
            n = batch["input_ids"].size(0)
            (n * model(**batch, compute_loss=True, return_dict=True).loss).backward()
            if native:
                acc.addcmul_(model.flat_grads, model.flat_grads)
            else:
                live = [(a, p.grad) for a, p in zip(acc, params) if p.grad is not None]
                torch._foreach_addcmul_([a for a, _ in live], [g for _, g in live], [g for _, g in live])
            seen += n
        model.zero_grad()
        if native:
            return _FlatDict(model, acc.div_(seen))
        torch._foreach_div_(acc, seen)
        return dict(zip(names, acc))

    # inside a step ---------------------------------------------------------------------------------------------------
    def compute_regularization(self, model, loss, task_id):
        fisher, old = self.fisher[task_id], self.old_params[task_id]
        if hasattr(model, "flat_params") and isinstance(fisher, _FlatDict) and model.flat_params.is_cuda:
            return loss + _EwcPenaltyFn.apply(model._anchor, model, old.flat, fisher.flat, float(self.reg_lambda))
        # foreign nn.Module: 0.5 * lambda * sum_k <F_k, (p_k - p*_k)^2> with multi-tensor ops (ewc.py:105-115)
        keys = [k for k, p in model.named_parameters() if p.requires_grad]
        cur = [dict(model.named_parameters())[k] for k in keys]
        delta = torch._foreach_sub(cur, [old[k] for k in keys])
        weighted = torch._foreach_mul(torch._foreach_mul(delta, delta), [fisher[k] for k in keys])
        return loss + 0.5 * self.reg_lambda * torch.stack([w.sum() for w in weighted]).sum()

    def compute_loss(self, model, loss, **kwargs):
        if self.task_id == 0:
            return loss
        if self.online:
            return self.compute_regularization(model, loss, 0)
        for t in range(self.task_id):
            loss = self.compute_regularization(model, loss, t)
        return loss
