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
        # {lr, 1-b1^t, sqrt(1-b2^t)} of the current step and the step counter live on the DEVICE (mafed_optim_advance):
        # the optimiser kernels carry no per-step host constants, so a whole step replays from a hipGraph
        self.lr_dev = torch.tensor([lr, 1.0, 1.0], dtype=torch.float32, device=dev)
        self.state_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._sched = (0, 0)  # (warmup_steps, total_steps); total 0 = constant lr
        self.clip_out = torch.ones(2, dtype=torch.float32, device=dev)  # {grad norm, clip scale}
        self.step_count = 0
        n_decay = model.decay_split()
        self.param_groups = [{"lr": lr, "initial_lr": lr, "weight_decay": weight_decay, "range": (0, n_decay)},
                             {"lr": lr, "initial_lr": lr, "weight_decay": 0.0, "range": (n_decay, model.flat_params.numel())}]

    def set_lr(self, lr: float) -> None:
        for g in self.param_groups:
            g["lr"] = lr

    def attach_schedule(self, warmup_steps: int, total_steps: int) -> None:
        """Linear warm-up / decay evaluated on the device each step (get_linear_schedule_with_warmup semantics)."""
        self._sched = (int(warmup_steps), int(total_steps))

    def advance(self) -> None:
        """First kernel of an optimiser step (capturable): t += 1 and {lr(t-1), 1-b1^t, sqrt(1-b2^t)} -> lr_dev.  A no-op when
        clip_grad_norm_(fuse_advance=True) has already done it for this step inside the norm's finish launch."""
        if getattr(self, "_advanced", False):
            self._advanced = False
            return
        # (with a clip pending the advance is guarded by it: a step whose gradient norm was not finite is skipped on the device --
        #  no parameter / state update in the AdamW kernel, no step-counter advance here; the host sees the non-finite norm in its log)
        ops.optim_advance_(self.state_dev, self.base_lr, self._sched[0], self._sched[1], self.betas[0], self.betas[1], self.lr_dev,
                           clip=self.clip_out if getattr(self, "_clip_pending", False) else None)

    def host_advance(self) -> None:
        """Host mirror of the step counter (logging, state_dict); no device work."""
        self.step_count += 1

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.model.zero_grad()

    # ---- global-norm clip ------------------------------------------------------------------------------------------------
    def _norm_plan(self):
        """{trigger id: [(lo, hi, first partial slot), ...]} over the flat gradient buffer, in the order the backward finishes the
        ranges (LM head = L, layers L-1 .. 0, embeddings / projector / every bias = -1), or None when the ranges do not tile the
        buffer exactly (then the norm is always the one-pass form)."""
        if not hasattr(self, "_norm_plan_cache"):
            plan = None
            try:
                from mafed_amd.dist import layer_ranges
                per_layer, head, tail = layer_ranges(self.model)
                L = len(per_layer)
                trig = {L: [head], -1: list(tail)}
                for i, r in enumerate(per_layer):
                    trig[i] = [r]
                flat = sorted(r for rs in trig.values() for r in rs)
                ok = flat[0][0] == 0 and flat[-1][1] == self.model.flat_grads.numel() and all(a[1] == b[0] for a, b in zip(flat, flat[1:]))
                ok = ok and all(lo % 4 == 0 for lo, _ in flat)
                if ok:
                    plan, slot = {}, 0
                    for t in [L] + list(range(L - 1, -1, -1)) + [-1]:
                        plan[t] = []
                        for lo, hi in trig[t]:
                            plan[t].append((lo, hi, slot))
                            slot += ops.gradnorm_blocks(hi - lo)
                    # behind the range partials: 16 slots per layer weight matrix (4 per layer) for the squares the weight-gradient GEMMs'
                    # epilogues leave (mafed_gemm_problem.sumsq) -- zero unless a backward uses them (begin_incremental_norm)
                    self._norm_dw_lo = slot
                    has_mat = hasattr(self.model, "layer_matrix_range") and all(
                        per_layer[i][0] <= self.model.layer_matrix_range(i)[0] and self.model.layer_matrix_range(i)[1] == per_layer[i][1] for i in range(L))
                    self._norm_dw_n = 16 * 4 * L if has_mat else 0
                    slot += self._norm_dw_n
                    self._norm_slots = slot
                    self._norm_partials = torch.zeros(slot, dtype=torch.float32, device=self.model.flat_grads.device)
            except Exception:
                plan = None
            self._norm_plan_cache = plan
        return self._norm_plan_cache

    def begin_incremental_norm(self, fused_matrix_squares: bool = False):
        """-> hook(i) for ``model.grad_ready_hook`` (or None): called by the backward on the stream that finished range i, it
        launches that range's sum-of-squares partials at once -- under the rest of the backward -- so that ``clip_grad_norm_`` only
        has the finish kernel left (the one-pass norm reads 1.6 GB on the optimiser step's critical path: 0.28 ms at 410M).
        ``fused_matrix_squares``: the grouped weight-gradient GEMMs of THIS backward leave the squares of the layers' matrix gradients in
        ``model.dw_sumsq`` slots (their epilogues hold the final values in registers); the hook then reads only a layer's LayerNorm
        weights -- 1.2 GB of the 1.63 GB pass gone.  The model clears ``dw_sumsq`` when a sweep could not use it (then the hook reads
        the whole range, as before)."""
        plan = self._norm_plan()
        self._norm_seen = {}
        if plan is None:
            return None
        g, part, model = self.model.flat_grads, self._norm_partials, self.model
        L = model.config.num_hidden_layers
        n_dw = getattr(self, "_norm_dw_n", 0)
        if n_dw:
            part[self._norm_dw_lo:].zero_()
        model.dw_sumsq = part[self._norm_dw_lo:self._norm_dw_lo + n_dw].view(L, 4, 16) if (fused_matrix_squares and n_dw) else None

        def hook(i):
            fused = model.dw_sumsq is not None and 0 <= i < L and getattr(model, "_dw_sumsq_used", None) == getattr(model, "_bw_serial", 0)
            for lo, hi, slot in plan.get(i, ()):
                if fused:
                    hi = model.layer_matrix_range(i)[0]   # the LayerNorm weights in front of the matrices; the matrices' squares are in dw_sumsq
                if hi > lo:
                    ops.gradnorm_partial(g[lo:hi], part[slot:])
            if self._norm_seen is None:     # a backward outside Trainer.step() while the hook is still installed (clip_grad_norm_ consumed the
                self._norm_seen = {}        # last window's record): its partials are simply never used
            self._norm_seen[i] = getattr(model, "_bw_serial", 0)   # which backward sweep this range's partials belong to
        hook.is_norm_hook = True   # Trainer replaces / removes hooks of this kind only
        return hook

    NORM_LOG_SLOTS = 4096

    def clip_grad_norm_(self, max_norm: float, fuse_advance: bool = False) -> torch.Tensor:
        """Global L2 norm + clip scale on the device; the scale is applied inside the AdamW kernel.  Uses the partials left by the
        backward's hooks when every range reported in this backward, the single pass otherwise.
        ``fuse_advance`` (Trainer): with the partials present, the schedule advance of this optimiser step runs in the finish launch
        (advance() then does nothing) and the returned norm is a view of a log slot of its own -- valid until NORM_LOG_SLOTS further
        optimiser steps have run -- instead of ``clip_out[0]``, which the next step overwrites (callers clone that one)."""
        plan = getattr(self, "_norm_plan_cache", None)
        seen = getattr(self, "_norm_seen", None)
        last = getattr(self.model, "_bw_serial", 0)
        # complete = every range reported during the LAST backward sweep (a sweep that skipped layers, or a plugin's extra sweep
        # before it, leaves older partials behind: then the buffer is read in one pass as before)
        if plan is not None and seen is not None and len(seen) == len(plan) and all(v == last for v in seen.values()):
            if fuse_advance:
                if not hasattr(self, "_norm_log"):
                    self._norm_log = torch.zeros(self.NORM_LOG_SLOTS, dtype=torch.float32, device=self.clip_out.device)
                    self._norm_log_i = 0
                slot = self._norm_log[self._norm_log_i % self.NORM_LOG_SLOTS: self._norm_log_i % self.NORM_LOG_SLOTS + 1]
                self._norm_log_i += 1
                ops.gradnorm_finish(self._norm_partials, max_norm, self.clip_out, norm_log=slot,
                                    advance=(self.state_dev, self.base_lr, self._sched[0], self._sched[1], self.betas[0], self.betas[1], self.lr_dev))
                self._advanced = True
                self._norm_seen = None
                self._clip_pending = True
                return slot[0]
            ops.gradnorm_finish(self._norm_partials, max_norm, self.clip_out)
        else:
            ops.gradnorm_clip(self.model.flat_grads, max_norm, self.clip_out)
        self._norm_seen = None
        self._clip_pending = True
        return self.clip_out[0]

    def step(self, grad_mul: float = 1.0) -> None:
        self.host_advance()
        self.advance()
        self.apply(grad_mul)

    def apply(self, grad_mul: float = 1.0, zero_grads: bool = False, skip_matrix_zero: bool = False) -> None:
        """Device half of a step: the AdamW kernels, reading this step's scalars from device memory.  ``zero_grads``: the same
        pass also zeroes the gradient buffer (optimizer.zero_grad() of the next window).  ``skip_matrix_zero`` (with ``zero_grads``): the
        layers' weight-matrix gradients are left as they are -- the next window's first backward overwrites them (model.grad_overwrite)."""
        m = self.model
        clip = self.clip_out if getattr(self, "_clip_pending", False) else None
        if zero_grads and skip_matrix_zero:
            self._apply_chunks(clip, grad_mul)
            self._clip_pending = False
            if m.flat_shadow is not None:
                m._shadow_dirty = False
            return
        for grp in self.param_groups:
            lo, hi = grp["range"]
            if hi <= lo:
                continue
            shadow = m.flat_shadow[lo:hi] if m.flat_shadow is not None else None
            ops.adamw_step_(m.flat_params[lo:hi], m.flat_grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr_dev,
                            self.betas[0], self.betas[1], self.eps, grp["weight_decay"], 0, clip, grad_mul, shadow, zero_grad=zero_grads)
        self._clip_pending = False
        if m.flat_shadow is not None:
            m._shadow_dirty = False

    def _apply_chunks(self, clip, grad_mul: float, skip_matrix_zero: bool = True, events: Optional[dict] = None, stream=None):
        """One AdamW launch per chunk of ``_chunks()`` on the current stream, the gradient zeroed in the same pass; a layer chunk zeroes only
        its LayerNorm-weight part when ``skip_matrix_zero`` (mafed_adamw_step_partial_zero)."""
        m = self.model
        for key, lo, hi, wd in self._chunks():
            shadow = m.flat_shadow[lo:hi] if m.flat_shadow is not None else None
            zn = None
            if skip_matrix_zero and isinstance(key, tuple) and key[0] == "layer":
                mlo, mhi = m.layer_matrix_range(key[1])
                assert lo <= mlo and mhi == hi, "layer chunk = [LayerNorm weights | weight matrices]"
                zn = mlo - lo
            ops.adamw_step_(m.flat_params[lo:hi], m.flat_grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr_dev,
                            self.betas[0], self.betas[1], self.eps, wd, 0, clip, grad_mul, shadow, zero_grad=True, zero_n=zn)
            if events is not None:
                events[key] = stream.record_event()
        if skip_matrix_zero:
            m._dw_stale = True   # (cleared by the next backward sweep: it overwrites the matrices, or zeroes them first)

    def _chunks(self):
        """(key, lo, hi, weight_decay) in the order the NEXT forward touches the parameters: everything the first kernels
        read (all biases / non-decayed tensors, token embedding, projector), then the layers bottom-up, then final LN + head."""
        from mafed_amd.dist import layer_ranges
        m = self.model
        wd = self.param_groups[0]["weight_decay"]
        per_layer, head, tail = layer_ranges(m)
        out = [("pre", tail[-1][0], tail[-1][1], 0.0)]               # the non-decayed segment
        out += [("pre", lo, hi, wd) for lo, hi in tail[:-1]]          # embed_in, projector
        out += [(("layer", i), lo, hi, wd) for i, (lo, hi) in enumerate(per_layer)]
        out.append(("head", head[0], head[1], wd))
        assert sum(hi - lo for _, lo, hi, _ in out) == m.flat_params.numel(), "optimizer chunks must tile the flat buffer"
        return [c for c in out if c[2] > c[1]]

    def apply_pipelined(self, stream, grad_mul: float = 1.0, zero_grads: bool = True, skip_matrix_zero: bool = False):
        """AdamW (and the gradient zeroing) chunk by chunk on ``stream``, one event per chunk group: the next forward waits
        for "pre", then for ("layer", i) right before layer i, then for "head" -- so the HBM-bound update of the upper layers
        runs under the MFMA-bound forward of the lower ones instead of in front of it.  The caller's stream must not touch
        parameters, optimiser state or gradients until it has waited for these events (the model's forward does)."""
        m = self.model
        clip = self.clip_out if getattr(self, "_clip_pending", False) else None
        main = torch.cuda.current_stream()
        stream.wait_event(main.record_event())  # gradients final, clip scale and {lr, bias corrections} on the device
        events = {}
        with torch.cuda.stream(stream):
            if zero_grads:
                self._apply_chunks(clip, grad_mul, skip_matrix_zero=skip_matrix_zero, events=events, stream=stream)
            else:
                for key, lo, hi, wd in self._chunks():
                    shadow = m.flat_shadow[lo:hi] if m.flat_shadow is not None else None
                    ops.adamw_step_(m.flat_params[lo:hi], m.flat_grads[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr_dev,
                                    self.betas[0], self.betas[1], self.eps, wd, 0, clip, grad_mul, shadow, zero_grad=False)
                    events[key] = stream.record_event()
        self._clip_pending = False
        if m.flat_shadow is not None:
            m._shadow_dirty = False
        return events

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "sched": self._sched}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.state_dev.fill_(self.step_count)
        self._sched = tuple(sd.get("sched", self._sched))


class LinearWarmupSchedule:
    """LambdaLR(get_linear_schedule_with_warmup) equivalent for FlatAdamW: lr is set at construction (epoch 0) and
    after every ``step()``."""

    def __init__(self, optimizer: FlatAdamW, warmup_steps: int, total_steps: int, last_epoch: int = -1):
        self.optimizer, self.warmup_steps, self.total_steps = optimizer, warmup_steps, total_steps
        self.last_epoch = last_epoch
        optimizer.attach_schedule(warmup_steps, total_steps)  # the device evaluates the same lambda from its own step counter
        self.step()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def step(self) -> None:
        """Host mirror (param_groups[...]["lr"], get_last_lr); the kernels read the device-side value."""
        self.last_epoch += 1
        self.optimizer.set_lr(self.optimizer.base_lr * lr_lambda(self.last_epoch, self.warmup_steps, self.total_steps))


def get_linear_schedule_with_warmup(optimizer: FlatAdamW, warmup_steps: int, total_steps: int, last_epoch: int = -1) -> LinearWarmupSchedule:
    return LinearWarmupSchedule(optimizer, warmup_steps, total_steps, last_epoch)
