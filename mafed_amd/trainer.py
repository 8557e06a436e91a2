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


import os as _os

_EARLY_TEACHER = _os.environ.get("MAFED_EARLY_TEACHER", "1") != "0"   # A/B switch


class Trainer:
    def __init__(self, model, cl_method, config: Optional[Any] = None, task_id: int = 0, n_batches_per_epoch: int = 1000,
                 process_group=None, ddp: bool = False, bucket_mb: float = 64.0, pipeline_optimizer: bool = False,
                 grad_dtype: Optional[torch.dtype] = None, reduce_mode: str = "all_reduce", incremental_norm: bool = True,
                 overwrite_weight_grads: bool = True, reducer=None):
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
        # (``reducer``: a ready-made GradReducer, e.g. dist.EmulatedReducer -- the one-GPU forecast of the N-GPU step)
        self.reducer = reducer if reducer is not None else (GradReducer(model, process_group, bucket_mb, grad_dtype=grad_dtype, mode=reduce_mode) if ddp else None)
        # AdamW + gradient zeroing chunk by chunk on their own stream, the next forward waiting per layer (FlatAdamW.
        # apply_pipelined).  Opt-in: between step() calls the caller's stream may then only reach the parameters through the
        # model's forward, or after join().
        self.pipeline_optimizer = bool(pipeline_optimizer) and torch.cuda.is_available()
        self._opt_stream = torch.cuda.Stream(device=model.flat_params.device) if self.pipeline_optimizer else None
        # global-norm clip from per-range partials launched by the backward as each range becomes final (single process only: under
        # DDP the gradient hook belongs to the reducer; plugins that touch gradients outside the model's backward keep the one-pass norm)
        self.incremental_norm = bool(incremental_norm)
        # DDP, the backward that runs beside the bucket collectives (last micro-batch of a window): "ticketed" = persistent GEMM kernels in
        # ticketed tile order, "128x128" = every GEMM of that backward on the 128 x 128 kernels (round 3), None = nothing special.
        # (model._engine_backward; per call -- MAFED_EPI_TICKETED / MAFED_EPI_NO_PERSISTENT --, no process-wide switch)
        self.contention_mode = "ticketed"
        # First micro-batch of a window WRITES the layers' weight-matrix gradients (beta = 0) and AdamW does not zero them: 1.2 GB less
        # written by the optimiser pass and 1.2 GB less read by the weight-gradient epilogues per step at 410M.  Single process, grouped
        # bf16 weight gradients only (`_overwrite_ok`); after a step the matrices' ``.grad`` holds the last gradient, not zeros.
        self.overwrite_weight_grads = bool(overwrite_weight_grads)
        # the clip's norm of the layers' weight matrices from the weight-gradient GEMMs' own epilogues (FlatAdamW.begin_incremental_norm)
        self.fused_norm_squares = True
        self.global_step = 0
        self._one = None
        self.optimizer.zero_grad()
        self.on_train_start()

    # ---- Lightning hooks kept by name -----------------------------------------------------------------------------------
    def on_train_start(self):
        self.cl_method.num_training_steps = getattr(self.scheduler, "total_steps", None)

    def _is_replay_step(self, batch_idx: int) -> bool:
        return self.task_id > 0 and (batch_idx + 1) % self.replay_interval == 0

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int):
        """Replay / MAFED step iff task_id > 0 and (batch_idx+1) % replay_interval == 0 -- the current-task batch is
        then dropped; otherwise plain CE through cl_method.compute_loss."""
        return self._training_step(batch, self._is_replay_step(batch_idx))

    def _training_step(self, batch, is_replay: bool):
        loss = None
        branch = "task"
        if is_replay:
            loss, _ = self.cl_method.replay(self.model)
            if loss is not None:
                branch = "replay"
        if loss is None:
            loss = self.model(**batch, compute_loss=True, return_dict=True).loss
            loss = self.cl_method.compute_loss(self.model, loss, batch=batch)
        return loss, branch

    @property
    def contention_aware(self) -> bool:   # (round-3 name: True = the 128 x 128 kernels beside collectives)
        return self.contention_mode == "128x128"

    @contention_aware.setter
    def contention_aware(self, v) -> None:
        self.contention_mode = "128x128" if v else None

    def _overwrite_ok(self) -> bool:
        m = self.model
        return (self.overwrite_weight_grads and self.reducer is None and getattr(m, "compute_dtype", None) == torch.bfloat16
                and int(getattr(m, "dw_group_layers", 0)) > 0 and hasattr(m, "layer_matrix_range") and m.flat_grads.is_cuda)

    def _device_step(self, batch, is_replay: bool, window_end: bool, window_start: bool = False) -> Dict[str, Any]:
        """Everything of a step that runs on the GPU, in Lightning's order (no host synchronisation).  Launches are eager: a
        hipGraph replay of this sequence measured slower than multi-stream eager launches on ROCm 7 (42.9 vs 39.7 ms at 410M),
        so the capture path of round 1 was removed rather than kept untested."""
        if self.reducer is not None:
            self.reducer.enabled = window_end  # the gradient mean runs only on the last micro-batch of an accumulation window
            if window_end:
                self.reducer.begin_window()
            # collectives run beside this backward: the model keeps its GEMMs off the one-block-per-CU persistent kernels meanwhile
            self.model.contended_backward = self.contention_mode if (window_end and self.reducer.world > 1 and self.contention_mode) else False
        inc_norm = (window_end and self.reducer is None and self.grad_norm and self.grad_norm > 0 and self.incremental_norm
                    and getattr(self.cl_method, "grads_only_through_model", False) and hasattr(self.model, "grad_ready_hook")
                    and self.model.flat_grads.is_cuda)
        if self.reducer is None and hasattr(self.model, "grad_ready_hook"):
            cur = self.model.grad_ready_hook
            if cur is None or getattr(cur, "is_norm_hook", False):   # (a hook installed by somebody else is left alone)
                self.model.grad_ready_hook = self.optimizer.begin_incremental_norm(fused_matrix_squares=self.fused_norm_squares) if inc_norm else None
                if not inc_norm and hasattr(self.model, "dw_sumsq"):
                    self.model.dw_sumsq = None
            else:
                self.optimizer._norm_seen = None
        ow = self._overwrite_ok()
        self.model.grad_overwrite = bool(ow and window_start)
        loss, branch = self._training_step(batch, is_replay)
        # (an explicit root gradient: autograd's implicit ones_like is a fill kernel per step on the chain between forward and backward)
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
            self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
        (loss / self.accumulate if self.accumulate != 1 else loss).backward(gradient=self._one)
        self.model.grad_overwrite = False
        if torch.cuda.is_available() and hasattr(self.cl_method, "_prefetch_teacher"):
            # lets the next step's frozen-teacher forward start here, under this step's clip + AdamW
            # (the model's own event marks the end of the dX chain: the teacher forward then starts under the parameter-gradient tail)
            ev = getattr(self.model, "dx_chain_event", None) if _EARLY_TEACHER else None
            self.cl_method.backward_done_event = ev if ev is not None else torch.cuda.current_stream().record_event()
            if ev is not None:
                self.model.dx_chain_event = None
        rec: Dict[str, Any] = {"loss": loss.detach(), "branch": branch, "stepped": False}
        if window_end:
            self.cl_method.update_after_backward(model=self.model)  # on_before_optimizer_step
            if self.reducer is not None:
                self.reducer.wait()
                if self.reducer.world > 1 and self.contention_mode == "128x128" and torch.cuda.is_available() and hasattr(self.cl_method, "_prefetch_teacher"):
                    # the next step's teacher forward (persistent GEMMs) starts behind the last bucket's collective, not beside it
                    self.cl_method.backward_done_event = torch.cuda.current_stream().record_event()
            if self.grad_norm and self.grad_norm > 0:
                gn = self.optimizer.clip_grad_norm_(self.grad_norm, fuse_advance=True)
                # (fused finish + advance leaves the norm in a log slot of its own; the one-pass form returns clip_out[0], overwritten next step)
                rec["grad_norm"] = gn if getattr(self.optimizer, "_advanced", False) else gn.clone()
            self.optimizer.advance()
            if self.pipeline_optimizer:
                self.model._param_events = self.optimizer.apply_pipelined(self._opt_stream, skip_matrix_zero=ow)
            else:
                self.optimizer.apply(zero_grads=True, skip_matrix_zero=ow)
            rec["stepped"] = True
        return rec

    def join(self) -> None:
        """Order the current stream behind a pipelined optimiser update (needed before parameters, gradients or optimiser
        state are read by anything but the model's forward)."""
        if self._opt_stream is not None:
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            self.model._param_events = None

    def step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, Any]:
        window_end = (batch_idx + 1) % self.accumulate == 0
        window_start = batch_idx % self.accumulate == 0
        is_replay = self._is_replay_step(batch_idx) and getattr(self.cl_method, "mem_dataloader", True) is not None
        lr_now = self.optimizer.param_groups[0]["lr"]
        rec = self._device_step(batch, is_replay, window_end, window_start)
        if window_end:
            rec["lr"] = lr_now
            self.optimizer.host_advance()
            self.scheduler.step()
            self.global_step += 1
        self.cl_method.update_after_step(model=self.model, batch_idx=batch_idx)  # on_train_batch_end
        return rec
