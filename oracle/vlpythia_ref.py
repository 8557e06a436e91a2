"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the MAFED training hot path.

Plain PyTorch-CPU tensor ops (no ``transformers``, no reference import): this file travels to
the GPU box, the reference does not.  Every function cites the reference lines it restates;
paths are relative to /root/reference unless they name the third-party ``transformers``
GPT-NeoX module (``tf:`` = transformers/models/gpt_neox/modeling_gpt_neox.py, installed 5.15.0,
upstream pin 4.37.1 -- same arithmetic on this path).

Pinned by ``tests/golden/*.npz`` (made by ``oracle/gen_golden.py`` from the reference classes).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# configuration + deterministic weights
# ----------------------------------------------------------------------------------------------
@dataclass
class RefConfig:
    """Shape parameters (config/vlpythia-base.json:13-31)."""

    vocab_size: int = 50304
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    rotary_pct: float = 0.25
    rotary_emb_base: float = 10000.0
    layer_norm_eps: float = 1e-5
    vision_hidden_size: int = 1024
    num_vision_tokens: int = 256
    initializer_range: float = 0.02

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def rotary_ndims(self) -> int:  # tf:189-190
        return int(self.head_dim * self.rotary_pct)


PRESETS = {
    # name: (h, L, H)  -- SURVEY.md section 8 sizes
    "160m": (768, 12, 12),
    "410m": (1024, 24, 16),
    "1b": (2048, 16, 8),
    "1.4b": (2048, 24, 16),
}


def preset(name: str, **kw) -> RefConfig:
    h, L, H = PRESETS[name]
    return RefConfig(hidden_size=h, num_hidden_layers=L, num_attention_heads=H, intermediate_size=4 * h, **kw)


def param_shapes(cfg: RefConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Trainable state-dict contract, in reference registration order (mafed/model/vl_pythia.py:209-237)."""
    h, ff, V, dv = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.vision_hidden_size
    out: List[Tuple[str, Tuple[int, ...]]] = [("gpt_neox.embed_in.weight", (V, h))]
    for i in range(cfg.num_hidden_layers):
        p = f"gpt_neox.layers.{i}."
        out += [
            (p + "input_layernorm.weight", (h,)),
            (p + "input_layernorm.bias", (h,)),
            (p + "post_attention_layernorm.weight", (h,)),
            (p + "post_attention_layernorm.bias", (h,)),
            (p + "attention.query_key_value.weight", (3 * h, h)),
            (p + "attention.query_key_value.bias", (3 * h,)),
            (p + "attention.dense.weight", (h, h)),
            (p + "attention.dense.bias", (h,)),
            (p + "mlp.dense_h_to_4h.weight", (ff, h)),
            (p + "mlp.dense_h_to_4h.bias", (ff,)),
            (p + "mlp.dense_4h_to_h.weight", (h, ff)),
            (p + "mlp.dense_4h_to_h.bias", (h,)),
        ]
    out += [
        ("gpt_neox.final_layer_norm.weight", (h,)),
        ("gpt_neox.final_layer_norm.bias", (h,)),
        ("embed_out.weight", (V, h)),
        ("vision_embed_tokens.0.weight", (h, dv)),
        ("vision_embed_tokens.0.bias", (h,)),
        ("vision_embed_tokens.2.weight", (h, h)),
        ("vision_embed_tokens.2.bias", (h,)),
    ]
    return out


def init_weights(cfg: RefConfig, seed: int = 0, bias_std: Optional[float] = None,
                 ln_jitter: float = 0.0) -> Dict[str, torch.Tensor]:
    """Deterministic weights owned by the build (numpy PCG64 stream; identical here and on the GPU box).

    Linear/Embedding ~ N(0, initializer_range), LayerNorm (1, 0) -- the HF init distribution.
    ``bias_std`` / ``ln_jitter`` make biases and LN affine non-trivial so parity tests exercise them.
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg):
        is_ln = "layernorm" in name or "layer_norm" in name
        if is_ln and name.endswith("weight"):
            w = np.ones(shape, np.float32)
            if ln_jitter:
                w = w + ln_jitter * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith("bias"):
            std = (ln_jitter if is_ln else bias_std) or 0.0
            w = (std * rng.standard_normal(shape, dtype=np.float32)) if std else np.zeros(shape, np.float32)
        else:
            w = cfg.initializer_range * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return sd


def perturb(sd: Dict[str, torch.Tensor], seed: int, std: float = 1e-3) -> Dict[str, torch.Tensor]:
    """Teacher = student + N(0, std) (SURVEY.md section 8d synthetic inputs)."""
    rng = np.random.default_rng(seed)
    return {k: v + torch.from_numpy((std * rng.standard_normal(tuple(v.shape), dtype=np.float32))) for k, v in sd.items()}


def make_batch(cfg: RefConfig, B: int, T: int, seed: int = 1234, pad: bool = True, n_answer: int = 4,
               vocab_cap: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Synthetic batch in the layout of vlpythia_vqa_collate (mafed/data/vl_pythia_vqa_dataset.py:128-158):
    left-padded ``input_ids`` (pad id 0), ``attention_mask`` 0 on pads, ``labels`` -100 except the last
    ``n_answer`` valid positions; ``patch_embeddings`` stand in for the frozen encoder output [B,P,Dv].
    """
    rng = np.random.default_rng(seed)
    P, dv = cfg.num_vision_tokens, cfg.vision_hidden_size
    V = vocab_cap or cfg.vocab_size
    feats = rng.standard_normal((B, P, dv), dtype=np.float32)
    ids = rng.integers(1, V, size=(B, T), dtype=np.int64)
    if pad:
        valid = rng.integers(max(1, T // 4), T + 1, size=(B,))
        valid[0] = T  # keep one full-length row
    else:
        valid = np.full((B,), T)
    am = np.zeros((B, T), np.int64)
    labels = np.full((B, T), -100, np.int64)
    for b in range(B):
        v = int(valid[b])
        am[b, T - v:] = 1
        ids[b, : T - v] = 0
        na = min(n_answer, v)
        labels[b, T - na:] = ids[b, T - na:]
    return {
        "input_ids": torch.from_numpy(ids),
        "attention_mask": torch.from_numpy(am),
        "labels": torch.from_numpy(labels),
        "patch_embeddings": torch.from_numpy(feats),
    }


# ----------------------------------------------------------------------------------------------
# forward (mafed/model/vl_pythia.py:247-326 + tf GPTNeoXModel.forward)
# ----------------------------------------------------------------------------------------------
def rotary_tables(cfg: RefConfig, S: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [S, rotary_ndims]; position ids are arange(S), pads NOT skipped (tf:342-345, 89, 103-106)."""
    rd = cfg.rotary_ndims
    inv_freq = 1.0 / (cfg.rotary_emb_base ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))
    freqs = torch.arange(S, dtype=torch.float32)[:, None] * inv_freq[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:  # tf:111-115
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_partial_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x [B,H,S,D]; rotate the first rotary_ndims dims (tf:118-151)."""
    rd = cos.shape[-1]
    xr, xp = x[..., :rd], x[..., rd:]
    xr = xr * cos + _rotate_half(xr) * sin
    return torch.cat((xr, xp), dim=-1)


def additive_mask(attention_mask_full: torch.Tensor) -> torch.Tensor:
    """[B,1,S,S] additive mask = causal AND key-padding (tf:347-353).  No row is ever fully masked here
    because the image prefix is always valid."""
    B, S = attention_mask_full.shape
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))
    keep = causal[None, :, :] & attention_mask_full[:, None, :].bool()
    m = torch.zeros(B, S, S, dtype=torch.float32)
    m.masked_fill_(~keep, torch.finfo(torch.float32).min)
    return m[:, None]


def attention(x_ln: torch.Tensor, sd, prefix: str, cfg: RefConfig, mask: torch.Tensor,
              cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """GPTNeoXAttention (tf:180-236): fused QKV with per-head [H,{q,k,v},D] row interleave, partial rotary,
    scores*D^-0.5 + mask, fp32 softmax, PV, merge heads, dense."""
    B, S, h = x_ln.shape
    H, D = cfg.num_attention_heads, cfg.head_dim
    qkv = F.linear(x_ln, sd[prefix + "query_key_value.weight"], sd[prefix + "query_key_value.bias"])
    qkv = qkv.view(B, S, H, 3 * D).transpose(1, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    q = apply_partial_rotary(q, cos, sin)
    k = apply_partial_rotary(k, cos, sin)
    w = torch.matmul(q, k.transpose(2, 3)) * (D ** -0.5)
    w = w + mask.to(w.dtype)
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, h)
    return F.linear(o, sd[prefix + "dense.weight"], sd[prefix + "dense.bias"])


def layer(x: torch.Tensor, sd, i: int, cfg: RefConfig, mask, cos, sin) -> torch.Tensor:
    """GPTNeoXLayer with use_parallel_residual=True (tf:239-281): h + attn(LN1(h)) + mlp(LN2(h))."""
    p = f"gpt_neox.layers.{i}."
    h = cfg.hidden_size
    ln1 = F.layer_norm(x, (h,), sd[p + "input_layernorm.weight"], sd[p + "input_layernorm.bias"], cfg.layer_norm_eps)
    ln2 = F.layer_norm(x, (h,), sd[p + "post_attention_layernorm.weight"], sd[p + "post_attention_layernorm.bias"],
                       cfg.layer_norm_eps)
    attn = attention(ln1, sd, p + "attention.", cfg, mask, cos, sin)
    u = F.linear(ln2, sd[p + "mlp.dense_h_to_4h.weight"], sd[p + "mlp.dense_h_to_4h.bias"])
    mlp = F.linear(F.gelu(u), sd[p + "mlp.dense_4h_to_h.weight"], sd[p + "mlp.dense_4h_to_h.bias"])
    return mlp + attn + x


def masked_mean_loss(labels: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """compute_loss -> average_task_loss -> masked_mean (mafed/model/vl_pythia.py:86-96, 64-83, 44-61):
    take the last T positions, shift, CE(reduction=none, ignore -100), per-sample mean over valid tokens
    (count clamped at 1e-13), then batch mean."""
    T = labels.size(1)
    lg = logits[:, -T:, :]
    sl = lg[..., :-1, :].contiguous()
    tl = labels[..., 1:].contiguous()
    B, Tm1 = tl.shape
    ce = F.cross_entropy(sl.view(-1, sl.size(-1)).float(), tl.view(-1), reduction="none", ignore_index=-100).view(B, Tm1)
    m = tl != -100
    s = ce.masked_fill(~m, 0.0).sum(-1)
    return (s / m.sum(-1).float().clamp(min=1e-13)).mean()


@dataclass
class RefOutput:
    loss: Optional[torch.Tensor]
    logits: torch.Tensor
    hidden_states: Tuple[torch.Tensor, ...]


def forward(sd, batch, cfg: RefConfig, autocast_bf16: bool = False) -> RefOutput:
    """VLCLIPGPTNeoXForCausalLM.forward (mafed/model/vl_pythia.py:247-326) from pre-computed patch embeddings
    (the frozen encoder + feature_select, :453-475, is the path's input boundary)."""
    ctx = torch.autocast("cpu", dtype=torch.bfloat16) if autocast_bf16 else torch.autocast("cpu", enabled=False)
    with ctx:
        feats = batch["patch_embeddings"]
        img = F.linear(F.gelu(F.linear(feats, sd["vision_embed_tokens.0.weight"], sd["vision_embed_tokens.0.bias"])),
                       sd["vision_embed_tokens.2.weight"], sd["vision_embed_tokens.2.bias"])  # :226-234,270
        txt = F.embedding(batch["input_ids"], sd["gpt_neox.embed_in.weight"])  # :282
        x = torch.cat([img, txt.to(img.device)], dim=1)  # :283 (promotes to fp32 under autocast)
        B, S, _ = x.shape
        am = torch.cat([torch.ones(B, feats.shape[1], dtype=torch.long), batch["attention_mask"]], dim=1)  # :292
        mask = additive_mask(am)
        cos, sin = rotary_tables(cfg, S)
        hs = [x]
        for i in range(cfg.num_hidden_layers):
            x = layer(x, sd, i, cfg, mask, cos, sin)
            if i < cfg.num_hidden_layers - 1:
                hs.append(x)
        x = F.layer_norm(x, (cfg.hidden_size,), sd["gpt_neox.final_layer_norm.weight"],
                         sd["gpt_neox.final_layer_norm.bias"], cfg.layer_norm_eps)
        hs.append(x)  # index L is post-final-LN
        logits = F.linear(x, sd["embed_out.weight"])  # :310, all S positions
        loss = None
        if batch.get("labels") is not None:
            loss = masked_mean_loss(batch["labels"], logits)  # :314
    return RefOutput(loss=loss, logits=logits, hidden_states=tuple(hs))


# ----------------------------------------------------------------------------------------------
# MAFED distillation (mafed/methods/distillation.py, distillation_loss_weights.py)
# ----------------------------------------------------------------------------------------------
def layer_coeffs(strategy: str, num_hidden_layers: int, gamma: float,
                 distillation_layer: Optional[int]) -> Tuple[List[int], Optional[torch.Tensor]]:
    """DistillationWeights.__init__/prepare_layer_coeffs/get_distillation_layers
    (distillation_loss_weights.py:10-60, 81-89).  ``num_hidden_layers`` is what train.py:133 passes: L-1."""
    if distillation_layer is not None and not (0 <= distillation_layer < num_hidden_layers):
        distillation_layer = None  # distillation.py:61-64
    if distillation_layer is None and strategy == "single":
        raise AssertionError("Invalid layer weighting strategy 'single'. Use 'equal' or 'discounted' instead!")
    if distillation_layer is None and strategy == "cumulative":
        raise AssertionError("Invalid layer weighting strategy 'cumulative'. Please pass the distillation layer!")
    nh = distillation_layer if strategy == "cumulative" else num_hidden_layers
    if distillation_layer is not None and strategy != "cumulative":
        strategy = "single"
    if strategy == "single":
        return [distillation_layer], None
    if strategy == "equal":
        return list(range(nh)), torch.ones(nh) / nh
    c = torch.tensor([gamma ** d for d in torch.arange(nh, 0, -1)])
    return list(range(nh)), c / c.sum()


def modality_masks(attention_mask: torch.Tensor, P: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """feature_distillation mask construction (distillation.py:134-144)."""
    B, T = attention_mask.shape
    lang = torch.zeros(B, T + P, dtype=attention_mask.dtype)
    lang[:, P:] = attention_mask
    img = torch.zeros(B, T + P, dtype=attention_mask.dtype)
    img[:, :P] = 1
    return lang, img


def masked_mse(s: torch.Tensor, t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """_compute_mse_distillation_loss (distillation.py:237-249)."""
    dim = s.shape[-1]
    d = ((s.reshape(-1, dim) - t.reshape(-1, dim)) ** 2).sum(-1) / dim
    m = mask.reshape(-1)
    return (d * m).sum() / m.sum()


def masked_cos(s: torch.Tensor, t: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """_compute_cosine_distillation_loss (distillation.py:226-235); CosineEmbeddingLoss(target=1) = 1 - cos,
    with torch's eps=1e-8 guard on the norms' product (EPSILON in aten CosineEmbeddingLoss)."""
    dim = s.shape[-1]
    a, b = s.reshape(-1, dim), t.reshape(-1, dim)
    m = mask.reshape(-1)
    d = F.cosine_embedding_loss(a, b, torch.ones_like(m), reduction="none")
    return (d * m).sum() / m.sum()


def cls_cos(s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """_compute_cls_distillation_loss with the cosine loss (distillation.py:251-257); token 0 = first image patch."""
    a, b = s[:, 0], t[:, 0]
    return F.cosine_embedding_loss(a, b, torch.ones(a.shape[0]), reduction="none").mean()


def modality_weights(strategy: str, lang_mask, img_mask, layer: int, lang_coeff=None):
    """get_modality_loss_weights (distillation_loss_weights.py:71-79, 148-174)."""
    if strategy == "equal":
        nt, nv = lang_mask.sum(), img_mask.sum()
        return nt / (nt + nv), nv / (nt + nv)
    if strategy == "balanced":
        return 0.5, 0.5
    if strategy == "adaptive":
        lc = lang_coeff
        lw = lc.item() if lc.shape[0] == 1 else lc[layer].item()
        return lw, 1 - lw
    raise NotImplementedError


@dataclass
class DistillSpec:
    modality: str = "balanced"       # equal | balanced | adaptive
    layer_strategy: str = "discounted"  # single | equal | discounted | cumulative
    gamma: float = 0.5
    distillation_layer: Optional[int] = None
    loss: str = "mse"                 # mse | cosine
    cls: bool = False
    distillation_coeff: float = 1.0
    replay_coeff: float = 1.0
    lang_coeff: Optional[torch.Tensor] = None  # adaptive


def distill_loss(student_hs: Sequence[torch.Tensor], teacher_hs: Sequence[torch.Tensor], attention_mask: torch.Tensor,
                 cfg: RefConfig, spec: DistillSpec) -> Tuple[torch.Tensor, Dict[int, Dict[str, torch.Tensor]]]:
    """FeatureDistillation.distill + feature_distillation (distillation.py:105-166)."""
    layers, coeffs = layer_coeffs(spec.layer_strategy, cfg.num_hidden_layers - 1, spec.gamma, spec.distillation_layer)
    lang, img = modality_masks(attention_mask, cfg.num_vision_tokens)
    total = 0.0
    per_layer: Dict[int, Dict[str, torch.Tensor]] = {}
    fn = masked_cos if spec.loss == "cosine" else masked_mse
    for l in layers:
        c = 1.0 if coeffs is None else coeffs[l]
        s, t = student_hs[l], teacher_hs[l]
        if spec.cls:
            dl = cls_cos(s, t)
            per_layer[l] = {"loss": dl}
        else:
            lw, vw = modality_weights(spec.modality, lang, img, l, spec.lang_coeff)
            ll, vl = fn(s, t, lang), fn(s, t, img)
            dl = lw * ll + vw * vl
            per_layer[l] = {"lang": ll, "vision": vl, "lang_w": torch.as_tensor(lw), "vision_w": torch.as_tensor(vw),
                            "loss": dl}
        total = total + c * spec.distillation_coeff * dl
    return total, per_layer


def mafed_replay_loss(student_sd, teacher_sd, batch, cfg: RefConfig, spec: DistillSpec, task_id: int = 1,
                      autocast_bf16: bool = False):
    """FeatureDistillation.replay (distillation.py:84-103): replay CE (iff replay_coeff>0 and task_id>0) +
    distillation against the frozen teacher's hidden states (:218-224)."""
    out = forward(student_sd, batch, cfg, autocast_bf16)
    do_replay = spec.replay_coeff > 0 and task_id > 0
    loss = spec.replay_coeff * out.loss if do_replay else None
    if spec.distillation_coeff == 0:
        return loss, out, {}
    with torch.no_grad():
        tb = {k: v for k, v in batch.items() if k != "labels"}
        t_out = forward(teacher_sd, tb, cfg, autocast_bf16)
        t_hs = [x.detach() for x in t_out.hidden_states]
    dl, per_layer = distill_loss(out.hidden_states, t_hs, batch["attention_mask"], cfg, spec)
    loss = dl if loss is None else loss + dl
    return loss, out, per_layer


# ----------------------------------------------------------------------------------------------
# optimiser side (mafed/optim/adamw.py, mafed/optim/sched.py, mafed/model/vqa_cont_learner.py:58-128)
# ----------------------------------------------------------------------------------------------
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "vqa_output_distill_loss_params")


def param_group_of(name: str) -> int:
    """configure_optimizers grouping (vqa_cont_learner.py:71-102): groups 0/1 'vqa_output' (lr*lr_mul), 2/3 rest;
    odd = no weight decay.  GPT-NeoX LN params are '*layernorm.*' (lower-case) so only 'bias' names match."""
    top = "vqa_output" in name
    nd = any(x in name for x in NO_DECAY)
    return (0 if top else 2) + (1 if nd else 0)


def lr_lambda(step: int, warmup_steps: int, total_steps: int) -> float:
    """get_linear_schedule_with_warmup (optim/sched.py:34-48)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warmup_steps)))


def compute_warmup(n_batches: int, accumulate: int, warmup_perc: float) -> Tuple[int, int]:
    """BaseModule.compute_warmup (vqa_cont_learner.py:58-69); the 60 is hard-coded upstream."""
    total = math.ceil(n_batches / accumulate) * 60
    return total, int(warmup_perc * total)


def clip_grad_norm(grads: Sequence[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, float]:
    """Lightning gradient_clip_val -> torch.nn.utils.clip_grad_norm_ (train.py:288): global L2 norm,
    scale = clamp(max_norm / (norm + 1e-6), max=1)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float()) for g in grads]))
    scale = float(torch.clamp(max_norm / (total + 1e-6), max=1.0))
    return total, scale


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               beta1: float, beta2: float, eps: float, wd: float) -> None:
    """AdamW.step (optim/adamw.py:86-111), in place: eps added to sqrt(v) un-corrected, bias correction folded
    into the step size, decoupled decay applied AFTER the update with the scheduled lr."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if wd > 0.0:
        p.add_(p, alpha=-lr * wd)


# ---------------------------------------------------------------------------------------------------------------
# Greedy validation decode (mafed/model/vqa_cont_learner.py:260-267, utils/eval_utils.py:170-177)
# ---------------------------------------------------------------------------------------------------------------
def generate_greedy(sd, batch, cfg: RefConfig, max_new_tokens: int = 10, eos_token_id: Optional[int] = 0,
                    pad_token_id: Optional[int] = None, logits_fn=None):
    """``generate(max_new_tokens=10, use_cache=False, pad_token_id=eos)`` as HF 4.37.1 ``greedy_search`` runs it: every step
    pushes the whole [image | prompt | generated] sequence through the model again, takes the argmax of the last position,
    overwrites the tokens of finished rows with ``pad_token_id``, appends a 1 to the attention mask, and stops once every row
    has emitted ``eos_token_id``.  ``logits_fn(input_ids, attention_mask) -> [B, T', V]`` lets the golden generator run the
    same loop around the reference model's own forward.  Returns (tokens [B, T + n], last-position logits [n, B, V])."""
    if eos_token_id is not None and pad_token_id is None:
        pad_token_id = eos_token_id
    if logits_fn is None:
        def logits_fn(ids, am):
            b = {"input_ids": ids, "attention_mask": am, "patch_embeddings": batch["patch_embeddings"]}
            with torch.no_grad():
                return forward(sd, b, cfg).logits
    ids, am = batch["input_ids"].clone(), batch["attention_mask"].clone()
    T0 = ids.shape[1]
    unfinished = torch.ones(ids.shape[0], dtype=torch.int64)
    steps = []
    for _ in range(max_new_tokens):
        last = logits_fn(ids, am)[:, -1, :].float()
        steps.append(last)
        nxt = last.argmax(dim=-1)
        if eos_token_id is not None:
            nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        am = torch.cat([am, torch.ones_like(nxt)[:, None]], dim=1)
        if eos_token_id is not None:
            unfinished = unfinished * (nxt != eos_token_id).to(torch.int64)
            if int(unfinished.max()) == 0:
                break
    assert ids.shape[1] == T0 + len(steps)
    return ids, torch.stack(steps, dim=0)


# ---------------------------------------------------------------------------------------------------------------
# Online EWC (mafed/methods/ewc.py)
# ---------------------------------------------------------------------------------------------------------------
def ewc_importances(sd: Dict[str, torch.Tensor], batches: Sequence[Dict[str, torch.Tensor]], cfg: RefConfig,
                    autocast_bf16: bool = True) -> Dict[str, torch.Tensor]:
    """compute_importances (ewc.py:70-103): Fisher diagonal = sum over batches of grad(B * CE)^2, divided by the number of
    samples.  The reference runs the forward under bf16 autocast on whatever device it is on (ewc.py:84-86)."""
    imp = {k: torch.zeros_like(v) for k, v in sd.items()}
    total = 0.0
    for batch in batches:
        params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        bsz = batch["input_ids"].size(0)
        loss = bsz * forward(params, batch, cfg, autocast_bf16).loss
        loss.backward()
        for k in imp:
            if params[k].grad is not None:
                imp[k] += params[k].grad.detach().pow(2)
        total += bsz
    return {k: v / total for k, v in imp.items()}


def ewc_online_update(old: Optional[Dict[str, torch.Tensor]], new: Dict[str, torch.Tensor], task_id: int,
                      online_factor: float = 0.95) -> Dict[str, torch.Tensor]:
    """update (ewc.py:52-62), online=True: the first two updates overwrite, later ones decay the running sum."""
    if old is None or task_id <= 1:
        return new
    return {k: new[k] + online_factor * old[k] for k in new}


def ewc_penalty(params: Dict[str, torch.Tensor], old_params: Dict[str, torch.Tensor], fisher: Dict[str, torch.Tensor],
                reg_lambda: float) -> torch.Tensor:
    """compute_regularization (ewc.py:105-115): sum over parameters of 0.5 * lambda * sum(F * (p - p*)^2)."""
    loss = torch.zeros((), dtype=torch.float32)
    for k, p in params.items():
        loss = loss + 0.5 * reg_lambda * (fisher[k] * (p - old_params[k]).pow(2)).sum()
    return loss


@dataclass
class RefTrainer:
    """Reproduces the Lightning automatic-optimisation order around VLPythiaVQACLearner.training_step
    (vqa_cont_learner.py:209-254) on the CPU, fp32."""

    cfg: RefConfig
    sd: Dict[str, torch.Tensor]
    lr: float = 5e-5
    betas: Tuple[float, float] = (0.9, 0.98)
    eps: float = 1e-6
    weight_decay: float = 0.01
    grad_clip: float = 2.0
    accumulate: int = 1
    replay_interval: int = 4
    warmup_steps: int = 0
    total_steps: int = 1000
    task_id: int = 0
    teacher_sd: Optional[Dict[str, torch.Tensor]] = None
    spec: DistillSpec = field(default_factory=DistillSpec)
    autocast_bf16: bool = False

    def __post_init__(self):
        self.params = {k: v.clone().requires_grad_(True) for k, v in self.sd.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.sd.items()}
        self.opt_step = 0
        self.sched_step = 0
        self.log: List[dict] = []

    def step(self, batch, batch_idx: int, mem_batch=None) -> dict:
        """One micro-batch: training_step -> loss/accum -> backward -> [clip, AdamW, LambdaLR on the last
        micro-batch of the accumulation window]."""
        rec = {"batch_idx": batch_idx}
        loss = None
        if self.task_id > 0 and (batch_idx + 1) % self.replay_interval == 0 and mem_batch is not None:
            loss, _, per_layer = mafed_replay_loss(self.params, self.teacher_sd, mem_batch, self.cfg, self.spec,
                                                   self.task_id, self.autocast_bf16)
            rec["branch"] = "replay"
            rec["per_layer"] = {l: {k: float(x) for k, x in d.items()} for l, d in per_layer.items()}
        if loss is None:
            loss = forward(self.params, batch, self.cfg, self.autocast_bf16).loss
            rec["branch"] = "task"
        rec["loss"] = float(loss.detach())
        (loss / self.accumulate).backward()
        if (batch_idx + 1) % self.accumulate == 0:
            names = list(self.params)
            grads = [self.params[k].grad for k in names]
            total, scale = clip_grad_norm(grads, self.grad_clip)
            rec["grad_norm"] = float(total)
            lr = self.lr * lr_lambda(self.sched_step, self.warmup_steps, self.total_steps)
            rec["lr"] = lr
            self.opt_step += 1
            with torch.no_grad():
                for k in names:
                    grp = param_group_of(k)
                    wd = self.weight_decay if grp % 2 == 0 else 0.0
                    adamw_step(self.params[k], self.params[k].grad * scale, self.m[k], self.v[k], self.opt_step, lr,
                               self.betas[0], self.betas[1], self.eps, wd)
                    self.params[k].grad = None
            self.sched_step += 1
            rec["param_checksum"] = float(sum(p.detach().double().sum() for p in self.params.values()))
        self.log.append(rec)
        return rec
