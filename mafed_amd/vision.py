"""Frozen CLIP vision tower on the gfx950 kernels.

Reference: ``build_vision_encoder`` -> ``CLIPVisionModel`` (mafed/model/vl_pythia.py:196-198), called by
``get_patch_embeddings`` / ``feature_select`` as ``vision_encoder(pixel_values, output_hidden_states=True)
.hidden_states[select_layer = -2][:, 1:]`` (vl_pythia.py:453-475) and frozen by the learner
(mafed/model/vqa_cont_learner.py:202-203).  The arithmetic is ``transformers/models/clip/modeling_clip.py`` (``clip:``).

What runs here: patch embedding as im2col + MFMA GEMM, class token + learned positions, pre-LayerNorm, then the first
``L + 1 + select_layer`` encoder layers (hidden_states[-2] is the output of layer L-2: the last layer and post_layernorm are
never needed on this path) -- LayerNorm kernel, ONE fused q|k|v GEMM per layer (the three biased projections of
clip:293-295 concatenated per head into the [H, {q,k,v}, D] layout the attention kernels read), bidirectional resident
attention (``mafed_attn_fwd_bidir``), output projection with the residual add in its epilogue, fc1 with the quick-GELU
epilogue, fc2 with the residual add.  Inference only, no autograd, no CPU path.

State-dict names are the checkpoint's (``vision_model.embeddings.class_embedding`` ... as in transformers 4.37.1 and the hub
files; names without the ``vision_model.`` prefix, as transformers 5.x registers them, load too).  EVA02 (the timm branch,
vl_pythia.py:181-193) needs weights and code that are not available offline and is not built.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn

from mafed_amd import ops
from mafed_amd._lib import EPI_QUICK_GELU


@dataclass
class ClipVisionConfig:
    """Fields of ``CLIPVisionConfig`` the tower reads (defaults: openai/clip-vit-large-patch14)."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    image_size: int = 224
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    select_layer: int = -2

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def layers_run(self) -> int:
        L = self.num_hidden_layers
        idx = self.select_layer if self.select_layer >= 0 else L + 1 + self.select_layer
        if not 0 <= idx <= L:
            raise ValueError(f"select_layer {self.select_layer} out of range for {L} layers")
        return idx

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "ClipVisionConfig":
        d = d.get("vision_config", d)
        keys = set(cls.__dataclass_fields__)
        return cls(**{k: v for k, v in d.items() if k in keys})


def _pad_rows(rows: int, widths) -> int:
    """Row padding of the activation matrices: the GEMM tiles are 128 or 144 rows tall and 512 of them are resident at a time
    (2 per CU), so pick the tile height whose tile counts over the layer's output widths need the fewest rounds of 512 --
    B = 32 images of 257 tokens: 8352 = 58 x 144 rows run the four GEMMs in 1 + 3 + 1 + 4 rounds, 8320 = 65 x 128 in 2 + 4 + 2 + 5."""
    best = None
    for m in (128, 144):
        rp = _pad_to(rows, m)
        rounds = sum(-(-((rp // m) * (w // 128)) // 512) * m for w in widths if w % 128 == 0)
        if best is None or (rounds, rp) < best[:2]:
            best = (rounds, rp)
    return best[1]


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class ClipVisionTower(nn.Module):
    PREFIX = "vision_model."

    def __init__(self, config: ClipVisionConfig, compute_dtype: torch.dtype = torch.bfloat16, device: Any = None, seed: Optional[int] = None):
        super().__init__()
        if config.hidden_act != "quick_gelu":
            raise ValueError(f"hidden_act {config.hidden_act!r}: only the OpenAI checkpoints' quick_gelu is built")
        assert compute_dtype in (torch.bfloat16, torch.float32)
        self.config, self.compute_dtype = config, compute_dtype
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        h, ff, ps, C, L = config.hidden_size, config.intermediate_size, config.patch_size, config.num_channels, config.num_hidden_layers
        g = torch.Generator().manual_seed(0 if seed is None else int(seed))

        def P(*shape, std=0.02, ones=False):
            t = torch.ones(shape) if ones else (torch.randn(shape, generator=g) * std if std else torch.zeros(shape))
            return nn.Parameter(t.to(dev), requires_grad=False)

        vm = nn.Module()
        emb = nn.Module()
        emb.class_embedding = P(h)
        emb.patch_embedding = nn.Module()
        emb.patch_embedding.weight = P(h, C, ps, ps)
        emb.position_embedding = nn.Module()
        emb.position_embedding.weight = P(config.num_patches + 1, h)
        vm.embeddings = emb
        vm.pre_layrnorm = nn.Module()      # (sic: the name the checkpoints carry)
        vm.pre_layrnorm.weight, vm.pre_layrnorm.bias = P(h, ones=True), P(h, std=0)
        enc = nn.Module()
        layers = nn.ModuleList()
        for _ in range(L):
            lyr = nn.Module()
            att = nn.Module()
            for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
                m = nn.Module()
                m.weight, m.bias = P(h, h), P(h, std=0)
                setattr(att, nm, m)
            lyr.self_attn = att
            lyr.layer_norm1 = nn.Module()
            lyr.layer_norm1.weight, lyr.layer_norm1.bias = P(h, ones=True), P(h, std=0)
            mlp = nn.Module()
            mlp.fc1, mlp.fc2 = nn.Module(), nn.Module()
            mlp.fc1.weight, mlp.fc1.bias = P(ff, h), P(ff, std=0)
            mlp.fc2.weight, mlp.fc2.bias = P(h, ff), P(h, std=0)
            lyr.mlp = mlp
            lyr.layer_norm2 = nn.Module()
            lyr.layer_norm2.weight, lyr.layer_norm2.bias = P(h, ones=True), P(h, std=0)
            layers.append(lyr)
        enc.layers = layers
        vm.encoder = enc
        vm.post_layernorm = nn.Module()
        vm.post_layernorm.weight, vm.post_layernorm.bias = P(h, ones=True), P(h, std=0)
        self.vision_model = vm
        self.num_features = h  # what the reference reads from a timm tower; CLIP towers expose config.hidden_size
        self._derived: Optional[Dict[str, Any]] = None
        self._bufs: Dict[Tuple, torch.Tensor] = {}
        self._register_load_state_dict_pre_hook(self._accept_unprefixed)
        self.register_load_state_dict_post_hook(lambda m, ik: setattr(m, "_derived", None))

    # ---- loading ---------------------------------------------------------------------------------------------------------
    def _accept_unprefixed(self, state_dict, prefix, *args):
        """transformers 5.x registers the tower's tensors without ``vision_model.``: accept both spellings."""
        own = prefix + self.PREFIX
        for k in [k for k in state_dict if k.startswith(prefix) and not k.startswith(own)]:
            tail = k[len(prefix):]
            if tail.split(".")[0] in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm"):
                state_dict[own + tail] = state_dict.pop(k)
        state_dict.pop(own + "embeddings.position_ids", None)  # non-persistent buffer in old checkpoints

    @classmethod
    def from_pretrained(cls, path: str, select_layer: int = -2, **kw) -> "ClipVisionTower":
        """A LOCAL directory with ``config.json`` (a CLIPVisionConfig, or a CLIPConfig with ``vision_config``) and
        ``model.safetensors`` / ``pytorch_model.bin``; hub names need network and are refused (vl_pythia.py:198)."""
        if not os.path.isdir(path):
            raise ValueError(f"{path!r} is not a local directory (no network access on this path)")
        with open(os.path.join(path, "config.json")) as fp:
            cfg = ClipVisionConfig.from_dict(json.load(fp))
        cfg.select_layer = select_layer
        tower = cls(cfg, **kw)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("vision_model.") or k.split(".")[0] in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm")}
        tower.load_state_dict(sd, strict=True)
        return tower

    def _apply(self, fn, recurse=True):
        self._derived = None
        self._bufs.clear()
        return super()._apply(fn, recurse)

    # ---- derived operands (compute-dtype weights in the layouts the kernels read) ----------------------------------------------
    def _prepare(self) -> Dict[str, Any]:
        if self._derived is not None:
            return self._derived
        cfg, cd = self.config, self.compute_dtype
        h, H, D = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
        vm = self.vision_model
        K = cfg.num_channels * cfg.patch_size ** 2
        Kpad = _pad_to(K, 64)
        with torch.no_grad():
            wp = torch.zeros((h, Kpad), dtype=cd, device=vm.embeddings.class_embedding.device)
            wp[:, :K] = vm.embeddings.patch_embedding.weight.reshape(h, K).to(cd)
            d: Dict[str, Any] = {"w_patch": wp, "Kpad": Kpad, "layers": []}
            for lyr in list(vm.encoder.layers)[: cfg.layers_run]:
                a = lyr.self_attn
                # rows (head, {q, k, v}, d): the fused-QKV layout of the attention kernels (one GEMM instead of three)
                w = torch.stack([a.q_proj.weight.view(H, D, h), a.k_proj.weight.view(H, D, h), a.v_proj.weight.view(H, D, h)], dim=1)
                b = torch.stack([a.q_proj.bias.view(H, D), a.k_proj.bias.view(H, D), a.v_proj.bias.view(H, D)], dim=1)
                d["layers"].append({
                    "w_qkv": w.reshape(3 * h, h).to(cd).contiguous(), "b_qkv": b.reshape(3 * h).float().contiguous(),
                    "w_out": a.out_proj.weight.to(cd).contiguous(), "w_fc1": lyr.mlp.fc1.weight.to(cd).contiguous(),
                    "w_fc2": lyr.mlp.fc2.weight.to(cd).contiguous()})
        self._derived = d
        return d

    def _buf(self, key, shape, dtype, device) -> torch.Tensor:
        """Zero-initialised scratch kept across calls (rows past B * S are padding for the 128-row GEMM tiles: they must hold
        finite values, nothing reads them back)."""
        k = (key, tuple(shape), dtype, torch.cuda.current_stream(device).cuda_stream)  # one set per stream: callers on two streams never share
        b = self._bufs.get(k)
        if b is None:
            b = self._bufs[k] = torch.zeros(shape, dtype=dtype, device=device)
        return b

    # ---- forward -----------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def hidden_state(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """hidden_states[select_layer] of ``CLIPVisionModel(pixel_values, output_hidden_states=True)``: [B, 1 + patches, h] fp32."""
        cfg, cd = self.config, self.compute_dtype
        vm = self.vision_model
        dev = vm.embeddings.class_embedding.device
        if dev.type != "cuda":
            raise RuntimeError("mafed_amd runs on the GPU only (no CPU fallback); move the tower with .cuda()")
        B, C, Hh, Ww = pixel_values.shape
        if C != cfg.num_channels or Hh != cfg.image_size or Ww != cfg.image_size:
            raise ValueError(f"Input image size ({Hh}*{Ww}) doesn't match model ({cfg.image_size}*{cfg.image_size}).")  # clip:204-207
        d = self._prepare()
        h, H, D, np_ = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.num_patches
        S = np_ + 1
        rows, prow = B * S, B * np_
        rows_pad, prow_pad = _pad_rows(rows, (h, 3 * h, cfg.intermediate_size)), _pad_to(prow, 128)
        pix = pixel_values.to(dev)
        if pix.dtype not in (torch.float32, torch.bfloat16):
            pix = pix.float()
        pix = pix.contiguous()
        cols = ops.patchify(pix, cfg.patch_size, prow_pad, d["Kpad"], cd)
        pe = ops.gemm(cols, d["w_patch"], False, True)                                        # [prow_pad, h]
        x = self._buf("x0", (rows_pad, h), torch.float32, dev)
        ops.vit_assemble(pe, vm.embeddings.class_embedding, vm.embeddings.position_embedding.weight, B, np_, h, x)
        x, _, _, _ = ops.layernorm_fwd(x, vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, None, None, cfg.layer_norm_eps, torch.float32, save_stats=False)
        ao = self._buf("ao", (rows_pad, h), cd, dev)
        for lyr, w in zip(vm.encoder.layers, d["layers"]):
            y, _, _, _ = ops.layernorm_fwd(x, lyr.layer_norm1.weight, lyr.layer_norm1.bias, None, None, cfg.layer_norm_eps, cd, save_stats=False)
            qkv = ops.gemm(y, w["w_qkv"], False, True, bias=w["b_qkv"])
            ops.attn_fwd_bidir(qkv, B, S, H, D, out=ao)
            x = ops.gemm(ao, w["w_out"], False, True, bias=lyr.self_attn.out_proj.bias, res2=x, out_dtype=torch.float32)
            y, _, _, _ = ops.layernorm_fwd(x, lyr.layer_norm2.weight, lyr.layer_norm2.bias, None, None, cfg.layer_norm_eps, cd, save_stats=False)
            a = ops.gemm(y, w["w_fc1"], False, True, bias=lyr.mlp.fc1.bias, epilogue=EPI_QUICK_GELU)
            x = ops.gemm(a, w["w_fc2"], False, True, bias=lyr.mlp.fc2.bias, res2=x, out_dtype=torch.float32)
        return x[:rows].view(B, S, h)

    def forward(self, pixel_values: torch.Tensor, output_hidden_states: bool = False, **kwargs) -> torch.Tensor:
        """[B, 1 + patches, h]: the tensor ``feature_select`` slices (vl_pythia.py:463-475)."""
        return self.hidden_state(pixel_values)

    forward_features = forward

    def patch_features(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """get_patch_embeddings + feature_select("patch"): [B, patches, h]."""
        return self.hidden_state(pixel_values)[:, 1:]
