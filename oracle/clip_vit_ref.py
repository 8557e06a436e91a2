"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the frozen CLIP vision tower as the reference uses it.

The reference's CLIP branch (mafed/model/vl_pythia.py:196-198) is ``transformers.CLIPVisionModel``; the path calls it as
``vision_encoder(pixel_values, output_hidden_states=True).hidden_states[select_layer = -2][:, 1:]``
(``get_patch_embeddings`` + ``feature_select``, vl_pythia.py:453-475; frozen at mafed/model/vqa_cont_learner.py:202-203).
The arithmetic lives in the third-party module ``transformers/models/clip/modeling_clip.py`` (``clip:`` below; installed 5.15.0,
upstream pin 4.37.1 -- same arithmetic for the vision tower: Conv2d patch embedding without bias, class token, learned
position embedding, pre-LayerNorm, pre-LN encoder layers with biased q/k/v/out projections, scale D^-0.5, softmax in fp32,
quick-GELU MLP; hidden_states[i] = input of layer i, so index -2 is the OUTPUT of layer L-2: the last layer and
``post_layernorm`` never run on this path).

Plain PyTorch-CPU tensor ops, no ``transformers`` import: this file travels to the GPU box, the reference does not.
Pinned by ``tests/golden/clip_*.npz`` (made by ``oracle/gen_golden.py::gen_clip_fixture`` from ``CLIPVisionModel`` itself and
from the reference's own ``VLCLIPGPTNeoXForCausalLM.get_patch_embeddings``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class ClipVisionRefConfig:
    """Fields of ``CLIPVisionConfig`` the tower reads (openai/clip-vit-large-patch14: 1024 / 24 / 16 / 4096 / 224 / 14)."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    image_size: int = 224
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    select_layer: int = -2

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def layers_run(self) -> int:
        """hidden_states has L + 1 entries (embeddings after pre-LN, then each layer's output): index -2 needs L - 1 layers."""
        L = self.num_hidden_layers
        idx = self.select_layer if self.select_layer >= 0 else L + 1 + self.select_layer
        assert 0 <= idx <= L
        return idx


def param_shapes(cfg: ClipVisionRefConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """State-dict names / shapes of ``CLIPVisionModel`` (prefix ``vision_model.``; clip:138-159, 280-296, 338-344, 353-362)."""
    h, ff, ps, C = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels
    p = "vision_model."
    out = [(p + "embeddings.class_embedding", (h,)), (p + "embeddings.patch_embedding.weight", (h, C, ps, ps)),
           (p + "embeddings.position_embedding.weight", (cfg.num_patches + 1, h)),
           (p + "pre_layrnorm.weight", (h,)), (p + "pre_layrnorm.bias", (h,))]
    for i in range(cfg.num_hidden_layers):
        q = f"{p}encoder.layers.{i}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out += [(q + f"self_attn.{nm}.weight", (h, h)), (q + f"self_attn.{nm}.bias", (h,))]
        out += [(q + "layer_norm1.weight", (h,)), (q + "layer_norm1.bias", (h,)),
                (q + "mlp.fc1.weight", (ff, h)), (q + "mlp.fc1.bias", (ff,)), (q + "mlp.fc2.weight", (h, ff)), (q + "mlp.fc2.bias", (h,)),
                (q + "layer_norm2.weight", (h,)), (q + "layer_norm2.bias", (h,))]
    out += [(p + "post_layernorm.weight", (h,)), (p + "post_layernorm.bias", (h,))]
    return out


def init_weights(cfg: ClipVisionRefConfig, seed: int = 0, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """Deterministic weights owned by the build (numpy PCG64): every tensor non-trivial so that the fixtures pin every term."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in param_shapes(cfg):
        if "norm" in name and name.endswith("weight"):
            w = 1.0 + 0.05 * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith("bias"):
            w = 0.02 * rng.standard_normal(shape, dtype=np.float32)
        else:
            w = std * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return sd


def make_pixels(cfg: ClipVisionRefConfig, B: int, seed: int) -> torch.Tensor:
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal((B, cfg.num_channels, cfg.image_size, cfg.image_size), dtype=np.float32))


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """``hidden_act = "quick_gelu"`` of the OpenAI CLIP checkpoints (transformers/activations.py QuickGELUActivation)."""
    return x * torch.sigmoid(1.702 * x)


def embeddings(sd, pixels: torch.Tensor, cfg: ClipVisionRefConfig) -> torch.Tensor:
    """CLIPVisionEmbeddings.forward (clip:202-218): stride-``patch`` convolution = one GEMM over flattened patches."""
    p = "vision_model.embeddings."
    pe = F.conv2d(pixels, sd[p + "patch_embedding.weight"], bias=None, stride=cfg.patch_size)      # [B, h, g, g]
    pe = pe.flatten(2).transpose(1, 2)                                                              # [B, g*g, h]
    cls = sd[p + "class_embedding"].expand(pixels.shape[0], 1, -1)
    return torch.cat([cls, pe], dim=1) + sd[p + "position_embedding.weight"][None]


def attention(x: torch.Tensor, sd, pre: str, cfg: ClipVisionRefConfig) -> torch.Tensor:
    """CLIPAttention.forward + eager_attention_forward (clip:259-277, 298-335): bidirectional, no mask for images."""
    B, S, h = x.shape
    H, D = cfg.num_attention_heads, cfg.head_dim
    q = F.linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"]).view(B, S, H, D).transpose(1, 2)
    k = F.linear(x, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"]).view(B, S, H, D).transpose(1, 2)
    v = F.linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"]).view(B, S, H, D).transpose(1, 2)
    w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (D ** -0.5), dim=-1, dtype=torch.float32)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, S, h)
    return F.linear(o, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def encoder_layer(x: torch.Tensor, sd, i: int, cfg: ClipVisionRefConfig) -> torch.Tensor:
    """CLIPEncoderLayer.forward (clip:364-383): pre-LN, sequential residuals."""
    p = f"vision_model.encoder.layers.{i}."
    h = cfg.hidden_size
    y = F.layer_norm(x, (h,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.layer_norm_eps)
    x = x + attention(y, sd, p + "self_attn.", cfg)
    y = F.layer_norm(x, (h,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.layer_norm_eps)
    y = F.linear(quick_gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y


def hidden_states(sd, pixels: torch.Tensor, cfg: ClipVisionRefConfig, n_layers: int = None) -> List[torch.Tensor]:
    """CLIPVisionTransformer.forward with output_hidden_states (clip: embeddings -> pre_layrnorm -> encoder): entry 0 is the
    pre-LayerNormed embedding, entry i the output of layer i - 1."""
    x = embeddings(sd, pixels, cfg)
    x = F.layer_norm(x, (cfg.hidden_size,), sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"], cfg.layer_norm_eps)
    hs = [x]
    for i in range(cfg.num_hidden_layers if n_layers is None else n_layers):
        x = encoder_layer(x, sd, i, cfg)
        hs.append(x)
    return hs


def patch_features(sd, pixels: torch.Tensor, cfg: ClipVisionRefConfig) -> torch.Tensor:
    """get_patch_embeddings + feature_select("patch") (mafed/model/vl_pythia.py:453-475): hidden_states[select_layer] without
    the class token -> [B, num_patches, hidden]."""
    return hidden_states(sd, pixels, cfg, n_layers=cfg.layers_run)[-1][:, 1:]
